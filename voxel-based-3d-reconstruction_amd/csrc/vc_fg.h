// gfx950 kernels for the data-parallel part of the step BEFORE the carve path: the front half of the reference's
// extract_foreground_mask, background_subtraction.py:153-168.  What they replace:
//   k_bgr2hsv        cv2.cvtColor(image, cv2.COLOR_BGR2HSV) on uint8 (:155) -- OpenCV's 8-bit fixed-point path (RGB2HSV_b, hrange 180)
//   k_morph3x3       one pass of cv2.erode / cv2.dilate with the 3x3 MORPH_RECT element of the pre open / close (:161-168)
// bg_model.apply (:158, a stateful mixture model) and findContours / fill (:171-193, sequential) stay with cv2 on the CPU.
// PARITY UNPINNED (no cv2 here, no intermediate image in the reference): checked against oracle/foreground_np.py.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vc {

constexpr int kHsvShift = 12;

// One thread per pixel: three bytes in, three bytes out.  Four pixels per thread as three dwords would save instructions; the
// images are 0.9 MB (644 x 486) to 6 MB (1080p), the kernel is launch-latency bound either way.
// sdiv / hdiv: OpenCV's two division tables (256 ints each, built on the host exactly as OpenCV builds them).
__global__ __launch_bounds__(256) void k_bgr2hsv(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ hsv, uint32_t npix,
                                                const int32_t *__restrict__ sdiv, const int32_t *__restrict__ hdiv)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix) return;
    const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
    int v = b > g ? b : g; v = v > r ? v : r;
    int vmin = b < g ? b : g; vmin = vmin < r ? vmin : r;
    const int diff = v - vmin;
    const int s = (diff * sdiv[v] + (1 << (kHsvShift - 1))) >> kHsvShift;
    int h = (v == r) ? (g - b) : (v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff);
    h = (h * hdiv[diff] + (1 << (kHsvShift - 1))) >> kHsvShift;       // (arithmetic shift of a negative int, as the reference's C++)
    h += h < 0 ? 180 : 0;
    hsv[3 * i] = (uint8_t)(h > 255 ? 255 : h);
    hsv[3 * i + 1] = (uint8_t)s;
    hsv[3 * i + 2] = (uint8_t)v;
}

// Anchor at the centre: output (y, x) looks at rows y-1..y+1, columns x-1..x+1; pixels outside the image never win
// (BORDER_CONSTANT with morphologyDefaultBorderValue).
template <bool DILATE>
__global__ __launch_bounds__(256) void k_morph3x3(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint32_t H, uint32_t W)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= H * W) return;
    const uint32_t y = i / W, x = i - y * W;
    uint32_t v = in[i];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = (int)y + dy, xx = (int)x + dx;
            if (yy < 0 || xx < 0 || yy >= (int)H || xx >= (int)W) continue;
            const uint32_t o = in[(uint32_t)yy * W + (uint32_t)xx];
            v = DILATE ? (o > v ? o : v) : (o < v ? o : v);
        }
    }
    out[i] = (uint8_t)v;
}

}  // namespace vc
