// gfx950 kernels for the data-parallel part of the step BEFORE the carve path: the front half of the reference's
// extract_foreground_mask, background_subtraction.py:153-168.  What they replace:
//   k_bgr2hsv        cv2.cvtColor(image, cv2.COLOR_BGR2HSV) on uint8 (:155) -- OpenCV's 8-bit fixed-point path (RGB2HSV_b, hrange 180)
//   k_morph3x3       one pass of cv2.erode / cv2.dilate with the 3x3 MORPH_RECT element of the pre open / close (:161-168)
//   k_mog_apply      bg_model.apply of the MOG model assignment.py trains (:158; training background_subtraction.py:75-92)
// findContours / fill (:171-193, sequential border following) stays with cv2 on the CPU.
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

// cv2.bgsegm.createBackgroundSubtractorMOG(...).apply(image, None, learningRate) on an 8-bit 3-channel image
// (background_subtraction.py:75-92 training, :158 inference): the mixture-of-Gaussians model of KaewTraKulPong & Bowden as
// opencv_contrib's bgsegm module implements it (bgfg_gaussmix.cpp, process8uC3), one thread per pixel, float32, operations in
// that code's order (no contraction: the library is built with -ffp-contract=off).  Per pixel K <= 8 components
// {sortKey, weight, mean[3], var[3]} kept sorted by sortKey = weight / sqrt(sum var), best first.
//   alpha > 0: the first component within varThreshold of the pixel is pulled towards it and bubbles up; if none matches, the
//              last (or first empty) component is replaced by {w0, pixel, var0}; weights renormalised; the pixel is foreground
//              iff the component that took it lies behind the components that make up backgroundRatio of the weight.
//   alpha == 0 (the reference's inference, learning_rate 0): the model is only read.
// State in HBM as planes: plane (8 k + f) holds field f of component k for all pixels (f: 0 sortKey, 1 weight, 2..4 mean, 5..7 var),
// so that a wave's 64 pixels read 256 contiguous bytes per field.  PARITY UNPINNED like the rest of this file (oracle/mog_np.py).
struct MogParams {
    float alpha, T, vT, w0, sk0, var0, minVar;
    uint32_t K, npix;
};
constexpr int kMogMaxMixtures = 8;
constexpr float kMogEps = 1.1920928955078125e-7f;                  // FLT_EPSILON

__global__ __launch_bounds__(256) void k_mog_apply(const uint8_t *__restrict__ img, uint8_t *__restrict__ fg, float *__restrict__ state, const MogParams p)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= p.npix) return;
    const int K = (int)p.K;
    const float pix[3] = {(float)img[3 * i], (float)img[3 * i + 1], (float)img[3 * i + 2]};
    float sk[kMogMaxMixtures], w[kMogMaxMixtures], mu[kMogMaxMixtures][3], var[kMogMaxMixtures][3];
#pragma unroll
    for (int k = 0; k < kMogMaxMixtures; ++k) {
        if (k < K) {
            const float *f = state + (size_t)(8 * k) * p.npix + i;
            sk[k] = f[0]; w[k] = f[(size_t)p.npix];
#pragma unroll
            for (int c = 0; c < 3; ++c) { mu[k][c] = f[(size_t)(2 + c) * p.npix]; var[k][c] = f[(size_t)(5 + c) * p.npix]; }
        } else {
            sk[k] = w[k] = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) mu[k][c] = var[k][c] = 0.f;
        }
    }
    int kHit = -1, kFg = -1;
    if (p.alpha > 0.f) {
        float wsum = 0.f;
        int kstop = K;                                              // the reference's loop variable k where its loop ends
        bool done = false;
#pragma unroll
        for (int k = 0; k < kMogMaxMixtures; ++k) {
            if (k < K && !done) {
                const float wk = w[k];
                wsum += wk;
                if (wk < kMogEps) { done = true; kstop = k; }
                else {
                    const float d0 = pix[0] - mu[k][0], d1 = pix[1] - mu[k][1], d2c = pix[2] - mu[k][2];
                    const float dist2 = (d0 * d0 + d1 * d1) + d2c * d2c;
                    if (dist2 < p.vT * ((var[k][0] + var[k][1]) + var[k][2])) {
                        wsum -= wk;
                        const float dw = p.alpha * (1.f - wk);
                        w[k] = wk + dw;
                        const float d[3] = {d0, d1, d2c};
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            mu[k][c] = mu[k][c] + p.alpha * d[c];
                            const float nv = var[k][c] + p.alpha * (d[c] * d[c] - var[k][c]);
                            var[k][c] = nv > p.minVar ? nv : p.minVar;
                        }
                        sk[k] = wk / sqrtf((var[k][0] + var[k][1]) + var[k][2]);
                        int pos = k;
                        bool stop = false;
#pragma unroll
                        for (int k1 = k - 1; k1 >= 0; --k1) {
                            if (!stop) {
                                if (sk[k1] >= sk[k1 + 1]) stop = true;
                                else {
                                    float t;
                                    t = sk[k1]; sk[k1] = sk[k1 + 1]; sk[k1 + 1] = t;
                                    t = w[k1]; w[k1] = w[k1 + 1]; w[k1 + 1] = t;
#pragma unroll
                                    for (int c = 0; c < 3; ++c) {
                                        t = mu[k1][c]; mu[k1][c] = mu[k1 + 1][c]; mu[k1 + 1][c] = t;
                                        t = var[k1][c]; var[k1][c] = var[k1 + 1][c]; var[k1 + 1][c] = t;
                                    }
                                    pos = k1;
                                }
                            }
                        }
                        kHit = pos; done = true; kstop = k;
                    }
                }
            }
        }
        if (kHit < 0) {                                             // nothing matched: the weakest (or first empty) component starts over
            const int kk = kstop < K - 1 ? kstop : K - 1;
            kHit = kk;
#pragma unroll
            for (int k = 0; k < kMogMaxMixtures; ++k) {
                if (k == kk) {
                    wsum += p.w0 - w[k];
                    w[k] = p.w0; sk[k] = p.sk0;
#pragma unroll
                    for (int c = 0; c < 3; ++c) { mu[k][c] = pix[c]; var[k][c] = p.var0; }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < kMogMaxMixtures; ++k) if (k >= kstop && k < K) wsum += w[k];
        }
        const float wscale = 1.f / wsum;
        wsum = 0.f;
#pragma unroll
        for (int k = 0; k < kMogMaxMixtures; ++k) {
            if (k < K) {
                w[k] *= wscale;
                wsum += w[k];
                sk[k] *= wscale;
                if (wsum > p.T && kFg < 0) kFg = k + 1;
            }
        }
        fg[i] = kHit >= kFg ? 255 : 0;
#pragma unroll
        for (int k = 0; k < kMogMaxMixtures; ++k) {
            if (k < K) {
                float *f = state + (size_t)(8 * k) * p.npix + i;
                f[0] = sk[k]; f[(size_t)p.npix] = w[k];
#pragma unroll
                for (int c = 0; c < 3; ++c) { f[(size_t)(2 + c) * p.npix] = mu[k][c]; f[(size_t)(5 + c) * p.npix] = var[k][c]; }
            }
        }
    } else {
        bool done = false;
#pragma unroll
        for (int k = 0; k < kMogMaxMixtures; ++k) {
            if (k < K && !done) {
                if (w[k] < kMogEps) done = true;
                else {
                    const float d0 = pix[0] - mu[k][0], d1 = pix[1] - mu[k][1], d2c = pix[2] - mu[k][2];
                    const float dist2 = (d0 * d0 + d1 * d1) + d2c * d2c;
                    if (dist2 < p.vT * ((var[k][0] + var[k][1]) + var[k][2])) { kHit = k; done = true; }
                }
            }
        }
        if (kHit >= 0) {
            float wsum = 0.f;
#pragma unroll
            for (int k = 0; k < kMogMaxMixtures; ++k) {
                if (k < K && kFg < 0) {
                    wsum += w[k];
                    if (wsum > p.T) kFg = k + 1;
                }
            }
        }
        fg[i] = (kHit < 0 || kHit >= kFg) ? 255 : 0;
    }
}

}  // namespace vc
