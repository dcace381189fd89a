// Device-side arithmetic shared by every voxcarve kernel (gfx950 only).
//
// Bit-exactness contract: float64 throughout, every multiply and add is its own
// rounding (contraction is switched off here AND on the hipcc command line), true IEEE
// division, and the expression order of OpenCV 4.x cvProjectPoints2Internal, which is
// what the reference calls at voxel_reconstruction.py:81.  The k4..k6 / s1..s4 / tilt
// slots of OpenCV's model are zero for the reference's 5-coefficient cameras and are
// left out: they only alter values that are already non-finite (rejected either way).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace vc {

struct CamDev {            // 21 doubles, lives in the kernel-argument segment (SGPR loads)
    double r[9];
    double t[3];
    double fx, fy, cx, cy;
    double k1, k2, p1, p2, k3;
};

// Camera-frame point -> pixel coordinates (everything after the rigid transform).
__device__ __forceinline__ void distort_and_project(const CamDev &c, double x, double y, double z,
                                                    double &u, double &v)
{
    z = (z != 0.0) ? 1.0 / z : 1.0;                 // z = z ? 1./z : 1  (no behind-camera cull)
    x = x * z;
    y = y * z;
    const double r2 = x * x + y * y;
    const double r4 = r2 * r2;
    const double r6 = r4 * r2;
    const double tx = 2 * x;                        // 2*x*y == (2*x)*y, 2*x*x == (2*x)*x
    const double ty = 2 * y;
    const double a1 = tx * y;
    const double a2 = r2 + tx * x;
    const double a3 = r2 + ty * y;
    const double cdist = 1 + c.k1 * r2 + c.k2 * r4 + c.k3 * r6;
    const double xd = x * cdist + c.p1 * a1 + c.p2 * a2;
    const double yd = y * cdist + c.p1 * a3 + c.p2 * a1;
    u = xd * c.fx + c.cx;
    v = yd * c.fy + c.cy;
}

__device__ __forceinline__ void project_point(const CamDev &c, double X, double Y, double Z,
                                              double &u, double &v)
{
    const double x = c.r[0] * X + c.r[1] * Y + c.r[2] * Z + c.t[0];
    const double y = c.r[3] * X + c.r[4] * Y + c.r[5] * Z + c.t[1];
    const double z = c.r[6] * X + c.r[7] * Y + c.r[8] * Z + c.t[2];
    distort_and_project(c, x, y, z, u, v);
}

// voxel_reconstruction.py:110-112: bounds test on the FLOAT coordinates (NaN and (-1,0)
// fail), then int() truncation.  Returns int(v)*W + int(u), or -1 when outside.
__device__ __forceinline__ int32_t pixel_offset(double u, double v, uint32_t H, uint32_t W)
{
    const bool inside = (u >= 0.0) && (u < (double)W) && (v >= 0.0) && (v < (double)H);
    if (!inside) return -1;
    return (int32_t)((uint32_t)(int32_t)v * W + (uint32_t)(int32_t)u);
}

// Foreground bit of pixel `off` in a bit-packed mask (bit b of word w = pixel 32*w + b).
__device__ __forceinline__ bool mask_bit(const uint32_t *__restrict__ bits, int32_t off)
{
    return (bits[(uint32_t)off >> 5] >> ((uint32_t)off & 31u)) & 1u;
}

}  // namespace vc
