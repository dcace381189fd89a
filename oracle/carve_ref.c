/* C/OpenMP restatement of the reference carve path.  TEST INFRASTRUCTURE ONLY.
 *
 * See oracle/__init__.py: PARITY UNPINNED vs cv2 (OpenCV absent, reference has
 * no golden vectors); this file is the second independent restatement and the
 * multi-core CPU baseline ("port") timed by bench.py.  It must agree bit-for-bit
 * with oracle/carve_np.py.  Build: oracle/Makefile (-O2 -ffp-contract=off, so a*b+c
 * stays two roundings as in numpy / an SSE2 OpenCV build).
 *
 * Reference lines followed (relative to /root/reference):
 *   voxel_reconstruction.py:52-57   grid axes (np.linspace) and voxel order
 *   voxel_reconstruction.py:81      cv2.projectPoints -> cvProjectPoints2Internal formula
 *   voxel_reconstruction.py:110-112 float bounds test, int() truncation, mask > 0
 *   assignment.py:119-133           all-views threshold, ascending order, colour camera
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* np.linspace(lo, hi, num=n): y[k] = k*step + lo (two roundings), y[n-1] = hi. */
void vco_axis(double lo, double hi, uint32_t n, double *out)
{
    if (n == 0) return;
    if (n == 1) { out[0] = lo; return; }
    double delta = hi - lo;
    double div = (double)(n - 1);
    double step = delta / div;
    for (uint32_t k = 0; k < n; ++k) {
        double kk = (double)k;
        double y = (step == 0.0) ? (kk / div) * delta : kk * step;
        out[k] = y + lo;
    }
    out[n - 1] = hi;
}

typedef struct {
    double R[9], t[3], fx, fy, cx, cy, k1, k2, p1, p2, k3;
} cam_t;

/* One voxel-view: OpenCV 4.x cvProjectPoints2Internal with a 5-term model.
 * Slots k4..k6, s1..s4 and the tilt are zero/identity in the reference's data and
 * are left out: they only change values that are already non-finite. */
static inline void project_one(const cam_t *c, double X, double Y, double Z, double *u, double *v)
{
    double x = c->R[0] * X + c->R[1] * Y + c->R[2] * Z + c->t[0];
    double y = c->R[3] * X + c->R[4] * Y + c->R[5] * Z + c->t[1];
    double z = c->R[6] * X + c->R[7] * Y + c->R[8] * Z + c->t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    double r2 = x * x + y * y;
    double r4 = r2 * r2;
    double r6 = r4 * r2;
    double a1 = 2 * x * y;
    double a2 = r2 + 2 * x * x;
    double a3 = r2 + 2 * y * y;
    double cdist = 1 + c->k1 * r2 + c->k2 * r4 + c->k3 * r6;
    double xd = x * cdist + c->p1 * a1 + c->p2 * a2;
    double yd = y * cdist + c->p1 * a3 + c->p2 * a1;
    *u = xd * c->fx + c->cx;
    *v = yd * c->fy + c->cy;
}

static inline int32_t pixel_offset(double u, double v, uint32_t H, uint32_t W)
{
    if (!(0 <= v && v < (double)H && 0 <= u && u < (double)W)) return -1;
    return (int32_t)((int64_t)v * (int64_t)W + (int64_t)u);
}

static void load_cams(cam_t *cams, uint32_t C, const double *K9, const double *dist5,
                      const double *R9, const double *t3)
{
    for (uint32_t c = 0; c < C; ++c) {
        memcpy(cams[c].R, R9 + 9 * c, sizeof(double) * 9);
        memcpy(cams[c].t, t3 + 3 * c, sizeof(double) * 3);
        cams[c].fx = K9[9 * c + 0]; cams[c].cx = K9[9 * c + 2];
        cams[c].fy = K9[9 * c + 4]; cams[c].cy = K9[9 * c + 5];
        cams[c].k1 = dist5[5 * c + 0]; cams[c].k2 = dist5[5 * c + 1];
        cams[c].p1 = dist5[5 * c + 2]; cams[c].p2 = dist5[5 * c + 3];
        cams[c].k3 = dist5[5 * c + 4];
    }
}

/* Project n arbitrary points for one camera (golden-vector checks). */
void vco_project(const double *pts, uint64_t n, const double *K9, const double *dist5,
                 const double *R9, const double *t3, double *uv)
{
    cam_t cam;
    load_cams(&cam, 1, K9, dist5, R9, t3);
    for (uint64_t i = 0; i < n; ++i)
        project_one(&cam, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], &uv[2 * i], &uv[2 * i + 1]);
}

/* Carve linear-index range [i0, i1) of an nx*ny*nz grid (i = iz*nx*ny + ix*ny + iy).
 * masks: u8 [C,H,W], foreground > 0.  frame: u8 [H,W,3] BGR of camera color_cam, or NULL.
 * Optional outputs (NULL to skip): viewmask u16 [i1-i0], lut i32 [C, i1-i0].
 * Survivors (views >= min_views and >= 1) are written ascending to idx_out / bgr_out up
 * to cap entries; the return value is the full survivor count, or -1 on bad arguments. */
int64_t vco_carve(uint32_t nx, uint32_t ny, uint32_t nz, const double *bounds6,
                  uint32_t C, const double *K9, const double *dist5, const double *R9, const double *t3,
                  uint32_t H, uint32_t W, const uint8_t *masks, const uint8_t *frame,
                  uint32_t min_views, uint32_t color_cam, uint64_t i0, uint64_t i1,
                  uint16_t *viewmask, int32_t *lut, uint32_t *idx_out, uint8_t *bgr_out, uint64_t cap,
                  int threads)
{
    uint64_t N = (uint64_t)nx * ny * nz;
    if (C == 0 || C > 16 || i1 > N || i0 > i1 || (i0 & 63)) return -1;
    if (min_views < 1) min_views = 1;
    uint64_t n = i1 - i0;
    cam_t cams[16];
    load_cams(cams, C, K9, dist5, R9, t3);
    double *xs = malloc(sizeof(double) * nx), *ys = malloc(sizeof(double) * ny), *zs = malloc(sizeof(double) * nz);
    vco_axis(bounds6[0], bounds6[1], nx, xs);
    vco_axis(bounds6[2], bounds6[3], ny, ys);
    vco_axis(bounds6[4], bounds6[5], nz, zs);
    uint64_t nwords = (n + 63) / 64;
    uint64_t *keep = calloc(nwords ? nwords : 1, sizeof(uint64_t));
    size_t HW = (size_t)H * W;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    (void)threads;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t w = 0; w < (int64_t)nwords; ++w) {
        uint64_t bits = 0;
        uint64_t base = i0 + (uint64_t)w * 64;
        for (uint32_t b = 0; b < 64 && base + b < i1; ++b) {
            uint64_t i = base + b;
            uint32_t iy = (uint32_t)(i % ny);
            uint64_t t = i / ny;
            uint32_t ix = (uint32_t)(t % nx), iz = (uint32_t)(t / nx);
            uint32_t vm = 0, cnt = 0;
            for (uint32_t c = 0; c < C; ++c) {
                double u, v;
                project_one(&cams[c], xs[ix], ys[iy], zs[iz], &u, &v);
                int32_t off = pixel_offset(u, v, H, W);
                if (lut) lut[(uint64_t)c * n + (i - i0)] = off;
                if (off >= 0 && masks[c * HW + (size_t)off] > 0) { vm |= 1u << c; ++cnt; }
            }
            if (viewmask) viewmask[i - i0] = (uint16_t)vm;
            if (cnt >= min_views) bits |= 1ull << b;
        }
        keep[w] = bits;
    }
    int64_t S = 0;
    for (uint64_t w = 0; w < nwords; ++w) {
        uint64_t bits = keep[w];
        while (bits) {
            uint32_t b = (uint32_t)__builtin_ctzll(bits);
            bits &= bits - 1;
            uint64_t i = i0 + w * 64 + b;
            if ((uint64_t)S < cap) {
                if (idx_out) idx_out[S] = (uint32_t)i;
                if (bgr_out) {
                    uint8_t px[3] = {0, 0, 0};
                    if (frame && color_cam < C) {
                        uint32_t iy = (uint32_t)(i % ny);
                        uint64_t t = i / ny;
                        uint32_t ix = (uint32_t)(t % nx), iz = (uint32_t)(t / nx);
                        double u, v;
                        project_one(&cams[color_cam], xs[ix], ys[iy], zs[iz], &u, &v);
                        int32_t off = pixel_offset(u, v, H, W);
                        if (off >= 0 && masks[color_cam * HW + (size_t)off] > 0)
                            memcpy(px, frame + 3 * (size_t)off, 3);
                    }
                    memcpy(bgr_out + 3 * S, px, 3);
                }
            }
            ++S;
        }
    }
    free(keep); free(xs); free(ys); free(zs);
    return S;
}

int vco_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
