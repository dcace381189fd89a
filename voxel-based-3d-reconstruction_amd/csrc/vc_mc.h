// Marching cubes over the dense ON/OFF volume on the device: the step after the carve path.
//
// Reference consumer: voxel_reconstruction.py:127-163 plot_marching_cubes -> skimage.measure.marching_cubes(voxels_status, 0)
// on the statuses in lookup-table order reshaped to (width, height*2, depth) (assignment.py:143-146).  The volume here is the
// carve's occupancy bit array itself (element i = voxel i), viewed as a C-ordered (d0, d1, d2) array -- with
// (d0, d1, d2) = (nx, ny, nz) exactly what that reshape yields, with (nz, nx, ny) the geometrically meaningful axes.
//
// Classic table-driven marching cubes (table: mc_table.h, generated; conventions: oracle/marching_np.py).  Vertices sit on
// the cube edges that join an ON and an OFF element, at off + level * (on - off); an edge belongs to its lower element.
// The unit of work is a WORD of 64 consecutive elements (one lane) and a GROUP of 64 words (one wave):
//
//   k_mc_count   per word: the crossed edges along each axis as three 64-bit masks (XOR of the bit stream with itself
//                shifted by 1, d2 and d1*d2 elements, volume borders masked out) and the triangle count of its cells;
//                per group: vertex and triangle totals (for the same two-level scan the record compaction uses).
//   k_mc_verts   per group: word bases by a wave scan; every crossed edge writes its vertex at
//                base + [axis-0 crossings of the word below it | all of axis 0 + axis-1 crossings below | ...].
//   k_mc_faces   per group: every cell writes its triangles; a triangle corner is an edge of some element of this or a
//                neighbouring word, whose vertex number = that word's base + rank inside its masks (stored by k_mc_count).
#pragma once
#include "mc_table.h"
#include "vc_kernels.h"

namespace vc {

struct McParams {
    const uint64_t *bits;       // [ceil(n / 64)] element i = bit (i & 63) of word i >> 6; bits past n are zero
    uint64_t n;                 // d0 * d1 * d2
    uint32_t d0, d1, d2;
    uint32_t nwords, ngroups;
    uint64_t *x;                // [3][nwords] crossing masks per axis
    uint32_t *wbase;            // [nwords] first vertex number of each word
    uint32_t *gv, *gt;          // [ngroups] vertices / triangles per group
    const uint32_t *gvoff, *gtoff;      // exclusive scans inside scan blocks ...
    const uint64_t *bvoff, *btoff;      // ... and of the scan blocks
    float *verts;               // [V][3]
    uint32_t *faces;            // [F][3]
    uint64_t vcap, fcap;
    float level;
};

// 64 elements starting at element index e (any alignment) of the bit stream
__device__ __forceinline__ uint64_t mc_window(const uint64_t *__restrict__ bits, uint64_t e, uint32_t nwords)
{
    const uint64_t w = e >> 6;
    const uint32_t o = (uint32_t)e & 63u;
    const uint64_t lo = w < nwords ? bits[w] : 0ull, hi = (w + 1 < nwords) ? bits[w + 1] : 0ull;
    return o ? (lo >> o) | (hi << (64u - o)) : lo;
}

// mask of the elements i in [e0, e0 + 64) whose coordinate along `axis` is NOT the last one (they own an edge along it)
__device__ __forceinline__ uint64_t mc_valid(uint64_t e0, uint32_t axis, uint32_t d0, uint32_t d1, uint32_t d2, uint64_t n)
{
    uint64_t m = 0;
    if (e0 >= n) return 0;
    const uint64_t cnt = n - e0 >= 64 ? 64 : n - e0;
    m = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
    if (axis == 2) {
        // clear the elements with c == d2 - 1
        uint64_t first = e0 - e0 % d2 + (d2 - 1);                  // the one of e0's row
        for (uint64_t i = first; i < e0 + cnt; i += d2)
            if (i >= e0) m &= ~(1ull << (i - e0));
    } else if (axis == 1) {
        // clear whole rows with b == d1 - 1
        for (uint64_t r = e0 / d2; r * d2 < e0 + cnt; ++r) {
            if (r % d1 != d1 - 1) continue;
            const uint64_t a = r * d2 > e0 ? r * d2 - e0 : 0, b = (r + 1) * d2 - e0 < cnt ? (r + 1) * d2 - e0 : cnt;
            const uint64_t span = (b - a == 64) ? ~0ull : (((1ull << (b - a)) - 1ull) << a);
            m &= ~span;
        }
    } else {
        const uint64_t lim = (uint64_t)(d0 - 1) * d1 * d2;         // elements of the last slab own no axis-0 edge
        if (e0 >= lim) m = 0;
        else if (lim - e0 < 64) m &= (1ull << (lim - e0)) - 1ull;
    }
    return m;
}

struct McWord {
    uint64_t b, s2, s1, s12, s0, s02, s01, s012;   // the eight corner streams of the word's cells
    uint64_t v0, v1, v2;                           // elements owning an edge along axis 0 / 1 / 2
};

__device__ __forceinline__ McWord mc_load(const McParams &p, uint32_t w)
{
    McWord m;
    const uint64_t e = (uint64_t)w << 6, s1 = p.d2, s0 = (uint64_t)p.d1 * p.d2;
    m.b = mc_window(p.bits, e, p.nwords);
    m.s2 = mc_window(p.bits, e + 1, p.nwords);
    m.s1 = mc_window(p.bits, e + s1, p.nwords);
    m.s12 = mc_window(p.bits, e + s1 + 1, p.nwords);
    m.s0 = mc_window(p.bits, e + s0, p.nwords);
    m.s02 = mc_window(p.bits, e + s0 + 1, p.nwords);
    m.s01 = mc_window(p.bits, e + s0 + s1, p.nwords);
    m.s012 = mc_window(p.bits, e + s0 + s1 + 1, p.nwords);
    m.v0 = mc_valid(e, 0, p.d0, p.d1, p.d2, p.n);
    m.v1 = mc_valid(e, 1, p.d0, p.d1, p.d2, p.n);
    m.v2 = mc_valid(e, 2, p.d0, p.d1, p.d2, p.n);
    return m;
}

__device__ __forceinline__ uint32_t mc_case(const McWord &m, uint32_t k)
{
    // corner number = a << 2 | b << 1 | c
    return (uint32_t)((m.b >> k) & 1ull) | (uint32_t)((m.s2 >> k) & 1ull) << 1 | (uint32_t)((m.s1 >> k) & 1ull) << 2 |
           (uint32_t)((m.s12 >> k) & 1ull) << 3 | (uint32_t)((m.s0 >> k) & 1ull) << 4 | (uint32_t)((m.s02 >> k) & 1ull) << 5 |
           (uint32_t)((m.s01 >> k) & 1ull) << 6 | (uint32_t)((m.s012 >> k) & 1ull) << 7;
}

__device__ __forceinline__ uint64_t mc_active(const McWord &m)
{
    const uint64_t differ = (m.b ^ m.s2) | (m.b ^ m.s1) | (m.b ^ m.s12) | (m.b ^ m.s0) | (m.b ^ m.s02) | (m.b ^ m.s01) | (m.b ^ m.s012);
    return differ & m.v0 & m.v1 & m.v2;
}

__global__ __launch_bounds__(kBlock) void k_mc_count(const McParams p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (g >= p.ngroups) return;
    const uint32_t w = g * 64 + lane;
    uint32_t nv = 0, nt = 0;
    if (w < p.nwords) {
        const McWord m = mc_load(p, w);
        const uint64_t x0 = (m.b ^ m.s0) & m.v0, x1 = (m.b ^ m.s1) & m.v1, x2 = (m.b ^ m.s2) & m.v2;
        p.x[w] = x0; p.x[(size_t)p.nwords + w] = x1; p.x[2 * (size_t)p.nwords + w] = x2;
        nv = (uint32_t)(__popcll(x0) + __popcll(x1) + __popcll(x2));
        for (uint64_t act = mc_active(m); act; act &= act - 1) nt += kMcNtri[mc_case(m, (uint32_t)__builtin_ctzll(act))];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { nv += __shfl_xor(nv, d); nt += __shfl_xor(nt, d); }
    if (lane == 0) { p.gv[g] = nv; p.gt[g] = nt; }
}

__global__ __launch_bounds__(kBlock) void k_mc_verts(const McParams p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (g >= p.ngroups) return;
    const uint32_t w = g * 64 + lane;
    uint64_t x[3] = {0, 0, 0};
    if (w < p.nwords) { x[0] = p.x[w]; x[1] = p.x[(size_t)p.nwords + w]; x[2] = p.x[2 * (size_t)p.nwords + w]; }
    const uint32_t c = (uint32_t)(__popcll(x[0]) + __popcll(x[1]) + __popcll(x[2]));
    const uint64_t base = p.bvoff[g / kScanBlock] + p.gvoff[g] + (wave_inclusive_scan(c, lane) - c);
    if (w >= p.nwords) return;
    p.wbase[w] = (uint32_t)base;
    if (c == 0) return;
    const uint64_t b = p.bits[w];
    const uint64_t s01 = (uint64_t)p.d1 * p.d2;
    uint64_t id = base;
    for (uint32_t axis = 0; axis < 3; ++axis)
        for (uint64_t m = x[axis]; m; m &= m - 1, ++id) {
            const uint32_t k = (uint32_t)__builtin_ctzll(m);
            const uint64_t e = ((uint64_t)w << 6) + k;
            float pos[3] = {(float)(e / s01), (float)((e / p.d2) % p.d1), (float)(e % p.d2)};
            pos[axis] += ((b >> k) & 1ull) ? 1.0f - p.level : p.level;      // off + level * (on - off), measured from the lower element
            if (id < p.vcap) { p.verts[3 * id] = pos[0]; p.verts[3 * id + 1] = pos[1]; p.verts[3 * id + 2] = pos[2]; }
        }
}

// vertex number of the edge along `axis` owned by element e
__device__ __forceinline__ uint32_t mc_vertex(const McParams &p, uint64_t e, uint32_t axis)
{
    const uint32_t w = (uint32_t)(e >> 6), k = (uint32_t)e & 63u;
    const uint64_t below = (1ull << k) - 1ull;
    uint32_t id = p.wbase[w];
    const uint64_t x0 = p.x[w];
    if (axis == 0) return id + (uint32_t)__popcll(x0 & below);
    id += (uint32_t)__popcll(x0);
    const uint64_t x1 = p.x[(size_t)p.nwords + w];
    if (axis == 1) return id + (uint32_t)__popcll(x1 & below);
    return id + (uint32_t)__popcll(x1) + (uint32_t)__popcll(p.x[2 * (size_t)p.nwords + w] & below);
}

__global__ __launch_bounds__(kBlock) void k_mc_faces(const McParams p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (g >= p.ngroups) return;
    if (p.gt[g] == 0) return;
    const uint32_t w = g * 64 + lane;
    McWord m;
    uint64_t act = 0;
    uint32_t nt = 0;
    if (w < p.nwords) {
        m = mc_load(p, w);
        act = mc_active(m);
        for (uint64_t a = act; a; a &= a - 1) nt += kMcNtri[mc_case(m, (uint32_t)__builtin_ctzll(a))];
    }
    uint64_t t = p.btoff[g / kScanBlock] + p.gtoff[g] + (wave_inclusive_scan(nt, lane) - nt);
    const uint64_t s1 = p.d2, s0 = (uint64_t)p.d1 * p.d2;
    for (; act; act &= act - 1) {
        const uint32_t k = (uint32_t)__builtin_ctzll(act);
        const uint32_t cs = mc_case(m, k);
        const uint64_t e = ((uint64_t)w << 6) + k;
        for (uint32_t q = 0; q < kMcNtri[cs]; ++q, ++t) {
            uint32_t f[3];
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) {
                const uint32_t ed = kMcTri[cs][3 * q + c3], axis = ed >> 2, mm = ed & 3u;
                // the edge's lower corner: the other two axes' offsets are the bits of mm (high bit = lower-numbered axis)
                const uint32_t oa = axis == 0 ? 0u : (mm >> 1), ob = axis == 1 ? 0u : (axis == 0 ? (mm >> 1) : (mm & 1u));
                const uint32_t oc = axis == 2 ? 0u : (mm & 1u);
                f[c3] = mc_vertex(p, e + oa * s0 + ob * s1 + oc, axis);
            }
            if (t < p.fcap) { p.faces[3 * t] = f[0]; p.faces[3 * t + 1] = f[1]; p.faces[3 * t + 2] = f[2]; }
        }
    }
}

}  // namespace vc
