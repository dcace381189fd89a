// gfx950 kernels of the carve path.  Host code and the C ABI: voxcarve.hip.
//
// The slab's n voxels are numbered j = 0..n-1 in the reference's order (y fastest,
// voxel_reconstruction.py:57).  The unit of work is a WORD of 64 voxels whose survivor bits come out of
// one __ballot, and a GROUP of 64 words (4096 voxels) is what one wavefront finishes per iteration and
// what the ordered compaction counts and scans.  A word is 64 consecutive y ("y-line") or, where the
// grid shape allows, a tile of 4 x-rows x 16 y (compact footprint, tighter pixel box); results always
// leave the kernels as y-major words.
//
//   per frame set   k_morph2x2 (optional), k_prep_pack, k_prep_grid (+ k_coarsen_grids for large grids): nothing returns to the host
//   per geometry    k_build_lut<TILE> (table and/or word boxes), k_brick_boxes_bm
//   carve           k_cull_bricks, k_brick_words, k_voxel_words<LUT,PAIR>, k_assemble   the brick pipeline (default for
//                                                    ny in {256, 512, 1024, 2048, 4096}; see its section below)
//                   k_lut_refine<B,HIER,PAIR,TILE>   one-launch hierarchical lookup-table kernel (other shapes, VC_MODE_LUT)
//                   k_carve_fused_hier<TILE,BOX>     the same with the projection in-kernel (other shapes, VC_MODE_FUSED)
//                   k_lut_first + k_lut_refine<.,false,.>   the table streamed without skipping (roofline_stream)
//                   k_carve_fused<KSUB,NY64>, k_carve_generic<LUT,VM>   chunked / one thread per voxel (any shape,
//                                                                     thresholds below C, camera bit masks)
//   compaction      k_count_groups, k_scan_groups, k_scan_blocks | k_finish_scan, k_emit_busy / k_emit_lanes / k_emit_words
//   multi-GPU       k_count_nz, k_pack_entries, k_count_entries, k_emit_lanes<.,INDIRECT>, k_zero_dead_groups
//
// Cameras are visited most-selective first (order[]); every level stops as soon as nothing it covers can
// still pass.  All projection arithmetic is float64 with contraction off (vc_device.h).
#pragma once
#include "vc_device.h"

#pragma clang fp contract(off)

namespace vc {

constexpr uint32_t kBlock = 256;            // 4 wavefronts
constexpr uint32_t kSeenFlag = 1u << 24;    // bit 56 of a record: the colour camera sees the voxel
constexpr uint32_t kGroupWords = 64;        // compaction group = 64 words = 4096 voxels (one word per lane)
constexpr uint32_t kScanBlock = 1024;       // groups per scan workgroup
constexpr uint32_t kMaxCameras = 16;
constexpr uint32_t kLutPad = 8192;          // voxels; every chunking below divides it
constexpr uint64_t kEmptyBox = ~0ull;       // pixel box of a word with no in-image voxel

// One camera's block grids cover only the word columns [w_lo, w_lo + cws) and block rows [v_lo, v_lo + ch)
// that hold foreground (a silhouette fills a few percent of its image); outside them "any" and "all"
// are zero by definition.  ch == 0: the camera sees no foreground at all.
struct GridCam {
    uint32_t off;               // first word of any[]; all[] follows at off + ch * cws
    uint16_t w_lo, v_lo, cws, ch;
};

// A frame set's grid buffer starts with a HEADER the per-frame preparation kernels write on the device (nothing
// of a frame set's derived state ever visits the host): the cameras' grid descriptors (3 words each; kernels
// index them in LDS -- indexing a kernel-argument copy with a run-time camera number costs the compiler 48
// VGPRs), the block size and buffer length the preparation chose, and the camera visiting order.
constexpr uint32_t kHdrShift = 3 * kMaxCameras;        // log2 of the block edge in pixels
constexpr uint32_t kHdrWords = kHdrShift + 1;          // u32 words of header + grids (what the hierarchical kernels stage in LDS)
constexpr uint32_t kGridHeader = kHdrShift + 4;        // 52 words = 13 x 16 bytes
// Beside the header, one 128-byte line per camera and quantity so that atomics of different cameras never meet on a line
// (atomics on one LINE serialise at ~12 ns each): the cameras' foreground pixel boxes (k_prep_pack -> k_prep_grid),
// double-buffered by frame parity, [2][kMaxCameras][kBoxStride]; then the pass counts of a voxel sample, [kMaxCameras][kBoxStride]
// (k_prep_grid -> the carve kernels' camera visiting order).
constexpr uint32_t kBoxStride = 32;
constexpr uint32_t kCountBase = 2 * kMaxCameras * kBoxStride;
__device__ __forceinline__ uint32_t hdr_u32(const uint32_t *hdr, uint32_t i)          // wave-uniform i
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[i]);
}
// Camera visiting order (most selective first) from the pass counts k_prep_grid left beside the header: position of
// camera t = number of cameras with a smaller count (ties: lower camera number first).  Every workgroup works it
// out for itself (C <= 16) into s_order; ends with a barrier.
__device__ __forceinline__ void stage_order(const uint32_t *counts, uint32_t C, uint32_t *s_order)
{
    if (threadIdx.x < C) {
        const uint32_t mine = counts[threadIdx.x * kBoxStride];
        uint32_t rank = 0;
        for (uint32_t c = 0; c < C; ++c) {
            const uint32_t o = counts[c * kBoxStride];
            rank += (o < mine || (o == mine && c < threadIdx.x)) ? 1u : 0u;
        }
        s_order[rank] = threadIdx.x;
    }
    __syncthreads();
}
__device__ __forceinline__ uint32_t ord(const uint32_t *s_order, uint32_t q)         // wave-uniform q
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)s_order[q]);
}
__device__ __forceinline__ GridCam load_gridcam(const uint32_t *grids, uint32_t c)
{
    const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)grids[3 * c]);          // c is wave-uniform
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)grids[3 * c + 1]);
    const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)grids[3 * c + 2]);
    return {a, (uint16_t)b, (uint16_t)(b >> 16), (uint16_t)d, (uint16_t)(d >> 16)};
}

// Header + grids of a frame set into LDS: 16 bytes per lane, four loads in flight per lane (a plain copy loop is compiled
// to load, wait, store per round: one memory round trip per 16 bytes and lane).  Ends with a barrier.
__device__ __forceinline__ void stage_grids(uint32_t *s_grid, const uint32_t *__restrict__ blockgrid)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u *__restrict__ src = reinterpret_cast<const v4u *>(blockgrid);
    v4u *dst = reinterpret_cast<v4u *>(s_grid);
    const uint32_t n = ((uint32_t)__builtin_amdgcn_readfirstlane((int)blockgrid[kHdrWords]) + 3u) / 4u;      // (buffer padded to 16 B)
    const uint32_t bd = blockDim.x;
    for (uint32_t i = threadIdx.x; i < n; i += 4 * bd) {
        v4u v0 = src[i], v1 = src[i + bd < n ? i + bd : n - 1u], v2 = src[i + 2 * bd < n ? i + 2 * bd : n - 1u],
            v3 = src[i + 3 * bd < n ? i + 3 * bd : n - 1u];           // (clamped: no select behind a load)
        // (keeps the four loads together, ahead of the conditional stores: the compiler would sink each into its store's branch)
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        dst[i] = v0;
        if (i + bd < n) dst[i + bd] = v1;
        if (i + 2 * bd < n) dst[i + 2 * bd] = v2;
        if (i + 3 * bd < n) dst[i + 3 * bd] = v3;
    }
    __syncthreads();
}

struct CarveParams {
    const double *xs, *ys, *zs;
    const uint32_t *maskbits;   // [C][mwords] of the active frame set
    const int32_t *lut;         // [C][n_pad]
    const uint64_t *bbox;       // [C][n_pad/64] pixel bounding box of each 64-voxel word (u16 u0,v0,u1,v1)
    const int32_t *lut_tile;    // [C][n_pad] the same table in tile order: word T = 4 x-rows x 16 y of one z-layer
    const uint64_t *tbox;       // [C][n_pad/64] pixel boxes of the tile words
    uint32_t tq;                // tile words per row quad = ny / 16
    uint32_t tile_whole;        // 64 % tq == 0: the 64 tile words of a wave are exactly one y-major group
    // bricks of 16 x 16 x 16 voxels (4 row quads x 1 tile column x 16 layers = 64 tile words): culled per frame set
    const uint64_t *kbox;       // [C][nbrick_pad] pixel boxes of the bricks (geometry only, built with tbox)
    uint64_t *live;             // [2][nbrick_pad / 64]: bit per brick "may hold survivors", then "every voxel survives"; null: no culling
    uint32_t nbx, nbz;          // bricks along x and (slab-local) z; along y there are tq
    uint32_t nbrick_pad;        // bricks rounded up to 64
    const uint32_t *counts;     // the frame set's per-camera pass counts of a voxel sample, kBoxStride apart (camera visiting order)
    const uint32_t *blockgrid;  // the frame set's header (block size, length, camera order) + per camera (crop[c].off):
                                // any[ch][cws] then all[ch][cws], one bit per block of 2^shift x 2^shift pixels
    const uint32_t *coarsegrid; // the same for 4 x 4 times coarser blocks (k_coarsen_grids; large frame sets only, else null): what
                                // the brick level looks at
    uint64_t *words;
    uint32_t *groupcnt;         // survivors per group of 64 words (kernels that know it write it)
    uint32_t *groupnz;          // non-zero words per group, or null (brick pipeline with tq <= 64: what the compact exchange packs by)
    uint16_t *viewmask;
    uint64_t n;                 // voxels in the slab (< 2^32)
    uint64_t n_pad;             // LUT camera stride: n rounded up to kLutPad, tail entries = -1
    uint32_t nx, ny, nz, z0;
    uint32_t C, H, W, mwords;
    uint32_t min_views;
    unsigned long long *stats;  // option timing_detail: work counters (stat_add), else null
    uint32_t cull_lds_words;    // k_cull_bricks: u32 words of dynamic LDS it was launched with when that is an ESTIMATE of the coarse grids' length
    uint32_t compact_off;       // k_brick_words_wide: where the waves' compaction words start in the dynamic LDS (u32 words; 0: no room, bricks in lockstep)
                                // (0: the launch reserved what the grids can take at most); grids that turn out longer are not staged at all
    uint32_t dbg;               // experiments only (vc_set_option("dbg", ...), scripts/exp_bricks.py): 1 = skip the voxel level (undecided words
                                // count as alive), 2 = skip the word level too; 8 = preparation, carve and scan kernels launch and return
                                // at once (32 / 64 / 128 / 256 / 512: only the preparation / cull + word level / voxel level /
                                // assembly / scans do), 16 = the record expansion does (scripts/exp_streams.py: what do the launches cost each
                                // other, apart from their work); results are then WRONG on purpose
    CamDev cam[kMaxCameras];
};

struct EmitParams {
    const double *xs, *ys, *zs;
    const uint32_t *maskbits;   // colour camera's mask bits (or null)
    const uint32_t *frame;      // colour camera's image as one dword per pixel, laid out as the upper half of a record: R | G << 8 |
                                // B << 16 | seen << 24 (or null)
    const int32_t *lut;         // colour camera's packed table (FROM_LUT), else null
    uint32_t lut_tq;            // != 0: that table is in TILE order (tile words per row quad = ny / 16); 0: y-major
    const uint64_t *words;
    const uint32_t *busylist;   // k_emit_busy: the groups with survivors; busycount[0] of them
    const uint32_t *busycount;
    const uint64_t *entries;    // INDIRECT expansion: {bits, global index of bit 0} pairs instead of words
    uint32_t entry_chunk;       // ... taken `entry_chunk` (16 or 64) per wave: see k_count_entries
    const uint32_t *groupcnt;   // survivors per group
    const uint32_t *groupoff;   // exclusive scan of groupcnt inside each scan block
    const uint64_t *blockoff;   // exclusive scan of the scan blocks' sums
    uint64_t *records;
    uint64_t capacity;
    uint64_t n;
    uint64_t i0;                // global linear index of slab-local voxel 0
    uint32_t ngroups;
    uint32_t nx, ny, z0;
    uint32_t H, W;
    int has_cam;
    unsigned long long *stats;  // option timing_detail: work counters, else null
    uint32_t dbg;               // experiments only (see CarveParams::dbg): 16 = the expansion launches and returns
    CamDev cam;
};

__device__ __forceinline__ void decompose(uint32_t j, uint32_t nx, uint32_t ny,
                                          uint32_t &ix, uint32_t &iy, uint32_t &izl)
{
    const uint32_t t = j / ny;
    iy = j - t * ny;
    izl = t / nx;
    ix = t - izl * nx;
}

// ---------------------------------------------------------------- mask post-filter
// One pass of cv2.erode / cv2.dilate with the 2x2 MORPH_RECT element the reference applies after
// contour filling (background_subtraction.py:195-203): OpenCV anchors an even element at ksize/2,
// so the window of output (y, x) is rows y-1..y, columns x-1..x, for erosion AND dilation (no
// reflection), and pixels outside the image are ignored (morphologyDefaultBorderValue).
template <bool DILATE>
__global__ __launch_bounds__(kBlock) void k_morph2x2(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                     uint32_t H, uint32_t W)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= H * W) return;
    const uint32_t y = i / W, x = i - y * W;
    uint32_t v = in[i];
    if (x > 0) { const uint32_t o = in[i - 1]; v = DILATE ? (o > v ? o : v) : (o < v ? o : v); }
    if (y > 0) { const uint32_t o = in[i - W]; v = DILATE ? (o > v ? o : v) : (o < v ? o : v); }
    if (x > 0 && y > 0) { const uint32_t o = in[i - W - 1]; v = DILATE ? (o > v ? o : v) : (o < v ? o : v); }
    out[i] = (uint8_t)v;
}

// ---------------------------------------------------------------- generic carve
// One thread per voxel, any grid shape, any min_views.  LUT = stream the packed table
// instead of projecting; VM = also store the per-voxel camera bitmask (no early exit).
template <bool LUT, bool VM>
__global__ __launch_bounds__(kBlock) void k_carve_generic(const CarveParams p)
{
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = j < p.n;
    uint32_t vm = 0, cnt = 0;
    if (valid) {
        double X = 0, Y = 0, Z = 0;
        if (!LUT) {
            uint32_t ix, iy, izl;
            decompose((uint32_t)j, p.nx, p.ny, ix, iy, izl);
            X = p.xs[ix];
            Y = p.ys[iy];
            Z = p.zs[p.z0 + izl];
        }
        for (uint32_t c = 0; c < p.C; ++c) {
            // Voxels that can no longer reach min_views stop early (result unchanged).
            if (!VM && cnt + (p.C - c) < p.min_views) break;
            int32_t off;
            if (LUT) {
                off = p.lut[(size_t)c * p.n_pad + j];
            } else {
                double u, v;
                project_point(p.cam[c], X, Y, Z, u, v);
                off = pixel_offset(u, v, p.H, p.W);
            }
            if (off >= 0 && mask_bit(p.maskbits + (size_t)c * p.mwords, off)) {
                vm |= 1u << c;
                ++cnt;
            }
        }
        if (VM) p.viewmask[j] = (uint16_t)vm;
    }
    const bool keep = valid && cnt >= p.min_views;
    const uint64_t ballot = __ballot(keep);
    if ((threadIdx.x & 63u) == 0) p.words[j >> 6] = ballot;
}

// Survivor bits of a finished chunk: lane k stores sub-chunk k's ballot.
template <int KSUB>
__device__ __forceinline__ void store_chunk(const CarveParams &p, uint32_t chunk, uint32_t lane, uint32_t alive)
{
    uint64_t mine = 0;
#pragma unroll
    for (int k = 0; k < KSUB; ++k) {
        const uint64_t b = __ballot((alive >> k) & 1u);
        if (lane == (uint32_t)k) mine = b;
    }
    if (lane < (uint32_t)KSUB) p.words[(uint64_t)chunk * KSUB + lane] = mine;
}

// ---------------------------------------------------------------- LUT-streaming carve
// All-views case (min_views == C), two launches:
//
//  k_lut_first   the most selective camera's table, read for EVERY voxel: a pure HBM
//                stream (4 B per voxel, 16 B per lane per load, software-pipelined one
//                chunk ahead) tested against that camera's bit-packed mask held in LDS
//                (the gather would otherwise be texture-addresser bound).  Writes the
//                alive bits, one u64 word per 64 voxels.
//  k_lut_refine  the other cameras, only where an alive bit is left: B alive words at a
//                time, per-lane predicated loads, so the tables are read sparsely.
//                Writes the final words and the per-tile survivor counts.
//
// A lane of k_lut_first holds 4 consecutive voxels (one dwordx4), so the survivor nibbles
// are transposed into 64-voxel words with four DPP row shifts (lane 15 of every row of 16
// lanes ends up with the word).
__device__ __forceinline__ uint32_t row_or_reduce(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    return v;
}

constexpr uint32_t kFirstBlock = 512;

template <int KV>
__global__ __launch_bounds__(kFirstBlock) void k_lut_first(const CarveParams p)
{
    extern __shared__ uint32_t s_mask[];                         // first camera's mask bits
    uint32_t c0 = 0;                                              // the most selective camera (lowest pass count)
    {
        uint32_t best = hdr_u32(p.counts, 0);
        for (uint32_t c = 1; c < p.C; ++c) {
            const uint32_t v = hdr_u32(p.counts, c * kBoxStride);
            if (v < best) { best = v; c0 = c; }
        }
    }
    {
        const uint32_t *__restrict__ mb = p.maskbits + (size_t)c0 * p.mwords;
        for (uint32_t i = threadIdx.x; i < p.mwords; i += kFirstBlock) s_mask[i] = mb[i];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kFirstBlock + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (kFirstBlock / 64);
    const uint32_t nchunks = (uint32_t)(p.n_pad / (256 * KV));
    if (wave0 >= nchunks) return;
    const int4 *__restrict__ L = reinterpret_cast<const int4 *>(p.lut + (size_t)c0 * p.n_pad);

    auto fetch = [&](uint32_t ch, int4 (&dst)[KV]) {
        if (ch >= nchunks) ch = nchunks - 1;                     // clamped: branch-free, in bounds
        const int4 *src = L + (size_t)ch * (64 * KV) + lane;
#pragma unroll
        for (int k = 0; k < KV; ++k) dst[k] = src[64 * k];
    };
    auto test = [&](int32_t o) -> uint32_t {
        const uint32_t w = s_mask[o >= 0 ? ((uint32_t)o >> 5) : 0u];
        return (o >= 0) ? ((w >> ((uint32_t)o & 31u)) & 1u) : 0u;
    };

    int4 nxt[KV];
    fetch(wave0, nxt);
    const uint32_t sh = (lane & 7u) * 4u;
    const bool upper = (lane & 8u) != 0;
    for (uint32_t chunk = wave0; chunk < nchunks; chunk += nwaves) {
        int4 off[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) off[k] = nxt[k];
        fetch(chunk + nwaves, nxt);
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const uint32_t nib = test(off[k].x) | (test(off[k].y) << 1) | (test(off[k].z) << 2) | (test(off[k].w) << 3);
            const uint32_t lo = row_or_reduce(upper ? 0u : (nib << sh));
            const uint32_t hi = row_or_reduce(upper ? (nib << sh) : 0u);
            if ((lane & 15u) == 15u)
                p.words[((uint64_t)chunk * KV + k) * 4 + (lane >> 4)] = ((uint64_t)hi << 32) | lo;
        }
    }
}

// A word's pixel box against one camera's block grids.  0: no foreground block in the box -- none of
// its voxels can pass; 2: every voxel lands inside the image (kBoxAllInside) and every block the box
// touches is entirely foreground -- all of its voxels pass, no table or mask read needed; 1: undecided.
// Conservative (block granularity); boxes taller than 16 or wider than 64 blocks are undecided.
constexpr uint64_t kBoxAllInside = 1ull << 63;    // flag in the box word (mask heights < 32768)

__device__ __forceinline__ uint32_t box_test(const uint32_t *__restrict__ grids, const GridCam gc, uint64_t bb, uint32_t gshift)
{
    if (bb == kEmptyBox || gc.ch == 0) return 0;
    const uint32_t bu0 = (uint32_t)(bb & 0xffffu) >> gshift, bv0 = (uint32_t)((bb >> 16) & 0xffffu) >> gshift;
    const uint32_t bu1 = (uint32_t)((bb >> 32) & 0xffffu) >> gshift, bv1 = (uint32_t)((bb >> 48) & 0x7fffu) >> gshift;
    const uint32_t w0 = bu0 >> 5, w1 = bu1 >> 5;
    const uint32_t cw_hi = (uint32_t)gc.w_lo + gc.cws - 1u, cv_hi = (uint32_t)gc.v_lo + gc.ch - 1u;
    if (w1 < gc.w_lo || w0 > cw_hi || bv1 < gc.v_lo || bv0 > cv_hi) return 0;      // nowhere near the foreground
    if (bv1 - bv0 > 15u || bu1 - bu0 > 63u) return 1;
    const uint32_t r0 = bv0 > gc.v_lo ? bv0 : gc.v_lo, r1 = bv1 < cv_hi ? bv1 : cv_hi;
    const uint32_t c0 = w0 > gc.w_lo ? w0 : gc.w_lo, c1 = w1 < cw_hi ? w1 : cw_hi;
    const uint32_t *__restrict__ g_any = grids + gc.off;
    const uint32_t *__restrict__ g_all = g_any + (uint32_t)gc.ch * gc.cws;
    uint32_t any = 0;
    uint32_t miss = (r0 != bv0 || r1 != bv1 || c0 != w0 || c1 != w1) ? 1u : 0u;    // part of the box lies outside the kept blocks
    const uint32_t m0 = 0xffffffffu << (bu0 & 31u), m1 = 0xffffffffu >> (31u - (bu1 & 31u));
    if (c0 == c1 && r1 - r0 < 4u) {
        // the common shape (a tile word's box: one grid word wide, up to 4 block rows): straight-line, no loop
        uint32_t m = 0xffffffffu;
        if (c0 == w0) m &= m0;
        if (c0 == w1) m &= m1;
        const uint32_t i0 = (r0 - gc.v_lo) * gc.cws + (c0 - gc.w_lo);
        const uint32_t nr = r1 - r0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t i = i0 + (k <= nr ? k : nr) * gc.cws;         // rows past the box repeat its last row
            any |= g_any[i] & m;
            miss |= ~g_all[i] & m;
        }
    } else {
#pragma nounroll
        for (uint32_t r = r0; r <= r1; ++r) {
#pragma nounroll
            for (uint32_t w = c0; w <= c1; ++w) {
                uint32_t m = 0xffffffffu;
                if (w == w0) m &= m0;
                if (w == w1) m &= m1;
                const uint32_t i = (r - gc.v_lo) * gc.cws + (w - gc.w_lo);
                any |= g_any[i] & m;
                miss |= ~g_all[i] & m;
            }
        }
    }
    if (any == 0) return 0;
    return (miss == 0 && (bb & kBoxAllInside)) ? 2u : 1u;
}

// The same answers for large images, where a word's box is often five or six block rows tall and box_test's loop (one row and one
// grid word per dependent trip, taken by the whole wave as soon as one lane has such a box) was a good part of k_brick_words at
// 512^3 x 16 x 1080p: a box up to 32 blocks wide is a 64-bit window on two neighbouring grid words, two rows per trip, every
// lane making the trips of the tallest box among the wave's (rows past a lane's box repeat its last row).  The neighbour of
// the first or last kept word may lie outside the camera's rectangle: read all the same (LDS; before the first row that is the
// word in front of the grid), never looked at (its half of the window is empty).
// N boxes side by side against one camera (the lanes' N list entries of one trip of brick_words_compact): no branches, one row
// loop for all.  Boxes wider than 32 blocks are undecided here (box_test looks at up to 64: work, not results).
constexpr uint32_t kRowsPerTrip = 2;
template <int N>
__device__ __forceinline__ void box_test_rows_n(const uint32_t *__restrict__ grids, const GridCam gc, const uint64_t (&bb)[N], const bool (&on)[N],
                                                uint32_t gshift, uint32_t (&res)[N])
{
    const uint32_t cw_hi = (uint32_t)gc.w_lo + gc.cws - 1u, cv_hi = (uint32_t)gc.v_lo + gc.ch - 1u;
    const uint32_t *__restrict__ g_any = grids + gc.off;
    const uint32_t *__restrict__ g_all = g_any + (uint32_t)gc.ch * gc.cws;
    uint32_t mlo[N], mhi[N], miss[N], any[N], nr[N], fixed[N];
    int32_t i0[N];
    bool look[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const bool live = on[j] && bb[j] != kEmptyBox && gc.ch != 0;
        const uint32_t bu0 = (uint32_t)(bb[j] & 0xffffu) >> gshift, bv0 = (uint32_t)((bb[j] >> 16) & 0xffffu) >> gshift;
        const uint32_t bu1 = (uint32_t)((bb[j] >> 32) & 0xffffu) >> gshift, bv1 = (uint32_t)((bb[j] >> 48) & 0x7fffu) >> gshift;
        const uint32_t w0 = bu0 >> 5, w1 = bu1 >> 5;
        const bool near = live && !(w1 < gc.w_lo || w0 > cw_hi || bv1 < gc.v_lo || bv0 > cv_hi);
        const bool big = bv1 - bv0 > 15u || bu1 - bu0 > 31u;
        look[j] = near && !big;
        fixed[j] = near && big ? 1u : 0u;                           // the answer of a test that does not look at the grids
        const uint32_t r0 = bv0 > gc.v_lo ? bv0 : gc.v_lo, r1 = bv1 < cv_hi ? bv1 : cv_hi;
        miss[j] = (r0 != bv0 || r1 != bv1) ? 1u : 0u;
        const uint64_t m = ((2ull << ((bu1 - bu0) & 31u)) - 1ull) << (bu0 & 31u);
        mlo[j] = (uint32_t)m; mhi[j] = (uint32_t)(m >> 32);
        if (w0 < gc.w_lo) { miss[j] = 1u; mlo[j] = 0u; }
        if (w0 + 1u > cw_hi) { miss[j] |= mhi[j]; mhi[j] = 0u; }
        if (!look[j]) { mlo[j] = 0u; mhi[j] = 0u; }
        i0[j] = look[j] ? (int32_t)__umul24(r0 - gc.v_lo, gc.cws) + (int32_t)w0 - (int32_t)gc.w_lo : 0;
        nr[j] = look[j] ? r1 - r0 : 0u;
        any[j] = 0u;
    }
    uint32_t nrmax = nr[0];
#pragma unroll
    for (int j = 1; j < N; ++j) nrmax = nr[j] > nrmax ? nr[j] : nrmax;
    for (uint32_t k = 0; __ballot(k <= nrmax) != 0ull; k += kRowsPerTrip) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
#pragma unroll
            for (uint32_t rr = 0; rr < kRowsPerTrip; ++rr) {
                const int32_t ia = i0[j] + (int32_t)__umul24(k + rr < nr[j] ? k + rr : nr[j], gc.cws);
                any[j] |= (g_any[ia] & mlo[j]) | (g_any[ia + 1] & mhi[j]);
                miss[j] |= (~g_all[ia] & mlo[j]) | (~g_all[ia + 1] & mhi[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j)
        res[j] = !look[j] ? fixed[j] : any[j] == 0u ? 0u : (miss[j] == 0u && (bb[j] & kBoxAllInside)) ? 2u : 1u;
}

// Sum over the 64 lanes with data-parallel-primitive moves (no trip through the LDS crossbar: six ds_bpermute in a row cost
// ~700 cycles of latency); every lane gets the total.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8: lane 15 of each row holds the row's sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Result of a wave that worked on tile words (lane = tile word gw + lane, 4 x-rows x 16 y each): lane (r, k)
// assembles the y-major word of row r, y chunk k from tile words 4k .. 4k+3 (16 bits each) and stores it.
// tile_whole: the wave's words are exactly y-major group g, its count is stored; otherwise the counts
// are accumulated with atomics into a zeroed groupcnt.
__device__ __forceinline__ uint64_t tile_transpose(uint64_t mine, uint32_t lane)
{
    const uint32_t r = lane >> 4, k = lane & 15u;
    uint64_t out = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int src = (int)(4 * k + q);
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)mine, src);
        const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(mine >> 32), src);
        const uint64_t m = ((uint64_t)hi << 32) | lo;
        out |= ((m >> (16 * r)) & 0xffffull) << (16 * q);
    }
    return out;
}
// y-major word index of row r of the row quad of tile word T, at T's y (T = first of the four tile words of a y-major word)
__device__ __forceinline__ uint64_t tile_word_index(const CarveParams &p, uint64_t T, uint32_t r)
{
    const uint32_t quad = (uint32_t)(T / p.tq), ty = (uint32_t)(T - (uint64_t)quad * p.tq);
    const uint32_t qpl = p.nx >> 2;                               // row quads per layer
    const uint32_t izl = quad / qpl, qx = quad - izl * qpl;
    return (((uint64_t)izl * p.nx + qx * 4 + r) * p.ny + (uint64_t)ty * 16) >> 6;
}
__device__ __forceinline__ void tile_store(const CarveParams &p, uint32_t g, uint64_t gw, uint32_t lane, uint64_t mine)
{
    const uint32_t r = lane >> 4, k = lane & 15u;
    const uint64_t out = tile_transpose(mine, lane);
    const uint64_t T = gw + 4 * k;                                // first of the four tile words
    if (T < (p.n >> 6)) {
        const uint64_t lw = tile_word_index(p, T, r);
        p.words[lw] = out;
        if (!p.tile_whole && (p.tq & 63u)) {                      // words of any group: one atomic per word with survivors
            const uint32_t pc = (uint32_t)__popcll(out);
            if (pc) atomicAdd(&p.groupcnt[lw >> 6], pc);
        }
    }
    if (!p.tile_whole && !(p.tq & 63u)) {
        // ny = 2048, 4096: the 16 words of row r are 1024 consecutive voxels of one group: one atomic per row
        uint32_t pc = T < (p.n >> 6) ? (uint32_t)__popcll(out) : 0u;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) pc += __shfl_xor(pc, d);
        if (k == 0 && pc) atomicAdd(&p.groupcnt[tile_word_index(p, T, r) >> 6], pc);
    }
    if (p.tile_whole) {
        const uint32_t cnt = wave_sum_u32((uint32_t)__popcll(mine));
        if (lane == 0) p.groupcnt[g] = cnt;
    }
}

// Brick culling (k_cull): what the coarse level decided for the brick that holds tile word T.  live = the brick may hold
// survivors; full = every voxel of it survives (inside every image, every block of every camera's box foreground).
__device__ __forceinline__ void brick_bits(const CarveParams &p, uint64_t T, bool &live, bool &full)
{
    const uint32_t quad = (uint32_t)(T / p.tq), ty = (uint32_t)(T - (uint64_t)quad * p.tq), qpl = p.nx >> 2;
    const uint32_t izl = quad / qpl, qx = quad - izl * qpl;
    const uint32_t b = ((izl >> 4) * p.nbx + (qx >> 2)) * p.tq + ty;
    const uint64_t lw = p.live[b >> 6], fw = p.live[(p.nbrick_pad >> 6) + (b >> 6)];
    live = (lw >> (b & 63u)) & 1ull;
    full = (fw >> (b & 63u)) & 1ull;
}

// HIER = false: refines the alive words k_lut_first left, cameras order[1..].
// HIER = true : no first pass at all.  Coarse pass, lane = word: each camera's pixel box of the word
//               against that camera's block grids (LDS).  Any camera with no foreground block in the
//               box kills the word; a camera whose box is all-foreground needs no further look; the
//               others are recorded in the word's `need` mask.  Words with an empty `need` mask are
//               final (all 64 voxels survive); the rest take the exact per-voxel test, only for the
//               cameras in their mask.  Exact: the box contains the pixel of every voxel of the word.
// TILE: the 64 voxels of a word are 4 neighbouring x-rows x 16 consecutive y of one z-layer instead of
// 64 consecutive y.  A compact footprint has a pixel box of ~2 blocks instead of ~5, so the box test
// decides more words (undecided 7 % -> 3.8 % in the hull's layers at 1024^3, scripts/exp_shapes.py).
// Needs nx % 4 == 0 and ny % 64 == 0; the table and the boxes are kept in that order too
// (k_build_lut<true>), and the wave transposes its 64 result words back to y-major before storing them.
template <int B, bool HIER, bool PAIR, bool TILE = false>
__device__ __forceinline__ void lut_refine_body(const CarveParams &p, uint32_t vblock, uint32_t nblocks, uint32_t *s_grid)
{
    if (HIER) stage_grids(s_grid, p.blockgrid);
    __shared__ uint32_t s_order[kMaxCameras];
    stage_order(p.counts, p.C, s_order);       // camera order (and block size) of this frame set
    const uint32_t gshift = HIER ? hdr_u32(s_grid, kHdrShift) : 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((vblock * kBlock + threadIdx.x) >> 6);
    const uint32_t nwaves = nblocks * (kBlock / 64);
    const uint32_t ngroups = (uint32_t)(p.n_pad / 4096);         // 64 words of 64 voxels
    const uint64_t nwords = p.n_pad >> 6;
    const uint32_t qfirst = HIER ? 0u : 1u;
    uint64_t next = 0;
    if (!HIER) next = (wave0 < ngroups) ? p.words[(uint64_t)wave0 * 64 + lane] : 0ull;
    for (uint32_t g = wave0; g < ngroups; g += nwaves) {
        const uint64_t gw = (uint64_t)g * 64;                     // first word of the group
        uint64_t mine;
        uint32_t need = 0xffffffffu;                              // cameras (positions in order[]) still to test exactly
        if (HIER) {
            const uint64_t j0 = (gw + lane) << 6;                 // TILE: n % 64 == 0, a word is whole or padding
            bool cand = j0 < p.n, full = false;
            if (TILE && p.live) {
                // brick level first: what k_cull decided for the 16^3 brick around this word.  With ny == 1024 the wave's
                // 64 words lie in the 64 bricks of one brick row: two scalar loads, and 4 of 5 groups end right here
                if (p.tq == 64) {
                    const uint32_t qpl = p.nx >> 2, izl = g / qpl, qx = g - izl * qpl;       // uniform
                    const uint32_t bw = (izl >> 4) * p.nbx + (qx >> 2);
                    const uint64_t lw = p.live[bw], fw = p.live[(p.nbrick_pad >> 6) + bw];
                    full = cand && ((fw >> lane) & 1ull);
                    cand = cand && ((lw >> lane) & 1ull);
                } else if (cand) {
                    bool lv;
                    brick_bits(p, gw + lane, lv, full);
                    cand = lv;
                }
                if (p.tile_whole && __ballot(cand) == 0) {        // dead group: count only (see below)
                    if (lane == 0) p.groupcnt[g] = 0;
                    continue;
                }
            }
            // coarse pass, four cameras' boxes in flight at a time (words of a "full" brick need no look at all)
            need = 0;
            for (uint32_t q0 = 0; q0 < p.C && __ballot(cand && !full) != 0; q0 += 4) {
                uint64_t bb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    bb[k] = (q0 + k < p.C && cand && !full) ? (TILE ? p.tbox : p.bbox)[(size_t)ord(s_order, q0 + k) * nwords + gw + lane] : 0ull;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (q0 + k < p.C && cand && !full) {
                        const uint32_t r = box_test(s_grid, load_gridcam(s_grid, ord(s_order, q0 + k)), bb[k], gshift);
                        cand = r != 0;
                        if (r == 1) need |= 1u << (q0 + k);
                    }
                }
            }
            // a live word starts with every voxel of the slab alive (padding excluded)
            mine = 0;
            if (cand) mine = (p.n - j0 >= 64) ? ~0ull : ((1ull << (p.n - j0)) - 1ull);
            if (!cand) need = 0;
        } else {
            mine = next;
            const uint32_t gn = (g + nwaves < ngroups) ? g + nwaves : g;   // clamped prefetch
            next = p.words[(uint64_t)gn * 64 + lane];
        }
        uint64_t nz = __ballot(mine != 0 && need != 0);           // words that need the exact pass
        while (p.C > qfirst && nz != 0) {                         // wave-uniform
            uint32_t li[B], nd[B];
            uint32_t alive = 0;
#pragma unroll
            for (int b = 0; b < B; ++b) {
                li[b] = 64; nd[b] = 0;
                if (nz != 0) {
                    li[b] = (uint32_t)__builtin_ctzll(nz);
                    nz &= nz - 1;
                    const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, (int)li[b]);
                    const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), (int)li[b]);
                    nd[b] = (uint32_t)__builtin_amdgcn_readlane((int)need, (int)li[b]);
                    const uint64_t wv = ((uint64_t)whi << 32) | wlo;
                    if ((wv >> lane) & 1ull) alive |= 1u << b;
                }
            }
            uint32_t ndany = 0;                                   // cameras any word of the batch still needs
#pragma unroll
            for (int b = 0; b < B; ++b) ndany |= nd[b];
            for (uint32_t q = qfirst; q < p.C; q += PAIR ? 2 : 1) {
                // PAIR: two cameras' entries per dependent round trip (their loads and gathers overlap)
                if (((ndany >> q) & (PAIR ? 3u : 1u)) == 0) continue;   // decided by the boxes for the whole batch
                const bool two = PAIR && q + 1 < p.C;
                const uint32_t c = ord(s_order, q), c2 = ord(s_order, two ? q + 1 : q);
                const int32_t *__restrict__ L = (TILE ? p.lut_tile : p.lut) + (size_t)c * p.n_pad + gw * 64 + lane;
                const int32_t *__restrict__ L2 = (TILE ? p.lut_tile : p.lut) + (size_t)c2 * p.n_pad + gw * 64 + lane;
                const uint32_t *__restrict__ mb = p.maskbits + (size_t)c * p.mwords;
                const uint32_t *__restrict__ mb2 = p.maskbits + (size_t)c2 * p.mwords;
                int32_t off[B], off2[B];
                uint32_t mw[B], mw2[B];
#pragma unroll
                for (int b = 0; b < B; ++b) {                     // a camera outside the word's mask counts as passed
                    const bool t1 = ((nd[b] >> q) & 1u) && ((alive >> b) & 1u);
                    const bool t2 = two && ((nd[b] >> (q + 1)) & 1u) && ((alive >> b) & 1u);
                    off[b] = t1 ? L[(size_t)li[b] * 64] : -2;
                    off2[b] = t2 ? L2[(size_t)li[b] * 64] : -2;
                }
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    mw[b] = (off[b] >= 0) ? mb[(uint32_t)off[b] >> 5] : (off[b] == -2 ? ~0u : 0u);
                    mw2[b] = !PAIR ? ~0u : (off2[b] >= 0) ? mb2[(uint32_t)off2[b] >> 5] : (off2[b] == -2 ? ~0u : 0u);
                }
#pragma unroll
                for (int b = 0; b < B; ++b)
                    if (!((mw[b] >> ((uint32_t)off[b] & 31u)) & (mw2[b] >> ((uint32_t)off2[b] & 31u)) & 1u)) alive &= ~(1u << b);
                if (__ballot(alive != 0) == 0) break;
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                if (li[b] < 64) {
                    const uint64_t nb = __ballot((alive >> b) & 1u);
                    if (lane == li[b]) mine = nb;
                }
            }
        }
        // The words of a group without survivors are NOT written (5 of 6 groups at 1024^3): every reader
        // looks at groupcnt first (expansion, packing), vc_fetch_occupancy zero-fills them on demand.
        const bool whole = HIER && (!TILE || p.tile_whole);       // this wave's 64 words are y-major group g
        if (whole && __ballot(mine != 0) == 0) {
            if (lane == 0) p.groupcnt[g] = 0;
            continue;
        }
        if (TILE) {
            tile_store(p, g, gw, lane, mine);
            continue;
        }
        p.words[gw + lane] = mine;
        uint32_t cnt = (uint32_t)__popcll(mine);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        if (lane == 0) p.groupcnt[g] = cnt;
    }
}

template <int B, bool HIER, bool PAIR, bool TILE = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lut_refine(const CarveParams p)
{
    extern __shared__ uint32_t s_grid[];                          // HIER: the cropped block grids of all cameras
    lut_refine_body<B, HIER, PAIR, TILE>(p, blockIdx.x, gridDim.x, s_grid);
}

// ---------------------------------------------------------------- fused carve
// All-views case.  FP64-VALU-bound: coordinates regenerated from the index, projection
// in-kernel, nothing streamed from HBM but the bit-packed masks (L2 resident).
// NY64: ny % 64 == 0, so a sub-chunk is 64 consecutive y of one (ix, iz) column and its
// index arithmetic is wave-uniform (scalar unit); otherwise per-lane division.
template <int KSUB, bool NY64>
__global__ __launch_bounds__(kBlock) void k_carve_fused(const CarveParams p)
{
    __shared__ uint32_t s_order[kMaxCameras];
    stage_order(p.counts, p.C, s_order);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (kBlock / 64);
    const uint32_t nchunks = (uint32_t)((p.n + 64 * KSUB - 1) / (64 * KSUB));
    for (uint32_t chunk = wave0; chunk < nchunks; chunk += nwaves) {
        const uint64_t base = (uint64_t)chunk * (64 * KSUB);
        uint32_t alive = 0;
        double X[KSUB], Y[KSUB], Z[KSUB];
        if (NY64) {
            uint32_t ix, iy, izl;                        // of the chunk's first voxel: uniform
            decompose((uint32_t)base, p.nx, p.ny, ix, iy, izl);
#pragma unroll
            for (int k = 0; k < KSUB; ++k) {
                if (base + 64u * k < p.n) {              // whole sub-chunk valid (n % 64 == 0)
                    alive |= 1u << k;
                    X[k] = p.xs[ix];
                    Y[k] = p.ys[iy + lane];
                    Z[k] = p.zs[p.z0 + izl];
                } else {
                    X[k] = Y[k] = Z[k] = 0.0;
                }
                iy += 64;
                if (iy == p.ny) {
                    iy = 0;
                    if (++ix == p.nx) { ix = 0; ++izl; }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < KSUB; ++k) {
                const uint64_t j = base + 64u * k + lane;
                X[k] = Y[k] = Z[k] = 0.0;
                if (j < p.n) {
                    uint32_t ix, iy, izl;
                    decompose((uint32_t)j, p.nx, p.ny, ix, iy, izl);
                    alive |= 1u << k;
                    X[k] = p.xs[ix];
                    Y[k] = p.ys[iy];
                    Z[k] = p.zs[p.z0 + izl];
                }
            }
        }
        for (uint32_t q = 0; q < p.C; ++q) {
            const uint32_t c = ord(s_order, q);
            const uint32_t *__restrict__ mb = p.maskbits + (size_t)c * p.mwords;
#pragma unroll
            for (int k = 0; k < KSUB; ++k) {
                if ((alive >> k) & 1u) {
                    double u, v;
                    project_point(p.cam[c], X[k], Y[k], Z[k], u, v);
                    const int32_t off = pixel_offset(u, v, p.H, p.W);
                    if (!(off >= 0 && mask_bit(mb, off))) alive &= ~(1u << k);
                }
            }
            if (__ballot(alive != 0) == 0) break;
        }
        store_chunk<KSUB>(p, chunk, lane, alive);
    }
}

// ---------------------------------------------------------------- fused carve, hierarchical
// Table-free counterpart of the hierarchical LUT carve (needs ny % 64 == 0, so a word is 64
// consecutive y of one (x, z) column).  Coarse pass, lane = word: the camera-frame coordinates
// are linear in y, so the word is a line segment; interval arithmetic through OpenCV's projection
// (division, distortion polynomial, intrinsics) gives a pixel box that CONTAINS every voxel's
// pixel.  The box is widened by a pixel and a relative 1e-9, orders of magnitude above the
// float64 rounding of the interval evaluation, and a word whose interval cannot be bounded
// (depth interval touching zero, non-finite values) is simply kept.  A word survives the coarse
// pass only if every camera's box holds a foreground block (LDS grid).  Fine pass: the exact
// per-voxel float64 test of k_carve_fused, for the candidate words only.
struct Iv { double lo, hi; };

__device__ __forceinline__ Iv iv_add(Iv a, Iv b) { return {a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ Iv iv_addc(Iv a, double c) { return {a.lo + c, a.hi + c}; }
__device__ __forceinline__ Iv iv_scale(Iv a, double k) { return k >= 0 ? Iv{a.lo * k, a.hi * k} : Iv{a.hi * k, a.lo * k}; }
__device__ __forceinline__ Iv iv_mul(Iv a, Iv b)
{
    const double p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    return {fmin(fmin(p0, p1), fmin(p2, p3)), fmax(fmax(p0, p1), fmax(p2, p3))};
}
__device__ __forceinline__ Iv iv_sqr(Iv a)
{
    const double l = a.lo * a.lo, h = a.hi * a.hi;
    if (a.lo <= 0.0 && a.hi >= 0.0) return {0.0, fmax(l, h)};
    return {fmin(l, h), fmax(l, h)};
}

// Pixel box of the voxel set {(x, y, Z): x in [xa, xb], y in [ya, yb]} for camera c (a y-line word has
// xa == xb), packed as bbox words are; kEmptyBox when no voxel of it can be inside the image; ~1ull
// ("maybe") when it cannot be bounded.
constexpr uint64_t kMaybeBox = ~1ull;
__device__ __forceinline__ uint64_t segment_box(const CamDev &c, double xa, double xb, double ya, double yb, double Z,
                                                uint32_t H, uint32_t W)
{
    const Iv X = {fmin(xa, xb), fmax(xa, xb)};
    const Iv Y = {fmin(ya, yb), fmax(ya, yb)};
    const Iv x = iv_addc(iv_add(iv_scale(X, c.r[0]), iv_scale(Y, c.r[1])), c.r[2] * Z + c.t[0]);
    const Iv y = iv_addc(iv_add(iv_scale(X, c.r[3]), iv_scale(Y, c.r[4])), c.r[5] * Z + c.t[1]);
    const Iv z = iv_addc(iv_add(iv_scale(X, c.r[6]), iv_scale(Y, c.r[7])), c.r[8] * Z + c.t[2]);
    // keep a margin around z = 0: there the projection takes the `z ? 1/z : 1` branch or blows up
    const double zmag = fmax(fabs(z.lo), fabs(z.hi));
    if (!(z.lo > 1e-6 * zmag || z.hi < -1e-6 * zmag) || zmag == 0.0) return kMaybeBox;
    const Iv inv = {1.0 / z.hi, 1.0 / z.lo};
    const Iv xn = iv_mul(x, inv), yn = iv_mul(y, inv);
    const Iv x2 = iv_sqr(xn), y2 = iv_sqr(yn);
    const Iv r2 = iv_add(x2, y2);
    const Iv r4 = iv_sqr(r2);
    const Iv r6 = iv_mul(r4, r2);
    const Iv cdist = iv_addc(iv_add(iv_add(iv_scale(r2, c.k1), iv_scale(r4, c.k2)), iv_scale(r6, c.k3)), 1.0);
    const Iv a1 = iv_scale(iv_mul(xn, yn), 2.0);
    const Iv a2 = iv_add(r2, iv_scale(x2, 2.0));
    const Iv a3 = iv_add(r2, iv_scale(y2, 2.0));
    const Iv xd = iv_add(iv_add(iv_mul(xn, cdist), iv_scale(a1, c.p1)), iv_scale(a2, c.p2));
    const Iv yd = iv_add(iv_add(iv_mul(yn, cdist), iv_scale(a3, c.p1)), iv_scale(a1, c.p2));
    Iv u = iv_addc(iv_scale(xd, c.fx), c.cx);
    Iv v = iv_addc(iv_scale(yd, c.fy), c.cy);
    if (!(isfinite(u.lo) && isfinite(u.hi) && isfinite(v.lo) && isfinite(v.hi))) return kMaybeBox;
    u.lo -= 1.0 + 1e-9 * fabs(u.lo); u.hi += 1.0 + 1e-9 * fabs(u.hi);
    v.lo -= 1.0 + 1e-9 * fabs(v.lo); v.hi += 1.0 + 1e-9 * fabs(v.hi);
    if (u.hi < 0.0 || v.hi < 0.0 || u.lo >= (double)W || v.lo >= (double)H) return kEmptyBox;
    const uint32_t u0 = u.lo > 0.0 ? (uint32_t)u.lo : 0u, v0 = v.lo > 0.0 ? (uint32_t)v.lo : 0u;
    const uint32_t u1 = u.hi < (double)(W - 1) ? (uint32_t)u.hi : W - 1, v1 = v.hi < (double)(H - 1) ? (uint32_t)v.hi : H - 1;
    // the (conservative) box inside the image => every voxel of the segment is inside the image
    const bool inside = u.lo >= 0.0 && v.lo >= 0.0 && u.hi < (double)W && v.hi < (double)H;
    return (uint64_t)u0 | ((uint64_t)v0 << 16) | ((uint64_t)u1 << 32) | ((uint64_t)v1 << 48) | (inside ? kBoxAllInside : 0ull);
}

// The same box with the nonlinear part (division, distortion polynomial, intrinsics) evaluated in
// float32 intervals: the coarse pass is FP64-VALU bound and this is most of its arithmetic.  The rigid
// transform stays in float64 (it is where cancellation lives: R.X + t with |t| in the thousands); its
// three intervals are widened by one float32 ulp when narrowed.  Everything after it only ever
// multiplies and adds quantities whose magnitudes are tracked (M, Mt below), so the float32 rounding
// (~1e-7 relative per operation, ~10 operations deep) is bounded by 1e-6 x those magnitudes; the box is
// widened by 1 px + 1e-5 x them, and a word that cannot be bounded is kept, as before.
struct Ivf { float lo, hi; };
__device__ __forceinline__ Ivf ivf_add(Ivf a, Ivf b) { return {a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ Ivf ivf_addc(Ivf a, float c) { return {a.lo + c, a.hi + c}; }
__device__ __forceinline__ Ivf ivf_scale(Ivf a, float k) { return k >= 0 ? Ivf{a.lo * k, a.hi * k} : Ivf{a.hi * k, a.lo * k}; }
__device__ __forceinline__ Ivf ivf_mul(Ivf a, Ivf b)
{
    const float p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    return {fminf(fminf(p0, p1), fminf(p2, p3)), fmaxf(fmaxf(p0, p1), fmaxf(p2, p3))};
}
__device__ __forceinline__ Ivf ivf_sqr(Ivf a)
{
    const float l = a.lo * a.lo, h = a.hi * a.hi;
    if (a.lo <= 0.0f && a.hi >= 0.0f) return {0.0f, fmaxf(l, h)};
    return {fminf(l, h), fmaxf(l, h)};
}
__device__ __forceinline__ Ivf ivf_from(Iv a)                      // outward by one float32 ulp
{
    const float l = (float)a.lo, h = (float)a.hi;
    return {l - fabsf(l) * 1.2e-7f, h + fabsf(h) * 1.2e-7f};
}

__device__ __forceinline__ uint64_t segment_box_f32(const CamDev &c, double xa, double xb, double ya, double yb, double Z,
                                                    uint32_t H, uint32_t W)
{
    const Iv X = {fmin(xa, xb), fmax(xa, xb)};
    const Iv Y = {fmin(ya, yb), fmax(ya, yb)};
    const Iv xd64 = iv_addc(iv_add(iv_scale(X, c.r[0]), iv_scale(Y, c.r[1])), c.r[2] * Z + c.t[0]);
    const Iv yd64 = iv_addc(iv_add(iv_scale(X, c.r[3]), iv_scale(Y, c.r[4])), c.r[5] * Z + c.t[1]);
    const Iv zd64 = iv_addc(iv_add(iv_scale(X, c.r[6]), iv_scale(Y, c.r[7])), c.r[8] * Z + c.t[2]);
    const double zmag = fmax(fabs(zd64.lo), fabs(zd64.hi));
    if (!(zd64.lo > 1e-6 * zmag || zd64.hi < -1e-6 * zmag) || zmag == 0.0) return kMaybeBox;
    const Ivf x = ivf_from(xd64), y = ivf_from(yd64), z = ivf_from(zd64);
    const Ivf inv = {1.0f / z.hi, 1.0f / z.lo};
    const Ivf xn = ivf_mul(x, inv), yn = ivf_mul(y, inv);
    const Ivf x2 = ivf_sqr(xn), y2 = ivf_sqr(yn);
    const Ivf r2 = ivf_add(x2, y2);
    const Ivf r4 = ivf_sqr(r2);
    const Ivf r6 = ivf_mul(r4, r2);
    const float k1 = (float)c.k1, k2 = (float)c.k2, k3 = (float)c.k3, p1 = (float)c.p1, p2 = (float)c.p2;
    const float fx = (float)c.fx, fy = (float)c.fy, cx = (float)c.cx, cy = (float)c.cy;
    const Ivf cdist = ivf_addc(ivf_add(ivf_add(ivf_scale(r2, k1), ivf_scale(r4, k2)), ivf_scale(r6, k3)), 1.0f);
    const Ivf a1 = ivf_scale(ivf_mul(xn, yn), 2.0f);
    const Ivf a2 = ivf_add(r2, ivf_scale(x2, 2.0f));
    const Ivf a3 = ivf_add(r2, ivf_scale(y2, 2.0f));
    const Ivf xd = ivf_add(ivf_add(ivf_mul(xn, cdist), ivf_scale(a1, p1)), ivf_scale(a2, p2));
    const Ivf yd = ivf_add(ivf_add(ivf_mul(yn, cdist), ivf_scale(a3, p1)), ivf_scale(a1, p2));
    Ivf u = ivf_addc(ivf_scale(xd, fx), cx);
    Ivf v = ivf_addc(ivf_scale(yd, fy), cy);
    // magnitudes every rounded quantity was built from (not the possibly cancelled values)
    const float M = 1.0f + fabsf(k1) * r2.hi + fabsf(k2) * r4.hi + fabsf(k3) * r6.hi;
    const float Mt = 3.0f * (fabsf(p1) + fabsf(p2)) * r2.hi;
    const float xm = fmaxf(fabsf(xn.lo), fabsf(xn.hi)), ym = fmaxf(fabsf(yn.lo), fabsf(yn.hi));
    const float mu = 1.0f + 1e-5f * (fabsf(fx) * (xm * M + Mt) + fabsf(cx));
    const float mv = 1.0f + 1e-5f * (fabsf(fy) * (ym * M + Mt) + fabsf(cy));
    if (!(isfinite(u.lo) && isfinite(u.hi) && isfinite(v.lo) && isfinite(v.hi) && isfinite(mu) && isfinite(mv))) return kMaybeBox;
    u.lo -= mu; u.hi += mu;
    v.lo -= mv; v.hi += mv;
    if (u.hi < 0.0f || v.hi < 0.0f || u.lo >= (float)W || v.lo >= (float)H) return kEmptyBox;
    const uint32_t u0 = u.lo > 0.0f ? (uint32_t)u.lo : 0u, v0 = v.lo > 0.0f ? (uint32_t)v.lo : 0u;
    const uint32_t u1 = u.hi < (float)(W - 1) ? (uint32_t)u.hi : W - 1, v1 = v.hi < (float)(H - 1) ? (uint32_t)v.hi : H - 1;
    const bool inside = u.lo >= 0.0f && v.lo >= 0.0f && u.hi < (float)W && v.hi < (float)H;
    return (uint64_t)u0 | ((uint64_t)v0 << 16) | ((uint64_t)u1 << 32) | ((uint64_t)v1 << 48) | (inside ? kBoxAllInside : 0ull);
}

// TILE: words of 4 x-rows x 16 y (see lut_refine_body); the interval of a compact word is tighter in
// both image directions, and the result words go back to y-major through tile_store.
// BOX: where a word's pixel box comes from -- 0: float64 intervals, 1: float32 intervals after a float64
// rigid transform, 2: read from the boxes k_build_lut reduced once from the exact pixels (8 bytes per
// word and camera, geometry only: they survive every new frame set; no table is built or read).
template <bool TILE, int BOX>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_carve_fused_hier(const CarveParams p)
{
    extern __shared__ uint32_t s_grid[];                          // the cropped block grids of all cameras
    stage_grids(s_grid, p.blockgrid);
    __shared__ uint32_t s_order[kMaxCameras];
    stage_order(p.counts, p.C, s_order);
    const uint32_t gshift = hdr_u32(s_grid, kHdrShift);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (kBlock / 64);
    const uint32_t ngroups = (uint32_t)(p.n_pad / 4096);
    for (uint32_t g = wave0; g < ngroups; g += nwaves) {
        const uint64_t gw = (uint64_t)g * 64;
        // ---- coarse: lane = word
        const uint64_t j0 = (gw + lane) << 6;
        bool cand = j0 < p.n;                                     // n % 64 == 0 here: whole words only
        uint32_t ix = 0, iy = 0, izl = 0;                         // first voxel of the word
        if (cand && TILE) {
            const uint32_t quad = (uint32_t)((gw + lane) / p.tq), qpl = p.nx >> 2;
            iy = ((uint32_t)(gw + lane) - quad * p.tq) * 16;
            izl = quad / qpl;
            ix = (quad - izl * qpl) * 4;
        } else if (cand) decompose((uint32_t)j0, p.nx, p.ny, ix, iy, izl);
        const double Z = p.zs[p.z0 + izl];
        const double xa = p.xs[ix], xb = TILE ? p.xs[ix + 3 < p.nx ? ix + 3 : p.nx - 1] : xa;
        const uint32_t ylast = TILE ? 15u : 63u;
        const double ya = p.ys[iy], yb = p.ys[iy + ylast < p.ny ? iy + ylast : p.ny - 1];
        uint32_t need = 0;                                       // cameras still to test voxel by voxel
        bool full = false;
        if (BOX == 2 && TILE && p.live) {                        // brick level first (see lut_refine_body)
            if (cand) {
                bool lv;
                brick_bits(p, gw + lane, lv, full);
                cand = lv;
            }
            if (p.tile_whole && __ballot(cand) == 0) {
                if (lane == 0) p.groupcnt[g] = 0;
                continue;
            }
        }
        if (BOX == 2) {
            const uint64_t nwords = p.n_pad >> 6;
            for (uint32_t q0 = 0; q0 < p.C && __ballot(cand && !full) != 0; q0 += 4) {      // four cameras' boxes in flight
                uint64_t bb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    bb[k] = (q0 + k < p.C && cand && !full) ? (TILE ? p.tbox : p.bbox)[(size_t)ord(s_order, q0 + k) * nwords + gw + lane] : 0ull;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (q0 + k < p.C && cand && !full) {
                        const uint32_t r = box_test(s_grid, load_gridcam(s_grid, ord(s_order, q0 + k)), bb[k], gshift);
                        cand = r != 0;
                        if (r == 1) need |= 1u << (q0 + k);
                    }
                }
            }
        }
        for (uint32_t q = 0; BOX != 2 && q < p.C && __ballot(cand) != 0; ++q) {
            const uint32_t c = ord(s_order, q);
            if (cand) {
                const uint64_t bb = BOX == 1 ? segment_box_f32(p.cam[c], xa, xb, ya, yb, Z, p.H, p.W)
                                             : segment_box(p.cam[c], xa, xb, ya, yb, Z, p.H, p.W);
                uint32_t r = 1;
                if (bb != kMaybeBox) {
                    r = box_test(s_grid, load_gridcam(s_grid, c), bb, gshift);
                }
                cand = r != 0;
                if (r == 1) need |= 1u << q;
            }
        }
        // ---- fine: lanes = the 64 voxels of one candidate word, exact float64 test for the cameras in
        // its mask; a candidate with an empty mask is final (every voxel passes every camera)
        uint64_t mine = (cand && need == 0) ? ~0ull : 0ull;
        uint64_t nz = __ballot(cand && need != 0);
        while (nz != 0) {                                         // wave-uniform
            const uint32_t l = (uint32_t)__builtin_ctzll(nz);
            nz &= nz - 1;
            const uint32_t nd = (uint32_t)__builtin_amdgcn_readlane((int)need, (int)l);
            const uint32_t wix = (uint32_t)__builtin_amdgcn_readlane((int)ix, (int)l);
            const uint32_t wiy = (uint32_t)__builtin_amdgcn_readlane((int)iy, (int)l);
            const uint32_t wiz = (uint32_t)__builtin_amdgcn_readlane((int)izl, (int)l);
            const double VX = p.xs[TILE ? wix + (lane >> 4) : wix], VY = p.ys[TILE ? wiy + (lane & 15u) : wiy + lane];
            const double VZ = p.zs[p.z0 + wiz];
            bool alive = true;
            for (uint32_t q = 0; q < p.C; ++q) {
                if (!((nd >> q) & 1u)) continue;                  // decided for the whole word by its box
                const uint32_t c = ord(s_order, q);
                if (alive) {
                    double u, v;
                    project_point(p.cam[c], VX, VY, VZ, u, v);
                    const int32_t off = pixel_offset(u, v, p.H, p.W);
                    alive = off >= 0 && mask_bit(p.maskbits + (size_t)c * p.mwords, off);
                }
                if (__ballot(alive) == 0) break;
            }
            const uint64_t nb = __ballot(alive);
            if (lane == l) mine = nb;
        }
        if ((!TILE || p.tile_whole) && __ballot(mine != 0) == 0) {   // dead group: count only (see lut_refine_body)
            if (lane == 0) p.groupcnt[g] = 0;
            continue;
        }
        if (TILE) {
            tile_store(p, g, gw, lane, mine);
            continue;
        }
        p.words[gw + lane] = mine;
        uint32_t cnt = (uint32_t)__popcll(mine);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        if (lane == 0) p.groupcnt[g] = cnt;
    }
}

// ---------------------------------------------------------------- LUT build
// create_lookup_table: every camera, every voxel of the slab.  Also reduces, per camera and
// 64-voxel word, the bounding box of the pixels its in-image voxels land on (u16 x 4); the
// hierarchical carve rejects whole words whose box holds no foreground.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(v, d); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(v, d); v = o > v ? o : v; }
    return v;
}

// Inclusive prefix sum over the 64 lanes (all of them active), with data-parallel-primitive moves: row-wise Hillis-Steele, then
// the totals of the rows before (the same steps as wave_sum_u32; six ds_bpermute in a row would cost ~700 cycles of latency).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane)
{
    (void)lane;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// TILE: thread t is element t % 64 of tile word t / 64 (4 x-rows x 16 y) instead of linear voxel t, so the
// boxes (and the table, when one is asked for) come out in tile order.  lut or bbox may be null: the
// table-free kernel only wants the boxes.
template <bool TILE>
__global__ __launch_bounds__(kBlock) void k_build_lut(const CarveParams p, int32_t *__restrict__ lut,
                                                      uint64_t *__restrict__ bbox)
{
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;      // grid covers n_pad exactly
    const bool valid = j < p.n;
    double X = 0, Y = 0, Z = 0;
    if (valid) {
        uint32_t ix, iy, izl;
        if (TILE) {
            const uint64_t T = j >> 6;
            const uint32_t e = (uint32_t)j & 63u;
            const uint32_t quad = (uint32_t)(T / p.tq), qpl = p.nx >> 2;
            iy = ((uint32_t)T - quad * p.tq) * 16 + (e & 15u);
            izl = quad / qpl;
            ix = (quad - izl * qpl) * 4 + (e >> 4);
        } else decompose((uint32_t)j, p.nx, p.ny, ix, iy, izl);
        X = p.xs[ix]; Y = p.ys[iy]; Z = p.zs[p.z0 + izl];
    }
    const uint64_t nwords = p.n_pad >> 6;
    for (uint32_t c = 0; c < p.C; ++c) {
        int32_t off = -1;                             // padding: never inside any image
        if (valid) {
            double u, v;
            project_point(p.cam[c], X, Y, Z, u, v);
            off = pixel_offset(u, v, p.H, p.W);
        }
        if (lut) lut[(size_t)c * p.n_pad + j] = off;
        const uint32_t pv = off >= 0 ? (uint32_t)off / p.W : 0u;
        const uint32_t pu = off >= 0 ? (uint32_t)off - pv * p.W : 0u;
        const uint32_t u0 = wave_min_u32(off >= 0 ? pu : 0xffffu), u1 = wave_max_u32(pu);
        const uint32_t v0 = wave_min_u32(off >= 0 ? pv : 0xffffu), v1 = wave_max_u32(pv);
        const bool inside = __ballot(off >= 0) == ~0ull;          // every voxel of the word lands in the image
        if (bbox && (threadIdx.x & 63u) == 0)
            bbox[(size_t)c * nwords + (j >> 6)] =
                (u0 == 0xffffu) ? kEmptyBox
                                : ((uint64_t)u0 | ((uint64_t)v0 << 16) | ((uint64_t)u1 << 32) | ((uint64_t)v1 << 48) |
                                   (inside ? kBoxAllInside : 0ull));
    }
}

// A table handed in from outside (vc_upload_lut: y-major int32 [C][n_pad], what vc_fetch_lut gives out) is adopted into the
// layout the kernels read.  TILE: element e of tile word T is row r = e / 16 of its row quad, y = 16 * ty + e % 16 -- the table
// is permuted into tile order and the tile words' pixel boxes are reduced from it; else only the y-line words' boxes are.
// A foreign table cannot point outside the masks, and -1 is the only negative value the kernels know ("outside the image"; the
// per-voxel level keeps -2 for itself): anything outside [-1, H*W) is stored as -1 -- into lut_tile (TILE), or back over the
// entry itself (y-major: `lut` and `lut_out` are the same buffer, every thread rewrites only what it read).
template <bool TILE>
__global__ __launch_bounds__(kBlock) void k_adopt_lut(const CarveParams p, const int32_t *lut,
                                                      int32_t *lut_out, uint64_t *__restrict__ box)
{
    int32_t *lut_tile = lut_out;
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;      // grid covers n_pad exactly
    const uint64_t T = t >> 6;
    const uint32_t e = (uint32_t)t & 63u, r = e >> 4, yy = e & 15u;
    const bool valid = t < p.n;
    uint64_t j = t;
    if (TILE && valid) {
        const uint32_t quad = (uint32_t)(T / p.tq), ty = (uint32_t)(T - (uint64_t)quad * p.tq);
        const uint32_t qpl = p.nx >> 2;
        const uint32_t izl = quad / qpl, qx = quad - izl * qpl;
        j = ((uint64_t)izl * p.nx + qx * 4 + r) * p.ny + (uint64_t)ty * 16 + yy;
    }
    const uint64_t nwords = p.n_pad >> 6;
    for (uint32_t c = 0; c < p.C; ++c) {
        int32_t off = valid ? lut[(size_t)c * p.n_pad + j] : -1;
        const int32_t raw = off;
        if (off >= (int32_t)(p.H * p.W) || off < -1) off = -1;
        if (TILE) lut_tile[(size_t)c * p.n_pad + t] = off;
        else if (valid && off != raw) lut_out[(size_t)c * p.n_pad + j] = off;
        const uint32_t pv = off >= 0 ? (uint32_t)off / p.W : 0u;
        const uint32_t pu = off >= 0 ? (uint32_t)off - pv * p.W : 0u;
        const uint32_t u0 = wave_min_u32(off >= 0 ? pu : 0xffffu), u1 = wave_max_u32(pu);
        const uint32_t v0 = wave_min_u32(off >= 0 ? pv : 0xffffu), v1 = wave_max_u32(pv);
        const bool inside = __ballot(off >= 0) == ~0ull;
        if (e == 0)
            box[(size_t)c * nwords + T] =
                (u0 == 0xffffu) ? kEmptyBox
                                : ((uint64_t)u0 | ((uint64_t)v0 << 16) | ((uint64_t)u1 << 32) | ((uint64_t)v1 << 48) |
                                   (inside ? kBoxAllInside : 0ull));
    }
}

__global__ __launch_bounds__(kBlock) void k_project(const CamDev cam, const double *__restrict__ xyz,
                                                    uint64_t n, double *__restrict__ uv)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double u, v;
    project_point(cam, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], u, v);
    uv[2 * i] = u;
    uv[2 * i + 1] = v;
}

// ---------------------------------------------------------------- brick level
// A brick is 16 x 16 x 16 voxels = 4 row quads x 1 tile column x 16 layers = 64 tile words; brick number
// b = (bz * nbx + bx) * tq + by, so the 64 bricks of one brick row along y are one u64 of the bit maps.
// The bricks' boxes are built by k_brick_boxes_bm (below).
// k_cull (every frame set, in front of the hierarchical carve): lane = brick, every camera's brick box against that
// camera's block grids (LDS).  No foreground block in some camera's box: no voxel of the brick can survive (exact: the box
// contains the pixel of every voxel of every word of the brick) -- its 64 words are never looked at.  Every block of every
// camera's box foreground and every voxel inside every image: all 4096 voxels survive, equally without a look.
__global__ __launch_bounds__(kBlock) void k_cull(const CarveParams p)
{
    extern __shared__ uint32_t s_grid[];
    stage_grids(s_grid, p.blockgrid);
    __shared__ uint32_t s_order[kMaxCameras];
    stage_order(p.counts, p.C, s_order);
    const uint32_t gshift = hdr_u32(s_grid, kHdrShift);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (kBlock / 64);
    const uint32_t nw = p.nbrick_pad >> 6;
    for (uint32_t w = wave0; w < nw; w += nwaves) {
        const uint32_t b = w * 64 + lane;
        bool cand = true, full = true;
        for (uint32_t q0 = 0; q0 < p.C && __ballot(cand) != 0; q0 += 4) {
            uint64_t bb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                bb[k] = (q0 + k < p.C && cand) ? p.kbox[(size_t)ord(s_order, q0 + k) * p.nbrick_pad + b] : 0ull;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (q0 + k < p.C && cand) {
                    const uint32_t r = box_test(s_grid, load_gridcam(s_grid, ord(s_order, q0 + k)), bb[k], gshift);
                    cand = r != 0;
                    full = full && r == 2;
                }
            }
        }
        const uint64_t lw = __ballot(cand), fw = __ballot(cand && full);
        if (lane == 0) { p.live[w] = lw; p.live[nw + w] = fw; }
    }
}

// ---------------------------------------------------------------- brick pipeline
// The default carve for ny in {256, 512, 1024, 2048, 4096} (groups of 4096 consecutive voxels then lie inside one brick column).
// Culling whole bricks leaves a wave of the group-wise kernels with a handful of live lanes (the hull crosses a line of
// 1024 voxels along y in a few bricks only), and what is left is latency: a wave walks through entry -> boxes -> table ->
// mask, one dependent round trip after the other, for a few words.  So the work is re-cut by what it needs, one
// launch per level, each a flat list of equal pieces, every access contiguous:
//
//  k_cull_bricks   lane = brick: every camera's brick box against the block grids (k_cull).  Appends the bricks that are
//                  neither dead nor full to the BRICK LIST and the brick columns with a live brick to the COLUMN LIST
//                  (one atomic per wave and list), zeroes the survivor counts of all groups.
//  k_brick_words   one wave per listed brick, lane (q, l) = its tile word of row quad q, layer l.  The word boxes come from
//                  a brick-major copy (512 contiguous bytes per camera).  Decided words (dead / all alive) are stored to the
//                  brick-major word buffer; undecided ones go to the WORD LIST with the cameras that still have to look.
//  k_voxel_words   one wave per 8 listed words, lanes = the 64 voxels of a word: table entry (LUT) or float64 projection,
//                  mask bit, for the listed cameras, two cameras per dependent round trip (PAIR; one above 4 cameras).  No
//                  block grids: no LDS to fill.  Table entries are loaded non-temporally (read once per step).
//  k_assemble      one wave per group of the listed columns: collects the group's 64 tile words (dead brick: 0, full
//                  brick: all ones, else the brick-major buffer), turns them into y-major words (tile_store) and stores
//                  words + count.  Groups of unlisted columns keep count 0 and nobody reads their words.
//
// List lengths stay on the device; the host sizes each launch from the lengths of an EARLIER step (a page-locked
// word the kernels write, read without any synchronisation), the waves stride over whatever the real length is.
// The lists are SHARDED: a returning atomic on one address takes ~12 ns, so 6 000 waves reserving their list space from a
// single counter would serialise for longer than their work.  Shard s (of kShards) has its own counter and its own region
// [s * cap, (s + 1) * cap) with cap = the most its producers (the waves w with w % kShards == s) can ever append, so no
// shard can overflow.  A consumer wave scans the kShards counts once (lane = shard) and finds the shard of flat item t by
// a ballot over the exclusive prefix.
constexpr uint32_t kShards = 64;
constexpr uint32_t kWideBlock = 1024;  // k_cull_bricks / k_brick_words run 256 or 1024 threads per workgroup (large grids: 16 waves share one LDS copy)
constexpr uint32_t kShardStride = 32;   // u32 between two shard counters: one 128-byte line each (atomics on one LINE serialise too)
// Work counters of the detail runs (vc_timing_t::work): one wave-level add per wave and kind, on kShards lines of their own.
constexpr uint32_t kStatStride = 16;    // u64 between two shards of a counter: one 128-byte line each
__device__ __forceinline__ void stat_add(unsigned long long *stats, uint32_t kind, uint32_t wave, uint32_t lane, uint64_t v)
{
    if (stats && lane == 0 && v) atomicAdd(&stats[((size_t)kind * kShards + (wave % kShards)) * kStatStride], (unsigned long long)v);
}
struct BrickLists {
    uint32_t *counters;         // [2][3][kShards * kShardStride]: bricks, columns, words, of this parity; k_cull_bricks zeroes the other set
    uint32_t *bricks;           // brick numbers (live, not full)
    uint32_t *columns;          // column numbers (bz * nbx + bx) with a live brick
    uint64_t *words;            // undecided words: tile word T | need mask (by camera NUMBER) << 32
    uint64_t *bm;               // [n_pad / 64] the tile words' results, tile order (k_assemble reads 512 contiguous bytes per group)
    const uint64_t *wbox;       // [C][nbrick_pad * 64] brick-major word boxes
    uint32_t *host_counts;      // page-locked [4]
    uint32_t parity;
    uint32_t cap_b, cap_c, cap_w;   // shard capacities of the three lists (entries)
};

// producer: space for popcount(mask) entries in shard `shard`; returns this lane's entry index (valid where its mask bit is set)
__device__ __forceinline__ uint32_t shard_append(uint32_t *counts, uint32_t cap, uint32_t shard, uint64_t mask, uint32_t lane)
{
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counts[shard * kShardStride], (uint32_t)__popcll(mask));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    return shard * cap + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
// consumer: lane s gets the exclusive prefix of the shard sizes (in units of `per`, rounded up per shard) and its own size
struct ShardView { uint32_t start, size, total; };
__device__ __forceinline__ ShardView shard_view(const uint32_t *counts, uint32_t per, uint32_t lane)
{
    ShardView v;
    v.size = counts[lane * kShardStride];
    const uint32_t units = (v.size + per - 1) / per;
    const uint32_t incl = wave_inclusive_scan(units, lane);
    v.start = incl - units;
    v.total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    return v;
}
// unit t (wave-uniform, < total) -> its shard and its unit number inside the shard
__device__ __forceinline__ void shard_locate(const ShardView &v, uint32_t t, uint32_t &shard, uint32_t &within, uint32_t &size)
{
    shard = (uint32_t)__popcll(__ballot(v.start <= t)) - 1u;
    within = t - (uint32_t)__builtin_amdgcn_readlane((int)v.start, (int)shard);
    size = (uint32_t)__builtin_amdgcn_readlane((int)v.size, (int)shard);
}

__global__ __launch_bounds__(kWideBlock) void k_cull_bricks(const CarveParams p, const BrickLists bl, uint32_t ngroups)
{
    if (p.dbg & 72u) return;                                       // experiment: launch only (scripts/exp_streams.py)
    extern __shared__ uint32_t s_grid[];
    const uint32_t *gsrc = p.coarsegrid ? p.coarsegrid : p.blockgrid;
    // (the launch sized its LDS from a bound on the coarse grids that oddly shaped crops can exceed: then nothing is staged and every
    // brick is listed for the word level, which decides it all the same -- slower, never wrong)
    const bool staged = p.cull_lds_words == 0 || hdr_u32(gsrc, kHdrWords) + 8u <= p.cull_lds_words;
    if (staged) stage_grids(s_grid, gsrc);
    __shared__ uint32_t s_order[kMaxCameras];
    stage_order(p.counts, p.C, s_order);
    if (blockIdx.x == 0)                                             // (any workgroup size: 192 counters)
        for (uint32_t i = threadIdx.x; i < 3 * kShards; i += blockDim.x) bl.counters[((bl.parity ^ 1u) * 3 * kShards + i) * kShardStride] = 0;
    uint32_t *cnt = bl.counters + bl.parity * 3 * kShards * kShardStride;
    const uint32_t gshift = staged ? hdr_u32(s_grid, kHdrShift) : 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x / 64);
    const uint32_t nw = p.nbrick_pad >> 6;
    const uint32_t nbricks = p.nbx * p.tq * p.nbz;
    for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x); i < ngroups; i += gridDim.x * blockDim.x) {
        p.groupcnt[i] = 0;
        if (p.groupnz) p.groupnz[i] = 0;
    }
    // a wave takes 64 bricks at a time; a brick column has tq of them: 64 / tq whole columns per wave (ny <= 1024), or one column
    // in tq / 64 rounds (ny = 2048, 4096), so that a column is listed once, by one wave
    const uint32_t ipw = p.tq > 64u ? p.tq / 64u : 1u;
    uint64_t nstat = 0;
    for (uint32_t W = wave0; W * ipw < nw; W += nwaves) {
        uint64_t colany = 0;
        for (uint32_t it = 0; it < ipw; ++it) {
            const uint32_t w = W * ipw + it;
            const uint32_t b = w * 64 + lane;
            bool cand = b < nbricks, full = staged;
            for (uint32_t q0 = 0; q0 < p.C && __ballot(cand) != 0 && staged; q0 += 4) {
                uint64_t bb[4];
                if (p.stats) nstat += (uint64_t)__popcll(__ballot(cand)) * (p.C - q0 < 4u ? p.C - q0 : 4u);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    bb[k] = (q0 + k < p.C && cand) ? p.kbox[(size_t)ord(s_order, q0 + k) * p.nbrick_pad + b] : 0ull;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (q0 + k < p.C && cand) {
                        const uint32_t r = box_test(s_grid, load_gridcam(s_grid, ord(s_order, q0 + k)), bb[k], gshift);
                        cand = r != 0;
                        full = full && r == 2;
                    }
                }
            }
            const uint64_t lw = __ballot(cand), fw = __ballot(cand && full);
            p.live[w] = lw;                                       // (all lanes, same value: one store)
            p.live[nw + w] = fw;
            colany |= lw;
            // bricks to look into
            const uint64_t bm = lw & ~fw;
            if (bm) {
                const uint32_t at = shard_append(cnt, bl.cap_b, W % kShards, bm, lane);
                if ((bm >> lane) & 1ull) bl.bricks[at] = b;
            }
        }
        if (colany == 0) continue;                                // (wave-uniform)
        // columns with a live brick; lane j < 64 / tq speaks for column j of the wave (lane 0 for the only one)
        const uint32_t ncol = p.tq > 64u ? 1u : 64u / p.tq;
        const uint64_t colbits = p.tq >= 64u ? colany : (colany >> ((lane < ncol ? lane : 0u) * p.tq)) & ((1ull << p.tq) - 1ull);
        const bool cwant = lane < ncol && colbits != 0;
        const uint64_t cm = __ballot(cwant);
        const uint32_t cat = shard_append(cnt + kShards * kShardStride, bl.cap_c, W % kShards, cm, lane);
        if (cwant) bl.columns[cat] = p.tq > 64u ? W : (W * 64) / p.tq + lane;
    }
    stat_add(p.stats, 4 /* VC_WORK_BRICK_BOXES */, wave0, lane, nstat);
}

// Once per grid / slab / camera set: the tile words' pixel boxes in brick-major order (brick b, lane (q, l): row quad
// 4 bx + q, layer 16 bz + l, tile column by), and from them the bricks' boxes (union; "every voxel inside the image"
// only if every word has it).  One wave per brick.
__global__ __launch_bounds__(kBlock) void k_brick_boxes_bm(const CarveParams p, const uint64_t *__restrict__ tbox,
                                                           uint64_t *__restrict__ wbox, uint64_t *__restrict__ kbox)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (b >= p.nbrick_pad) return;
    const uint32_t qpl = p.nx >> 2, nzl = (uint32_t)(p.n / ((uint64_t)p.nx * p.ny));
    const uint32_t col = b / p.tq, by = b - col * p.tq, bz = col / p.nbx, bx = col - bz * p.nbx;
    const uint32_t qx = 4 * bx + (lane >> 4), izl = 16 * bz + (lane & 15u);
    const bool valid = bz < p.nbz && qx < qpl && izl < nzl;
    const uint64_t nwords = p.n_pad >> 6;
    for (uint32_t c = 0; c < p.C; ++c) {
        const uint64_t w = valid ? tbox[(size_t)c * nwords + ((uint64_t)izl * qpl + qx) * p.tq + by] : kEmptyBox;
        wbox[((size_t)c * p.nbrick_pad + b) * 64 + lane] = w;
        const bool some = w != kEmptyBox;
        const uint32_t u0 = wave_min_u32(some ? (uint32_t)(w & 0xffffu) : 0xffffu), v0 = wave_min_u32(some ? (uint32_t)((w >> 16) & 0xffffu) : 0xffffu);
        const uint32_t u1 = wave_max_u32(some ? (uint32_t)((w >> 32) & 0xffffu) : 0u), v1 = wave_max_u32(some ? (uint32_t)((w >> 48) & 0x7fffu) : 0u);
        // a word that does not exist (grid edge) does not spoil "inside"; one whose voxels all miss the image does
        const bool inside = __ballot(valid && !(some && (w >> 63))) == 0;
        const bool none = __ballot(some) == 0;
        if (lane == 0)
            kbox[(size_t)c * p.nbrick_pad + b] = none ? kEmptyBox
                : ((uint64_t)u0 | ((uint64_t)v0 << 16) | ((uint64_t)u1 << 32) | ((uint64_t)v1 << 48) | (inside ? kBoxAllInside : 0ull));
    }
}

// DUAL: the 1024-thread form for block grids that fill the LDS (one workgroup per compute unit, four waves per SIMD, 128 registers each).  At that
// occupancy nothing hides a brick's chain of rounds {four cameras' boxes, four box tests}: a wave takes TWO listed bricks at a time and
// interleaves their rounds, so that one brick's box loads are under way while the other's boxes are tested.
template <int NB>
__device__ __forceinline__ void brick_words_body(const CarveParams &p, const BrickLists &bl, uint32_t *s_grid)
{
    uint32_t *cnt = bl.counters + bl.parity * 3 * kShards * kShardStride;
    const ShardView sv = shard_view(cnt, 1, threadIdx.x & 63u);
    const uint32_t nlist = sv.total;
    if (blockIdx.x == 0 && threadIdx.x == 0) bl.host_counts[0] = nlist;
    if (blockIdx.x * (blockDim.x / 64) >= nlist) return;              // fewer bricks than waves launched
    const bool passall = (p.dbg & 4u) != 0;                       // experiment: no word-level tests, every word of a listed brick goes to the voxel level
    stage_grids(s_grid, p.blockgrid);
    __shared__ uint32_t s_order[kMaxCameras];
    stage_order(p.counts, p.C, s_order);
    const uint32_t gshift = passall ? 0u : hdr_u32(s_grid, kHdrShift);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x / 64);
    const uint32_t qpl = p.nx >> 2, nzl = (uint32_t)(p.n / ((uint64_t)p.nx * p.ny));
    const bool tests = !(p.dbg & 6u);
    uint64_t nstat = 0;
    for (uint32_t t0 = wave0; t0 < nlist; t0 += NB * nwaves) {
        uint32_t b[NB], need[NB];
        bool cand[NB], exists[NB], have[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint32_t t = t0 + (uint32_t)i * nwaves;
            have[i] = t < nlist;                                   // (wave-uniform)
            uint32_t shard, within, ssize;
            shard_locate(sv, have[i] ? t : t0, shard, within, ssize);
            b[i] = hdr_u32(bl.bricks, shard * bl.cap_b + within);
            const uint32_t col = b[i] / p.tq, bz = col / p.nbx, bx = col - bz * p.nbx;     // wave-uniform
            exists[i] = have[i] && 4 * bx + (lane >> 4) < qpl && 16 * bz + (lane & 15u) < nzl;
            cand[i] = exists[i];
            need[i] = 0;                                           // by camera NUMBER (k_voxel_words has no use for the order)
        }
        for (uint32_t q0 = 0; q0 < p.C && tests; q0 += 4) {
            bool any = false;
#pragma unroll
            for (int i = 0; i < NB; ++i) any = any || __ballot(cand[i]) != 0;
            if (!any) break;
            uint64_t bb[NB][4];
            uint32_t cn[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cn[k] = q0 + k < p.C ? ord(s_order, q0 + k) : 0u;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (p.stats) nstat += (uint64_t)__popcll(__ballot(cand[i])) * (p.C - q0 < 4u ? p.C - q0 : 4u);
                const size_t slot = (size_t)b[i] * 64 + lane;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    bb[i][k] = (q0 + k < p.C && cand[i]) ? bl.wbox[(size_t)cn[k] * p.nbrick_pad * 64 + slot] : 0ull;
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (q0 + k < p.C && cand[i]) {
                        const uint32_t r = box_test(s_grid, load_gridcam(s_grid, cn[k]), bb[i][k], gshift);
                        cand[i] = r != 0;
                        if (r == 1) need[i] |= 1u << cn[k];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (!have[i]) continue;
            const uint32_t col = b[i] / p.tq, by = b[i] - col * p.tq, bz = col / p.nbx, bx = col - bz * p.nbx;
            const uint32_t qx = 4 * bx + (lane >> 4), izl = 16 * bz + (lane & 15u);
            if (!cand[i]) need[i] = 0;
            else if (passall) need[i] = (1u << p.C) - 1u;
            const uint64_t T = ((uint64_t)(exists[i] ? izl : 0u) * qpl + (exists[i] ? qx : 0u)) * p.tq + by;
            if (exists[i] && (need[i] == 0 || (p.dbg & 1u))) bl.bm[T] = cand[i] ? ~0ull : 0ull;     // decided here
            const uint64_t um = (p.dbg & 1u) ? 0ull : __ballot(need[i] != 0);
            if (um) {
                const size_t o = shard_append(cnt + 2 * kShards * kShardStride, bl.cap_w, (t0 + (uint32_t)i * nwaves) % kShards, um, lane);
                if (need[i]) bl.words[o] = T | ((uint64_t)need[i] << 32);
            }
        }
    }
    stat_add(p.stats, 0 /* VC_WORK_WORD_BOXES */, wave0, lane, nstat);
}

// The 1024-thread form where the LDS has 256 more words per wave (CarveParams::compact_off) and a camera mask fits in 24 bits.
// Most words meet the camera that rejects them in the first round (the cameras are asked in order of emptiness) and the rest of
// a brick's rounds would run with a few lanes alive.  The words of four listed bricks -- {brick i, lane, need} in 32 bits --
// are packed into the wave's LDS words, tested 64 NP at a time, side by side (box_test_rows_n), the survivors packed again in
// place after every round; at the end they are scattered back to their own lanes, so that the stores stay whole rows.
// One copy of the test code for all rounds (the unrolled lockstep form of four bricks is 10 000 instructions: more than the
// instruction cache).  One wave's LDS operations execute in order; the fences are for the compiler.
template <int NP>
__device__ __forceinline__ void brick_words_compact(const CarveParams &p, const BrickLists &bl, uint32_t *s_grid)
{
    constexpr int NB = 4;
    uint32_t *cnt = bl.counters + bl.parity * 3 * kShards * kShardStride;
    const ShardView sv = shard_view(cnt, 1, threadIdx.x & 63u);
    const uint32_t nlist = sv.total;
    if (blockIdx.x == 0 && threadIdx.x == 0) bl.host_counts[0] = nlist;
    if (blockIdx.x * (blockDim.x / 64) >= nlist) return;
    const bool passall = (p.dbg & 4u) != 0;
    stage_grids(s_grid, p.blockgrid);
    __shared__ uint32_t s_order[kMaxCameras];
    stage_order(p.counts, p.C, s_order);
    const uint32_t gshift = passall ? 0u : hdr_u32(s_grid, kHdrShift);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t below = (1ull << lane) - 1ull;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x / 64);
    const uint32_t qpl = p.nx >> 2, nzl = (uint32_t)(p.n / ((uint64_t)p.nx * p.ny));
    const bool tests = !(p.dbg & 6u);
    uint32_t *scr = s_grid + p.compact_off + (threadIdx.x >> 6) * 256u;
    uint64_t nstat = 0;
    for (uint32_t t0 = wave0; t0 < nlist; t0 += NB * nwaves) {
        uint32_t b[NB];
        bool exists[NB], have[NB];
        uint32_t n = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint32_t t = t0 + (uint32_t)i * nwaves;
            have[i] = t < nlist;                                   // (wave-uniform)
            uint32_t shard, within, ssize;
            shard_locate(sv, have[i] ? t : t0, shard, within, ssize);
            b[i] = hdr_u32(bl.bricks, shard * bl.cap_b + within);
            const uint32_t col = b[i] / p.tq, bz = col / p.nbx, bx = col - bz * p.nbx;
            exists[i] = have[i] && 4 * bx + (lane >> 4) < qpl && 16 * bz + (lane & 15u) < nzl;
            const uint64_t m = __ballot(exists[i]);
            if (exists[i]) scr[n + (uint32_t)__popcll(m & below)] = ((uint32_t)i << 6) | lane;
            n += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t q0 = 0; q0 < p.C && n != 0u && tests; q0 += 4) {
            uint32_t cn[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cn[k] = q0 + k < p.C ? ord(s_order, q0 + k) : 0u;
            uint32_t nout = 0;
#pragma nounroll
            for (uint32_t base = 0; base < n; base += 64 * NP) {
                uint32_t e[NP], nd[NP]; bool c[NP]; uint64_t bb[4][NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    c[u] = base + 64u * u + lane < n;
                    e[u] = c[u] ? scr[base + 64u * u + lane] : 0u;
                    nd[u] = e[u] >> 8;
                    const uint32_t bi = (e[u] >> 6) & 3u;
                    const uint32_t bsel = bi == 0u ? b[0] : bi == 1u ? b[1] : bi == 2u ? b[2] : b[3];
                    const size_t slot = (size_t)bsel * 64 + (e[u] & 63u);
                    if (p.stats) nstat += (uint64_t)__popcll(__ballot(c[u])) * (p.C - q0 < 4u ? p.C - q0 : 4u);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        bb[k][u] = (q0 + k < p.C && c[u]) ? bl.wbox[(size_t)cn[k] * p.nbrick_pad * 64 + slot] : 0ull;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (q0 + k >= p.C) break;
                    bool some = false;
#pragma unroll
                    for (int u = 0; u < NP; ++u) some = some || c[u];
                    if (__ballot(some) == 0ull) break;
                    uint32_t r[NP];
                    box_test_rows_n<NP>(s_grid, load_gridcam(s_grid, cn[k]), bb[k], c, gshift, r);
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        if (c[u] && r[u] == 1u) nd[u] |= 1u << cn[k];
                        c[u] = c[u] && r[u] != 0u;
                    }
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const uint64_t m = __ballot(c[u]);              // (the survivors land at or below what this trip has read)
                    if (c[u]) scr[nout + (uint32_t)__popcll(m & below)] = (e[u] & 255u) | (nd[u] << 8);
                    nout += (uint32_t)__popcll(m);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            n = nout;
        }
        uint32_t ent[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) ent[i] = 64u * i + lane < n ? scr[64u * i + lane] : 0xffffffffu;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int i = 0; i < NB; ++i) scr[64u * i + lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int i = 0; i < NB; ++i) if (ent[i] != 0xffffffffu) scr[ent[i] & 255u] = (ent[i] >> 8) | 0x80000000u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t res[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) res[i] = scr[64u * i + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (!have[i]) continue;
            const bool cand = exists[i] && (res[i] >> 31);
            uint32_t need = cand ? res[i] & 0x7fffffffu : 0u;
            if (cand && passall) need = (1u << p.C) - 1u;
            const uint32_t col = b[i] / p.tq, by = b[i] - col * p.tq, bz = col / p.nbx, bx = col - bz * p.nbx;
            const uint32_t qx = 4 * bx + (lane >> 4), izl = 16 * bz + (lane & 15u);
            const uint64_t T = ((uint64_t)(exists[i] ? izl : 0u) * qpl + (exists[i] ? qx : 0u)) * p.tq + by;
            if (exists[i] && (need == 0 || (p.dbg & 1u))) bl.bm[T] = cand ? ~0ull : 0ull;       // decided here
            const uint64_t um = (p.dbg & 1u) ? 0ull : __ballot(need != 0);
            if (um) {
                const size_t o = shard_append(cnt + 2 * kShards * kShardStride, bl.cap_w, (t0 + (uint32_t)i * nwaves) % kShards, um, lane);
                if (need) bl.words[o] = T | ((uint64_t)need << 32);
            }
        }
    }
    stat_add(p.stats, 0 /* VC_WORK_WORD_BOXES */, wave0, lane, nstat);
}

__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_brick_words(const CarveParams p, const BrickLists bl)
{
    if (p.dbg & 72u) return;
    extern __shared__ uint32_t s_grid[];
    brick_words_body<1>(p, bl, s_grid);
}
__global__ __launch_bounds__(kWideBlock) void k_brick_words_wide(const CarveParams p, const BrickLists bl)
{
    if (p.dbg & 72u) return;
    extern __shared__ uint32_t s_grid[];
    if (p.compact_off) brick_words_compact<2>(p, bl, s_grid);
    else brick_words_body<4>(p, bl, s_grid);
}

// B undecided words per wave (list entries B t .. B t + B - 1), lanes = the 64 voxels of a tile word (4 x-rows x 16 y).
template <bool LUT, bool PAIR = true>
__global__ __launch_bounds__(kBlock) void k_voxel_words(const CarveParams p, const BrickLists bl)
{
    if (p.dbg & 136u) return;
    constexpr int B = 8;
    const uint32_t lane = threadIdx.x & 63u;
    const ShardView sv = shard_view(bl.counters + (bl.parity * 3 + 2) * kShards * kShardStride, B, lane);    // in batches of B words, per shard
    const uint32_t nbatch = sv.total;
    if (blockIdx.x == 0 && threadIdx.x == 0) bl.host_counts[2] = nbatch * B;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x / 64);
    const uint32_t qpl = p.nx >> 2;
    uint64_t nstat = 0;
    // lane b < B fetches entry b of a batch; everybody gets them by cross-lane reads.  A wave that takes several batches has the
    // next one's entries under way while it works on this one (one dependent round trip less per batch).
    auto fetch = [&](uint32_t t, bool &valid) -> uint64_t {
        uint32_t shard, within, ssize;
        shard_locate(sv, t < nbatch ? t : 0u, shard, within, ssize);
        valid = t < nbatch && lane < (uint32_t)B && within * B + lane < ssize;
        const size_t at = (size_t)shard * bl.cap_w + within * B + lane;
        return valid ? bl.words[at] : 0ull;
    };
    bool nvalid = false;
    uint64_t enext = nbatch ? fetch(wave0, nvalid) : 0ull;
    for (uint32_t t = wave0; t < nbatch; t += nwaves) {
        const bool mine_valid = nvalid;
        const uint64_t e0 = enext;
        enext = fetch(t + nwaves, nvalid);
        uint32_t nd[B], Tb[B];
        uint32_t alive = 0, ndany = 0;
#pragma unroll
        for (int b = 0; b < B; ++b) {
            Tb[b] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e0, b);
            nd[b] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(e0 >> 32), b);
            if (nd[b]) alive |= 1u << b;
            ndany |= nd[b];
        }
        if (LUT) {
            // cameras in pairs (two tables' entries and mask words per dependent round trip), skipping cameras no word of the batch needs
            uint32_t left = ndany;
            while (left) {
                const uint32_t c = (uint32_t)__builtin_ctz(left);
                left &= left - 1;
                uint32_t c2 = c;
                if (PAIR && left) { c2 = (uint32_t)__builtin_ctz(left); left &= left - 1; }
                const int32_t *__restrict__ L1 = p.lut_tile + (size_t)c * p.n_pad + lane;
                const int32_t *__restrict__ L2 = p.lut_tile + (size_t)c2 * p.n_pad + lane;
                const uint32_t *__restrict__ mb = p.maskbits + (size_t)c * p.mwords;
                const uint32_t *__restrict__ mb2 = p.maskbits + (size_t)c2 * p.mwords;
                int32_t off[B], off2[B];
                uint32_t mw[B], mw2[B];
#pragma unroll
                for (int b = 0; b < B; ++b) {                     // a camera outside the word's mask counts as passed
                    const bool t1 = ((nd[b] >> c) & 1u) && ((alive >> b) & 1u);
                    const bool t2 = PAIR && c2 != c && ((nd[b] >> c2) & 1u) && ((alive >> b) & 1u);
                    // (table entries are read once per step: streamed past the caches, which keep the mask bits)
                    off[b] = t1 ? __builtin_nontemporal_load(&L1[(size_t)Tb[b] * 64]) : -2;
                    off2[b] = t2 ? __builtin_nontemporal_load(&L2[(size_t)Tb[b] * 64]) : -2;
                    if (p.stats) nstat += (uint64_t)__popcll(__ballot(t1)) + (uint64_t)__popcll(__ballot(t2));
                }
                // (the value of a lane that asks nothing is set first, branch-free; ONE lane mask -- "has an entry" -- around the gather)
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    mw[b] = off[b] == -2 ? ~0u : 0u;
                    mw2[b] = (!PAIR || off2[b] == -2) ? ~0u : 0u;
                }
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    if (off[b] >= 0) mw[b] = mb[(uint32_t)off[b] >> 5];
                    if (PAIR && off2[b] >= 0) mw2[b] = mb2[(uint32_t)off2[b] >> 5];
                }
#pragma unroll
                for (int b = 0; b < B; ++b)
                    if (!((mw[b] >> ((uint32_t)off[b] & 31u)) & (mw2[b] >> ((uint32_t)off2[b] & 31u)) & 1u)) alive &= ~(1u << b);
                if (__ballot(alive != 0) == 0) break;
            }
        } else {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                if (nd[b] == 0) continue;
                const uint32_t quad = Tb[b] / p.tq, ty = Tb[b] - quad * p.tq, zl = quad / qpl, qx = quad - zl * qpl;   // wave-uniform
                const double VX = p.xs[4 * qx + (lane >> 4)], VY = p.ys[16 * ty + (lane & 15u)], VZ = p.zs[p.z0 + zl];
                bool ok = true;
                for (uint32_t left = nd[b]; left; left &= left - 1) {
                    const uint32_t c = (uint32_t)__builtin_ctz(left);
                    if (p.stats) nstat += (uint64_t)__popcll(__ballot(ok));
                    if (ok) {
                        double u, v;
                        project_point(p.cam[c], VX, VY, VZ, u, v);
                        const int32_t off = pixel_offset(u, v, p.H, p.W);
                        ok = off >= 0 && mask_bit(p.maskbits + (size_t)c * p.mwords, off);
                    }
                    if (__ballot(ok) == 0) break;
                }
                if (!ok) alive &= ~(1u << b);
            }
        }
        uint64_t mine = 0;
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const uint64_t nb = __ballot((alive >> b) & 1u);
            if (lane == (uint32_t)b) mine = nb;
        }
        if (mine_valid) bl.bm[(uint32_t)e0] = mine;
    }
    stat_add(p.stats, LUT ? 1 /* VC_WORK_TABLE_ENTRIES */ : 2 /* VC_WORK_PROJECTIONS */, wave0, lane, nstat);
}

// One wave per (listed brick column, layer): the 4 / qpg groups (4096 consecutive voxels = 64 tile words each) that lie
// side by side along x in that layer.  Their loads are issued together; the column's live / full bits are read once.
__global__ __launch_bounds__(kBlock) void k_assemble(const CarveParams p, const BrickLists bl)
{
    if (p.dbg & 264u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const ShardView sv = shard_view(bl.counters + (bl.parity * 3 + 1) * kShards * kShardStride, 1, lane);
    const uint32_t ncols = sv.total;
    if (blockIdx.x == 0 && threadIdx.x == 0) bl.host_counts[1] = ncols;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x / 64);
    const uint32_t qpl = p.nx >> 2, nzl = (uint32_t)(p.n / ((uint64_t)p.nx * p.ny));
    const uint32_t nunits = ncols * 16u;
    const uint32_t nw = p.nbrick_pad >> 6;
    if (p.tq > 64u) {
        // ny = 2048, 4096: a row quad holds tq / 64 wave loads of tile words and 2 or 4 groups (of 2 rows or 1); every word of
        // the column's layer is stored, the groups' counts are added up with atomics (tile_store, tile_whole == 0)
        const uint32_t parts = p.tq / 64u;
        for (uint32_t u = wave0; u < nunits; u += nwaves) {
            const uint32_t ci = u >> 4, l = u & 15u;              // wave-uniform
            uint32_t shard, within, ssize;
            shard_locate(sv, ci, shard, within, ssize);
            const uint32_t col = hdr_u32(bl.columns, shard * bl.cap_c + within);
            const uint32_t bz = col / p.nbx, bx = col - bz * p.nbx;
            const uint32_t izl = 16 * bz + l;
            if (izl >= nzl) continue;
            for (uint32_t h = 0; h < parts; ++h) {
                const uint32_t b = col * p.tq + 64u * h + lane;   // my brick (the same for the four row quads)
                const uint64_t lw = p.live[b >> 6], fw = p.live[nw + (b >> 6)];
                const bool live = (lw >> (b & 63u)) & 1ull, full = (fw >> (b & 63u)) & 1ull;
                uint64_t mine[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    const uint64_t gw = ((uint64_t)izl * qpl + 4 * bx + q) * p.tq + 64u * h;
                    mine[q] = (4 * bx + q < qpl && live) ? (full ? ~0ull : bl.bm[gw + lane]) : 0ull;
                }
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    if (4 * bx + q < qpl) {
                        const uint64_t gw = ((uint64_t)izl * qpl + 4 * bx + q) * p.tq + 64u * h;
                        tile_store(p, 0u, gw, lane, mine[q]);
                    }
                }
            }
        }
        return;
    }
    const uint32_t qpg = 64u / p.tq;                              // row quads per group (1, 2 or 4)
    const uint32_t gq = 4u / qpg;                                 // groups along x inside a brick column (4, 2 or 1)
    const uint32_t tqs = (uint32_t)__builtin_ctz(p.tq);           // (tq = 16, 32 or 64)
    const uint32_t dq = lane >> tqs, ty = lane & (p.tq - 1u);     // my tile word inside a group: row quad dq, tile column ty
    for (uint32_t u = wave0; u < nunits; u += nwaves) {
        const uint32_t ci = u >> 4, l = u & 15u;                  // wave-uniform
        uint32_t shard, within, ssize;
        shard_locate(sv, ci, shard, within, ssize);
        const uint32_t col = hdr_u32(bl.columns, shard * bl.cap_c + within);
        const uint32_t bz = col / p.nbx, bx = col - bz * p.nbx;
        const uint32_t izl = 16 * bz + l;
        if (izl >= nzl) continue;
        const uint32_t b = col * p.tq + ty;                       // my brick (the same for every group of the column)
        const uint64_t lw = p.live[b >> 6], fw = p.live[nw + (b >> 6)];
        const bool live = (lw >> (b & 63u)) & 1ull, full = (fw >> (b & 63u)) & 1ull;
        uint64_t mine[4];
#pragma unroll
        for (uint32_t xg = 0; xg < 4; ++xg) {
            const uint32_t qx0 = 4 * bx + xg * qpg;               // first row quad of group xg
            const uint64_t gw = ((uint64_t)izl * qpl + qx0) * p.tq;
            mine[xg] = (xg < gq && qx0 + dq < qpl && live) ? (full ? ~0ull : bl.bm[gw + lane]) : 0ull;
        }
        // (tq divides 64 here: the wave's 64 tile words ARE group gw / 64.  All four groups are transposed and counted before
        // anything is stored: a store between two transposes makes the compiler wait for it)
        uint64_t out[4];
        uint32_t cnt[4];
#pragma unroll
        for (uint32_t xg = 0; xg < 4; ++xg) {
            out[xg] = tile_transpose(mine[xg], lane);
            cnt[xg] = wave_sum_u32((uint32_t)__popcll(mine[xg]));
        }
#pragma unroll
        for (uint32_t xg = 0; xg < 4; ++xg) {
            const uint32_t qx0 = 4 * bx + xg * qpg;
            if (xg < gq && qx0 < qpl) {
                const uint64_t gw = ((uint64_t)izl * qpl + qx0) * p.tq;   // first tile word of the group = 64 x its group number
                // lane (r, k) holds row r, y chunk k of the group: tile words 4 k .. 4 k + 3 = row quad qx0 + 4 k / tq, tile column
                // 4 k % tq (tq is a power of two here)
                const uint32_t k4 = 4u * (lane & 15u), dqk = k4 >> tqs, tyk = k4 & (p.tq - 1u);
                if (qx0 + dqk < qpl)
                    p.words[(((uint64_t)izl * p.nx + (qx0 + dqk) * 4 + (lane >> 4)) * p.ny + (uint64_t)tyk * 16) >> 6] = out[xg];
                const uint64_t nzw = __ballot(out[xg] != 0ull);   // (lane (r, k) holds y-major word 16 r + k of the group: all 64 of them)
                if (lane == 0) {
                    p.groupcnt[gw >> 6] = cnt[xg];
                    if (p.groupnz) p.groupnz[gw >> 6] = (uint32_t)__popcll(nzw);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- per-frame preparation
// What a new frame set needs before the carve kernels can run on it, in two launches and without a
// host round trip (the byte masks of update_visible_voxels_and_extract_colors' fg_masks argument,
// voxel_reconstruction.py:89, are already on the device).  All of it lands in the frame set's header.
//
//  k_prep_pack  byte masks -> bit masks (foreground where byte > 0, voxel_reconstruction.py:112) for all cameras,
//               BGR images -> one dword per pixel in the layout of a record's upper half (a colour sample is then a single aligned load, no byte swap), and each
//               camera's foreground pixel bounding box: workgroup reduction, then atomics only where they still move
//               the box.  The boxes are double-buffered by the slot's frame parity: this launch fills one set and
//               empties the other for the next frame, so no memset (and no fence) sits anywhere.
//  k_prep_grid  every workgroup derives the PLAN of the cropped block grids from the boxes for itself (finest
//               power-of-two block whose grids of all cameras fit the LDS budget; a few dozen scalar operations), then
//               classifies its 256 blocks: two bits per block, "some pixel is foreground" and "every pixel is
//               foreground", written as whole words from ballots (no zero-fill, no atomics).  Extra workgroups count,
//               per camera, how many voxels of a sample of the slab it passes: the carve kernels visit the most
//               selective camera first (stage_order).  That changes the work done, never the result (the all-views
//               test is a conjunction).
struct PrepParams {
    const uint8_t *src[kMaxCameras];   // byte mask of each camera (as uploaded, or post-filtered)
    const uint8_t *fsrc[kMaxCameras];  // BGR images to expand (nframes of them) ...
    uint32_t *fdst[kMaxCameras];       // ... into R | G << 8 | B << 16 | seen << 24
    uint32_t *bits;                    // [C][mwords]
    uint32_t *grid;                    // the frame set's header + grids
    uint32_t *boxes;                   // [2][kMaxCameras][kBoxStride] foreground boxes
    uint32_t C, H, W, HW, mwords, nframes;
    uint32_t parity;                   // which of the two box sets this frame fills
    uint32_t iters;                    // 256-word chunks per packing workgroup (large frame sets: fewer workgroups meet on the boxes)
    uint32_t dbg;                      // experiments only (see CarveParams::dbg)
};

__global__ __launch_bounds__(kBlock) void k_prep_pack(const PrepParams p)
{
    if (p.dbg & 40u) return;
    __shared__ uint32_t s_red[kBlock / 64][4];
    // workgroups [0, C * pw): camera y packs iters x 256 mask words each; then fw per image: 1024 pixels each
    const uint32_t pw = (p.mwords + kBlock * p.iters - 1) / (kBlock * p.iters), fw = (p.HW + 4 * kBlock - 1) / (4 * kBlock);
    const bool packing = blockIdx.x < p.C * pw;
    const uint32_t y = packing ? blockIdx.x / pw : p.C + (blockIdx.x - p.C * pw) / fw;
    const uint32_t t0 = (packing ? (blockIdx.x - y * pw) * p.iters : (blockIdx.x - p.C * pw) - (y - p.C) * fw) * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x < kMaxCameras) {
        // the sample counts of this frame start at zero (k_prep_grid adds to them); the other parity's boxes are emptied
        p.boxes[kCountBase + threadIdx.x * kBoxStride] = 0;
        uint32_t *ob = p.boxes + ((p.parity ^ 1u) * kMaxCameras + threadIdx.x) * kBoxStride;
        ob[0] = 0xffffffffu; ob[1] = 0; ob[2] = 0xffffffffu; ob[3] = 0;
    }
    if (packing) {
        uint32_t u0 = 0xffffffffu, u1 = 0, v0 = 0xffffffffu, v1 = 0;
        for (uint32_t it = 0; it < p.iters; ++it) {
            const uint32_t t = t0 + it * kBlock;
            if (t >= p.mwords) break;
            uint32_t wu0 = 0xffffffffu, wu1 = 0, wv0 = 0xffffffffu, wv1 = 0;
            const uint8_t *src = p.src[y];
            const uint32_t p0 = t * 32u;
            uint32_t out = 0;
            if (p0 + 32u <= p.HW && ((reinterpret_cast<uintptr_t>(src + p0) & 15u) == 0)) {
                const uint4 *s16 = reinterpret_cast<const uint4 *>(src + p0);
                const uint4 a = s16[0], b = s16[1];
                const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    out |= ((v[q] & 0x000000ffu) ? 1u : 0u) << (4 * q + 0);
                    out |= ((v[q] & 0x0000ff00u) ? 1u : 0u) << (4 * q + 1);
                    out |= ((v[q] & 0x00ff0000u) ? 1u : 0u) << (4 * q + 2);
                    out |= ((v[q] & 0xff000000u) ? 1u : 0u) << (4 * q + 3);
                }
            } else {
                for (uint32_t b = 0; b < 32u && p0 + b < p.HW; ++b) out |= (src[p0 + b] ? 1u : 0u) << b;
            }
            p.bits[(size_t)y * p.mwords + t] = out;
            if (out) {
                const uint32_t lo = p0 + (uint32_t)__builtin_ctz(out), hi = p0 + 31u - (uint32_t)__builtin_clz(out);
                wv0 = lo / p.W; wv1 = hi / p.W;
                if (wv0 == wv1) { wu0 = lo - wv0 * p.W; wu1 = hi - wv1 * p.W; }
                else {                                            // the word spans image rows: pixel by pixel
                    for (uint32_t bits = out; bits; bits &= bits - 1) {
                        const uint32_t o = p0 + (uint32_t)__builtin_ctz(bits);
                        const uint32_t u = o - (o / p.W) * p.W;
                        wu0 = u < wu0 ? u : wu0; wu1 = u > wu1 ? u : wu1;
                    }
                }
                u0 = wu0 < u0 ? wu0 : u0; u1 = wu1 > u1 ? wu1 : u1; v0 = wv0 < v0 ? wv0 : v0; v1 = wv1 > v1 ? wv1 : v1;
            }
        }
        u0 = wave_min_u32(u0); u1 = wave_max_u32(u1); v0 = wave_min_u32(v0); v1 = wave_max_u32(v1);
        if (lane == 0) { s_red[wave][0] = u0; s_red[wave][1] = u1; s_red[wave][2] = v0; s_red[wave][3] = v1; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (uint32_t k = 1; k < kBlock / 64; ++k) {
                u0 = s_red[k][0] < u0 ? s_red[k][0] : u0; u1 = s_red[k][1] > u1 ? s_red[k][1] : u1;
                v0 = s_red[k][2] < v0 ? s_red[k][2] : v0; v1 = s_red[k][3] > v1 ? s_red[k][3] : v1;
            }
            if (u0 != 0xffffffffu) {                              // boxes only ever grow: skip what would not move them
                uint32_t *bb = p.boxes + (p.parity * kMaxCameras + y) * kBoxStride;
                if (u0 < __hip_atomic_load(bb + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(bb + 0, u0);
                if (u1 > __hip_atomic_load(bb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(bb + 1, u1);
                if (v0 < __hip_atomic_load(bb + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(bb + 2, v0);
                if (v1 > __hip_atomic_load(bb + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(bb + 3, v1);
            }
        }
    } else {
        // four pixels per thread: 12 bytes in as three dwords, 16 bytes out
        const uint8_t *src = p.fsrc[y - p.C];
        uint32_t *dst = p.fdst[y - p.C];
        const uint32_t t = t0;
        const uint32_t i4 = t * 4u;
        if (i4 + 4u <= p.HW) {
            const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src) + 3u * t;
            const uint32_t w0 = s4[0], w1 = s4[1], w2 = s4[2];
            // B G R bytes -> the upper half of a record as it is: R | G << 8 | B << 16 | "seen" << 24
            const uint32_t x0 = w0 & 0xffffffu, x1 = (w0 >> 24) | ((w1 & 0xffffu) << 8), x2 = (w1 >> 16) | ((w2 & 0xffu) << 16), x3 = w2 >> 8;
            uint4 o;
            o.x = kSeenFlag | ((x0 & 0xffu) << 16) | (x0 & 0xff00u) | (x0 >> 16);
            o.y = kSeenFlag | ((x1 & 0xffu) << 16) | (x1 & 0xff00u) | (x1 >> 16);
            o.z = kSeenFlag | ((x2 & 0xffu) << 16) | (x2 & 0xff00u) | (x2 >> 16);
            o.w = kSeenFlag | ((x3 & 0xffu) << 16) | (x3 & 0xff00u) | (x3 >> 16);
            reinterpret_cast<uint4 *>(dst)[t] = o;
        } else {
            for (uint32_t i = i4; i < p.HW; ++i)
                dst[i] = kSeenFlag | ((uint32_t)src[3 * i] << 16) | ((uint32_t)src[3 * i + 1] << 8) | (uint32_t)src[3 * i + 2];
        }
    }
}

// "any" and "all" of the pixels [o, o + len) of a bit mask, len >= 1
__device__ __forceinline__ void span_any_all(const uint32_t *__restrict__ mb, uint32_t o, uint32_t len, bool &any, bool &all)
{
    uint32_t w = o >> 5;
    const uint32_t wl = (o + len - 1u) >> 5;
    for (; w <= wl; ++w) {
        uint32_t m = 0xffffffffu;
        if (w == (o >> 5)) m &= 0xffffffffu << (o & 31u);
        if (w == wl) m &= 0xffffffffu >> (31u - ((o + len - 1u) & 31u));
        const uint32_t v = mb[w];
        any = any || (v & m) != 0;
        all = all && (v & m) == m;
    }
}

constexpr uint32_t kEstPerThread = 4;

__global__ __launch_bounds__(kBlock) void k_prep_grid(const CarveParams p, uint32_t *grid, uint32_t *boxes_rw, uint32_t parity,
                                                      uint32_t min_shift, uint32_t budget_words, uint32_t nsamples, uint32_t grid_wgs)
{
    if (p.dbg & 40u) return;
    const uint32_t *boxes = boxes_rw;
    uint32_t *counts = boxes_rw + kCountBase;
    __shared__ uint32_t s_hits[kMaxCameras];
    if (threadIdx.x < kMaxCameras) s_hits[threadIdx.x] = 0;
    __syncthreads();
    // workgroups [0, grid_wgs): block grids -- the cameras' blocks one after the other (each camera rounded up to whole
    // workgroups; all grids together hold at most 16 blocks per budgeted word); the rest: the sample counts
    const uint32_t lane = threadIdx.x & 63u;
    if (blockIdx.x < grid_wgs) {
        // ---- the plan, by every wave for itself.  lane = candidate shift: words the grids of all cameras would take
        const uint32_t *box = boxes + parity * kMaxCameras * kBoxStride;
        // lane c fetches camera c's box (one round trip for all cameras); everybody reads them across lanes
        uint32_t mb0 = 0xffffffffu, mb1 = 0, mb2 = 0xffffffffu, mb3 = 0;
        if (lane < p.C) { mb0 = box[kBoxStride * lane]; mb1 = box[kBoxStride * lane + 1]; mb2 = box[kBoxStride * lane + 2]; mb3 = box[kBoxStride * lane + 3]; }
        const uint32_t shift_c = lane < 15u ? lane : 14u;
        uint32_t total = kGridHeader;
        for (uint32_t c = 0; c < p.C; ++c) {
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)mb0, (int)c), b1 = (uint32_t)__builtin_amdgcn_readlane((int)mb1, (int)c);
            const uint32_t b2 = (uint32_t)__builtin_amdgcn_readlane((int)mb2, (int)c), b3 = (uint32_t)__builtin_amdgcn_readlane((int)mb3, (int)c);
            if (b0 > b1) continue;                                // no foreground in this camera
            total += 2u * (((b1 >> shift_c) >> 5) - ((b0 >> shift_c) >> 5) + 1u) * ((b3 >> shift_c) - (b2 >> shift_c) + 1u);
        }
        // W <= 65535, H <= 32767: at 2^14-pixel blocks every grid is a few words, so a shift is always found
        const uint64_t fits = __ballot(lane >= min_shift && lane < 15u && (total <= budget_words || lane == 14u));
        const uint32_t shift = (uint32_t)__builtin_ctzll(fits);
        // lane = camera: its descriptor at the chosen shift, offsets by a scan over the cameras
        uint32_t w_lo = 0, v_lo = 0, cws = 0, ch = 0;
        if (lane < p.C && mb0 <= mb1) {
            w_lo = (mb0 >> shift) >> 5; cws = ((mb1 >> shift) >> 5) - w_lo + 1u;
            v_lo = mb2 >> shift; ch = (mb3 >> shift) - v_lo + 1u;
        }
        const uint32_t size = 2u * cws * ch;
        const uint32_t incl = wave_inclusive_scan(size, lane);
        if (blockIdx.x == 0 && threadIdx.x < 64u) {              // one wave of the launch records the plan for the carve kernels
            if (lane < kMaxCameras) {
                grid[3 * lane] = kGridHeader + incl - size;
                grid[3 * lane + 1] = w_lo | (v_lo << 16);
                grid[3 * lane + 2] = cws | (ch << 16);
            }
            if (lane == 63) { grid[kHdrShift] = shift; grid[kHdrWords] = kGridHeader + incl; }
        }
        // ---- which camera's blocks this workgroup classifies: prefix of the cameras' workgroup counts
        const uint32_t cam_wgs = (32u * cws * ch + kBlock - 1) / kBlock;
        const uint32_t wincl = wave_inclusive_scan(cam_wgs, lane);
        const uint32_t y = (uint32_t)__popcll(__ballot(wincl <= blockIdx.x));          // cameras that end before this workgroup
        if (y >= p.C) return;
        const uint32_t t = (blockIdx.x - (uint32_t)__builtin_amdgcn_readlane((int)(wincl - cam_wgs), (int)y)) * kBlock + threadIdx.x;
        // ---- this workgroup's 256 blocks of camera y
        const uint32_t g_off = kGridHeader + (uint32_t)__builtin_amdgcn_readlane((int)(incl - size), (int)y);
        const uint32_t g_wlo = (uint32_t)__builtin_amdgcn_readlane((int)w_lo, (int)y), g_vlo = (uint32_t)__builtin_amdgcn_readlane((int)v_lo, (int)y);
        const uint32_t g_cws = (uint32_t)__builtin_amdgcn_readlane((int)cws, (int)y), g_ch = (uint32_t)__builtin_amdgcn_readlane((int)ch, (int)y);
        const uint32_t bw = g_cws * 32u, nb = bw * g_ch;          // block columns kept, blocks kept
        bool any = false, all = false;
        if (t < nb) {
            const uint32_t rv = t / bw, ru = t - rv * bw;
            const uint32_t bv = g_vlo + rv, bu = g_wlo * 32u + ru;
            const uint32_t x0 = bu << shift, y0 = bv << shift;
            if (x0 < p.W && y0 < p.H) {
                const uint32_t x1 = ((bu + 1u) << shift) < p.W ? ((bu + 1u) << shift) : p.W;
                const uint32_t y1 = ((bv + 1u) << shift) < p.H ? ((bv + 1u) << shift) : p.H;
                const uint32_t *mb = p.maskbits + (size_t)y * p.mwords;
                all = true;
                const uint32_t len = x1 - x0;
                if (len <= 32u) {
                    // a row of the block is a window of at most 32 bits: two words per row, the rows' loads independent of each other
                    const uint32_t fullm = len == 32u ? 0xffffffffu : ((1u << len) - 1u);
                    uint32_t acc_any = 0, acc_all = fullm;
#pragma unroll 8
                    for (uint32_t yy = y0; yy < y1; ++yy) {
                        const uint32_t o = yy * p.W + x0, wi = o >> 5;
                        const uint64_t two = ((uint64_t)mb[wi + 1 < p.mwords ? wi + 1 : wi] << 32) | mb[wi];
                        const uint32_t win = (uint32_t)(two >> (o & 31u)) & fullm;
                        acc_any |= win; acc_all &= win;
                    }
                    any = acc_any != 0; all = acc_all == fullm;
                }
                else for (uint32_t yy = y0; yy < y1; ++yy) span_any_all(mb, yy * p.W + x0, x1 - x0, any, all);
            }
        }
        // bw is a multiple of 32 and a wave starts at a multiple of 64: each half-wave is one grid word
        const uint64_t bany = __ballot(any), ball = __ballot(all);
        if ((lane & 31u) == 0 && t < nb) {
            const uint32_t rv = t / bw, ru = t - rv * bw;
            const uint32_t w = g_off + rv * g_cws + (ru >> 5);
            grid[w] = (uint32_t)(bany >> lane);
            grid[w + g_ch * g_cws] = (uint32_t)(ball >> lane);
        }
        return;
    }
    // ---- pass count of each camera on `nsamples` evenly spaced voxels of the slab, kEstPerThread per thread
    const uint32_t eblock = blockIdx.x - grid_wgs;
    if (eblock * kBlock * kEstPerThread >= nsamples) return;
    double X[kEstPerThread], Y[kEstPerThread], Z[kEstPerThread];
    uint32_t valid = 0;
#pragma unroll
    for (uint32_t k = 0; k < kEstPerThread; ++k) {
        const uint32_t sidx = (blockIdx.x * kEstPerThread + k) * kBlock + threadIdx.x;
        uint32_t ix = 0, iy = 0, izl = 0;
        if (sidx < nsamples) {
            valid |= 1u << k;
            decompose((uint32_t)(((uint64_t)sidx * p.n) / nsamples), p.nx, p.ny, ix, iy, izl);
        }
        X[k] = p.xs[ix]; Y[k] = p.ys[iy]; Z[k] = p.zs[p.z0 + izl];
    }
    for (uint32_t c = 0; c < p.C; ++c) {
        uint32_t hits = 0;
#pragma unroll
        for (uint32_t k = 0; k < kEstPerThread; ++k) {
            double u, v;
            project_point(p.cam[c], X[k], Y[k], Z[k], u, v);
            const int32_t off = pixel_offset(u, v, p.H, p.W);
            const bool hit = ((valid >> k) & 1u) && off >= 0 && mask_bit(p.maskbits + (size_t)c * p.mwords, off);
            hits += (uint32_t)__popcll(__ballot(hit));
        }
        if (lane == 0 && hits) atomicAdd(&s_hits[c], hits);
    }
    __syncthreads();
    if (threadIdx.x < p.C && s_hits[threadIdx.x]) atomicAdd(&counts[threadIdx.x * kBoxStride], s_hits[threadIdx.x]);   // one line per camera
}

// ---------------------------------------------------------------- compaction
// Ordered compaction without a sort: survivors per 64-word group -> exclusive scan (two
// levels) -> one wave per group expands its words into records.

// Large frame sets (blocks of 8 x 8 pixels at 16 cameras x 1080p): a brick's pixel box spans 8 x 8 such blocks and more, the loops
// of box_test over them cost k_cull_bricks 45 us, and the brick level needs no such resolution.  Once per frame set, behind
// k_prep_grid: grids of 4 x 4 times coarser blocks in the same format -- "any" = OR, "all" = AND of the 16 blocks (blocks outside
// a camera's kept rectangle hold no foreground: any = all = 0), aligned to absolute block coordinates, so that box_test reads them
// unchanged with the shift raised by 2.  Coarser is conservative in both directions ("dead" and "full" stay true statements
// about the fine blocks).  blockIdx.y = camera; every workgroup works out all descriptors for itself (lane = camera).
__device__ __forceinline__ uint32_t gather_every_4th_bit(uint32_t x)        // bits 0, 4, .., 28 -> bits 0..7
{
    x &= 0x11111111u;
    x = (x | (x >> 3)) & 0x03030303u;
    x = (x | (x >> 6)) & 0x000f000fu;
    return (x | (x >> 12)) & 0xffu;
}
__global__ __launch_bounds__(kBlock) void k_coarsen_grids(const uint32_t *__restrict__ fine, uint32_t *__restrict__ coarse, uint32_t C)
{
    __shared__ uint32_t s_d[3];                                   // this workgroup's camera: coarse offset, origin, size
    const uint32_t lane = threadIdx.x & 63u, c = blockIdx.y;
    if (threadIdx.x < 64u) {
        uint32_t f1 = 0, f2 = 0;
        if (lane < C) { f1 = fine[3 * lane + 1]; f2 = fine[3 * lane + 2]; }
        const uint32_t w_lo = f1 & 0xffffu, v_lo = f1 >> 16, cws = f2 & 0xffffu, ch = f2 >> 16;
        uint32_t cw_lo = 0, cv_lo = 0, ccws = 0, cch = 0;
        if (lane < C && ch != 0) {
            cw_lo = w_lo >> 2; cv_lo = v_lo >> 2;
            ccws = ((w_lo + cws - 1u) >> 2) - cw_lo + 1u;
            cch = ((v_lo + ch - 1u) >> 2) - cv_lo + 1u;
        }
        const uint32_t size = 2u * ccws * cch;
        const uint32_t incl = wave_inclusive_scan(size, lane);
        if (lane == c) { s_d[0] = kGridHeader + incl - size; s_d[1] = cw_lo | (cv_lo << 16); s_d[2] = ccws | (cch << 16); }
        if (blockIdx.x == 0 && c == 0) {                          // one workgroup writes the header
            if (lane < kMaxCameras) {
                coarse[3 * lane] = kGridHeader + incl - size;
                coarse[3 * lane + 1] = cw_lo | (cv_lo << 16);
                coarse[3 * lane + 2] = ccws | (cch << 16);
            }
            if (lane == 63u) { coarse[kHdrShift] = fine[kHdrShift] + 2u; coarse[kHdrWords] = kGridHeader + incl; }
        }
    }
    __syncthreads();
    const uint32_t f0 = fine[3 * c], f1 = fine[3 * c + 1], f2 = fine[3 * c + 2];
    const uint32_t w_lo = f1 & 0xffffu, v_lo = f1 >> 16, cws = f2 & 0xffffu, ch = f2 >> 16;
    const uint32_t cw_lo = s_d[1] & 0xffffu, cv_lo = s_d[1] >> 16, ccws = s_d[2] & 0xffffu, cch = s_d[2] >> 16;
    const uint32_t *__restrict__ g_any = fine + f0;
    const uint32_t *__restrict__ g_all = g_any + ch * cws;
    uint32_t *o_any = coarse + s_d[0];
    uint32_t *o_all = o_any + cch * ccws;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < ccws * cch; i += gridDim.x * kBlock) {
        const uint32_t R = i / ccws, Cw = i - R * ccws;
        const uint32_t fr0 = 4u * (cv_lo + R), fw0 = 4u * (cw_lo + Cw);        // first fine row / word under this coarse word
        uint32_t cany = 0, call = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t fw = fw0 + j;
            const bool wok = fw >= w_lo && fw < w_lo + cws;
            uint32_t a[4], l[4];
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) {                    // (clamped addresses: the eight loads go out together)
                const uint32_t fr = fr0 + r;
                const bool ok = wok && fr >= v_lo && fr < v_lo + ch;
                const uint32_t at = ok ? (fr - v_lo) * cws + (fw - w_lo) : 0u;
                a[r] = g_any[at]; l[r] = g_all[at];
                if (!ok) { a[r] = 0u; l[r] = 0u; }
            }
            uint32_t anyrow = a[0] | a[1] | a[2] | a[3], allrow = l[0] & l[1] & l[2] & l[3];
            anyrow |= anyrow >> 1; anyrow |= anyrow >> 2;         // bit 4 q = OR of bits 4 q .. 4 q + 3
            allrow &= allrow >> 1; allrow &= allrow >> 2;         // bit 4 q = AND of bits 4 q .. 4 q + 3
            cany |= gather_every_4th_bit(anyrow) << (8u * j);
            call |= gather_every_4th_bit(allrow) << (8u * j);
        }
        o_any[i] = cany;
        o_all[i] = call;
    }
}


// For the kernels that do not write groupcnt themselves (fused, generic): one wave per group.
__global__ __launch_bounds__(kBlock) void k_count_groups(const uint64_t *__restrict__ words, uint64_t nwords,
                                                         uint32_t ngroups, uint32_t *__restrict__ groupcnt)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (g >= ngroups) return;
    const uint64_t w = (uint64_t)g * kGroupWords + lane;
    uint32_t cnt = (w < nwords) ? (uint32_t)__popcll(words[w]) : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    if (lane == 0) groupcnt[g] = cnt;
}

// Zero-fills the words of the groups the hierarchical kernels left unwritten (no survivors).
__global__ __launch_bounds__(kBlock) void k_zero_dead_groups(uint64_t *__restrict__ words, uint64_t nwords, uint32_t ngroups,
                                                             const uint32_t *__restrict__ survcnt)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (g >= ngroups || survcnt[g] != 0) return;
    const uint64_t w = (uint64_t)g * kGroupWords + lane;
    if (w < nwords) words[w] = 0ull;
}

// ---- compact exchange form of a carve result (multi-GPU): the non-zero words of the slab as
// {bits, global index of bit 0} pairs, ascending.  A slab's hull fills ~1 word in 25, so the pairs
// are ~30x smaller than the survivor records they expand to.
__global__ __launch_bounds__(kBlock) void k_count_nz(const uint64_t *__restrict__ words, uint64_t nwords,
                                                     uint32_t ngroups, const uint32_t *__restrict__ survcnt,
                                                     uint32_t *__restrict__ groupcnt)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (g >= ngroups) return;
    if (survcnt[g] == 0) {                                              // its words may be unwritten
        if (lane == 0) groupcnt[g] = 0;
        return;
    }
    const uint64_t w = (uint64_t)g * kGroupWords + lane;
    const uint64_t nz = __ballot(w < nwords && words[w] != 0ull);
    if (lane == 0) groupcnt[g] = (uint32_t)__popcll(nz);
}

// mine[0] = entry count, mine[1] = survivor count of this rank (what the counts all-gather sends).
__global__ __launch_bounds__(kBlock) void k_pack_entries(const uint64_t *__restrict__ words, uint64_t nwords,
                                                         uint32_t ngroups, const uint32_t *__restrict__ survcnt,
                                                         const uint32_t *__restrict__ groupoff,
                                                         const uint64_t *__restrict__ blockoff, uint32_t nscan,
                                                         uint64_t i0, const uint64_t *__restrict__ survivors,
                                                         uint64_t *__restrict__ entries, uint64_t *__restrict__ mine)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) { mine[0] = blockoff[nscan]; mine[1] = *survivors; }
    if (g >= ngroups) return;
    if (survcnt[g] == 0) return;
    const uint64_t w = (uint64_t)g * kGroupWords + lane;
    const uint64_t bits = (w < nwords) ? words[w] : 0ull;
    const uint64_t nz = __ballot(bits != 0ull);
    if (bits != 0ull) {
        const uint64_t o = blockoff[g / kScanBlock] + groupoff[g] + (uint32_t)__popcll(nz & ((1ull << lane) - 1ull));
        entries[2 * o] = bits;
        entries[2 * o + 1] = i0 + (w << 6);
    }
}

// The same over the list of groups with survivors (k_finish_scan), offsets from the scan of the counts k_assemble left.
__global__ __launch_bounds__(kBlock) void k_pack_busy(const uint64_t *__restrict__ words, uint64_t nwords,
                                                      const uint32_t *__restrict__ busylist, const uint32_t *__restrict__ busycount,
                                                      const uint32_t *__restrict__ groupoff, const uint64_t *__restrict__ blockoff,
                                                      uint32_t nscan, uint64_t i0, const uint64_t *__restrict__ survivors,
                                                      uint64_t *__restrict__ entries, uint64_t *__restrict__ mine)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (kBlock / 64);
    if (blockIdx.x == 0 && threadIdx.x == 0) { mine[0] = blockoff[nscan]; mine[1] = *survivors; }
    const uint32_t nbusy = busycount[0];
    for (uint32_t t = w0; t < nbusy; t += nwaves) {
        const uint32_t g = busylist[t];
        const uint64_t w = (uint64_t)g * kGroupWords + lane;
        const uint64_t bits = (w < nwords) ? words[w] : 0ull;
        const uint64_t nz = __ballot(bits != 0ull);
        if (bits != 0ull) {
            const uint64_t o = blockoff[g / kScanBlock] + groupoff[g] + (uint32_t)__popcll(nz & ((1ull << lane) - 1ull));
            entries[2 * o] = bits;
            entries[2 * o + 1] = i0 + (w << 6);
        }
    }
}

// The gathered entries are expanded `chunk` entries per wave (EmitParams::entry_chunk), not 64: a wave that owns 64 dense words
// writes ~2 800 records and keeps its slot for the whole launch -- one generation of long-lived waves that a higher-priority
// stream's kernels cannot get past (the per-voxel level beside it went 41 -> 80 us); with 16 the slots turn over four times as often.
__global__ __launch_bounds__(kBlock) void k_count_entries(const uint64_t *__restrict__ entries, uint64_t nent,
                                                          uint32_t ngroups, uint32_t *__restrict__ groupcnt, uint32_t chunk)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (chunk == 16u) {
        // four chunks per wave, one per row of 16 lanes (42 000 waves with a quarter of their lanes alive were 16-21 us of this stream's step)
        const uint32_t g = 4u * wv + (lane >> 4);
        const uint64_t w = (uint64_t)wv * 64u + lane;
        uint32_t cnt = (g < ngroups && w < nent) ? (uint32_t)__popcll(entries[2 * w]) : 0u;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        if ((lane & 15u) == 0u && g < ngroups) groupcnt[g] = cnt;
        return;
    }
    const uint32_t g = wv;
    if (g >= ngroups) return;
    const uint64_t w = (uint64_t)g * chunk + lane;
    uint32_t cnt = (lane < chunk && w < nent) ? (uint32_t)__popcll(entries[2 * w]) : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    if (lane == 0) groupcnt[g] = cnt;
}

// Level 1: exclusive scan of the group counts inside blocks of kScanBlock groups.
// A single-block launch (<= 1024 groups: grids up to 4 M voxels) also finishes level 2 itself.
// The grand total goes to device memory AND straight to a page-locked host word (no copy kernel).
// Workgroups of 256 threads, four consecutive groups each: a 1024-thread workgroup needs sixteen free wave slots on ONE compute
// unit, and while another stream's kernel keeps refilling every slot that frees up (the record expansion's 16 384 four-wave
// workgroups) it finds them only once that kernel has nothing left to dispatch -- scripts/exp_streams.py: the two scan launches,
// 11 us of work, cost the pipelined step 68 us.
constexpr uint32_t kScanThreads = 256, kScanPer = kScanBlock / kScanThreads;
__global__ __launch_bounds__(kScanThreads) void k_scan_groups(const uint32_t *__restrict__ cnt, uint32_t ngroups,
                                                              uint32_t *__restrict__ off, uint64_t *__restrict__ blocksum,
                                                              uint64_t *__restrict__ blockoff, uint64_t *__restrict__ total_host,
                                                              uint32_t *__restrict__ busyoff, uint32_t *__restrict__ busysum,
                                                              uint32_t *__restrict__ busyblock, uint32_t dbg = 0)
{
    if (dbg & (8u | 512u)) return;
    __shared__ uint32_t wsum[kScanThreads / 64], wbusy[kScanThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t i0 = blockIdx.x * kScanBlock + kScanPer * t;
    uint32_t c[kScanPer];                                // <= 4096 each: a block total fits u32
    if (i0 + kScanPer <= ngroups) {                      // (16-byte aligned: i0 is a multiple of four)
        const uint4 v = *reinterpret_cast<const uint4 *>(cnt + i0);
        c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < kScanPer; ++k) c[k] = (i0 + k < ngroups) ? cnt[i0 + k] : 0u;
    }
    static_assert(kScanPer == 4, "four groups per thread");
    const uint32_t own = c[0] + c[1] + c[2] + c[3];
    const uint32_t incl = wave_inclusive_scan(own, lane);
    // busyoff != null: the same scan over "group has survivors", for the list of busy groups (k_finish_scan)
    uint32_t f[kScanPer], fown = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanPer; ++k) { f[k] = (busyoff && c[k]) ? 1u : 0u; fown += f[k]; }
    const uint32_t fincl = busyoff ? wave_inclusive_scan(fown, lane) : 0u;
    if (lane == 63) { wsum[wave] = incl; wbusy[wave] = fincl; }
    __syncthreads();
    uint32_t before = 0, total = 0, fbefore = 0, ftotal = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanThreads / 64; ++k) {
        const uint32_t s = wsum[k], fs = wbusy[k];
        if (k < wave) { before += s; fbefore += fs; }
        total += s; ftotal += fs;
    }
    uint32_t run = before + incl - own, frun = fbefore + fincl - fown;
    uint32_t o[kScanPer], fo[kScanPer];
#pragma unroll
    for (uint32_t k = 0; k < kScanPer; ++k) { o[k] = run; run += c[k]; fo[k] = frun; frun += f[k]; }
    if (i0 + kScanPer <= ngroups) {
        *reinterpret_cast<uint4 *>(off + i0) = make_uint4(o[0], o[1], o[2], o[3]);
        if (busyoff) *reinterpret_cast<uint4 *>(busyoff + i0) = make_uint4(fo[0], fo[1], fo[2], fo[3]);
    } else {
#pragma unroll
        for (uint32_t k = 0; k < kScanPer; ++k) {
            if (i0 + k < ngroups) {
                off[i0 + k] = o[k];
                if (busyoff) busyoff[i0 + k] = fo[k];
            }
        }
    }
    if (t == 0) {
        blocksum[blockIdx.x] = total;
        if (busyoff) busysum[blockIdx.x] = ftotal;
        if (gridDim.x == 1) {
            blockoff[0] = 0;
            blockoff[1] = total;
            *total_host = total;
            if (busyoff) busyblock[0] = ftotal;                  // single block: this is the count of busy groups
        }
    }
}

// Level 2: exclusive scan of the (at most kScanBlock) block sums; blockoff[nblocks] = total.  Thread t owns blocks 4t .. 4t+3.
__global__ __launch_bounds__(kScanThreads) void k_scan_blocks(const uint64_t *__restrict__ blocksum, uint32_t nblocks,
                                                              uint64_t *__restrict__ blockoff, uint64_t *__restrict__ total_host)
{
    __shared__ uint64_t wsum[kScanThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint64_t v[kScanPer], own = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanPer; ++k) { v[k] = (kScanPer * t + k < nblocks) ? blocksum[kScanPer * t + k] : 0ull; own += v[k]; }
    uint64_t incl = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t base = incl - own, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanThreads / 64; ++k) {
        if (k < wave) base += wsum[k];
        total += wsum[k];
    }
#pragma unroll
    for (uint32_t k = 0; k < kScanPer; ++k) {
        if (kScanPer * t + k < nblocks) blockoff[kScanPer * t + k] = base;
        base += v[k];
    }
    if (t == 0) {
        blockoff[nblocks] = total;
        *total_host = total;
    }
}

// Level 2 of both scans and the list of busy groups in one launch (grids large enough for the list: see
// kBusyListMinGroups).  A workgroup covers 256 consecutive groups, all inside one scan block b, so all it needs
// of the busy scan is the sum of the blocks before b; workgroup 0 also does the whole level-2 scan of the
// survivor sums (at most 1024 values, four per thread) that the expansion reads through blockoff.
__global__ __launch_bounds__(kBlock) void k_finish_scan(const uint64_t *__restrict__ blocksum, uint32_t nblocks,
                                                        uint64_t *__restrict__ blockoff, uint64_t *__restrict__ total_host,
                                                        const uint32_t *__restrict__ busysum, uint32_t *__restrict__ busycount,
                                                        const uint32_t *__restrict__ cnt, uint32_t ngroups,
                                                        const uint32_t *__restrict__ busyoff, uint32_t *__restrict__ list, uint32_t dbg = 0)
{
    if (dbg & (8u | 512u)) return;
    __shared__ uint32_t wred[kBlock / 64];
    __shared__ uint64_t wsum[kBlock / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t i = blockIdx.x * kBlock + t;
    const uint32_t b = (blockIdx.x * kBlock) / kScanBlock;            // scan block of this workgroup's groups
    // busy groups in the scan blocks before b (for workgroup 0 too: it reports the total)
    uint32_t part = 0, all = 0;
    for (uint32_t k = t; k < nblocks; k += kBlock) {
        const uint32_t v = busysum[k];
        if (k < b) part += v;
        all += v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { part += __shfl_xor(part, d); all += __shfl_xor(all, d); }
    if (lane == 0) wred[wave] = part;
    __syncthreads();
    const uint32_t before = wred[0] + wred[1] + wred[2] + wred[3];
    if (i < ngroups && cnt[i]) list[before + busyoff[i]] = i;
    if (blockIdx.x != 0) return;
    __syncthreads();
    if (lane == 0) wred[wave] = all;
    // level 2 of the survivor scan: thread t owns blocks 4t .. 4t+3
    uint64_t v[4], own = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (4 * t + k < nblocks) ? blocksum[4 * t + k] : 0ull; own += v[k]; }
    uint64_t incl = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t base = incl - own, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kBlock / 64; ++k) {
        if (k < wave) base += wsum[k];
        total += wsum[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (4 * t + k < nblocks) blockoff[4 * t + k] = base;
        base += v[k];
    }
    if (t == 0) {
        blockoff[nblocks] = total;
        *total_host = total;
        *busycount = wred[0] + wred[1] + wred[2] + wred[3];
    }
}

// Slab-local voxel j -> its entry in a tile-ordered table (word T = 4 x-rows x 16 y of one z-layer, element = row * 16 + y).
__device__ __forceinline__ uint32_t tile_index(uint32_t j, uint32_t nx, uint32_t ny, uint32_t tq)
{
    uint32_t ix, iy, izl;
    decompose(j, nx, ny, ix, iy, izl);
    return (((izl * (nx >> 2) + (ix >> 2)) * tq + (iy >> 4)) << 6) + ((ix & 3u) << 4) + (iy & 15u);
}

// The y-major table back out of the tile-ordered one (a table that came in through vc_upload_lut is never re-projected).
__global__ __launch_bounds__(kBlock) void k_untile_lut(const CarveParams p, const int32_t *__restrict__ lut_tile, int32_t *__restrict__ lut)
{
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;      // grid covers n_pad exactly
    for (uint32_t c = 0; c < p.C; ++c)
        lut[(size_t)c * p.n_pad + j] = j < p.n ? lut_tile[(size_t)c * p.n_pad + tile_index((uint32_t)j, p.nx, p.ny, p.tq)] : -1;
}

// r-th (0-based) set bit of x; requires r < popcount(x).
__device__ __forceinline__ uint32_t select_bit(uint64_t x, uint32_t r)
{
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t s = 32; s >= 1; s >>= 1) {
        const uint32_t c = (uint32_t)__popcll((x >> pos) & ((1ull << s) - 1ull));
        if (r >= c) { r -= c; pos += s; }
    }
    return pos;
}

// One wave per group of 64 words (one word per lane), one lane per SURVIVOR: survivor k of
// the group finds its word by a 6-step search over the lanes' inclusive popcounts (cross-lane
// reads, no LDS arrays, no barriers) and its voxel by a 6-step bit select, so every load and
// store runs with all 64 lanes busy.  EU survivors per lane are in flight together so that
// the dependent loads (table entry -> pixel) overlap.  Record = {u32 idx, r, g, b, seen}
// (assignment.py:133).  ALLSEEN: every survivor is seen by every camera (min_views == C), so
// the colour camera's test is known to pass; FROM_LUT: its pixel offset is read from the table
// instead of being re-projected.
template <bool FROM_LUT, bool ALLSEEN, int EU>
__device__ __forceinline__ void emit_body(const EmitParams &p, uint32_t vblock)
{
    const uint32_t lane = threadIdx.x & 63u;
    // One wave per group, no loop: the grid is the group list and the hardware dispatcher
    // balances the busy groups of the compact hull; an empty group (5 of 6) costs its wave one
    // scalar load of the group count.
    const uint32_t g = __builtin_amdgcn_readfirstlane((vblock * kBlock + threadIdx.x) >> 6);
    if (g >= p.ngroups) return;
    if (p.groupcnt[g] == 0) return;
    const uint64_t nwords = (p.n + 63) >> 6;
    const uint64_t out0 = p.blockoff[g / kScanBlock] + p.groupoff[g];
    {
        const uint64_t gw = (uint64_t)g * kGroupWords;
        const uint64_t mine = (gw + lane < nwords) ? p.words[gw + lane] : 0ull;
        const uint32_t c = (uint32_t)__popcll(mine);
        const uint32_t incl = wave_inclusive_scan(c, lane);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
        for (uint32_t k0 = 0; k0 < total; k0 += 64 * EU) {           // wave-uniform
            uint32_t j[EU];
            bool live[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const uint32_t k = k0 + 64 * u + lane;
                live[u] = k < total;
                const uint32_t kk = live[u] ? k : total - 1;
                uint32_t lo = 0;                                      // first lane whose inclusive count exceeds kk
#pragma unroll
                for (uint32_t step = 32; step >= 1; step >>= 1)
                    if ((uint32_t)__shfl((int)incl, (int)(lo + step - 1)) <= kk) lo += step;
                const uint64_t wb = ((uint64_t)(uint32_t)__shfl((int)mhi, (int)lo) << 32) | (uint32_t)__shfl((int)mlo, (int)lo);
                const uint32_t before = (uint32_t)__shfl((int)(incl - c), (int)lo);
                j[u] = (uint32_t)((gw + lo) << 6) + select_bit(wb, kk - before);
            }
            int32_t off[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                off[u] = -1;
                if (p.has_cam) {
                    if (FROM_LUT) {
                        off[u] = p.lut[p.lut_tq ? tile_index(j[u], p.nx, p.ny, p.lut_tq) : j[u]];
                    } else {
                        uint32_t ix, iy, izl;
                        decompose(j[u], p.nx, p.ny, ix, iy, izl);
                        double uu, vv;
                        project_point(p.cam, p.xs[ix], p.ys[iy], p.zs[p.z0 + izl], uu, vv);
                        off[u] = pixel_offset(uu, vv, p.H, p.W);
                    }
                }
            }
            uint64_t rec[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                rec[u] = (uint32_t)(p.i0 + j[u]);
                if (off[u] >= 0 && (ALLSEEN || (p.maskbits && mask_bit(p.maskbits, off[u])))) {
                    const uint64_t px = p.frame ? (uint64_t)p.frame[off[u]] : (uint64_t)kSeenFlag;   // R | G<<8 | B<<16 | seen<<24
                    rec[u] |= px << 32;
                }
            }
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const uint64_t o = out0 + k0 + 64 * u + lane;
                if (live[u] && o < p.capacity) __builtin_nontemporal_store(rec[u], &p.records[o]);
            }
        }
    }
}

template <bool FROM_LUT, bool ALLSEEN, int EU>
__global__ __launch_bounds__(kBlock) void k_emit_words(const EmitParams p)
{
    emit_body<FROM_LUT, ALLSEEN, EU>(p, blockIdx.x);
}

// Variant of the expansion with lanes = the 64 voxels of a word: a survivor's rank inside its
// word is the count of set bits below its lane (no search at all), at the price of idle lanes in
// sparse words (the hull's words are dense: ~44 of 64 bits).  EB words are expanded together.
// INDIRECT: the "words" are gathered {bits, base} entries of all ranks (p.n = 64 x entries, p.lut =
// the colour camera's table over the WHOLE grid, p.z0 = p.i0 = 0): voxel index = base + lane.
// One batch of EB words of a group in flight: what is wave-uniform about each word (scalar registers) and each lane's table
// entry / pixel offset for its voxel of that word.
template <int EB>
struct EmitBatch {
    uint32_t wl[EB];            // first record of the word inside the group (<= 4096) | word number inside the group << 16
    uint32_t jb[EB];            // INDIRECT only: voxel index of the word's bit 0 (else it follows from the word number)
    uint64_t wv[EB];
    int32_t off[EB];
    bool any;
};

// takes the next (up to) EB non-zero words of the group off `nz` and starts their table loads (or projects)
template <bool FROM_LUT, int EB, bool INDIRECT>
__device__ __forceinline__ void emit_prepare(const EmitParams &p, uint64_t &nz, uint64_t mine, uint32_t mybase, uint32_t &wstart,
                                             uint32_t tbase, uint64_t gw, uint32_t lane, EmitBatch<EB> &B,
                                             bool rowwords = false, uint32_t wix = 0, uint32_t wiy = 0, uint32_t wiz = 0)
{
    B.any = nz != 0;
    uint32_t tb[EB];
#pragma unroll
    for (int b = 0; b < EB; ++b) {
        B.wv[b] = 0; B.wl[b] = 0; B.jb[b] = 0; tb[b] = 0;
        if (nz != 0) {
            const uint32_t li = (uint32_t)__builtin_ctzll(nz);
            nz &= nz - 1;
            if (INDIRECT) B.jb[b] = (uint32_t)__builtin_amdgcn_readlane((int)mybase, (int)li);
            const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, (int)li);
            const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), (int)li);
            B.wv[b] = ((uint64_t)whi << 32) | wlo;
            // (first record of the word inside the group = survivors of the words before it: they are taken in ascending order,
            // so a running scalar sum does it -- no prefix scan over the group, no cross-lane read per word)
            B.wl[b] = wstart | (li << 16);
            wstart += (uint32_t)__popcll(B.wv[b]);
            tb[b] = (uint32_t)__builtin_amdgcn_readlane((int)tbase, (int)li);
        }
    }
#pragma unroll
    for (int b = 0; b < EB; ++b) {
        B.off[b] = -1;
        // (the word's bits, wave-uniform, ARE the lane mask of "my voxel survives": handed to the compiler as such, the branch is
        // one s_and_saveexec instead of shift, and, compare per lane)
        if (p.has_cam && __builtin_amdgcn_inverse_ballot_w64(B.wv[b])) {
            const uint32_t j = (INDIRECT ? B.jb[b] : (uint32_t)((gw + (B.wl[b] >> 16)) << 6)) + lane;
            if (FROM_LUT) {
                // (tile order: the word starts at a multiple of 64 and ny % 64 == 0, so its 64 voxels are 4 runs of 16 entries;
                // everything but the lane terms is wave-uniform)
                B.off[b] = p.lut[(!INDIRECT && p.lut_tq) ? tb[b] + ((lane >> 4) << 6) + (lane & 15u) : j];
            } else {
                uint32_t ix, iy, izl;
                if (rowwords) {                                   // (wave-uniform) the word lies in one x-row: its lanes differ in y only
                    const uint32_t li = B.wl[b] >> 16;
                    ix = (uint32_t)__builtin_amdgcn_readlane((int)wix, (int)li);
                    iy = (uint32_t)__builtin_amdgcn_readlane((int)wiy, (int)li) + lane;
                    izl = (uint32_t)__builtin_amdgcn_readlane((int)wiz, (int)li);
                } else decompose(j, p.nx, p.ny, ix, iy, izl);
                double u, v;
                project_point(p.cam, p.xs[ix], p.ys[iy], p.zs[p.z0 + izl], u, v);
                B.off[b] = pixel_offset(u, v, p.H, p.W);
            }
        }
    }
}

// colour gathers and record stores of a prepared batch
// ROOM: the group's records (at most 4 096) all fit the buffer -- no test per record
template <bool ALLSEEN, int EB, bool INDIRECT, bool ROOM = false>
__device__ __forceinline__ void emit_finish(const EmitParams &p, uint64_t out0, uint64_t gw, uint32_t lane, const EmitBatch<EB> &B)
{
    const uint64_t below = (1ull << lane) - 1ull;
    uint64_t *__restrict__ grp = p.records + out0;                // (wave-uniform: the stores take a 32-bit offset from it)
    // all gathers of the batch are issued before any of them is used (a load consumed inside its own branch is waited for
    // inside it: eight round trips in a row)
    uint32_t mw[EB], px[EB];
#pragma unroll
    for (int b = 0; b < EB; ++b) {
        mw[b] = ~0u;
        if (!ALLSEEN && p.maskbits && B.off[b] >= 0) mw[b] = p.maskbits[(uint32_t)B.off[b] >> 5];
    }
    bool seen[EB];
#pragma unroll
    for (int b = 0; b < EB; ++b) {
        seen[b] = B.off[b] >= 0 && (ALLSEEN || (p.maskbits && ((mw[b] >> ((uint32_t)B.off[b] & 31u)) & 1u)));
        px[b] = kSeenFlag;
        if (seen[b] && p.frame) px[b] = p.frame[B.off[b]];                          // R | G<<8 | B<<16 | seen<<24: a record's upper half
    }
    uint64_t rec[EB];
#pragma unroll
    for (int b = 0; b < EB; ++b) {
        rec[b] = (uint32_t)(p.i0 + (INDIRECT ? B.jb[b] : (uint32_t)((gw + (B.wl[b] >> 16)) << 6)) + lane);
        if (seen[b]) rec[b] |= (uint64_t)px[b] << 32;
    }
#pragma unroll
    for (int b = 0; b < EB; ++b) {
        if (__builtin_amdgcn_inverse_ballot_w64(B.wv[b])) {
            const uint32_t k = (B.wl[b] & 0xffffu) + (uint32_t)__popcll(B.wv[b] & below);
            // (streamed past the caches: 238 MB per step that nothing on the device reads again would evict the masks, images
            // and block grids the next kernels want)
            if (ROOM || out0 + k < p.capacity) __builtin_nontemporal_store(rec[b], &grp[k]);
        }
    }
}

// What a wave needs of a group before it can start: its 64 words (one per lane) and where its records begin.
struct EmitGroup {
    uint64_t mine, out0;
    uint32_t mybase;
};
template <bool INDIRECT>
__device__ __forceinline__ EmitGroup emit_load_group(const EmitParams &p, uint32_t g, uint32_t lane)
{
    EmitGroup h;
    const uint64_t nwords = (p.n + 63) >> 6;
    h.out0 = p.blockoff[g / kScanBlock] + p.groupoff[g];
    const uint64_t gw = (uint64_t)g * (INDIRECT ? p.entry_chunk : kGroupWords);
    h.mine = 0ull;
    h.mybase = 0;
    if (gw + lane < nwords && (!INDIRECT || lane < p.entry_chunk)) {
        if (INDIRECT) {
            const ulonglong2 e = reinterpret_cast<const ulonglong2 *>(p.entries)[gw + lane];
            h.mine = e.x; h.mybase = (uint32_t)e.y;
        } else h.mine = p.words[gw + lane];
    }
    return h;
}

// The dependent chain of a batch is table entry -> pixel -> store; the next batch's table loads are issued before the
// current batch's pixels are waited for, so a group of B batches costs about B + 1 round trips instead of 2 B.
template <bool FROM_LUT, bool ALLSEEN, int EB, bool INDIRECT, bool ROOM = false>
__device__ __forceinline__ void emit_group_lanes(const EmitParams &p, const uint32_t g, const uint32_t lane, const EmitGroup &h)
{
    const uint64_t gw = (uint64_t)g * kGroupWords;
    uint32_t wstart = 0;                                                // survivors of the group's words taken so far (wave-uniform)
    // colour look-up in a TILE-ordered table: where my word's 64 entries start (4 runs of 16; all lanes at once, once per group)
    const uint32_t tbase = (FROM_LUT && !INDIRECT && p.lut_tq) ? tile_index((uint32_t)((gw + lane) << 6), p.nx, p.ny, p.lut_tq) : 0u;
    uint64_t nz = __ballot(h.mine != 0);
    EmitBatch<EB> A;
    if (!FROM_LUT) {
        // projecting every survivor is arithmetic bound: nothing to overlap, and a second batch's state would not fit the registers.
        // ny % 64 == 0: a word is 64 consecutive y of one x-row, so (ix, iy of bit 0, iz) are found once per group for all 64
        // words (lane = word) instead of by two integer divisions per survivor
        const bool rowwords = (p.ny & 63u) == 0;
        uint32_t wix = 0, wiy = 0, wiz = 0;
        if (rowwords) decompose(INDIRECT ? h.mybase : (uint32_t)((gw + lane) << 6), p.nx, p.ny, wix, wiy, wiz);
        while (nz != 0) {                                               // wave-uniform
            emit_prepare<FROM_LUT, EB, INDIRECT>(p, nz, h.mine, h.mybase, wstart, tbase, gw, lane, A, rowwords, wix, wiy, wiz);
            emit_finish<ALLSEEN, EB, INDIRECT>(p, h.out0, gw, lane, A);
        }
        if (p.has_cam) stat_add(p.stats, 3 /* VC_WORK_EMIT_PROJECTIONS */, g, lane, wstart);   // (every survivor of the group was projected)
        return;
    }
    EmitBatch<EB> B;
    emit_prepare<FROM_LUT, EB, INDIRECT>(p, nz, h.mine, h.mybase, wstart, tbase, gw, lane, A);
    while (A.any) {                                                     // wave-uniform; A and B take turns (no copies)
        emit_prepare<FROM_LUT, EB, INDIRECT>(p, nz, h.mine, h.mybase, wstart, tbase, gw, lane, B);
        emit_finish<ALLSEEN, EB, INDIRECT, ROOM>(p, h.out0, gw, lane, A);
        if (!B.any) break;
        emit_prepare<FROM_LUT, EB, INDIRECT>(p, nz, h.mine, h.mybase, wstart, tbase, gw, lane, A);
        emit_finish<ALLSEEN, EB, INDIRECT, ROOM>(p, h.out0, gw, lane, B);
    }
}

template <bool FROM_LUT, bool ALLSEEN, int EB, bool INDIRECT = false>
__global__ __launch_bounds__(kBlock) void k_emit_lanes(const EmitParams p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    if (g >= p.ngroups) return;
    if (p.groupcnt[g] == 0) return;
    emit_group_lanes<FROM_LUT, ALLSEEN, EB, INDIRECT>(p, g, lane, emit_load_group<INDIRECT>(p, g, lane));
}

// The same expansion driven by the list of busy groups (k_finish_scan): a fixed grid of waves strides over
// it, so no wave is launched only to find its group empty (5 of 6 are).  A wave loads its next group's words and
// offsets before it works on the current one (and that group's number one step earlier still).
template <bool FROM_LUT, bool ALLSEEN, int EB>
__global__ __launch_bounds__(kBlock) void k_emit_busy(const EmitParams p)
{
    if (p.dbg & 16u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (kBlock / 64);
    const uint32_t nbusy = p.busycount[0];
    if (w >= nbusy) return;
    uint32_t g = p.busylist[w];
    uint32_t g1 = w + nwaves < nbusy ? p.busylist[w + nwaves] : 0u;
    EmitGroup h = emit_load_group<false>(p, g, lane);
    for (uint32_t t = w; t < nbusy; t += nwaves) {
        const bool more = t + nwaves < nbusy;                          // (wave-uniform)
        const uint32_t g2 = t + 2 * nwaves < nbusy ? p.busylist[t + 2 * nwaves] : 0u;
        EmitGroup hn = h;
        if (more) hn = emit_load_group<false>(p, g1, lane);
        // (a group holds at most 4 096 records: where that many fit behind its first one, nothing is tested per record)
        if (FROM_LUT && h.out0 + 4096u <= p.capacity) emit_group_lanes<FROM_LUT, ALLSEEN, EB, false, true>(p, g, lane, h);
        else emit_group_lanes<FROM_LUT, ALLSEEN, EB, false>(p, g, lane, h);
        h = hn; g = g1; g1 = g2;
    }
}

}  // namespace vc
