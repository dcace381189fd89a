// libvoxcarve.so -- MI355X (gfx950) visual-hull carve engine: kernels + C ABI.
//
// Path replaced (reference root = ChristosP1/Voxel-Based-3D-Reconstruction):
//   create_voxel_volume                        voxel_reconstruction.py:35-59
//   create_lookup_table (cv2.projectPoints)    voxel_reconstruction.py:62-86
//   update_visible_voxels_and_extract_colors   voxel_reconstruction.py:89-124
//   selection loop of set_voxel_positions      assignment.py:116-133
//
// Data layout in HBM (per context = per rank = one z-slab of n voxels, slab-local j):
//   axes      f64 xs[nx], ys[ny], zs[nz]        np.linspace tables (host-built, exact)
//   maskbits  u32 [slot][C][ceil(H*W/32)]       bit b of word w = pixel 32w+b foreground
//   frames    u8  [slot][C][H*W*3]              BGR, only the colour camera is read
//   lut       i32 [C][n]                        pixel offset or -1 (VC_MODE_LUT)
//   words     u64 [ceil(n/64)]                  survivor bit per voxel (= dense occupancy)
//   groupcnt  u32 [n_pad/4096]                  survivors per group of 64 words
//   groupoff  u32 [groups], blockoff u64        two-level exclusive scan of groupcnt
//   records   u64 [S]                           {u32 idx, r, g, b, seen}, ascending idx
//
// There is no CPU path in this library: without a GPU vc_create fails (VC_ERR_NODEV).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only; the functions are resolved with dlsym

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/voxcarve.h"
#include "vc_kernels.h"
#include "vc_mc.h"
#include "vc_fg.h"

#pragma clang fp contract(off)

using namespace vc;

namespace {

// ================================================================ host side
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;
std::string g_create_error;

// RCCL is resolved at first use so single-GPU runs never load it, and so that a
// process which already holds a librccl.so.1 (any host framework) shares that copy.
bool load_rccl(std::string &err)
{
    if (g_rccl.handle) return true;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { err = std::string("dlopen librccl: ") + dlerror(); return false; }
#define VC_SYM(field, name)                                                             \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));            \
    if (!g_rccl.field) { err = std::string("dlsym ") + name + " failed"; return false; }
    VC_SYM(GetUniqueId, "ncclGetUniqueId")
    VC_SYM(CommInitRank, "ncclCommInitRank")
    VC_SYM(CommDestroy, "ncclCommDestroy")
    VC_SYM(AllGather, "ncclAllGather")
    VC_SYM(Broadcast, "ncclBroadcast")
    VC_SYM(AllReduce, "ncclAllReduce")
    VC_SYM(GroupStart, "ncclGroupStart")
    VC_SYM(GroupEnd, "ncclGroupEnd")
    VC_SYM(GetErrorString, "ncclGetErrorString")
#undef VC_SYM
    g_rccl.handle = h;
    return true;
}

template <typename T>
struct DevBuf {
    T *ptr = nullptr;
    size_t cap = 0;     // elements
};

// One resident frame set.  The uploaded bytes stay on the device (bytes / fbytes), so the derived state (bit masks,
// record-layout images, block grids, camera order -- all made on the device by k_prep_pack / k_prep_grid, queued in front of
// the first carve that uses the slot) can be re-derived without another transfer (vc_touch_masks).
struct Slot {
    DevBuf<uint8_t> bytes;      // [C][H*W] byte masks as uploaded
    uint8_t *h_bytes = nullptr; // page-locked staging of the same size: uploads are asynchronous
    size_t h_bytes_cap = 0;
    DevBuf<uint8_t> fbytes[VC_MAX_CAMERAS];   // [H*W*3] BGR image of a camera as uploaded
    uint8_t *h_fbytes[VC_MAX_CAMERAS] = {nullptr};
    DevBuf<uint32_t> bits;      // [C][mwords]
    DevBuf<uint32_t> frames;    // [C][H*W] one dword per pixel: R | G << 8 | B << 16 | seen << 24 (a record's upper half)
    std::vector<uint8_t> have_frame, frame_dirty;
    bool have_masks = false;    // byte masks staged
    bool bits_valid = false;    // bits, record-layout images and the grid plan match the staged bytes
    bool grids_valid = false;   // block grids and camera order too (they also depend on grid, slab and cameras)
    DevBuf<uint32_t> grid;      // header + cropped block grids of all cameras (hierarchical kernels stage it in LDS)
    DevBuf<uint32_t> coarse;    // the same with 4 x 4 times coarser blocks, for the brick level (frame sets with large grids only)
    bool has_coarse = false;
    DevBuf<uint32_t> boxes;     // the cameras' foreground pixel boxes, two sets (see kBoxStride)
    uint32_t budget_words = 0;  // LDS budget the plan was made for (fixes the dynamic LDS size of the carve launch)
    uint32_t parity = 0;        // which of the header's two foreground-box sets the current frame filled
    bool counts_zero = false;   // the header's sample counts are zero (k_prep_pack just ran)
    hipEvent_t e_up = nullptr;  // last upload into this slot (owned; upload stream)
    // The per-frame preparation runs on the UPLOAD stream, right behind the copy it works on and beside whatever the carve
    // stream is doing for the step before; e_prep (owned) marks its end, the carve waits for it.  e_p0: its start when timed.
    hipEvent_t e_prep = nullptr, e_p0 = nullptr;
    bool prep_pending = false, prep_timed = false;
    // borrowed from the step that used the slot last (recording an event between two kernels costs ~10 us of stream time,
    // so the slot rides on events a step records anyway): behind the last carve kernels that read the slot's bits / grids,
    // and behind the last record expansion that read its bits / images on the second stream.  The next preparation waits for both.
    hipEvent_t e_carve = nullptr, e_emit = nullptr;
    bool up_pending = false, carve_pending = false, emit_pending = false;
    uint32_t gen = 0;           // preparations so far: a step remembers the one it ran on (vc_carve_end's regrow path)
};

// np.linspace(lo, hi, num=n) in float64: y[k] = k*step + lo (two roundings), y[n-1] = hi
// (voxel_reconstruction.py:52-54).  Host code of this file is built contraction-free too.
void linspace(double lo, double hi, uint32_t n, std::vector<double> &out)
{
    out.resize(n);
    if (n == 0) return;
    if (n == 1) { out[0] = lo; return; }
    const double delta = hi - lo;
    const double div = (double)(n - 1);
    const double step = delta / div;
    for (uint32_t k = 0; k < n; ++k) {
        const double kk = (double)k;
        const double y = (step == 0.0) ? (kk / div) * delta : kk * step;
        out[k] = y + lo;
    }
    out[n - 1] = hi;
}

}  // namespace

// One carve step's device state.  Two of them exist so that step i+1 can be enqueued before the
// host has collected step i (vc_carve_begin / vc_carve_end): the device never idles between steps.
struct StepBuf {
    DevBuf<uint64_t> words;
    DevBuf<uint32_t> groupcnt, groupoff;
    DevBuf<uint32_t> groupnz;                // non-zero words per group (brick pipeline, compact exchange: k_assemble counts them in passing)
    bool nz_valid = false;
    DevBuf<uint64_t> blocksum, blockoff;     // blockoff[nscan] = total
    DevBuf<uint64_t> records;
    uint64_t *h_total = nullptr;             // pinned
    hipEvent_t e0 = nullptr, e_first = nullptr, e1 = nullptr, e_prep = nullptr;
    hipEvent_t e2 = nullptr, e_scan = nullptr, e_emit0 = nullptr;   // borrowed from vc_ctx::step_ev for the step in this set (see there)
    bool emit_ridden = false;                // e_emit0 / e2 are the expansion launch's own begin and end
    // option timing_detail: begin / end events of this step's kernels by kind (owned, made on first use; they ride on the launches),
    // which pair each kind used (the expansion and k_finish_scan may carry the step's own events instead), and which kinds ran
    hipEvent_t kev[VC_KERNEL_KINDS][2] = {};
    hipEvent_t kused[VC_KERNEL_KINDS][2] = {};
    uint32_t kmask = 0;
    bool prepped = false, prep_timed = false; // this step queued preparation kernels in front of its carve (timed: e_prep .. e0)
    bool carve_timed = false;                // e0 / e1 were recorded around the carve kernels (synchronous calls, timing_detail)
    bool emit_timed = false;                 // e_scan / e2 bracket the record expansion
    bool pending = false, used = false;
    EmitParams emit;                         // kept for a re-run after a records regrow
    bool allseen = false, want_vm = false, has_first = false;
    bool no_records = false;                 // VC_FLAG_NO_RECORDS: occupancy words + count only
    bool sparse_words = false;               // words of groups with groupcnt == 0 were left unwritten
    DevBuf<uint32_t> busyoff, busysum, busyblock, busylist;   // the groups with survivors, listed by the scan
    bool busy = false;
    // compact exchange form of this step: non-zero words as {bits, global index of bit 0} pairs
    DevBuf<uint64_t> ent, mine, counts;      // pairs | {entries, survivors} of this rank | of all ranks
    uint64_t *h_counts = nullptr;            // pinned, 2 per rank
    bool counts_exchanged = false;           // vc_carve_begin already packed and all-gathered the counts
    int mode = 0, color_cam = -1;            // what the step was run with (vc_expand_entries colours the same way)
    uint32_t slot = 0, slot_gen = 0;         // frame set the step read, and which preparation of it
    uint64_t n = 0, survivors = 0;
};

constexpr int kDepth = 3;                   // sets of result buffers = carve steps that may be in flight: with two, the host cannot queue step i + 1
                                            // before it has collected step i - 1, whose expansion ends when the carve of step i does -- the carve
                                            // stream then idles for the host's round trip (20 us of a 155 us step)
constexpr uint32_t kStepRing = 64;
constexpr uint32_t kGatherRing = 32;        // steps before a gather's events are recorded again (more than the resident frame sets a stream cycles through)
struct vc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // scan + record expansion of step i, beside the carve of step i+1 on `stream`
    hipStream_t stream_up = nullptr; // host-to-device copies of masks and images (overlap the carve in flight)
    hipStream_t stream_x = nullptr;  // a rank of a communicator: packing + collectives of step i, beside the carve of step i + 1 (they waited in
                                     // line on the carve stream: three launches, two collectives and their events, ~60 us per step)
    hipEvent_t ev_h[2] = {nullptr, nullptr};   // around the last mask upload (h2d_ms)
    bool h2d_pending = false;
    int overlap = 1;                 // (one stream when a communicator is attached: its collectives order everything)
    // How the streams share the chip.  The record expansion fills every wave slot (65 536 waves of streaming work); the carve
    // chain is a row of short, latency-bound launches that then queue for slots behind it.  stream_priority: carve + preparation
    // streams at the highest queue priority, the expansion stream at the lowest -- the dispatcher hands a freed slot to the
    // carve chain first.  reserve_cus: k compute units per XCD (k x 8 of 256) are left out of the expansion stream's CU mask
    // (hipExtStreamCreateWithCUMask; no priority then: that call has none), so the carve chain always finds free slots there.
    int stream_priority = 1;
    int reserve_cus = 0;
    int launch_events = 1;           // the events the streams exchange ride on the launch that precedes them (hipExtLaunchKernelGGL's stop event: the
                                     // kernel's own completion signal) instead of a barrier packet of their own behind it (~3.5 us of stream time each)
    int event_scope = 1;             // 1: the events the streams exchange release to the DEVICE only (no system-scope write-back)
    StepBuf sb[kDepth];
    int head = 0, npending = 0, cur = -1;    // next set to issue into, steps in flight, set holding the fetched result
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // the compact all-gather's events {start, expansion done, payload arrived}, a RING of them: a frame set's next preparation
    // waits for the expansion that read it (Slot::e_emit), many steps later -- one event re-recorded every step would make it
    // wait for the newest expansion instead and put carve, exchange, expansion and preparation in one line
    hipEvent_t gx[kGatherRing][3] = {};
    // the same for a step's {scan done, step done}: frame sets remember them (Slot::e_carve, e_emit) for their next preparation,
    // kStepRing steps of distance keep that wait on the step that read the frame set and not on a newer one
    hipEvent_t step_ev[kStepRing][3] = {};      // {scan done, step done, expansion begun}
    uint32_t step_next = 0;
    uint32_t gx_next = 0;
    std::string err;

    // grid
    uint32_t nx = 0, ny = 0, nz = 0, z0 = 0, z1 = 0;
    double bounds[6] = {0, 0, 0, 0, 0, 0};
    std::vector<double> xs, ys, zs;
    DevBuf<double> d_axes;           // xs | ys | zs
    bool have_grid = false;

    // cameras
    uint32_t C = 0, H = 0, W = 0, mwords = 0;
    CamDev cams[VC_MAX_CAMERAS];
    bool have_cams = false;

    std::vector<Slot> slots;
    uint8_t post_open[VC_MAX_CAMERAS] = {0}, post_close[VC_MAX_CAMERAS] = {0};   // 2x2 open / close per camera
    DevBuf<uint8_t> d_morph;         // post-filtered byte masks [C][H*W] + one scratch image

    DevBuf<int32_t> d_lut;
    DevBuf<uint64_t> d_bbox;         // [C][n_pad/64] per-word pixel boxes (built with the LUT)
    DevBuf<int32_t> d_lut_tile;      // the table in tile order (words of 4 x-rows x 16 y), when the grid allows
    DevBuf<uint64_t> d_tbox;         // pixel boxes of the tile words
    DevBuf<uint64_t> d_kbox;         // pixel boxes of the 16^3 bricks (64 tile words each)
    DevBuf<uint64_t> d_live;         // per frame set: bit per brick "may hold survivors" | "all voxels survive" (k_cull)
    // brick pipeline (k_cull_bricks -> k_brick_words -> k_voxel_words -> k_assemble)
    DevBuf<uint64_t> d_wbox;         // [C][nbrick_pad * 64] brick-major word boxes (geometry only)
    DevBuf<uint64_t> d_bm;           // [n_pad / 64] tile-word results of the current step, tile order
    DevBuf<uint32_t> d_blist;        // counters [8] | brick list [nbrick_pad] | column list
    DevBuf<uint64_t> d_wlist;        // undecided words (worst case: every word of the slab)
    uint32_t *h_lists = nullptr;     // pinned [4]: list lengths of an earlier step, to size launches by
    // marching cubes (vc_marching_cubes)
    DevBuf<uint64_t> d_mcbits, d_mcx;
    DevBuf<uint32_t> d_mcwbase, d_mcgv, d_mcgt, d_mcgvoff, d_mcgtoff, d_mcfaces;
    DevBuf<uint64_t> d_mcbv, d_mcbvoff, d_mcbt, d_mcbtoff;
    DevBuf<float> d_mcverts;
    uint64_t mc_verts = 0, mc_faces = 0;
    bool mc_valid = false;
    uint32_t list_parity = 0;
    uint32_t cull_probe = 0;         // steps that skipped the brick-level tests (launch_bricks: every 64th looks again)
    int bricks = 1;                  // the brick pipeline where the grid shape allows (ny in {256, 512, 1024})
    int dbg = 0;
    bool big_lds_ok = false;
    int voxel_batches = 0;                   // k_voxel_words: batches of 8 words a wave takes one after the other; 0 = by kernel form (launch_bricks)
    int voxel_pairs = 0;                     // k_voxel_words: 0 = by camera count, 1 = two cameras per round, 2 = one
    bool kbox_valid = false;
    int cull = 1;                    // hierarchical kernels on tile words: cull whole bricks first
    bool tile_valid = false;
    bool bbox_valid = false, tbox_valid = false;   // boxes match the grid, slab and cameras (also built without a table)
    int grid_lds_kb = 0;             // LDS budget of a frame set's block grids (picks their resolution); 0 = by frame-set size (16 or 64)
    int grid_min_shift = 1;          // finest block: 2^shift pixels
    int lut_tile = 1;                // hierarchical LUT kernel on tile words (needs nx % 4 == 0, ny % 64 == 0)
    int fused_tile = 1;              // the same word shape for the hierarchical table-free kernel
    int fused_f32box = 1;            // its word boxes from float32 intervals after a float64 rigid transform ...
    int fused_color_table = 1;       // VC_MODE_FUSED: colour the survivors from the colour camera's table (one camera, whole grid; 0: project each survivor again)
    int fused_boxes = 1;             // ... or read from boxes reduced once from the exact pixels (no table involved)
    bool lut_valid = false;          // vc_build_lut ran for this grid / slab / cameras (tile-ordered table, or y-major where tiles do not apply)
    uint32_t upload_mask = 0;        // cameras handed in by vc_upload_lut so far
    bool lut_foreign = false;        // the table in use came in through vc_upload_lut
    bool ymajor_valid = false;       // the y-major table + y-line boxes exist (built on demand: streaming / generic kernels, vc_fetch_lut)
    // tuning knobs (vc_set_option); defaults are the measured best on MI355X
    bool force_generic = false;      // one-thread-per-voxel kernels only (cross-check path)
    int first_kv = 1;                // dwordx4 loads per lane per chunk in k_lut_first: 1, 2 or 4
    int first_blocks_per_cu = 3;     // k_lut_first workgroups (512 threads) per CU
    int refine_b = 8;                // alive words per batch in k_lut_refine: 8 or 16
    int refine_blocks_per_cu = 8;    // k_lut_refine workgroups (256 threads) per CU
    int fused_blocks_per_cu = 8;     // k_carve_fused workgroups (256 threads) per CU
    int reorder = 1;                 // visit the most selective camera first
    int hier_blocks_per_cu = 48;     // hierarchical kernel: oversubscribed grid, the dispatcher balances uneven groups
    int refine_pair = 1;             // hierarchical LUT kernel: two cameras per dependent round trip
    int emit_lanes = 1;              // record expansion: lanes = voxels of a word (1) or lanes = survivors (0)
    int emit_busy = 1;               // ... driven by the list of busy groups (grids of >= kBusyListMinGroups groups)
    int emit_waves_per_cu = 256;     // waves of that launch per CU (a wave strides over the list when there are more busy groups)
    int fused_hier = 1;              // VC_MODE_FUSED: interval-arithmetic word rejection (needs ny % 64 == 0)
    int lut_hier = 1;                // VC_MODE_LUT: hierarchical kernel (boxes + block grid) instead of stream + refine
    int timing_detail = 0;           // also time preparation and carve kernels of pipelined steps (three more events on the carve stream)
    int kernel_events = 0;           // every launch of a step carries begin / end events of its own (no packet of their own: vc_timing_t::kernel_ms_sum)
    bool sync_call = false;          // inside vc_carve: the step is collected at once, events between its kernels cost nothing that matters
    DevBuf<uint16_t> d_viewmask;
    DevBuf<double> d_scratch;
    uint64_t *h_total = nullptr;     // pinned scalar (all-gather count)
    bool viewmask_valid = false, carved = false;
    uint64_t survivors = 0;

    // comm
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
    DevBuf<uint64_t> d_counts, d_gathered;
    uint64_t *h_counts = nullptr;    // pinned, n_ranks
    uint64_t gathered_total = 0;
    bool gathered = false;
    // compact exchange: non-zero words of the slab as {bits, global index of bit 0} pairs
    int gather_compact = 1;          // vc_allgather exchanges the pairs and expands them on every rank
    int gather_sync = 1;             // 0: vc_allgather returns once its work is queued (count known from the ranks' counts)
    // Two compact gathers may be in flight (gather_sync 0): gather k uses half k & 1 of what follows -- its payload buffer
    // (all ranks' pairs in rank order: the next payload arrives while the expansion of this one still reads it), its events, its
    // expected total.  The records go to the one d_gathered: expansions are in order on their stream and a read-back ends them.
    bool gpend[2] = {false, false};  // a queued all-gather whose completion has not been observed yet
    uint32_t gx_idx[2] = {0, 0};
    uint64_t gexpect[2] = {0, 0};
    uint32_t gseq = 0;               // compact gathers issued
    DevBuf<uint64_t> d_ent_all[2];
    DevBuf<uint32_t> d_xcnt, d_xoff;         // scan scratch of the pack pass ...
    DevBuf<uint64_t> d_xbsum, d_xboff;
    DevBuf<uint32_t> d_ycnt, d_yoff;         // ... and of the expansion, which may run on the second stream beside a pack
    DevBuf<uint64_t> d_ybsum, d_yboff;
    uint64_t *h_xtotal = nullptr;            // pinned
    uint64_t packed_entries = 0;
    bool packed = false;
    DevBuf<int32_t> d_lut_color;             // colour camera's table over the WHOLE grid (expansion of remote words)
    int lut_color_cam = -1;

    DevBuf<uint8_t> d_fg;            // vc_bgr_to_hsv / vc_mask_morphology: input | output | scratch images
    DevBuf<int32_t> d_hsvdiv;        // OpenCV's two division tables of the 8-bit HSV conversion (sdiv | hdiv)
    struct MogModel {                // vc_mog_*: one background model (the reference keeps one per camera, assignment.py:79)
        bool used = false;
        int history = 200, nmixtures = 5;
        double background_ratio = 0.7, noise_sigma = 15.0;
        uint32_t H = 0, W = 0, nframes = 0;
        DevBuf<float> state;         // [8 nmixtures][H W] planes, see k_mog_apply
    } mog[VC_MAX_MOG_MODELS];
    vc_timing_t tm;
    StepBuf *kev_sb = nullptr;       // timing_detail: the step whose kernels are being queued (their launches carry its per-kind events)
    DevBuf<unsigned long long> d_stats;   // timing_detail: the kernels' work counters, [VC_WORK_KINDS][kShards][kStatStride]

    uint64_t n_voxels() const { return (uint64_t)nx * ny * (z1 - z0); }
    uint64_t i0() const { return (uint64_t)z0 * nx * ny; }
};

namespace {

int fail(vc_ctx *ctx, int code, const char *fmt, ...);

// timing_detail: the begin / end events launch `kind` of the step being queued is to carry (null otherwise: an ordinary launch)
void kev_pick(vc_ctx *ctx, int kind, hipEvent_t &start, hipEvent_t &stop)
{
    start = stop = nullptr;
    StepBuf *sb = ctx->kev_sb;
    if (!sb) return;
    for (int i = 0; i < 2; ++i)
        if (!sb->kev[kind][i] && hipEventCreate(&sb->kev[kind][i]) != hipSuccess) return;
    start = sb->kev[kind][0]; stop = sb->kev[kind][1];
    sb->kused[kind][0] = start; sb->kused[kind][1] = stop;
    sb->kmask |= 1u << kind;
}
#define VC_KLAUNCH(kind, kernel, grid, block, lds, st, ...)                                             \
    do {                                                                                                \
        hipEvent_t ks_, ke_;                                                                            \
        kev_pick(ctx, kind, ks_, ke_);                                                                  \
        hipExtLaunchKernelGGL(kernel, grid, block, lds, st, ks_, ke_, 0, __VA_ARGS__);                  \
    } while (0)

// (Re)creates the three streams for ctx->stream_priority / ctx->reserve_cus.  Nothing may be in flight.
hipError_t make_streams(vc_ctx *ctx)
{
    hipStream_t *all[4] = {&ctx->stream, &ctx->stream2, &ctx->stream_up, &ctx->stream_x};
    for (hipStream_t *st : all) {
        if (!*st) continue;
        hipError_t e = hipStreamSynchronize(*st);
        if (e == hipSuccess) e = hipStreamDestroy(*st);
        if (e != hipSuccess) return e;
        *st = nullptr;
    }
    int least = 0, greatest = 0;                                 // numerically: greatest priority <= least priority
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    const bool prio = ctx->stream_priority && least != greatest;
    if (ctx->reserve_cus > 0) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, ctx->device);
        if (e != hipSuccess) return e;
        // the driver deals the mask's bits round-robin over the XCDs (bit i -> XCD i % 8): the first 8 k bits are k compute
        // units of every XCD
        const uint32_t ncu = (uint32_t)prop.multiProcessorCount, words = (ncu + 31) / 32;
        const uint32_t reserved = (uint32_t)ctx->reserve_cus * 8u < ncu ? (uint32_t)ctx->reserve_cus * 8u : ncu / 2;
        std::vector<uint32_t> rest(words, 0u);
        for (uint32_t i = reserved; i < ncu; ++i) rest[i >> 5] |= 1u << (i & 31u);
        e = hipExtStreamCreateWithCUMask(&ctx->stream2, words, rest.data());
        if (e != hipSuccess) return e;
    } else {
        e = prio ? hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, least)
                 : hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    e = prio ? hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    e = prio ? hipStreamCreateWithPriority(&ctx->stream_up, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&ctx->stream_up, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    return prio ? hipStreamCreateWithPriority(&ctx->stream_x, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&ctx->stream_x, hipStreamNonBlocking);
}

// The events that only order one stream behind another ({scan done} of a step: carve stream -> expansion stream; a frame set's
// {prepared}: upload stream -> carve stream) are recorded between two kernels of the critical path.  By default an event
// performs a SYSTEM-scope release when it is recorded (the host may want to look at what came before it): on this chip that is
// a write-back of every XCD's L2 -- with the expansion's 238 MB of records in flight beside it, ~12 us during which the recording
// stream stands still.  Nobody on the host ever looks at anything through these events: hipEventReleaseToDevice.  event_scope 2
// does the same to a step's {done} event, which the host DOES wait for (the survivor count it then reads sits in page-locked
// host memory, written past the caches).
hipError_t make_events(vc_ctx *ctx)
{
    for (uint32_t r = 0; r < kStepRing; ++r) {
        for (int i = 0; i < 3; ++i) {
            if (ctx->step_ev[r][i]) { hipError_t e = hipEventDestroy(ctx->step_ev[r][i]); if (e != hipSuccess) return e; ctx->step_ev[r][i] = nullptr; }
            const bool dev = ctx->event_scope >= (i == 1 ? 2 : 1);
            hipError_t e = hipEventCreateWithFlags(&ctx->step_ev[r][i], dev ? hipEventReleaseToDevice : hipEventDefault);
            if (e != hipSuccess) return e;
        }
    }
    for (Slot &s : ctx->slots) {
        if (!s.e_prep) continue;
        hipError_t e = hipEventDestroy(s.e_prep);
        if (e != hipSuccess) return e;
        e = hipEventCreateWithFlags(&s.e_prep, ctx->event_scope >= 1 ? hipEventReleaseToDevice : hipEventDefault);
        if (e != hipSuccess) return e;
        s.prep_pending = s.carve_pending = s.emit_pending = false;   // (everything has drained: nothing to wait for)
    }
    return hipSuccess;
}

int fail(vc_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define VC_HIP(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, e_ == hipErrorOutOfMemory ? VC_ERR_OOM : VC_ERR_HIP, "%s: %s",    \
                        #call, hipGetErrorString(e_));                                         \
    } while (0)

#define VC_NCCL(ctx, call)                                                                     \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess)                                                                 \
            return fail(ctx, VC_ERR_RCCL, "%s: %s", #call, g_rccl.GetErrorString(r_));         \
    } while (0)

template <typename T>
int ensure(vc_ctx *ctx, DevBuf<T> &b, size_t elems)
{
    if (elems <= b.cap && b.ptr) return VC_OK;
    if (b.ptr) { VC_HIP(ctx, hipFree(b.ptr)); b.ptr = nullptr; b.cap = 0; }
    if (elems == 0) elems = 1;
    VC_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&b.ptr), elems * sizeof(T)));
    b.cap = elems;
    return VC_OK;
}

template <typename T>
void release(DevBuf<T> &b)
{
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.cap = 0;
}

#define VC_TRY(expr)              \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != VC_OK) return rc_; \
    } while (0)

uint32_t grid_for(uint64_t n);

void fill_params(const vc_ctx *ctx, CarveParams &p)
{
    memset(&p, 0, sizeof p);
    p.xs = ctx->d_axes.ptr;
    p.ys = p.xs + ctx->nx;
    p.zs = p.ys + ctx->ny;
    p.n = ctx->n_voxels();
    p.n_pad = (p.n + kLutPad - 1) / kLutPad * kLutPad;
    p.nx = ctx->nx; p.ny = ctx->ny; p.nz = ctx->nz; p.z0 = ctx->z0;
    p.C = ctx->C; p.H = ctx->H; p.W = ctx->W; p.mwords = ctx->mwords;
    memcpy(p.cam, ctx->cams, sizeof(CamDev) * ctx->C);
    p.bbox = ctx->d_bbox.ptr;
    p.lut_tile = ctx->d_lut_tile.ptr; p.tbox = ctx->d_tbox.ptr; p.tq = ctx->ny / 16;
    p.tile_whole = (p.tq != 0 && 64 % p.tq == 0) ? 1u : 0u;
    p.kbox = ctx->d_kbox.ptr;
    p.dbg = (uint32_t)ctx->dbg;
    p.live = nullptr;                                            // set by the launches that cull
    p.nbx = ((ctx->nx >> 2) + 3) / 4;
    p.nbz = (ctx->z1 - ctx->z0 + 15) / 16;
    p.nbrick_pad = (uint32_t)(((uint64_t)p.nbx * p.tq * p.nbz + 63) / 64 * 64);
}

constexpr int VC_MAX_RANKS = 64;
constexpr uint32_t kBusyListMinGroups = 16384;   // below 64 M voxels a wave per group is as fast and one launch shorter
constexpr uint32_t kMaxScanBlocks = 1024;  // 2^32 voxels / 4096 per group / 1024 groups per scan block
constexpr int kSub = 4;                    // 64-voxel sub-chunks per wavefront chunk (fused kernel)
constexpr size_t kLdsBytes = 160 * 1024;   // LDS per CU on gfx950
constexpr size_t kMaxFirstLds = 64 * 1024; // static limit of one workgroup's dynamic LDS without opt-in
constexpr size_t kWideGridBytes = 20 * 1024; // grids above this: 1024-thread workgroups share a copy, the brick level reads coarser blocks
constexpr size_t kMaxWideLds = 152 * 1024; // what the brick pipeline's grid-staging kernels may take (one 1024-thread workgroup per CU)
constexpr uint32_t kEstimateSamples = 1u << 16;

// The bricks' pixel boxes and the brick-major copy of the word boxes (once per grid / slab / camera set, right behind the
// tile boxes).
int build_brick_boxes(vc_ctx *ctx)
{
    CarveParams p;
    fill_params(ctx, p);
    ctx->kbox_valid = false;
    if (p.nbrick_pad == 0) return VC_OK;
    VC_TRY(ensure(ctx, ctx->d_kbox, (size_t)p.nbrick_pad * ctx->C));
    VC_TRY(ensure(ctx, ctx->d_live, (size_t)(p.nbrick_pad / 64) * 2));
    VC_TRY(ensure(ctx, ctx->d_wbox, (size_t)p.nbrick_pad * 64 * ctx->C));
    p.kbox = ctx->d_kbox.ptr;
    hipLaunchKernelGGL(k_brick_boxes_bm, dim3(p.nbrick_pad / 4), dim3(kBlock), 0, ctx->stream, p, (const uint64_t *)ctx->d_tbox.ptr,
                       ctx->d_wbox.ptr, ctx->d_kbox.ptr);
    VC_HIP(ctx, hipGetLastError());
    ctx->kbox_valid = true;
    if (ctx->h_lists) ctx->h_lists[0] = ctx->h_lists[1] = ctx->h_lists[2] = 0xffffffffu;
    return VC_OK;
}

// Grid shapes the brick pipeline takes: a group of 4096 consecutive voxels must lie inside one brick column (see k_assemble).
bool brick_shape(const vc_ctx *ctx, const CarveParams &p)
{
    if (!ctx->bricks || !ctx->cull || !ctx->kbox_valid) return false;
    if (ctx->ny != 256 && ctx->ny != 512 && ctx->ny != 1024 && ctx->ny != 2048 && ctx->ny != 4096) return false;
    if (ctx->nx % 4 != 0 || (ctx->ny < 1024 && ctx->nx % (4096u / ctx->ny) != 0)) return false;
    return true;
}

uint32_t sized(uint32_t known, uint64_t unknown_guess, uint64_t cap, uint32_t per_wg)
{
    uint64_t est = known == 0xffffffffu ? unknown_guess : (uint64_t)known + known / 4 + 64;
    if (est > cap) est = cap;
    uint64_t wgs = (est + per_wg - 1) / per_wg;
    if (wgs < 64) wgs = 64;
    if (wgs > 65536) wgs = 65536;
    return (uint32_t)wgs;
}

template <bool LUT>
int launch_bricks(vc_ctx *ctx, CarveParams &p, size_t lds, uint32_t ngroups)
{
    const uint32_t ncolumns = p.nbx * p.nbz;
    const uint32_t ipw = p.tq > 64 ? p.tq / 64 : 1;                           // rounds of 64 bricks per wave of k_cull_bricks (a whole column)
    const uint32_t nw = p.nbrick_pad / 64 / ipw;                              // waves of k_cull_bricks
    const uint32_t wps = (nw + kShards - 1) / kShards;                        // producers per shard
    BrickLists bl;
    bl.cap_b = wps * 64 * ipw; bl.cap_c = wps * 4;
    bl.cap_w = (p.nbrick_pad + kShards - 1) / kShards * 64;                   // a listed brick appends at most its 64 words
    VC_TRY(ensure(ctx, ctx->d_bm, (size_t)(p.n_pad / 64)));
    VC_TRY(ensure(ctx, ctx->d_wlist, (size_t)bl.cap_w * kShards));
    const size_t ncounters = 6 * (size_t)kShards * kShardStride;
    const size_t need = ncounters + (size_t)bl.cap_b * kShards + (size_t)bl.cap_c * kShards + 64;
    if (ctx->d_blist.cap < need) {
        VC_TRY(ensure(ctx, ctx->d_blist, need));
        VC_HIP(ctx, hipMemsetAsync(ctx->d_blist.ptr, 0, ncounters * sizeof(uint32_t), ctx->stream));    // both sets of list lengths
        ctx->list_parity = 0;
    }
    if (!ctx->h_lists) {
        VC_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_lists), 4 * sizeof(uint32_t), hipHostMallocDefault));
        ctx->h_lists[0] = ctx->h_lists[1] = ctx->h_lists[2] = ctx->h_lists[3] = 0xffffffffu;       // unknown yet
    }
    bl.counters = ctx->d_blist.ptr;
    bl.bricks = ctx->d_blist.ptr + ncounters;
    bl.columns = bl.bricks + (size_t)bl.cap_b * kShards;
    bl.words = ctx->d_wlist.ptr;
    bl.bm = ctx->d_bm.ptr;
    bl.wbox = ctx->d_wbox.ptr;
    bl.host_counts = ctx->h_lists;
    bl.parity = (ctx->list_parity ^= 1u);
    p.live = ctx->d_live.ptr;
    const volatile uint32_t *known = ctx->h_lists;                // lengths of an earlier step (any size is correct: the waves stride)
    const uint32_t k_bricks = known[0], k_cols = known[1], k_words = known[2];
    // large grids (many cameras x large images): 16 waves share one LDS copy, so that the compute units stay full of waves
    // with two or three workgroups each; no more workgroups than fit the chip at once (they stride over the lists)
    const bool wide = lds > kWideGridBytes;
    const uint32_t wpg = wide ? kWideBlock / 64 : kBlock / 64;                // waves per workgroup
    const dim3 block(kBlock), gblock(wpg * 64);
    const uint32_t fit = 256u * (uint32_t)(kLdsBytes / (lds ? lds : 1) < 1 ? 1 : kLdsBytes / (lds ? lds : 1));
    const uint32_t lds_cap = wide ? fit : 65536u;
    const uint32_t cw = nw / wpg;
    uint32_t cull_wgs = cw < 1 ? 1 : (cw > 1024 ? 1024 : cw);
    if (cull_wgs > lds_cap) cull_wgs = lds_cap;
    uint32_t word_wgs = sized(k_bricks, p.nbrick_pad / 8, p.nbrick_pad, wpg);
    if (word_wgs > lds_cap) word_wgs = lds_cap;
    if (lds > kMaxFirstLds && !ctx->big_lds_ok) {                             // more than 64 KB of dynamic LDS is opt-in
        VC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cull_bricks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxWideLds));
        VC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_brick_words_wide), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kMaxWideLds + 6 * 1024)));
        ctx->big_lds_ok = true;
    }
    // Frame sets on which the brick level decides next to nothing (16 noisy cameras: every brick's box holds some foreground block
    // in every camera) pay its 35 us for nothing: when the earlier step whose list lengths are known listed at least nine bricks in ten,
    // this step lists them all without looking (the kernel stages no grids and tests no boxes; the word level decides them all the
    // same), and every 64th such step looks again.  Work only, never results.
    const uint32_t nbricks_all = p.nbx * p.tq * p.nbz;
    const bool cull_pays = k_bricks == 0xffffffffu || (uint64_t)k_bricks * 10u < (uint64_t)nbricks_all * 9u || (ctx->cull_probe++ & 63u) == 0u;
    if (!cull_pays || (ctx->dbg & 8192)) {
        p.cull_lds_words = 1;                                     // (nothing fits: nothing is staged)
        const uint32_t w4 = nw / (kBlock / 64);
        VC_KLAUNCH(VC_K_CULL_BRICKS, k_cull_bricks, dim3(w4 < 1 ? 1u : (w4 > 1024 ? 1024u : w4)), block, 64, ctx->stream, p, bl, ngroups);
        p.cull_lds_words = 0;
    }
    else if (wide && p.coarsegrid) {
        // the brick level reads the 4 x 4 times coarser grids (about a sixteenth of the words): no reason to share an LDS copy among 16
        // waves -- with nw / 16 workgroups it ran on 32 of the 256 compute units at 512^3.  Ordinary workgroups, as many as there are
        // wave loads of bricks; the bound on the coarse grids' length: a sixteenth of the fine ones + one more row and column per camera
        const size_t clds = ((size_t)(lds / sizeof(uint32_t)) / 8 + 64u * p.C + kGridHeader + 8) * sizeof(uint32_t);
        const uint32_t w4 = nw / (kBlock / 64);
        p.cull_lds_words = (uint32_t)((clds < lds ? clds : lds) / sizeof(uint32_t));      // (an estimate: the kernel checks it against the real length)
        VC_KLAUNCH(VC_K_CULL_BRICKS, k_cull_bricks, dim3(w4 < 1 ? 1u : (w4 > 1024 ? 1024u : w4)), block, clds < lds ? clds : lds, ctx->stream, p, bl, ngroups);
        p.cull_lds_words = 0;
    }
    else VC_KLAUNCH(VC_K_CULL_BRICKS, k_cull_bricks, dim3(cull_wgs), gblock, lds, ctx->stream, p, bl, ngroups);
    if (wide) {
        // + 1 KB per wave for the survivors' compaction (vc_kernels.h, brick_words_body) where the LDS has it: a camera mask and a lane number in 32 bits
        const size_t grid_words = (lds / sizeof(uint32_t) + 63u) & ~(size_t)63u;
        const size_t with = (grid_words + (kWideBlock / 64) * 256u) * sizeof(uint32_t);
        const bool compact = with <= kMaxWideLds + 6 * 1024 && p.C > 4 && p.C <= 23 && !(ctx->dbg & 16384);
        p.compact_off = compact ? (uint32_t)grid_words : 0u;
        VC_KLAUNCH(VC_K_BRICK_WORDS, k_brick_words_wide, dim3(word_wgs), gblock, compact ? with : lds, ctx->stream, p, bl);
        p.compact_off = 0;
    }
    else VC_KLAUNCH(VC_K_BRICK_WORDS, k_brick_words, dim3(word_wgs), gblock, lds, ctx->stream, p, bl);
    // many cameras: most voxels fail the first camera they ask, a second camera's entries read in the same round trip would be
    // wasted on them; few cameras: two per dependent round (the lists are short, the kernel is latency bound)
    const bool pairs = !LUT || ctx->voxel_pairs == 1 || (ctx->voxel_pairs == 0 && p.C <= 4);
    // batches a wave takes one after the other (the next one's entries under way): 8 where a batch is short (table look-ups, two cameras
    // per round: 0.1368 -> 0.1345 ms per step at 1024^3 x 4; 16: 0.147), 1 where it is long (projection 0.175 -> 0.19, one camera per round 0.30 -> 0.32)
    const uint32_t vb = ctx->voxel_batches ? (uint32_t)ctx->voxel_batches : (LUT && pairs ? 8u : 1u);
    const dim3 vgrid(sized(k_words, (uint64_t)p.nbrick_pad * 4, (uint64_t)p.nbrick_pad * 64, (pairs ? 32u : 64u) * vb));
    // one wave per workgroup for the two list-driven kernels that need no LDS: a workgroup of four has to find four free wave slots on one
    // compute unit at once, beside an expansion that refills every slot as it frees (the scans' lesson, in small: step -1.3 %)
    const uint32_t wpw = 1u;
    const dim3 vgrid2(vgrid.x * (4u / wpw)), sblock(64u * wpw);
    if (pairs) VC_KLAUNCH(VC_K_VOXEL_WORDS, (k_voxel_words<LUT, true>), vgrid2, sblock, 0, ctx->stream, p, bl);
    else VC_KLAUNCH(VC_K_VOXEL_WORDS, (k_voxel_words<LUT, false>), vgrid2, sblock, 0, ctx->stream, p, bl);
    VC_KLAUNCH(VC_K_ASSEMBLE, k_assemble, dim3(sized(k_cols == 0xffffffffu ? k_cols : k_cols * 16u, (uint64_t)ncolumns * 4, (uint64_t)ncolumns * 16, 4) * (4u / wpw)),
                       sblock, 0, ctx->stream, p, bl);
    VC_HIP(ctx, hipGetLastError());
    return VC_OK;
}

int slot_at(vc_ctx *ctx, uint32_t slot, Slot **out)
{
    if (!ctx->have_cams) return fail(ctx, VC_ERR_ARG, "vc_set_cameras must precede frame uploads");
    if (slot >= 64) return fail(ctx, VC_ERR_ARG, "slot %u out of range (max 64 resident frame sets)", slot);
    if (slot >= ctx->slots.size()) ctx->slots.resize(slot + 1);
    Slot &s = ctx->slots[slot];
    if (s.have_frame.size() != ctx->C) { s.have_frame.assign(ctx->C, 0); s.frame_dirty.assign(ctx->C, 0); }
    if (!s.e_up) {
        VC_HIP(ctx, hipEventCreateWithFlags(&s.e_up, hipEventDisableTiming));
        VC_HIP(ctx, hipEventCreateWithFlags(&s.e_prep, ctx->event_scope >= 1 ? hipEventReleaseToDevice : hipEventDefault));
        VC_HIP(ctx, hipEventCreate(&s.e_p0));
    }
    *out = &s;
    return VC_OK;
}

void release_slot(Slot &s)
{
    release(s.bytes); release(s.bits); release(s.frames); release(s.grid); release(s.coarse); release(s.boxes);
    s.has_coarse = false;
    for (int c = 0; c < VC_MAX_CAMERAS; ++c) {
        release(s.fbytes[c]);
        if (s.h_fbytes[c]) { (void)hipHostFree(s.h_fbytes[c]); s.h_fbytes[c] = nullptr; }
    }
    if (s.h_bytes) { (void)hipHostFree(s.h_bytes); s.h_bytes = nullptr; s.h_bytes_cap = 0; }
    s.have_masks = s.bits_valid = s.grids_valid = false;
    s.have_frame.clear(); s.frame_dirty.clear();
}

uint32_t grid_for(uint64_t n);

// Queues, on the UPLOAD stream (behind the copy of the bytes it reads, beside the carve stream's work for the step before),
// whatever the slot's derived state is missing: bit masks + record-layout images + grid plan (k_prep_pack, after the optional 2x2
// post-filter), and for the chunked / hierarchical kernels the block grids and the camera order (k_prep_grid).  No host
// synchronisation: the kernels leave their results in the slot's header; e_prep marks their end for the carve stream.
int ensure_prepared(vc_ctx *ctx, Slot &s, bool want_grids, const CarveParams *cp, bool timed = false, bool wide_ok = false)
{
    const uint32_t C = ctx->C;
    const size_t HW = (size_t)ctx->H * ctx->W;
    hipStream_t st = ctx->stream_up;
    // grids made for the brick pipeline's 1024-thread workgroups do not fit the other kernels' LDS: prepare again
    if (want_grids && !wide_ok && s.bits_valid && (size_t)s.budget_words * sizeof(uint32_t) + 32 > kMaxFirstLds) s.bits_valid = s.grids_valid = false;
    if (s.bits_valid && !(want_grids && !s.grids_valid)) return VC_OK;
    // the kernels that still read what is about to be overwritten: carve kernels (bits, grids), record expansion (bits, images)
    if (s.carve_pending) { VC_HIP(ctx, hipStreamWaitEvent(st, s.e_carve, 0)); s.carve_pending = false; }
    if (s.emit_pending) { VC_HIP(ctx, hipStreamWaitEvent(st, s.e_emit, 0)); s.emit_pending = false; }
    s.prep_timed = timed;
    s.gen++;
    if (timed) VC_HIP(ctx, hipEventRecord(s.e_p0, st));
    if (!s.bits_valid) {
        VC_TRY(ensure(ctx, s.bits, (size_t)ctx->mwords * C));
        // LDS budget of header + grids: 16 KB (eight workgroups per CU) unless the frame set is so large that 16 KB would
        // force blocks of 32 x 32 pixels on it (16 cameras at 1080p).  The kernels that stage the grids are then launched
        // with few, persistent workgroups; the brick pipeline's run 1024 threads per workgroup, one or two per CU, so the
        // grids may take most of a CU's LDS (blocks of 8 x 8 pixels for 16 cameras at 1080p: masks with salt noise leave
        // 3 in 4 such blocks clean, 1 in 4 blocks of 16 x 16)
        const uint32_t cap_words = (uint32_t)((wide_ok ? 148u : 64u) * 256u);
        uint32_t budget = (uint32_t)ctx->grid_lds_kb * 256u;                  // u32 words of header + grids
        if (budget > cap_words) budget = cap_words;
        if (ctx->grid_lds_kb == 0) {
            budget = 16u * 256u;
            if ((uint64_t)ctx->mwords * C * 4 > (2u << 20)) {
                // what the UNCROPPED grids of all cameras take at the finest block that keeps them within the cap: cropping can
                // then only make the blocks finer, and the workgroups do not reserve more LDS than the grids can fill
                for (uint32_t sh = (uint32_t)ctx->grid_min_shift; sh < 15; ++sh) {
                    const uint64_t bw = ((uint64_t)ctx->W + (1u << sh) - 1) >> sh, bh = ((uint64_t)ctx->H + (1u << sh) - 1) >> sh;
                    const uint64_t total = kGridHeader + (uint64_t)C * 2 * ((bw + 31) / 32) * bh;
                    if (total <= cap_words || sh == 14) { budget = (uint32_t)(total < 16u * 256u ? 16u * 256u : total); break; }
                }
            }
        }
        if (budget < kGridHeader + 128u) budget = kGridHeader + 128u;         // room for every camera's grid at the coarsest block
        if (budget + 8 > s.grid.cap || !s.grid.ptr) {
            VC_TRY(ensure(ctx, s.grid, (size_t)budget + 8));                   // + padding: kernels copy it 16 bytes at a time
            VC_HIP(ctx, hipMemsetAsync(s.grid.ptr, 0, s.grid.cap * sizeof(uint32_t), st));
        }
        if (!s.boxes.ptr) {
            VC_TRY(ensure(ctx, s.boxes, (size_t)3 * kMaxCameras * kBoxStride));
            std::vector<uint32_t> init(s.boxes.cap, 0u);                       // both box sets start empty
            for (uint32_t k = 0; k < 2 * kMaxCameras; ++k) { init[kBoxStride * k] = 0xffffffffu; init[kBoxStride * k + 2] = 0xffffffffu; }
            VC_HIP(ctx, hipMemcpyAsync(s.boxes.ptr, init.data(), init.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            VC_HIP(ctx, hipStreamSynchronize(st));
            s.parity = 0;
        }
        s.budget_words = budget;
        s.parity ^= 1u;
        PrepParams pp;
        memset(&pp, 0, sizeof pp);
        for (uint32_t c = 0; c < C; ++c) {
            pp.src[c] = s.bytes.ptr + HW * c;
            if (!ctx->post_open[c] && !ctx->post_close[c]) continue;
            // MORPH_OPEN = erode, dilate; MORPH_CLOSE = dilate, erode (opening first when both are set); the uploaded
            // bytes stay as they are, the filtered image of camera c goes to d_morph[c]
            VC_TRY(ensure(ctx, ctx->d_morph, HW * (C + 1)));
            const uint8_t *img = s.bytes.ptr + HW * c;
            uint8_t *fin = ctx->d_morph.ptr + HW * c, *tmp = ctx->d_morph.ptr + HW * C;
            const dim3 mg(grid_for(HW)), mb(kBlock);
            if (ctx->post_open[c]) {
                hipLaunchKernelGGL((k_morph2x2<false>), mg, mb, 0, st, img, tmp, ctx->H, ctx->W);
                hipLaunchKernelGGL((k_morph2x2<true>), mg, mb, 0, st, (const uint8_t *)tmp, fin, ctx->H, ctx->W);
                img = fin;
            }
            if (ctx->post_close[c]) {
                hipLaunchKernelGGL((k_morph2x2<true>), mg, mb, 0, st, img, tmp, ctx->H, ctx->W);
                hipLaunchKernelGGL((k_morph2x2<false>), mg, mb, 0, st, (const uint8_t *)tmp, fin, ctx->H, ctx->W);
            }
            VC_HIP(ctx, hipGetLastError());
            pp.src[c] = fin;
        }
        for (uint32_t c = 0; c < C; ++c) {
            if (!s.frame_dirty[c]) continue;
            pp.fsrc[pp.nframes] = s.fbytes[c].ptr;
            pp.fdst[pp.nframes] = s.frames.ptr + HW * c;
            pp.nframes++;
            s.frame_dirty[c] = 0;
        }
        pp.bits = s.bits.ptr; pp.grid = s.grid.ptr; pp.boxes = s.boxes.ptr;
        pp.C = C; pp.H = ctx->H; pp.W = ctx->W; pp.HW = (uint32_t)HW; pp.mwords = ctx->mwords;
        pp.parity = s.parity;
        pp.dbg = (uint32_t)ctx->dbg;
        // about a thousand packing workgroups at most: every one of them looks at (and may update) its camera's box
        const uint64_t total_words = (uint64_t)ctx->mwords * C;
        pp.iters = (uint32_t)(total_words / (256ull * 1024ull));
        pp.iters = pp.iters < 1 ? 1 : (pp.iters > 16 ? 16 : pp.iters);
        const uint32_t pw = (ctx->mwords + kBlock * pp.iters - 1) / (kBlock * pp.iters), fw = (uint32_t)((HW + 4 * kBlock - 1) / (4 * kBlock));
        VC_KLAUNCH(VC_K_PREP_PACK, k_prep_pack, dim3(C * pw + pp.nframes * fw), dim3(kBlock), 0, st, pp);
        VC_HIP(ctx, hipGetLastError());
        s.bits_valid = true;
        s.grids_valid = false;
        s.counts_zero = true;
    }
    if (want_grids && !s.grids_valid) {
        CarveParams p = *cp;
        p.maskbits = s.bits.ptr;
        const uint64_t n = p.n;
        const uint32_t ns = (uint32_t)(n < kEstimateSamples ? n : kEstimateSamples);
        const uint32_t est_wgs = ctx->reorder ? (ns + kBlock * kEstPerThread - 1) / (kBlock * kEstPerThread) : 0u;   // no counts: cameras in index order
        if (!s.counts_zero) VC_HIP(ctx, hipMemsetAsync(s.boxes.ptr + kCountBase, 0, sizeof(uint32_t) * kMaxCameras * kBoxStride, st));
        s.counts_zero = false;
        // all grids together hold at most 16 blocks per budgeted word; every camera's blocks are rounded up to whole workgroups
        const uint32_t grid_wgs = (16u * s.budget_words + kBlock - 1) / kBlock + C;
        VC_KLAUNCH(VC_K_PREP_GRID, k_prep_grid, dim3(grid_wgs + est_wgs), dim3(kBlock), 0, st, p,
                           s.grid.ptr, s.boxes.ptr, s.parity, (uint32_t)ctx->grid_min_shift, s.budget_words, ns, grid_wgs);
        VC_HIP(ctx, hipGetLastError());
        // large grids (the brick pipeline's 1024-thread workgroups): the brick level gets 4 x 4 times coarser blocks
        s.has_coarse = false;
        if ((size_t)s.budget_words * sizeof(uint32_t) > kWideGridBytes) {
            VC_TRY(ensure(ctx, s.coarse, (size_t)s.budget_words + 8));       // (never larger than the fine grids)
            hipLaunchKernelGGL(k_coarsen_grids, dim3(8, C), dim3(kBlock), 0, st, (const uint32_t *)s.grid.ptr, s.coarse.ptr, C);
            VC_HIP(ctx, hipGetLastError());
            s.has_coarse = true;
        }
        s.grids_valid = true;
    }
    VC_HIP(ctx, hipEventRecord(s.e_prep, st));
    s.prep_pending = true;
    return VC_OK;
}

uint32_t grid_for(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

constexpr int kEmitBatch = 4;              // survivors per lane in flight together in k_emit_words

// start / stop: events that are to carry the launch's own begin and end (its packet's signals, hipExtLaunchKernelGGL), or null
int launch_emit(vc_ctx *ctx, StepBuf &sb, hipStream_t st, hipEvent_t start = nullptr, hipEvent_t stop = nullptr)
{
    const EmitParams &e = sb.emit;
    const dim3 eg((e.ngroups + 3) / 4), block(kBlock);
#define VC_EMIT(kernel, grid) hipExtLaunchKernelGGL((kernel), grid, block, 0, st, start, stop, 0, e)
    if (sb.busy && ctx->emit_lanes) {
        const dim3 bg(256u * (uint32_t)ctx->emit_waves_per_cu / 4u);
        if (e.lut && sb.allseen) VC_EMIT((k_emit_busy<true, true, 8>), bg);
        else if (e.lut) VC_EMIT((k_emit_busy<true, false, 8>), bg);
        else if (sb.allseen) VC_EMIT((k_emit_busy<false, true, 4>), bg);
        else VC_EMIT((k_emit_busy<false, false, 4>), bg);
    }
    else if (ctx->emit_lanes) {
        if (e.lut && sb.allseen) VC_EMIT((k_emit_lanes<true, true, 8>), eg);
        else if (e.lut) VC_EMIT((k_emit_lanes<true, false, 8>), eg);
        else if (sb.allseen) VC_EMIT((k_emit_lanes<false, true, 4>), eg);
        else VC_EMIT((k_emit_lanes<false, false, 4>), eg);
    }
    else if (e.lut && sb.allseen) VC_EMIT((k_emit_words<true, true, kEmitBatch>), eg);
    else if (e.lut) VC_EMIT((k_emit_words<true, false, kEmitBatch>), eg);
    else if (sb.allseen) VC_EMIT((k_emit_words<false, true, kEmitBatch>), eg);
    else VC_EMIT((k_emit_words<false, false, kEmitBatch>), eg);
#undef VC_EMIT
    VC_HIP(ctx, hipGetLastError());
    return VC_OK;
}


// counts -> exclusive offsets (two levels) on the context's stream; the total also lands in *total_host
int scan_counts(vc_ctx *ctx, hipStream_t st, const uint32_t *cnt, uint32_t ngroups, uint32_t *off, uint64_t *bsum, uint64_t *boff,
                uint64_t *total_host)
{
    const uint32_t nscan = (ngroups + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(k_scan_groups, dim3(nscan), dim3(kScanThreads), 0, st, cnt, ngroups, off, bsum, boff, total_host,
                       (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u);
    VC_HIP(ctx, hipGetLastError());
    if (nscan > 1) {
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(kScanThreads), 0, st, bsum, nscan, boff, total_host);
        VC_HIP(ctx, hipGetLastError());
    }
    return VC_OK;
}

int ensure_exchange_scratch(vc_ctx *ctx, uint32_t ngroups)
{
    VC_TRY(ensure(ctx, ctx->d_xcnt, ngroups));
    VC_TRY(ensure(ctx, ctx->d_xoff, ngroups));
    VC_TRY(ensure(ctx, ctx->d_xbsum, kMaxScanBlocks));
    VC_TRY(ensure(ctx, ctx->d_xboff, kMaxScanBlocks + 1));
    if (!ctx->h_xtotal)
        VC_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_xtotal), 4 * sizeof(uint64_t), hipHostMallocDefault));   // [2], [3]: the two gathers in flight
    return VC_OK;
}

// Pixel boxes of the slab's words (tile or y-line order) without a lookup table: the table-free
// hierarchical kernel reads them instead of bounding each word by interval arithmetic.  Geometry only:
// built once per grid / slab / camera set (the projection of every voxel, as long as vc_build_lut).
int ensure_boxes(vc_ctx *ctx, bool tile)
{
    if (tile ? ctx->tbox_valid : ctx->bbox_valid) return VC_OK;
    const uint64_t n = ctx->n_voxels();
    const uint64_t n_pad = (n + kLutPad - 1) / kLutPad * kLutPad;
    DevBuf<uint64_t> &buf = tile ? ctx->d_tbox : ctx->d_bbox;
    VC_TRY(ensure(ctx, buf, (size_t)(n_pad / 64) * ctx->C));
    CarveParams p;
    fill_params(ctx, p);
    if (tile) hipLaunchKernelGGL(k_build_lut<true>, dim3(grid_for(n_pad)), dim3(kBlock), 0, ctx->stream, p, (int32_t *)nullptr, buf.ptr);
    else hipLaunchKernelGGL(k_build_lut<false>, dim3(grid_for(n_pad)), dim3(kBlock), 0, ctx->stream, p, (int32_t *)nullptr, buf.ptr);
    VC_HIP(ctx, hipGetLastError());
    (tile ? ctx->tbox_valid : ctx->bbox_valid) = true;
    if (tile) VC_TRY(build_brick_boxes(ctx));
    return VC_OK;
}

// Enqueues the packing of the current result's non-zero words into ctx->d_ent ({bits, base} pairs) and
// {entries, survivors} into ctx->d_mine.  No host synchronisation; *h_xtotal holds the entry count
// once the stream has drained.
int enqueue_pack(vc_ctx *ctx, StepBuf &cur, hipStream_t st)
{
    const uint64_t n = cur.n;
    VC_TRY(ensure_exchange_scratch(ctx, 1));
    VC_TRY(ensure(ctx, cur.mine, 2));
    if (n == 0) {
        VC_HIP(ctx, hipMemsetAsync(cur.mine.ptr, 0, 2 * sizeof(uint64_t), st));
        *ctx->h_xtotal = 0;
        return VC_OK;
    }
    const uint64_t nwords = (n + 63) / 64;
    const uint64_t n_pad = (n + kLutPad - 1) / kLutPad * kLutPad;
    const uint32_t ngroups = (uint32_t)(n_pad / (64 * kGroupWords));
    const uint32_t nscan = (ngroups + kScanBlock - 1) / kScanBlock;
    VC_TRY(ensure_exchange_scratch(ctx, ngroups));
    VC_TRY(ensure(ctx, cur.ent, (size_t)(2 * nwords)));          // worst case: every word non-zero (n / 4 bytes)
    const dim3 grid((ngroups + 3) / 4), block(kBlock);
    if (cur.nz_valid && cur.busy) {
        // the carve left the counts of non-zero words and the list of groups with survivors: no counting pass, and the packing
        // strides over the list (5 of 6 groups are empty; a launch over all of them is dispatch bound)
        VC_TRY(scan_counts(ctx, st, cur.groupnz.ptr, ngroups, ctx->d_xoff.ptr, ctx->d_xbsum.ptr, ctx->d_xboff.ptr, ctx->h_xtotal));
        hipLaunchKernelGGL(k_pack_busy, dim3(1024), block, 0, st, (const uint64_t *)cur.words.ptr, nwords, (const uint32_t *)cur.busylist.ptr,
                           (const uint32_t *)cur.busyblock.ptr, (const uint32_t *)ctx->d_xoff.ptr, (const uint64_t *)ctx->d_xboff.ptr, nscan, ctx->i0(),
                           (const uint64_t *)(cur.blockoff.ptr + nscan), cur.ent.ptr, cur.mine.ptr);
        VC_HIP(ctx, hipGetLastError());
        return VC_OK;
    }
    hipLaunchKernelGGL(k_count_nz, grid, block, 0, st, cur.words.ptr, nwords, ngroups, cur.groupcnt.ptr,
                       ctx->d_xcnt.ptr);
    VC_HIP(ctx, hipGetLastError());
    VC_TRY(scan_counts(ctx, st, ctx->d_xcnt.ptr, ngroups, ctx->d_xoff.ptr, ctx->d_xbsum.ptr, ctx->d_xboff.ptr, ctx->h_xtotal));
    hipLaunchKernelGGL(k_pack_entries, grid, block, 0, st, cur.words.ptr, nwords, ngroups, cur.groupcnt.ptr, ctx->d_xoff.ptr,
                       ctx->d_xboff.ptr, nscan, ctx->i0(), cur.blockoff.ptr + nscan, cur.ent.ptr, cur.mine.ptr);
    VC_HIP(ctx, hipGetLastError());
    return VC_OK;
}

// {entries, survivors} of every rank into cur.h_counts (valid once the stream has drained).
int enqueue_counts_exchange(vc_ctx *ctx, StepBuf &cur, hipStream_t st)
{
    const int G = ctx->n_ranks;
    VC_TRY(ensure(ctx, cur.counts, (size_t)2 * G));
    if (!cur.h_counts)
        VC_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&cur.h_counts), sizeof(uint64_t) * 2 * VC_MAX_RANKS, hipHostMallocDefault));
    VC_NCCL(ctx, g_rccl.AllGather(cur.mine.ptr, cur.counts.ptr, 2, ncclUint64, ctx->comm, st));
    VC_HIP(ctx, hipMemcpyAsync(cur.h_counts, cur.counts.ptr, sizeof(uint64_t) * 2 * G, hipMemcpyDeviceToHost, st));
    return VC_OK;
}

// The colour camera's table over the whole grid (4 B per voxel of the WHOLE grid, built once per
// camera): what lets a rank colour survivors of words another rank carved.
int ensure_color_table(vc_ctx *ctx, int cam)
{
    if (ctx->lut_color_cam == cam && ctx->d_lut_color.ptr) return VC_OK;
    const uint64_t n = (uint64_t)ctx->nx * ctx->ny * ctx->nz;
    const uint64_t n_pad = (n + kLutPad - 1) / kLutPad * kLutPad;
    VC_TRY(ensure(ctx, ctx->d_lut_color, (size_t)n_pad));
    CarveParams p;
    fill_params(ctx, p);
    p.n = n; p.n_pad = n_pad; p.z0 = 0; p.C = 1;
    p.cam[0] = ctx->cams[cam];
    hipLaunchKernelGGL(k_build_lut<false>, dim3(grid_for(n_pad)), dim3(kBlock), 0, ctx->stream, p, ctx->d_lut_color.ptr,
                       (uint64_t *)nullptr);
    VC_HIP(ctx, hipGetLastError());
    ctx->lut_color_cam = cam;
    return VC_OK;
}

// Observes the completion of a queued all-gather: its timing, and that the expansion produced
// the survivor count the ranks announced.
static int finish_one(vc_ctx *ctx, uint32_t half)
{
    if (!ctx->gpend[half]) return VC_OK;
    ctx->gpend[half] = false;
    hipEvent_t *E = ctx->gx[ctx->gx_idx[half]];
    VC_HIP(ctx, hipEventSynchronize(E[1]));
    VC_HIP(ctx, hipEventElapsedTime(&ctx->tm.gather_ms, E[0], E[1]));
    VC_HIP(ctx, hipEventElapsedTime(&ctx->tm.exchange_ms, E[0], E[2]));
    ctx->tm.gather_ms_sum += ctx->tm.gather_ms;
    ctx->tm.gathers += 1;
    if (ctx->gexpect[half] && ctx->h_xtotal[2 + half] != ctx->gexpect[half])
        return fail(ctx, VC_ERR_RCCL, "gathered words expand to %llu survivors, the ranks reported %llu",
                    (unsigned long long)ctx->h_xtotal[2 + half], (unsigned long long)ctx->gexpect[half]);
    return VC_OK;
}
// every queued compact gather, oldest first
int finish_gather(vc_ctx *ctx)
{
    VC_TRY(finish_one(ctx, ctx->gseq & 1u));
    return finish_one(ctx, (ctx->gseq + 1u) & 1u);
}

// Expands M gathered entries (device, ascending) into the ordered survivor records of the whole grid
// in ctx->d_gathered, coloured the way the current step was (its mode, colour camera and frame set).
// S_hint = expected survivor count (0 = unknown: sized after a host synchronisation).
// st: the stream the expansion runs on (the second stream lets it run beside the next step's carve; its scan
// scratch is its own because the next step's packing may be under way on the first).
int enqueue_expand(vc_ctx *ctx, hipStream_t st, const uint64_t *d_entries, uint64_t M, uint64_t S_hint, uint64_t *h_total = nullptr)
{
    StepBuf &cur = ctx->sb[ctx->cur];
    if (!h_total) h_total = ctx->h_xtotal + 1;                   // (page-locked word the scan leaves the survivor total in)
    *h_total = 0;
    if (M == 0) return VC_OK;
    const uint32_t chunk = M <= (1ull << 24) ? 16u : kGroupWords;   // entries per wave (the scan takes 2^20 groups at most)
    const uint32_t ngroups = (uint32_t)((M + chunk - 1) / chunk);
    VC_TRY(ensure_exchange_scratch(ctx, 1));
    VC_TRY(ensure(ctx, ctx->d_ycnt, ngroups));
    VC_TRY(ensure(ctx, ctx->d_yoff, ngroups));
    VC_TRY(ensure(ctx, ctx->d_ybsum, kMaxScanBlocks));
    VC_TRY(ensure(ctx, ctx->d_yboff, kMaxScanBlocks + 1));
    // (the table-free mode colours from the colour camera's table too, unless fused_color_table is off: the whole-grid table the
    // expansion of other ranks' words needs is the very one)
    const bool from_lut = (cur.mode == VC_MODE_LUT || ctx->fused_color_table) && cur.color_cam >= 0;
    if (from_lut && !(ctx->lut_color_cam == cur.color_cam && ctx->d_lut_color.ptr)) {
        VC_TRY(ensure_color_table(ctx, cur.color_cam));           // built on the first stream, once per camera
        if (st != ctx->stream) VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const dim3 grid((ngroups + 3) / 4), block(kBlock);
    hipLaunchKernelGGL(k_count_entries, chunk == 16u ? dim3((ngroups + 15) / 16) : grid, block, 0, st, d_entries, M, ngroups, ctx->d_ycnt.ptr, chunk);
    VC_HIP(ctx, hipGetLastError());
    VC_TRY(scan_counts(ctx, st, ctx->d_ycnt.ptr, ngroups, ctx->d_yoff.ptr, ctx->d_ybsum.ptr, ctx->d_yboff.ptr, h_total));
    if (S_hint == 0) {
        VC_HIP(ctx, hipStreamSynchronize(st));
        S_hint = *h_total;
    }
    if (S_hint > ctx->d_gathered.cap)            // survivor counts drift from frame to frame: grow with slack
        VC_TRY(ensure(ctx, ctx->d_gathered, (size_t)(S_hint + S_hint / 8 + 1024)));
    EmitParams e;                                // axes, camera, mask bits and frame of the step
    memset(&e, 0, sizeof e);
    e.xs = ctx->d_axes.ptr; e.ys = e.xs + ctx->nx; e.zs = e.ys + ctx->ny;
    e.nx = ctx->nx; e.ny = ctx->ny; e.H = ctx->H; e.W = ctx->W;
    if (cur.color_cam >= 0) {
        const Slot &s = ctx->slots[cur.slot];
        e.has_cam = 1;
        e.cam = ctx->cams[cur.color_cam];
        e.maskbits = s.bits.ptr + (size_t)cur.color_cam * ctx->mwords;
        if (s.frames.ptr && s.have_frame[cur.color_cam]) e.frame = s.frames.ptr + (size_t)cur.color_cam * ctx->H * ctx->W;
    }
    e.entries = d_entries;
    e.groupcnt = ctx->d_ycnt.ptr; e.groupoff = ctx->d_yoff.ptr; e.blockoff = ctx->d_yboff.ptr;
    e.n = M * 64; e.i0 = 0; e.z0 = 0; e.ngroups = ngroups; e.entry_chunk = chunk;
    e.records = ctx->d_gathered.ptr; e.capacity = ctx->d_gathered.cap;
    e.lut = from_lut ? ctx->d_lut_color.ptr : nullptr;
    if (from_lut && cur.allseen) hipLaunchKernelGGL((k_emit_lanes<true, true, 8, true>), grid, block, 0, st, e);
    else if (from_lut) hipLaunchKernelGGL((k_emit_lanes<true, false, 8, true>), grid, block, 0, st, e);
    else if (cur.allseen) hipLaunchKernelGGL((k_emit_lanes<false, true, 4, true>), grid, block, 0, st, e);
    else hipLaunchKernelGGL((k_emit_lanes<false, false, 4, true>), grid, block, 0, st, e);
    VC_HIP(ctx, hipGetLastError());
    return VC_OK;
}

}  // namespace

// ================================================================ C ABI
extern "C" {

int vc_device_count(int *n_out)
{
    if (!n_out) return VC_ERR_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *n_out = 0; return fail(nullptr, VC_ERR_NODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n_out = n;
    return VC_OK;
}

int vc_create(int device, vc_ctx **out)
{
    if (!out) return VC_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, VC_ERR_NODEV, "no HIP device (%s); voxcarve has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "count 0");
    if (device < 0 || device >= n) return fail(nullptr, VC_ERR_ARG, "device %d not in [0,%d)", device, n);
    hipDeviceProp_t prop;
    VC_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, VC_ERR_NODEV, "device %d is %s; this library carries gfx950 code only",
                    device, prop.gcnArchName);
    VC_HIP(nullptr, hipSetDevice(device));
    vc_ctx *ctx = new vc_ctx();
    ctx->device = device;
    memset(&ctx->tm, 0, sizeof ctx->tm);
    hipError_t e1 = make_streams(ctx);
    for (int k = 0; k < 2 && e1 == hipSuccess; ++k) e1 = hipEventCreate(&ctx->ev_h[k]);
    for (int k = 0; k < kDepth && e1 == hipSuccess; ++k) {
        StepBuf &b = ctx->sb[k];
        e1 = hipEventCreate(&b.e0);
        if (e1 == hipSuccess) e1 = hipEventCreate(&b.e_first);
        if (e1 == hipSuccess) e1 = hipEventCreate(&b.e1);
            if (e1 == hipSuccess) e1 = hipEventCreate(&b.e_prep);
        if (e1 == hipSuccess) e1 = hipHostMalloc(reinterpret_cast<void **>(&b.h_total), sizeof(uint64_t), hipHostMallocDefault);
    }
    for (int i = 0; i < 4 && e1 == hipSuccess; ++i) e1 = hipEventCreate(&ctx->ev[i]);
    for (uint32_t r = 0; r < kGatherRing && e1 == hipSuccess; ++r)
        for (int i = 0; i < 3 && e1 == hipSuccess; ++i) e1 = hipEventCreate(&ctx->gx[r][i]);
    if (e1 == hipSuccess) e1 = make_events(ctx);
    if (e1 == hipSuccess) e1 = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_total), sizeof(uint64_t), hipHostMallocDefault);
    const char *fg = getenv("VOXCARVE_FORCE_GENERIC");
    ctx->force_generic = fg && fg[0] == '1';
    if (e1 != hipSuccess) {
        int rc = fail(nullptr, VC_ERR_HIP, "context setup: %s", hipGetErrorString(e1));
        vc_destroy(ctx);                 // releases whatever was created so far
        return rc;
    }
    *out = ctx;
    return VC_OK;
}

int vc_destroy(vc_ctx *ctx)
{
    if (!ctx) return VC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream_up) (void)hipStreamSynchronize(ctx->stream_up);
    if (ctx->stream_x) (void)hipStreamSynchronize(ctx->stream_x);
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
    for (Slot &s : ctx->slots) {
        release_slot(s);
        if (s.e_up) (void)hipEventDestroy(s.e_up);
        if (s.e_prep) (void)hipEventDestroy(s.e_prep);
        if (s.e_p0) (void)hipEventDestroy(s.e_p0);
    }
    for (int k = 0; k < 2; ++k) if (ctx->ev_h[k]) (void)hipEventDestroy(ctx->ev_h[k]);
    for (uint32_t r = 0; r < kGatherRing; ++r)
        for (int i = 0; i < 3; ++i) if (ctx->gx[r][i]) (void)hipEventDestroy(ctx->gx[r][i]);
    for (uint32_t r = 0; r < kStepRing; ++r)
        for (int i = 0; i < 3; ++i) if (ctx->step_ev[r][i]) (void)hipEventDestroy(ctx->step_ev[r][i]);
    release(ctx->d_axes); release(ctx->d_morph); release(ctx->d_lut); release(ctx->d_bbox); release(ctx->d_lut_tile); release(ctx->d_tbox); release(ctx->d_kbox); release(ctx->d_live); release(ctx->d_wbox); release(ctx->d_bm); release(ctx->d_blist); release(ctx->d_wlist);
    release(ctx->d_mcbits); release(ctx->d_mcx); release(ctx->d_mcwbase); release(ctx->d_mcgv); release(ctx->d_mcgt); release(ctx->d_mcgvoff);
    release(ctx->d_mcgtoff); release(ctx->d_mcfaces); release(ctx->d_mcbv); release(ctx->d_mcbvoff); release(ctx->d_mcbt); release(ctx->d_mcbtoff);
    release(ctx->d_mcverts);
    for (StepBuf &b : ctx->sb) {
        release(b.words); release(b.groupcnt); release(b.groupoff); release(b.groupnz); release(b.blocksum); release(b.blockoff); release(b.records);
        release(b.ent); release(b.mine); release(b.counts);
        release(b.busyoff); release(b.busysum); release(b.busyblock); release(b.busylist);
        if (b.h_counts) (void)hipHostFree(b.h_counts);
        if (b.h_total) (void)hipHostFree(b.h_total);
        if (b.e0) (void)hipEventDestroy(b.e0);
        if (b.e_first) (void)hipEventDestroy(b.e_first);
        if (b.e1) (void)hipEventDestroy(b.e1);
        if (b.e_prep) (void)hipEventDestroy(b.e_prep);
        for (int kk = 0; kk < VC_KERNEL_KINDS; ++kk)
            for (int i = 0; i < 2; ++i) if (b.kev[kk][i]) (void)hipEventDestroy(b.kev[kk][i]);
    }
    release(ctx->d_stats); release(ctx->d_fg); release(ctx->d_hsvdiv);
    for (auto &m : ctx->mog) release(m.state);
    release(ctx->d_viewmask); release(ctx->d_scratch); release(ctx->d_counts); release(ctx->d_gathered);
    release(ctx->d_ent_all[0]); release(ctx->d_ent_all[1]); release(ctx->d_xcnt); release(ctx->d_xoff); release(ctx->d_xbsum);
    release(ctx->d_xboff); release(ctx->d_lut_color);
    release(ctx->d_ycnt); release(ctx->d_yoff); release(ctx->d_ybsum); release(ctx->d_yboff);
    if (ctx->h_xtotal) (void)hipHostFree(ctx->h_xtotal);
    if (ctx->h_lists) (void)hipHostFree(ctx->h_lists);
    if (ctx->h_total) (void)hipHostFree(ctx->h_total);
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    for (int i = 0; i < 4; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream_up) (void)hipStreamDestroy(ctx->stream_up);
    if (ctx->stream_x) (void)hipStreamDestroy(ctx->stream_x);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VC_OK;
}

const char *vc_last_error(const vc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int vc_synchronize(vc_ctx *ctx)
{
    if (!ctx) return VC_ERR_ARG;
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream_up));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream_x));
    return finish_gather(ctx);
}

int vc_set_grid(vc_ctx *ctx, uint32_t nx, uint32_t ny, uint32_t nz, const double bounds[6])
{
    if (!ctx || !bounds) return VC_ERR_ARG;
    if (ctx->npending) return fail(ctx, VC_ERR_ARG, "carve steps are in flight: collect them with vc_carve_end first");
    if (nx == 0 || ny == 0 || nz == 0) return fail(ctx, VC_ERR_ARG, "grid dimensions must be >= 1");
    const uint64_t N = (uint64_t)nx * ny * nz;
    if (N > 0xffffffffull || (uint64_t)nx * ny > 0xffffffffull)
        return fail(ctx, VC_ERR_ARG, "grid of %llu voxels exceeds the u32 voxel index", (unsigned long long)N);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    ctx->nx = nx; ctx->ny = ny; ctx->nz = nz; ctx->z0 = 0; ctx->z1 = nz;
    memcpy(ctx->bounds, bounds, sizeof ctx->bounds);
    linspace(bounds[0], bounds[1], nx, ctx->xs);
    linspace(bounds[2], bounds[3], ny, ctx->ys);
    linspace(bounds[4], bounds[5], nz, ctx->zs);
    VC_TRY(ensure(ctx, ctx->d_axes, (size_t)nx + ny + nz));
    VC_HIP(ctx, hipMemcpyAsync(ctx->d_axes.ptr, ctx->xs.data(), sizeof(double) * nx, hipMemcpyHostToDevice, ctx->stream));
    VC_HIP(ctx, hipMemcpyAsync(ctx->d_axes.ptr + nx, ctx->ys.data(), sizeof(double) * ny, hipMemcpyHostToDevice, ctx->stream));
    VC_HIP(ctx, hipMemcpyAsync(ctx->d_axes.ptr + nx + ny, ctx->zs.data(), sizeof(double) * nz, hipMemcpyHostToDevice, ctx->stream));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_grid = true;
    for (Slot &sl : ctx->slots) sl.grids_valid = false;       // the camera order was sampled on the old geometry
    if (ctx->h_lists) ctx->h_lists[0] = ctx->h_lists[1] = ctx->h_lists[2] = 0xffffffffu;
    ctx->lut_valid = false; ctx->upload_mask = 0; ctx->ymajor_valid = false; ctx->tile_valid = false; ctx->bbox_valid = false; ctx->tbox_valid = false; ctx->kbox_valid = false; ctx->carved = false; ctx->gathered = false; ctx->viewmask_valid = false;
    ctx->lut_color_cam = -1; ctx->packed = false;
    return VC_OK;
}

int vc_set_slab(vc_ctx *ctx, uint32_t z0, uint32_t z1)
{
    if (!ctx) return VC_ERR_ARG;
    if (ctx->npending) return fail(ctx, VC_ERR_ARG, "carve steps are in flight: collect them with vc_carve_end first");
    if (!ctx->have_grid) return fail(ctx, VC_ERR_ARG, "vc_set_grid must precede vc_set_slab");
    if (z0 > z1 || z1 > ctx->nz) return fail(ctx, VC_ERR_ARG, "slab [%u,%u) outside [0,%u]", z0, z1, ctx->nz);
    ctx->z0 = z0; ctx->z1 = z1;
    for (Slot &sl : ctx->slots) sl.grids_valid = false;
    ctx->lut_valid = false; ctx->upload_mask = 0; ctx->ymajor_valid = false; ctx->tile_valid = false; ctx->bbox_valid = false; ctx->tbox_valid = false; ctx->kbox_valid = false; ctx->carved = false; ctx->gathered = false; ctx->viewmask_valid = false;
    ctx->packed = false;
    return VC_OK;
}

int vc_get_axes(vc_ctx *ctx, double *xs, double *ys, double *zs)
{
    if (!ctx || !ctx->have_grid) return ctx ? fail(ctx, VC_ERR_ARG, "no grid") : VC_ERR_ARG;
    VC_HIP(ctx, hipSetDevice(ctx->device));
    if (xs) VC_HIP(ctx, hipMemcpy(xs, ctx->d_axes.ptr, sizeof(double) * ctx->nx, hipMemcpyDeviceToHost));
    if (ys) VC_HIP(ctx, hipMemcpy(ys, ctx->d_axes.ptr + ctx->nx, sizeof(double) * ctx->ny, hipMemcpyDeviceToHost));
    if (zs) VC_HIP(ctx, hipMemcpy(zs, ctx->d_axes.ptr + ctx->nx + ctx->ny, sizeof(double) * ctx->nz, hipMemcpyDeviceToHost));
    return VC_OK;
}

int vc_set_cameras(vc_ctx *ctx, uint32_t C, const double *K9, const double *dist5, const double *R9,
                   const double *t3, uint32_t H, uint32_t W)
{
    if (!ctx || !K9 || !dist5 || !R9 || !t3) return VC_ERR_ARG;
    if (ctx->npending) return fail(ctx, VC_ERR_ARG, "carve steps are in flight: collect them with vc_carve_end first");
    if (C == 0 || C > VC_MAX_CAMERAS) return fail(ctx, VC_ERR_ARG, "camera count %u not in [1,%d]", C, VC_MAX_CAMERAS);
    if (H == 0 || W == 0 || H > 32767 || W > 65535) return fail(ctx, VC_ERR_ARG, "mask size %ux%u outside 1..32767 x 1..65535", H, W);
    for (uint32_t c = 0; c < C; ++c) {
        CamDev &d = ctx->cams[c];
        memcpy(d.r, R9 + 9 * c, sizeof d.r);
        memcpy(d.t, t3 + 3 * c, sizeof d.t);
        d.fx = K9[9 * c + 0]; d.cx = K9[9 * c + 2];
        d.fy = K9[9 * c + 4]; d.cy = K9[9 * c + 5];
        d.k1 = dist5[5 * c + 0]; d.k2 = dist5[5 * c + 1];
        d.p1 = dist5[5 * c + 2]; d.p2 = dist5[5 * c + 3];
        d.k3 = dist5[5 * c + 4];
    }
    const bool reshaped = (C != ctx->C || H != ctx->H || W != ctx->W);
    ctx->C = C; ctx->H = H; ctx->W = W;
    ctx->mwords = (uint32_t)(((uint64_t)H * W + 31) / 32);
    ctx->have_cams = true;
    (void)hipSetDevice(ctx->device);
    if (reshaped) {
        VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        VC_HIP(ctx, hipStreamSynchronize(ctx->stream2));
        VC_HIP(ctx, hipStreamSynchronize(ctx->stream_up));
        for (Slot &s : ctx->slots) release_slot(s);
    }
    for (Slot &sl : ctx->slots) sl.grids_valid = false;
    ctx->lut_valid = false; ctx->upload_mask = 0; ctx->ymajor_valid = false; ctx->tile_valid = false; ctx->bbox_valid = false; ctx->tbox_valid = false; ctx->kbox_valid = false; ctx->carved = false; ctx->gathered = false; ctx->viewmask_valid = false;
    ctx->lut_color_cam = -1; ctx->packed = false;
    return VC_OK;
}

// Host -> device copy of `bytes` into dst through the page-locked buffer *h_stage (grown on demand), on the upload
// stream.  The host only ever waits for ITS OWN previous copy out of that staging buffer; the copy itself waits (on
// the device) for the kernels that still read the bytes it replaces.
static int stage_upload(vc_ctx *ctx, Slot &s, uint8_t **h_stage, size_t *h_cap, uint8_t *dst, const uint8_t *src, size_t bytes, bool timed)
{
    if (s.up_pending) { VC_HIP(ctx, hipEventSynchronize(s.e_up)); s.up_pending = false; }
    if (!*h_stage || (h_cap && *h_cap < bytes)) {
        if (*h_stage) VC_HIP(ctx, hipHostFree(*h_stage));
        *h_stage = nullptr;
        VC_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(h_stage), bytes, hipHostMallocDefault));
        if (h_cap) *h_cap = bytes;
    }
    memcpy(*h_stage, src, bytes);
    if (timed) {
        if (ctx->h2d_pending) { (void)hipEventSynchronize(ctx->ev_h[1]); (void)hipEventElapsedTime(&ctx->tm.h2d_ms, ctx->ev_h[0], ctx->ev_h[1]); }
        VC_HIP(ctx, hipEventRecord(ctx->ev_h[0], ctx->stream_up));
    }
    VC_HIP(ctx, hipMemcpyAsync(dst, *h_stage, bytes, hipMemcpyHostToDevice, ctx->stream_up));
    if (timed) { VC_HIP(ctx, hipEventRecord(ctx->ev_h[1], ctx->stream_up)); ctx->h2d_pending = true; }
    VC_HIP(ctx, hipEventRecord(s.e_up, ctx->stream_up));
    s.up_pending = true;
    return VC_OK;
}

int vc_upload_masks(vc_ctx *ctx, uint32_t slot, const uint8_t *masks)
{
    if (!ctx || !masks) return VC_ERR_ARG;
    Slot *s = nullptr;
    VC_TRY(slot_at(ctx, slot, &s));
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t HW = (size_t)ctx->H * ctx->W;
    VC_TRY(ensure(ctx, s->bytes, HW * ctx->C + 64));
    VC_TRY(stage_upload(ctx, *s, &s->h_bytes, &s->h_bytes_cap, s->bytes.ptr, masks, HW * ctx->C, true));
    s->have_masks = true;
    s->bits_valid = false;           // the next carve on this slot re-derives bits, grids and camera order on the device
    s->grids_valid = false;
    return VC_OK;
}

int vc_touch_masks(vc_ctx *ctx, uint32_t slot)
{
    if (!ctx) return VC_ERR_ARG;
    if (slot >= ctx->slots.size() || !ctx->slots[slot].have_masks) return fail(ctx, VC_ERR_ARG, "no masks uploaded in slot %u", slot);
    Slot &s = ctx->slots[slot];
    s.bits_valid = false;
    s.grids_valid = false;
    for (uint32_t c = 0; c < ctx->C; ++c) if (s.have_frame[c]) s.frame_dirty[c] = 1;
    return VC_OK;
}

int vc_set_mask_postfilter(vc_ctx *ctx, const uint8_t *open2x2, const uint8_t *close2x2)
{
    if (!ctx) return VC_ERR_ARG;
    if (!ctx->have_cams) return fail(ctx, VC_ERR_ARG, "vc_set_cameras must precede vc_set_mask_postfilter");
    for (uint32_t c = 0; c < VC_MAX_CAMERAS; ++c) {
        ctx->post_open[c] = (open2x2 && c < ctx->C) ? (open2x2[c] != 0) : 0;
        ctx->post_close[c] = (close2x2 && c < ctx->C) ? (close2x2[c] != 0) : 0;
    }
    return VC_OK;
}

int vc_fetch_mask(vc_ctx *ctx, uint32_t slot, uint32_t cam, uint8_t *out)
{
    if (!ctx || !out) return VC_ERR_ARG;
    if (slot >= ctx->slots.size() || !ctx->slots[slot].have_masks) return fail(ctx, VC_ERR_ARG, "no masks uploaded in slot %u", slot);
    if (cam >= ctx->C) return fail(ctx, VC_ERR_ARG, "camera %u not in [0,%u)", cam, ctx->C);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_TRY(ensure_prepared(ctx, ctx->slots[slot], false, nullptr));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream_up));
    std::vector<uint32_t> bits(ctx->mwords);
    VC_HIP(ctx, hipMemcpy(bits.data(), ctx->slots[slot].bits.ptr + (size_t)cam * ctx->mwords, sizeof(uint32_t) * ctx->mwords,
                          hipMemcpyDeviceToHost));
    const size_t HW = (size_t)ctx->H * ctx->W;
    for (size_t p = 0; p < HW; ++p) out[p] = ((bits[p >> 5] >> (p & 31)) & 1u) ? 255 : 0;
    return VC_OK;
}

int vc_upload_frame(vc_ctx *ctx, uint32_t slot, uint32_t cam, const uint8_t *bgr)
{
    if (!ctx || !bgr) return VC_ERR_ARG;
    Slot *s = nullptr;
    VC_TRY(slot_at(ctx, slot, &s));
    if (cam >= ctx->C) return fail(ctx, VC_ERR_ARG, "camera %u not in [0,%u)", cam, ctx->C);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)ctx->H * ctx->W;
    VC_TRY(ensure(ctx, s->frames, npix * ctx->C));
    VC_TRY(ensure(ctx, s->fbytes[cam], npix * 3 + 64));
    VC_TRY(stage_upload(ctx, *s, &s->h_fbytes[cam], nullptr, s->fbytes[cam].ptr, bgr, npix * 3, false));
    s->have_frame[cam] = 1;
    s->frame_dirty[cam] = 1;
    s->bits_valid = false;           // the image expansion rides in the same launch as the bit-packing
    return VC_OK;
}

// The y-major table [C][n_pad] (+ the y-line words' boxes): what the streaming and the one-thread-per-voxel kernels, the
// y-line hierarchical kernel and vc_fetch_lut read.  The default kernels read the tile-ordered table only, so this one is
// built when something first asks for it (19 ms at 1024^3 x 4, 17 GB).
static int ensure_ymajor(vc_ctx *ctx)
{
    if (ctx->ymajor_valid) return VC_OK;
    const uint64_t n = ctx->n_voxels();
    const uint64_t n_pad = (n + kLutPad - 1) / kLutPad * kLutPad;
    VC_TRY(ensure(ctx, ctx->d_lut, (size_t)n_pad * ctx->C));
    VC_TRY(ensure(ctx, ctx->d_bbox, (size_t)(n_pad / 64) * ctx->C));
    if (n) {
        CarveParams p;
        fill_params(ctx, p);
        if (ctx->lut_foreign && ctx->tile_valid) {               // a table that was handed in: permuted back, never re-projected
            hipLaunchKernelGGL(k_untile_lut, dim3(grid_for(n_pad)), dim3(kBlock), 0, ctx->stream, p, (const int32_t *)ctx->d_lut_tile.ptr, ctx->d_lut.ptr);
            hipLaunchKernelGGL(k_adopt_lut<false>, dim3(grid_for(n_pad)), dim3(kBlock), 0, ctx->stream, p, (const int32_t *)ctx->d_lut.ptr,
                               ctx->d_lut.ptr, ctx->d_bbox.ptr);
        }
        else hipLaunchKernelGGL(k_build_lut<false>, dim3(grid_for(n_pad)), dim3(kBlock), 0, ctx->stream, p, ctx->d_lut.ptr, ctx->d_bbox.ptr);
        VC_HIP(ctx, hipGetLastError());
    }
    ctx->bbox_valid = true;
    ctx->ymajor_valid = true;
    return VC_OK;
}

int vc_build_lut(vc_ctx *ctx)
{
    if (!ctx) return VC_ERR_ARG;
    if (ctx->npending) return fail(ctx, VC_ERR_ARG, "carve steps are in flight: collect them with vc_carve_end first");
    if (!ctx->have_grid || !ctx->have_cams) return fail(ctx, VC_ERR_ARG, "grid and cameras must be set before vc_build_lut");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = ctx->n_voxels();
    const uint64_t n_pad = (n + kLutPad - 1) / kLutPad * kLutPad;
    VC_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    ctx->tile_valid = false;
    ctx->lut_foreign = false; ctx->ymajor_valid = false; ctx->upload_mask = 0;
    if (ctx->lut_tile && ctx->nx % 4 == 0 && ctx->ny % 64 == 0) {
        // ONE table, in tile order (words of 4 x-rows x 16 y), projected straight into that order; the colour look-up of
        // the record expansion reads it too (closed-form index).  No y-major copy unless something asks for one.
        VC_TRY(ensure(ctx, ctx->d_lut_tile, (size_t)n_pad * ctx->C));
        VC_TRY(ensure(ctx, ctx->d_tbox, (size_t)(n_pad / 64) * ctx->C));
        if (n) {
            CarveParams p;
            fill_params(ctx, p);
            hipLaunchKernelGGL(k_build_lut<true>, dim3(grid_for(n_pad)), dim3(kBlock), 0, ctx->stream, p, ctx->d_lut_tile.ptr, ctx->d_tbox.ptr);
            VC_HIP(ctx, hipGetLastError());
            ctx->tile_valid = true;
            ctx->tbox_valid = true;
            VC_TRY(build_brick_boxes(ctx));
        }
    } else {
        VC_TRY(ensure_ymajor(ctx));
    }
    VC_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    VC_HIP(ctx, hipEventElapsedTime(&ctx->tm.lut_ms, ctx->ev[0], ctx->ev[1]));
    ctx->lut_valid = true;
    ctx->lut_foreign = false;
    return VC_OK;
}

// One camera's table from the host (the counterpart of vc_fetch_lut; reference: the pickled lookup table that
// assignment.py:12-15 loads).  When every camera has been handed in, the tables are adopted: permuted into tile order
// where the grid allows (the y-major copy is released again) and the word / brick boxes are reduced from them.
int vc_upload_lut(vc_ctx *ctx, uint32_t cam, const int32_t *lut)
{
    if (!ctx || !lut) return VC_ERR_ARG;
    if (ctx->npending) return fail(ctx, VC_ERR_ARG, "carve steps are in flight: collect them with vc_carve_end first");
    if (!ctx->have_grid || !ctx->have_cams) return fail(ctx, VC_ERR_ARG, "grid and cameras must be set before vc_upload_lut");
    if (cam >= ctx->C) return fail(ctx, VC_ERR_ARG, "camera %u not in [0,%u)", cam, ctx->C);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = ctx->n_voxels();
    const uint64_t n_pad = (n + kLutPad - 1) / kLutPad * kLutPad;
    if (ctx->upload_mask == 0) {                                 // first camera of a new table: whatever was there is void
        ctx->lut_valid = false; ctx->ymajor_valid = false; ctx->tile_valid = false; ctx->bbox_valid = false; ctx->tbox_valid = false; ctx->kbox_valid = false;
        VC_TRY(ensure(ctx, ctx->d_lut, (size_t)n_pad * ctx->C));
        VC_HIP(ctx, hipMemsetAsync(ctx->d_lut.ptr, 0xff, (size_t)n_pad * ctx->C * sizeof(int32_t), ctx->stream));   // padding = -1
    }
    if (n) VC_HIP(ctx, hipMemcpyAsync(ctx->d_lut.ptr + (size_t)cam * n_pad, lut, n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->upload_mask |= 1u << cam;
    if (ctx->upload_mask != (ctx->C >= 32 ? 0xffffffffu : (1u << ctx->C) - 1u)) return VC_OK;
    ctx->upload_mask = 0;
    CarveParams p;
    fill_params(ctx, p);
    if (ctx->lut_tile && ctx->nx % 4 == 0 && ctx->ny % 64 == 0) {
        VC_TRY(ensure(ctx, ctx->d_lut_tile, (size_t)n_pad * ctx->C));
        VC_TRY(ensure(ctx, ctx->d_tbox, (size_t)(n_pad / 64) * ctx->C));
        if (n) {
            hipLaunchKernelGGL(k_adopt_lut<true>, dim3(grid_for(n_pad)), dim3(kBlock), 0, ctx->stream, p, (const int32_t *)ctx->d_lut.ptr,
                               ctx->d_lut_tile.ptr, ctx->d_tbox.ptr);
            VC_HIP(ctx, hipGetLastError());
            ctx->tile_valid = true;
            ctx->tbox_valid = true;
            VC_TRY(build_brick_boxes(ctx));
        }
        VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        release(ctx->d_lut);                                     // one table, as after vc_build_lut
    } else {
        VC_TRY(ensure(ctx, ctx->d_bbox, (size_t)(n_pad / 64) * ctx->C));
        if (n) {
            // (y-major path: the table stays where it is, so entries outside [-1, H*W) are rewritten to -1 IN PLACE)
            hipLaunchKernelGGL(k_adopt_lut<false>, dim3(grid_for(n_pad)), dim3(kBlock), 0, ctx->stream, p, (const int32_t *)ctx->d_lut.ptr,
                               ctx->d_lut.ptr, ctx->d_bbox.ptr);
            VC_HIP(ctx, hipGetLastError());
        }
        VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->bbox_valid = true;
        ctx->ymajor_valid = true;
    }
    ctx->lut_valid = true;
    ctx->lut_foreign = true;
    return VC_OK;
}

int vc_fetch_lut(vc_ctx *ctx, uint32_t cam, int32_t *out)
{
    if (!ctx || !out) return VC_ERR_ARG;
    if (!ctx->lut_valid) return fail(ctx, VC_ERR_ARG, "no lookup table: call vc_build_lut");
    if (cam >= ctx->C) return fail(ctx, VC_ERR_ARG, "camera %u not in [0,%u)", cam, ctx->C);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_TRY(ensure_ymajor(ctx));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n = ctx->n_voxels();
    const uint64_t n_pad = (n + kLutPad - 1) / kLutPad * kLutPad;
    if (n) VC_HIP(ctx, hipMemcpy(out, ctx->d_lut.ptr + (size_t)cam * n_pad, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return VC_OK;
}

int vc_project(vc_ctx *ctx, uint32_t cam, const double *xyz, uint64_t n, double *uv)
{
    if (!ctx || !xyz || !uv) return VC_ERR_ARG;
    if (!ctx->have_cams || cam >= ctx->C) return fail(ctx, VC_ERR_ARG, "camera %u not set", cam);
    if (n == 0) return VC_OK;
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_TRY(ensure(ctx, ctx->d_scratch, (size_t)n * 5));
    double *d_xyz = ctx->d_scratch.ptr, *d_uv = d_xyz + 3 * n;
    VC_HIP(ctx, hipMemcpyAsync(d_xyz, xyz, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_project, dim3(grid_for(n)), dim3(kBlock), 0, ctx->stream, ctx->cams[cam], d_xyz, n, d_uv);
    VC_HIP(ctx, hipGetLastError());
    VC_HIP(ctx, hipMemcpyAsync(uv, d_uv, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VC_OK;
}

int vc_carve_begin(vc_ctx *ctx, uint32_t slot, uint32_t min_views, int color_cam, int mode, uint32_t flags)
{
    if (!ctx) return VC_ERR_ARG;
    if (ctx->npending >= kDepth) return fail(ctx, VC_ERR_ARG, "%d carve steps are already in flight: call vc_carve_end", kDepth);
    if (!ctx->have_grid || !ctx->have_cams) return fail(ctx, VC_ERR_ARG, "grid and cameras must be set before vc_carve");
    if (slot >= ctx->slots.size() || !ctx->slots[slot].have_masks)
        return fail(ctx, VC_ERR_ARG, "no masks uploaded in slot %u", slot);
    if (mode != VC_MODE_FUSED && mode != VC_MODE_LUT) return fail(ctx, VC_ERR_ARG, "unknown mode %d", mode);
    if (mode == VC_MODE_LUT && !ctx->lut_valid) return fail(ctx, VC_ERR_ARG, "VC_MODE_LUT needs vc_build_lut first");
    if (color_cam >= (int)ctx->C) return fail(ctx, VC_ERR_ARG, "colour camera %d not in [0,%u)", color_cam, ctx->C);
    if (min_views < 1) min_views = 1;            // a voxel no camera sees never enters voxels_visible
    VC_HIP(ctx, hipSetDevice(ctx->device));
    Slot &s = ctx->slots[slot];
    const uint64_t n = ctx->n_voxels();
    const bool want_vm = (flags & VC_FLAG_VIEWMASK) != 0;
    ctx->gathered = false;
    ctx->tm.voxels = n;
    if (ctx->head == ctx->cur) {
        // this step is queued into the buffers that hold the result the vc_fetch_* functions read: it is gone from here on
        ctx->carved = false; ctx->viewmask_valid = false; ctx->packed = false;
    }
    StepBuf &sb = ctx->sb[ctx->head];
    sb.e_scan = ctx->step_ev[ctx->step_next][0];
    sb.e2 = ctx->step_ev[ctx->step_next][1];
    sb.e_emit0 = ctx->step_ev[ctx->step_next][2];
    ctx->step_next = (ctx->step_next + 1) % kStepRing;
    sb.n = n; sb.survivors = 0; sb.want_vm = want_vm; sb.has_first = false;
    sb.allseen = min_views == ctx->C;
    sb.no_records = (flags & VC_FLAG_NO_RECORDS) != 0;
    sb.sparse_words = false;
    sb.mode = mode; sb.color_cam = color_cam; sb.slot = slot;
    // a rank of a communicator packs and exchanges the counts right behind the carve, so that
    // vc_allgather finds them on the host and only has the payload and the expansion left
    sb.counts_exchanged = false;
    const bool auto_exchange = sb.no_records && ctx->comm && ctx->gather_compact;
    // packing + collectives of a step run on the exchange stream, beside the next step's carve
    hipStream_t sx = ctx->overlap ? ctx->stream_x : ctx->stream;
    if (n == 0) {
        if (auto_exchange) {                     // an empty slab still takes part in the collective
            VC_TRY(enqueue_pack(ctx, sb, sx));
            VC_TRY(enqueue_counts_exchange(ctx, sb, sx));
            VC_HIP(ctx, hipEventRecord(sb.e2, sx));
            sb.counts_exchanged = true;
        }
        sb.pending = true; sb.used = false; ctx->head = (ctx->head + 1) % kDepth; ctx->npending++;
        return VC_OK;
    }

    const uint64_t nwords = (n + 63) / 64;
    const uint64_t n_pad = (n + kLutPad - 1) / kLutPad * kLutPad;
    const uint32_t ngroups = (uint32_t)(n_pad / (64 * kGroupWords));
    const uint32_t nscan = (ngroups + kScanBlock - 1) / kScanBlock;
    VC_TRY(ensure(ctx, sb.words, n_pad / 64));
    VC_TRY(ensure(ctx, sb.groupcnt, ngroups));
    VC_TRY(ensure(ctx, sb.groupoff, ngroups));
    VC_TRY(ensure(ctx, sb.blocksum, kMaxScanBlocks));
    VC_TRY(ensure(ctx, sb.blockoff, kMaxScanBlocks + 1));
    if (want_vm) VC_TRY(ensure(ctx, ctx->d_viewmask, n));
    if (!sb.records.ptr && !sb.no_records) VC_TRY(ensure(ctx, sb.records, (size_t)(n / 16 + 1024)));

    CarveParams p;
    fill_params(ctx, p);
    sb.kmask = 0;
    ctx->kev_sb = nullptr;
    if (ctx->timing_detail || ctx->kernel_events) ctx->kev_sb = &sb;   // every launch of this step carries begin / end events
    if (ctx->timing_detail) {                                     // ... and the kernels count their work
        if (!ctx->d_stats.ptr) {
            VC_TRY(ensure(ctx, ctx->d_stats, (size_t)VC_WORK_KINDS * kShards * kStatStride));
            VC_HIP(ctx, hipMemset(ctx->d_stats.ptr, 0, ctx->d_stats.cap * sizeof(unsigned long long)));
        }
        p.stats = ctx->d_stats.ptr;
    }
    p.words = sb.words.ptr;
    p.groupcnt = sb.groupcnt.ptr;
    p.viewmask = ctx->d_viewmask.ptr;
    p.min_views = min_views;

    // The chunked kernels cover the reference's case (seen by ALL cameras); the one-thread-per-voxel kernels cover
    // thresholds below C, the camera bitmask, and thresholds above C (no voxel can be seen by more cameras than
    // there are: the reference's sum(views.values()) >= views_threshold is never true, the result is empty).
    bool fast = !ctx->force_generic && !want_vm && min_views == ctx->C;
    // k_lut_first keeps one camera's mask bits in LDS; larger masks take the generic kernel.
    if (mode == VC_MODE_LUT && !ctx->lut_hier && (size_t)ctx->mwords * sizeof(uint32_t) > kMaxFirstLds) fast = false;
    // per-frame preparation, on the device, in front of the carve (nothing to do when the slot has been used before)
    sb.prepped = !s.bits_valid || (fast && !s.grids_valid);
    sb.carve_timed = ctx->timing_detail || ctx->sync_call;
    sb.prep_timed = sb.prepped && sb.carve_timed;
    // (the brick pipeline's grid-staging kernels take up to 148 KB of LDS, the others 64: the preparation sizes the grids for it)
    const bool fused_tiles = mode == VC_MODE_FUSED && fast && ctx->ny % 64 == 0 && ctx->fused_hier && ctx->fused_tile && ctx->nx % 4 == 0 && ctx->fused_boxes;
    if (fused_tiles) VC_TRY(ensure_boxes(ctx, true));
    const bool bricks = fast && brick_shape(ctx, p) && (fused_tiles || (mode == VC_MODE_LUT && ctx->lut_hier && ctx->lut_tile && ctx->tile_valid));
    VC_TRY(ensure_prepared(ctx, s, fast, &p, sb.prep_timed, bricks));
    sb.nz_valid = false;
    if (auto_exchange && bricks && p.tile_whole) {               // k_assemble counts the groups' non-zero words for the packing
        VC_TRY(ensure(ctx, sb.groupnz, ngroups));
        p.groupnz = sb.groupnz.ptr;
        sb.nz_valid = true;
    }
    sb.slot_gen = s.gen;
    if (s.prep_pending) { VC_HIP(ctx, hipStreamWaitEvent(ctx->stream, s.e_prep, 0)); s.prep_pending = false; }
    p.maskbits = s.bits.ptr;
    p.blockgrid = s.grid.ptr;
    p.coarsegrid = s.has_coarse ? s.coarse.ptr : nullptr;
    p.counts = s.boxes.ptr + kCountBase;
    // which table the step reads: the tile-ordered one (hierarchical kernels on tile words), else the y-major one
    const bool lut_tiled = mode == VC_MODE_LUT && fast && ctx->lut_hier && ctx->lut_tile && ctx->tile_valid;
    if (mode == VC_MODE_LUT && !lut_tiled) VC_TRY(ensure_ymajor(ctx));
    p.lut = ctx->d_lut.ptr;
    p.bbox = ctx->d_bbox.ptr;
    const size_t grid_lds = ((size_t)s.budget_words + 8) * sizeof(uint32_t);

    // VC_MODE_FUSED colours the survivors from the colour camera's table over the whole grid: projected once, HERE -- in front of
    // the carve kernels on their stream, so that {scan done}, which the expansion waits for on its own stream and which rides
    // on the k_finish_scan launch, is behind it (queued where the expansion's parameters are set up it would follow that
    // launch, and the first expansion would read a table still being written)
    if (mode == VC_MODE_FUSED && ctx->fused_color_table && color_cam >= 0 && !sb.no_records) VC_TRY(ensure_color_table(ctx, color_cam));
    if (sb.carve_timed) VC_HIP(ctx, hipEventRecord(sb.e0, ctx->stream));
    const dim3 block(kBlock);
    if (fast) {
        const uint64_t nchunks = (n + 64 * kSub - 1) / (64 * kSub);
        const uint64_t want = (nchunks + 3) / 4;
        const uint64_t gmax = 256ull * (uint64_t)ctx->fused_blocks_per_cu;
        const dim3 grid((uint32_t)(want < gmax ? want : gmax));
        if (mode == VC_MODE_LUT && ctx->lut_hier) {
            // one launch: word-level rejection by pixel box x foreground-block grid, exact test for the rest
            const size_t lds = grid_lds;
            const uint64_t groups = p.n_pad / 4096;
            const uint64_t rwant = (groups + 3) / 4;
            const uint64_t rmax = 256ull * (uint64_t)ctx->hier_blocks_per_cu;
            const dim3 rgrid((uint32_t)(rwant < rmax ? rwant : rmax));
            sb.sparse_words = true;
            if (ctx->lut_tile && ctx->tile_valid) {
                if (!p.tile_whole) {             // waves and groups do not coincide: counts by atomics, every word stored
                    VC_HIP(ctx, hipMemsetAsync(sb.groupcnt.ptr, 0, sizeof(uint32_t) * ngroups, ctx->stream));
                    sb.sparse_words = false;
                }
                if (brick_shape(ctx, p)) {
                    sb.sparse_words = true;              // (k_cull_bricks zeroes every group's count itself)
                    VC_TRY(launch_bricks<true>(ctx, p, lds, ngroups));
                } else {
                    if (ctx->cull && ctx->kbox_valid) {
                        p.live = ctx->d_live.ptr;
                        const uint32_t cw = p.nbrick_pad / 256;
                        VC_KLAUNCH(VC_K_CULL, k_cull, dim3(cw < 1 ? 1 : (cw > 1024 ? 1024 : cw)), block, lds, ctx->stream, p);
                    }
                    VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_lut_refine<8, true, true, true>), rgrid, block, lds, ctx->stream, p);
                }
            }
            else if (ctx->refine_pair) VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_lut_refine<8, true, true>), rgrid, block, lds, ctx->stream, p);
            else if (ctx->refine_b == 8) VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_lut_refine<8, true, false>), rgrid, block, lds, ctx->stream, p);
            else VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_lut_refine<16, true, false>), rgrid, block, lds, ctx->stream, p);
        } else if (mode == VC_MODE_LUT) {
            const size_t lds = (size_t)ctx->mwords * sizeof(uint32_t);
            const int kv = ctx->first_kv;
            const uint64_t chunks = p.n_pad / (256 * kv);
            const uint64_t fwant = (chunks + 7) / 8;
            uint32_t per_cu = (uint32_t)(kLdsBytes / (lds ? lds : 1));
            if (per_cu > (uint32_t)ctx->first_blocks_per_cu) per_cu = (uint32_t)ctx->first_blocks_per_cu;
            if (per_cu < 1) per_cu = 1;
            const uint32_t fmax = 256u * per_cu;
            const dim3 fgrid((uint32_t)(fwant < fmax ? fwant : fmax)), fblock(kFirstBlock);
            if (kv == 1) VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_lut_first<1>), fgrid, fblock, lds, ctx->stream, p);
            else if (kv == 4) VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_lut_first<4>), fgrid, fblock, lds, ctx->stream, p);
            else VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_lut_first<2>), fgrid, fblock, lds, ctx->stream, p);
            VC_HIP(ctx, hipGetLastError());
            if (sb.carve_timed) { VC_HIP(ctx, hipEventRecord(sb.e_first, ctx->stream)); sb.has_first = true; }
            const uint64_t groups = p.n_pad / 4096;
            const uint64_t rwant = (groups + 3) / 4;
            const uint64_t rmax = 256ull * (uint64_t)ctx->refine_blocks_per_cu;
            const dim3 rgrid((uint32_t)(rwant < rmax ? rwant : rmax));
            if (ctx->refine_b == 8) VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_lut_refine<8, false, false>), rgrid, block, 0, ctx->stream, p);
            else VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_lut_refine<16, false, false>), rgrid, block, 0, ctx->stream, p);
        }
        else if (ctx->ny % 64 == 0 && ctx->fused_hier) {
            const size_t lds = grid_lds;
            const uint64_t groups = p.n_pad / 4096;
            const uint64_t rwant = (groups + 3) / 4;
            const uint64_t rmax = 256ull * (uint64_t)ctx->hier_blocks_per_cu;
            const dim3 rgrid((uint32_t)(rwant < rmax ? rwant : rmax));
            sb.sparse_words = true;
            if (ctx->fused_tile && ctx->nx % 4 == 0) {            // ny % 64 == 0 here: words of 4 x-rows x 16 y
                if (!p.tile_whole) {
                    VC_HIP(ctx, hipMemsetAsync(sb.groupcnt.ptr, 0, sizeof(uint32_t) * ngroups, ctx->stream));
                    sb.sparse_words = false;
                }
                if (ctx->fused_boxes) {
                    VC_TRY(ensure_boxes(ctx, true));
                    p.tbox = ctx->d_tbox.ptr;
                    p.kbox = ctx->d_kbox.ptr;
                    if (brick_shape(ctx, p)) {
                        sb.sparse_words = true;
                        VC_TRY(launch_bricks<false>(ctx, p, lds, ngroups));
                    } else {
                        if (ctx->cull && ctx->kbox_valid) {
                            p.live = ctx->d_live.ptr;
                            const uint32_t cw = p.nbrick_pad / 256;
                            VC_KLAUNCH(VC_K_CULL, k_cull, dim3(cw < 1 ? 1 : (cw > 1024 ? 1024 : cw)), block, lds, ctx->stream, p);
                        }
                        VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_fused_hier<true, 2>), rgrid, block, lds, ctx->stream, p);
                    }
                }
                else if (ctx->fused_f32box) VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_fused_hier<true, 1>), rgrid, block, lds, ctx->stream, p);
                else VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_fused_hier<true, 0>), rgrid, block, lds, ctx->stream, p);
            }
            else if (ctx->fused_boxes) {
                VC_TRY(ensure_boxes(ctx, false));
                p.bbox = ctx->d_bbox.ptr;
                VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_fused_hier<false, 2>), rgrid, block, lds, ctx->stream, p);
            }
            else if (ctx->fused_f32box) VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_fused_hier<false, 1>), rgrid, block, lds, ctx->stream, p);
            else VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_fused_hier<false, 0>), rgrid, block, lds, ctx->stream, p);
        }
        else if (ctx->ny % 64 == 0) VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_fused<kSub, true>), grid, block, 0, ctx->stream, p);
        else VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_fused<kSub, false>), grid, block, 0, ctx->stream, p);
    } else {
        const dim3 grid(grid_for(n));
        if (mode == VC_MODE_LUT) {
            if (want_vm) VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_generic<true, true>), grid, block, 0, ctx->stream, p);
            else VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_generic<true, false>), grid, block, 0, ctx->stream, p);
        } else {
            if (want_vm) VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_generic<false, true>), grid, block, 0, ctx->stream, p);
            else VC_KLAUNCH(VC_K_CARVE_ONE_LAUNCH, (k_carve_generic<false, false>), grid, block, 0, ctx->stream, p);
        }
    }
    VC_HIP(ctx, hipGetLastError());
    if (sb.carve_timed) VC_HIP(ctx, hipEventRecord(sb.e1, ctx->stream));

    // ---- compaction: group counts -> two-level scan -> record expansion
    // the carve kernels are VALU-issue bound, the expansion is memory bound: on its own stream the expansion of this
    // step runs beside the carve of the next one (two steps in flight) instead of after it
    // (the small scan kernels stay on the first stream, right behind the carve: on the second they would queue for
    // compute units against the next carve's thousands of workgroups and delay the expansion they feed)
    hipStream_t s3 = (ctx->overlap && !ctx->comm) ? ctx->stream2 : ctx->stream;
    hipStream_t s2 = ctx->stream;
    // kernels that do not know their group totals (fused, generic) get them counted
    const bool counted = fast && (mode == VC_MODE_LUT || (ctx->ny % 64 == 0 && ctx->fused_hier));
    if (!counted) {
        VC_KLAUNCH(VC_K_COUNT_GROUPS, k_count_groups, dim3((ngroups + 3) / 4), block, 0, s2, sb.words.ptr, nwords, ngroups,
                           sb.groupcnt.ptr);
        VC_HIP(ctx, hipGetLastError());
    }
    // the expansion of a large grid iterates over the list of groups that have survivors, made by the same scan
    // (a rank of a communicator has no expansion of its own, but its packing strides over the same list)
    sb.busy = ctx->emit_lanes && (!sb.no_records || (auto_exchange && sb.nz_valid)) &&
              (ctx->emit_busy == 2 || (ctx->emit_busy == 1 && ngroups >= kBusyListMinGroups));
    if (sb.busy) {
        VC_TRY(ensure(ctx, sb.busyoff, ngroups));
        VC_TRY(ensure(ctx, sb.busylist, ngroups));
        VC_TRY(ensure(ctx, sb.busysum, kMaxScanBlocks));
        VC_TRY(ensure(ctx, sb.busyblock, 1));                    // the count of busy groups
    }
    VC_KLAUNCH(VC_K_SCAN_GROUPS, k_scan_groups, dim3(nscan), dim3(kScanThreads), 0, s2, (const uint32_t *)sb.groupcnt.ptr, ngroups, sb.groupoff.ptr,
               sb.blocksum.ptr, sb.blockoff.ptr, sb.h_total, sb.busy ? sb.busyoff.ptr : (uint32_t *)nullptr, sb.busysum.ptr,
               sb.busyblock.ptr, (uint32_t)ctx->dbg);
    VC_HIP(ctx, hipGetLastError());
    // the two events a pipelined step hands from stream to stream ride on the launches in front of them where those are the
    // list-driven ones (large grids): {scan done} on k_finish_scan, {step done} on the expansion
    const bool xstream = auto_exchange && sx != ctx->stream;     // the packing waits for the scan across streams
    const bool want_scan_ev = (!sb.no_records && (s3 != ctx->stream || sb.carve_timed)) || xstream;
    const bool ride = ctx->launch_events && !sb.no_records;
    sb.emit_ridden = ride;
    bool scan_ridden = false;
    if (sb.busy) {
        // level 2 of both scans + the list in one launch (k_scan_groups has left the count in busyblock[0] when nscan == 1)
        scan_ridden = ctx->launch_events && want_scan_ev;
        hipEvent_t fs0 = nullptr, fs1 = nullptr;
        kev_pick(ctx, VC_K_FINISH_SCAN, fs0, fs1);
        if (scan_ridden) { fs1 = sb.e_scan; if (fs0) sb.kused[VC_K_FINISH_SCAN][1] = fs1; }
        hipExtLaunchKernelGGL(k_finish_scan, dim3(grid_for(ngroups)), block, 0, s2, fs0, fs1, 0,
                              (const uint64_t *)sb.blocksum.ptr, nscan, sb.blockoff.ptr, sb.h_total, (const uint32_t *)sb.busysum.ptr, sb.busyblock.ptr,
                              (const uint32_t *)sb.groupcnt.ptr, ngroups, (const uint32_t *)sb.busyoff.ptr, sb.busylist.ptr, (uint32_t)ctx->dbg);
        VC_HIP(ctx, hipGetLastError());
    } else if (nscan > 1) {
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(kScanThreads), 0, s2, sb.blocksum.ptr, nscan, sb.blockoff.ptr, sb.h_total);
        VC_HIP(ctx, hipGetLastError());
    }

    EmitParams &e = sb.emit;
    memset(&e, 0, sizeof e);
    e.xs = p.xs; e.ys = p.ys; e.zs = p.zs;
    e.words = sb.words.ptr; e.groupcnt = sb.groupcnt.ptr; e.groupoff = sb.groupoff.ptr; e.blockoff = sb.blockoff.ptr;
    e.n = n; e.i0 = ctx->i0(); e.ngroups = ngroups;
    e.nx = ctx->nx; e.ny = ctx->ny; e.z0 = ctx->z0; e.H = ctx->H; e.W = ctx->W;
    if (color_cam >= 0) {
        e.has_cam = 1;
        e.cam = ctx->cams[color_cam];
        e.maskbits = s.bits.ptr + (size_t)color_cam * ctx->mwords;
        if (s.frames.ptr && s.have_frame[color_cam])
            e.frame = s.frames.ptr + (size_t)color_cam * ctx->H * ctx->W;
        if (mode == VC_MODE_LUT && ctx->tile_valid && !ctx->ymajor_valid) {
            e.lut = ctx->d_lut_tile.ptr + (size_t)color_cam * p.n_pad;      // the only table there is: tile order
            e.lut_tq = p.tq;
        }
        else if (mode == VC_MODE_LUT) e.lut = ctx->d_lut.ptr + (size_t)color_cam * p.n_pad;
        else if (ctx->fused_color_table && !sb.no_records) {
            // table-free carve, but the colour look-up of the survivors reads the colour camera's table (4 B per
            // voxel of the whole grid, one camera: made in front of this step's carve kernels, see above) instead of
            // projecting every survivor again
            e.lut = ctx->d_lut_color.ptr + ctx->i0();
        }
    }
    e.records = sb.records.ptr;
    e.capacity = sb.records.cap;
    e.busylist = sb.busylist.ptr; e.busycount = sb.busyblock.ptr;
    e.dbg = (uint32_t)ctx->dbg;
    e.stats = p.stats;
    sb.emit_timed = false;
    bool scan_recorded = false;                                  // e_scan recorded by THIS step (Slot::carve_pending may still be set by an earlier one)
    if (want_scan_ev) {
        if (!scan_ridden) VC_HIP(ctx, hipEventRecord(sb.e_scan, ctx->stream));     // cross-stream dependency
        if (!sb.no_records && s3 != ctx->stream) VC_HIP(ctx, hipStreamWaitEvent(s3, sb.e_scan, 0));
        if (xstream) VC_HIP(ctx, hipStreamWaitEvent(sx, sb.e_scan, 0));
        sb.emit_timed = !sb.no_records;
        scan_recorded = true;
    }
    if (!sb.no_records) {
        VC_TRY(launch_emit(ctx, sb, s3, ride ? sb.e_emit0 : nullptr, ride ? sb.e2 : nullptr));
        if (ride) sb.emit_timed = true;
    }
    if (auto_exchange) {
        VC_TRY(enqueue_pack(ctx, sb, sx));
        VC_TRY(enqueue_counts_exchange(ctx, sb, sx));
        sb.counts_exchanged = true;
    }
    if (!ride || auto_exchange) VC_HIP(ctx, hipEventRecord(sb.e2, auto_exchange ? sx : sb.no_records ? s2 : s3));
    if (!sb.no_records && s3 != ctx->stream) { s.e_emit = sb.e2; s.emit_pending = true; }   // the expansion reads the slot's bits / images
    // the slot's next preparation waits for the kernels of THIS step that read its bits / grids: always this step's own event
    // (a flag left set by an earlier step on the same slot must not keep that step's event in place: the preparation would
    // then overwrite bits and grids under this step's carve).  Without e_scan, e2 is behind the carve kernels too.
    s.e_carve = scan_recorded ? sb.e_scan : sb.e2;
    s.carve_pending = true;
    ctx->kev_sb = nullptr;
    sb.pending = true;
    sb.used = true;
    ctx->head = (ctx->head + 1) % kDepth;
    ctx->npending++;
    return VC_OK;
}

// Completes the oldest step in flight: its survivor count, and its records become what the
// vc_fetch_* functions and vc_allgather read.
int vc_carve_end(vc_ctx *ctx, uint64_t *n_out)
{
    if (!ctx || !n_out) return VC_ERR_ARG;
    *n_out = 0;
    if (ctx->npending == 0) return fail(ctx, VC_ERR_ARG, "no carve step in flight");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const int k = (ctx->head - ctx->npending + kDepth) % kDepth;           // oldest pending set
    StepBuf &sb = ctx->sb[k];
    ctx->carved = false; ctx->viewmask_valid = false; ctx->gathered = false; ctx->packed = false;
    if (sb.n != 0) {
        VC_HIP(ctx, hipEventSynchronize(sb.e2));
        uint64_t total = *sb.h_total;
        if (!sb.no_records && total > sb.records.cap) {                    // regrow once, expand again
            // the second expansion reads the step's frame set (bits, image) NOW: if a later vc_carve_begin has prepared the slot
            // again in the meantime, colours and seen flags would come from the newer frame -- refuse instead of mixing
            if (sb.color_cam >= 0 && ctx->slots[sb.slot].gen != sb.slot_gen) {
                sb.pending = false; ctx->npending--;
                return fail(ctx, VC_ERR_ARG, "step overflowed its record buffer (%llu > %llu) and frame set %u has been prepared again "
                            "since: its records cannot be re-expanded; collect a step before re-using its slot with new input",
                            (unsigned long long)total, (unsigned long long)sb.records.cap, sb.slot);
            }
            VC_TRY(ensure(ctx, sb.records, (size_t)(total + total / 8 + 1024)));
            sb.emit.records = sb.records.ptr;
            sb.emit.capacity = sb.records.cap;
            VC_TRY(launch_emit(ctx, sb, ctx->stream));
            sb.emit_ridden = false;
            VC_HIP(ctx, hipEventRecord(sb.e2, ctx->stream));
            VC_HIP(ctx, hipEventSynchronize(sb.e2));
        }
        sb.survivors = total;
        if (sb.carve_timed) {
            float ms = 0;
            VC_HIP(ctx, hipEventElapsedTime(&ms, sb.e0, sb.e1));
            ctx->tm.carve_ms = ms;
            ctx->tm.first_ms = ms;                               // one kernel sequence does the whole carve ...
            if (sb.has_first) VC_HIP(ctx, hipEventElapsedTime(&ctx->tm.first_ms, sb.e0, sb.e_first));   // ... unless a streaming first pass was timed
            ctx->tm.first_ms_sum += ctx->tm.first_ms;
            ctx->tm.carve_ms_sum += ms;
            ctx->tm.carve_launches += 1;
            VC_HIP(ctx, hipEventElapsedTime(&ctx->tm.compact_ms, sb.e1, sb.e2));
        }
        if (sb.emit_timed) {
            // the launch's own begin .. end where they ride on it; else from {scan done}, which includes whatever the expansion
            // stream still had to finish first
            VC_HIP(ctx, hipEventElapsedTime(&ctx->tm.emit_ms, sb.emit_ridden ? sb.e_emit0 : sb.e_scan, sb.e2));
            ctx->tm.emit_ms_sum += ctx->tm.emit_ms;
            ctx->tm.emit_launches += 1;
        }
        if (sb.emit_timed && sb.emit_ridden) { sb.kused[VC_K_EMIT][0] = sb.e_emit0; sb.kused[VC_K_EMIT][1] = sb.e2; sb.kmask |= 1u << VC_K_EMIT; }
        for (int kk = 0; kk < VC_KERNEL_KINDS; ++kk) {
            if (!((sb.kmask >> kk) & 1u)) continue;
            if (kk <= VC_K_PREP_GRID) VC_HIP(ctx, hipEventSynchronize(sb.kused[kk][1]));   // (upload stream: not ordered before e2 by itself)
            float ms = 0;
            if (hipEventElapsedTime(&ms, sb.kused[kk][0], sb.kused[kk][1]) == hipSuccess) {
                ctx->tm.kernel_ms_sum[kk] += ms;
                ctx->tm.kernel_launches[kk] += 1;
            }
        }
        sb.kmask = 0;
        ctx->tm.prep_ms = 0;
        if (sb.prepped) ctx->tm.preps += 1;
        if (sb.prep_timed) {
            Slot &sl = ctx->slots[sb.slot];
            VC_HIP(ctx, hipEventElapsedTime(&ctx->tm.prep_ms, sl.e_p0, sl.e_prep));
            ctx->tm.prep_ms_sum += ctx->tm.prep_ms;
            ctx->tm.preps_timed += 1;
        }
    } else {
        if (sb.counts_exchanged) VC_HIP(ctx, hipEventSynchronize(sb.e2));
        sb.survivors = 0;
    }
    sb.pending = false;
    ctx->npending--;
    ctx->cur = k;
    ctx->survivors = sb.survivors;
    ctx->tm.survivors = sb.survivors;
    ctx->carved = true;
    ctx->viewmask_valid = sb.want_vm;
    *n_out = sb.survivors;
    return VC_OK;
}

int vc_carve(vc_ctx *ctx, uint32_t slot, uint32_t min_views, int color_cam, int mode, uint32_t flags,
             uint64_t *n_out)
{
    if (!ctx || !n_out) return VC_ERR_ARG;
    *n_out = 0;
    if (ctx->npending != 0) return fail(ctx, VC_ERR_ARG, "vc_carve with steps in flight: drain them with vc_carve_end");
    ctx->sync_call = true;
    const int rc = vc_carve_begin(ctx, slot, min_views, color_cam, mode, flags);
    ctx->sync_call = false;
    if (rc != VC_OK) return rc;
    return vc_carve_end(ctx, n_out);
}

// Page-locked host memory for the caller's output buffers: device-to-host copies into it run
// at PCIe rate instead of the pageable-memory rate (the reference has no counterpart; its
// outputs are Python lists).
int vc_host_alloc(vc_ctx *ctx, uint64_t bytes, void **out)
{
    if (!ctx || !out) return VC_ERR_ARG;
    *out = nullptr;
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_HIP(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return VC_OK;
}

int vc_host_free(vc_ctx *ctx, void *ptr)
{
    if (!ctx) return VC_ERR_ARG;
    if (ptr) VC_HIP(ctx, hipHostFree(ptr));
    return VC_OK;
}

int vc_fetch_records(vc_ctx *ctx, uint64_t *records)
{
    if (!ctx || !records) return VC_ERR_ARG;
    if (!ctx->carved) return fail(ctx, VC_ERR_ARG, "no carve result to fetch");
    if (ctx->sb[ctx->cur].no_records)
        return fail(ctx, VC_ERR_ARG, "last carve ran with VC_FLAG_NO_RECORDS: use vc_allgather / vc_expand_entries");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->survivors)
        VC_HIP(ctx, hipMemcpy(records, ctx->sb[ctx->cur].records.ptr, ctx->survivors * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return VC_OK;
}

int vc_fetch(vc_ctx *ctx, uint32_t *idx, uint8_t *rgb, uint8_t *seen)
{
    if (!ctx) return VC_ERR_ARG;
    if (!ctx->carved) return fail(ctx, VC_ERR_ARG, "no carve result to fetch");
    const uint64_t S = ctx->survivors;
    if (S == 0) return VC_OK;
    std::vector<uint64_t> rec(S);
    VC_TRY(vc_fetch_records(ctx, rec.data()));
    for (uint64_t k = 0; k < S; ++k) {
        const uint64_t r = rec[k];
        if (idx) idx[k] = (uint32_t)r;
        if (rgb) { rgb[3 * k] = (uint8_t)(r >> 32); rgb[3 * k + 1] = (uint8_t)(r >> 40); rgb[3 * k + 2] = (uint8_t)(r >> 48); }
        if (seen) seen[k] = (uint8_t)((r >> 56) & 1);
    }
    return VC_OK;
}

int vc_fetch_viewmask(vc_ctx *ctx, uint16_t *viewmask)
{
    if (!ctx || !viewmask) return VC_ERR_ARG;
    if (!ctx->carved || !ctx->viewmask_valid) return fail(ctx, VC_ERR_ARG, "last carve did not keep the view mask (VC_FLAG_VIEWMASK)");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = ctx->n_voxels();
    if (n) VC_HIP(ctx, hipMemcpy(viewmask, ctx->d_viewmask.ptr, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return VC_OK;
}

int vc_fetch_occupancy(vc_ctx *ctx, uint8_t *bits)
{
    if (!ctx || !bits) return VC_ERR_ARG;
    if (!ctx->carved) return fail(ctx, VC_ERR_ARG, "no carve result to fetch");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t nwords = (ctx->n_voxels() + 63) / 64;
    StepBuf &cur = ctx->sb[ctx->cur];
    if (nwords && cur.sparse_words) {            // the hierarchical kernels skip the words of groups without survivors
        const uint32_t ngroups = (uint32_t)((nwords + kGroupWords - 1) / kGroupWords);
        hipLaunchKernelGGL(k_zero_dead_groups, dim3((ngroups + 3) / 4), dim3(kBlock), 0, ctx->stream, cur.words.ptr, nwords,
                           ngroups, cur.groupcnt.ptr);
        VC_HIP(ctx, hipGetLastError());
        VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        cur.sparse_words = false;
    }
    if (nwords) VC_HIP(ctx, hipMemcpy(bits, cur.words.ptr, nwords * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return VC_OK;
}

// ---- marching cubes over the dense ON/OFF volume (reference consumer voxel_reconstruction.py:127-163) ----
int vc_marching_cubes(vc_ctx *ctx, const uint8_t *volume_bits, uint32_t d0, uint32_t d1, uint32_t d2, float level,
                      uint64_t *n_verts, uint64_t *n_faces)
{
    if (!ctx || !n_verts || !n_faces) return VC_ERR_ARG;
    *n_verts = *n_faces = 0;
    ctx->mc_valid = false;
    if (d0 == 0 || d1 == 0 || d2 == 0) return fail(ctx, VC_ERR_ARG, "volume dimensions must be >= 1");
    if (!(level >= 0.0f && level < 1.0f)) return fail(ctx, VC_ERR_ARG, "level %g not in [0, 1): ON is 1, OFF is 0", (double)level);
    const uint64_t n = (uint64_t)d0 * d1 * d2;
    if (n > 0xffffffffull) return fail(ctx, VC_ERR_ARG, "volume of %llu elements exceeds the u32 index", (unsigned long long)n);
    if (ctx->npending) return fail(ctx, VC_ERR_ARG, "carve steps are in flight: collect them with vc_carve_end first");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t nwords = (uint32_t)((n + 63) / 64), ngroups = (nwords + 63) / 64;
    const uint32_t nscan = (ngroups + kScanBlock - 1) / kScanBlock;
    McParams p;
    memset(&p, 0, sizeof p);
    if (volume_bits) {
        VC_TRY(ensure(ctx, ctx->d_mcbits, (size_t)nwords));
        VC_HIP(ctx, hipMemsetAsync(ctx->d_mcbits.ptr, 0, (size_t)nwords * sizeof(uint64_t), ctx->stream));
        VC_HIP(ctx, hipMemcpyAsync(ctx->d_mcbits.ptr, volume_bits, (size_t)((n + 7) / 8), hipMemcpyHostToDevice, ctx->stream));
        p.bits = ctx->d_mcbits.ptr;
    } else {
        if (!ctx->carved) return fail(ctx, VC_ERR_ARG, "no carve result: run vc_carve first or pass a volume");
        if (n != ctx->n_voxels()) return fail(ctx, VC_ERR_ARG, "%u x %u x %u is not the %llu voxels of the carved slab", d0, d1, d2,
                                              (unsigned long long)ctx->n_voxels());
        StepBuf &cur = ctx->sb[ctx->cur];
        if (cur.sparse_words) {                  // the hierarchical kernels skip the words of groups without survivors
            const uint32_t cg = (uint32_t)((nwords + kGroupWords - 1) / kGroupWords);
            hipLaunchKernelGGL(k_zero_dead_groups, dim3((cg + 3) / 4), dim3(kBlock), 0, ctx->stream, cur.words.ptr, (uint64_t)nwords, cg, cur.groupcnt.ptr);
            VC_HIP(ctx, hipGetLastError());
            cur.sparse_words = false;
        }
        p.bits = cur.words.ptr;
    }
    VC_TRY(ensure(ctx, ctx->d_mcx, (size_t)nwords * 3));
    VC_TRY(ensure(ctx, ctx->d_mcwbase, (size_t)nwords));
    VC_TRY(ensure(ctx, ctx->d_mcgv, ngroups)); VC_TRY(ensure(ctx, ctx->d_mcgt, ngroups));
    VC_TRY(ensure(ctx, ctx->d_mcgvoff, ngroups)); VC_TRY(ensure(ctx, ctx->d_mcgtoff, ngroups));
    VC_TRY(ensure(ctx, ctx->d_mcbv, kMaxScanBlocks)); VC_TRY(ensure(ctx, ctx->d_mcbt, kMaxScanBlocks));
    VC_TRY(ensure(ctx, ctx->d_mcbvoff, kMaxScanBlocks + 1)); VC_TRY(ensure(ctx, ctx->d_mcbtoff, kMaxScanBlocks + 1));
    VC_TRY(ensure_exchange_scratch(ctx, 1));
    p.n = n; p.d0 = d0; p.d1 = d1; p.d2 = d2; p.nwords = nwords; p.ngroups = ngroups;
    p.x = ctx->d_mcx.ptr; p.wbase = ctx->d_mcwbase.ptr; p.gv = ctx->d_mcgv.ptr; p.gt = ctx->d_mcgt.ptr;
    p.gvoff = ctx->d_mcgvoff.ptr; p.gtoff = ctx->d_mcgtoff.ptr; p.bvoff = ctx->d_mcbvoff.ptr; p.btoff = ctx->d_mcbtoff.ptr;
    p.level = level;
    const dim3 grid((ngroups + 3) / 4), block(kBlock);
    hipLaunchKernelGGL(k_mc_count, grid, block, 0, ctx->stream, p);
    VC_HIP(ctx, hipGetLastError());
    VC_TRY(scan_counts(ctx, ctx->stream, ctx->d_mcgv.ptr, ngroups, ctx->d_mcgvoff.ptr, ctx->d_mcbv.ptr, ctx->d_mcbvoff.ptr, ctx->h_xtotal));
    VC_TRY(scan_counts(ctx, ctx->stream, ctx->d_mcgt.ptr, ngroups, ctx->d_mcgtoff.ptr, ctx->d_mcbt.ptr, ctx->d_mcbtoff.ptr, ctx->h_xtotal + 1));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)nscan;
    const uint64_t V = ctx->h_xtotal[0], F = ctx->h_xtotal[1];
    if (V > 0xffffffffull) return fail(ctx, VC_ERR_ARG, "%llu vertices exceed the u32 vertex number", (unsigned long long)V);
    VC_TRY(ensure(ctx, ctx->d_mcverts, (size_t)(3 * V + 3)));
    VC_TRY(ensure(ctx, ctx->d_mcfaces, (size_t)(3 * F + 3)));
    p.verts = ctx->d_mcverts.ptr; p.faces = ctx->d_mcfaces.ptr; p.vcap = V; p.fcap = F;
    hipLaunchKernelGGL(k_mc_verts, grid, block, 0, ctx->stream, p);
    hipLaunchKernelGGL(k_mc_faces, grid, block, 0, ctx->stream, p);
    VC_HIP(ctx, hipGetLastError());
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->mc_verts = V; ctx->mc_faces = F; ctx->mc_valid = true;
    *n_verts = V; *n_faces = F;
    return VC_OK;
}

int vc_fetch_mesh(vc_ctx *ctx, float *verts, uint32_t *faces)
{
    if (!ctx) return VC_ERR_ARG;
    if (!ctx->mc_valid) return fail(ctx, VC_ERR_ARG, "no mesh: call vc_marching_cubes");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    if (verts && ctx->mc_verts) VC_HIP(ctx, hipMemcpy(verts, ctx->d_mcverts.ptr, ctx->mc_verts * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (faces && ctx->mc_faces) VC_HIP(ctx, hipMemcpy(faces, ctx->d_mcfaces.ptr, ctx->mc_faces * 3 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return VC_OK;
}

// ---- the step before the path, its data-parallel part (reference background_subtraction.py:153-168) ----
static int hsv_tables(vc_ctx *ctx);

int vc_bgr_to_hsv(vc_ctx *ctx, const uint8_t *bgr, uint32_t H, uint32_t W, uint8_t *hsv)
{
    if (!ctx || !bgr || !hsv) return VC_ERR_ARG;
    if (H == 0 || W == 0 || (uint64_t)H * W > 0x3fffffffull) return fail(ctx, VC_ERR_ARG, "image size %u x %u", H, W);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)H * W;
    hipStream_t st = ctx->stream_up;
    VC_TRY(hsv_tables(ctx));
    VC_TRY(ensure(ctx, ctx->d_fg, npix * 6 + 64));
    uint8_t *d_in = ctx->d_fg.ptr, *d_out = d_in + npix * 3;
    VC_HIP(ctx, hipMemcpyAsync(d_in, bgr, npix * 3, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_bgr2hsv, dim3((uint32_t)((npix + 255) / 256)), dim3(256), 0, st, (const uint8_t *)d_in, d_out, (uint32_t)npix,
                       (const int32_t *)ctx->d_hsvdiv.ptr, (const int32_t *)(ctx->d_hsvdiv.ptr + 256));
    VC_HIP(ctx, hipGetLastError());
    VC_HIP(ctx, hipMemcpyAsync(hsv, d_out, npix * 3, hipMemcpyDeviceToHost, st));
    VC_HIP(ctx, hipStreamSynchronize(st));
    return VC_OK;
}

int vc_mask_morphology(vc_ctx *ctx, const uint8_t *mask, uint32_t H, uint32_t W, uint32_t ksize, int open, int close, uint8_t *out)
{
    if (!ctx || !mask || !out) return VC_ERR_ARG;
    if (ksize != 2 && ksize != 3) return fail(ctx, VC_ERR_ARG, "structuring element %u x %u: the reference uses 3 x 3 (pre) and 2 x 2 (post)", ksize, ksize);
    if (H == 0 || W == 0 || (uint64_t)H * W > 0xffffffffull) return fail(ctx, VC_ERR_ARG, "image size %u x %u", H, W);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)H * W;
    hipStream_t st = ctx->stream_up;
    VC_TRY(ensure(ctx, ctx->d_fg, npix * 6 + 64));
    uint8_t *a = ctx->d_fg.ptr, *b = a + npix;
    VC_HIP(ctx, hipMemcpyAsync(a, mask, npix, hipMemcpyHostToDevice, st));
    const dim3 g((uint32_t)((npix + 255) / 256)), blk(256);
    auto pass = [&](bool dilate) {                               // a -> b, then the two swap
        if (ksize == 3) {
            if (dilate) hipLaunchKernelGGL(k_morph3x3<true>, g, blk, 0, st, (const uint8_t *)a, b, H, W);
            else hipLaunchKernelGGL(k_morph3x3<false>, g, blk, 0, st, (const uint8_t *)a, b, H, W);
        } else {
            if (dilate) hipLaunchKernelGGL(k_morph2x2<true>, g, blk, 0, st, (const uint8_t *)a, b, H, W);
            else hipLaunchKernelGGL(k_morph2x2<false>, g, blk, 0, st, (const uint8_t *)a, b, H, W);
        }
        uint8_t *t = a; a = b; b = t;
    };
    if (open) { pass(false); pass(true); }                       // MORPH_OPEN = erode, dilate
    if (close) { pass(true); pass(false); }                      // MORPH_CLOSE = dilate, erode (opening first when both are asked)
    VC_HIP(ctx, hipGetLastError());
    VC_HIP(ctx, hipMemcpyAsync(out, a, npix, hipMemcpyDeviceToHost, st));
    VC_HIP(ctx, hipStreamSynchronize(st));
    return VC_OK;
}

int vc_mog_create(vc_ctx *ctx, int history, int nmixtures, double background_ratio, double noise_sigma, uint32_t *model)
{
    if (!ctx || !model) return VC_ERR_ARG;
    for (uint32_t i = 0; i < VC_MAX_MOG_MODELS; ++i) {
        vc_ctx::MogModel &m = ctx->mog[i];
        if (m.used) continue;
        // the constructor's clamps (bgfg_gaussmix.cpp, BackgroundSubtractorMOGImpl): non-positive arguments select the defaults
        m.nmixtures = nmixtures > 0 ? nmixtures : 5;
        if (m.nmixtures > kMogMaxMixtures) m.nmixtures = kMogMaxMixtures;
        m.history = history > 0 ? history : 200;
        m.background_ratio = background_ratio > 0 ? background_ratio : 0.95;
        if (m.background_ratio > 1.0) m.background_ratio = 1.0;
        m.noise_sigma = noise_sigma <= 0 ? 15.0 : noise_sigma;
        m.H = m.W = m.nframes = 0;
        m.used = true;
        *model = i;
        return VC_OK;
    }
    return fail(ctx, VC_ERR_ARG, "all %d background models of this context are in use", VC_MAX_MOG_MODELS);
}

int vc_mog_destroy(vc_ctx *ctx, uint32_t model)
{
    if (!ctx) return VC_ERR_ARG;
    if (model >= VC_MAX_MOG_MODELS || !ctx->mog[model].used) return fail(ctx, VC_ERR_ARG, "no background model %u", model);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream_up));
    release(ctx->mog[model].state);
    ctx->mog[model].used = false;
    return VC_OK;
}

// One frame through a background model: d_img (device, [H W 3]) -> d_mask (device, [H W]) on the upload stream.
static int mog_enqueue(vc_ctx *ctx, uint32_t model, const uint8_t *d_img, uint32_t H, uint32_t W, double learning_rate, uint8_t *d_mask)
{
    vc_ctx::MogModel &m = ctx->mog[model];
    const size_t npix = (size_t)H * W;
    hipStream_t st = ctx->stream_up;
    // apply(): the model starts over on its first frame, on a learning rate >= 1 and when the image size changes
    if (m.nframes == 0 || learning_rate >= 1 || m.H != H || m.W != W) {
        VC_TRY(ensure(ctx, m.state, npix * 8 * (size_t)m.nmixtures));
        VC_HIP(ctx, hipMemsetAsync(m.state.ptr, 0, npix * 8 * (size_t)m.nmixtures * sizeof(float), st));
        m.H = H; m.W = W; m.nframes = 0;
    }
    ++m.nframes;
    const double lr = learning_rate >= 0 && m.nframes > 1 ? learning_rate : 1.0 / (double)(m.nframes < (uint32_t)m.history ? m.nframes : (uint32_t)m.history);
    const double default_noise_sigma = 30 * 0.5, w0 = 0.05;
    MogParams p;
    p.alpha = (float)lr; p.T = (float)m.background_ratio; p.vT = (float)(2.5 * 2.5);
    p.w0 = (float)w0;
    p.sk0 = (float)(w0 / (default_noise_sigma * 2 * std::sqrt(3.)));
    p.var0 = (float)(default_noise_sigma * default_noise_sigma * 4);
    p.minVar = (float)(m.noise_sigma * m.noise_sigma);
    p.K = (uint32_t)m.nmixtures; p.npix = (uint32_t)npix;
    hipLaunchKernelGGL(k_mog_apply, dim3((uint32_t)((npix + 255) / 256)), dim3(256), 0, st, d_img, d_mask, m.state.ptr, p);
    VC_HIP(ctx, hipGetLastError());
    return VC_OK;
}

static int hsv_tables(vc_ctx *ctx)
{
    if (ctx->d_hsvdiv.ptr) return VC_OK;
    // as OpenCV builds them (color_hsv: RGB2HSV_b): saturate_cast<int>(double) = round half to even
    int32_t t[512];
    t[0] = t[256] = 0;
    for (int i = 1; i < 256; ++i) {
        t[i] = (int32_t)std::nearbyint((double)(255 << kHsvShift) / (1.0 * i));
        t[256 + i] = (int32_t)std::nearbyint((double)(180 << kHsvShift) / (6.0 * i));
    }
    VC_TRY(ensure(ctx, ctx->d_hsvdiv, 512));
    VC_HIP(ctx, hipMemcpy(ctx->d_hsvdiv.ptr, t, sizeof t, hipMemcpyHostToDevice));
    return VC_OK;
}

int vc_mog_apply(vc_ctx *ctx, uint32_t model, const uint8_t *image, uint32_t H, uint32_t W, double learning_rate, uint8_t *fgmask)
{
    if (!ctx || !image || !fgmask) return VC_ERR_ARG;
    if (model >= VC_MAX_MOG_MODELS || !ctx->mog[model].used) return fail(ctx, VC_ERR_ARG, "no background model %u", model);
    if (H == 0 || W == 0 || (uint64_t)H * W > 0x0fffffffull) return fail(ctx, VC_ERR_ARG, "image size %u x %u", H, W);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)H * W;
    hipStream_t st = ctx->stream_up;
    VC_TRY(ensure(ctx, ctx->d_fg, npix * 6 + 64));
    uint8_t *d_in = ctx->d_fg.ptr, *d_out = d_in + npix * 3;
    VC_HIP(ctx, hipMemcpyAsync(d_in, image, npix * 3, hipMemcpyHostToDevice, st));
    VC_TRY(mog_enqueue(ctx, model, d_in, H, W, learning_rate, d_out));
    VC_HIP(ctx, hipMemcpyAsync(fgmask, d_out, npix, hipMemcpyDeviceToHost, st));
    VC_HIP(ctx, hipStreamSynchronize(st));
    return VC_OK;
}

int vc_foreground_front(vc_ctx *ctx, uint32_t model, const uint8_t *bgr, uint32_t H, uint32_t W, int to_hsv, double learning_rate,
                        int open, int close, uint8_t *mask)
{
    if (!ctx || !bgr || !mask) return VC_ERR_ARG;
    if (model >= VC_MAX_MOG_MODELS || !ctx->mog[model].used) return fail(ctx, VC_ERR_ARG, "no background model %u", model);
    if (H == 0 || W == 0 || (uint64_t)H * W > 0x0fffffffull) return fail(ctx, VC_ERR_ARG, "image size %u x %u", H, W);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)H * W;
    hipStream_t st = ctx->stream_up;
    VC_TRY(hsv_tables(ctx));
    VC_TRY(ensure(ctx, ctx->d_fg, npix * 8 + 64));
    uint8_t *d_in = ctx->d_fg.ptr, *d_hsv = d_in + npix * 3, *a = d_hsv + npix * 3, *b = a + npix;
    VC_HIP(ctx, hipMemcpyAsync(d_in, bgr, npix * 3, hipMemcpyHostToDevice, st));
    const dim3 g((uint32_t)((npix + 255) / 256)), blk(256);
    if (to_hsv) {
        hipLaunchKernelGGL(k_bgr2hsv, g, blk, 0, st, (const uint8_t *)d_in, d_hsv, (uint32_t)npix, (const int32_t *)ctx->d_hsvdiv.ptr,
                           (const int32_t *)(ctx->d_hsvdiv.ptr + 256));
        VC_HIP(ctx, hipGetLastError());
    }
    VC_TRY(mog_enqueue(ctx, model, to_hsv ? d_hsv : d_in, H, W, learning_rate, a));
    auto pass = [&](bool dilate) {
        if (dilate) hipLaunchKernelGGL(k_morph3x3<true>, g, blk, 0, st, (const uint8_t *)a, b, H, W);
        else hipLaunchKernelGGL(k_morph3x3<false>, g, blk, 0, st, (const uint8_t *)a, b, H, W);
        uint8_t *t = a; a = b; b = t;
    };
    if (open) { pass(false); pass(true); }
    if (close) { pass(true); pass(false); }
    VC_HIP(ctx, hipGetLastError());
    VC_HIP(ctx, hipMemcpyAsync(mask, a, npix, hipMemcpyDeviceToHost, st));
    VC_HIP(ctx, hipStreamSynchronize(st));
    return VC_OK;
}

int vc_mog_state(vc_ctx *ctx, uint32_t model, float *state, uint64_t capacity, uint32_t *H, uint32_t *W, uint32_t *nmixtures, uint32_t *nframes)
{
    if (!ctx) return VC_ERR_ARG;
    if (model >= VC_MAX_MOG_MODELS || !ctx->mog[model].used) return fail(ctx, VC_ERR_ARG, "no background model %u", model);
    const vc_ctx::MogModel &m = ctx->mog[model];
    if (H) *H = m.H;
    if (W) *W = m.W;
    if (nmixtures) *nmixtures = (uint32_t)m.nmixtures;
    if (nframes) *nframes = m.nframes;
    if (!state) return VC_OK;
    const size_t nfloat = (size_t)m.H * m.W * 8 * (size_t)m.nmixtures;
    if (capacity < nfloat) return fail(ctx, VC_ERR_ARG, "state buffer holds %llu floats, the model has %llu", (unsigned long long)capacity, (unsigned long long)nfloat);
    if (nfloat == 0) return VC_OK;
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream_up));
    VC_HIP(ctx, hipMemcpy(state, m.state.ptr, nfloat * sizeof(float), hipMemcpyDeviceToHost));
    return VC_OK;
}

int vc_set_option(vc_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name) return VC_ERR_ARG;
    const std::string k(name);
    if (k == "force_generic") ctx->force_generic = value != 0;
    else if (k == "reorder") ctx->reorder = value != 0;
    else if (k == "lut_hier") ctx->lut_hier = value != 0;
    else if (k == "fused_hier") ctx->fused_hier = value != 0;
    else if (k == "emit_lanes") ctx->emit_lanes = value != 0;
    else if (k == "overlap") ctx->overlap = value != 0;
    else if (k == "timing_detail") ctx->timing_detail = value != 0;
    else if (k == "kernel_events") ctx->kernel_events = value != 0;
    else if (k == "launch_events") ctx->launch_events = value != 0;
    else if (k == "event_scope" && value >= 0 && value <= 2) {
        if (ctx->npending) return fail(ctx, VC_ERR_ARG, "carve steps are in flight: collect them with vc_carve_end first");
        VC_HIP(ctx, hipSetDevice(ctx->device));
        VC_TRY(vc_synchronize(ctx));
        ctx->event_scope = value;
        VC_HIP(ctx, make_events(ctx));
    }
    else if ((k == "stream_priority" && (value == 0 || value == 1)) || (k == "reserve_cus" && value >= 0 && value <= 16)) {
        if (ctx->npending) return fail(ctx, VC_ERR_ARG, "carve steps are in flight: collect them with vc_carve_end first");
        VC_HIP(ctx, hipSetDevice(ctx->device));
        VC_TRY(vc_synchronize(ctx));
        (k == "stream_priority" ? ctx->stream_priority : ctx->reserve_cus) = value;
        VC_HIP(ctx, make_streams(ctx));
    }
    else if (k == "cull") ctx->cull = value != 0;
    else if (k == "bricks") ctx->bricks = value != 0;
    else if (k == "dbg") ctx->dbg = value;
    else if (k == "voxel_pairs") ctx->voxel_pairs = value;
    else if (k == "voxel_batches" && value >= 0 && value <= 16) ctx->voxel_batches = value;
    else if (k == "emit_busy" && value >= 0 && value <= 2) ctx->emit_busy = value;          // 0 never, 1 large grids, 2 always
    else if (k == "emit_waves_per_cu" && value >= 4 && value <= 1024) ctx->emit_waves_per_cu = value;
    else if (k == "lut_tile") ctx->lut_tile = value != 0;
    else if (k == "grid_lds_kb" && value >= 0 && value <= 148) ctx->grid_lds_kb = value;
    else if (k == "grid_min_shift" && value >= 0 && value <= 8) ctx->grid_min_shift = value;
    else if (k == "fused_tile") ctx->fused_tile = value != 0;
    else if (k == "fused_f32box") ctx->fused_f32box = value != 0;
    else if (k == "fused_boxes") ctx->fused_boxes = value != 0;
    else if (k == "fused_color_table") ctx->fused_color_table = value != 0;
    else if (k == "gather_compact") ctx->gather_compact = value != 0;
    else if (k == "gather_sync") ctx->gather_sync = value != 0;
    else if (k == "refine_pair") ctx->refine_pair = value != 0;
    else if (k == "hier_blocks_per_cu" && value >= 1 && value <= 4096) ctx->hier_blocks_per_cu = value;
    else if (k == "first_kv" && (value == 1 || value == 2 || value == 4)) ctx->first_kv = value;
    else if (k == "first_blocks_per_cu" && value >= 1 && value <= 8) ctx->first_blocks_per_cu = value;
    else if (k == "refine_b" && (value == 8 || value == 16)) ctx->refine_b = value;
    else if (k == "refine_blocks_per_cu" && value >= 1 && value <= 64) ctx->refine_blocks_per_cu = value;
    else if (k == "fused_blocks_per_cu" && value >= 1 && value <= 16) ctx->fused_blocks_per_cu = value;
    else return fail(ctx, VC_ERR_ARG, "unknown option or bad value: %s = %d", name, value);
    return VC_OK;
}

int vc_debug_counters(vc_ctx *ctx, uint64_t out[8])
{
    if (!ctx || !out) return VC_ERR_ARG;
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memset(out, 0, 8 * sizeof(uint64_t));
    CarveParams p;
    fill_params(ctx, p);
    if (ctx->d_blist.ptr) {
        std::vector<uint32_t> c(6 * (size_t)kShards * kShardStride);
        VC_HIP(ctx, hipMemcpy(c.data(), ctx->d_blist.ptr, c.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t *q = c.data() + (size_t)ctx->list_parity * 3 * kShards * kShardStride;
        for (uint32_t k = 0; k < kShards; ++k) {
            out[0] += q[k * kShardStride]; out[4] += q[(kShards + k) * kShardStride]; out[5] += q[(2 * kShards + k) * kShardStride];
        }
    }
    if (ctx->d_live.ptr && ctx->kbox_valid) {
        const size_t nw = p.nbrick_pad / 64;
        std::vector<uint64_t> bits(2 * nw);
        VC_HIP(ctx, hipMemcpy(bits.data(), ctx->d_live.ptr, bits.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nw; ++i) { out[1] += (uint64_t)__builtin_popcountll(bits[i]); out[2] += (uint64_t)__builtin_popcountll(bits[nw + i]); }
        out[3] = (uint64_t)p.nbx * p.tq * p.nbz;
    }
    return VC_OK;
}

int vc_timing(vc_ctx *ctx, vc_timing_t *out)
{
    if (!ctx || !out) return VC_ERR_ARG;
    VC_TRY(finish_gather(ctx));
    if (ctx->h2d_pending && hipEventQuery(ctx->ev_h[1]) == hipSuccess) {
        (void)hipEventElapsedTime(&ctx->tm.h2d_ms, ctx->ev_h[0], ctx->ev_h[1]);
        ctx->h2d_pending = false;
    }
    memset(ctx->tm.work, 0, sizeof ctx->tm.work);
    if (ctx->d_stats.ptr && ctx->npending == 0) {
        std::vector<unsigned long long> h(ctx->d_stats.cap);
        VC_HIP(ctx, hipSetDevice(ctx->device));
        VC_HIP(ctx, hipMemcpy(h.data(), ctx->d_stats.ptr, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int w = 0; w < VC_WORK_KINDS; ++w)
            for (uint32_t k = 0; k < kShards; ++k) ctx->tm.work[w] += h[((size_t)w * kShards + k) * kStatStride];
    }
    *out = ctx->tm;
    return VC_OK;
}

uint32_t vc_timing_struct_size(void) { return (uint32_t)sizeof(vc_timing_t); }

int vc_timing_reset(vc_ctx *ctx)
{
    if (!ctx) return VC_ERR_ARG;
    ctx->tm.carve_launches = 0;
    ctx->tm.carve_ms_sum = 0;
    ctx->tm.first_ms_sum = 0;
    ctx->tm.gather_ms_sum = 0;
    ctx->tm.gathers = 0;
    ctx->tm.prep_ms_sum = 0;
    ctx->tm.preps = 0;
    ctx->tm.preps_timed = 0;
    ctx->tm.emit_ms_sum = 0;
    ctx->tm.emit_launches = 0;
    memset(ctx->tm.kernel_ms_sum, 0, sizeof ctx->tm.kernel_ms_sum);
    memset(ctx->tm.kernel_launches, 0, sizeof ctx->tm.kernel_launches);
    if (ctx->d_stats.ptr && ctx->npending == 0) {
        VC_HIP(ctx, hipSetDevice(ctx->device));
        VC_HIP(ctx, hipMemset(ctx->d_stats.ptr, 0, ctx->d_stats.cap * sizeof(unsigned long long)));
    }
    return VC_OK;
}

// ---------------------------------------------------------------- multi-GPU
int vc_comm_unique_id(uint8_t out[VC_UNIQUE_ID_BYTES])
{
    if (!out) return VC_ERR_ARG;
    std::string err;
    if (!load_rccl(err)) return fail(nullptr, VC_ERR_RCCL, "%s", err.c_str());
    static_assert(sizeof(ncclUniqueId) == VC_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    VC_NCCL(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(out, &id, sizeof id);
    return VC_OK;
}

int vc_comm_init(vc_ctx *ctx, int n_ranks, int rank, const uint8_t uid[VC_UNIQUE_ID_BYTES])
{
    if (!ctx || !uid) return VC_ERR_ARG;
    if (n_ranks < 1 || n_ranks > VC_MAX_RANKS || rank < 0 || rank >= n_ranks) return fail(ctx, VC_ERR_ARG, "rank %d of %d", rank, n_ranks);
    if (ctx->npending) return fail(ctx, VC_ERR_ARG, "carve steps are in flight: collect them with vc_carve_end first");
    std::string err;
    if (!load_rccl(err)) return fail(ctx, VC_ERR_RCCL, "%s", err.c_str());
    VC_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->comm) { g_rccl.CommDestroy(ctx->comm); ctx->comm = nullptr; }
    ncclUniqueId id;
    memcpy(&id, uid, sizeof id);
    VC_NCCL(ctx, g_rccl.CommInitRank(&ctx->comm, n_ranks, id, rank));
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    VC_TRY(ensure(ctx, ctx->d_counts, (size_t)n_ranks + 1));
    if (ctx->h_counts) { (void)hipHostFree(ctx->h_counts); ctx->h_counts = nullptr; }
    VC_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_counts), sizeof(uint64_t) * n_ranks, hipHostMallocDefault));
    return VC_OK;
}

int vc_comm_destroy(vc_ctx *ctx)
{
    if (!ctx) return VC_ERR_ARG;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamSynchronize(ctx->stream_x);
        ctx->gpend[0] = ctx->gpend[1] = false;
        VC_NCCL(ctx, g_rccl.CommDestroy(ctx->comm));
        ctx->comm = nullptr;
    }
    ctx->n_ranks = 1; ctx->rank = 0;
    return VC_OK;
}

// ---- compact exchange form: the slab's non-zero occupancy words ----------------------------------
int vc_pack_entries(vc_ctx *ctx, uint64_t *n_entries_out)
{
    if (!ctx || !n_entries_out) return VC_ERR_ARG;
    if (!ctx->carved) return fail(ctx, VC_ERR_ARG, "no carve result to pack");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_TRY(enqueue_pack(ctx, ctx->sb[ctx->cur], ctx->stream));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->packed_entries = *ctx->h_xtotal;
    ctx->packed = true;
    *n_entries_out = ctx->packed_entries;
    return VC_OK;
}

int vc_fetch_entries(vc_ctx *ctx, uint64_t *entries)
{
    if (!ctx || !entries) return VC_ERR_ARG;
    if (!ctx->packed) return fail(ctx, VC_ERR_ARG, "no packed result: call vc_pack_entries");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->packed_entries)
        VC_HIP(ctx, hipMemcpy(entries, ctx->sb[ctx->cur].ent.ptr, ctx->packed_entries * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return VC_OK;
}

int vc_expand_entries(vc_ctx *ctx, const uint64_t *entries, uint64_t n_entries, uint64_t *total_out)
{
    if (!ctx || !total_out || (!entries && n_entries)) return VC_ERR_ARG;
    if (!ctx->carved) return fail(ctx, VC_ERR_ARG, "vc_expand_entries colours like the last carve: run one first");
    if (n_entries > (1ull << 26)) return fail(ctx, VC_ERR_ARG, "%llu entries exceed a u32 grid", (unsigned long long)n_entries);
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_TRY(finish_gather(ctx));
    ctx->gathered = false;
    VC_TRY(ensure_exchange_scratch(ctx, 1));
    VC_TRY(ensure(ctx, ctx->d_ent_all[0], (size_t)(2 * n_entries)));
    if (n_entries)
        VC_HIP(ctx, hipMemcpyAsync(ctx->d_ent_all[0].ptr, entries, n_entries * 2 * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    VC_TRY(enqueue_expand(ctx, ctx->stream, ctx->d_ent_all[0].ptr, n_entries, 0));
    VC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->gathered_total = n_entries ? *(ctx->h_xtotal + 1) : 0;
    ctx->gathered = true;
    *total_out = ctx->gathered_total;
    return VC_OK;
}

// Compact form of vc_allgather: every rank packs its non-zero words, the {bits, base} pairs are
// exchanged (~12x fewer bytes over xGMI than the records they stand for at 1024^3), and every rank
// expands all pairs itself -- colours from its own copy of the colour camera's table and frame.
static int allgather_compact(vc_ctx *ctx, uint64_t *counts_out, uint64_t *total_out)
{
    const int G = ctx->n_ranks;
    StepBuf &cur = ctx->sb[ctx->cur];
    hipStream_t sx = ctx->overlap ? ctx->stream_x : ctx->stream;    // every collective of the compact form is queued here
    const uint32_t half = ctx->gseq & 1u;
    VC_TRY(finish_one(ctx, half));                               // the gather before last owned this half; the last one may still run
    DevBuf<uint64_t> &ent_all = ctx->d_ent_all[half];
    hipEvent_t *E = ctx->gx[ctx->gx_next];                       // this gather's own events: see vc_ctx::gx
    ctx->gx_idx[half] = ctx->gx_next;
    ctx->gx_next = (ctx->gx_next + 1) % kGatherRing;
    VC_HIP(ctx, hipEventRecord(E[0], sx));
    if (!cur.counts_exchanged) {                 // vc_carve_begin did not do it (records were kept)
        VC_TRY(enqueue_pack(ctx, cur, sx));
        VC_TRY(enqueue_counts_exchange(ctx, cur, sx));
        VC_HIP(ctx, hipStreamSynchronize(sx));
    }
    uint64_t M = 0, S = 0;
    for (int r = 0; r < G; ++r) { M += cur.h_counts[2 * r]; S += cur.h_counts[2 * r + 1]; }
    if (2 * M > ent_all.cap) VC_TRY(ensure(ctx, ent_all, (size_t)(2 * M + M / 4 + 1024)));
    VC_TRY(ensure(ctx, cur.ent, 2));
    VC_NCCL(ctx, g_rccl.GroupStart());
    uint64_t disp = 0;
    for (int r = 0; r < G; ++r) {
        const uint64_t cnt = cur.h_counts[2 * r];
        if (cnt) {
            ncclResult_t rc = g_rccl.Broadcast(cur.ent.ptr, ent_all.ptr + 2 * disp, 2 * cnt, ncclUint64, r,
                                               ctx->comm, sx);
            if (rc != ncclSuccess) {
                g_rccl.GroupEnd();
                return fail(ctx, VC_ERR_RCCL, "ncclBroadcast(root %d): %s", r, g_rccl.GetErrorString(rc));
            }
        }
        disp += cnt;
    }
    VC_NCCL(ctx, g_rccl.GroupEnd());
    VC_HIP(ctx, hipEventRecord(E[2], sx));
    // the expansion runs beside the next step's carve (second stream) when the call does not wait for it anyway
    hipStream_t xs = (ctx->overlap && !ctx->gather_sync) ? ctx->stream2 : sx;
    if (xs != sx) VC_HIP(ctx, hipStreamWaitEvent(xs, E[2], 0));
    if (S) VC_TRY(enqueue_expand(ctx, xs, ent_all.ptr, M, S, ctx->h_xtotal + 2 + half));
    VC_HIP(ctx, hipEventRecord(E[1], xs));
    if (S && cur.color_cam >= 0) {          // the expansion reads the slot's bits / images beside the carve stream
        Slot &sl = ctx->slots[cur.slot];
        sl.e_emit = E[1];
        sl.emit_pending = true;
    }
    ctx->gpend[half] = true;
    ctx->gexpect[half] = S;
    ctx->gseq++;
    if (counts_out) for (int r = 0; r < G; ++r) counts_out[r] = cur.h_counts[2 * r + 1];
    ctx->gathered_total = S;
    ctx->gathered = true;
    *total_out = S;
    if (ctx->gather_sync) VC_TRY(finish_gather(ctx));
    return VC_OK;
}

// Variable-length all-gather in rank order.  Default (option "gather_compact" = 1): the compact form
// above.  With the option off: counts first (one u64 per rank), then one grouped broadcast of the
// 8-byte records per root straight into its displacement of the gathered buffer.
int vc_allgather(vc_ctx *ctx, uint64_t *counts_out, uint64_t *total_out)
{
    if (!ctx || !total_out) return VC_ERR_ARG;
    if (!ctx->comm) return fail(ctx, VC_ERR_ARG, "vc_comm_init must precede vc_allgather");
    if (!ctx->carved) return fail(ctx, VC_ERR_ARG, "no carve result to gather");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    const int G = ctx->n_ranks;
    if (ctx->gather_compact) return allgather_compact(ctx, counts_out, total_out);
    VC_TRY(finish_gather(ctx));
    if (ctx->sb[ctx->cur].no_records)
        return fail(ctx, VC_ERR_ARG, "last carve ran with VC_FLAG_NO_RECORDS: the record exchange needs records");
    hipStream_t sx = ctx->overlap ? ctx->stream_x : ctx->stream;      // (the exchange stream: every collective of the communicator is queued there)
    uint64_t *d_mine = ctx->d_counts.ptr + G;
    *ctx->h_total = ctx->survivors;
    VC_HIP(ctx, hipEventRecord(ctx->ev[0], sx));
    VC_HIP(ctx, hipMemcpyAsync(d_mine, ctx->h_total, sizeof(uint64_t), hipMemcpyHostToDevice, sx));
    VC_NCCL(ctx, g_rccl.AllGather(d_mine, ctx->d_counts.ptr, 1, ncclUint64, ctx->comm, sx));
    VC_HIP(ctx, hipMemcpyAsync(ctx->h_counts, ctx->d_counts.ptr, sizeof(uint64_t) * G, hipMemcpyDeviceToHost, sx));
    VC_HIP(ctx, hipStreamSynchronize(sx));
    uint64_t total = 0;
    for (int r = 0; r < G; ++r) total += ctx->h_counts[r];
    VC_TRY(ensure(ctx, ctx->d_gathered, (size_t)total));
    StepBuf &cur = ctx->sb[ctx->cur];
    if (!cur.records.ptr) VC_TRY(ensure(ctx, cur.records, 1024));
    VC_NCCL(ctx, g_rccl.GroupStart());
    uint64_t disp = 0;
    for (int r = 0; r < G; ++r) {
        const uint64_t cnt = ctx->h_counts[r];
        if (cnt) {
            ncclResult_t rc = g_rccl.Broadcast(cur.records.ptr, ctx->d_gathered.ptr + disp, cnt, ncclUint64, r,
                                               ctx->comm, sx);
            if (rc != ncclSuccess) {
                g_rccl.GroupEnd();
                return fail(ctx, VC_ERR_RCCL, "ncclBroadcast(root %d): %s", r, g_rccl.GetErrorString(rc));
            }
        }
        disp += cnt;
    }
    VC_NCCL(ctx, g_rccl.GroupEnd());
    VC_HIP(ctx, hipEventRecord(ctx->ev[1], sx));
    VC_HIP(ctx, hipStreamSynchronize(sx));
    VC_HIP(ctx, hipEventElapsedTime(&ctx->tm.gather_ms, ctx->ev[0], ctx->ev[1]));
    ctx->tm.exchange_ms = ctx->tm.gather_ms;
    ctx->tm.gather_ms_sum += ctx->tm.gather_ms;
    ctx->tm.gathers += 1;
    if (counts_out) memcpy(counts_out, ctx->h_counts, sizeof(uint64_t) * G);
    ctx->gathered_total = total;
    ctx->gathered = true;
    *total_out = total;
    return VC_OK;
}

// Max over ranks of one double, through the device (RCCL all-reduce): a barrier and a timing
// reduction for host code that must not load a second ROCm runtime.
int vc_comm_allreduce_max(vc_ctx *ctx, double *inout)
{
    if (!ctx || !inout) return VC_ERR_ARG;
    if (!ctx->comm) return fail(ctx, VC_ERR_ARG, "vc_comm_init must precede vc_comm_allreduce_max");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_TRY(ensure(ctx, ctx->d_scratch, 16));
    double *d = ctx->d_scratch.ptr;
    hipStream_t sx = ctx->overlap ? ctx->stream_x : ctx->stream;      // (every collective of the communicator on the one exchange stream)
    VC_HIP(ctx, hipMemcpyAsync(d, inout, sizeof(double), hipMemcpyHostToDevice, sx));
    VC_NCCL(ctx, g_rccl.AllReduce(d, d + 1, 1, ncclFloat64, ncclMax, ctx->comm, sx));
    VC_HIP(ctx, hipMemcpyAsync(inout, d + 1, sizeof(double), hipMemcpyDeviceToHost, sx));
    VC_HIP(ctx, hipStreamSynchronize(sx));
    return VC_OK;
}

int vc_fetch_gathered(vc_ctx *ctx, uint64_t *records)
{
    if (!ctx || !records) return VC_ERR_ARG;
    if (!ctx->gathered) return fail(ctx, VC_ERR_ARG, "no gathered result: call vc_allgather");
    VC_HIP(ctx, hipSetDevice(ctx->device));
    VC_TRY(finish_gather(ctx));
    if (ctx->gathered_total)
        VC_HIP(ctx, hipMemcpy(records, ctx->d_gathered.ptr, ctx->gathered_total * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return VC_OK;
}

}  // extern "C"
