/* voxcarve.h -- C ABI of libvoxcarve.so, the MI355X (gfx950) visual-hull carve engine.
 *
 * Drop-in boundary for ONE path of ChristosP1/Voxel-Based-3D-Reconstruction: the
 * per-voxel x per-camera projection-and-mask test behind set_voxel_positions().
 * The reference is pure Python; a maintainer binds this library with ctypes (stub in
 * INTEGRATION.md).  Each entry point names the reference interface it replaces
 * (paths relative to the reference root).
 *
 * Conventions: every function returns 0 (VC_OK) or a negative vc_status; the message
 * of the last failure is vc_last_error(ctx).  A context owns one HIP device + stream
 * and all device buffers; it is NOT thread-safe (the reference calls the path from one
 * thread, executable.py:182-188).  Host buffers belong to the caller.  An empty
 * result is count 0, not an error (reference returns [], []).
 *
 * Voxel numbering (voxel_reconstruction.py:52-57): linear index
 *     i = iz*nx*ny + ix*ny + iy,   centre = (xs[ix], ys[iy], zs[iz]),
 * axes = np.linspace(lo, hi, n).  Survivor lists are ascending in i, which is the
 * order the reference's dicts yield (assignment.py:121-133).
 */
#ifndef VOXCARVE_H
#define VOXCARVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vc_ctx vc_ctx;

typedef enum {
    VC_OK = 0,
    VC_ERR_ARG = -1,    /* bad argument / call order */
    VC_ERR_HIP = -2,    /* HIP runtime error (message has hipGetErrorString) */
    VC_ERR_RCCL = -3,   /* RCCL missing or failed */
    VC_ERR_OOM = -4,    /* device allocation failed */
    VC_ERR_NODEV = -5   /* no usable GPU: there is NO CPU fallback */
} vc_status;

typedef enum {
    VC_MODE_FUSED = 0,  /* project in-kernel (fp64), nothing precomputed            */
    VC_MODE_LUT = 1     /* stream the packed int32 LUT built by vc_build_lut()      */
} vc_mode;

enum {
    VC_FLAG_VIEWMASK = 1u,   /* also keep the per-voxel camera bitmask (compat dicts) */
    VC_FLAG_NO_RECORDS = 2u  /* count + occupancy only: the records are produced by vc_allgather /
                                vc_expand_entries (a rank of a multi-GPU job never reads its own slab's list).
                                With a communicator attached (vc_comm_init) such a step is a COLLECTIVE call:
                                it also packs the slab's words and all-gathers the counts, so every rank
                                must issue the same sequence of them. */
};

#define VC_MAX_CAMERAS 16
#define VC_UNIQUE_ID_BYTES 128

typedef enum {
    VC_K_PREP_PACK = 0, VC_K_PREP_GRID, VC_K_CULL_BRICKS, VC_K_BRICK_WORDS, VC_K_VOXEL_WORDS, VC_K_ASSEMBLE,
    VC_K_SCAN_GROUPS, VC_K_FINISH_SCAN, VC_K_EMIT, VC_K_CARVE_ONE_LAUNCH /* k_lut_refine, k_carve_fused*, k_carve_generic, k_lut_first */,
    VC_K_CULL /* in front of a one-launch kernel */, VC_K_COUNT_GROUPS
} vc_kernel_kind;
#define VC_KERNEL_KINDS 12
enum {
    VC_WORK_WORD_BOXES = 0,   /* 8-byte word boxes k_brick_words read (listed bricks x 64 words x cameras asked)           */
    VC_WORK_TABLE_ENTRIES,    /* 4-byte table entries the per-voxel level read (VC_MODE_LUT)                                */
    VC_WORK_PROJECTIONS,      /* float64 projections the per-voxel level did (VC_MODE_FUSED)                                */
    VC_WORK_EMIT_PROJECTIONS, /* float64 projections the record expansion did (VC_MODE_FUSED without the colour table)      */
    VC_WORK_BRICK_BOXES       /* 8-byte brick boxes k_cull_bricks read                                                      */
};
#define VC_WORK_KINDS 8

typedef struct {
    float carve_ms;     /* the carve kernels alone (HIP events on the context's stream).  carve_ms, first_ms, compact_ms
                           and prep_ms are measured for vc_carve calls and, with option timing_detail = 1, for
                           vc_carve_begin steps: an event between two kernels costs the stream ~10 us, so pipelined
                           steps record only the events they need anyway (emit_ms comes from those) */
    float compact_ms;   /* scan + emit kernels                                         */
    float gather_ms;    /* RCCL all-gather (vc_allgather)                              */
    float lut_ms;       /* last vc_build_lut                                           */
    float h2d_ms;       /* last vc_upload_masks / vc_upload_frame incl. bit-packing    */
    uint64_t voxels;    /* voxels of this rank's slab                                  */
    uint64_t survivors; /* survivors of the last carve (this rank)                     */
    uint32_t carve_launches; /* carve kernel launches since vc_timing_reset            */
    float carve_ms_sum; /* summed carve kernel time since vc_timing_reset              */
    float first_ms;     /* VC_MODE_LUT: the first-camera streaming kernel of the last carve */
    float first_ms_sum; /* summed since vc_timing_reset                                 */
    float exchange_ms;  /* vc_allgather, compact form: pack + RCCL part of gather_ms         */
    float gather_ms_sum; /* summed since vc_timing_reset                                */
    uint32_t gathers;   /* vc_allgather calls since vc_timing_reset                      */
    float prep_ms;      /* per-frame preparation queued in front of the last carve (bit-pack, boxes, grids, camera order);
                           measured only with option timing_detail = 1 (one more event on the carve stream) */
    float prep_ms_sum;  /* summed since vc_timing_reset                                  */
    uint32_t preps;     /* carve steps that had to prepare their frame set since vc_timing_reset */
    uint32_t preps_timed; /* ... of which prep_ms_sum holds the time                     */
    float emit_ms;      /* record expansion of the last step: the launch's own begin .. end (the events ride on the launch) */
    float emit_ms_sum;  /* summed since vc_timing_reset                                  */
    uint32_t emit_launches;
    /* Option timing_detail = 1: every kernel of a step carries its own begin / end events (on its launch: no extra packet on
     * the stream) and the kernels count the work they do.  Index = vc_kernel_kind; summed since vc_timing_reset. */
    float kernel_ms_sum[VC_KERNEL_KINDS];
    uint32_t kernel_launches[VC_KERNEL_KINDS];
    /* work[VC_WORK_*]: what the kernels of those steps actually touched (counted on the device, lanes that really asked);
     * valid when no step is in flight */
    uint64_t work[VC_WORK_KINDS];
} vc_timing_t;

/* ---- lifetime ------------------------------------------------------------------ */
int vc_device_count(int *n_out);
int vc_create(int device, vc_ctx **out);
int vc_destroy(vc_ctx *ctx);
const char *vc_last_error(const vc_ctx *ctx);      /* ctx may be NULL: last create error */
int vc_synchronize(vc_ctx *ctx);

/* ---- geometry: replaces create_voxel_volume, voxel_reconstruction.py:35-59 ------ */
/* bounds = {x_min,x_max,y_min,y_max,z_min,z_max}.  No point array is materialised:
 * kernels regenerate the np.linspace coordinates from the index. */
int vc_set_grid(vc_ctx *ctx, uint32_t nx, uint32_t ny, uint32_t nz, const double bounds[6]);
/* This rank's block of the grid: iz in [z0, z1) (multi-GPU z-slab split).  Default all. */
int vc_set_slab(vc_ctx *ctx, uint32_t z0, uint32_t z1);
/* The three np.linspace axes as the device uses them (tests; out arrays of nx, ny, nz). */
int vc_get_axes(vc_ctx *ctx, double *xs, double *ys, double *zs);

/* ---- cameras: replaces load_config_info, voxel_reconstruction.py:10-32 ---------- */
/* K9: [C,9] row-major camera matrices; dist5: [C,5] (k1,k2,p1,p2,k3); R9: [C,9] rotation
 * matrices (host does Rodrigues so fixtures pin R); t3: [C,3]; H, W: mask size. */
int vc_set_cameras(vc_ctx *ctx, uint32_t n_cameras, const double *K9, const double *dist5,
                   const double *R9, const double *t3, uint32_t H, uint32_t W);

/* ---- per-frame inputs: the fg_masks / images arguments of ------------------------
 *      update_visible_voxels_and_extract_colors, voxel_reconstruction.py:89 --------- */
/* masks: u8 [C,H,W], foreground where > 0 (line 112).  slot selects one of the resident frame sets (0..63, created
 * on first use).  ASYNCHRONOUS: the bytes are copied to a page-locked staging buffer and from there to the device
 * on an upload stream of their own, so the call returns at once and the copy runs beside the carve in flight; the
 * caller's buffer is free when the call returns.  Everything derived from the bytes (bit masks, foreground boxes,
 * cropped block grids, camera visiting order, images in the records' byte order) is made ON THE DEVICE by two kernels queued in front of
 * the first vc_carve / vc_carve_begin that uses the slot -- no host round trip anywhere. */
int vc_upload_masks(vc_ctx *ctx, uint32_t slot, const uint8_t *masks);
/* The byte masks and images resident in `slot` are to be taken as NEW input: the next carve on the slot derives
 * everything from them again (what a producer that writes the masks on the device, or a benchmark that wants every
 * step to pay for its own preparation, calls instead of uploading the same bytes again). */
int vc_touch_masks(vc_ctx *ctx, uint32_t slot);
/* Tail of extract_foreground_mask on the device (background_subtraction.py:195-206): per camera,
 * optional 2x2 MORPH_OPEN then 2x2 MORPH_CLOSE applied to the byte masks of every following
 * vc_upload_masks, before the final > 0 binarisation.  Arrays of C flags, NULL = none. */
int vc_set_mask_postfilter(vc_ctx *ctx, const uint8_t *open2x2, const uint8_t *close2x2);
/* The device's binarised mask of one camera as u8 [H,W] in {0,255} (tests). */
int vc_fetch_mask(vc_ctx *ctx, uint32_t slot, uint32_t cam, uint8_t *out);
/* bgr: u8 [H,W,3] image of camera cam (0-based) for colour sampling (lines 119-122).  Asynchronous like
 * vc_upload_masks. */
int vc_upload_frame(vc_ctx *ctx, uint32_t slot, uint32_t cam, const uint8_t *bgr);

/* ---- lookup table: replaces create_lookup_table, voxel_reconstruction.py:62-86 --- */
/* Projects this rank's slab once into int32 [C][n]: int(y)*W + int(x), or -1 when the
 * float coordinates fail the bounds test of line 110.  Needed by VC_MODE_LUT only. */
int vc_build_lut(vc_ctx *ctx);
int vc_fetch_lut(vc_ctx *ctx, uint32_t cam, int32_t *out);   /* n entries, voxel order */
/* The way back: replaces the pickled lookup table the reference can load instead of rebuilding it (load_lookup_table,
 * assignment.py:12-15).  One camera's n entries in voxel order, exactly what vc_fetch_lut gives out; once all cameras of
 * the context have been handed in the table is adopted (tile order, word and brick boxes reduced from it) and VC_MODE_LUT
 * runs on it.  Entries outside [-1, H*W) count as -1.  The host side (CarveEngine.save_lut / load_lut) wraps the table
 * with grid, slab, bounds, mask size and a digest of the camera parameters and refuses a file that does not match. */
int vc_upload_lut(vc_ctx *ctx, uint32_t cam, const int32_t *lut);
/* Device projection of arbitrary points with camera cam: uv = [n,2] float64 (tests). */
int vc_project(vc_ctx *ctx, uint32_t cam, const double *xyz, uint64_t n, double *uv);

/* ---- the hot path: replaces update_visible_voxels_and_extract_colors (:89-124) ----
 *      plus the selection loop of set_voxel_positions, assignment.py:116-133 -------- */
/* Keeps voxels seen by >= min_views cameras (reference: 4 of 4).  color_cam is the
 * 0-based camera whose image colours the survivors (reference key 2 -> index 1), or
 * -1 for none.  Leaves the ordered survivor records on the device; *n_out = count. */
int vc_carve(vc_ctx *ctx, uint32_t slot, uint32_t min_views, int color_cam, int mode,
             uint32_t flags, uint64_t *n_out);
/* The same step split in two so that step i+1 is queued on the device before the host collects
 * step i (no idle gap between steps).  At most THREE steps may be in flight (with two, the host cannot queue step i + 1
 * before it has collected step i - 1, whose record expansion ends about when the carve of step i does: the carve stream would
 * idle for the host's round trip); vc_carve_end completes the OLDEST one, whose records are then what vc_fetch_* /
 * vc_allgather read.  The steps rotate through three sets of result buffers: a vc_carve_begin that is queued into the set
 * holding the collected result (the third one after it was issued) takes it away -- fetch before that call; afterwards the
 * vc_fetch_* functions fail with VC_ERR_ARG until the next vc_carve_end.
 * min_views > n_cameras is legal and yields the empty result (as the reference's threshold test would). */
int vc_carve_begin(vc_ctx *ctx, uint32_t slot, uint32_t min_views, int color_cam, int mode, uint32_t flags);
int vc_carve_end(vc_ctx *ctx, uint64_t *n_out);
/* Survivors of the last carve: idx u32 [S] (global linear index, ascending), rgb u8 [S,3]
 * (RGB order, i.e. the reference's BGR[::-1]) and seen u8 [S] (1 if the colour camera
 * sees the voxel -- the reference raises KeyError when it does not).  Any may be NULL. */
int vc_fetch(vc_ctx *ctx, uint32_t *idx, uint8_t *rgb, uint8_t *seen);
/* Raw 8-byte records {u32 idx, u8 r, g, b, seen} of the last carve (S of them). */
int vc_fetch_records(vc_ctx *ctx, uint64_t *records);
/* Page-locked host buffers for the fetch destinations (PCIe-rate read-back). */
int vc_host_alloc(vc_ctx *ctx, uint64_t bytes, void **out);
int vc_host_free(vc_ctx *ctx, void *ptr);
/* Per-voxel camera bitmask u16 [n] of the last carve run with VC_FLAG_VIEWMASK. */
int vc_fetch_viewmask(vc_ctx *ctx, uint16_t *viewmask);
/* Dense occupancy of the last carve: ceil(n/64)*8 bytes, bit (j & 7) of byte j >> 3 for
 * slab-local voxel j (consumer shape of assignment.py:143-146). */
int vc_fetch_occupancy(vc_ctx *ctx, uint8_t *bits);

/* ---- the step before the path, its data-parallel part ------------------------------------------------------
 * Front half of extract_foreground_mask, background_subtraction.py:153-168.  Host buffers in and out (findContours / fill
 * :171-193 sits between the pre- and the post-filter and stays with cv2 on the CPU).
 * vc_bgr_to_hsv: replaces cv2.cvtColor(image, cv2.COLOR_BGR2HSV) (:155) on uint8 [H,W,3] -- OpenCV's 8-bit fixed-point
 * conversion (H in 0..179).  vc_mask_morphology: replaces cv2.morphologyEx(mask, MORPH_OPEN / MORPH_CLOSE,
 * getStructuringElement(MORPH_RECT, (ksize, ksize))) on uint8 [H,W], opening first when both flags are set: ksize 3 = the
 * pre-filter (:161-168), ksize 2 = the post-filter (:195-203; the carve path applies that one itself on upload, see
 * vc_set_mask_postfilter).  Parity with cv2 is unpinned (oracle/foreground_np.py restates OpenCV's published code). */
int vc_bgr_to_hsv(vc_ctx *ctx, const uint8_t *bgr, uint32_t H, uint32_t W, uint8_t *hsv);
int vc_mask_morphology(vc_ctx *ctx, const uint8_t *mask, uint32_t H, uint32_t W, uint32_t ksize, int open, int close, uint8_t *out);
/* The background model between them: cv2.bgsegm.createBackgroundSubtractorMOG(history, nmixtures, backgroundRatio, noiseSigma)
 * (background_subtraction.py:75-76; assignment.py:79 trains one per camera) and its apply(image, None, learningRate)
 * (:91 training, :158 inference with learning rate 0) on uint8 [H,W,3] images -> uint8 [H,W] {0, 255}.  The model lives on the
 * device (8 floats per mixture and pixel); it starts over on its first frame, on a learning rate >= 1 and when the image size
 * changes; a negative learning rate means 1 / min(frames seen, history), as in OpenCV.  Non-positive constructor arguments select
 * OpenCV's defaults (history 200, 5 mixtures (at most 8), backgroundRatio 0.95 when not given, noiseSigma 15).  vc_mog_state
 * copies the model out ([8 nmixtures][H W] float planes: plane 8 k + f = field f of component k; f: 0 sort key, 1 weight,
 * 2..4 mean, 5..7 variance; state may be null to ask for the sizes only) -- tests and persistence.  Restated from the published
 * algorithm of opencv_contrib's bgsegm module (bgfg_gaussmix.cpp); parity with cv2 unpinned (oracle/mog_np.py). */
#define VC_MAX_MOG_MODELS 64
int vc_mog_create(vc_ctx *ctx, int history, int nmixtures, double background_ratio, double noise_sigma, uint32_t *model);
int vc_mog_apply(vc_ctx *ctx, uint32_t model, const uint8_t *image, uint32_t H, uint32_t W, double learning_rate, uint8_t *fgmask);
int vc_mog_state(vc_ctx *ctx, uint32_t model, float *state, uint64_t capacity, uint32_t *H, uint32_t *W, uint32_t *nmixtures, uint32_t *nframes);
int vc_mog_destroy(vc_ctx *ctx, uint32_t model);
/* Everything of extract_foreground_mask in front of the contour stage in one call, one copy each way: BGR -> HSV where to_hsv
 * (:155; the reference always does), the model's apply with `learning_rate` (:158), the 3x3 opening / closing where asked
 * (:161-168).  bgr uint8 [H,W,3] in, the model's mask uint8 [H,W] out. */
int vc_foreground_front(vc_ctx *ctx, uint32_t model, const uint8_t *bgr, uint32_t H, uint32_t W, int to_hsv, double learning_rate,
                        int open, int close, uint8_t *mask);

/* ---- the step after the path: marching cubes over the dense ON/OFF volume -------------------------------
 * Replaces skimage.measure.marching_cubes(voxels_status, 0) of plot_marching_cubes, voxel_reconstruction.py:127-163, whose
 * input the reference builds as the statuses in voxel order reshaped to (width, height*2, depth) (assignment.py:143-146).
 * volume_bits: d0*d1*d2 bits, element i = bit (i & 7) of byte i >> 3, C order (axis 2 fastest); NULL = the occupancy of
 * the last carve of this context, viewed as a (d0, d1, d2) array over the voxel index (d0*d1*d2 must equal the slab's
 * voxel count: (nx, ny, nz) is literally the reference's reshape, (nz, nx, ny) the geometric axes).  Vertices lie on the
 * cube edges between an ON and an OFF element at off + level * (on - off), 0 <= level < 1 (the reference passes 0), in
 * index coordinates (axis 0, 1, 2); faces index them, oriented from ON to OFF.  Classic table-driven marching cubes; the
 * table is generated (csrc/mc_table.h), NOT scikit-image's Lewiner variant: parity with skimage is unpinned. */
int vc_marching_cubes(vc_ctx *ctx, const uint8_t *volume_bits, uint32_t d0, uint32_t d1, uint32_t d2, float level,
                      uint64_t *n_verts, uint64_t *n_faces);
/* verts: float [n_verts][3], faces: u32 [n_faces][3] of the last vc_marching_cubes; either may be NULL. */
int vc_fetch_mesh(vc_ctx *ctx, float *verts, uint32_t *faces);

/* Tuning knobs: which of the equivalent kernels runs and with what launch geometry; NEVER changes results
 * (tests/test_gpu_parity.py runs every family against the oracle).  Defaults are the measured best on MI355X.
 *   kernel choice   lut_hier (1)  hierarchical lookup-table kernel, 0 = stream the table (k_lut_first + refine)
 *                   lut_tile, fused_tile (1)  words of 4 x-rows x 16 y where nx % 4 == 0 and ny % 64 == 0
 *                   bricks (1)  ny in {256, 512, 1024, 2048, 4096}: the brick pipeline (whole 16^3-voxel bricks decided from their pixel
 *                                  boxes, flat lists of bricks / undecided words / columns, one launch per level) instead of
 *                                  the one-launch hierarchical kernels
 *                   cull (1)  the one-launch kernels on tile words skip whole bricks too (k_cull); 0 also switches `bricks` off
 *                   fused_hier (1), fused_boxes (1), fused_f32box (1)  table-free kernel: word rejection; boxes read /
 *                                  bounded on the fly in float32 / float64 intervals
 *                   fused_color_table (1)  table-free carve, survivors coloured from the colour camera's table (4 B per
 *                                  voxel of the whole grid, ONE camera, projected at the first step that wants it) instead of
 *                                  by projecting each of them again; 0: no table of any kind (the expansion is then FP64-bound)
 *                   refine_pair (1), reorder (1)  two cameras per round trip; most selective camera first
 *                   voxel_pairs (0)  per-voxel level of the brick pipeline: 0 = two cameras per round trip up to 4 cameras, one
 *                                  above; 1 = always two; 2 = always one
 *                   voxel_batches (0 = 8 for table look-ups with two cameras per round, else 1)  batches of 8 undecided words a wave of
 *                                  the per-voxel level takes one after the other
 *                   emit_lanes (1), emit_busy (1: grids >= 64 M voxels, 2: always, 0: never)  record expansion form
 *                   force_generic (0)  one thread per voxel everywhere (also env VOXCARVE_FORCE_GENERIC=1)
 *   frame sets      grid_lds_kb (0 = 16; frame sets above 2 MB of mask bits: what their uncropped grids need at the finest block
 *                                  that fits 148 KB, brick pipeline, or 64 KB, other kernels; explicit values above 64 are
 *                                  clamped to 64 for those), grid_min_shift (1)  LDS budget / finest block of the cropped
 *                                  block grids (read when a frame set is next prepared)
 *   launch shape    hier_blocks_per_cu (48), emit_waves_per_cu (256), first_kv (1), first_blocks_per_cu (3),
 *                   refine_b (8), refine_blocks_per_cu (8), fused_blocks_per_cu (8)
 *   streams         overlap (1)  scan + record expansion of a step on a second stream, beside the next step's carve
 *                                  (single stream while a communicator is attached)
 *   experiments     dbg (0)  bit 0: skip the per-voxel level (undecided words count as alive), bit 1: skip the word level
 *                                  too -- WRONG results on purpose, to time the levels apart (scripts/exp_bricks.py); bit 2:
 *                                  no word-level tests, every word of a listed brick goes to the per-voxel level (right results)
 *                                  bit 13: the brick level lists every brick without testing (what it does by itself once a step has
 *                                  listed nine bricks in ten), bit 14: wide frame sets without the survivors' compaction at the word
 *                                  level -- right results, other paths (tests)
 *   timing          timing_detail (0)  1: vc_carve_begin steps record the events around preparation and carve kernels too
 *                                  (vc_carve always does; see vc_timing_t), every kernel carries begin / end events on its own
 *                                  launch (kernel_ms_sum) and counts its work (vc_timing_t::work)
 *                   kernel_events (0)  1: only the per-launch begin / end events (kernel_ms_sum), nothing else changes
 *   streams         stream_priority (1)  carve + preparation streams at the highest queue priority, the expansion stream at
 *                                  the lowest (the expansion fills every wave slot; the carve chain is a row of short launches
 *                                  that would queue behind it); launch_events (1)  the events the streams exchange ride on
 *                                  the launches in front of them; event_scope (1)  those events release to the device only;
 *                                  reserve_cus (0)  k compute units per XCD kept out of the expansion stream's CU mask
 *   multi-GPU       gather_compact (1)  exchange occupancy words instead of records;
 *                   gather_sync (1)  0: vc_allgather returns once its work is queued (two gathers may be in flight; a read-back,
 *                                  vc_timing or vc_synchronize waits for them; vc_fetch_gathered returns the last one's list)
 * Unknown names or out-of-range values return VC_ERR_ARG. */
int vc_set_option(vc_ctx *ctx, const char *name, int value);
int vc_timing(vc_ctx *ctx, vc_timing_t *out);
/* sizeof(vc_timing_t) as the library was built: a binding checks its own mirror of the struct against it before the first call. */
uint32_t vc_timing_struct_size(void);
/* Diagnostics of the last brick-pipeline carve (scripts/, DESIGN figures): out[0] bricks listed for a look at their words,
 * out[1] bricks that may hold survivors, out[2] bricks whose voxels all survive, out[3] bricks of the slab, out[4] brick
 * columns listed, out[5] tile words that took the per-voxel test; the rest 0. */
int vc_debug_counters(vc_ctx *ctx, uint64_t out[8]);
int vc_timing_reset(vc_ctx *ctx);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (no reference counterpart) ---- */
int vc_comm_unique_id(uint8_t out[VC_UNIQUE_ID_BYTES]);
int vc_comm_init(vc_ctx *ctx, int n_ranks, int rank, const uint8_t uid[VC_UNIQUE_ID_BYTES]);
int vc_comm_destroy(vc_ctx *ctx);
/* All-gather of every rank's survivor records in rank (= z-slab = index) order.
 * counts_out (NULL ok): n_ranks entries.  *total_out = global survivor count.
 * What crosses xGMI is the compact form below (option "gather_compact", default 1): each rank's
 * non-zero occupancy words; every rank expands all of them into the full record list itself, taking
 * colours from its own copy of the colour camera's table over the whole grid (VC_MODE_LUT: 4 B per
 * voxel of the whole grid, built at the first call) or by re-projection (VC_MODE_FUSED).  With the
 * option off the 8-byte records themselves are exchanged. */
int vc_allgather(vc_ctx *ctx, uint64_t *counts_out, uint64_t *total_out);
/* The compact form for host-side transports (and tests): entries = pairs of u64 {occupancy bits of one
 * 64-voxel word, global linear index of its bit 0}, non-zero words only, ascending.
 * vc_pack_entries packs the last carve's slab; vc_fetch_entries copies 2*n u64 out;
 * vc_expand_entries takes the concatenation of all ranks' entries in rank order and leaves the ordered
 * records of the whole grid where vc_fetch_gathered reads them, coloured like the last carve of THIS
 * context (same mode, colour camera and frame set). */
int vc_pack_entries(vc_ctx *ctx, uint64_t *n_entries_out);
int vc_fetch_entries(vc_ctx *ctx, uint64_t *entries);
int vc_expand_entries(vc_ctx *ctx, const uint64_t *entries, uint64_t n_entries, uint64_t *total_out);
int vc_fetch_gathered(vc_ctx *ctx, uint64_t *records);
/* Max of one double over all ranks via RCCL (doubles as a barrier for host code). */
int vc_comm_allreduce_max(vc_ctx *ctx, double *inout);

#ifdef __cplusplus
}
#endif
#endif /* VOXCARVE_H */
