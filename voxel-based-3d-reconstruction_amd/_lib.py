"""ctypes binding of libvoxcarve.so (C ABI: include/voxcarve.h).

The library is the only compute path: if it is missing, or there is no gfx950
device, loading / context creation raises -- nothing here falls back to the CPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvoxcarve.so")

VC_OK = 0
VC_MODE_FUSED = 0
VC_MODE_LUT = 1
VC_FLAG_VIEWMASK = 1
VC_FLAG_NO_RECORDS = 2
VC_MAX_CAMERAS = 16
VC_UNIQUE_ID_BYTES = 128

STATUS_NAMES = {0: "VC_OK", -1: "VC_ERR_ARG", -2: "VC_ERR_HIP", -3: "VC_ERR_RCCL",
                -4: "VC_ERR_OOM", -5: "VC_ERR_NODEV"}

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u16p = ctypes.POINTER(ctypes.c_uint16)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_ctx = ctypes.c_void_p


VC_KERNEL_KINDS, VC_WORK_KINDS = 12, 8
KERNEL_KINDS = ("k_prep_pack", "k_prep_grid", "k_cull_bricks", "k_brick_words", "k_voxel_words", "k_assemble", "k_scan_groups",
                "k_finish_scan", "k_emit", "one_launch_carve", "k_cull", "k_count_groups")
WORK_KINDS = ("word_boxes", "table_entries", "projections", "emit_projections", "brick_boxes")


class VcTiming(ctypes.Structure):
    _fields_ = [("carve_ms", ctypes.c_float), ("compact_ms", ctypes.c_float),
                ("gather_ms", ctypes.c_float), ("lut_ms", ctypes.c_float),
                ("h2d_ms", ctypes.c_float), ("voxels", ctypes.c_uint64),
                ("survivors", ctypes.c_uint64), ("carve_launches", ctypes.c_uint32),
                ("carve_ms_sum", ctypes.c_float), ("first_ms", ctypes.c_float),
                ("first_ms_sum", ctypes.c_float), ("exchange_ms", ctypes.c_float),
                ("gather_ms_sum", ctypes.c_float), ("gathers", ctypes.c_uint32),
                ("prep_ms", ctypes.c_float), ("prep_ms_sum", ctypes.c_float), ("preps", ctypes.c_uint32),
                ("preps_timed", ctypes.c_uint32), ("emit_ms", ctypes.c_float), ("emit_ms_sum", ctypes.c_float),
                ("emit_launches", ctypes.c_uint32),
                ("kernel_ms_sum", ctypes.c_float * VC_KERNEL_KINDS), ("kernel_launches", ctypes.c_uint32 * VC_KERNEL_KINDS),
                ("work", ctypes.c_uint64 * VC_WORK_KINDS)]


# name -> (restype, argtypes); every symbol include/voxcarve.h declares.
SIGNATURES = {
    "vc_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "vc_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_ctx)]),
    "vc_destroy": (ctypes.c_int, [c_ctx]),
    "vc_last_error": (ctypes.c_char_p, [c_ctx]),
    "vc_synchronize": (ctypes.c_int, [c_ctx]),
    "vc_set_grid": (ctypes.c_int, [c_ctx, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, c_f64p]),
    "vc_set_slab": (ctypes.c_int, [c_ctx, ctypes.c_uint32, ctypes.c_uint32]),
    "vc_get_axes": (ctypes.c_int, [c_ctx, c_f64p, c_f64p, c_f64p]),
    "vc_set_cameras": (ctypes.c_int, [c_ctx, ctypes.c_uint32, c_f64p, c_f64p, c_f64p, c_f64p,
                                      ctypes.c_uint32, ctypes.c_uint32]),
    "vc_upload_masks": (ctypes.c_int, [c_ctx, ctypes.c_uint32, c_u8p]),
    "vc_touch_masks": (ctypes.c_int, [c_ctx, ctypes.c_uint32]),
    "vc_set_mask_postfilter": (ctypes.c_int, [c_ctx, c_u8p, c_u8p]),
    "vc_fetch_mask": (ctypes.c_int, [c_ctx, ctypes.c_uint32, ctypes.c_uint32, c_u8p]),
    "vc_bgr_to_hsv": (ctypes.c_int, [c_ctx, c_u8p, ctypes.c_uint32, ctypes.c_uint32, c_u8p]),
    "vc_mask_morphology": (ctypes.c_int, [c_ctx, c_u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, c_u8p]),
    "vc_mog_create": (ctypes.c_int, [c_ctx, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_uint32)]),
    "vc_mog_apply": (ctypes.c_int, [c_ctx, ctypes.c_uint32, c_u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, c_u8p]),
    "vc_mog_state": (ctypes.c_int, [c_ctx, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float), ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32),
                                    ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
    "vc_mog_destroy": (ctypes.c_int, [c_ctx, ctypes.c_uint32]),
    "vc_foreground_front": (ctypes.c_int, [c_ctx, ctypes.c_uint32, c_u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_double,
                                           ctypes.c_int, ctypes.c_int, c_u8p]),
    "vc_upload_frame": (ctypes.c_int, [c_ctx, ctypes.c_uint32, ctypes.c_uint32, c_u8p]),
    "vc_build_lut": (ctypes.c_int, [c_ctx]),
    "vc_fetch_lut": (ctypes.c_int, [c_ctx, ctypes.c_uint32, c_i32p]),
    "vc_upload_lut": (ctypes.c_int, [c_ctx, ctypes.c_uint32, c_i32p]),
    "vc_project": (ctypes.c_int, [c_ctx, ctypes.c_uint32, c_f64p, ctypes.c_uint64, c_f64p]),
    "vc_carve": (ctypes.c_int, [c_ctx, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                ctypes.c_uint32, c_u64p]),
    "vc_carve_begin": (ctypes.c_int, [c_ctx, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_uint32]),
    "vc_carve_end": (ctypes.c_int, [c_ctx, c_u64p]),
    "vc_fetch": (ctypes.c_int, [c_ctx, c_u32p, c_u8p, c_u8p]),
    "vc_fetch_records": (ctypes.c_int, [c_ctx, c_u64p]),
    "vc_host_alloc": (ctypes.c_int, [c_ctx, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]),
    "vc_host_free": (ctypes.c_int, [c_ctx, ctypes.c_void_p]),
    "vc_fetch_viewmask": (ctypes.c_int, [c_ctx, c_u16p]),
    "vc_fetch_occupancy": (ctypes.c_int, [c_ctx, c_u8p]),
    "vc_marching_cubes": (ctypes.c_int, [c_ctx, c_u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float, c_u64p, c_u64p]),
    "vc_fetch_mesh": (ctypes.c_int, [c_ctx, ctypes.POINTER(ctypes.c_float), c_u32p]),
    "vc_set_option": (ctypes.c_int, [c_ctx, ctypes.c_char_p, ctypes.c_int]),
    "vc_timing_struct_size": (ctypes.c_uint32, []),
    "vc_timing": (ctypes.c_int, [c_ctx, ctypes.POINTER(VcTiming)]),
    "vc_debug_counters": (ctypes.c_int, [c_ctx, c_u64p]),
    "vc_timing_reset": (ctypes.c_int, [c_ctx]),
    "vc_comm_unique_id": (ctypes.c_int, [c_u8p]),
    "vc_comm_init": (ctypes.c_int, [c_ctx, ctypes.c_int, ctypes.c_int, c_u8p]),
    "vc_comm_destroy": (ctypes.c_int, [c_ctx]),
    "vc_allgather": (ctypes.c_int, [c_ctx, c_u64p, c_u64p]),
    "vc_fetch_gathered": (ctypes.c_int, [c_ctx, c_u64p]),
    "vc_pack_entries": (ctypes.c_int, [c_ctx, c_u64p]),
    "vc_fetch_entries": (ctypes.c_int, [c_ctx, c_u64p]),
    "vc_expand_entries": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_uint64, c_u64p]),
    "vc_comm_allreduce_max": (ctypes.c_int, [c_ctx, c_f64p]),
}

_lib = None


class VoxcarveError(RuntimeError):
    pass


def load():
    """Load libvoxcarve.so and bind every entry point; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("VOXCARVE_LIB", LIB_PATH)          # an alternative build of the same library (sanitizer runs)
    if not os.path.exists(path):
        raise VoxcarveError(
            "libvoxcarve.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    # multi-process GPU work on this platform needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise);
    # must be in the environment before the HIP runtime initialises
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = ABI drift, let it surface
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.vc_timing_struct_size() != ctypes.sizeof(VcTiming):
        raise VoxcarveError("libvoxcarve.so at %s was built with a vc_timing_t of %d bytes, this binding mirrors one of %d: rebuild the library"
                            % (path, lib.vc_timing_struct_size(), ctypes.sizeof(VcTiming)))
    _lib = lib
    return lib


def check(rc, ctx=None, what=""):
    if rc == VC_OK:
        return
    msg = load().vc_last_error(ctx)
    raise VoxcarveError("%s failed: %s (%s)" % (what or "voxcarve call", STATUS_NAMES.get(rc, rc),
                                                msg.decode() if msg else ""))
