"""CarveEngine: Python face of one libvoxcarve context (one GPU, one z-slab of the grid).

Array-shaped fast path of the reference's carve step:
voxel_reconstruction.py:35-124 + assignment.py:116-133.  All compute happens in the HIP
library; this class only marshals numpy buffers across the C ABI.
"""
import ctypes

import numpy as np

from . import _lib
from .camera import Camera

# reference voxel_reconstruction.py:35-36 (x_min, x_max, y_min, y_max, z_min, z_max)
DEFAULT_BOUNDS = (-512.0, 1024.0, -1024.0, 1024.0, -2048.0, 512.0)
SCALING_FACTOR = 64          # reference assignment.py:118
COLOR_CAMERA_INDEX = 1       # reference assignment.py:133 uses camera key 2 (1-based)

MODES = {"fused": _lib.VC_MODE_FUSED, "lut": _lib.VC_MODE_LUT}


def _ptr(a, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


class CarveEngine:
    def __init__(self, device=0):
        self._L = _lib.load()
        self._ctx = _lib.c_ctx()
        _lib.check(self._L.vc_create(int(device), ctypes.byref(self._ctx)), None, "vc_create")
        self.device = device
        self.grid = None
        self.slab = None
        self.n_cameras = 0
        self.image_size = None
        self.count = 0

    # -- lifetime -----------------------------------------------------------------
    def close(self):
        if self._ctx:
            self._release_pinned()
            self._L.vc_destroy(self._ctx)
            self._ctx = _lib.c_ctx()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        _lib.check(rc, self._ctx, what)

    # -- geometry -----------------------------------------------------------------
    def set_grid(self, nx, ny, nz, bounds=DEFAULT_BOUNDS):
        b = np.asarray(bounds, dtype=np.float64).reshape(6)
        self._check(self._L.vc_set_grid(self._ctx, nx, ny, nz, _ptr(b, ctypes.c_double)), "vc_set_grid")
        self.grid = (int(nx), int(ny), int(nz))
        self.bounds = tuple(float(v) for v in b)
        self.slab = (0, int(nz))

    def set_slab(self, z0, z1):
        self._check(self._L.vc_set_slab(self._ctx, z0, z1), "vc_set_slab")
        self.slab = (int(z0), int(z1))

    @property
    def n_voxels(self):
        nx, ny, _ = self.grid
        return nx * ny * (self.slab[1] - self.slab[0])

    @property
    def index_base(self):
        nx, ny, _ = self.grid
        return self.slab[0] * nx * ny

    def axes(self):
        nx, ny, nz = self.grid
        xs, ys, zs = np.empty(nx), np.empty(ny), np.empty(nz)
        self._check(self._L.vc_get_axes(self._ctx, _ptr(xs, ctypes.c_double), _ptr(ys, ctypes.c_double),
                                        _ptr(zs, ctypes.c_double)), "vc_get_axes")
        return xs, ys, zs

    # -- cameras ------------------------------------------------------------------
    def set_cameras(self, cameras, H, W):
        cams = [c if isinstance(c, Camera) else Camera(*c) for c in cameras]
        K9 = np.ascontiguousarray([c.K.reshape(9) for c in cams], dtype=np.float64)
        d5 = np.ascontiguousarray([c.dist for c in cams], dtype=np.float64)
        R9 = np.ascontiguousarray([c.R.reshape(9) for c in cams], dtype=np.float64)
        t3 = np.ascontiguousarray([c.tvec for c in cams], dtype=np.float64)
        dp = ctypes.c_double
        self._check(self._L.vc_set_cameras(self._ctx, len(cams), _ptr(K9, dp), _ptr(d5, dp), _ptr(R9, dp),
                                           _ptr(t3, dp), H, W), "vc_set_cameras")
        self.n_cameras = len(cams)
        self._cams = cams
        self.image_size = (int(H), int(W))

    # -- per-frame inputs -----------------------------------------------------------
    def upload_masks(self, masks, slot=0):
        m = np.ascontiguousarray(np.stack([np.asarray(x) for x in masks]), dtype=np.uint8)
        if m.shape != (self.n_cameras,) + self.image_size:
            raise ValueError("masks shape %s, expected %s" % (m.shape, (self.n_cameras,) + self.image_size))
        self._check(self._L.vc_upload_masks(self._ctx, slot, _ptr(m, ctypes.c_uint8)), "vc_upload_masks")

    def touch_masks(self, slot=0):
        """Treat the slot's resident byte masks / images as new input: the next carve re-derives bit masks, block
        grids and camera order from them on the device (no transfer)."""
        self._check(self._L.vc_touch_masks(self._ctx, slot), "vc_touch_masks")

    def set_mask_postfilter(self, open2x2=None, close2x2=None):
        """Per-camera 2x2 MORPH_OPEN / MORPH_CLOSE applied on the device to every following
        upload_masks (tail of the reference's extract_foreground_mask, background_subtraction.py:195-206)."""
        def arr(flags):
            if flags is None:
                return None
            a = np.ascontiguousarray([1 if f else 0 for f in flags], dtype=np.uint8)
            if a.size != self.n_cameras:
                raise ValueError("need one flag per camera")
            return a
        o, c = arr(open2x2), arr(close2x2)
        self._check(self._L.vc_set_mask_postfilter(self._ctx, _ptr(o, ctypes.c_uint8) if o is not None else None,
                                                   _ptr(c, ctypes.c_uint8) if c is not None else None), "vc_set_mask_postfilter")

    # -- the step before the path: data-parallel part of extract_foreground_mask --------------------------------
    def bgr_to_hsv(self, image):
        """cv2.cvtColor(image, cv2.COLOR_BGR2HSV) on uint8 [H,W,3], on the device (background_subtraction.py:155)."""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("image shape %s, expected [H, W, 3]" % (a.shape,))
        out = np.empty_like(a)
        self._check(self._L.vc_bgr_to_hsv(self._ctx, _ptr(a, ctypes.c_uint8), a.shape[0], a.shape[1], _ptr(out, ctypes.c_uint8)), "vc_bgr_to_hsv")
        return out

    def mask_morphology(self, mask, ksize, opening=False, closing=False):
        """cv2.morphologyEx with a ksize x ksize MORPH_RECT element on uint8 [H,W]: MORPH_OPEN, then MORPH_CLOSE, as asked
        (background_subtraction.py:161-168 with ksize 3, :195-203 with ksize 2)."""
        a = np.ascontiguousarray(mask, dtype=np.uint8)
        if a.ndim != 2:
            raise ValueError("mask shape %s, expected [H, W]" % (a.shape,))
        out = np.empty_like(a)
        self._check(self._L.vc_mask_morphology(self._ctx, _ptr(a, ctypes.c_uint8), a.shape[0], a.shape[1], int(ksize), int(bool(opening)),
                                               int(bool(closing)), _ptr(out, ctypes.c_uint8)), "vc_mask_morphology")
        return out

    # ---- the MOG background model (cv2.bgsegm.createBackgroundSubtractorMOG; background_subtraction.py:75-92, :158)
    def mog_create(self, history=200, nmixtures=5, background_ratio=0.7, noise_sigma=0):
        model = ctypes.c_uint32(0)
        self._check(self._L.vc_mog_create(self._ctx, int(history), int(nmixtures), float(background_ratio), float(noise_sigma),
                                          ctypes.byref(model)), "vc_mog_create")
        return model.value

    def mog_apply(self, model, image, learning_rate=-1):
        a = np.ascontiguousarray(image, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("image shape %s, expected [H, W, 3]" % (a.shape,))
        out = np.empty(a.shape[:2], dtype=np.uint8)
        self._check(self._L.vc_mog_apply(self._ctx, int(model), _ptr(a, ctypes.c_uint8), a.shape[0], a.shape[1], float(learning_rate),
                                         _ptr(out, ctypes.c_uint8)), "vc_mog_apply")
        return out

    def mog_state(self, model):
        """(state float32 [8 nmixtures, H W] planes -- plane 8 k + f: field f (sort key, weight, mean[3], var[3]) of component k --,
        (H, W), frames seen)."""
        H, W, K, nf = (ctypes.c_uint32(0) for _ in range(4))
        self._check(self._L.vc_mog_state(self._ctx, int(model), None, 0, ctypes.byref(H), ctypes.byref(W), ctypes.byref(K), ctypes.byref(nf)),
                    "vc_mog_state")
        state = np.zeros((8 * K.value, H.value * W.value), dtype=np.float32)
        if state.size:
            self._check(self._L.vc_mog_state(self._ctx, int(model), _ptr(state, ctypes.c_float), state.size, None, None, None, None), "vc_mog_state")
        return state, (H.value, W.value), nf.value

    def foreground_front(self, model, image, learning_rate=0, opening=False, closing=False, to_hsv=True):
        """BGR -> HSV, the model's apply, 3x3 open / close: extract_foreground_mask up to its contour stage, one call
        (background_subtraction.py:155-168)."""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("image shape %s, expected [H, W, 3]" % (a.shape,))
        out = np.empty(a.shape[:2], dtype=np.uint8)
        self._check(self._L.vc_foreground_front(self._ctx, int(model), _ptr(a, ctypes.c_uint8), a.shape[0], a.shape[1], int(bool(to_hsv)),
                                                float(learning_rate), int(bool(opening)), int(bool(closing)), _ptr(out, ctypes.c_uint8)),
                    "vc_foreground_front")
        return out

    def mog_destroy(self, model):
        self._check(self._L.vc_mog_destroy(self._ctx, int(model)), "vc_mog_destroy")

    def fetch_mask(self, cam, slot=0):
        out = np.empty(self.image_size, dtype=np.uint8)
        self._check(self._L.vc_fetch_mask(self._ctx, slot, cam, _ptr(out, ctypes.c_uint8)), "vc_fetch_mask")
        return out

    def upload_frame(self, cam, bgr, slot=0):
        f = np.ascontiguousarray(bgr, dtype=np.uint8)
        if f.shape != self.image_size + (3,):
            raise ValueError("frame shape %s, expected %s" % (f.shape, self.image_size + (3,)))
        self._check(self._L.vc_upload_frame(self._ctx, slot, cam, _ptr(f, ctypes.c_uint8)), "vc_upload_frame")

    # -- lookup table ---------------------------------------------------------------
    def build_lut(self):
        self._check(self._L.vc_build_lut(self._ctx), "vc_build_lut")

    def fetch_lut(self, cam):
        out = np.empty(self.n_voxels, dtype=np.int32)
        self._check(self._L.vc_fetch_lut(self._ctx, cam, _ptr(out, ctypes.c_int32)), "vc_fetch_lut")
        return out

    # -- lookup-table persistence (reference: the pickled table of assignment.py:12-15; here a .npz, nothing executable) ----
    def _lut_meta(self):
        import hashlib
        h = hashlib.sha256()
        for c in self._cams:
            for a in (c.K, c.dist, c.R, c.tvec):
                h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        return {"format": "voxcarve-lut-1", "grid": list(self.grid), "slab": list(self.slab), "bounds": list(self.bounds),
                "image_size": list(self.image_size), "cameras": self.n_cameras, "cameras_sha256": h.hexdigest(),
                "entry": "int32 pixel offset int(v) * W + int(u) of the voxel's projection, -1 outside the image "
                         "(voxel_reconstruction.py:110-112); voxel order i = iz*nx*ny + ix*ny + iy of the slab"}

    def save_lut(self, path):
        """Writes the packed table of this context (grid, slab, cameras) as <path> (.npz: 'lut' int32 [C, n] + 'meta' JSON)."""
        import json
        lut = np.stack([self.fetch_lut(c) for c in range(self.n_cameras)])
        with open(path, "wb") as f:
            np.savez(f, lut=lut, meta=np.frombuffer(json.dumps(self._lut_meta()).encode(), dtype=np.uint8))

    def load_lut(self, path):
        """Hands a table written by save_lut to the device instead of projecting it again.  The file must have been made
        for exactly this grid, slab, bounds, mask size and these cameras: anything else raises VoxcarveError."""
        import json
        with np.load(path, allow_pickle=False) as z:
            meta = json.loads(bytes(z["meta"]).decode())
            want = self._lut_meta()
            bad = [k for k in want if k != "entry" and meta.get(k) != want[k]]
            if bad:
                raise _lib.VoxcarveError("lookup table %s was made for another configuration (differs in: %s)" % (path, ", ".join(bad)))
            lut = z["lut"]
            if lut.dtype != np.int32 or lut.shape != (self.n_cameras, self.n_voxels):
                raise _lib.VoxcarveError("lookup table %s: array %s %s, expected int32 %s" % (path, lut.dtype, lut.shape, (self.n_cameras, self.n_voxels)))
            hw = int(self.image_size[0]) * int(self.image_size[1])            # (equal to the file's: checked above)
            lo, hi = (int(lut.min()), int(lut.max())) if lut.size else (-1, -1)
            if lo < -1 or hi >= hw:
                raise _lib.VoxcarveError("lookup table %s holds entries outside [-1, H*W) (min %d, max %d): stale or corrupt file" % (path, lo, hi))
            self.upload_lut(lut)

    def upload_lut(self, lut):
        """int32 [C, n] in voxel order (what fetch_lut gives out per camera); adopted once all cameras are in."""
        lut = np.ascontiguousarray(lut, dtype=np.int32)
        if lut.shape != (self.n_cameras, self.n_voxels):
            raise ValueError("lut shape %s, expected %s" % (lut.shape, (self.n_cameras, self.n_voxels)))
        for c in range(self.n_cameras):
            self._check(self._L.vc_upload_lut(self._ctx, c, _ptr(lut[c], ctypes.c_int32)), "vc_upload_lut")

    def project(self, cam, points):
        p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        uv = np.empty((p.shape[0], 2), dtype=np.float64)
        self._check(self._L.vc_project(self._ctx, cam, _ptr(p, ctypes.c_double), p.shape[0],
                                       _ptr(uv, ctypes.c_double)), "vc_project")
        return uv

    # -- hot path -------------------------------------------------------------------
    def carve(self, slot=0, min_views=None, color_cam=COLOR_CAMERA_INDEX, mode="fused", viewmask=False, records=True):
        """Runs the carve; returns the survivor count (records stay on the device).  records=False keeps
        only the count and the occupancy words (multi-GPU ranks: allgather() / expand_entries() make the list)."""
        n = ctypes.c_uint64(0)
        mv = self.n_cameras if min_views is None else int(min_views)
        cc = -1 if color_cam is None else int(color_cam)
        flags = (_lib.VC_FLAG_VIEWMASK if viewmask else 0) | (0 if records else _lib.VC_FLAG_NO_RECORDS)
        self._check(self._L.vc_carve(self._ctx, slot, mv, cc, MODES[mode], flags, ctypes.byref(n)), "vc_carve")
        self.count = int(n.value)
        return self.count

    def carve_begin(self, slot=0, min_views=None, color_cam=COLOR_CAMERA_INDEX, mode="fused", viewmask=False,
                    records=True):
        """Enqueue a carve step without waiting (at most two in flight), so the device has the next
        step queued while the host collects this one.  carve_end() completes the oldest step."""
        mv = self.n_cameras if min_views is None else int(min_views)
        cc = -1 if color_cam is None else int(color_cam)
        flags = (_lib.VC_FLAG_VIEWMASK if viewmask else 0) | (0 if records else _lib.VC_FLAG_NO_RECORDS)
        self._check(self._L.vc_carve_begin(self._ctx, slot, mv, cc, MODES[mode], flags), "vc_carve_begin")

    def carve_end(self):
        n = ctypes.c_uint64(0)
        self._check(self._L.vc_carve_end(self._ctx, ctypes.byref(n)), "vc_carve_end")
        self.count = int(n.value)
        return self.count

    def fetch(self):
        """(idx u32 [S] ascending global linear index, rgb u8 [S,3], seen bool [S])."""
        S = self.count
        idx = np.empty(S, dtype=np.uint32)
        rgb = np.empty((S, 3), dtype=np.uint8)
        seen = np.empty(S, dtype=np.uint8)
        self._check(self._L.vc_fetch(self._ctx, _ptr(idx, ctypes.c_uint32), _ptr(rgb, ctypes.c_uint8),
                                     _ptr(seen, ctypes.c_uint8)), "vc_fetch")
        return idx, rgb, seen.astype(bool)

    def fetch_records(self, pinned=False):
        """u64 records of the last carve.  pinned=True returns a view of a page-locked buffer
        owned by the engine (valid until the next pinned fetch): PCIe-rate read-back."""
        if not pinned:
            rec = np.empty(self.count, dtype=np.uint64)
            self._check(self._L.vc_fetch_records(self._ctx, _ptr(rec, ctypes.c_uint64)), "vc_fetch_records")
            return rec
        need = max(self.count, 1) * 8
        if getattr(self, "_pin_bytes", 0) < need:
            self._release_pinned()
            ptr = ctypes.c_void_p()
            grow = (need + need // 4 + 7) // 8 * 8
            self._check(self._L.vc_host_alloc(self._ctx, grow, ctypes.byref(ptr)), "vc_host_alloc")
            self._pin_ptr, self._pin_bytes = ptr, grow
            self._pin_arr = np.frombuffer((ctypes.c_uint8 * grow).from_address(ptr.value), dtype=np.uint64)
        out = self._pin_arr[:self.count]
        self._check(self._L.vc_fetch_records(self._ctx, _ptr(out, ctypes.c_uint64)), "vc_fetch_records")
        return out

    def _release_pinned(self):
        if getattr(self, "_pin_bytes", 0):
            self._pin_arr = None
            self._L.vc_host_free(self._ctx, self._pin_ptr)
            self._pin_bytes = 0

    def fetch_viewmask(self):
        vm = np.empty(self.n_voxels, dtype=np.uint16)
        self._check(self._L.vc_fetch_viewmask(self._ctx, _ptr(vm, ctypes.c_uint16)), "vc_fetch_viewmask")
        return vm

    def fetch_occupancy(self):
        """Dense survivor bits of the slab: bool [n] (slab-local voxel order)."""
        n = self.n_voxels
        raw = np.empty(((n + 63) // 64) * 8, dtype=np.uint8)
        self._check(self._L.vc_fetch_occupancy(self._ctx, _ptr(raw, ctypes.c_uint8)), "vc_fetch_occupancy")
        return np.unpackbits(raw, bitorder="little")[:n].astype(bool)

    def marching_cubes(self, volume=None, level=0.0, axes="reference"):
        """Triangle mesh of an ON/OFF volume on the device -> (verts float32 [V, 3], faces uint32 [F, 3]).
        volume: 3-D boolean array (what the reference hands to skimage.measure.marching_cubes, voxel_reconstruction.py:141);
        None = the occupancy of the last carve, reshaped as the reference does it (axes="reference": (nx, ny, nz) over the
        voxel order, assignment.py:144) or on its geometric axes (axes="grid": (nz, nx, ny), i.e. vertex = (iz, ix, iy))."""
        nv, nf = ctypes.c_uint64(0), ctypes.c_uint64(0)
        if volume is None:
            nx, ny, _ = self.grid
            nzl = self.slab[1] - self.slab[0]
            dims = (nx, ny, nzl) if axes == "reference" else (nzl, nx, ny)
            bits = None
        else:
            vol = np.ascontiguousarray(volume).astype(bool)
            if vol.ndim != 3:
                raise ValueError("volume must be 3-D")
            dims = vol.shape
            packed = np.packbits(vol.reshape(-1), bitorder="little")
            bits = _ptr(packed, ctypes.c_uint8)
        self._check(self._L.vc_marching_cubes(self._ctx, bits, dims[0], dims[1], dims[2], float(level), ctypes.byref(nv), ctypes.byref(nf)),
                    "vc_marching_cubes")
        verts = np.empty((int(nv.value), 3), dtype=np.float32)
        faces = np.empty((int(nf.value), 3), dtype=np.uint32)
        self._check(self._L.vc_fetch_mesh(self._ctx, _ptr(verts, ctypes.c_float), _ptr(faces, ctypes.c_uint32)), "vc_fetch_mesh")
        return verts, faces

    def set_option(self, name, value):
        """Launch-geometry tuning knobs (never change results); see vc_set_option."""
        self._check(self._L.vc_set_option(self._ctx, name.encode(), int(value)), "vc_set_option")

    def debug_counters(self):
        out = np.zeros(8, dtype=np.uint64)
        self._check(self._L.vc_debug_counters(self._ctx, _ptr(out, ctypes.c_uint64)), "vc_debug_counters")
        return {"bricks_listed": int(out[0]), "bricks_live": int(out[1]), "bricks_full": int(out[2]), "bricks": int(out[3]),
                "columns_listed": int(out[4]), "words_undecided": int(out[5])}

    def synchronize(self):
        self._check(self._L.vc_synchronize(self._ctx), "vc_synchronize")

    def timing(self, reset=False):
        t = _lib.VcTiming()
        self._check(self._L.vc_timing(self._ctx, ctypes.byref(t)), "vc_timing")
        if reset:
            self._check(self._L.vc_timing_reset(self._ctx), "vc_timing_reset")
        out = {name: getattr(t, name) for name, _ in _lib.VcTiming._fields_ if name not in ("kernel_ms_sum", "kernel_launches", "work")}
        out["kernels"] = {k: {"ms_sum": float(t.kernel_ms_sum[i]), "launches": int(t.kernel_launches[i])}
                          for i, k in enumerate(_lib.KERNEL_KINDS) if t.kernel_launches[i]}
        out["work"] = {k: int(t.work[i]) for i, k in enumerate(_lib.WORK_KINDS)}
        return out

    # -- multi-GPU -------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        L = _lib.load()
        buf = (ctypes.c_uint8 * _lib.VC_UNIQUE_ID_BYTES)()
        _lib.check(L.vc_comm_unique_id(buf), None, "vc_comm_unique_id")
        return bytes(buf)

    def comm_init(self, n_ranks, rank, uid):
        import os
        import sys
        if "torch" in sys.modules and os.environ.get("VOXCARVE_ALLOW_TORCH") != "1":
            # Measured on the GPU box: a torch wheel bundles its own libhsa-runtime64 / librccl; once it is
            # loaded, RCCL resolves HSA from that uninitialised copy and ncclCommInitRank fails with
            # "no ROCm-capable device is detected".  Exchange the unique id without a framework
            # (slabs.file_rendezvous) or set VOXCARVE_ALLOW_TORCH=1 if your torch uses the system ROCm.
            raise _lib.VoxcarveError("comm_init in a process that imported torch: its bundled ROCm runtime breaks RCCL "
                                     "(see voxcarve.slabs.file_rendezvous); set VOXCARVE_ALLOW_TORCH=1 to try anyway")
        buf = (ctypes.c_uint8 * _lib.VC_UNIQUE_ID_BYTES).from_buffer_copy(uid)
        self._check(self._L.vc_comm_init(self._ctx, n_ranks, rank, buf), "vc_comm_init")
        self.n_ranks, self.rank = n_ranks, rank

    def comm_destroy(self):
        self._check(self._L.vc_comm_destroy(self._ctx), "vc_comm_destroy")

    def allgather(self):
        """RCCL all-gather of all ranks' survivor records; returns (counts per rank, total)."""
        counts = np.zeros(getattr(self, "n_ranks", 1), dtype=np.uint64)
        total = ctypes.c_uint64(0)
        self._check(self._L.vc_allgather(self._ctx, _ptr(counts, ctypes.c_uint64), ctypes.byref(total)),
                    "vc_allgather")
        self.gathered_total = int(total.value)
        return counts, self.gathered_total

    def comm_max(self, value=0.0):
        """Max of `value` over all ranks through RCCL; with the default it is a barrier."""
        v = ctypes.c_double(float(value))
        self._check(self._L.vc_comm_allreduce_max(self._ctx, ctypes.byref(v)), "vc_comm_allreduce_max")
        return v.value

    def fetch_gathered(self):
        rec = np.empty(self.gathered_total, dtype=np.uint64)
        self._check(self._L.vc_fetch_gathered(self._ctx, _ptr(rec, ctypes.c_uint64)), "vc_fetch_gathered")
        return rec

    # -- compact exchange form (host-side transports, tests) ---------------------------
    def pack_entries(self):
        """The last carve's non-zero occupancy words as u64 [M,2] = {bits, global index of bit 0}, ascending."""
        m = ctypes.c_uint64(0)
        self._check(self._L.vc_pack_entries(self._ctx, ctypes.byref(m)), "vc_pack_entries")
        ent = np.empty((int(m.value), 2), dtype=np.uint64)
        self._check(self._L.vc_fetch_entries(self._ctx, _ptr(ent, ctypes.c_uint64)), "vc_fetch_entries")
        return ent

    def expand_entries(self, entries):
        """All ranks' entries in rank order -> ordered records of the whole grid on this device (read them with
        fetch_gathered()), coloured like this engine's last carve.  Returns the survivor count."""
        ent = np.ascontiguousarray(entries, dtype=np.uint64).reshape(-1, 2)
        total = ctypes.c_uint64(0)
        self._check(self._L.vc_expand_entries(self._ctx, _ptr(ent, ctypes.c_uint64), ent.shape[0],
                                              ctypes.byref(total)), "vc_expand_entries")
        self.gathered_total = int(total.value)
        return self.gathered_total


# -- record / viewer helpers (host arithmetic of assignment.py:127-133) ----------------
def unpack_records(rec):
    """u64 records -> (idx u32, rgb u8 [S,3], seen bool)."""
    rec = np.ascontiguousarray(rec, dtype=np.uint64)
    b = rec.view(np.uint8).reshape(-1, 8)
    return rec.astype(np.uint32), b[:, 4:7].copy(), (b[:, 7] & 1).astype(bool)


def voxel_keys(idx, grid, axes):
    """tuple(map(int, voxel)) of voxel_reconstruction.py:84: truncated coordinates int64 [S,3]."""
    nx, ny, _ = grid
    xs, ys, zs = axes
    idx = np.asarray(idx, dtype=np.int64)
    iy = idx % ny
    t = idx // ny
    return np.stack([np.trunc(xs[t % nx]), np.trunc(ys[iy]), np.trunc(zs[t // nx])], axis=1).astype(np.int64)


def viewer_positions(keys, scaling_factor=SCALING_FACTOR):
    """assignment.py:127-130: x = vx/64, y = -(vz/64), z = vy/64 -> float32 [S,3] (as mesh.py:82 casts)."""
    k = np.asarray(keys, dtype=np.int64)
    pos = np.stack([k[:, 0] / scaling_factor, -(k[:, 2] / scaling_factor), k[:, 1] / scaling_factor], axis=1)
    return pos.astype(np.float32)


def viewer_colors(rgb):
    """assignment.py:133: BGR[::-1] / 255.0 -> float32 [S,3] (as mesh.py:88 casts)."""
    return (np.asarray(rgb, dtype=np.uint8) / 255.0).astype(np.float32)
