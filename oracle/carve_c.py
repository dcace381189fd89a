"""ctypes front for oracle/carve_ref.c.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libcarve_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcarve_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        dp, u8p = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint8)
        L.vco_axis.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_uint32, dp]
        L.vco_axis.restype = None
        L.vco_project.argtypes = [dp, ctypes.c_uint64, dp, dp, dp, dp, dp]
        L.vco_project.restype = None
        L.vco_carve.argtypes = [ctypes.c_uint32] * 3 + [dp, ctypes.c_uint32, dp, dp, dp, dp,
                                ctypes.c_uint32, ctypes.c_uint32, u8p, u8p,
                                ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64,
                                ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_int32),
                                ctypes.POINTER(ctypes.c_uint32), u8p, ctypes.c_uint64, ctypes.c_int]
        L.vco_carve.restype = ctypes.c_int64
        L.vco_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def pack_cams(cams):
    """list of (K, dist, R, t) -> contiguous K9, dist5, R9, t3 float64 arrays."""
    K9 = np.ascontiguousarray([np.asarray(c[0], np.float64).reshape(9) for c in cams])
    d5 = np.ascontiguousarray([np.asarray(c[1], np.float64).reshape(-1)[:5] for c in cams])
    R9 = np.ascontiguousarray([np.asarray(c[2], np.float64).reshape(9) for c in cams])
    t3 = np.ascontiguousarray([np.asarray(c[3], np.float64).reshape(3) for c in cams])
    return K9, d5, R9, t3


def axis(lo, hi, n):
    out = np.empty(n, np.float64)
    lib().vco_axis(lo, hi, n, _dp(out))
    return out


def project(points, cam):
    K9, d5, R9, t3 = pack_cams([cam])
    pts = np.ascontiguousarray(points, np.float64)
    uv = np.empty((pts.shape[0], 2), np.float64)
    lib().vco_project(_dp(pts), pts.shape[0], _dp(K9), _dp(d5), _dp(R9), _dp(t3), _dp(uv))
    return uv


def carve(nx, ny, nz, cams, masks, frames=None, bounds=(-512.0, 1024.0, -1024.0, 1024.0, -2048.0, 512.0),
          min_views=None, color_cam=1, index_range=None, want_viewmask=False, want_lut=False,
          threads=0, cap=None):
    C = len(cams)
    K9, d5, R9, t3 = pack_cams(cams)
    m = np.ascontiguousarray(np.stack(masks), np.uint8)
    H, W = m.shape[1:]
    N = nx * ny * nz
    i0, i1 = (0, N) if index_range is None else index_range
    n = i1 - i0
    b = np.asarray(bounds, np.float64)
    vm = np.empty(n, np.uint16) if want_viewmask else None
    lut = np.empty((C, n), np.int32) if want_lut else None
    frame = np.ascontiguousarray(frames[color_cam], np.uint8) if frames is not None else None
    u8p = ctypes.POINTER(ctypes.c_uint8)

    def run(capacity, idx, bgr):
        return lib().vco_carve(
            nx, ny, nz, _dp(b), C, _dp(K9), _dp(d5), _dp(R9), _dp(t3), H, W,
            m.ctypes.data_as(u8p), frame.ctypes.data_as(u8p) if frame is not None else None,
            C if min_views is None else min_views, color_cam, i0, i1,
            vm.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)) if vm is not None else None,
            lut.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if lut is not None else None,
            idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)) if idx is not None else None,
            bgr.ctypes.data_as(u8p) if bgr is not None else None, capacity, threads)

    capacity = n if cap is None else cap
    idx = np.empty(capacity, np.uint32)
    bgr = np.empty((capacity, 3), np.uint8) if frame is not None else None
    S = run(capacity, idx, bgr)
    if S < 0:
        raise ValueError("vco_carve: bad arguments")
    k = min(S, capacity)
    out = {"idx": idx[:k].copy(), "count": int(S)}
    if bgr is not None:
        out["bgr"] = bgr[:k].copy()
    if vm is not None:
        out["viewmask"] = vm
    if lut is not None:
        out["offsets"] = lut
    return out
