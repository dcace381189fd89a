"""Stability: many steps, many contexts; device memory must come back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx

_hip = __import__("ctypes").CDLL("libamdhip64.so")

def used_mb():
    """Device memory in use (hipMemGetInfo; rocm-smi's figure lags behind frees by seconds)."""
    import ctypes
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    _hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return (t.value - f.value) / 1e6

cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
base = used_mb()
for rep in range(6):
    eng = voxcarve.CarveEngine(0)
    g = (256, 512, 1024)[rep % 3]
    eng.set_grid(g, g, g); eng.set_cameras(cams, *masks[0].shape)
    eng.upload_masks(masks); eng.upload_frame(1, frames[1]); eng.build_lut()
    first = None
    t0 = time.perf_counter()
    for i in range(600):
        n = eng.carve(mode=("lut", "fused")[i & 1])
        if first is None:
            first = n
        assert n == first
    dt = time.perf_counter() - t0
    mid = used_mb()
    eng.close()
    print("rep %d grid %d^3: 600 steps %.2f s, survivors stable at %d, vram in use %.0f MB -> after close %.0f MB (start %.0f)" %
          (rep, g, dt, first, mid, used_mb(), base), flush=True)
