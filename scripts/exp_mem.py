"""Device memory footprint of a context by mode (hipMemGetInfo)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxcarve, fixtures_util as fx
hip = ctypes.CDLL("libamdhip64.so")
def free_gb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 1e9
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
for label, grid, lut in (("1024^3 fused only", (1024, 1024, 1024), False), ("1024^3 lut + fused", (1024, 1024, 1024), True),
                         ("2048x2048x1023 fused only", (2048, 2048, 1023), False)):
    eng = voxcarve.CarveEngine(0)
    a = free_gb()
    eng.set_grid(*grid); eng.set_cameras(cams, *masks[0].shape)
    eng.upload_masks(masks); eng.upload_frame(1, frames[1])
    if lut:
        eng.build_lut(); eng.carve(mode="lut")
    n = eng.carve(mode="fused")
    b = free_gb()
    eng.close()
    print("%-28s survivors %d: %.2f GB in use, %.2f GB after close" % (label, n, a - b, a - free_gb()), flush=True)
