"""Experiment: record expansion time with / without the colour look-up (1024^3 x 4, LUT mode, one stream, synchronous calls)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
eng.set_grid(1024, 1024, 1024); eng.set_cameras(cams, *masks[0].shape)
eng.upload_masks(masks); eng.upload_frame(1, frames[1])
eng.build_lut(); eng.set_option("overlap", 0)
for o in sys.argv[1:]:
    k, v = o.split("="); eng.set_option(k, int(v))
for label, kw in (("colour from table + frame", dict(color_cam=1)), ("no colour camera", dict(color_cam=None))):
    for _ in range(3): eng.carve(mode="lut", **kw)
    eng.timing(reset=True)
    for _ in range(20): eng.carve(mode="lut", **kw)
    tm = eng.timing()
    print("%-28s emit %.4f ms (carve %.4f)" % (label, tm["emit_ms_sum"] / tm["emit_launches"], tm["carve_ms_sum"] / tm["carve_launches"]))
