"""Tuning aid: what the emit costs with and without the colour sample."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
eng.set_grid(1024, 1024, 1024); eng.set_cameras(cams, *masks[0].shape)
eng.upload_masks(masks); eng.build_lut()
for label, cc, frame in (("no colour camera", None, False), ("colour cam, no frame uploaded", 1, False), ("colour cam + frame", 1, True)):
    if frame:
        eng.upload_frame(1, frames[1])
    ts = []
    for it in range(6):
        n = eng.carve(mode="lut", color_cam=cc)
        ts.append(eng.timing()["compact_ms"])
    print("%-32s survivors %d compact med %.4f min %.4f" % (label, n, np.median(ts[1:]), min(ts[1:])), flush=True)
