"""Tuning aid: cost floors -- all-background masks (coarse pass + empty emit launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
eng.set_grid(1024, 1024, 1024); eng.set_cameras(cams, *masks[0].shape)
eng.upload_masks([np.zeros_like(m) for m in masks], slot=0)
eng.upload_masks(masks, slot=1)
eng.upload_frame(1, frames[1], slot=0); eng.upload_frame(1, frames[1], slot=1)
eng.build_lut()
for slot, name in ((0, "all-background"), (1, "real masks")):
    for mode, hier in (("lut", 1), ("lut", 0), ("fused", 1)):
        eng.set_option("lut_hier", hier)
        ts = []
        for it in range(6):
            n = eng.carve(slot=slot, mode=mode)
            t = eng.timing(); ts.append((t["carve_ms"], t["compact_ms"]))
        a = np.array(ts[1:])
        print(name, mode, "hier", hier, "survivors", n, "carve med %.4f compact med %.4f" % (np.median(a[:, 0]), np.median(a[:, 1])), flush=True)
