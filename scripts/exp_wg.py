"""cull / brick_words with 1, 2 or 4 waves per workgroup (dbg 262144 / 524288 / 0): same records first, then the step time."""
import os, sys, hashlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
for grid in ((64, 256, 64), (1024, 1024, 1024)):
    eng.set_grid(*grid); eng.set_cameras(cams, *masks[0].shape)
    eng.upload_masks(masks); eng.upload_frame(1, frames[1]); eng.build_lut()
    ref = None
    for dbg in (0,):          # (the workgroup-shape switches were an experiment build: DESIGN section 8)
        eng.set_option("dbg", dbg)
        for rep in range(3):
            eng.touch_masks(0)
            n = eng.carve(mode="lut", color_cam=1)
            d = hashlib.sha256(eng.fetch_records().tobytes()).hexdigest()
            if ref is None:
                ref = (n, d)
            assert (n, d) == ref, (grid, dbg, rep, n, ref[0])
    print(grid, "same records with 4, 2, 1 waves per workgroup:", ref[0], flush=True)
eng.set_option("dbg", 0)
