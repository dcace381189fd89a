"""Tuning aid: record expansion time (scan + list + expansion kernels), 1024^3 x 4, both modes; options as k=v."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
eng.set_grid(1024, 1024, 1024); eng.set_cameras(cams, *masks[0].shape)
for opt in sys.argv[1:]:
    k, v = opt.split("=")
    eng.set_option(k, int(v))
eng.upload_masks(masks); eng.upload_frame(1, frames[1]); eng.build_lut()
for mode in ("lut", "fused"):
    ts = []
    for it in range(8):
        n = eng.carve(mode=mode)
        ts.append(eng.timing()["compact_ms"])
    print("%-5s survivors %d compact med %.4f min %.4f" % (mode, n, np.median(ts[2:]), min(ts[2:])), flush=True)
