"""Tuning aid: table-free hierarchical kernel time (1024^3 x 4, real fixtures)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng.set_grid(G, G, G); eng.set_cameras(cams, *masks[0].shape)
for opt in sys.argv[2:]:
    k, v = opt.split("=")
    eng.set_option(k, int(v))
eng.upload_masks(masks); eng.upload_frame(1, frames[1])
ts = []
for it in range(12):
    n = eng.carve(mode="fused")
    ts.append(eng.timing()["carve_ms"])
print("survivors %d carve med %.4f min %.4f" % (n, np.median(ts[2:]), min(ts[2:])), flush=True)
