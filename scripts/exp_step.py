"""Tuning aid: steady-state step time (two steps in flight, as bench.py runs them), 1024^3 x 4 (or grid=nx,ny,nz), options as k=v."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
grid = (1024, 1024, 1024)
for opt in sys.argv[1:]:
    if opt.startswith("grid="):
        grid = tuple(int(x) for x in opt[5:].split(","))
eng.set_grid(*grid); eng.set_cameras(cams, *masks[0].shape)
mode = "lut"
for opt in sys.argv[1:]:
    k, v = opt.split("=")
    if k == "grid":
        continue
    if k == "mode":
        mode = v
    else:
        eng.set_option(k, int(v))
for s in range(4):
    eng.upload_masks([np.roll(m, 3 * s, axis=1) for m in masks], slot=s)
    eng.upload_frame(1, np.roll(frames[1], 3 * s, axis=1), slot=s)
if mode == "lut":
    eng.build_lut()
depth = int(os.environ.get("DEPTH", "2"))
def run(n):
    pending = 0
    for i in range(n):
        eng.carve_begin(slot=i % 4, mode=mode)
        pending += 1
        if pending == depth:
            eng.carve_end(); pending -= 1
    while pending:
        eng.carve_end(); pending -= 1
    eng.synchronize()
run(30)
best = 1e9
for rep in range(5):
    t0 = time.perf_counter(); run(200); dt = (time.perf_counter() - t0) / 200 * 1e3
    best = min(best, dt)
print("%s %s: step %.4f ms (best of 5 x 200)" % (mode, " ".join(sys.argv[1:]), best), flush=True)
