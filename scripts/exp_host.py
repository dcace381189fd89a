"""Host time per step of the bench's loop (touch_masks + carve_begin, and carve_end apart): is the enqueueing thread the limit?
usage: python scripts/exp_host.py [k=v,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx

cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
eng.set_grid(1024, 1024, 1024); eng.set_cameras(cams, *masks[0].shape)
NS = 8
for s in range(NS):
    eng.upload_masks([np.roll(m, 3 * s, axis=1) for m in masks], slot=s)
    eng.upload_frame(1, np.roll(frames[1], 3 * s, axis=1), slot=s)
eng.build_lut()
for a in sys.argv[1:]:
    for kv in a.split(","):
        k, v = kv.split("="); eng.set_option(k, int(v))


def run(n, depth=3):
    tb = te = 0.0
    pending = 0
    for i in range(n):
        t0 = time.perf_counter()
        eng.touch_masks(i % NS)
        eng.carve_begin(slot=i % NS, mode="lut")
        t1 = time.perf_counter()
        tb += t1 - t0
        pending += 1
        if pending == depth:
            eng.carve_end(); pending -= 1
            te += time.perf_counter() - t1
    while pending:
        eng.carve_end(); pending -= 1
    eng.synchronize()
    return tb / n * 1e6, te / n * 1e6


run(200)
for rep in range(3):
    t0 = time.perf_counter(); b, e = run(400); dt = (time.perf_counter() - t0) / 400 * 1e6
    print("step %.1f us: enqueue (touch_masks + carve_begin) %.1f us, carve_end (mostly waiting) %.1f us" % (dt, b, e), flush=True)
