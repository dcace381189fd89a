"""Tuning aid: steady-state step time of BASELINE config 5 (512^3 x 16 cameras x 1080p, salt noise), options as k=v;
prints the records' digest so that variants can be checked against each other."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve
from voxcarve import synthetic
H, W, C = 1080, 1920, 16
cams = synthetic.ring_cameras(C, H, W)
masks = synthetic.ellipsoid_masks(cams, H, W)
frames = synthetic.random_frames(C, H, W)
eng = voxcarve.CarveEngine(0)
eng.set_grid(512, 512, 512); eng.set_cameras(cams, H, W)
mode = "lut"
fresh = 1
for opt in sys.argv[1:]:
    k, v = opt.split("=")
    if k == "mode": mode = v
    elif k == "fresh": fresh = int(v)
    else: eng.set_option(k, int(v))
for s in range(2):
    eng.upload_masks(masks, slot=s)
    eng.upload_frame(1, frames[1], slot=s)
eng.build_lut()
n = eng.carve(slot=0, mode=mode, color_cam=1)
rec = eng.fetch_records()
eng.debug_counters()
n = eng.carve(slot=0, mode=mode, color_cam=1)
print(eng.debug_counters(), flush=True)
print("survivors %d digest %s" % (n, hashlib.sha256(np.ascontiguousarray(rec).tobytes()).hexdigest()[:16]), flush=True)
def run(k):
    eng.carve_begin(slot=0, mode=mode, color_cam=1)
    for i in range(1, k):
        if fresh: eng.touch_masks(i % 2)
        eng.carve_begin(slot=i % 2, mode=mode, color_cam=1)
        eng.carve_end()
    eng.carve_end()
    eng.synchronize()
run(20)
best = 1e9
for rep in range(4):
    t0 = time.perf_counter(); run(100); dt = (time.perf_counter() - t0) / 100 * 1e3
    best = min(best, dt)
print("%s %s: step %.4f ms (best of 4 x 100)" % (mode, " ".join(sys.argv[1:]), best), flush=True)
