"""Tuning aid: host-side time of each call of a pipelined step on the 1-rank communicator path (where does the host wait?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
eng.set_grid(1024, 1024, 1024); eng.set_cameras(cams, *masks[0].shape)
for s in range(8):
    eng.upload_masks([np.roll(m, 3 * s, axis=1) for m in masks], slot=s)
    eng.upload_frame(1, np.roll(frames[1], 3 * s, axis=1), slot=s)
eng.build_lut()
eng.comm_init(1, 0, voxcarve.CarveEngine.comm_unique_id())
eng.set_option("gather_sync", 0)
T = []
def begin(i):
    t0 = time.perf_counter(); eng.touch_masks(i % 8); eng.carve_begin(slot=i % 8, mode="lut", records=False); T.append(("begin", i, time.perf_counter() - t0))
def finish(i):
    t0 = time.perf_counter(); n = eng.carve_end(); t1 = time.perf_counter(); eng.allgather(); T.append(("end+gather", i, t1 - t0, time.perf_counter() - t1))
begin(0)
for i in range(1, 40):
    begin(i); finish(i - 1)
finish(39); eng.synchronize()
T.clear()
t0 = time.perf_counter()
begin(0)
for i in range(1, 200):
    begin(i); finish(i - 1)
finish(199); eng.synchronize()
print("step %.4f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
for row in T[300:320]:
    print(row[0], row[1], " ".join("%.1f us" % (x * 1e6) for x in row[2:]))
