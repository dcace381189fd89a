"""Experiment: per-frame preparation + carve phases for a workload (real | config5), synchronous calls with all events on."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="real"); ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--mode", default="lut"); ap.add_argument("opts", nargs="*")
a = ap.parse_args()
grid, cams, masks, frames, cc, dm, text = bench.make_workload(a)
eng = voxcarve.CarveEngine(0)
eng.set_grid(*grid); eng.set_cameras(cams, *masks[0].shape)
for o in a.opts:
    k, v = o.split("="); eng.set_option(k, int(v))
eng.upload_masks(masks); eng.upload_frame(cc, frames[cc])
if a.mode == "lut":
    eng.build_lut()
eng.set_option("overlap", 0)
for _ in range(3):
    eng.touch_masks(0); eng.carve(mode=a.mode, color_cam=cc)
eng.timing(reset=True)
for _ in range(20):
    eng.touch_masks(0); eng.carve(mode=a.mode, color_cam=cc)
tm = eng.timing()
print("%s %s %s: prep %.4f carve %.4f emit %.4f ms, survivors %d, %s" % (a.workload, a.mode, a.opts, tm["prep_ms_sum"] / max(1, tm["preps_timed"]),
      tm["carve_ms_sum"] / tm["carve_launches"], tm["emit_ms_sum"] / max(1, tm["emit_launches"]), eng.count, eng.debug_counters()))
