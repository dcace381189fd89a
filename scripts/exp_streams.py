"""How the carve and the expansion streams share the chip: the bench's step (every step prepares its frame set, three steps
in flight) under different stream settings, alternating on ONE device (devices of the pool differ by +-5 %).
usage: python scripts/exp_streams.py [mode=lut|fused] [workload=real|config5] "stream_priority=0" "stream_priority=1" "reserve_cus=2" ...
Each quoted argument is one setting (comma-separated k=v pairs applied on top of the defaults)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx

mode, workload, settings = "lut", "real", []
for a in sys.argv[1:]:
    if a.startswith("mode="):
        mode = a[5:]
    elif a.startswith("workload="):
        workload = a[9:]
    else:
        settings.append(a)
if workload == "config5":
    from voxcarve import synthetic
    H, W, C = 1080, 1920, 16
    cams = synthetic.ring_cameras(C, H, W)
    masks = synthetic.ellipsoid_masks(cams, H, W)
    frames = synthetic.random_frames(C, H, W)
    grid = (512, 512, 512)
else:
    cams, masks = fx.golden_cameras(), fx.golden_masks()
    frames = fx.synthetic_frames(4, *masks[0].shape)
    grid = (1024, 1024, 1024)
eng = voxcarve.CarveEngine(0)
eng.set_grid(*grid); eng.set_cameras(cams, *masks[0].shape)
NS = 8
for s in range(NS):
    eng.upload_masks([np.roll(m, 3 * s, axis=1) for m in masks], slot=s)
    eng.upload_frame(1, np.roll(frames[1], 3 * s, axis=1), slot=s)
if mode == "lut":
    eng.build_lut()
DEFAULTS = {"stream_priority": 1, "reserve_cus": 0, "emit_waves_per_cu": 256, "overlap": 1, "event_scope": 1, "dbg": 0, "launch_events": 1, "kernel_events": 0, "timing_detail": 0, "voxel_batches": 0}


AHEAD = 0                                  # frame sets prepared this many steps before the step that carves them (the slots are there)


def run(n, depth=3):
    pending = 0
    for a in range(AHEAD):
        eng.touch_masks(a % NS)
    for i in range(n):
        eng.touch_masks((i + AHEAD) % NS)
        eng.carve_begin(slot=i % NS, mode=mode)
        pending += 1
        if pending == depth:
            eng.carve_end(); pending -= 1
    while pending:
        eng.carve_end(); pending -= 1
    eng.synchronize()


def apply(text):
    global AHEAD
    opts = dict(DEFAULTS)
    AHEAD = 0
    for kv in text.split(","):
        if kv:
            k, v = kv.split("=")
            if k == "ahead":
                AHEAD = int(v)
            else:
                opts[k] = int(v)
    for k, v in opts.items():
        eng.set_option(k, v)


t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    run(20)
best = {s: 1e9 for s in settings}
for rep in range(4):
    for s in settings:
        apply(s)
        run(20)
        eng.timing(reset=True)
        t0 = time.perf_counter(); run(200); dt = (time.perf_counter() - t0) / 200 * 1e3
        tm = eng.timing()
        emit = tm["emit_ms_sum"] / max(1, tm["emit_launches"])
        best[s] = min(best[s], dt)
        print("rep %d %-44s step %.4f ms  emit %.4f ms" % (rep, s, dt, emit), flush=True)
        if rep == 3 and tm["kernels"]:
            print("      " + "  ".join("%s %.1f" % (k[2:], v["ms_sum"] / v["launches"] * 1e3) for k, v in tm["kernels"].items()), " us;  work per step:",
                  {k: v // 220 for k, v in tm["work"].items() if v}, flush=True)
for s in settings:
    print("BEST %-44s %.4f ms" % (s, best[s]), flush=True)
