"""Would whole steps on alternating streams (no cross-stream event on the critical path) beat carve || expansion of neighbouring
steps?  Emulated with K independent contexts on one device, each running its steps on ONE stream (overlap=0), the host dealing
steps round-robin.  usage: python scripts/exp_lanes.py [mode=lut|fused] [workload=config5] K[,K..] [depth per context]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx

mode = "lut"
args = [a for a in sys.argv[1:]]
workload = "real"
for a in list(args):
    if a.startswith("mode="):
        mode = a[5:]; args.remove(a)
    elif a.startswith("workload="):
        workload = a[9:]; args.remove(a)
Ks = [int(x) for x in (args[0] if args else "1,2,3").split(",")]
depth = int(args[1]) if len(args) > 1 else 1
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
NS = 8
engs = []
for k in range(max(Ks)):
    e = voxcarve.CarveEngine(0)
    e.set_grid(*grid); e.set_cameras(cams, *masks[0].shape)
    for s in range(NS):
        e.upload_masks([np.roll(m, 3 * s, axis=1) for m in masks], slot=s)
        e.upload_frame(1, np.roll(frames[1], 3 * s, axis=1), slot=s)
    if mode == "lut":
        e.build_lut()
    e.set_option("overlap", int(os.environ.get("LANES_OVERLAP", "0")))
    engs.append(e)


def run(K, n):
    pend = []
    for i in range(n):
        e = engs[i % K]
        e.touch_masks((i // K) % NS)
        e.carve_begin(slot=(i // K) % NS, mode=mode)
        pend.append(e)
        if len(pend) == K * depth:
            pend.pop(0).carve_end()
    while pend:
        pend.pop(0).carve_end()
    for e in engs[:K]:
        e.synchronize()


for K in Ks:
    run(K, 60)
for rep in range(3):
    for K in Ks:
        run(K, 30)
        t0 = time.perf_counter(); run(K, 300); dt = (time.perf_counter() - t0) / 300 * 1e3
        print("rep %d  %d contexts x %d in flight: %.4f ms per step" % (rep, K, depth, dt), flush=True)
