"""Step latency / throughput on the other BASELINE configurations (not bench lines)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
from voxcarve import synthetic

def run(label, grid, cams, masks, frames, cc, modes=("lut", "fused"), reps=30):
    eng = voxcarve.CarveEngine(0)
    eng.set_grid(*grid); eng.set_cameras(cams, *masks[0].shape)
    eng.upload_masks(masks); eng.upload_frame(cc, frames[cc])
    t0 = time.perf_counter(); eng.build_lut(); lut_s = time.perf_counter() - t0
    vv = float(np.prod(grid)) * len(cams)
    for mode in modes:
        for _ in range(3):
            n = eng.carve(mode=mode, color_cam=cc)
        t0 = time.perf_counter()
        for _ in range(reps):
            n = eng.carve(mode=mode, color_cam=cc)
        dt = (time.perf_counter() - t0) / reps
        tm = eng.timing()
        print("%-34s %-5s survivors %9d  step %8.4f ms (kernel %.4f + compact %.4f)  %10.1f Mvv/s  lut build %.1f ms" %
              (label, mode, n, dt * 1e3, tm["carve_ms"], tm["compact_ms"], vv / dt / 1e6, lut_s * 1e3), flush=True)
    eng.close()

cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
for g in (64, 128, 256, 512, 1024):
    run("%d^3 x 4 real cams 644x486" % g, (g, g, g), cams, masks, frames, 1)
H, W, C = 1080, 1920, 16
scams = synthetic.ring_cameras(C, H, W)
smasks = synthetic.ellipsoid_masks(scams, H, W)
sframes = [np.zeros((H, W, 3), np.uint8) for _ in range(C)]
sframes[5] = synthetic.random_frames(6, H, W)[5]
run("512^3 x 16 synthetic cams 1080p", (512, 512, 512), scams, smasks, sframes, 5, reps=10)
# the same scene without the specified salt noise (what a post-filtered mask looks like), and with the
# noise removed on the device by the reference's 2x2 open + close
clean = synthetic.ellipsoid_masks(scams, H, W, noise=0.0)
run("512^3 x 16 cams 1080p, no noise", (512, 512, 512), scams, clean, sframes, 5, reps=10)
