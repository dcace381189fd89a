"""The three CPU baselines of BASELINE.md section 4 on this host (B0 literal dict/loop, B1 numpy, B2 C/OpenMP)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fixtures_util as fx
from oracle import carve_c, carve_literal, carve_np

cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
oc = fx.oracle_cams(cams)
cores = len(os.sched_getaffinity(0))
print("host threads available:", cores)

t0 = time.perf_counter()
data, cols = carve_literal.set_voxel_positions(64, 32, 64, oc, masks, frames)
dt = time.perf_counter() - t0
print("B0 literal dict/loop   64^3  1 core   %8.2f s  %10.3f Mvv/s  survivors %d" % (dt, 64 ** 3 * 4 / dt / 1e6, len(data)))

for g in (64, 256):
    t0 = time.perf_counter()
    r = carve_np.carve(g, g, g, oc, masks, frames)
    dt = time.perf_counter() - t0
    print("B1 numpy vectorised   %4d^3  1 core   %8.2f s  %10.3f Mvv/s  survivors %d" % (g, dt, g ** 3 * 4 / dt / 1e6, r["idx"].size))

carve_c.carve(64, 64, 64, oc, masks, frames, threads=cores)          # warm the thread pool
for g, rng in ((64, None), (256, None), (1024, (384 * 1024 * 1024, 640 * 1024 * 1024))):
    for th in (1, cores):
        if g == 1024 and th == 1:
            rng1 = (500 * 1024 * 1024, 516 * 1024 * 1024)
        else:
            rng1 = rng
        n = (rng1[1] - rng1[0]) if rng1 else g ** 3
        t0 = time.perf_counter()
        r = carve_c.carve(g, g, g, oc, masks, frames, index_range=rng1, threads=th, cap=1 << 26)
        dt = time.perf_counter() - t0
        print("B2 C/OpenMP           %4d^3  %3d thr  %8.3f s  %10.3f Mvv/s  (%s)" %
              (g, th, dt, n * 4 / dt / 1e6, "whole grid" if rng1 is None else "z-layers %d..%d" % (rng1[0] >> 20, rng1[1] >> 20)))
