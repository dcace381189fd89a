"""Latency of the drop-in call the reference's viewer makes: set_voxel_positions(128, 64, 128)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fixtures_util as fx
from voxcarve import assignment
masks = fx.golden_masks(); frames = fx.synthetic_frames(4, *masks[0].shape)
for mode in ("fused", "lut"):
    sets = [(frames, [np.roll(m, i, axis=1) for m in masks]) for i in range(60)]
    assignment.configure(frame_source=assignment.StaticFrameSource(sets), data_path=os.path.join(fx.GOLDEN, "data"), mode=mode)
    t0 = time.perf_counter(); pos, col = assignment.set_voxel_positions(128, 64, 128); first = time.perf_counter() - t0
    ts = []
    for i in range(50):
        t0 = time.perf_counter(); pos, col = assignment.set_voxel_positions(128, 64, 128); ts.append(time.perf_counter() - t0)
    print("mode %-5s first call %.1f ms (context, cameras%s), then median %.3f ms per call, %d voxels, dtypes %s %s" %
          (mode, first * 1e3, ", table" if mode == "lut" else "", np.median(ts) * 1e3, len(pos), pos.dtype, col.dtype))
