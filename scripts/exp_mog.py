"""The background model's kernels under rocprofv3 (k_bgr2hsv, k_mog_apply, k_morph3x3): python scripts/exp_mog.py [H W]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import voxcarve
from voxcarve import background_subtraction as bs

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (486, 644)
eng = voxcarve.CarveEngine(0)
rng = np.random.default_rng(0)
bg = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
frames = [np.clip(bg.astype(np.int16) + rng.integers(-5, 6, bg.shape), 0, 255).astype(np.uint8) for _ in range(8)]
model = bs.train_MOG_background_model(frames=frames * 4, engine=eng)
t0 = time.perf_counter()
for i in range(100):
    eng.foreground_front(model._model, frames[i % 8], 0, True, True)
dt = (time.perf_counter() - t0) / 100 * 1e3
print("%d x %d: foreground_front (HSV + apply + 3x3 open + close, host buffers in and out) %.3f ms per frame" % (H, W, dt))
