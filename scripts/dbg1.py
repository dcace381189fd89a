import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
idx_want, bgr_want, summary = fx.expected(64)
eng = voxcarve.CarveEngine(0)
eng.set_grid(64, 64, 64); eng.set_cameras(cams, *masks[0].shape); eng.upload_masks(masks)
for c, f in enumerate(frames): eng.upload_frame(c, f)
eng.build_lut()
for mode in ("fused", "lut"):
    n = eng.carve(mode=mode)
    idx, rgb, seen = eng.fetch()
    bad = np.nonzero(idx != idx_want)[0]
    print(mode, n, "mismatches", bad.size, bad[:10], idx[bad[:10]], idx_want[bad[:10]])
    occ = np.nonzero(eng.fetch_occupancy())[0]
    print("  occupancy equal to want:", np.array_equal(occ, idx_want), "sorted idx equal:", np.array_equal(np.sort(idx), idx_want),
          "unique", np.unique(idx).size)
