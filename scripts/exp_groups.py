"""Tuning aid: how the survivors are spread over 4096-voxel groups at 1024^3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
eng = voxcarve.CarveEngine(0)
eng.set_grid(1024, 1024, 1024); eng.set_cameras(cams, *masks[0].shape)
eng.upload_masks(masks)
n = eng.carve(mode="fused", color_cam=None)
occ = eng.fetch_occupancy()
g = occ.reshape(-1, 4096).sum(axis=1)
w = occ.reshape(-1, 64).sum(axis=1)
print("survivors", n, "groups", g.size, "busy groups", (g > 0).sum(), "mean per busy", g[g > 0].mean(), "max", g.max())
print("hist of busy group sizes (bins of 512):", np.bincount(g[g > 0] // 512))
print("busy words", (w > 0).sum(), "mean per busy word", w[w > 0].mean())
it = np.ceil(g / 256).sum()
print("sum over groups of ceil(cnt/256) =", it)
# waves: strided ownership
for nw in (8192,):
    per = np.array([np.ceil(g[wv::nw] / 256).sum() for wv in range(0, nw, 64)])
    print("iterations per wave (sampled): mean %.1f max %d min %d" % (per.mean(), per.max(), per.min()))
