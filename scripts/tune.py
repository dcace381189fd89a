"""Tuning aid: interleaved A/B of launch options in ONE process (same device, same clocks)."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else "lut"
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
eng.set_grid(G, G, G); eng.set_cameras(cams, *masks[0].shape)
eng.upload_masks(masks); eng.upload_frame(1, frames[1])
if mode == "lut":
    eng.build_lut()
if mode == "lut":
    variants = [dict(lut_hier=0)]
    variants += [dict(lut_hier=1, refine_pair=pr, refine_b=rb, hier_blocks_per_cu=rc) for (pr, rb) in ((1, 8),) for rc in (12, 16, 24, 32, 48, 96)]
else:
    variants = [dict(fused_hier=0)] + [dict(fused_hier=1, hier_blocks_per_cu=b) for b in (8, 24, 48, 96)]
defaults = dict(first_kv=1, first_blocks_per_cu=3, refine_b=8, refine_blocks_per_cu=8, fused_blocks_per_cu=8, lut_hier=1, fused_hier=1, hier_blocks_per_cu=48, refine_pair=1, emit_lanes=1)
res = {i: [] for i in range(len(variants))}
for rnd in range(5):
    for i, v in enumerate(variants):
        for k, x in {**defaults, **v}.items():
            eng.set_option(k, x)
        eng.carve(mode=mode)
        t = eng.timing()
        res[i].append((t["carve_ms"], t["first_ms"], t["compact_ms"]))
for i, v in enumerate(variants):
    a = np.array(res[i][1:])
    print(v, "carve med %.4f min %.4f | first med %.4f | compact med %.4f" % (np.median(a[:, 0]), a[:, 0].min(), np.median(a[:, 1]), np.median(a[:, 2])), flush=True)
