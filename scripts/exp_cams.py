"""Tuning aid: carve kernel time vs number of cameras (which phase costs what)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["lut", "fused"]
for sel in ([2], [2, 1], [2, 1, 3], [2, 1, 3, 0]):
    eng = voxcarve.CarveEngine(0)
    eng.set_grid(G, G, G)
    eng.set_cameras([cams[c] for c in sel], *masks[0].shape)
    eng.upload_masks([masks[c] for c in sel])
    if "lut" in modes:
        eng.build_lut()
    for mode in modes:
        ts = []
        for it in range(6):
            n = eng.carve(mode=mode, color_cam=None)
            ts.append(eng.timing()["carve_ms"])
        print("cams", sel, mode, "survivors", n, "kernel_ms min %.4f med %.4f" % (min(ts[1:]), sorted(ts[1:])[2]),
              "compact_ms %.3f" % eng.timing()["compact_ms"], flush=True)
    eng.close()
