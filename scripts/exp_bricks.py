"""Experiment: where the strip carve's time goes (1024^3 x 4): carve kernel time alone under debug switches + brick statistics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
G = int(os.environ.get("GRID", "1024"))
eng.set_grid(G, G, G); eng.set_cameras(cams, *masks[0].shape)
eng.upload_masks(masks); eng.upload_frame(1, frames[1])
eng.build_lut()
eng.set_option("overlap", 0)
def t(mode, n=20, **opts):
    for k, v in opts.items():
        eng.set_option(k, v)
    for _ in range(3):
        eng.carve(mode=mode)
    eng.timing(reset=True)
    for _ in range(n):
        eng.carve(mode=mode)
    tm = eng.timing()
    cnt = eng.count
    dc = eng.debug_counters()
    for k in opts:
        eng.set_option(k, {"dbg": 0, "bricks": 1, "cull": 1, "strips_per_wave": 1, "grid_lds_kb": 16}[k])
    return "%-6s %-40s carve %.4f ms  compact %.4f  survivors %d  %s" % (mode, opts, tm["carve_ms_sum"] / tm["carve_launches"], tm["compact_ms"], cnt, dc)
for mode in ("lut",):
    print(t(mode))
    print(t(mode, dbg=1))
    print(t(mode, dbg=3))
    
    

    print(t(mode, bricks=0))
    print(t(mode, bricks=0, cull=0))

