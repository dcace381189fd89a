"""PMC target: N synchronous carves of one mode at 1024^3 x 4 with options k=v (run under rocprofv3 --pmc ...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxcarve, fixtures_util as fx
cams, masks = fx.golden_cameras(), fx.golden_masks()
frames = fx.synthetic_frames(4, *masks[0].shape)
eng = voxcarve.CarveEngine(0)
G = int(os.environ.get("GRID", "1024"))
eng.set_grid(G, G, G); eng.set_cameras(cams, *masks[0].shape)
eng.upload_masks(masks); eng.upload_frame(1, frames[1])
mode = "lut"
for opt in sys.argv[1:]:
    k, v = opt.split("=")
    if k == "mode":
        mode = v
    else:
        eng.set_option(k, int(v))
if mode == "lut":
    eng.build_lut()
eng.set_option("overlap", 0)
for _ in range(12):
    eng.carve(mode=mode)
print(mode, eng.count, eng.timing()["carve_ms"])
