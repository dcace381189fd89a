"""One-GPU rehearsal of what a rank of an N-GPU job does per step (1024^3 x 4, real fixtures):
slab carve (no records) + pack, then the expansion of ALL ranks' entries (uploaded once, expanded from the device
copy by re-running vc_expand_entries is avoided: the 1-rank communicator path of bench.py --force-comm times that).
Prints per-N: slab carve ms, pack ms, entries (count, bytes) per rank."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import voxcarve
import fixtures_util as fx
from voxcarve import slabs

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cams, masks = fx.golden_cameras(), fx.golden_masks()
H, W = masks[0].shape
frames = fx.synthetic_frames(len(cams), H, W)
eng = voxcarve.CarveEngine(0)
eng.set_grid(G, G, G)
eng.set_cameras(cams, H, W)
eng.upload_masks(masks)
eng.upload_frame(1, frames[1])
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 16
t0 = time.perf_counter()
weights = slabs.measure_chunk_cost(eng, G, CH) if len(sys.argv) > 3 else None
if weights:
    print("chunk cost (%d layers each, %.2f s to measure):" % (CH, time.perf_counter() - t0), " ".join("%.3f" % w for w in weights))
for N in (1, 2, 4, 8):
    rows = []
    bounds = slabs.balanced_bounds(weights, CH, G, N) if weights else [slabs.slab_range(G, N, r)[0] for r in range(N)] + [G]
    print("bounds", bounds)
    for r in range(N):
        z0, z1 = bounds[r], bounds[r + 1]
        eng.set_slab(z0, z1)
        eng.build_lut()
        for _ in range(3):
            eng.carve(mode="lut", records=False)
        eng.timing(reset=True)
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            n = eng.carve(mode="lut", records=False)
        wall = (time.perf_counter() - t0) / K * 1e3
        tm = eng.timing()
        t0 = time.perf_counter()
        ent = eng.pack_entries()
        pack_wall = (time.perf_counter() - t0) * 1e3
        rows.append((r, n, ent.shape[0], tm["carve_ms_sum"] / tm["carve_launches"], tm["compact_ms"], wall, pack_wall))
    tot_e = sum(x[2] for x in rows)
    print("N=%d  total entries %d (%.2f MB) for %d survivors (%.1f MB of records)" % (
        N, tot_e, tot_e * 16 / 1e6, sum(x[1] for x in rows), sum(x[1] for x in rows) * 8 / 1e6))
    for x in rows:
        print("   rank %d: survivors %9d entries %7d carve %.4f ms scan %.4f ms sync-call %.4f ms" % x[:6])
