"""Shared loaders for the committed fixtures (tests/golden/) and seeded synthetic inputs."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _unhex(lst, shape):
    return np.array([float.fromhex(h) for h in lst], dtype=np.float64).reshape(shape)


def golden_cameras():
    """The 4 real cameras with R pinned to the committed hex floats."""
    from voxcarve.camera import Camera
    g = json.load(open(os.path.join(GOLDEN, "cameras.json")))
    return [Camera(_unhex(c["K"], (3, 3)), _unhex(c["dist"], 5), _unhex(c["rvec"], 3), _unhex(c["tvec"], 3),
                   R=_unhex(c["R"], (3, 3))) for c in g["cameras"]]


def golden_masks():
    z = np.load(os.path.join(GOLDEN, "masks_mog.npz"))
    H, W = int(z["H"]), int(z["W"])
    return [(np.unpackbits(b, bitorder="little")[:H * W].reshape(H, W) * 255).astype(np.uint8) for b in z["bits"]]


def synthetic_frames(C, H, W):
    return [np.random.default_rng(2000 + c).integers(0, 256, (H, W, 3), dtype=np.uint8) for c in range(C)]


def oracle_cams(cams):
    return [(c.K, c.dist, c.R, c.tvec) for c in cams]


def expected(n):
    z = np.load(os.path.join(GOLDEN, "expected_%d.npz" % n))
    summary = json.load(open(os.path.join(GOLDEN, "expected_summary.json")))[str(n)]
    return z["idx"], z["bgr"], summary


def random_scene(seed, C=3, H=37, W=53, fg=0.4):
    """Random but plausible cameras around the default volume + random masks / frames."""
    from voxcarve.camera import Camera
    rng = np.random.default_rng(seed)
    cams = []
    for _ in range(C):
        rvec = rng.normal(size=3)
        rvec *= rng.uniform(0.2, 3.0) / np.linalg.norm(rvec)
        K = np.array([[rng.uniform(20, 60), 0, W / 2 + rng.normal()], [0, rng.uniform(20, 60), H / 2 + rng.normal()],
                      [0, 0, 1.0]])
        dist = np.array([rng.normal(0, 0.2), rng.normal(0, 0.1), rng.normal(0, 1e-3), rng.normal(0, 1e-3),
                         rng.normal(0, 0.05)])
        tvec = np.array([rng.normal(0, 300), rng.normal(0, 300), rng.uniform(1500, 5000)])
        cams.append(Camera(K, dist, rvec, tvec))
    masks = [np.where(rng.random((H, W)) < fg, rng.integers(1, 256, (H, W)), 0).astype(np.uint8) for _ in range(C)]
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(C)]
    return cams, masks, frames


def arrow_tips():
    """The reference-held cv2.projectPoints pin (tests/golden/make_arrow_tips.py): world points [3,3] and,
    per camera, the integer arrow-tip pixels [3,2] measured in the reference's data/cam*/test.jpg."""
    t = json.load(open(os.path.join(GOLDEN, "arrow_tips.json")))
    return np.array(t["object_points"], dtype=np.float64), [np.array(c["tips_xy"], dtype=np.float64) for c in t["cameras"]]


ARROW_TIP_TOL_PX = 2.0     # 2-px pen, JPEG chroma blur, the reference's int32 truncation


def arrow_tip_error(project, cams):
    """max over cameras / tips of the Chebyshev distance between project(cam_index, cam, points) and the drawn tip."""
    pts, tips = arrow_tips()
    return max(float(np.abs(np.asarray(project(c, cam, pts)) - tips[c]).max()) for c, cam in enumerate(cams))
