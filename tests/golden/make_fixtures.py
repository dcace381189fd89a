"""Generates the committed fixtures under tests/golden/ (run once, in the build container).

Inputs are DATA files of the reference only (never its source):
  /root/reference/data/cam{1..4}/config.xml   calibration, exact float64 text
  /root/reference/data/cam{1..4}/mask_MOG.jpg frame-0 foreground masks (lossy JPEG)
The JPEGs are decoded ONCE here with Pillow and binarised at >= 128 (raw > 0 would
admit JPEG ringing); decoders differ, so tests use the committed bits, never the JPEG.

Expected outputs come from the numpy oracle (oracle/carve_np.py) and are cross-checked
here against the C oracle and the literal dict/loop oracle before being written.
The reference itself cannot be run (cv2 absent): parity vs OpenCV is UNPINNED.

Usage: python tests/golden/make_fixtures.py
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import carve_c, carve_literal, carve_np  # noqa: E402

REF = "/root/reference/data"
H, W = 486, 644


def hexlist(a):
    return [float(v).hex() for v in np.asarray(a, np.float64).reshape(-1)]


def synthetic_frames(C, H, W):
    return [np.random.default_rng(2000 + c).integers(0, 256, (H, W, 3), dtype=np.uint8) for c in range(C)]


def main():
    cams, cam_json, masks = [], [], []
    for c in range(1, 5):
        src = os.path.join(REF, "cam%d" % c, "config.xml")
        dst_dir = os.path.join(HERE, "data", "cam%d" % c)
        os.makedirs(dst_dir, exist_ok=True)
        shutil.copyfile(src, os.path.join(dst_dir, "config.xml"))
        K, dist, rvec, tvec = carve_np.read_config_xml(src)
        R = carve_np.rodrigues(rvec)
        cams.append((K, dist, R, tvec))
        cam_json.append({"K": hexlist(K), "dist": hexlist(dist), "rvec": hexlist(rvec),
                         "tvec": hexlist(tvec), "R": hexlist(R),
                         "K_dec": [repr(float(v)) for v in K.reshape(-1)]})
        img = np.asarray(Image.open(os.path.join(REF, "cam%d" % c, "mask_MOG.jpg")).convert("L"))
        assert img.shape == (H, W)
        masks.append(np.where(img >= 128, 255, 0).astype(np.uint8))
    json.dump({"H": H, "W": W, "cameras": cam_json}, open(os.path.join(HERE, "cameras.json"), "w"), indent=1)
    bits = np.stack([np.packbits(m.reshape(-1) > 0, bitorder="little") for m in masks])
    np.savez_compressed(os.path.join(HERE, "masks_mog.npz"), bits=bits, H=H, W=W)

    frames = synthetic_frames(4, H, W)
    summary = {}
    for n, half in ((64, 32), (128, 64)):
        res = carve_np.carve(n, 2 * half, n, cams, masks, frames)
        resc = carve_c.carve(n, 2 * half, n, cams, masks, frames, want_viewmask=True, want_lut=True)
        assert np.array_equal(res["idx"], resc["idx"]), "numpy vs C oracle: survivors differ"
        assert np.array_equal(res["viewmask"], resc["viewmask"])
        assert np.array_equal(res["offsets"], resc["offsets"])
        assert np.array_equal(res["bgr"], resc["bgr"])
        if n == 64:
            data, cols = carve_literal.set_voxel_positions(n, half, n, cams, masks, frames)
            keys = carve_np.voxel_keys(res["idx"], n, 2 * half, n)
            assert np.array_equal(np.array(data), carve_np.viewer_positions(keys)), "literal vs numpy: positions"
            assert np.array_equal(np.array(cols), carve_np.viewer_colors(res["bgr"])), "literal vs numpy: colours"
        any_view = int((res["viewmask"] != 0).sum())
        np.savez_compressed(os.path.join(HERE, "expected_%d.npz" % n), idx=res["idx"], bgr=res["bgr"],
                            viewmask_hist=np.bincount(res["viewmask"], minlength=16))
        summary[str(n)] = {
            "survivors": int(res["idx"].size), "any_view": any_view,
            "idx_sha256": hashlib.sha256(res["idx"].tobytes()).hexdigest(),
            "viewmask_sha256": hashlib.sha256(res["viewmask"].tobytes()).hexdigest(),
            "offsets_sha256": hashlib.sha256(res["offsets"].tobytes()).hexdigest(),
            "bgr_sha256": hashlib.sha256(res["bgr"].tobytes()).hexdigest(),
        }
        print(n, summary[str(n)])

    # Projected-pixel goldens: 256 voxels of the 128^3 grid per camera, incl. out-of-image ones.
    rng = np.random.default_rng(7)
    sample_idx = np.sort(rng.choice(128 ** 3, 256, replace=False))
    pts = carve_np.points_of_indices(sample_idx, 128, 128, 128)
    proj = {"grid": [128, 128, 128], "idx": [int(i) for i in sample_idx], "uv": []}
    for cam in cams:
        uv = carve_np.project_points(pts, cam[2], cam[3], cam[0], cam[1])
        assert np.array_equal(uv, carve_c.project(pts, cam))
        proj["uv"].append(hexlist(uv))
    json.dump(proj, open(os.path.join(HERE, "projected_samples.json"), "w"))
    json.dump(summary, open(os.path.join(HERE, "expected_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
