"""Extracts the only cv2.projectPoints OUTPUT the reference holds and commits it as a fixture.

The reference's calibration script draws three arrows on frame 0 of checkerboard.avi and saves the
picture as data/cam{1..4}/test.jpg (camera_calibration.py:753-789, called from :967-969 with
chessboard_square_size = 115 mm (data/checkerboard.xml) and a span of 3 squares, :847-849):

    axes  = np.float32([[1,0,0],[0,1,0],[0,0,-1]]) * 345
    tips  = cv2.projectPoints(axes, rvecs, tvecs, mtx, dist)[0].astype(np.int32)
    cv2.arrowedLine(image, origin_corner, tips[0], (0,0,255), 2)     # red   = +X
    cv2.arrowedLine(image, origin_corner, tips[1], (0,255,0), 2)     # green = +Y
    cv2.arrowedLine(image, origin_corner, tips[2], (255,0,0), 2)     # blue  = -Z

with exactly the (mtx, dist, rvecs, tvecs) written to config.xml three lines later (:972-974).  The
arrow tips in those four JPEGs are therefore cv2.projectPoints results for the committed cameras,
truncated to integers and blurred by a 2-px pen and JPEG chroma subsampling: a pixel-level pin of
K / distortion / Rodrigues / translation conventions and of the axis order -- not of the last ulp.

This script only reads DATA files of the reference (the JPEGs), decodes them ONCE with Pillow and
measures the drawings; it uses nothing of the oracle.  Per camera and colour: the pixels of that
pure colour, their principal axis (the shaft), the two extreme pixels along it; the three arrows
share their start point, so the extreme NEAR the other two colours' pixels is the origin corner
and the FAR one is the tip.

Usage: python tests/golden/make_arrow_tips.py      (writes tests/golden/arrow_tips.json)
"""
import json
import os

import numpy as np
from PIL import Image
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data"
AXIS_LENGTH_MM = 3 * 115          # camera_calibration.py:849 (span 3) x data/checkerboard.xml (115 mm)
# world points of the three tips, camera_calibration.py:770
AXES = [[AXIS_LENGTH_MM, 0, 0], [0, AXIS_LENGTH_MM, 0], [0, 0, -AXIS_LENGTH_MM]]
COLOURS = ("red", "green", "blue")


def colour_pixels(rgb, which):
    r, g, b = (rgb[..., k].astype(np.int32) for k in range(3))
    if which == "red":
        m = (r > 170) & (g < 90) & (b < 90)
    elif which == "green":
        m = (g > 170) & (r < 90) & (b < 90)
    else:
        m = (b > 170) & (r < 90) & (g < 90)
    # the frame itself holds a few saturated pixels: keep the largest 8-connected blob (the arrow)
    lab, n = ndimage.label(m, structure=np.ones((3, 3), dtype=int))
    if n > 1:
        m = lab == (1 + int(np.argmax(ndimage.sum(m, lab, index=np.arange(1, n + 1)))))
    ys, xs = np.nonzero(m)
    return np.stack([xs, ys], axis=1).astype(np.float64)


def extremes(pts):
    c = pts.mean(axis=0)
    _, _, vt = np.linalg.svd(pts - c, full_matrices=False)
    t = (pts - c) @ vt[0]
    return pts[np.argmin(t)], pts[np.argmax(t)]


def main():
    out = {"axis_length_mm": AXIS_LENGTH_MM, "object_points": AXES, "colours": COLOURS,
           "source": "reference data/cam{1..4}/test.jpg, decoded once with Pillow %s" % Image.__version__,
           "cameras": []}
    for c in range(1, 5):
        rgb = np.asarray(Image.open(os.path.join(REF, "cam%d" % c, "test.jpg")).convert("RGB"))
        sets = {k: colour_pixels(rgb, k) for k in COLOURS}
        tips, starts, counts = [], [], []
        for k in COLOURS:
            a, b = extremes(sets[k])
            others = np.concatenate([sets[o] for o in COLOURS if o != k])
            da = np.min(np.linalg.norm(others - a, axis=1))
            db = np.min(np.linalg.norm(others - b, axis=1))
            tip, start = (b, a) if da < db else (a, b)
            tips.append([int(tip[0]), int(tip[1])])
            starts.append([int(start[0]), int(start[1])])
            counts.append(int(sets[k].shape[0]))
        out["cameras"].append({"image_size": [int(rgb.shape[0]), int(rgb.shape[1])], "tips_xy": tips,
                               "shaft_starts_xy": starts, "pixels": counts})
        print("cam%d" % c, "tips", tips, "starts", starts, "pixels", counts)
    json.dump(out, open(os.path.join(HERE, "arrow_tips.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
