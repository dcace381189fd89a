"""numpy restatement of the tail of the reference's extract_foreground_mask.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED: cv2.morphologyEx is absent here
and the reference holds no fixture of an intermediate mask, so this restates OpenCV's documented
behaviour (reference lines: background_subtraction.py:195-206):

  * cv2.getStructuringElement(MORPH_RECT, (2, 2)) with the default anchor (-1, -1) -> ksize/2 = (1, 1),
    so output (y, x) looks at rows y-1..y and columns x-1..x, for erode AND dilate (no reflection);
  * BORDER_CONSTANT with morphologyDefaultBorderValue(): pixels outside the image never win
    (ignored by both min and max);
  * MORPH_OPEN = dilate(erode(src)), MORPH_CLOSE = erode(dilate(src)), opening first when both apply;
  * final threshold: foreground[foreground > 0] = 255.
"""
import numpy as np


def _window_reduce(img, use_max):
    a = np.asarray(img, dtype=np.uint8)
    pad_val = 0 if use_max else 255
    p = np.full((a.shape[0] + 1, a.shape[1] + 1), pad_val, dtype=np.uint8)
    p[1:, 1:] = a
    f = np.maximum if use_max else np.minimum
    return f(f(p[1:, 1:], p[1:, :-1]), f(p[:-1, 1:], p[:-1, :-1]))


def erode2x2(img):
    return _window_reduce(img, use_max=False)


def dilate2x2(img):
    return _window_reduce(img, use_max=True)


def post_filter(mask, apply_opening_post=False, apply_closing_post=False):
    out = np.asarray(mask, dtype=np.uint8)
    if apply_opening_post:
        out = dilate2x2(erode2x2(out))
    if apply_closing_post:
        out = erode2x2(dilate2x2(out))
    return np.where(out > 0, 255, 0).astype(np.uint8)
