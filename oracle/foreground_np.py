"""numpy restatement of the data-parallel part of the FRONT half of the reference's extract_foreground_mask.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED: cv2 is absent here and the reference holds no
intermediate image of this stage (the only mask fixtures are the END of the pipeline, behind a stateful MOG model trained on
a video that cannot be decoded here), so this restates OpenCV 4.x's implementation as published; the reference lines it
stands for are background_subtraction.py:153-168:

  * :155  cv2.cvtColor(image, cv2.COLOR_BGR2HSV) on uint8 -- OpenCV's 8-bit path (imgproc, color_hsv: RGB2HSV_b, hrange 180) is
          FIXED POINT, not the float formula rounded: with hsv_shift = 12,
              sdiv[v]  = cvRound((255 << 12) / v)        hdiv[d] = cvRound((180 << 12) / (6 d))      (0 for v, d = 0)
              v = max(b, g, r);  d = v - min(b, g, r)
              s = (d * sdiv[v] + 2048) >> 12
              h = (g - b) if v == r else (b - r + 2 d) if v == g else (r - g + 4 d)          (first match wins: r, then g)
              h = (h * hdiv[d] + 2048) >> 12;  h += 180 if h < 0                              (arithmetic shift)
          cvRound = round half to even (lrint).  Output channel order H, S, V.
  * :161-168  cv2.morphologyEx(mask, MORPH_OPEN / MORPH_CLOSE, getStructuringElement(MORPH_RECT, (3, 3))): anchor at the centre,
          output (y, x) looks at rows y-1..y+1, columns x-1..x+1; BORDER_CONSTANT with morphologyDefaultBorderValue(), i.e.
          pixels outside the image never win; OPEN = dilate(erode(src)), CLOSE = erode(dilate(src)); opening first when both.

bg_model.apply (:158) and findContours / fill (:171-193) are NOT restated: stateful and sequential, they stay with cv2 on the CPU.
"""
import numpy as np

HSV_SHIFT = 12


def _tables():
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    sdiv[1:] = np.rint((255 << HSV_SHIFT) / i).astype(np.int64)          # np.rint: half to even, as cvRound
    hdiv[1:] = np.rint((180 << HSV_SHIFT) / (6.0 * i)).astype(np.int64)
    return sdiv, hdiv


SDIV, HDIV = _tables()


def bgr_to_hsv(image):
    """uint8 [..., 3] BGR -> uint8 [..., 3] HSV with H in 0..179 (cv2.COLOR_BGR2HSV on 8-bit input)."""
    a = np.asarray(image, dtype=np.uint8).astype(np.int64)
    b, g, r = a[..., 0], a[..., 1], a[..., 2]
    v = np.maximum(np.maximum(b, g), r)
    d = v - np.minimum(np.minimum(b, g), r)
    s = (d * SDIV[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * d, r - g + 4 * d))
    h = (h * HDIV[d] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT              # numpy >> on int64 is arithmetic, like C on int
    h = np.where(h < 0, h + 180, h)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


def _window3(img, use_max):
    a = np.asarray(img, dtype=np.uint8)
    pad = 0 if use_max else 255
    p = np.full((a.shape[0] + 2, a.shape[1] + 2), pad, dtype=np.uint8)
    p[1:-1, 1:-1] = a
    f = np.maximum if use_max else np.minimum
    out = p[1:-1, 1:-1]
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            out = f(out, p[dy:dy + a.shape[0], dx:dx + a.shape[1]])
    return out


def erode3x3(img):
    return _window3(img, use_max=False)


def dilate3x3(img):
    return _window3(img, use_max=True)


def pre_filter(mask, apply_opening_pre=False, apply_closing_pre=False):
    """background_subtraction.py:161-168 on the background model's mask (grey levels kept: MOG2 marks shadows 127)."""
    out = np.asarray(mask, dtype=np.uint8)
    if apply_opening_pre:
        out = dilate3x3(erode3x3(out))
    if apply_closing_pre:
        out = erode3x3(dilate3x3(out))
    return out
