"""The reference's ``extract_foreground_mask`` (background_subtraction.py:129-208) and its MOG background model
(``train_MOG_background_model``, :49-92) with their data-parallel stages on the GPU.

SURVEY 8(f)-2, the step BEFORE the carve path.  Same names, parameters and defaults as the reference's functions.  What runs where:

  BGR -> HSV (:155)                      GPU   CarveEngine.bgr_to_hsv          (OpenCV's 8-bit fixed-point conversion)
  bg_model.apply (:158)                  GPU   BackgroundSubtractorMOG.apply   (the model assignment.py:79 trains; per pixel mixture
                                               of Gaussians, state in HBM) -- or whatever model the caller hands in (a cv2 MOG2 / KNN
                                               object works as before, on the CPU)
  3x3 open / close before the contours   GPU   CarveEngine.mask_morphology(.., 3, ..)
  contours: fill the figures, re-open
  their large holes (:171-193)           CPU   cv2.findContours / fillPoly / drawContours: sequential border following
  2x2 open / close after them (:195-203) GPU   CarveEngine.mask_morphology(.., 2, ..)
  final threshold (:206)                 host  one comparison

The CPU stage needs cv2 (as the reference does), and so does decoding the training video; without it those calls fail by name --
there is no substitute for them in this package.  Parity of the GPU stages with cv2 is unpinned (see oracle/foreground_np.py,
oracle/mog_np.py)."""
import numpy as np

from ._lib import VoxcarveError

_engine = None


def _default_engine():
    global _engine
    if _engine is None:
        from .engine import CarveEngine
        _engine = CarveEngine(0)
    return _engine


class BackgroundSubtractorMOG:
    """cv2.bgsegm.createBackgroundSubtractorMOG(history, nmixtures, backgroundRatio, noiseSigma) with the model on the device
    (background_subtraction.py:75-76).  ``apply(image, fgmask=None, learningRate=-1)`` as cv2's: uint8 [H,W,3] in, uint8 [H,W]
    {0, 255} out; -1 = 1 / min(frames seen, history), 0 = the model is only read (the reference's inference, :158)."""

    def __init__(self, history=200, nmixtures=5, backgroundRatio=0.7, noiseSigma=0, engine=None):
        self._eng = engine if engine is not None else _default_engine()
        self._model = self._eng.mog_create(history, nmixtures, backgroundRatio, noiseSigma)

    def apply(self, image, fgmask=None, learningRate=-1):
        out = self._eng.mog_apply(self._model, image, learningRate)
        if fgmask is not None:
            fgmask[...] = out
            return fgmask
        return out

    def state(self):
        return self._eng.mog_state(self._model)

    def close(self):
        if self._model is not None:
            self._eng.mog_destroy(self._model)
            self._model = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _video_frames(path):
    try:
        import cv2
    except ImportError as exc:
        raise VoxcarveError("train_MOG_background_model: decoding %s runs on cv2.VideoCapture, which is not importable here (%s); "
                            "pass the frames themselves (frames=...)" % (path, exc))
    cap = cv2.VideoCapture(path)
    if not cap.isOpened():
        return None

    def gen():
        while True:
            ok, frame = cap.read()
            if not ok:
                return
            yield frame
    return gen()


def train_MOG_background_model(bg_video_input_path="data/cam", bg_video_input_filename="background.avi", use_hsv=True,
                               history=200, n_mixtures=5, bg_ratio=0.7, noise_sigma=0, learning_rate=-1, engine=None, frames=None):
    """A MOG model trained on a background video; reference background_subtraction.py:49-92, same parameters (None if the video
    cannot be opened, as there).  ``frames``: an iterable of BGR frames instead of the video file (no cv2 needed then)."""
    import os
    if frames is None:
        frames = _video_frames(os.path.join(bg_video_input_path, bg_video_input_filename))
        if frames is None:
            return None
    eng = engine if engine is not None else _default_engine()
    model = BackgroundSubtractorMOG(history=history, nmixtures=n_mixtures, backgroundRatio=bg_ratio, noiseSigma=noise_sigma, engine=eng)
    for frame in frames:
        if use_hsv:
            frame = eng.bgr_to_hsv(frame)
        model.apply(frame, None, learning_rate)
    return model


def fill_figures(mask, figure_threshold, figure_inner_threshold):
    """The contour stage (:171-193) with cv2: every contour of the RETR_TREE hierarchy whose area reaches ``figure_threshold``
    is drawn filled; each of its direct children whose oriented area reaches ``figure_inner_threshold`` is cleared again and
    its outline kept."""
    try:
        import cv2
    except ImportError as exc:
        raise VoxcarveError("extract_foreground_mask: the contour stage (findContours / fillPoly, background_subtraction.py:171-193) "
                            "runs on cv2, which is not importable here (%s)" % exc)
    contours, hierarchy = cv2.findContours(mask, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)
    out = np.zeros(mask.shape, dtype=np.uint8)
    for k, outer in enumerate(contours):
        if cv2.contourArea(outer) < figure_threshold:
            continue
        cv2.drawContours(out, [outer], -1, 255)
        cv2.fillPoly(out, [outer], 255)
        child = hierarchy[0][k][2]                            # first child; siblings follow through field 0
        while child != -1:
            hole = contours[child]
            if cv2.contourArea(hole, True) >= figure_inner_threshold:
                cv2.fillPoly(out, [hole], 0)
                cv2.drawContours(out, [hole], -1, 255)
            child = hierarchy[0][child][0]
    return out


def extract_foreground_mask(image, bg_model, learning_rate=0, figure_threshold=5000, figure_inner_threshold=115,
                            apply_opening_pre=False, apply_closing_pre=False, apply_opening_post=False,
                            apply_closing_post=False, engine=None, contour_stage=None):
    """Foreground mask (uint8 {0, 255} [H, W]) of a BGR image; reference background_subtraction.py:129-208, same parameters.
    ``engine``: the CarveEngine whose device does the work (default: one on device 0); ``contour_stage``: what stands in for
    ``fill_figures`` (tests; default: the cv2 one)."""
    eng = engine if engine is not None else _default_engine()
    if isinstance(bg_model, BackgroundSubtractorMOG) and bg_model._eng is eng:
        # the model lives on this device: colour conversion, apply and pre-filter without leaving it
        model_mask = eng.foreground_front(bg_model._model, image, learning_rate, apply_opening_pre, apply_closing_pre)
    else:
        hsv = eng.bgr_to_hsv(image)
        model_mask = np.ascontiguousarray(bg_model.apply(hsv, None, learning_rate), dtype=np.uint8)
        if apply_opening_pre or apply_closing_pre:
            model_mask = eng.mask_morphology(model_mask, 3, apply_opening_pre, apply_closing_pre)
    figures = (contour_stage or fill_figures)(model_mask, figure_threshold, figure_inner_threshold)
    if apply_opening_post or apply_closing_post:
        figures = eng.mask_morphology(figures, 2, apply_opening_post, apply_closing_post)
    return np.where(figures > 0, 255, 0).astype(np.uint8)
