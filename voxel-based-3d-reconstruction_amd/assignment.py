"""Drop-in for ``set_voxel_positions`` of the reference's assignment.py:54-149.

Same signature and module-level lazy state as the reference; the carve itself runs on the
GPU through the array fast path and the result comes back as two float32 ndarrays that
``Mesh.set_multiple_positions`` (engine/renderable/mesh.py:80-94) accepts as they are.

Frame and mask acquisition (video decode + MOG background subtraction, reference
assignment.py:68-82,93-109) is OUT of this path: it is delegated to a *frame source*.
The default source reuses the reference's own ``background_subtraction`` module and cv2
when this file is dropped into a reference checkout; tests and benchmarks install a
``StaticFrameSource``.
"""
import os

import numpy as np

from ._lib import VoxcarveError
from .camera import load_cameras
from .engine import (COLOR_CAMERA_INDEX, DEFAULT_BOUNDS, CarveEngine, viewer_colors, viewer_positions,
                     voxel_keys)

# reference assignment.py:28-33: figure_threshold, figure_inner_threshold, opening/closing pre/post
cam_bg_model_params = [
    [5000, 115, False, False, True, True],
    [5000, 115, False, False, True, True],
    [5000, 175, False, True, True, True],
    [5000, 115, False, False, False, True],
]


class StaticFrameSource:
    """Yields pre-computed (frames, masks) pairs; ``None`` when exhausted (end of video)."""

    def __init__(self, frame_sets):
        self._sets = list(frame_sets)
        self._pos = 0

    def next(self):
        if self._pos >= len(self._sets):
            return None
        item = self._sets[self._pos]
        self._pos += 1
        return item


class ReferenceVideoSource:
    """Reference acquisition (assignment.py:68-82, 93-109) through the reference's own modules.

    Needs cv2 (opencv-contrib) and the reference's ``background_subtraction`` / ``utils`` on
    sys.path -- true when this package is used from inside a reference checkout."""

    def __init__(self, data_path="data", num_cameras=4, post_on_device=False):
        # post_on_device: leave the 2x2 open/close tail of extract_foreground_mask to the GPU
        # (CarveEngine.set_mask_postfilter with cam_bg_model_params[c][4:6]); masks then come out unfiltered.
        self.post_on_device = post_on_device
        try:
            import cv2
            import background_subtraction
            import utils
        except ImportError as exc:
            # fail HERE, once and by name -- not with a bare ImportError out of the viewer's key callback
            raise VoxcarveError(
                "voxcarve.assignment.set_voxel_positions has no frame source: the default one decodes the videos and "
                "subtracts the background with the reference's own modules (cv2 / opencv-contrib, background_subtraction, "
                "utils), and %r is not importable here.  Run from inside a reference checkout with OpenCV installed, or "
                "call voxcarve.assignment.configure(frame_source=...) with an object whose next() returns "
                "(frames, masks) or None (e.g. StaticFrameSource)." % exc.name) from exc
        self._bs = background_subtraction
        self.videos, self.bg_models = [], []
        for camera in range(num_cameras):
            directory = os.path.join(data_path, "cam" + str(camera + 1))
            self.videos.append(cv2.VideoCapture(os.path.join(directory, "video.avi")))
            _, _, n_frames = utils.get_video_properties(directory, "background.avi")
            self.bg_models.append(background_subtraction.train_MOG_background_model(
                directory, "background.avi", use_hsv=True, history=n_frames, n_mixtures=50, bg_ratio=0.90,
                noise_sigma=0))

    def next(self):
        frames = [video.read()[1] for video in self.videos]
        if any(frame is None for frame in frames):
            return None
        masks = []
        for camera, frame in enumerate(frames):
            p = cam_bg_model_params[camera]
            post = (False, False) if self.post_on_device else (p[4], p[5])
            masks.append(np.array(self._bs.extract_foreground_mask(frame, self.bg_models[camera], 0, p[0], p[1],
                                                                   p[2], p[3], post[0], post[1])))
        return frames, masks


# module state, as the reference keeps it (assignment.py:22-40)
initialized = False
frame_count = 0
_engine = None
_source = None
_settings = {"data_path": "data", "num_cameras": 4, "device": 0, "mode": "fused",
             "views_threshold": 4, "color_camera": COLOR_CAMERA_INDEX, "bounds": DEFAULT_BOUNDS}


def configure(frame_source=None, **settings):
    """Install a frame source / override data_path, num_cameras, device, mode, ...; resets state."""
    global _source, _engine, initialized, frame_count
    unknown = set(settings) - set(_settings)
    if unknown:
        raise TypeError("unknown settings: %s" % sorted(unknown))
    _settings.update(settings)
    _source = frame_source
    if _engine is not None:
        _engine.close()
    _engine = None
    initialized = False
    frame_count = 0


def set_voxel_positions(width, height, depth):
    """Voxels seen by all cameras and their colours; reference assignment.py:54-149.

    :param width: voxel volume width
    :param height: HALF of the voxel volume height (the volume has 2*height cells in y)
    :param depth: voxel volume depth
    :return: (positions float32 [S,3], colors float32 [S,3]); ([], []) at the end of the video
    """
    global initialized, frame_count, _engine, _source
    if not initialized:
        if _source is None:
            _source = ReferenceVideoSource(_settings["data_path"], _settings["num_cameras"])
        _engine = CarveEngine(_settings["device"])
        _engine.set_grid(width, height * 2, depth, _settings["bounds"])        # assignment.py:85
        _engine._cameras = load_cameras(_settings["data_path"], _settings["num_cameras"])   # :88
        _engine._sized = None
        initialized = True

    item = _source.next()                                                       # assignment.py:94-96
    if item is None:
        return [], []
    frames, masks = item
    frame_count += 1

    H, W = np.asarray(masks[0]).shape[:2]
    if _engine._sized != (H, W):
        _engine.set_cameras(_engine._cameras, H, W)
        if getattr(_source, "post_on_device", False):
            n = _settings["num_cameras"]
            _engine.set_mask_postfilter([cam_bg_model_params[c][4] for c in range(n)],
                                        [cam_bg_model_params[c][5] for c in range(n)])
        if _settings["mode"] == "lut":
            _engine.build_lut()
        _engine._sized = (H, W)
    cc = _settings["color_camera"]
    _engine.upload_masks(masks, slot=0)
    _engine.upload_frame(cc, frames[cc], slot=0)
    _engine.carve(slot=0, min_views=_settings["views_threshold"], color_cam=cc, mode=_settings["mode"])
    idx, rgb, _ = _engine.fetch()
    keys = voxel_keys(idx, _engine.grid, _engine.axes())
    return viewer_positions(keys), viewer_colors(rgb)


def voxels_status():
    """Dense ON/OFF volume of the last set_voxel_positions call, shaped (width, height*2, depth) exactly as the
    reference builds it for voxel_reconstruction.plot_marching_cubes (assignment.py:143-146: the list of
    statuses in lookup-table order, reshaped): feed it to skimage.measure.marching_cubes as the reference does.
    The bits come straight off the device (vc_fetch_occupancy), no pass over Python dicts."""
    if _engine is None or not initialized:
        raise RuntimeError("set_voxel_positions has not run")
    nx, ny, nz = _engine.grid
    return _engine.fetch_occupancy().reshape(nx, ny, nz)

