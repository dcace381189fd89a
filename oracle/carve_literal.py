"""Dict-and-loop restatement of the reference's own Python control flow.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py`` (parity unpinned vs cv2).

Where ``carve_np`` is array-shaped, this file keeps the reference's data shapes
(lookup table = dict camera -> list of (voxel key, (x, y)); visibility = dict of
dicts) so that insertion order, key truncation and key collisions behave as in
the reference.  Pure-Python loops: small grids only (<= 64^3).  The only thing
not literal is ``cv2.projectPoints`` itself, replaced by
``carve_np.project_points`` (the OpenCV formula restated).
"""
import numpy as np

from . import carve_np


def build_lookup_table(voxel_points, cams):
    """voxel_reconstruction.py:74-86 -- cameras keyed 1..C, entries (int-key, (x, y))."""
    table = {}
    for cam_key, (K, dist, R, t) in enumerate(cams, start=1):
        projected = carve_np.project_points(voxel_points, R, t, K, dist)
        entries = []
        for point, uv in zip(voxel_points, projected):
            key = (int(point[0]), int(point[1]), int(point[2]))      # tuple(map(int, voxel))
            entries.append((key, (uv[0], uv[1])))
        table[cam_key] = entries
    return table


def visible_voxels_and_colors(table, fg_masks, images):
    """voxel_reconstruction.py:89-124 -- float bounds test, int() truncation, mask > 0."""
    visible, colors = {}, {}
    for cam_key, entries in table.items():
        mask = fg_masks[cam_key - 1]
        image = images[cam_key - 1]
        rows, cols = mask.shape[0], mask.shape[1]
        for key, (x, y) in entries:
            if not (0 <= y < rows and 0 <= x < cols):
                continue
            r, c = int(y), int(x)
            if mask[r, c] > 0:
                visible.setdefault(key, {})[cam_key] = True
                colors.setdefault(key, {})[cam_key] = np.array(image[r, c, :])
    return visible, colors


def select_for_viewer(visible, colors, views_threshold=carve_np.VIEWS_THRESHOLD,
                      color_camera_key=carve_np.COLOR_CAMERA_KEY,
                      scaling_factor=carve_np.SCALING_FACTOR):
    """assignment.py:116-133,149 -- threshold on sum(views), axis swap, /64, BGR->RGB/255."""
    data, cols = [], []
    for key, views in visible.items():
        if sum(views.values()) >= views_threshold:
            data.append([key[0] / scaling_factor, -(key[2] / scaling_factor), key[1] / scaling_factor])
            cols.append(colors[key][color_camera_key][::-1] / 255.0)
    return data, cols


def set_voxel_positions(width, height, depth, cams, fg_masks, images):
    """assignment.py:54-149 minus video / background-model acquisition."""
    points = carve_np.create_voxel_volume(width, height * 2, depth)            # assignment.py:85
    table = build_lookup_table(points, cams)                                   # assignment.py:88
    visible, colors = visible_voxels_and_colors(table, fg_masks, images)       # assignment.py:113
    return select_for_viewer(visible, colors, views_threshold=len(cams))
