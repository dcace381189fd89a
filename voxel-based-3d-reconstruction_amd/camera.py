"""Camera parameters for the carve path: config.xml reader and Rodrigues (host side).

Replaces the cv2 touch points around the hot path so it runs without OpenCV:
``cv2.FileStorage`` (reference utils.py:115-152 via voxel_reconstruction.py:10-32) and the
``cv2.Rodrigues`` that ``cv2.projectPoints`` applies to the rotation vector
(voxel_reconstruction.py:81).
"""
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass

import numpy as np

NODE_TAGS = ("CameraMatrix", "DistortionCoeffs", "RotationVector", "TranslationVector")


def read_opencv_matrix(node):
    """One ``type_id="opencv-matrix"`` node -> float64 array [rows, cols]."""
    rows = int(node.findtext("rows").strip())
    cols = int(node.findtext("cols").strip())
    dt = node.findtext("dt").strip()
    if dt not in ("d", "f"):
        raise ValueError("unsupported opencv-matrix dt %r" % dt)
    values = [float(tok) for tok in node.findtext("data").split()]
    if len(values) != rows * cols:
        raise ValueError("opencv-matrix holds %d values, expected %dx%d" % (len(values), rows, cols))
    return np.array(values, dtype=np.float64).reshape(rows, cols)


def load_xml_matrices(directory_path, filename, node_tags=NODE_TAGS):
    """Matrices of an OpenCV FileStorage XML by tag (reference utils.py:115: appends .xml)."""
    if not filename.lower().endswith(".xml"):
        filename += ".xml"
    root = ET.parse(os.path.join(directory_path, filename)).getroot()
    out = {}
    for tag in node_tags:
        node = root.find(tag)
        if node is None:
            raise KeyError("node %s not found in %s" % (tag, os.path.join(directory_path, filename)))
        out[tag] = read_opencv_matrix(node)
    return out


def rodrigues(rvec):
    """Rotation vector -> matrix with OpenCV 4.x's formula and evaluation order:
    theta = sqrt(rx^2+ry^2+rz^2); R = cos*I + (1-cos)*r r^T + sin*[r]x with r scaled by 1/theta."""
    rx, ry, rz = (float(v) for v in np.asarray(rvec, dtype=np.float64).reshape(3))
    theta = math.sqrt(rx * rx + ry * ry + rz * rz)
    if theta < np.finfo(np.float64).eps:
        return np.eye(3, dtype=np.float64)
    c, s = math.cos(theta), math.sin(theta)
    c1 = 1.0 - c
    inv = 1.0 / theta
    rx, ry, rz = rx * inv, ry * inv, rz * inv
    outer = ((rx * rx, rx * ry, rx * rz), (rx * ry, ry * ry, ry * rz), (rx * rz, ry * rz, rz * rz))
    cross = ((0.0, -rz, ry), (rz, 0.0, -rx), (-ry, rx, 0.0))
    R = np.empty((3, 3), dtype=np.float64)
    for i in range(3):
        for j in range(3):
            R[i, j] = (c * (1.0 if i == j else 0.0) + c1 * outer[i][j]) + s * cross[i][j]
    return R


@dataclass
class Camera:
    """One calibrated view: K 3x3, dist (k1,k2,p1,p2,k3), rvec 3, tvec 3 (mm), R 3x3."""
    K: np.ndarray
    dist: np.ndarray
    rvec: np.ndarray
    tvec: np.ndarray
    R: np.ndarray = None

    def __post_init__(self):
        self.K = np.asarray(self.K, dtype=np.float64).reshape(3, 3)
        d = np.asarray(self.dist, dtype=np.float64).reshape(-1)
        if d.size > 5 and np.any(d[5:] != 0):
            raise ValueError("only the 5-coefficient distortion model (k1,k2,p1,p2,k3) is supported")
        self.dist = np.concatenate([d[:5], np.zeros(max(0, 5 - d.size))])
        self.tvec = np.asarray(self.tvec, dtype=np.float64).reshape(3)
        if self.rvec is not None:
            self.rvec = np.asarray(self.rvec, dtype=np.float64).reshape(3)
        if self.R is None:
            self.R = rodrigues(self.rvec)
        self.R = np.asarray(self.R, dtype=np.float64).reshape(3, 3)

    @classmethod
    def from_config(cls, directory_path, filename="config.xml"):
        m = load_xml_matrices(directory_path, filename)
        return cls(m["CameraMatrix"], m["DistortionCoeffs"], m["RotationVector"], m["TranslationVector"])


def load_cameras(cam_input_path="data", num_cameras=4, config_input_filename="config.xml"):
    """Cameras cam1..camN under cam_input_path (reference voxel_reconstruction.py:76-78)."""
    return [Camera.from_config(os.path.join(cam_input_path, "cam" + str(c)), config_input_filename)
            for c in range(1, num_cameras + 1)]
