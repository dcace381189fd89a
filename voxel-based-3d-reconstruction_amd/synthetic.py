"""Deterministic synthetic cameras, masks and frames.

Used by bench.py and the large-size property tests: there is no decoder for the
reference's videos here, and BASELINE.json's larger configurations (16 cameras at 1080p)
have no real data at all.  Cameras sit on a ring around the volume centre looking at it,
with a real-camera-like distortion; mask c = pixels whose ray hits an ellipsoid at the
centre, XOR 0.5 % salt noise; frames are seeded random BGR.
"""
import math

import numpy as np

from .camera import Camera

VOLUME_CENTRE = (256.0, 0.0, -768.0)          # centre of the reference's default bounds
ELLIPSOID_RADII = (300.0, 250.0, 800.0)
DIST = (-0.36, 0.19, 2e-4, 2e-4, -0.06)


def ring_cameras(n_cameras, H, W, radius=4000.0, elevation_deg=20.0, centre=VOLUME_CENTRE):
    if (H, W) == (1080, 1920):
        f, cx, cy = 1500.0, 960.0, 540.0
    else:
        f, cx, cy = 0.78 * W, W / 2.0, H / 2.0
    K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1.0]])
    ctr = np.asarray(centre, dtype=np.float64)
    cams = []
    for c in range(n_cameras):
        az = 2.0 * math.pi * c / n_cameras
        el = math.radians(elevation_deg if c % 2 == 0 else -elevation_deg)
        # world "up" of the reference's volume is -z (z runs -2048..512 below the floor plane)
        pos = ctr + radius * np.array([math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), -math.sin(el)])
        fwd = ctr - pos
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, -1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])            # world -> camera rows
        t = -R @ pos
        cams.append(Camera(K.copy(), np.array(DIST), None, t, R=R))   # R given directly, no rvec
    return cams


def ellipsoid_masks(cams, H, W, radii=ELLIPSOID_RADII, centre=VOLUME_CENTRE, noise=0.005, seed=1000):
    """uint8 {0,255} masks: undistorted pixel rays against the ellipsoid, XOR salt noise."""
    ctr = np.asarray(centre, dtype=np.float64)
    inv_r = 1.0 / np.asarray(radii, dtype=np.float64)
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    masks = []
    for c, cam in enumerate(cams):
        fx, fy, cx, cy = cam.K[0, 0], cam.K[1, 1], cam.K[0, 2], cam.K[1, 2]
        xd, yd = (u + 0.5 - cx) / fx, (v + 0.5 - cy) / fy
        x, y = xd.copy(), yd.copy()
        k1, k2, p1, p2, k3 = cam.dist
        for _ in range(8):                                   # fixed-point undistortion
            r2 = x * x + y * y
            cd = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (xd - dx) / cd, (yd - dy) / cd
        d_cam = np.stack([x, y, np.ones_like(x)], axis=-1)
        d = (d_cam @ cam.R) * inv_r                           # R^T d, scaled to the unit sphere
        o = ((-cam.R.T @ cam.tvec) - ctr) * inv_r
        a = (d * d).sum(-1)
        b = 2.0 * (d * o).sum(-1)
        cc = float((o * o).sum()) - 1.0
        hit = (b * b - 4 * a * cc) >= 0
        salt = np.random.default_rng(seed + c).random((H, W)) < noise
        masks.append(np.where(hit ^ salt, 255, 0).astype(np.uint8))
    return masks


def random_frames(n_cameras, H, W, seed=2000):
    return [np.random.default_rng(seed + c).integers(0, 256, (H, W, 3), dtype=np.uint8) for c in range(n_cameras)]


def shifted_masks(masks, step):
    """A different but equally sized workload per step: rotate every mask by `step` columns."""
    return [np.roll(m, 3 * step, axis=1) for m in masks]
