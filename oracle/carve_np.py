"""Vectorised numpy restatement of the reference carve path (float64, no FMA).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py`` (parity unpinned vs cv2).

Each function cites the reference lines it follows (paths relative to
``/root/reference``).  numpy evaluates every ``*`` and ``+`` as its own ufunc,
so no multiply-add is ever contracted: the evaluation order written here IS the
rounding order.
"""
import re

import numpy as np

# Default carve volume, reference voxel_reconstruction.py:35-36 (millimetres).
DEFAULT_BOUNDS = (-512.0, 1024.0, -1024.0, 1024.0, -2048.0, 512.0)
SCALING_FACTOR = 64          # reference assignment.py:118
VIEWS_THRESHOLD = 4          # reference assignment.py:119
COLOR_CAMERA_KEY = 2         # reference assignment.py:133 (1-based camera key)


# --------------------------------------------------------------------------- a-1
def read_config_xml(path):
    """Camera parameters of one ``config.xml``.

    Follows voxel_reconstruction.py:10-32 / utils.py:115-152: nodes
    CameraMatrix (3x3), DistortionCoeffs (1x5), RotationVector (3x1),
    TranslationVector (3x1), all ``<dt>d</dt>`` row-major.  cv2.FileStorage is
    replaced by a regex scan; Python's float() is correctly rounded like
    OpenCV's strtod, so the float64 values are the same bits.
    """
    text = open(path, "r").read()
    out = {}
    for tag in ("CameraMatrix", "DistortionCoeffs", "RotationVector", "TranslationVector"):
        m = re.search(r"<%s[^>]*>(.*?)</%s>" % (tag, tag), text, re.S)
        if m is None:
            raise ValueError("node %s missing in %s" % (tag, path))
        body = m.group(1)
        rows = int(re.search(r"<rows>\s*(\d+)\s*</rows>", body).group(1))
        cols = int(re.search(r"<cols>\s*(\d+)\s*</cols>", body).group(1))
        data = re.search(r"<data>(.*?)</data>", body, re.S).group(1).split()
        vals = np.array([float(tok) for tok in data], dtype=np.float64)
        out[tag] = vals.reshape(rows, cols)
    return out["CameraMatrix"], out["DistortionCoeffs"], out["RotationVector"], out["TranslationVector"]


# --------------------------------------------------------------------------- Rodrigues
def rodrigues(rvec):
    """Rotation vector -> 3x3 matrix, OpenCV 4.x ``cv::Rodrigues`` formula.

    theta = sqrt(rx*rx + ry*ry + rz*rz); theta < DBL_EPSILON -> identity;
    otherwise c = cos, s = sin, c1 = 1 - c, r *= 1/theta and
    R = c*I + c1*r*r^T + s*[r]x, element by element, left to right.
    Called implicitly by cv2.projectPoints (voxel_reconstruction.py:81).
    """
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    rx, ry, rz = float(r[0]), float(r[1]), float(r[2])
    theta = np.sqrt(np.float64(rx * rx + ry * ry + rz * rz))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3, dtype=np.float64)
    c = np.cos(theta)
    s = np.sin(theta)
    c1 = 1.0 - c
    itheta = 1.0 / theta
    rx, ry, rz = rx * itheta, ry * itheta, rz * itheta
    rrt = np.array([[rx * rx, rx * ry, rx * rz],
                    [rx * ry, ry * ry, ry * rz],
                    [rx * rz, ry * rz, rz * rz]], dtype=np.float64)
    r_x = np.array([[0.0, -rz, ry],
                    [rz, 0.0, -rx],
                    [-ry, rx, 0.0]], dtype=np.float64)
    eye = np.eye(3, dtype=np.float64)
    return (c * eye + c1 * rrt) + s * r_x


# --------------------------------------------------------------------------- a-2
def axis_tables(nx, ny, nz, bounds=DEFAULT_BOUNDS):
    """The three ``np.linspace`` axes of voxel_reconstruction.py:52-54."""
    x0, x1, y0, y1, z0, z1 = bounds
    return (np.linspace(x0, x1, num=nx), np.linspace(y0, y1, num=ny), np.linspace(z0, z1, num=nz))


def create_voxel_volume(nx=128, ny=128, nz=128, bounds=DEFAULT_BOUNDS):
    """Voxel centres float64 [N,3], voxel_reconstruction.py:35-59 (literal form).

    Row i = iz*nx*ny + ix*ny + iy holds (x[ix], y[iy], z[iz]).
    """
    xs, ys, zs = axis_tables(nx, ny, nz, bounds)
    return np.array(np.meshgrid(xs, ys, zs)).T.reshape(-1, 3)


def points_of_indices(idx, nx, ny, nz, bounds=DEFAULT_BOUNDS):
    """Voxel centres of linear indices ``idx`` without materialising the volume."""
    xs, ys, zs = axis_tables(nx, ny, nz, bounds)
    idx = np.asarray(idx, dtype=np.int64)
    iy = idx % ny
    t = idx // ny
    ix = t % nx
    iz = t // nx
    return np.stack([xs[ix], ys[iy], zs[iz]], axis=1)


# --------------------------------------------------------------------------- a-3
def project_points(points, R, tvec, K, dist):
    """OpenCV 4.x ``cvProjectPoints2Internal`` in float64 (voxel_reconstruction.py:81).

    Written with all 14 distortion slots and the tilt stage present but zero /
    identity, exactly as OpenCV evaluates them for a 5-coefficient model, so the
    shortened forms used by the C oracle and the HIP kernels can be checked
    against this one (they differ only where a value is already non-finite).
    Returns float64 [N,2] (u, v).
    """
    P = np.asarray(points, dtype=np.float64)
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    t = np.asarray(tvec, dtype=np.float64).reshape(3)
    A = np.asarray(K, dtype=np.float64).reshape(3, 3)
    k = np.zeros(14, dtype=np.float64)
    d = np.asarray(dist, dtype=np.float64).reshape(-1)
    k[:d.size] = d
    fx, fy, cx, cy = A[0, 0], A[1, 1], A[0, 2], A[1, 2]
    X, Y, Z = P[:, 0], P[:, 1], P[:, 2]
    with np.errstate(all="ignore"):
        x = R[0, 0] * X + R[0, 1] * Y + R[0, 2] * Z + t[0]
        y = R[1, 0] * X + R[1, 1] * Y + R[1, 2] * Z + t[1]
        z = R[2, 0] * X + R[2, 1] * Y + R[2, 2] * Z + t[2]
        z = np.where(z != 0.0, 1.0 / z, 1.0)           # z = z ? 1./z : 1
        x = x * z
        y = y * z
        r2 = x * x + y * y
        r4 = r2 * r2
        r6 = r4 * r2
        a1 = 2 * x * y
        a2 = r2 + 2 * x * x
        a3 = r2 + 2 * y * y
        cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6
        icdist2 = 1.0 / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6)
        xd0 = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4
        yd0 = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4
        # tilt stage with matTilt = I (k[12] = k[13] = 0): Matx33d * Vec3d(xd0, yd0, 1)
        one = np.ones_like(xd0)
        vt0 = ((0.0 + 1.0 * xd0) + 0.0 * yd0) + 0.0 * one
        vt1 = ((0.0 + 0.0 * xd0) + 1.0 * yd0) + 0.0 * one
        vt2 = ((0.0 + 0.0 * xd0) + 0.0 * yd0) + 1.0 * one
        inv_proj = np.where(vt2 != 0.0, 1.0 / vt2, 1.0)
        xd = inv_proj * vt0
        yd = inv_proj * vt1
        u = xd * fx + cx
        v = yd * fy + cy
    return np.stack([u, v], axis=1)


def pixel_offsets(uv, H, W):
    """Packed LUT entry per voxel-view: ``int(v)*W + int(u)`` or -1 when outside.

    Lossless for the test at voxel_reconstruction.py:110-112: the bounds test is
    on the float coordinates (NaN and (-1,0) are outside), int() truncates.
    """
    u, v = uv[:, 0], uv[:, 1]
    with np.errstate(invalid="ignore"):
        inside = (0 <= v) & (v < H) & (0 <= u) & (u < W)
    off = np.full(u.shape, -1, dtype=np.int32)
    ui = u[inside].astype(np.int64)
    vi = v[inside].astype(np.int64)
    off[inside] = (vi * W + ui).astype(np.int32)
    return off


# --------------------------------------------------------------------------- a-4 / a-5
def carve(nx, ny, nz, cams, masks, frames=None, bounds=DEFAULT_BOUNDS, min_views=None,
          color_cam=COLOR_CAMERA_KEY - 1, index_range=None, chunk=1 << 20):
    """Visual hull over the grid, chunked so 256^3 fits in memory.

    cams: list of (K, dist, R, tvec).  masks: list of uint8 [H,W] (foreground > 0).
    frames: list of uint8 [H,W,3] BGR or None.  min_views defaults to len(cams)
    (reference VIEWS_THRESHOLD = 4 with 4 cameras, assignment.py:119-122).
    index_range = (i0, i1) restricts to a contiguous linear-index range (a z-slab).

    Follows voxel_reconstruction.py:105-122 and assignment.py:116-133.  Returns
    dict with
      idx      uint32 [S]  ascending linear voxel indices of the survivors,
      viewmask uint16 [n]  per-voxel bitmask of cameras that see it (bit c),
      offsets  int32 [C,n] packed LUT (pixel offset or -1),
      bgr      uint8 [S,3] colour-camera BGR sample of each survivor (if frames).
    """
    C = len(cams)
    if min_views is None:
        min_views = C
    N = nx * ny * nz
    i0, i1 = (0, N) if index_range is None else index_range
    n = i1 - i0
    H, W = masks[0].shape
    viewmask = np.zeros(n, dtype=np.uint16)
    offsets = np.empty((C, n), dtype=np.int32)
    for s in range(i0, i1, chunk):
        e = min(s + chunk, i1)
        pts = points_of_indices(np.arange(s, e, dtype=np.int64), nx, ny, nz, bounds)
        for c, (K, dist, R, t) in enumerate(cams):
            off = pixel_offsets(project_points(pts, R, t, K, dist), H, W)
            offsets[c, s - i0:e - i0] = off
            seen = np.zeros(e - s, dtype=bool)
            ok = off >= 0
            seen[ok] = masks[c].reshape(-1)[off[ok]] > 0
            viewmask[s - i0:e - i0] |= (seen.astype(np.uint16) << c)
    nviews = np.zeros(n, dtype=np.int32)
    for c in range(C):
        nviews += (viewmask >> c) & 1
    # A voxel enters voxels_visible only if >=1 camera sees it; sum(views) >= threshold.
    keep = (nviews >= min_views) & (nviews >= 1)
    local = np.nonzero(keep)[0]
    idx = (local + i0).astype(np.uint32)
    out = {"idx": idx, "viewmask": viewmask, "offsets": offsets}
    if frames is not None:
        # voxels_visible_colors[voxel][2] exists only when the colour camera sees the
        # voxel (voxel_reconstruction.py:119-122); the reference would raise KeyError
        # otherwise, so survivors not seen by it get 0,0,0 here and are flagged.
        seen_cc = ((viewmask[local] >> color_cam) & 1).astype(bool)
        bgr = np.zeros((local.size, 3), dtype=np.uint8)
        offc = offsets[color_cam, local]
        bgr[seen_cc] = frames[color_cam].reshape(-1, 3)[offc[seen_cc]]
        out["bgr"] = bgr
        out["color_seen"] = seen_cc
    return out


def voxel_keys(idx, nx, ny, nz, bounds=DEFAULT_BOUNDS):
    """``tuple(map(int, voxel))`` of voxel_reconstruction.py:84 as int64 [S,3] (trunc toward 0)."""
    return np.trunc(points_of_indices(idx, nx, ny, nz, bounds)).astype(np.int64)


def viewer_positions(keys):
    """assignment.py:127-130: [vx/64, -(vz/64), vy/64] as float64 [S,3]."""
    k = np.asarray(keys, dtype=np.int64)
    return np.stack([k[:, 0] / SCALING_FACTOR, -(k[:, 2] / SCALING_FACTOR), k[:, 1] / SCALING_FACTOR], axis=1)


def viewer_colors(bgr):
    """assignment.py:133: BGR uint8 -> RGB float64 in [0,1]."""
    return np.asarray(bgr)[:, ::-1] / 255.0
