"""Drop-in for the reference module ``voxel_reconstruction`` (carve functions only).

Same function names, parameters, defaults and return shapes as
voxel_reconstruction.py:10-124 of the reference, so ``assignment.py`` can call them
unchanged.  The point array and the lookup table become light handles (the reference
only passes them back in); the visibility dictionaries are real dicts, materialised from
the GPU's per-voxel camera bitmask ("compat" surface -- sized for the reference's 128^3;
use ``voxcarve.engine.CarveEngine`` / ``voxcarve.assignment`` for large grids).
``plot_marching_cubes`` (reference :127-163) runs the marching cubes on the device instead of in scikit-image.
"""
import os

import numpy as np

from .camera import Camera, load_xml_matrices
from .engine import CarveEngine


def load_config_info(config_info_path="data/cam", config_input_filename="config.xml"):
    """(mtx 3x3, dist 1x5, rvecs 3x1, tvecs 3x1) of one camera; reference :10-32."""
    m = load_xml_matrices(config_info_path, config_input_filename)
    return m["CameraMatrix"], m["DistortionCoeffs"], m["RotationVector"], m["TranslationVector"]


class VoxelVolume:
    """Stands for the float64 [N,3] array of reference :52-57 without allocating it.

    Row i = iz*nx*ny + ix*ny + iy is (x[ix], y[iy], z[iz]); np.asarray(volume) builds the
    real array on demand (small grids)."""

    def __init__(self, nx, ny, nz, bounds):
        self.shape_xyz = (int(nx), int(ny), int(nz))
        self.bounds = tuple(float(b) for b in bounds)

    def __len__(self):
        nx, ny, nz = self.shape_xyz
        return nx * ny * nz

    @property
    def shape(self):
        return (len(self), 3)

    def axes(self):
        b = self.bounds
        nx, ny, nz = self.shape_xyz
        return (np.linspace(b[0], b[1], num=nx), np.linspace(b[2], b[3], num=ny), np.linspace(b[4], b[5], num=nz))

    def __array__(self, dtype=None, copy=None):
        xs, ys, zs = self.axes()
        pts = np.array(np.meshgrid(xs, ys, zs)).T.reshape(-1, 3)
        return pts if dtype is None else pts.astype(dtype)


def create_voxel_volume(num_voxels_x=128, num_voxels_y=128, num_voxels_z=128, x_min=-512, x_max=1024,
                        y_min=-1024, y_max=1024, z_min=-2048, z_max=512):
    """Voxel volume handle; reference :35-59."""
    return VoxelVolume(num_voxels_x, num_voxels_y, num_voxels_z, (x_min, x_max, y_min, y_max, z_min, z_max))


class LookupTable:
    """Stands for the dict of reference :74-86: owns the GPU context with grid and cameras.

    The reference stores float pixel coordinates and applies the image-size test per call
    (:110), so the packed device table is (re)built when the mask size is first seen."""

    def __init__(self, volume, cameras, device=0):
        self.volume = volume
        self.cameras = cameras
        self.engine = CarveEngine(device)
        nx, ny, nz = volume.shape_xyz
        self.engine.set_grid(nx, ny, nz, volume.bounds)
        self._size = None

    def __len__(self):
        return len(self.cameras)

    def keys(self):
        return range(1, len(self.cameras) + 1)

    def prepare(self, H, W):
        if self._size != (H, W):
            self.engine.set_cameras(self.cameras, H, W)
            self.engine.build_lut()
            self._size = (H, W)
        return self.engine


def create_lookup_table(voxel_points, num_cameras, cam_input_path="data", config_input_filename="config.xml"):
    """Lookup-table handle for cameras cam1..camN; reference :62-86."""
    if not isinstance(voxel_points, VoxelVolume):
        raise TypeError("voxel_points must come from this module's create_voxel_volume()")
    cams = [Camera(*load_config_info(os.path.join(cam_input_path, "cam" + str(c)), config_input_filename))
            for c in range(1, num_cameras + 1)]
    return LookupTable(voxel_points, cams)


def update_visible_voxels_and_extract_colors(lookup_table, fg_masks, images):
    """(voxels_visible, voxels_visible_colors) as in reference :89-124.

    voxels_visible[voxel][camera] = True and voxels_visible_colors[voxel][camera] =
    np.array(BGR) for every camera (1-based) that sees voxel = tuple of truncated ints;
    insertion order as the reference: camera by camera, voxels in table order."""
    H, W = np.asarray(fg_masks[0]).shape[:2]
    eng = lookup_table.prepare(H, W)
    eng.upload_masks(fg_masks, slot=0)
    eng.carve(slot=0, min_views=1, color_cam=None, mode="lut", viewmask=True)
    viewmask = eng.fetch_viewmask()
    nx, ny, nz = lookup_table.volume.shape_xyz
    xs, ys, zs = lookup_table.volume.axes()
    voxels_visible, voxels_visible_colors = {}, {}
    for cam_key in lookup_table.keys():
        c = cam_key - 1
        sel = np.nonzero((viewmask >> c) & 1)[0]
        if sel.size == 0:
            continue
        off = eng.fetch_lut(c)[sel]
        bgr = np.asarray(images[c]).reshape(-1, 3)[off]
        iy = sel % ny
        t = sel // ny
        kx = np.trunc(xs[t % nx]).astype(np.int64).tolist()
        ky = np.trunc(ys[iy]).astype(np.int64).tolist()
        kz = np.trunc(zs[t // nx]).astype(np.int64).tolist()
        for k in range(sel.size):
            voxel = (kx[k], ky[k], kz[k])
            views = voxels_visible.get(voxel)
            if views is None:
                voxels_visible[voxel] = views = {}
                voxels_visible_colors[voxel] = {}
            views[cam_key] = True
            voxels_visible_colors[voxel][cam_key] = np.array(bgr[k])
    return voxels_visible, voxels_visible_colors


def marching_cubes(voxels_status, level=0, device=0):
    """(verts float32 [V, 3], faces uint32 [F, 3]) of a 3-D ON/OFF array: the device's counterpart of the
    ``measure.marching_cubes(voxels_status, 0)`` call of reference :141 (classic table-driven marching cubes; vertex
    coordinates in array-index space like skimage's; parity with skimage's Lewiner variant is unpinned)."""
    with CarveEngine(device) as eng:
        return eng.marching_cubes(np.asarray(voxels_status), level=level)


def plot_marching_cubes(voxels_status, rotate=True, plot_output_path="plots", plot_output_filename="marching_cubes.png"):
    """Runs marching cubes on activated voxels and plots the result; reference :127-163, same parameters.

    The mesh comes from the GPU (``marching_cubes`` above); the figure is drawn as the reference draws it."""
    if rotate:
        voxels_status = np.rot90(voxels_status, 2)                   # reference :138-139
    verts, faces = marching_cubes(voxels_status, 0)
    # drawn on a Figure of its own with the Agg canvas: the caller's global matplotlib backend is left alone
    from matplotlib.backends.backend_agg import FigureCanvasAgg
    from matplotlib.figure import Figure
    from mpl_toolkits.mplot3d.art3d import Poly3DCollection
    fig = Figure(figsize=(10, 10))
    FigureCanvasAgg(fig)
    ax = fig.add_subplot(111, projection="3d")
    surface = Poly3DCollection(verts[faces.astype(np.int64)], edgecolor="k")
    ax.add_collection3d(surface)
    d0, d1, d2 = voxels_status.shape
    ax.set(xlabel="X", ylabel="Y", zlabel="z-axis", xlim=(0, d2), ylim=(0, d1), zlim=(0, d0))
    fig.tight_layout()
    os.makedirs(plot_output_path, exist_ok=True)
    fig.savefig(os.path.join(plot_output_path, plot_output_filename))
    return verts, faces
