"""voxcarve -- MI355X-native visual-hull voxel carving (one hot path, one drop-in).

Drop-in for the carve path of ChristosP1/Voxel-Based-3D-Reconstruction
(voxel_reconstruction.py:10-124 + assignment.py:54-149): Python host code over a ctypes
C ABI (include/voxcarve.h) onto hand-written gfx950 HIP kernels.  No PyTorch, no CPU path.

  voxcarve.engine.CarveEngine      array fast path (one context = one GPU = one z-slab)
  voxcarve.voxel_reconstruction    the reference module's call surface, names unchanged
  voxcarve.assignment              set_voxel_positions(width, height, depth)
  voxcarve.slabs                   z-slab split + survivor all-gather (RCCL over xGMI)
"""
from . import _lib, camera, engine  # noqa: F401
from .camera import Camera, load_cameras, rodrigues  # noqa: F401
from .engine import CarveEngine, DEFAULT_BOUNDS  # noqa: F401

__all__ = ["CarveEngine", "Camera", "load_cameras", "rodrigues", "DEFAULT_BOUNDS"]
