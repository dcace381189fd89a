"""Carves the reference's own scene (4 calibrated cameras + frame-0 MOG masks, committed fixtures) on the
GPU and writes the visual hull as a coloured point cloud (PLY) -- what the reference hands to its OpenGL
viewer after `G` is pressed.   python scripts/demo.py [grid=128] [out=hull.ply]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fixtures_util as fx
from voxcarve import assignment

g = int(sys.argv[1]) if len(sys.argv) > 1 else 128
out = sys.argv[2] if len(sys.argv) > 2 else "hull.ply"
masks = fx.golden_masks()
frames = [np.dstack([m // 2 + 60, m // 3 + 40, 255 - m // 2]).astype(np.uint8) for m in masks]   # any BGR image
assignment.configure(frame_source=assignment.StaticFrameSource([(frames, masks)]),
                     data_path=os.path.join(fx.GOLDEN, "data"))
pos, col = assignment.set_voxel_positions(g, g // 2, g)          # the reference's call: (width, height, depth)
rgb = (col * 255.0 + 0.5).astype(np.uint8)
with open(out, "w") as f:
    f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
            "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(pos))
    for p, c in zip(pos, rgb):
        f.write("%g %g %g %d %d %d\n" % (p[0], p[1], p[2], c[0], c[1], c[2]))
print("%d voxels of the %dx%dx%d grid survive all 4 views -> %s; extent x %.2f..%.2f, y %.2f..%.2f, z %.2f..%.2f" %
      (len(pos), g, g, g, out, pos[:, 0].min(), pos[:, 0].max(), pos[:, 1].min(), pos[:, 1].max(), pos[:, 2].min(), pos[:, 2].max()))
