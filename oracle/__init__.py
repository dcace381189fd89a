"""CPU oracle for the visual-hull carve path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import,
link or execute it, and only as the checker.  The product path
(``voxel-based-3d-reconstruction_amd/``) never routes through here and fails
loudly when its HIP library is missing.

PARITY UNPINNED (vs. OpenCV): the arithmetic core of the reference path is
``cv2.projectPoints`` (reference ``voxel_reconstruction.py:81``), a third-party
binary that is absent from this image (opencv-contrib-python, version unpinned
in the reference's ``requirements.txt``) and the reference has no tests or
golden vectors of its own.  The oracle therefore restates OpenCV 4.x's
published ``cvProjectPoints2Internal`` / ``Rodrigues`` formulas (float64,
separate multiply/add, left-to-right) and is pinned only by

  * the reference's own data files (``data/cam{1..4}/config.xml``,
    ``mask_MOG.jpg``) committed as fixtures under ``tests/golden/``;
  * three independent restatements that must agree bit-for-bit
    (``carve_np`` vectorised numpy, ``carve_literal`` dict/loop mirror of the
    reference's Python, ``carve_ref.c`` C/OpenMP).
"""
