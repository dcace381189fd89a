"""CPU oracle for the visual-hull carve path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import,
link or execute it, and only as the checker.  The product path
(``voxel-based-3d-reconstruction_amd/``) never routes through here and fails
loudly when its HIP library is missing.

PARITY PINNED AT PIXEL LEVEL, UNPINNED IN THE LAST ULP (vs. OpenCV): the arithmetic core of the reference path is
``cv2.projectPoints`` (reference ``voxel_reconstruction.py:81``), a third-party
binary that is absent from this image (opencv-contrib-python, version unpinned
in the reference's ``requirements.txt``) and the reference has no tests or
golden vectors of its own.  ONE artefact of the reference carries real
``cv2.projectPoints`` output: the three axis arrows its calibration script drew
into ``data/cam{1..4}/test.jpg`` (``camera_calibration.py:753-789``, ``:847-849``,
``:967-974``) with exactly the parameters saved to ``config.xml``.  Their 12 tips,
measured once from the JPEGs by ``tests/golden/make_arrow_tips.py`` (Pillow, no
oracle code involved) and committed as ``tests/golden/arrow_tips.json``, are hit
by every restatement here and by the device projection within 2 px
(``tests/test_oracle.py::test_oracle_pinned_by_reference_arrow_tips``,
``tests/test_gpu_parity.py::test_device_projection_hits_reference_arrow_tips``);
eight plausible misreadings of the conventions (R transposed, axis order, sign
of z, distortion dropped or k1 negated, ...) all miss by more than 4 px.  That
pins K / distortion / Rodrigues / translation conventions and the axis order;
it cannot pin the last ulp of the float64 evaluation (which decides
``int(x)`` only within ~1e-13 px of a pixel boundary).  Beyond that the oracle restates OpenCV 4.x's
published ``cvProjectPoints2Internal`` / ``Rodrigues`` formulas (float64,
separate multiply/add, left-to-right) and is pinned only by

  * the reference's own data files (``data/cam{1..4}/config.xml``,
    ``mask_MOG.jpg``) committed as fixtures under ``tests/golden/``;
  * three independent restatements that must agree bit-for-bit
    (``carve_np`` vectorised numpy, ``carve_literal`` dict/loop mirror of the
    reference's Python, ``carve_ref.c`` C/OpenMP).
"""
