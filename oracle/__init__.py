"""oracle/ -- CPU restatement of the ShapeClipper hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *checker*, never the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * nothing under ``shapeclipper_amd/`` (the product) imports it -- the product
    path raises when the HIP library is missing instead of falling back here.

Parity status
-------------
Pinned.  Every function in ``reference_ops.py`` was checked in the build
container against the reference's own modules imported from ``/root/reference``
(``model/implicit.py``, ``model/renderer.py``, ``model/loss.py``,
``utils/camera.py``); the frozen input/output vectors live in
``tests/golden/*.npz`` next to the script that made them
(``tests/golden/make_golden.py``).

  * Chamfer3D: ``build_chamfer_ref.py`` builds the reference's OWN extension
    (``external/chamfer3D``: chamfer_cuda.cpp + chamfer3D.cu, untouched) for
    gfx950 into ``oracle/_ref/chamfer_3D_ref.so``; ``chamfer_ref.c`` (fp32,
    d = fmaf(y, y, x*x) + z*z, strict ``<`` lowest-index tie rule) and the HIP
    kernels agree with it bit for bit (``tests/test_gpu_chamfer_ref.py``);
  * ``utils/eval_3D.py`` metric helpers (module imports mcubes/trimesh, absent
    here) -- restated from the file and checked with sys.modules stubs.

CLIP ViT (third-party openai/CLIP, un-vendored, un-pinned, weights absent):
"parity unpinned" -- architecture-level oracle only (see clip_ref.py).
"""
