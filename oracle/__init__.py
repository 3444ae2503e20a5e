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
(``tests/golden/make_golden.py``).  Two pieces are restated from source that
cannot run here and are therefore pinned only by known-answer vectors:

  * Chamfer3D (``external/chamfer3D/chamfer3D.cu`` needs nvcc + a GPU) --
    ``chamfer_ref.c`` follows the kernel line by line (fp32, fma contraction
    as nvcc's default ``-fmad=true``, strict ``<`` lowest-index tie rule);
  * ``utils/eval_3D.py`` metric helpers (module imports mcubes/trimesh, absent
    here) -- restated from the file and checked with sys.modules stubs.

CLIP ViT (third-party openai/CLIP, un-vendored, un-pinned, weights absent):
"parity unpinned" -- architecture-level oracle only (see clip_ref.py).
"""
