"""TEST INFRASTRUCTURE (checker only; never imported by the product).

Builds the REFERENCE's own Chamfer3D extension (external/chamfer3D/chamfer_cuda.cpp + chamfer3D.cu, the pybind module the reference
imports as `chamfer_3D`) for gfx950 from the sources where they lie under /root/reference, the way any PyTorch CUDA extension is
built on ROCm: torch.utils.cpp_extension hipifies the two translation units and compiles them with hipcc.  Only the resulting
shared object is kept, as oracle/_ref/chamfer_3D_ref.so (git-ignored, travels to the GPU box with the snapshot); the translated
sources live and die in a temporary directory outside the repository, nothing is written to /root/reference.

    python oracle/build_chamfer_ref.py          (no GPU needed: hipcc cross-compiles; called by __graft_entry__.build())

tests/test_gpu_chamfer_ref.py then runs the reference kernels and this build's on the same point clouds.
"""
import os
import shutil
import sys
import tempfile

REF = "/root/reference/external/chamfer3D"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
NAME = "chamfer_3D_ref"


def build(verbose=False):
    if not os.path.isdir(REF):
        return None                      # GPU box / fresh checkout without the reference: use the prebuilt file if it is there
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="sc_chamfer_ref_")
    try:
        srcs = []
        for f in ("chamfer_cuda.cpp", "chamfer3D.cu"):
            shutil.copy(os.path.join(REF, f), tmp)
            srcs.append(os.path.join(tmp, f))
        from torch.utils.cpp_extension import load
        load(name=NAME, sources=srcs, build_directory=tmp, verbose=verbose, is_python_module=False)
        so = os.path.join(tmp, NAME + ".so")
        out = os.path.join(OUT_DIR, NAME + ".so")
        shutil.copy(so, out)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def load_module():
    """The built extension as a Python module (forward / backward with the reference's signatures), or None if it was never built."""
    path = os.path.join(OUT_DIR, NAME + ".so")
    if not os.path.exists(path):
        return None
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_or_build():
    """load_module(), building the extension first when it is missing and the reference's sources are present."""
    mod = load_module()
    if mod is None and os.path.isdir(REF):
        build()
        mod = load_module()
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
