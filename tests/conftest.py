import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built artefacts are git-ignored: (re)build them when a fresh checkout runs the suite (hipcc cross-compiles
    without a GPU; `make` is a no-op when everything is up to date)."""
    lib = os.path.join(ROOT, "shapeclipper_amd", "lib", "libshapeclipper_hip.so")
    ref = os.path.join(ROOT, "oracle", "libchamfer_ref.so")
    if not (os.path.exists(lib) and os.path.exists(ref)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


# Order of the `-m gpu` suite (VERDICT r05 #1c; the driver runs `-x`): kernel-vs-oracle parity first, in the order of SURVEY.md section 8's
# rows; whole-step tests next; tests that start other processes (bench.py launches, multi-rank runs sharing the one GPU: timing- and
# contention-dependent) LAST, so that a failure there can never hide a parity test.  Files not named keep their alphabetical place
# in the middle group.
_FIRST = ["test_gpu_sdf.py", "test_gpu_sdf_backward.py", "test_gpu_camera.py", "test_gpu_sampling.py", "test_gpu_render_train.py",
          "test_gpu_render_eval.py", "test_gpu_render_hits.py", "test_gpu_render_cabi.py", "test_gpu_rgb_stash.py", "test_gpu_parity_large.py",
          "test_gpu_arch_variants.py", "test_gpu_other_architectures.py", "test_gpu_weight_norm.py", "test_gpu_loss.py",
          "test_gpu_camera_prior.py", "test_gpu_full_step_parity.py", "test_gpu_chamfer.py", "test_gpu_chamfer_grid.py",
          "test_gpu_chamfer_ref.py", "test_gpu_eval.py", "test_gpu_clip.py", "test_gpu_conv.py", "test_gpu_bn.py", "test_gpu_block.py",
          "test_gpu_fused_block.py", "test_gpu_bottleneck.py", "test_gpu_isosurface.py"]
_LAST = ["test_gpu_train_step.py", "test_gpu_determinism.py", "test_gpu_clip_anno.py", "test_gpu_two_ranks.py", "test_gpu_bench_contract.py"]


def suite_order_key(filename):
    """(group, position) of a test file: 0 = parity kernels in section-8 order, 1 = everything else alphabetically, 2 = subprocess tests."""
    if filename in _FIRST:
        return (0, _FIRST.index(filename), "")
    if filename in _LAST:
        return (2, _LAST.index(filename), "")
    return (1, 0, filename)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=lambda it: suite_order_key(os.path.basename(str(it.fspath))))        # stable: the order inside a file is kept
