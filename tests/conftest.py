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
