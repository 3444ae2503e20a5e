"""The C-ABI library builds, loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "shapeclipper_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|long long) (sc_\w+)\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from shapeclipper_amd import _lib
    names = declared_symbols()
    assert len(names) >= 9
    assert os.path.exists(_lib.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    cdll = ctypes.CDLL(_lib.LIB_PATH)          # loads without a GPU; no compute calls here
    for n in names:
        assert hasattr(cdll, n), n
    assert sorted(_lib.SYMBOLS + _lib.SYMBOLS_OTHER) == names, (sorted(_lib.SYMBOLS + _lib.SYMBOLS_OTHER), names)


def test_product_fails_loudly_without_device_tensors():
    import pytest
    import torch
    from shapeclipper_amd import ops, packing
    from oracle import reference_ops as R
    W = R.init_sdf_weights(R.Cfg(), 0)
    pack, cb = packing.pack_sdf(W, torch.zeros(1, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.sdf_forward(torch.zeros(4, 3), pack, cb, 4)


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "shapeclipper_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r"#.*", "", src).replace("oracle/", ""), os.path.join(dirpath, f)
    for f in ("chamfer_3D.py", "train.py", "evaluate.py", "pretrain.py"):
        assert "import oracle" not in open(os.path.join(ROOT, f)).read() and "from oracle" not in open(os.path.join(ROOT, f)).read()


def test_device_guard_raises_when_a_tensor_is_not_on_the_current_device(monkeypatch):
    """Kernels are launched on the CURRENT device's stream (shapeclipper_amd/_lib.py): every entry point passes its tensors through ptr()
    and then asks for the stream, which refuses a tensor of another device (rank != device index in a multi-GPU process).  No GPU here:
    the device queries are replaced, the tensor is a stand-in with the four methods ptr() uses."""
    import torch
    from shapeclipper_amd import _lib

    class FakeCudaTensor:
        is_cuda = True
        def __init__(self, dev): self._dev = dev
        def is_contiguous(self): return True
        def get_device(self): return self._dev
        def data_ptr(self): return 4096

    monkeypatch.setattr(torch._C, "_cuda_getCurrentRawStream", lambda dev: 1234 + dev, raising=False)
    monkeypatch.setattr(torch._C, "_cuda_getDevice", lambda: 1, raising=False)
    assert _lib.ptr(FakeCudaTensor(1)).value == 4096
    assert _lib.raw_stream() == 1235                       # tensor on cuda:1, current device cuda:1: the stream of device 1
    _lib.ptr(FakeCudaTensor(0))
    with pytest.raises(RuntimeError, match="current device is cuda:1"):
        _lib.stream()
    #  the remembered device is consumed by the check: an unrelated later call is not judged by the stale cuda:0 tensor ...
    assert _lib.raw_stream() == 1235
    # ... callers that take raw data_ptr()s name the device themselves (BatchNorm / block / workspace paths) ...
    assert _lib.raw_stream(1) == 1235
    with pytest.raises(RuntimeError, match="tensor on cuda:0"):
        _lib.raw_stream(0)
    # ... and the memory is per thread (autograd and side-stream threads launch concurrently)
    import threading
    _lib.ptr(FakeCudaTensor(0))
    seen = []
    t = threading.Thread(target=lambda: seen.append(_lib.raw_stream()))
    t.start(); t.join()
    assert seen == [1235]
    with pytest.raises(RuntimeError):
        _lib.raw_stream()
