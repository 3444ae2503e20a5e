"""The C-ABI library builds, loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import os
import re

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
