"""ctypes loader for oracle/chamfer_ref.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libchamfer_ref.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "chamfer_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libchamfer_ref.so"], stdout=subprocess.DEVNULL)
    return _SO


def _lib():
    lib = ctypes.CDLL(build())
    f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    lib.sc_ref_chamfer_forward.argtypes = [f32p, f32p, f32p, f32p, i32p, i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.sc_ref_chamfer_backward.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p, i32p, i32p,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def chamfer_forward(xyz1: np.ndarray, xyz2: np.ndarray):
    """-> dist1 [B,N] f32 (squared), dist2 [B,M], idx1 [B,N] i32, idx2 [B,M]."""
    xyz1 = np.ascontiguousarray(xyz1, np.float32)
    xyz2 = np.ascontiguousarray(xyz2, np.float32)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    d1, d2 = np.zeros((B, N), np.float32), np.zeros((B, M), np.float32)
    i1, i2 = np.zeros((B, N), np.int32), np.zeros((B, M), np.int32)
    f, i = ctypes.c_float, ctypes.c_int32
    _lib().sc_ref_chamfer_forward(_p(xyz1, f), _p(xyz2, f), _p(d1, f), _p(d2, f), _p(i1, i), _p(i2, i), B, N, M)
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, gd1, gd2, idx1, idx2):
    xyz1 = np.ascontiguousarray(xyz1, np.float32)
    xyz2 = np.ascontiguousarray(xyz2, np.float32)
    gd1 = np.ascontiguousarray(gd1, np.float32)
    gd2 = np.ascontiguousarray(gd2, np.float32)
    idx1 = np.ascontiguousarray(idx1, np.int32)
    idx2 = np.ascontiguousarray(idx2, np.int32)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    g1, g2 = np.zeros_like(xyz1), np.zeros_like(xyz2)
    f, i = ctypes.c_float, ctypes.c_int32
    _lib().sc_ref_chamfer_backward(_p(xyz1, f), _p(xyz2, f), _p(g1, f), _p(g2, f), _p(gd1, f), _p(gd2, f),
                                   _p(idx1, i), _p(idx2, i), B, N, M)
    return g1, g2
