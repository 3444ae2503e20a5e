"""chamfer_3D -- drop-in for the reference's pybind extension of the same name.

Same two entry points and argument meaning as external/chamfer3D/chamfer_cuda.cpp:17-33:

    forward(xyz1, xyz2, dist1, dist2, idx1, idx2) -> int
    backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2) -> int

The caller allocates every output (zero-filled; the backward *accumulates* into the gradients),
tensors are fp32 / int32, contiguous, on one ROCm device.  Returns 1 on success like the
reference (chamfer3D.cu:151,193); unlike the reference a failed launch raises instead of
printf-and-return-0, and kernels go to torch's current stream rather than the legacy default one.
The work is done by the hand-written gfx950 kernels in shapeclipper_amd/csrc/chamfer.hip (all pairs) and
chamfer_grid.hip (exact uniform-grid search, same results bit for bit, used for clouds of GRID_MIN_POINTS points or more)
through the C ABI (include/shapeclipper_hip.h); there is no CPU fallback.  SEARCH = "brute" (or the environment
variable SHAPECLIPPER_CHAMFER_SEARCH=brute) keeps every call on the all-pairs kernels.
"""
import ctypes
import os

import torch

from shapeclipper_amd import _lib


SEARCH = os.environ.get("SHAPECLIPPER_CHAMFER_SEARCH", "grid")
GRID_MIN_POINTS = 2048
NSPLIT = int(os.environ.get("SHAPECLIPPER_CHAMFER_NSPLIT", "0"))      # all pairs: target slices per cloud (0: chosen by the library)


def _dims(xyz1, xyz2):
    assert xyz1.dim() == 3 and xyz2.dim() == 3 and xyz1.shape[2] == 3 and xyz2.shape[2] == 3
    assert xyz1.shape[0] == xyz2.shape[0]
    return xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]


def _check(dev, **tensors):
    """The C ABI takes raw pointers: a wrong dtype / shape / device would silently corrupt memory.  The reference
    extension raises on a dtype mismatch (ATen accessor checks, chamfer3D.cu:144-149); so does this."""
    for name, (t, dtype, shape) in tensors.items():
        if t.dtype != dtype:
            raise TypeError("chamfer_3D: %s must be %s, got %s" % (name, dtype, t.dtype))
        if tuple(t.shape) != tuple(shape):
            raise ValueError("chamfer_3D: %s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
        if t.device != dev:
            raise ValueError("chamfer_3D: %s is on %s, xyz1 is on %s" % (name, t.device, dev))
        if not t.is_contiguous():
            raise ValueError("chamfer_3D: %s must be contiguous" % name)


def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    lib = _lib.load()
    b, n, m = _dims(xyz1, xyz2)
    f32, i32 = torch.float32, torch.int32
    _check(xyz1.device, xyz1=(xyz1, f32, (b, n, 3)), xyz2=(xyz2, f32, (b, m, 3)), dist1=(dist1, f32, (b, n)),
           dist2=(dist2, f32, (b, m)), idx1=(idx1, i32, (b, n)), idx2=(idx2, i32, (b, m)))
    with torch.cuda.device(xyz1.device):
        return _forward(lib, b, n, m, xyz1, xyz2, dist1, dist2, idx1, idx2)


def _forward(lib, b, n, m, xyz1, xyz2, dist1, dist2, idx1, idx2):
    if SEARCH not in ("grid", "brute"):
        raise ValueError("chamfer_3D.SEARCH must be 'grid' or 'brute', got %r" % (SEARCH,))
    if SEARCH == "grid" and min(n, m) >= GRID_MIN_POINTS:
        ws = torch.empty(int(lib.sc_chamfer3d_grid_workspace_bytes(ctypes.c_int(b), ctypes.c_int(n), ctypes.c_int(m))),
                         dtype=torch.uint8, device=xyz1.device)
        code = lib.sc_chamfer3d_forward_grid(_lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(dist1), _lib.ptr(dist2), _lib.ptr(idx1),
                                             _lib.ptr(idx2), ctypes.c_int(b), ctypes.c_int(n), ctypes.c_int(m), _lib.ptr(ws),
                                             _lib.stream())
        _lib.check(code, "sc_chamfer3d_forward_grid")
        return 1
    if max(n, m) >= 4096:
        # all pairs with the target cloud cut into slices so that the launch is a whole number of full rounds of the chip (nsplit 0: the
        # library chooses per direction; evaluation at b = 1: 13 slices -> one round; b = 32: 2 slices -> 4.9 rounds instead of 2.45)
        ws = torch.empty(b * (n + m), dtype=torch.int64, device=xyz1.device)
        code = lib.sc_chamfer3d_forward_split(_lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(dist1), _lib.ptr(dist2),
                                              _lib.ptr(idx1), _lib.ptr(idx2), ctypes.c_int(b), ctypes.c_int(n),
                                              ctypes.c_int(m), ctypes.c_int(NSPLIT), _lib.ptr(ws), _lib.stream())
        _lib.check(code, "sc_chamfer3d_forward_split")
        return 1
    code = lib.sc_chamfer3d_forward(_lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(dist1), _lib.ptr(dist2),
                                    _lib.ptr(idx1), _lib.ptr(idx2), ctypes.c_int(b), ctypes.c_int(n),
                                    ctypes.c_int(m), _lib.stream())
    _lib.check(code, "sc_chamfer3d_forward")
    return 1


def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    lib = _lib.load()
    b, n, m = _dims(xyz1, xyz2)
    f32, i32 = torch.float32, torch.int32
    _check(xyz1.device, xyz1=(xyz1, f32, (b, n, 3)), xyz2=(xyz2, f32, (b, m, 3)), gradxyz1=(gradxyz1, f32, (b, n, 3)),
           gradxyz2=(gradxyz2, f32, (b, m, 3)), graddist1=(graddist1, f32, (b, n)), graddist2=(graddist2, f32, (b, m)),
           idx1=(idx1, i32, (b, n)), idx2=(idx2, i32, (b, m)))
    with torch.cuda.device(xyz1.device):
        code = lib.sc_chamfer3d_backward(_lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(gradxyz1), _lib.ptr(gradxyz2),
                                         _lib.ptr(graddist1), _lib.ptr(graddist2), _lib.ptr(idx1), _lib.ptr(idx2),
                                         ctypes.c_int(b), ctypes.c_int(n), ctypes.c_int(m), _lib.stream())
    _lib.check(code, "sc_chamfer3d_backward")
    return 1
