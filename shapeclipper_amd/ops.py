"""Thin Python wrappers over the C ABI (allocation + pointer plumbing only; no arithmetic here)."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from .packing import n_tiles

c_int = ctypes.c_int


def sdf_forward(points: torch.Tensor, w_pack: torch.Tensor, cbias: torch.Tensor, n_per_image: int,
                symmetric: bool = True, want_grad: bool = True, want_feat: bool = True,
                stash: bool = False):
    """points [N,3] -> (sdf [N], grad [N,3] | None, feat TBL64 | None[, stash_a, stash_p])."""
    lib = _lib.load()
    n = points.shape[0]
    dev = points.device
    nt = n_tiles(n)
    sdf = torch.empty(n, device=dev, dtype=torch.float32)
    grad = torch.empty(n, 3, device=dev, dtype=torch.float32) if want_grad else None
    feat = torch.empty(nt * 1024, device=dev, dtype=torch.float32) if want_feat else None
    sa = torch.empty(5 * nt * 1024, device=dev, dtype=torch.float32) if stash else None
    sp = torch.empty(4 * nt * 1024, device=dev, dtype=torch.float32) if (stash and want_grad) else None
    code = lib.sc_sdf_forward(_lib.ptr(points), _lib.ptr(w_pack), _lib.ptr(cbias), c_int(n), c_int(n_per_image),
                              c_int(cbias.shape[0]), c_int(1 if symmetric else 0), _lib.ptr(sdf), _lib.ptr(grad),
                              _lib.ptr(feat), _lib.ptr(sa), _lib.ptr(sp), _lib.stream())
    _lib.check(code, "sc_sdf_forward")
    if stash:
        return sdf, grad, feat, sa, sp
    return sdf, grad, feat
