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


def rgb_composite_forward(points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta_param,
                          rays_per_image: int, symmetric: bool, beta_min: float, bgcolor: float,
                          normal_pow: float, keep_samples: bool = False):
    """Per-ray outputs of the renderer from the per-point SDF results.

    points [n_rays*64,3], z_vals [n_rays,64], depth_fac [n_rays], sdf [P], grad [P,3], feat TBL64.
    Returns dict(rgb [n_rays,3], mask, mask_hard, depth [n_rays], normal [n_rays,3]
    [, weights, alpha [n_rays,64], rgb_flat [P,3]])."""
    lib = _lib.load()
    n_rays = z_vals.shape[0]
    assert z_vals.shape[1] == 64, "the compositing kernel maps one 64-lane wavefront to the 64 samples of a ray"
    dev = points.device
    f32 = dict(device=dev, dtype=torch.float32)
    out = dict(rgb=torch.empty(n_rays, 3, **f32), mask=torch.empty(n_rays, **f32),
               mask_hard=torch.empty(n_rays, **f32), depth=torch.empty(n_rays, **f32),
               normal=torch.empty(n_rays, 3, **f32))
    if keep_samples:
        out.update(weights=torch.empty(n_rays, 64, **f32), alpha=torch.empty(n_rays, 64, **f32),
                   rgb_flat=torch.empty(n_rays * 64, 3, **f32))
    code = lib.sc_rgb_composite_forward(
        _lib.ptr(points), _lib.ptr(z_vals), _lib.ptr(depth_fac), _lib.ptr(sdf), _lib.ptr(grad), _lib.ptr(feat),
        _lib.ptr(v_pack), _lib.ptr(dbias), _lib.ptr(beta_param), c_int(n_rays), c_int(rays_per_image),
        c_int(dbias.shape[0]), c_int(1 if symmetric else 0), ctypes.c_float(beta_min), ctypes.c_float(bgcolor),
        ctypes.c_float(normal_pow), _lib.ptr(out["rgb"]), _lib.ptr(out["mask"]), _lib.ptr(out["mask_hard"]),
        _lib.ptr(out["depth"]), _lib.ptr(out["normal"]), _lib.ptr(out.get("weights")), _lib.ptr(out.get("alpha")),
        _lib.ptr(out.get("rgb_flat")), _lib.stream())
    _lib.check(code, "sc_rgb_composite_forward")
    return out
