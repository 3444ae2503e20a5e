"""Host-side weight packing for the HIP MLP kernels (pure torch, differentiable).

The kernels (csrc/mlp_tile.hpp) want
  * positional-encoding columns in *slot order* (48 columns: 39 real + 9 zero pads) so that lane
    group g of a point owns frequencies {2^(2g), 2^(2g+1)} and the raw coordinates,
  * the per-image latent folded into per-image biases  c_l = b_l + W_l[:, latent] @ z,
  * the skip-connection scale 1/sqrt(2) (model/implicit.py:155) pre-multiplied into W1, W2.
Doing this with torch ops keeps autograd exact for every original parameter and for the latents:
the kernels return d/d(w_pack) and d/d(cbias), torch propagates to lin*.weight / lin*.bias / z.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

PE_COLS = 48
SDF_PACK_FLOATS = 64 * 48 + 2 * 64 * 112 + 2 * 64 * 64 + 65 * 64 + 65
RGB_PACK_FLOATS = 64 * 112 + 2 * 64 * 64 + 3 * 64 + 4
TILE = 16
# float offsets inside the packed images (must match SdfPack / RgbPack in csrc/mlp_tile.hpp)
SDF_OFF = dict(W0=0, W1=64 * 48, W2=64 * 48 + 64 * 112, W3=64 * 48 + 2 * 64 * 112,
               W4=64 * 48 + 2 * 64 * 112 + 64 * 64, W5=64 * 48 + 2 * 64 * 112 + 2 * 64 * 64,
               B5=64 * 48 + 2 * 64 * 112 + 2 * 64 * 64 + 65 * 64)
RGB_OFF = dict(V0=0, V1=64 * 112, V2=64 * 112 + 64 * 64, V3=64 * 112 + 2 * 64 * 64,
               B3=64 * 112 + 2 * 64 * 64 + 3 * 64)


def pe_slot_col(col: int) -> int:
    """Reference PE column (model/implicit.py:12-34 order) held by packed column `col`; -1 = zero pad.
    Must match pe_slot_col() in csrc/mlp_tile.hpp."""
    g, cj = col & 3, col >> 2
    c, j = cj >> 2, cj & 3
    if g < 3:
        return 3 + 6 * (2 * g + (j >> 1)) + 3 * (j & 1) + c
    return c if j == 0 else -1


_SLOT_IDX = [pe_slot_col(c) if pe_slot_col(c) >= 0 else 39 for c in range(PE_COLS)]


_SLOT_IDX_DEV = {}


def _slots(w_pe: torch.Tensor) -> torch.Tensor:
    """[out, 39] -> [out, 48] in slot order (pads are exact zeros)."""
    ext = torch.cat([w_pe, w_pe.new_zeros(w_pe.shape[0], 1)], dim=1)
    key = str(w_pe.device)
    if key not in _SLOT_IDX_DEV:           # built once per device: a list -> device tensor copy synchronises
        _SLOT_IDX_DEV[key] = torch.as_tensor(_SLOT_IDX, device=w_pe.device)
    return ext[:, _SLOT_IDX_DEV[key]]


def check_arch(opt) -> None:
    a = opt.arch
    ok = (a.impl_sdf.n_hidden_layers == 5 and a.impl_sdf.n_channels == 64 and a.impl_sdf.pos_enc == 6
          and list(a.impl_sdf.skip_connection) == [1, 2] and not a.impl_sdf.weight_norm
          and a.impl_rgb.n_hidden_layers == 3 and a.impl_rgb.n_channels == 64 and a.impl_rgb.pos_enc == 6
          and not a.impl_rgb.weight_norm)
    if not ok:
        raise NotImplementedError(
            "shapeclipper_amd HIP kernels are specialised for the shipped architecture "
            "(impl_sdf: 5x64, pos_enc 6, skip [1,2]; impl_rgb: 3x64, pos_enc 6; no weight_norm)")


def sdf_cbias(W: Dict[str, torch.Tensor], z: torch.Tensor) -> torch.Tensor:
    """Per-image biases c_l = b_l + W_l[:, latent] @ z (skip layers scaled by 1/sqrt2) -> [B, 5, 64]."""
    r = 1.0 / math.sqrt(2.0)
    w0, w1, w2 = W["lin0.weight"], W["lin1.weight"], W["lin2.weight"]
    B = z.shape[0]
    c0 = W["lin0.bias"] + z @ w0[:, 39:].t()
    c1 = W["lin1.bias"] + (z @ w1[:, 103:].t()) * r
    c2 = W["lin2.bias"] + (z @ w2[:, 103:].t()) * r
    c3 = W["lin3.bias"].unsqueeze(0).expand(B, 64)
    c4 = W["lin4.bias"].unsqueeze(0).expand(B, 64)
    return torch.stack([c0, c1, c2, c3, c4], dim=1).contiguous()


def pack_sdf(W: Dict[str, torch.Tensor], z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """SDFNetwork parameters (state-dict names lin{l}.weight/.bias) + latent z [B, Z]
    -> (w_pack [SDF_PACK_FLOATS], cbias [B, 5, 64])."""
    r = 1.0 / math.sqrt(2.0)
    w0, w1, w2 = W["lin0.weight"], W["lin1.weight"], W["lin2.weight"]
    Z = z.shape[1]
    assert w0.shape == (64, 39 + Z) and w1.shape == (64, 64 + 39 + Z)
    cbias = sdf_cbias(W, z)
    pack = torch.cat([
        _slots(w0[:, :39]).reshape(-1),
        torch.cat([w1[:, :64] * r, _slots(w1[:, 64:103]) * r], dim=1).reshape(-1),
        torch.cat([w2[:, :64] * r, _slots(w2[:, 64:103]) * r], dim=1).reshape(-1),
        W["lin3.weight"].reshape(-1), W["lin4.weight"].reshape(-1),
        W["lin5.weight"].reshape(-1), W["lin5.bias"].reshape(-1),
    ]).contiguous()
    assert pack.numel() == SDF_PACK_FLOATS
    return pack, cbias


def rgb_dbias(W: Dict[str, torch.Tensor], z: torch.Tensor) -> torch.Tensor:
    v0 = W["lin0.weight"]
    Z, B = z.shape[1], z.shape[0]
    d0 = W["lin0.bias"] + z @ v0[:, 39:39 + Z].t()
    d1 = W["lin1.bias"].unsqueeze(0).expand(B, 64)
    d2 = W["lin2.bias"].unsqueeze(0).expand(B, 64)
    return torch.stack([d0, d1, d2], dim=1).contiguous()


def pack_rgb(W: Dict[str, torch.Tensor], z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """RGBNetwork parameters + latent z_rgb [B, Z] -> (v_pack [RGB_PACK_FLOATS], dbias [B, 3, 64]).
    lin0 input order is [PE(39), z_rgb(Z), sdf_feature(64)] (model/implicit.py:231)."""
    v0 = W["lin0.weight"]
    Z = z.shape[1]
    assert v0.shape == (64, 39 + Z + 64)
    dbias = rgb_dbias(W, z)
    pack = torch.cat([
        torch.cat([_slots(v0[:, :39]), v0[:, 39 + Z:]], dim=1).reshape(-1),
        W["lin1.weight"].reshape(-1), W["lin2.weight"].reshape(-1),
        W["lin3.weight"].reshape(-1), W["lin3.bias"].reshape(-1), v0.new_zeros(1),
    ]).contiguous()
    assert pack.numel() == RGB_PACK_FLOATS
    return pack, dbias


def n_tiles(n_points: int) -> int:
    return (n_points + TILE - 1) // TILE


def tbl_to_rows(t: torch.Tensor, n_points: int) -> torch.Tensor:
    """TBL64 tensor [ntiles*1024] -> [n_points, 64] (testing / interop helper)."""
    nt = n_tiles(n_points)
    return t.view(nt, 16, 16, 4).permute(0, 2, 1, 3).reshape(nt * 16, 64)[:n_points]


def rows_to_tbl(x: torch.Tensor) -> torch.Tensor:
    """[n_points, 64] -> TBL64 [ntiles*1024] (zero padded)."""
    n = x.shape[0]
    nt = n_tiles(n)
    pad = x.new_zeros(nt * 16, 64)
    pad[:n] = x
    return pad.view(nt, 16, 16, 4).permute(0, 2, 1, 3).contiguous().view(-1)
