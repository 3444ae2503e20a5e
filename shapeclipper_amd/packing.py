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


def arch_supported(opt) -> bool:
    """True when the HIP chain kernels take opt.arch (see check_arch); otherwise the networks run on model/eager_path.py."""
    a = opt.arch
    return (a.impl_sdf.n_hidden_layers == 5 and 1 <= a.impl_sdf.n_channels <= 64 and 0 <= a.impl_sdf.pos_enc <= 6
            and set(a.impl_sdf.skip_connection) <= {1, 2}
            and a.impl_rgb.n_hidden_layers == 3 and 1 <= a.impl_rgb.n_channels <= 64 and 0 <= a.impl_rgb.pos_enc <= 6)


def arch_summary(opt) -> str:
    a = opt.arch
    return "impl_sdf %d x %d, pos_enc %d, skip %s; impl_rgb %d x %d, pos_enc %d" % (
        a.impl_sdf.n_hidden_layers, a.impl_sdf.n_channels, a.impl_sdf.pos_enc, list(a.impl_sdf.skip_connection),
        a.impl_rgb.n_hidden_layers, a.impl_rgb.n_channels, a.impl_rgb.pos_enc)


def check_arch(opt) -> None:
    """What the HIP chain kernels take.  They are compiled for 5 hidden SDF layers / 3 hidden RGB layers of 64 channels with the positional
    encoding in 48 slots (6 octaves) and optional skip inputs at SDF layers 1 and 2; every SMALLER member of the reference's config family
    (model/implicit.py:89-113,197-214) is embedded exactly by the weight packing below -- unused rows / columns / octaves / skip inputs of
    the kernel's image are exact zeros, so the extra channels carry softplus(0) or relu(0) into zero weights: n_channels <= 64 (both
    networks, independently), pos_enc 0..6 (both), skip_connection any subset of [1, 2], any proj_latent_dim.  Depth is not embeddable
    (a softplus layer cannot be made an identity) and wider layers / more octaves need other kernels: those raise."""
    a = opt.arch
    ok = (a.impl_sdf.n_hidden_layers == 5 and 1 <= a.impl_sdf.n_channels <= 64 and 0 <= a.impl_sdf.pos_enc <= 6
          and set(a.impl_sdf.skip_connection) <= {1, 2}
          and a.impl_rgb.n_hidden_layers == 3 and 1 <= a.impl_rgb.n_channels <= 64 and 0 <= a.impl_rgb.pos_enc <= 6)
    if not ok:
        raise NotImplementedError(
            "shapeclipper_amd HIP kernels take impl_sdf: 5 hidden layers, n_channels <= 64, pos_enc <= 6, skip_connection within [1, 2]; "
            "impl_rgb: 3 hidden layers, n_channels <= 64, pos_enc <= 6 (weight_norm on or off, any proj_latent_dim); got sdf %dx%d pos_enc %d skip %s, "
            "rgb %dx%d pos_enc %d" % (a.impl_sdf.n_hidden_layers, a.impl_sdf.n_channels, a.impl_sdf.pos_enc, list(a.impl_sdf.skip_connection),
                                      a.impl_rgb.n_hidden_layers, a.impl_rgb.n_channels, a.impl_rgb.pos_enc))


# ---- gather plans ---------------------------------------------------------------------------------------------------------
# Packing used to be ~70 small torch operators per forward pass (slices, cats, the slot permutation, three latent matmuls) and ~300
# in backward (every Select / Slice backward is a fill + a copy): a tenth of the launches of a training step.  Now ONE index_select
# over the concatenated raw parameters builds, in a single vector,   [ packed weight image | latent columns of the conditioned layers
# | bias rows ],   with the 1/sqrt(2) skip scale as one element-wise multiply: 3 launches forward and 3 backward (every source element is
# gathered at most once, so the index_add_ of the backward pass has no colliding addresses besides the shared zero pad and is
# deterministic); the per-image biases are one matmul on the gathered latent block.  The packed WEIGHT image is bit-identical to the
# operator-by-operator construction (each element is still `parameter` or `parameter * r`); the per-image latent bias is the same sum
# to within GEMM rounding only -- one [B,Z] x [Z,192] product instead of three [B,Z] x [Z,64], for which rocBLAS may pick another
# kernel / summation order (last ulps) -- so tests compare cbias / dbias with a tolerance, never for equality.
_PLANS = {}


def _plan(kind: str, Z: int, device, arch=None):
    """arch: sdf -> (C, pe, skips) = channels, positional-encoding columns 3 + 6 L, tuple of skip layers; rgb -> (C, pe, C_sdf).
    None = the shipped architecture."""
    if arch is None:
        arch = (64, 39, (1, 2)) if kind == "sdf" else (64, 39, 64)
    key = (kind, Z, str(device), arch)
    if key in _PLANS:
        return _PLANS[key]
    r = 1.0 / math.sqrt(2.0)
    idx, scl = [], []
    C, pe = arch[0], arch[1]
    if kind == "sdf":
        skips = tuple(arch[2])
        d0 = pe + Z
        win = lambda l: C + (d0 if l in skips else 0)
        shapes = [(C, d0), (C,), (C, win(1)), (C,), (C, win(2)), (C,), (C, C), (C,), (C, C), (C,), (1 + C, C), (1 + C,)]
    else:
        Cs = arch[2]
        shapes = [(C, pe + Z + Cs), (C,), (C, C), (C,), (C, C), (C,), (3, C), (3,)]
    offs, o = [], 0
    for sh in shapes:
        offs.append(o)
        o += int(torch.Size(sh).numel())
    zero = o                                           # index of the appended exact zero

    def mat(k, row, col):                              # flat index of element (row, col) of parameter k
        return offs[k] + row * shapes[k][1] + col

    def put(i, sc=1.0):
        idx.append(i); scl.append(sc)

    def slot(c):                                       # reference PE column of packed slot c, or -1 (pad / octave the network does not have)
        return _SLOT_IDX[c] if _SLOT_IDX[c] < pe else -1
    if kind == "sdf":
        for row in range(64):                          # W0: PE slots
            for c in range(PE_COLS):
                put(mat(0, row, slot(c)) if (row < C and slot(c) >= 0) else zero)
        for l, k in ((1, 2), (2, 4)):                  # W1, W2: [hidden 64 | PE slots 48]; skip layers take [h, input] / sqrt 2
            sc = r if l in skips else 1.0
            for row in range(64):
                for c in range(64):
                    put(mat(k, row, c) if (row < C and c < C) else zero, sc)
                for c in range(PE_COLS):
                    put(mat(k, row, C + slot(c)) if (l in skips and row < C and slot(c) >= 0) else zero, sc)
        for k, rows in ((6, 64), (8, 64), (10, 65)):   # W3, W4, W5 (row 0 = sdf, rows 1.. = feature)
            nr = C if rows == 64 else 1 + C
            for row in range(rows):
                for c in range(64):
                    put(mat(k, row, c) if (row < nr and c < C) else zero)
        for row in range(65):                          # b5
            put(offs[11] + row if row < 1 + C else zero)
        n_pack = len(idx)
        assert n_pack == SDF_PACK_FLOATS
        for l, k, c0 in ((0, 0, pe), (1, 2, C + pe), (2, 4, C + pe)):    # latent columns [3 x 64 rows][Z]
            has = l == 0 or l in skips
            for row in range(64):
                for c in range(Z):
                    put(mat(k, row, c0 + c) if (has and row < C) else zero)
        for k in (1, 3, 5, 7, 9):                      # bias rows b0..b4
            for row in range(64):
                put(offs[k] + row if row < C else zero)
        n_lat, n_bias = 3 * 64 * Z, 5 * 64
        post = torch.tensor([1.0, r if 1 in skips else 1.0, r if 2 in skips else 1.0], device=device).view(1, 3, 1)    # (z @ W_lat^T) * r for the skip layers
    else:
        for row in range(64):                          # V0: [PE slots 48 | sdf feature 64]
            for c in range(PE_COLS):
                put(mat(0, row, slot(c)) if (row < C and slot(c) >= 0) else zero)
            for c in range(64):
                put(mat(0, row, pe + Z + c) if (row < C and c < Cs) else zero)
        for k, rows in ((2, 64), (4, 64), (6, 3)):
            for row in range(rows):
                for c in range(64):
                    put(mat(k, row, c) if ((rows == 3 or row < C) and c < C) else zero)
        for row in range(3):
            put(offs[7] + row)
        put(zero)
        n_pack = len(idx)
        assert n_pack == RGB_PACK_FLOATS
        for row in range(64):
            for c in range(Z):
                put(mat(0, row, pe + c) if row < C else zero)
        for k in (1, 3, 5):
            for row in range(64):
                put(offs[k] + row if row < C else zero)
        n_lat, n_bias = 64 * Z, 3 * 64
        post = None
    scale = torch.tensor(scl, dtype=torch.float32)
    plan = dict(idx=torch.tensor(idx, dtype=torch.int64).to(device), scale=None if bool((scale == 1).all()) else scale.to(device),
                n_pack=n_pack, n_lat=n_lat, n_bias=n_bias, post=post, shapes=shapes, total=zero)
    _PLANS[key] = plan
    return plan


def arch_of(kind: str, W: Dict[str, torch.Tensor], Z: int, n_sdf: int = 64):
    """The architecture tuple _plan() wants, read off the parameter shapes (reference layer construction: model/implicit.py:93-113,199-214)."""
    C, d_in = W["lin0.weight"].shape
    if kind == "sdf":
        pe = d_in - Z
        skips = tuple(l for l in (1, 2) if W["lin%d.weight" % l].shape[1] == C + d_in)
        return (C, pe, skips)
    return (C, d_in - Z - n_sdf, n_sdf)


def _names(kind):
    n = 6 if kind == "sdf" else 4
    return [k for l in range(n) for k in ("lin%d.weight" % l, "lin%d.bias" % l)]


class _PackParams(torch.autograd.Function):
    """(packed weight image, latent block [L*64, Z], bias rows [NL, 64]) of a network from its raw parameters: ONE gather forward, ONE
    scatter-add backward.  Three separate outputs instead of slices of one vector: a slice node's backward is a full-size zero fill + a copy,
    and the pieces have 3-6 consumers per step (round 5: 7 fills + 6 adds of the 42,433-element vector per step went away)."""

    @staticmethod
    def forward(ctx, kind, Z, arch, *params):
        plan = _plan(kind, Z, params[0].device, arch)
        src = torch.cat([p.reshape(-1) for p in params] + [params[0].new_zeros(1)])
        out = torch.index_select(src, 0, plan["idx"])
        if plan["scale"] is not None:
            out = out * plan["scale"]
        ctx.plan, ctx.shapes = plan, [tuple(p.shape) for p in params]
        n_pack, n_lat = plan["n_pack"], plan["n_lat"]
        return out[:n_pack], out[n_pack:n_pack + n_lat].view(n_lat // Z, Z), out[n_pack + n_lat:].view(-1, 64)

    @staticmethod
    def backward(ctx, g_w, g_lat, g_bias):
        plan = ctx.plan
        ref = next(t for t in (g_w, g_lat, g_bias) if t is not None)
        z = lambda n: ref.new_zeros(n)
        g = torch.cat([g_w.reshape(-1) if g_w is not None else z(plan["n_pack"]), g_lat.reshape(-1) if g_lat is not None else z(plan["n_lat"]),
                       g_bias.reshape(-1) if g_bias is not None else z(plan["n_bias"])])
        if plan["scale"] is not None:
            g = g * plan["scale"]
        flat = ref.new_zeros(plan["total"] + 1).index_add_(0, plan["idx"], g)     # (every source element is gathered at most once besides the zero pad)
        outs, o = [], 0
        for sh in ctx.shapes:
            n = int(torch.Size(sh).numel())
            outs.append(flat[o:o + n].view(sh))
            o += n
        return (None, None, None) + tuple(outs)


class _LatentBias(torch.autograd.Function):
    """c = bias + post * (z @ lat^T) per image in one HIP launch each way (csrc/latent_bias.hip); batch-size invariant."""

    @staticmethod
    def forward(ctx, z, lat, bias, post):
        from . import ops
        z, lat, bias = ops._aligned(z), ops._aligned(lat), ops._aligned(bias)
        out = ops.latent_bias_forward(z, lat, bias, post)
        ctx.save_for_backward(z, lat, post)
        ctx.nl = bias.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        from . import ops
        z, lat, post = ctx.saved_tensors
        g_z, g_lat, g_bias = ops.latent_bias_backward(ops._aligned(g), z, lat, post, ctx.nl, ctx.needs_input_grad[0])
        return g_z, g_lat, g_bias, None


def gather_params(kind: str, W: Dict[str, torch.Tensor], Z: int, arch=None):
    """(packed image, latent block [L*64, Z], bias rows [NL, 64]) of a network in one gather (differentiable w.r.t. every parameter)."""
    names = _names(kind)
    plan = _plan(kind, Z, W[names[0]].device, arch)
    for n, sh in zip(names, plan["shapes"]):
        assert tuple(W[n].shape) == tuple(sh), (n, tuple(W[n].shape), sh)
    return _PackParams.apply(kind, Z, arch, *[W[n] for n in names])


def _bias_from(kind: str, parts, z: torch.Tensor, arch=None) -> torch.Tensor:
    lat, bias = parts
    B, Z = z.shape
    plan = _plan(kind, Z, lat.device, arch)
    L, NL = lat.shape[0] // 64, bias.shape[0]
    post = plan["post"]
    if z.is_cuda:                                       # one launch each way; no CPU fallback on the device path (raises if the library is missing)
        return _LatentBias.apply(z, lat, bias, post.reshape(-1) if post is not None else None)
    zw = (z @ lat.t()).view(B, L, 64)                   # host tensors (config[0] plumbing, CPU tests of the packing): stock operators
    if post is not None:
        zw = zw * post
    return torch.nn.functional.pad(zw, (0, 0, 0, NL - L)) + bias.unsqueeze(0)


def sdf_cbias(W: Dict[str, torch.Tensor], z: torch.Tensor, gathered=None, arch=None) -> torch.Tensor:
    """Per-image biases c_l = b_l + W_l[:, latent] @ z (skip layers scaled by 1/sqrt2) -> [B, 5, 64].  gathered: the (latent block, bias rows)
    pair pack_sdf(..., return_gathered=True) handed out (shared by the renders of a step)."""
    if arch is None and W is not None:
        arch = arch_of("sdf", W, z.shape[1])
    parts = gathered if gathered is not None else gather_params("sdf", W, z.shape[1], arch)[1:]
    return _bias_from("sdf", parts, z, arch)


def pack_sdf(W: Dict[str, torch.Tensor], z: torch.Tensor, return_gathered: bool = False):
    """SDFNetwork parameters (state-dict names lin{l}.weight/.bias) + latent z [B, Z]
    -> (w_pack [SDF_PACK_FLOATS], cbias [B, 5, 64]).  The architecture (channels, octaves, skip inputs) is read off the shapes."""
    arch = arch_of("sdf", W, z.shape[1])
    w_pack, lat, bias = gather_params("sdf", W, z.shape[1], arch)
    out = (w_pack, _bias_from("sdf", (lat, bias), z, arch))
    return out + ((lat, bias),) if return_gathered else out


def rgb_dbias(W: Dict[str, torch.Tensor], z: torch.Tensor, gathered=None, arch=None, n_sdf: int = 64) -> torch.Tensor:
    if arch is None and W is not None:
        arch = arch_of("rgb", W, z.shape[1], n_sdf)
    parts = gathered if gathered is not None else gather_params("rgb", W, z.shape[1], arch)[1:]
    return _bias_from("rgb", parts, z, arch)


def pack_rgb(W: Dict[str, torch.Tensor], z: torch.Tensor, return_gathered: bool = False, n_sdf: int = 64):
    """RGBNetwork parameters + latent z_rgb [B, Z] -> (v_pack [RGB_PACK_FLOATS], dbias [B, 3, 64]).
    lin0 input order is [PE(39), z_rgb(Z), sdf_feature(64)] (model/implicit.py:231)."""
    arch = arch_of("rgb", W, z.shape[1], n_sdf)
    v_pack, lat, bias = gather_params("rgb", W, z.shape[1], arch)
    out = (v_pack, _bias_from("rgb", (lat, bias), z, arch))
    return out + ((lat, bias),) if return_gathered else out


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
