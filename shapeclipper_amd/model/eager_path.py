"""Stock PyTorch-ROCm path for implicit-network configurations OUTSIDE the family the HIP chain kernels are compiled for.

The hand-written kernels (csrc/sdf_fwd.hip, sdf_bwdw.hip, rgb_fwd.hip, rgb_bwd.hip) keep a whole network's weights in the 160 KiB of LDS
of a compute unit: 5 x 64 SDF layers + 3 x 64 RGB layers, 6 octaves, 64 samples per ray, and every smaller member by zero padding
(packing.check_arch).  The rest of the reference's configuration space (model/implicit.py:93-113,199-214: any n_hidden_layers, n_channels,
pos_enc, skip_connection; model/renderer.py:13-37: any n_samples_uniform) does not fit that structure -- 128 channels are 320 KiB of
weights.  Like the convolutions of shapes the trunk kernels do not take, those configurations run here on stock
device operators and torch autograd (incl. the create_graph double backward), so a reference user's YAML with another architecture trains
and evaluates instead of raising.  NOT a CPU fallback and not used by the shipped configuration: tensors must be on a ROCm device, and a
one-line warning says which path is taken and why.

Formulated for the device rather than transcribed from the reference: the positional encoding is evaluated once per point and shared by
the skip layers, and the latent never gets repeated per point (reference implicit.py:166, renderer.py:89) -- its columns of a layer's
weight act on the [B, Z] latent and enter as a per-image bias, exactly as the HIP kernels fold them.
"""
from __future__ import annotations

import math
import warnings

import torch
import torch.nn.functional as torch_F

_WARNED = set()


def warn_once(what: str):
    if what not in _WARNED:
        _WARNED.add(what)
        warnings.warn("shapeclipper_amd: %s -- outside the family of the HIP chain kernels (5x64 SDF / 3x64 RGB layers, pos_enc <= 6, skip "
                      "within [1, 2], 64 samples per ray): this configuration runs on stock PyTorch-ROCm operators (model/eager_path.py), "
                      "several times slower than the compiled family" % what, stacklevel=3)


def require_device(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("shapeclipper_amd: the stock-operator path for other architectures is a DEVICE path (no CPU fallback): got a CPU tensor")


def positional_encoding(x: torch.Tensor, octaves: int) -> torch.Tensor:
    """[..., 3] -> [..., 3 + 6 L]: x, then per octave m sin(2^m x) (3 columns), cos(2^m x) (3 columns) (reference implicit.py:12-34)."""
    if octaves <= 0:
        return x
    f = 2.0 ** torch.arange(octaves, device=x.device, dtype=x.dtype)
    xf = x.unsqueeze(-2) * f.view(-1, 1)                                      # [..., L, 3]
    sc = torch.stack([torch.sin(xf), torch.cos(xf)], dim=-2)                   # [..., L, 2, 3]
    return torch.cat([x, sc.flatten(-3)], dim=-1)


def _symmetric(points, on):
    return torch.cat([points[..., :1].abs(), points[..., 1:]], dim=-1) if on else points


def sdf_mlp(net, points: torch.Tensor, latent: torch.Tensor) -> torch.Tensor:
    """SDFNetwork.forward for points [B, n, 3] and ONE latent per image [B, Z] -> [B, n, 1 + C] (reference implicit.py:138-161)."""
    W = net.weight_dict()
    pe_dim = 3 + 6 * net.posenc_res
    e = positional_encoding(_symmetric(points, net.force_symmetry), net.posenc_res)      # [B, n, pe]
    r = 1.0 / math.sqrt(2.0)
    x = None
    n_lin = net.num_layers - 1
    for l in range(n_lin):
        w, b = W["lin%d.weight" % l], W["lin%d.bias" % l]
        if l == 0:                                               # input = [PE | latent]
            y = e @ w[:, :pe_dim].t() + (latent @ w[:, pe_dim:].t() + b).unsqueeze(1)
        elif l in net.skip_in:                                   # input = [h | PE | latent] / sqrt 2
            C = x.shape[-1]
            y = (x @ w[:, :C].t() + e @ w[:, C:C + pe_dim].t()) * r + (latent @ w[:, C + pe_dim:].t() * r + b).unsqueeze(1)
        else:
            y = x @ w.t() + b
        x = torch_F.softplus(y, beta=100) if l < n_lin - 1 else y
    return x


def sdf_conditional_output(net, batch_size: int, points_flat: torch.Tensor, latent: torch.Tensor, compute_grad: bool):
    """get_conditional_output (reference implicit.py:163-189): (sdf [N,1], feature [N,C], d sdf / d point [N,3] | None)."""
    require_device(points_flat)
    if compute_grad:
        latent = latent.detach()
    with torch.enable_grad():
        pts = points_flat if points_flat.requires_grad else points_flat.detach().requires_grad_(True)
        out = sdf_mlp(net, pts.view(batch_size, -1, 3), latent).reshape(-1, 1 + net.n_channel)
        sdf, feat = out[:, :1], out[:, 1:]
        grad = torch.autograd.grad(sdf, pts, torch.ones_like(sdf), create_graph=True, retain_graph=True)[0] if compute_grad else None
    return sdf, feat, grad


def rgb_mlp(net, points: torch.Tensor, latent: torch.Tensor, feature: torch.Tensor) -> torch.Tensor:
    """RGBNetwork.forward for points [B, n, 3], latent [B, Z], SDF feature [B, n, C_sdf] -> colours [B, n, 3] (implicit.py:220-239)."""
    W = net.weight_dict()
    pe_dim = 3 + 6 * net.posenc_res
    Z = latent.shape[1]
    e = positional_encoding(_symmetric(points, net.force_symmetry), net.posenc_res)
    n_lin = net.num_layers - 1
    x = None
    for l in range(n_lin):
        w, b = W["lin%d.weight" % l], W["lin%d.bias" % l]
        if l == 0:                                               # input = [PE | latent | sdf feature]
            y = e @ w[:, :pe_dim].t() + feature @ w[:, pe_dim + Z:].t() + (latent @ w[:, pe_dim:pe_dim + Z].t() + b).unsqueeze(1)
        else:
            y = x @ w.t() + b
        x = torch.relu(y) if l < n_lin - 1 else torch.sigmoid(y)
    return x


def render(renderer, opt, cam_loc, ray_dirs, depth_fac, z_vals, z_eik, eik_uniform, B, R, latent_sdf, latent_rgb, training):
    """Renderer.forward after the rays and depths are known (reference renderer.py:79-170) -> the six outputs (flat per-sample tensors
    for `visualize` as a seventh item).  cam_loc / ray_dirs [B R, 3], depth_fac [B R], z_vals [B R, S], eik_uniform [B, R, 3] | None."""
    require_device(ray_dirs)
    S = z_vals.shape[1]
    sdf_net, rgb_net = renderer.sdf_network, renderer.rgb_network
    points_flat = (cam_loc.unsqueeze(1) + z_vals.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
    with torch.enable_grad():                       # the normals need d density / d point even under no_grad (renderer.py:94-107)
        pts = points_flat if points_flat.requires_grad else points_flat.detach().requires_grad_(True)
        out = sdf_mlp(sdf_net, pts.view(B, R * S, 3), latent_sdf)
        sdf, feat = out[..., :1], out[..., 1:]
        density = renderer.density(sdf)
        normal_flat = -torch.autograd.grad(density, pts, torch.ones_like(density), create_graph=True, retain_graph=True)[0]
    rgb_flat = rgb_mlp(rgb_net, pts.view(B, R * S, 3), latent_rgb, feat).reshape(-1, S, 3)
    weights, alphas = renderer.volume_rendering(z_vals, sdf.reshape(-1, 1))
    depth = (weights * z_vals * depth_fac.view(-1, 1)).sum(1)
    nrm = torch_F.normalize(normal_flat, dim=-1, p=2).reshape(-1, S, 3)
    normal = torch_F.normalize(((weights.unsqueeze(-1) ** opt.reg.normal_pow) * nrm).sum(1), dim=-1, p=2)
    acc = weights.sum(-1)
    rgb = (weights.unsqueeze(-1) * rgb_flat).sum(1) + (1.0 - acc.unsqueeze(1)) * renderer.bg_color
    grad_eikonal = None
    if training:
        near = (cam_loc + z_eik * ray_dirs).reshape(B, R, 3)
        eik_points = torch.cat([eik_uniform, near], 1).reshape(-1, 3)
        grad_eikonal = sdf_conditional_output(sdf_net, B, eik_points, latent_sdf, True)[2].norm(2, dim=1)
    return (rgb.view(B, R, 3), acc.view(B, R, 1), (acc > 0.5).float().view(B, R, 1), depth.view(B, R, 1), normal.view(B, R, 3), grad_eikonal,
            (points_flat, alphas, rgb_flat))
