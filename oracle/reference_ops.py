"""CPU oracle for the ShapeClipper hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Plain, functional PyTorch restatement of the reference algorithm.  Weights are
passed as a flat ``dict[str, Tensor]`` using the reference's state-dict keys
(``lin0.weight`` ... torch ``Linear`` layout ``[out, in]``), configuration as a
small ``Cfg`` object.  Every function cites the reference file:line it follows
(paths relative to /root/reference).  All tensors fp32 unless stated.

Autograd: everything here is differentiable with ``create_graph=True`` exactly
where the reference is, so ``torch.autograd.grad`` of a scalar functional of
``render()`` gives the oracle gradients (including the second-order terms
through d(density)/dx and d(sdf)/dx).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration (the subset of options/pix3d/config.yaml the hot path reads)
# --------------------------------------------------------------------------------------
@dataclass
class Cfg:
    H: int = 224
    W: int = 224
    cam_dist: float = 5.0            # camera.dist
    cam_focal: float = 4.0           # camera.focal
    n_samples: int = 64              # render.n_samples_uniform
    bgcolor: float = 1.0             # data.bgcolor
    normal_pow: float = 1.0          # reg.normal_pow
    eik_range: Tuple[float, float] = (-1.0, 1.0)   # arch.impl_sdf.eikonal_sample_range
    force_symmetry: bool = True      # arch.force_symmetry
    latent_sdf: int = 64             # arch.impl_sdf.proj_latent_dim
    latent_rgb: int = 64             # arch.impl_rgb.proj_latent_dim
    hidden_sdf: int = 64             # arch.impl_sdf.n_channels
    hidden_rgb: int = 64             # arch.impl_rgb.n_channels
    n_hidden_sdf: int = 5            # arch.impl_sdf.n_hidden_layers
    n_hidden_rgb: int = 3            # arch.impl_rgb.n_hidden_layers
    posenc_sdf: int = 6              # arch.impl_sdf.pos_enc
    posenc_rgb: int = 6              # arch.impl_rgb.pos_enc
    skip_in: Sequence[int] = (1, 2)  # arch.impl_sdf.skip_connection
    beta_init: float = 0.1           # arch.impl_sdf.beta_init
    beta_min: float = 1e-4           # LaplaceDensity(beta_min)
    init_sphere_radius: float = 0.5
    normal_l1: float = 5.0           # reg.normal_l1
    mask_mse: float = 0.0            # reg.mask_mse
    emd_p: int = 2                   # reg.emd_p


# --------------------------------------------------------------------------------------
# positional encoding  (model/implicit.py:7-52)
# --------------------------------------------------------------------------------------
def posenc(x: Tensor, n_freq: int) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]  -> [..., 3 + 6 L].

    Embedder.create_embedding_fn, implicit.py:12-34: freq_bands = 2**linspace(0, L-1, L);
    for each freq: sin block (3) then cos block (3); include_input first.
    """
    if n_freq <= 0:
        return x
    freqs = 2.0 ** torch.linspace(0.0, n_freq - 1, n_freq)
    out = [x]
    for f in freqs:
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, dim=-1)


def symmetrize(points: Tensor, force_symmetry: bool) -> Tensor:
    """implicit.py:139-145 / 221-227: x0 <- |x0| on a clone."""
    if not force_symmetry:
        return points
    p = points.clone()
    p[..., 0] = torch.abs(p[..., 0].clone())
    return p


# --------------------------------------------------------------------------------------
# SDF / RGB networks   (model/implicit.py:85-239)
# --------------------------------------------------------------------------------------
def sdf_layer_dims(cfg: Cfg):
    """implicit.py:93-112 -- (in_dim, out_dim) per Linear."""
    d0 = 3 + cfg.latent_sdf + (6 * cfg.posenc_sdf if cfg.posenc_sdf > 0 else 0)
    dims = [d0] + [cfg.hidden_sdf] * cfg.n_hidden_sdf + [1 + cfg.hidden_sdf]
    out = []
    for l in range(len(dims) - 1):
        in_dim = dims[l] + dims[0] if l in cfg.skip_in else dims[l]
        out.append((in_dim, dims[l + 1]))
    return out


def rgb_layer_dims(cfg: Cfg):
    """implicit.py:199-212."""
    d0 = 3 + cfg.latent_rgb + cfg.hidden_sdf + (6 * cfg.posenc_rgb if cfg.posenc_rgb > 0 else 0)
    dims = [d0] + [cfg.hidden_rgb] * cfg.n_hidden_rgb + [3]
    return [(dims[l], dims[l + 1]) for l in range(len(dims) - 1)]


def init_sdf_weights(cfg: Cfg, generator_seed: Optional[int] = None) -> Dict[str, Tensor]:
    """Geometric init, same RNG consumption order as SDFNetwork.__init__ (implicit.py:103-133).

    nn.Linear's own kaiming init draws first (weight then bias), then the geometric
    init overwrites -- reproduced so that ``torch.manual_seed(s)`` gives identical
    tensors to constructing the reference module.
    """
    if generator_seed is not None:
        torch.manual_seed(generator_seed)
    dims = sdf_layer_dims(cfg)
    d0 = dims[0][0]
    n_lin = len(dims)
    W = {}
    for l, (in_dim, out_dim) in enumerate(dims):
        lin = torch.nn.Linear(in_dim, out_dim)
        with torch.no_grad():
            if l == n_lin - 1:
                torch.nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(in_dim), std=0.0001)
                torch.nn.init.constant_(lin.bias, -cfg.init_sphere_radius)
            elif cfg.posenc_sdf > 0 and l == 0:
                torch.nn.init.constant_(lin.bias, 0.0)
                torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
            elif cfg.posenc_sdf > 0 and l in cfg.skip_in:
                torch.nn.init.constant_(lin.bias, 0.0)
                torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                torch.nn.init.constant_(lin.weight[:, -(d0 - 3):], 0.0)
            else:
                torch.nn.init.constant_(lin.bias, 0.0)
                torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
        W[f"lin{l}.weight"] = lin.weight.detach().clone()
        W[f"lin{l}.bias"] = lin.bias.detach().clone()
    return W


def init_rgb_weights(cfg: Cfg, generator_seed: Optional[int] = None) -> Dict[str, Tensor]:
    """Default nn.Linear init in construction order (implicit.py:209-215)."""
    if generator_seed is not None:
        torch.manual_seed(generator_seed)
    W = {}
    for l, (in_dim, out_dim) in enumerate(rgb_layer_dims(cfg)):
        lin = torch.nn.Linear(in_dim, out_dim)
        W[f"lin{l}.weight"] = lin.weight.detach().clone()
        W[f"lin{l}.bias"] = lin.bias.detach().clone()
    return W


def sdf_mlp(cfg: Cfg, W: Dict[str, Tensor], points_raw: Tensor, latent_rep: Tensor) -> Tensor:
    """SDFNetwork.forward, implicit.py:138-161.  points [N,3], latent_rep [N,Z] -> [N, 1+C]."""
    pts = symmetrize(points_raw, cfg.force_symmetry)
    pts = posenc(pts, cfg.posenc_sdf)
    inputs = torch.cat([pts, latent_rep], dim=-1)
    x = inputs
    n_lin = cfg.n_hidden_sdf + 1
    for l in range(n_lin):
        if l in cfg.skip_in:
            x = torch.cat([x, inputs], 1) / np.sqrt(2)
        x = F.linear(x, W[f"lin{l}.weight"], W[f"lin{l}.bias"])
        if l < n_lin - 1:
            x = F.softplus(x, beta=100)
    return x


def sdf_conditional(cfg: Cfg, W: Dict[str, Tensor], batch_size: int, points_flat: Tensor,
                    proj_latent: Tensor, compute_grad: bool = True):
    """SDFNetwork.get_conditional_output, implicit.py:163-189.

    Returns (sdf [N,1], feat [N,C], grad [N,3] | None).  With compute_grad the latent is
    detached (implicit.py:168-169) and d(sdf)/d(points) is taken with create_graph=True.
    """
    N = points_flat.shape[0] // batch_size
    lat = proj_latent.unsqueeze(1).repeat(1, N, 1).view(batch_size * N, -1)
    assert lat.shape[1] == cfg.latent_sdf
    if compute_grad:
        lat = lat.detach()
    points_flat.requires_grad_(True)
    out = sdf_mlp(cfg, W, points_flat, lat)
    sdf, feat = out[:, :1], out[:, 1:]
    grad = None
    if compute_grad:
        grad = torch.autograd.grad(sdf, points_flat, torch.ones_like(sdf),
                                   create_graph=True, retain_graph=True, only_inputs=True)[0]
    return sdf, feat, grad


def rgb_mlp(cfg: Cfg, W: Dict[str, Tensor], points_raw: Tensor, latent_rep: Tensor,
            sdf_feature: Tensor) -> Tensor:
    """RGBNetwork.forward, implicit.py:220-239: [PE, z_rgb, feat] -> ReLU MLP -> sigmoid."""
    pts = symmetrize(points_raw, cfg.force_symmetry)
    pts = posenc(pts, cfg.posenc_rgb)
    x = torch.cat([pts, latent_rep, sdf_feature], dim=-1)
    n_lin = cfg.n_hidden_rgb + 1
    for l in range(n_lin):
        x = F.linear(x, W[f"lin{l}.weight"], W[f"lin{l}.bias"])
        if l < n_lin - 1:
            x = torch.relu(x)
    return torch.sigmoid(x)


def laplace_density(sdf: Tensor, beta_param: Tensor, beta_min: float = 1e-4) -> Tensor:
    """LaplaceDensity.density_func, implicit.py:65-83 (masked scatter form kept)."""
    beta = beta_param.abs() + beta_min
    alpha = 1 / beta
    out = torch.zeros_like(sdf)
    pos = sdf >= 0
    out[pos] = 0.5 * torch.exp(-sdf[pos] / beta)
    out[~pos] = 1 - 0.5 * torch.exp(sdf[~pos] / beta)
    return alpha * out


# --------------------------------------------------------------------------------------
# camera   (utils/camera.py)
# --------------------------------------------------------------------------------------
def make_pose(R: Optional[Tensor] = None, t: Optional[Tensor] = None) -> Tensor:
    """Pose.__call__, camera.py:7-23 -> [...,3,4]."""
    assert R is not None or t is not None
    if R is None:
        R = torch.eye(3).repeat(*t.shape[:-1], 1, 1)
    elif t is None:
        t = torch.zeros(R.shape[:-1])
    return torch.cat([R.float(), t.float()[..., None]], dim=-1)


def pose_invert(pose: Tensor) -> Tensor:
    """Pose.invert, camera.py:25-30 (R^T, -R^T t)."""
    R, t = pose[..., :3], pose[..., 3:]
    R_inv = R.transpose(-1, -2)
    t_inv = (-R_inv @ t)[..., 0]
    return make_pose(R_inv, t_inv)


def pose_compose_pair(a: Tensor, b: Tensor) -> Tensor:
    """Pose.compose_pair, camera.py:39-46: x -> b(a(x))."""
    R_a, t_a = a[..., :3], a[..., 3:]
    R_b, t_b = b[..., :3], b[..., 3:]
    return make_pose(R_b @ R_a, (R_b @ t_a + t_b)[..., 0])


def to_hom(X: Tensor) -> Tensor:
    return torch.cat([X, torch.ones_like(X[..., :1])], dim=-1)


def cam2world(X: Tensor, pose: Tensor) -> Tensor:
    """camera.py:90-96."""
    return to_hom(X) @ pose_invert(pose).transpose(-1, -2)


def get_intr(cfg: Cfg, scale_focal: Tensor) -> Tensor:
    """camera.py:198-211: K = [[f W,0,W/2],[0,f H,H/2],[0,0,1]], f = focal * scale_focal."""
    z, o = torch.zeros_like(scale_focal), torch.ones_like(scale_focal)
    f = cfg.cam_focal * scale_focal
    return torch.stack([f * cfg.W, z, o * cfg.W / 2, z, f * cfg.H, o * cfg.H / 2, z, z, o],
                       dim=-1).view(-1, 3, 3).contiguous()


def get_center_and_ray(cfg: Cfg, pose: Tensor, intr: Tensor):
    """camera.py:157-196, perspective model: pixel centres (+0.5) -> K^-1 -> cam2world.

    Returns center [B,1,3], ray [B,HW,3] (ray = grid - center, un-normalised).
    """
    B = pose.shape[0]
    y = torch.arange(cfg.H, dtype=torch.float32).add_(0.5)
    x = torch.arange(cfg.W, dtype=torch.float32).add_(0.5)
    Y, X = torch.meshgrid(y, x, indexing="ij")
    xy = torch.stack([X, Y], dim=-1).view(-1, 2).repeat(B, 1, 1)
    grid = to_hom(xy) @ intr.inverse().transpose(-1, -2)
    center = torch.zeros(B, 1, 3)
    grid_w = cam2world(grid, pose)
    center_w = cam2world(center, pose)
    return center_w, grid_w - center_w


def transform_normal(normals: Tensor, pose: Tensor) -> Tensor:
    """camera.py:98-103: rotate normals with R^T (translation zeroed)."""
    rot = pose[:, :, :3]
    trans = torch.zeros(1, 3, 1).expand(rot.shape[0], 3, 1)
    return cam2world(normals, torch.cat([rot, trans], dim=-1))


def azim_to_R(trig: Tensor) -> Tensor:
    """camera.py:105-122, representation='trig' ([cos, sin])."""
    c, s = trig[:, 0], trig[:, 1]
    R = torch.eye(3)[None].repeat(len(trig), 1, 1)
    z = torch.zeros(len(trig))
    R[:, 0, :] = torch.stack([c, z, s], dim=-1)
    R[:, 2, :] = torch.stack([-s, z, c], dim=-1)
    return R


def elev_to_R(trig: Tensor) -> Tensor:
    """camera.py:124-139."""
    c, s = trig[:, 0], trig[:, 1]
    R = torch.eye(3)[None].repeat(len(trig), 1, 1)
    R[:, 1, 1:] = torch.stack([c, -s], dim=-1)
    R[:, 2, 1:] = torch.stack([s, c], dim=-1)
    return R


def roll_to_R(trig: Tensor) -> Tensor:
    """camera.py:141-155."""
    c, s = trig[:, 0], trig[:, 1]
    R = torch.eye(3)[None].repeat(len(trig), 1, 1)
    R[:, 0, :2] = torch.stack([c, s], dim=-1)
    R[:, 1, :2] = torch.stack([-s, c], dim=-1)
    return R


def pose_from_trig(cfg: Cfg, trig_azim, trig_elev, trig_theta, scale_dist):
    """Graph.pred_pose math, model/graph.py:272-289: R = Rz Rx Ry P, t = (0,0,dist*scale)."""
    P = torch.tensor([[-1, 0, 0], [0, 0, -1], [0, -1, 0]]).float().unsqueeze(0)
    R = roll_to_R(trig_theta) @ elev_to_R(trig_elev) @ azim_to_R(trig_azim) @ P.expand(len(trig_azim), 3, 3)
    pose_R = make_pose(R=R)
    tz = scale_dist * cfg.cam_dist
    pose_T = make_pose(t=torch.stack([torch.zeros_like(tz), torch.zeros_like(tz), tz], dim=-1))
    return pose_compose_pair(pose_R, pose_T)


# --------------------------------------------------------------------------------------
# renderer   (model/renderer.py)
# --------------------------------------------------------------------------------------
def get_z_vals(cfg: Cfg, n_total_rays: int, scale_dist: Tensor, training: bool,
               t_rand: Optional[Tensor], eik_idx: Tensor):
    """UniformSampler.get_z_vals, renderer.py:13-37.

    ``t_rand`` [n_rays_total, S] (training only) and ``eik_idx`` [n_rays_total] int64 are the
    CPU-generator draws the reference makes at renderer.py:29 and :33, passed explicitly.
    """
    B = scale_dist.shape[0]
    n_rays = n_total_rays // B
    sd = scale_dist.unsqueeze(-1).repeat(1, n_rays).view(n_total_rays, 1)
    near = cfg.cam_dist * sd - 0.7
    far = cfg.cam_dist * sd + 0.7
    t = torch.linspace(0.0, 1.0, steps=cfg.n_samples)
    z = near * (1.0 - t) + far * t
    if training:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    z_eik = torch.gather(z, 1, eik_idx.unsqueeze(-1))
    return z, z_eik


def volume_rendering(z_vals: Tensor, sdf: Tensor, beta_param: Tensor, beta_min: float = 1e-4):
    """Renderer.volume_rendering, renderer.py:187-209 -> (weights, alpha) [n_rays, S]."""
    density = laplace_density(sdf, beta_param, beta_min).reshape(-1, z_vals.shape[1])
    dists = z_vals[:, 1:] - z_vals[:, :-1]
    dists = torch.cat([dists, torch.zeros(dists.shape[0], 1)], -1)
    free_energy = dists * density
    shifted = torch.cat([torch.zeros(dists.shape[0], 1), free_energy[:, :-1]], dim=-1)
    alpha = 1 - torch.exp(-free_energy)
    transmittance = torch.exp(-torch.cumsum(shifted, dim=-1))
    return alpha * transmittance, alpha


def draw_render_randoms(n_total_rays: int, n_samples: int, training: bool, eik_range=(-1.0, 1.0)):
    """The reference's CPU-generator draws in their exact order (renderer.py:29, :33, :158)."""
    t_rand = torch.rand(n_total_rays, n_samples) if training else None
    eik_idx = torch.randint(n_samples, (n_total_rays,))
    eik_pts = torch.empty(n_total_rays, 3).uniform_(eik_range[0], eik_range[1]) if training else None
    return t_rand, eik_idx, eik_pts


def render(cfg: Cfg, W_sdf, W_rgb, beta_param: Tensor, pose: Tensor, intr: Tensor,
           scale_dist: Tensor, z_sdf: Tensor, z_rgb: Tensor, ray_idx: Optional[Tensor],
           training: bool, t_rand: Optional[Tensor], eik_idx: Tensor, eik_pts: Optional[Tensor]):
    """Renderer.forward, renderer.py:57-185 (normal_model='volume', perspective camera).

    Returns dict(rgb [B,R,3], mask [B,R,1], mask_hard [B,R,1], depth [B,R,1], normal [B,R,3],
    grad_eikonal [2BR] | None) plus intermediates (weights, alpha, sdf, z_vals) for finer tests.
    """
    cam_loc, ray_raw = get_center_and_ray(cfg, pose, intr)
    ray_dirs = F.normalize(ray_raw, dim=-1)
    depth_fac = ray_dirs.norm(dim=-1, keepdim=True) / ray_raw.norm(dim=-1, keepdim=True)
    if ray_idx is not None:
        g3 = ray_idx[..., None].repeat(1, 1, 3)
        ray_dirs = ray_dirs.gather(dim=1, index=g3)
        depth_fac = depth_fac.gather(dim=1, index=ray_idx[..., None])
    B, R, _ = ray_dirs.shape
    S = cfg.n_samples
    cam_loc = cam_loc.repeat(1, R, 1).reshape(-1, 3)
    ray_dirs = ray_dirs.reshape(-1, 3)
    depth_fac = depth_fac.reshape(-1, 1)

    z_vals, z_eik = get_z_vals(cfg, B * R, scale_dist, training, t_rand, eik_idx)
    points = cam_loc.unsqueeze(1) + z_vals.unsqueeze(2) * ray_dirs.unsqueeze(1)
    points_flat = points.reshape(-1, 3)
    lat_rgb = z_rgb.unsqueeze(1).repeat(1, R * S, 1).view(B * R * S, -1)

    with torch.enable_grad():
        points_flat.requires_grad_(True)
        sdf, feat, _ = sdf_conditional(cfg, W_sdf, B, points_flat, z_sdf, compute_grad=False)
        density = laplace_density(sdf, beta_param, cfg.beta_min)
        normal_flat = -torch.autograd.grad(density, points_flat, torch.ones_like(density),
                                           create_graph=True, retain_graph=True, only_inputs=True)[0]
    rgb_flat = rgb_mlp(cfg, W_rgb, points_flat, lat_rgb, feat)
    rgb = rgb_flat.reshape(-1, S, 3)

    weights, alphas = volume_rendering(z_vals, sdf, beta_param, cfg.beta_min)
    depth = torch.sum(weights * (z_vals * depth_fac), 1).unsqueeze(-1).view(B, -1, 1)

    normal = F.normalize(normal_flat, dim=-1, p=2).reshape(-1, S, 3)
    nw = weights.unsqueeze(-1) ** cfg.normal_pow
    normal_out = F.normalize(torch.sum(nw * normal, 1), dim=-1, p=2).view(B, -1, 3)

    acc = torch.sum(weights, -1)
    rgb_out = torch.sum(weights.unsqueeze(-1) * rgb, 1) + (1.0 - acc.unsqueeze(1).repeat(1, 3)) * cfg.bgcolor
    mask_hard = (acc > 0.5).float()

    grad_eik = None
    if training:
        eik = eik_pts.reshape(B, R, 3)
        near = (cam_loc.unsqueeze(1) + z_eik.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(B, R, 3)
        eik_all = torch.cat([eik, near], 1).reshape(-1, 3)
        _, _, g = sdf_conditional(cfg, W_sdf, B, eik_all, z_sdf, compute_grad=True)
        grad_eik = g.norm(2, dim=1)

    return dict(rgb=rgb_out.view(B, -1, 3), mask=acc.view(B, -1, 1), mask_hard=mask_hard.view(B, -1, 1),
                depth=depth, normal=normal_out, grad_eikonal=grad_eik,
                weights=weights, alpha=alphas, sdf=sdf, z_vals=z_vals, points=points_flat,
                normal_flat=normal_flat, rgb_flat=rgb_flat, feat=feat)


# --------------------------------------------------------------------------------------
# losses   (model/loss.py)
# --------------------------------------------------------------------------------------
def aggregate_loss(loss, weight=None):
    """loss.py:69-73."""
    if weight is not None:
        loss = loss * weight
    return loss.mean()


def l1_loss(pred, label=0, weight=None):
    """loss.py:15-17."""
    return aggregate_loss((pred.contiguous() - label).abs(), weight)


def mse_loss(pred, label=0, weight=None, tolerance=0.0):
    """loss.py:19-32 (incl. the sorted 'tolerance' path)."""
    loss = (pred.contiguous() - label) ** 2
    if tolerance > 1.0e-5:
        assert len(pred.shape) == 3 and pred.shape[2] in [1, 3]
        loss_pixel = loss.mean(dim=2).view(-1) if pred.shape[2] == 3 else loss.view(-1)
        loss_sorted = torch.sort(loss_pixel, dim=0, descending=False)[0]
        end_idx = int((1 - tolerance) * loss_sorted.shape[0])
        assert weight is None
        return aggregate_loss(loss_sorted[:end_idx].contiguous(), None)
    return aggregate_loss(loss, weight)


def iou_loss(inputs, targets, weight=None, tolerance=0.0):
    """loss.py:75-91."""
    B = inputs.shape[0]
    a = inputs.view(B, -1).contiguous()
    b = targets.view(B, -1).contiguous()
    if tolerance > 1.0e-5:
        assert weight is None
        n = a.shape[1]
        diff = (a - b).abs().view(B * n)
        idx_sorted = torch.sort(diff, dim=0, descending=False)[1]
        end_idx = int((1 - tolerance) * diff.shape[0])
        out = idx_sorted[end_idx:]
        a.view(B * n)[out] = b.view(B * n)[out]
    loss = 1 - (a * b).sum(dim=1) / (a + b - a * b + 1.0e-8).sum(dim=1)
    if weight is not None:
        loss = loss * weight.squeeze(1).squeeze(1)
    return loss.mean()


def mask_loss(cfg: Cfg, inputs, targets, weight=None, tolerance=0.0):
    """loss.py:93-97."""
    return iou_loss(inputs, targets, weight, tolerance) + cfg.mask_mse * mse_loss(inputs, targets, weight, tolerance)


def normal_loss(cfg: Cfg, normal_pred, normal_gt, mask, weight=None, tolerance=0.0):
    """loss.py:52-67: masked compaction, 5*L1 + angular, keep int(n*(1-tol)) smallest angular."""
    mask = mask.squeeze(-1)
    assert normal_pred.shape == normal_gt.shape and len(normal_pred.shape) == 3 and len(mask.shape) == 2
    cos_sim = torch.sum(normal_pred[mask] * normal_gt[mask], dim=-1)
    ang = 1 - cos_sim
    L1 = (normal_pred[mask] - normal_gt[mask]).abs().sum(dim=-1)
    loss = cfg.normal_l1 * L1 + ang
    idx = torch.sort(ang, dim=0, descending=False)[1][:int(loss.shape[0] * (1 - tolerance))]
    if weight is not None:
        loss = loss * weight.expand_as(normal_pred)[mask][..., 0]
    return loss[idx].mean()


def cam_margin(trig, ranges, eps=5):
    """loss.py:99-105."""
    assert ranges[0] > -180 and ranges[1] < 180
    angle = torch.atan2(trig[:, 1], trig[:, 0]) * 180 / np.pi
    return l1_loss((-angle + ranges[0] - eps).relu()) + l1_loss((angle - ranges[1] - eps).relu())


def cam_uniform_loss(cfg: Cfg, trig):
    """loss.py:134-167: sliced Wasserstein of (cos, sin, cos*sin) against a uniform grid."""
    B = trig.shape[0]
    cos_e, sin_e = trig[:, 0], trig[:, 1]
    prod_e = cos_e * sin_e
    grid = torch.arange(1.0, 2 * B, 2.0).float() * np.pi / B
    cos_p, sin_p = torch.cos(grid), torch.sin(grid)
    prod_p = cos_p * sin_p
    srt = lambda v: v.sort(dim=0, descending=False)[0]
    dc, ds, dp = srt(cos_p) - srt(cos_e), srt(sin_p) - srt(sin_e), srt(prod_p) - srt(prod_e)
    if cfg.emd_p == 1:
        return (dc.abs().mean() + ds.abs().mean() + dp.abs().mean()) / 3
    p = cfg.emd_p
    return (torch.norm(dc, dim=0, p=p) + torch.norm(ds, dim=0, p=p) + torch.norm(dp, dim=0, p=p)) / (3 * B)


def cam_sym_terms(trig_azim, trig_elev, trig_theta, flipped):
    """loss.py:113-132 given the estimator outputs on the flipped image (estimator itself is out of scope)."""
    fa, fe, ft = flipped
    la = (trig_azim[:, 0] - fa[:, 0]) ** 2 + (-trig_azim[:, 1] - fa[:, 1]) ** 2
    le = (trig_elev[:, 0] - fe[:, 0]) ** 2 + (trig_elev[:, 1] - fe[:, 1]) ** 2
    lt = (trig_theta[:, 0] - ft[:, 0]) ** 2 + (-trig_theta[:, 1] - ft[:, 1]) ** 2
    return la.mean() + le.mean() + lt.mean()


def nn_view_scores(mask_input, mask_input_NN, sample_temp):
    """Graph.forward_NN selection scores, graph.py:119-134 -> probs [B,K]."""
    B = mask_input.shape[0]
    K = mask_input_NN.shape[-1]
    ious = []
    for i in range(K):
        cur = mask_input_NN[..., i].view(B, -1)
        inp = mask_input.view(B, -1)
        ious.append((cur * inp).sum(dim=1) / (cur + inp - cur * inp + 1.0e-8).sum(dim=1))
    scores = (1 - torch.stack(ious, dim=-1)) ** sample_temp
    return F.normalize(scores, dim=-1, p=1)


# --------------------------------------------------------------------------------------
# eval geometry   (utils/eval_3D.py)
# --------------------------------------------------------------------------------------
def dense_grid(range_min: float, range_max: float, N: int, batch: int) -> Tensor:
    """get_dense_3D_grid, eval_3D.py:9-18 ('ij' meshgrid, N+1 samples/axis) -> [B,N+1,N+1,N+1,3]."""
    g = torch.linspace(range_min, range_max, N + 1)
    pts = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1)
    return pts.repeat(batch, 1, 1, 1, 1)


@torch.no_grad()
def level_grid(cfg: Cfg, W_sdf, z_sdf: Tensor, points_3D: Tensor) -> Tensor:
    """compute_level_grid, eval_3D.py:21-38: one x-slab at a time -> [B,N,N,N]."""
    B, N = points_3D.shape[0], points_3D.shape[1]
    out = []
    for i in range(N):
        flat = points_3D[:, i:i + 1].reshape(-1, 3)
        lat = z_sdf.unsqueeze(1).repeat(1, N * N, 1).view(B * N * N, -1)
        out.append(sdf_mlp(cfg, W_sdf, flat, lat)[:, :1].view(B, 1, N, N, 1))
    return torch.cat(out, dim=1)[..., 0]


def normalize_pc(pc: Tensor) -> Tensor:
    """eval_3D.py:41-49: zero-mean, divide by max(x-extent, y-extent)+1e-7."""
    assert len(pc.shape) == 3
    z = pc - pc.mean(dim=1, keepdim=True)
    lx = z[:, :, 0].max(dim=-1)[0] - z[:, :, 0].min(dim=-1)[0]
    ly = z[:, :, 1].max(dim=-1)[0] - z[:, :, 1].min(dim=-1)[0]
    lm = torch.stack([lx, ly], dim=-1).max(dim=-1)[0].unsqueeze(-1).unsqueeze(-1)
    return z / (lm + 1.0e-7)


def compute_fscore(dist1: Tensor, dist2: Tensor, thresholds=(0.005, 0.01, 0.02, 0.05, 0.1, 0.2)) -> Tensor:
    """eval_3D.py:105-121."""
    fs = []
    for th in thresholds:
        p = torch.mean((dist1 < th).float(), dim=1)
        r = torch.mean((dist2 < th).float(), dim=1)
        f = 2 * p * r / (p + r)
        f[torch.isnan(f)] = 0
        fs.append(f)
    return torch.stack(fs, dim=1)


def chamfer_forward_f32(xyz1: np.ndarray, xyz2: np.ndarray):
    """NmDistanceKernel semantics (chamfer3D.cu:12-134) in numpy fp32, for small clouds.

    d = fma(dz,dz, fma(dy,dy, dx*dx)) emulated in float64-then-round is NOT identical to fp32
    fma; use chamfer_ref.c (compiled) for bit-level checks.  This numpy version is the
    independent brute force used to cross-check the C oracle: squared distance in fp32
    (separate mul/add roundings), first-minimum (lowest index) tie rule.
    """
    a = xyz1.astype(np.float32)
    b = xyz2.astype(np.float32)
    d = ((a[:, :, None, :] - b[:, None, :, :]) ** 2)
    d = (d[..., 0] + d[..., 1]) + d[..., 2]
    idx1 = d.argmin(axis=2).astype(np.int32)
    idx2 = d.argmin(axis=1).astype(np.int32)
    return d.min(axis=2), d.min(axis=1), idx1, idx2


def chamfer_backward_ref(xyz1, xyz2, gd1, gd2, idx1, idx2):
    """NmDistanceGradKernel (chamfer3D.cu:155-174), both launches (:184-185), float64 accumulate."""
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.zeros((B, n, 3), np.float64)
    g2 = np.zeros((B, m, 3), np.float64)
    for b in range(B):
        j2 = idx1[b].astype(np.int64)
        g = (2.0 * gd1[b])[:, None] * (xyz1[b].astype(np.float64) - xyz2[b][j2].astype(np.float64))
        g1[b] += g
        np.add.at(g2[b], j2, -g)
        j1 = idx2[b].astype(np.int64)
        g = (2.0 * gd2[b])[:, None] * (xyz2[b].astype(np.float64) - xyz1[b][j1].astype(np.float64))
        g2[b] += g
        np.add.at(g1[b], j1, -g)
    return g1, g2
