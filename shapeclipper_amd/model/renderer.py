"""VolSDF-style volume renderer of the MI355X build.

Call surface = reference model/renderer.py:  Renderer(opt, sdf_network, rgb_network) and
forward(opt, pose, intr, scale_dist, proj_latent_sdf, proj_latent_rgb, ray_idx=None, training=True,
visualize=False) -> 6-tuple (9-tuple with visualize).  What differs is where the work happens:

  reference                                   this build
  ---------                                   ----------
  all H*W rays, then gather(ray_idx)          rays only for the rendered pixels (utils/camera.py)
  ~40 torch ops / ~40 KB of temporaries       3 fused HIP kernels per render, ~0.6 KB/pt of HBM traffic
  per sample point, autograd double backward  hand-derived backward kernels (sdf_bwd / rgb_bwd / wgrad)

Random numbers: exactly the reference's CPU-generator draws, in its order (rand [BR,64] -> randint
[BR] -> uniform_ [BR,3]; renderer.py:29,33,158), so a seeded run consumes the same stream.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as torch_F

from ..functional import CameraRaysFunction, RaySampleEikFunction, RaySampleFunction, RgbCompositeFunction, SdfFunction
from ..utils import camera
from .implicit import LaplaceDensity



UPLOAD_STREAM = True          # `--hip.upload_stream!`: the CPU-generator draws are copied in the render's own stream
_upload_streams = {}


def _upload(x, dev):
    """Pinned host tensor -> device on the device's upload stream; the current stream waits for it (an event, no host wait)."""
    main = torch.cuda.current_stream(dev)
    side = _upload_streams.get(dev.index)
    if side is None:
        side = _upload_streams[dev.index] = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        d = x.to(dev, non_blocking=True)        # the pinned block is held by the host allocator until this copy has run
    main.wait_stream(side)
    d.record_stream(main)                       # allocated on the upload stream, used on the render's
    return d


class UniformSampler(nn.Module):
    """Stratified depth samples in [dist*s - 0.7, dist*s + 0.7] (reference model/renderer.py:8-37)."""

    def __init__(self, opt):
        super().__init__()
        self.N_samples = opt.render.n_samples_uniform

    def get_z_vals(self, opt, ray_dirs, scale_dist, training=True):
        n_total = ray_dirs.shape[0]
        dev = ray_dirs.device
        n_rays = n_total // scale_dist.shape[0]
        centre = (opt.camera.dist * scale_dist).repeat_interleave(n_rays).view(n_total, 1)
        near, far = centre - 0.7, centre + 0.7
        t = torch.linspace(0.0, 1.0, steps=self.N_samples).to(dev)
        z_vals = near * (1.0 - t) + far * t
        if training:
            mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
            upper = torch.cat([mids, z_vals[..., -1:]], -1)
            lower = torch.cat([z_vals[..., :1], mids], -1)
            t_rand = torch.rand(z_vals.shape).to(dev)              # CPU generator, as the reference
            z_vals = lower + (upper - lower) * t_rand
        idx = torch.randint(z_vals.shape[-1], (n_total,)).to(dev)  # CPU generator
        return z_vals, torch.gather(z_vals, 1, idx.unsqueeze(-1))


class Renderer(nn.Module):

    def __init__(self, opt, sdf_network, rgb_network):
        super().__init__()
        self.bg_color = float(opt.data.bgcolor)
        self.eik_range = opt.arch.impl_sdf.eikonal_sample_range
        self.normal_model = opt.render.normal_model
        self.sdf_network = sdf_network
        self.rgb_network = rgb_network
        self.density = LaplaceDensity(params_init={"beta": opt.arch.impl_sdf.beta_init})
        if opt.render.sampler != "uniform":
            raise NotImplementedError(opt.render.sampler)
        if self.normal_model != "volume":
            raise NotImplementedError("only render.normal_model=volume is implemented (the shipped setting)")
        self.ray_sampler = UniformSampler(opt)
        self.N_samples = opt.render.n_samples_uniform
        # The compositing kernel maps the 64 samples of a ray onto one 64-lane wavefront and the chain kernels hold one architecture family
        # in LDS: any other render.n_samples_uniform / arch.impl_* runs on stock device operators (model/eager_path.py, round 5).
        self.eager = bool(getattr(sdf_network, "eager", False) or getattr(rgb_network, "eager", False) or self.N_samples != 64)
        if self.eager:
            from . import eager_path
            eager_path.warn_once("render.n_samples_uniform = %d" % self.N_samples if self.N_samples != 64 else "implicit networks")
            sdf_network.eager = rgb_network.eager = True          # one path for the whole render (the HIP kernels hand TBL64 features to each other)

    def forward(self, opt, pose, intr, scale_dist, proj_latent_sdf, proj_latent_rgb, ray_idx=None, training=True,
                visualize=False):
        S = self.N_samples
        sym = bool(self.sdf_network.force_symmetry)
        if opt.camera.model == "perspective" and pose.is_cuda:
            # one launch: pixel centres -> K^-1 -> camera-to-world -> unit rays + depth factor (csrc/camera.hip)
            B = pose.shape[0]
            R = ray_idx.shape[1] if ray_idx is not None else opt.H * opt.W
            cam_loc, ray_dirs, depth_fac = CameraRaysFunction.apply(pose, intr, ray_idx, R, int(opt.W))
        else:
            cam_loc, ray_raw = camera.get_center_and_ray(opt, pose, intr=intr, device=pose.device, ray_idx=ray_idx)
            ray_dirs = torch_F.normalize(ray_raw, dim=-1)
            depth_fac = ray_dirs.norm(dim=-1, keepdim=True) / ray_raw.norm(dim=-1, keepdim=True)
            B, R, _ = ray_dirs.shape
            if opt.camera.model == "perspective":
                cam_loc = cam_loc.expand(B, R, 3)
            cam_loc = cam_loc.reshape(-1, 3)
            ray_dirs = ray_dirs.reshape(-1, 3)
            depth_fac = depth_fac.reshape(-1)

        # reference CPU-generator draws, in its order (renderer.py:29,33): jitter then the eikonal sample index
        dev_rng = bool(opt.get("hip", {}).get("device_rng", False))   # True: draw on the GPU (no 4 MB H2D copy per render,
        rdev = ray_dirs.device if dev_rng else "cpu"                   # but a different random stream than the reference)
        # CPU draws land in pinned memory and are copied asynchronously: a pageable H2D copy would drain the stream
        # (one host sync per draw, three per render) and let the GPU idle while the host catches up.
        pin = (not dev_rng) and ray_dirs.is_cuda
        # ... and they travel on an UPLOAD stream of their own (round 5): the host is milliseconds ahead of the GPU when it reaches a render, so
        # the 4 MB of jitter are on the device long before the render's stream gets there -- in that stream the copy was ~100 us per render
        # with nothing else running (`--hip.upload_stream!`: copy in the render's stream)
        if pin and UPLOAD_STREAM:
            up = lambda x: _upload(x, ray_dirs.device)
        else:
            up = lambda x: x.to(ray_dirs.device, non_blocking=True)
        t_rand = up(torch.rand(B * R, S, device=rdev, pin_memory=pin)) if training else None
        eik_idx = up(torch.randint(S, (B * R,), device=rdev, pin_memory=pin))
        if self.eager:
            return self._forward_eager(opt, cam_loc, ray_dirs, depth_fac, scale_dist, t_rand, eik_idx, B, R, proj_latent_sdf, proj_latent_rgb,
                                       training, visualize, up, rdev, pin)
        eik_points = None
        if training and ray_dirs.is_cuda:
            # the eikonal points come out of the sampling launch: the uniform draw (CPU generator, the reference's third draw of a render,
            # renderer.py:158 -- nothing else touches the generator in between) and, per ray, the sample eik_idx as its near-surface point
            eik_u = up(torch.empty(B * R, 3, device=rdev, pin_memory=pin).uniform_(self.eik_range[0], self.eik_range[1]))
            z_vals, points_flat, eik_points = RaySampleEikFunction.apply(cam_loc, ray_dirs, scale_dist, t_rand, eik_idx, eik_u, R, float(opt.camera.dist))
        else:
            z_vals, points_flat = RaySampleFunction.apply(cam_loc, ray_dirs, scale_dist, t_rand, R, float(opt.camera.dist))
            z_eik = torch.gather(z_vals, 1, eik_idx.unsqueeze(-1))
        assert proj_latent_rgb.shape[1] == opt.arch.impl_rgb.proj_latent_dim

        # fused SDF value + feature + d(sdf)/dx, then RGB MLP + density + compositing
        w_pack, cbias = self.sdf_network.packed(proj_latent_sdf)
        fused_bwd = bool(opt.get("hip", {}).get("fused_backward", True))
        sdf, grad, feat = SdfFunction.apply(points_flat, w_pack, cbias, R * S, sym, True, True, fused_bwd)
        v_pack, dbias = self.rgb_network.packed(proj_latent_rgb)
        outs = RgbCompositeFunction.apply(points_flat, z_vals, depth_fac.contiguous(), sdf, grad, feat,
                                          v_pack, dbias, self.density.beta, R, sym, float(self.density.beta_min),
                                          self.bg_color, float(opt.reg.normal_pow), bool(visualize))
        rgb, mask, mask_hard, depth, normal = outs[:5]
        rgb_output = rgb.view(B, R, 3)
        mask_output = mask.view(B, R, 1)
        mask_hard_output = mask_hard.view(B, R, 1)
        depth_output = depth.view(B, R, 1)
        normal_output = normal.view(B, R, 3)

        grad_eikonal = None
        if training:
            # uniform points (CPU generator, as the reference) + one near-surface point per ray
            if eik_points is None:
                n_eik = B * R
                eik = up(torch.empty(n_eik, 3, device=rdev, pin_memory=pin).uniform_(self.eik_range[0], self.eik_range[1])).reshape(B, R, 3)
                near = (cam_loc + z_eik * ray_dirs).reshape(B, R, 3)
                eik_points = torch.cat([eik, near], 1)
            eik_points = eik_points.reshape(-1, 3)
            _, _, g_eik = self.sdf_network.get_conditional_output(opt, B, eik_points, proj_latent_sdf, compute_grad=True)
            grad_eikonal = g_eik.norm(2, dim=1)

        if visualize:
            weights, alphas, rgb_flat = outs[5:8]
            opacity = alphas.reshape(B, -1, 1)
            transp = torch.cat([opacity, 1 - opacity, torch.zeros_like(opacity)], dim=-1)
            rgba = torch.cat([rgb_flat.reshape(B, -1, 3), opacity], dim=-1)
            idx = torch.randperm(R)[:200].to(opacity.device)
            pick = lambda x: self.sample_rays_visualize(idx, x.reshape(B, R, S, -1))
            return (rgb_output, mask_output, mask_hard_output, depth_output, normal_output, grad_eikonal,
                    pick(points_flat.detach()), pick(transp), pick(rgba))
        return rgb_output, mask_output, mask_hard_output, depth_output, normal_output, grad_eikonal

    def _forward_eager(self, opt, cam_loc, ray_dirs, depth_fac, scale_dist, t_rand, eik_idx, B, R, latent_sdf, latent_rgb, training, visualize,
                       up, rdev, pin):
        """The render on stock device operators (other architectures / sample counts): same random draws in the same order, same outputs."""
        from . import eager_path
        S = self.N_samples
        cam_loc, ray_dirs, depth_fac = cam_loc.reshape(-1, 3), ray_dirs.reshape(-1, 3), depth_fac.reshape(-1)
        centre = (opt.camera.dist * scale_dist).repeat_interleave(R).view(B * R, 1)
        t = torch.linspace(0.0, 1.0, steps=S, device=ray_dirs.device)
        z_vals = (centre - 0.7) * (1.0 - t) + (centre + 0.7) * t                      # reference renderer.py:17-24
        if training:
            mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
            upper, lower = torch.cat([mids, z_vals[..., -1:]], -1), torch.cat([z_vals[..., :1], mids], -1)
            z_vals = lower + (upper - lower) * t_rand
        z_eik = torch.gather(z_vals, 1, eik_idx.unsqueeze(-1))
        eik = None
        if training:
            eik = up(torch.empty(B * R, 3, device=rdev, pin_memory=pin).uniform_(self.eik_range[0], self.eik_range[1])).reshape(B, R, 3)
        out = eager_path.render(self, opt, cam_loc, ray_dirs, depth_fac, z_vals, z_eik, eik, B, R, latent_sdf, latent_rgb, training)
        if not visualize:
            return out[:6]
        points_flat, alphas, rgb_flat = out[6]
        opacity = alphas.reshape(B, -1, 1)
        transp = torch.cat([opacity, 1 - opacity, torch.zeros_like(opacity)], dim=-1)
        rgba = torch.cat([rgb_flat.reshape(B, -1, 3), opacity], dim=-1)
        idx = torch.randperm(R)[:200].to(opacity.device)
        pick = lambda x: self.sample_rays_visualize(idx, x.reshape(B, R, S, -1))
        return out[:6] + (pick(points_flat.detach()), pick(transp), pick(rgba))

    def volume_rendering(self, z_vals, sdf):
        """Standalone torch form of reference renderer.py:187-209 for external callers (small tensors)."""
        density = self.density(sdf).reshape(-1, z_vals.shape[1])
        dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.zeros_like(z_vals[:, :1])], -1)
        free_energy = dists * density
        shifted = torch.cat([torch.zeros_like(free_energy[:, :1]), free_energy[:, :-1]], dim=-1)
        alpha = 1 - torch.exp(-free_energy)
        return alpha * torch.exp(-torch.cumsum(shifted, dim=-1)), alpha

    @torch.no_grad()
    def sample_rays_visualize(self, idx, item):
        item = item[:, idx].clone()
        return item.reshape(item.shape[0], -1, item.shape[-1])
