"""Computation graph of one ShapeClipper step -- call surface of the reference's model/graph.py.

`Graph(opt)` is an nn.Module whose direct children are named exactly as in the reference
(estimator, sdf_network, rgb_network, renderer, encoder, latent_proj_shape, latent_proj_rgb,
loss_fns): those names are the checkpoint ABI (utils/util.py restore iterates named_children).
`forward(opt, var, training, get_loss, visualize)` fills the same `var` keys and returns the same
loss dictionary keys as reference graph.py:68-112,220-265.  The two render calls per training step
(input view, graph.py:92-93; CLIP-nearest-neighbour view, :207-209) run on the HIP renderer.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as torch_F

from ..utils import camera
from ..utils.util import EasyDict as edict
from . import loss, resnet
from .implicit import RGBNetwork, SDFNetwork
from .renderer import Renderer
from .view_estimator import Bottleneck_Linear as _EstimatorBottleneck
from .view_estimator import Estimator


class Bottleneck_Linear(_EstimatorBottleneck):
    """Latent projector block; unlike the estimator's heads bn2 is NOT zero-initialised (graph.py:16-40)."""

    def __init__(self, n_channels):
        super().__init__(n_channels, zero_init=False)


# world axes -> camera axes permutation applied before the Euler rotations (graph.py:278-283)
_AXIS_PERMUTE = [[-1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, -1.0, 0.0]]


def rotation_from_trig(trig_azim, trig_elev, trig_theta):
    Ry = camera.azim_to_rotation_matrix(trig_azim, representation="trig")
    Rx = camera.elev_to_rotation_matrix(trig_elev, representation="trig")
    Rz = camera.roll_to_rotation_matrix(trig_theta, representation="trig")
    return Rz @ Rx @ Ry @ _axis_permute(Ry.device).unsqueeze(0).expand_as(Ry)


_PERM_CACHE = {}
_SIDE_STREAMS = {}


def _axis_permute(device):
    """The constant lives on the device once (building it per call is a synchronising pageable H2D copy)."""
    key = str(device)
    if key not in _PERM_CACHE:
        _PERM_CACHE[key] = torch.tensor(_AXIS_PERMUTE, device=device)
    return _PERM_CACHE[key]


def _latent_projector(dim_in, dim_out):
    return nn.Sequential(Bottleneck_Linear(dim_in), Bottleneck_Linear(dim_in), nn.Linear(dim_in, dim_out))


def choice_from_uniform(probs, u):
    """np.random.choice(K, size=(1,), replace=False, p=probs[i] / np.sum(probs[i])) for every row i, given the uniform
    u[i] that call would draw (numpy mtrand.pyx legacy `choice`: cdf = cumsum(float64(p)); cdf /= cdf[-1];
    searchsorted(u, side='right')).  Same operation order as numpy (float32 sequential sum and division, float64
    sequential cumsum), so the picks are identical.  probs [B,K] float32, u [B] float64 -> (idx [B] int64, any-NaN flag)."""
    K = probs.shape[1]
    s = probs[:, 0]
    for k in range(1, K):
        s = s + probs[:, k]
    p = (probs / s[:, None]).double()
    cols = [p[:, 0]]
    for k in range(1, K):
        cols.append(cols[-1] + p[:, k])
    cdf = torch.stack(cols, dim=1)
    cdf = cdf / cdf[:, -1:]
    return (cdf <= u[:, None].to(cdf.dtype)).sum(dim=1), torch.isnan(cdf).any()


class Graph(nn.Module):

    def __init__(self, opt):
        super().__init__()
        self.estimator = Estimator(opt)
        self.sdf_network = SDFNetwork(opt)
        self.rgb_network = RGBNetwork(opt)
        self.renderer = Renderer(opt, self.sdf_network, self.rgb_network)
        self.encoder = resnet.build(opt.arch.enc_network, pretrained=opt.arch.enc_pretrained,
                                    allow_random=bool(opt.get("load")) or bool(opt.get("resume")))
        self.encoder.fc = nn.Linear(self.encoder.fc.in_features, opt.arch.latent_dim_shape + opt.arch.latent_dim_rgb)
        self.latent_proj_shape = _latent_projector(opt.arch.latent_dim_shape, opt.arch.impl_sdf.proj_latent_dim)
        self.latent_proj_rgb = _latent_projector(opt.arch.latent_dim_rgb, opt.arch.impl_rgb.proj_latent_dim)
        self.loss_fns = loss.Loss(opt)

    # ------------------------------------------------------------------------------------------------
    def forward(self, opt, var, training=False, get_loss=True, visualize=False):
        B = len(var.idx)
        sampled = bool(opt.render.rand_sample and training)
        ray_idx = var.ray_idx if sampled else None
        self.sdf_network.begin_step()       # one packed weight image per forward pass, shared by all renders
        self.rgb_network.begin_step()
        use_NN = (opt.loss_weight.nearest_img is not None or opt.loss_weight.nearest_mask is not None) and training
        # The neighbour choice depends only on the input masks; it needs one device->host read (numpy RNG, as the
        # reference).  Done first, while the stream is empty, the host never has to wait for the main render.
        for stale in ("_latent_batched", "_estim_input", "_estim_flip"):      # hand-over keys of a previous forward
            var.pop(stale, None)
        idx_NN = self.select_neighbours(opt, var) if use_NN else None
        views = self.gather_neighbour_views(opt, var, idx_NN, sampled) if use_NN else []
        if use_NN and "latent" not in var and opt.get("hip", {}).get("batched_encoders", True):
            self.encode_all_views(opt, var, views)

        # always (re)written, as in the reference (graph.py:73): a second forward of the same `var` after an optimiser
        # step or a train()/eval() switch must not reuse features of the old weights
        batched = var.pop("_latent_batched", None)
        var.latent_raw = var.latent if "latent" in var else (batched if batched is not None else self.encoder(var.rgb_input_map))
        var.latent_shape = var.latent_raw[:, :opt.arch.latent_dim_shape]
        var.latent_rgb = var.latent_raw[:, opt.arch.latent_dim_shape:]
        var.proj_latent_sdf = self.latent_proj_shape(var.latent_shape)
        # The colour projector runs on the input view here and on every neighbour view in forward_NN (reference graph.py:78, :198): when the
        # neighbours' latents are already known (batched encoders) all of them go through it ONCE, stacked, with per-set BatchNorm
        # statistics updated in the reference's call order -- one set of launches and one gradient per shared weight instead of one per view.
        nn_lat = [nn_in.latent_raw[:, opt.arch.latent_dim_shape:] for nn_in in views if "latent_raw" in nn_in] if use_NN else []
        if nn_lat and len(nn_lat) == len(views) and var.latent_rgb.is_cuda and opt.get("hip", {}).get("batched_encoders", True):
            G = 1 + len(nn_lat)
            proj = self._project_stacked(self.latent_proj_rgb, torch.cat([var.latent_rgb] + nn_lat, 0), G)
            pieces = proj.reshape(G, B, proj.shape[1]).unbind(0)
            var.proj_latent_rgb = pieces[0]
            for nn_in, piece in zip(views, pieces[1:]):
                nn_in.proj_latent_rgb = piece
        else:
            var.proj_latent_rgb = self.latent_proj_rgb(var.latent_rgb)

        var.pose, var.intr, var.scale_dist = self.pred_pose(opt, var)
        var.normal_transformed = self.transform_normal(var.normal_gt if "normal_gt" in var else var.normal_input, var.pose)

        out = self.renderer(opt, var.pose, var.intr, var.scale_dist, var.proj_latent_sdf, var.proj_latent_rgb,
                            ray_idx=ray_idx, training=training, visualize=visualize)
        var.rgb_recon, var.mask_recon, var.mask_hard, var.depth_recon, var.normal_recon, var.grad_eikonal = out[:6]
        if visualize:
            var.rendering_points, var.rendering_transparency, var.rendering_rgb = out[6:9]

        if not sampled:
            as_map = lambda x, c, h, w: x.view(B, h, w, c).permute(0, 3, 1, 2).contiguous()
            var.rgb_recon_map = as_map(var.rgb_recon, 3, opt.H, opt.W)
            var.mask_recon_map = as_map(var.mask_recon, 1, opt.H, opt.W)
            var.mask_hard_map = as_map(var.mask_hard, 1, opt.H, opt.W)
            var.normal_recon_map = as_map(var.normal_recon, 3, opt.H, opt.W)
            var.normal_transformed_map = as_map(var.normal_transformed, 3, opt.image_size[0], opt.image_size[1])

        if use_NN:
            self.forward_NN(opt, var, views=views)

        # the shared weight images are only valid inside this forward pass: drop them (they carry autograd history and
        # must not end up in a deepcopy / pickle of the module)
        self.sdf_network.begin_step(False)
        self.rgb_network.begin_step(False)
        if get_loss:
            return var, self.compute_loss(opt, var, training)
        return var

    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def select_neighbours(self, opt, var):
        """IoU-weighted choice of n_views of the K CLIP neighbours per image (graph.py:119-142).

        The reference reads the probabilities back to the host once per image and calls np.random.choice.  For
        n_views == 1 (the shipped config) that call consumes exactly one uniform of numpy's global stream per image and
        does a searchsorted on the normalised cdf: here the B uniforms are drawn on the host in image order (same
        stream position afterwards), uploaded asynchronously, and the cdf search runs on the device in numpy's own
        operation order (`choice_from_uniform`, bit-identical picks, tests/test_host_logic.py) -- NO device->host copy,
        the host never waits for the stream.  n_views > 1 draws a data-dependent number of uniforms: host path."""
        B, K = len(var.idx), opt.data.k_nearest
        inp = var.mask_input.view(B, -1, 1)
        nn_masks = var.mask_input_NN.reshape(B, -1, K)
        inter = (nn_masks * inp).sum(dim=1)
        union = (nn_masks + inp - nn_masks * inp + 1.e-8).sum(dim=1)
        probs = torch_F.normalize((1 - inter / union) ** opt.reg.sample_temp, dim=-1, p=1)
        dev = var.rgb_input_map.device
        if opt.reg.n_views == 1 and opt.get("hip", {}).get("device_choice", True):
            u = torch.from_numpy(np.random.random_sample(B))
            if probs.is_cuda:
                u = u.pin_memory().to(probs.device, non_blocking=True)
            idx, nan = choice_from_uniform(probs, u)
            var._bad_choice = nan                  # np.random.choice raises on NaN probabilities: joins the finite check
            return idx.view(B, 1).to(dev)
        probs = probs.cpu().numpy()
        picks = []
        for i in range(B):
            p = probs[i] / np.sum(probs[i])
            picks.append(np.random.choice(K, size=(opt.reg.n_views,), replace=False, p=p))
        return torch.tensor(np.stack(picks, axis=0)).long().to(dev)

    def gather_neighbour_views(self, opt, var, idx_NN, sampled):
        """The n_views chosen neighbours of every image as batch dictionaries (reference graph.py:144-193)."""
        B = len(var.idx)
        assert opt.reg.n_views <= opt.data.k_nearest
        if sampled:
            assert len(var.ray_idx.shape) == 2
        rows = torch.arange(B, device=idx_NN.device)
        views = []
        for v in range(opt.reg.n_views):
            pick = lambda stack: stack[rows, ..., idx_NN[:, v]]          # [B, ..., K] -> [B, ...]
            nn_in = edict()
            nn_in.rgb_input_map = pick(var.rgb_input_map_NN)
            nn_in.mask_input_map = pick(var.mask_input_map_NN)
            nn_in.normal_input_map = pick(var.normal_input_map_NN)
            nn_in.rgb_input = pick(var.rgb_input_NN)
            nn_in.mask_input = pick(var.mask_input_NN)
            nn_in.normal_input = pick(var.normal_input_NN)
            if sampled:
                nn_in.ray_idx = pick(var.ray_idx_NN)
            nn_in.pose_gt = pick(var.pose_gt_NN)
            var["input_NN_{}".format(v)] = nn_in
            views.append(nn_in)
        return views

    def encode_all_views(self, opt, var, views):
        """ONE encoder pass and ONE estimator pass for everything a training step needs.

        The reference calls the encoder on the input images and again on each neighbour view (graph.py:73,197), and
        the estimator on the input, on each neighbour and on the mirrored input (graph.py:272, :204, loss.py:114): up to
        five ResNet passes whose inputs are all known at the start of the step.  Here the image sets are stacked along
        the batch dimension and go through each network once with `groups` = number of sets: convolutions see one
        large batch, BatchNorm keeps per-set statistics and applies its running-statistics updates in the reference's
        call order, so the result equals the sequential calls while the step spends 2 instead of 5 passes worth of
        launches and every shared weight receives a single gradient."""
        B = len(var.idx)
        images = [var.rgb_input_map] + [nn_in.rgb_input_map for nn_in in views]
        mirrored = [var.rgb_input_map.flip(dims=[3])] if opt.loss_weight.cam_sym is not None else []
        est_in = torch.cat(images + mirrored, 0)
        # The two networks are independent: the estimator pass runs on a second HIP stream next to the encoder pass
        # (HBM-bound BatchNorm of one overlaps MFMA-bound convolutions of the other; autograd replays the backward of
        # every operator on the stream of its forward, so the backward passes overlap the same way).
        side = self._side_stream(est_in.device) if (est_in.is_cuda and opt.get("hip", {}).get("two_streams", True)) else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                est = self.estimator(est_in, groups=len(images) + len(mirrored), split=True)
        latent = self.encoder(torch.cat(images, 0), groups=len(images))
        if side is not None:
            main.wait_stream(side)
            for st in est:
                for t in st:
                    t.record_stream(main)
            est_in.record_stream(side)
        else:
            est = self.estimator(est_in, groups=len(images) + len(mirrored), split=True)
        # one unbind node (a stack in the backward) instead of a slice node per image set (fill + copy + add each)
        latents = latent.reshape(len(images), B, latent.shape[1]).unbind(0)
        var._latent_batched = latents[0]
        for v, nn_in in enumerate(views):
            nn_in.latent_raw = latents[v + 1]
        var._estim_input = est[0]
        for v, nn_in in enumerate(views):
            nn_in.estim = est[v + 1]
        if mirrored:
            var._estim_flip = est[len(images)]

    @staticmethod
    def _project_stacked(projector, x, groups):
        """A latent projector (Bottleneck_Linear, Bottleneck_Linear, Linear) on `groups` stacked sub-batches: the result of `groups` calls."""
        for m in projector:
            x = m(x, groups=groups) if isinstance(m, _EstimatorBottleneck) else m(x)
        return x

    @staticmethod
    def _side_stream(device):
        """Second HIP stream of a device (process-wide: module attributes must stay deep-copyable / picklable)."""
        key = str(device)
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        return _SIDE_STREAMS[key]

    def forward_NN(self, opt, var, training=True, idx_NN=None, views=None):
        B = len(var.idx)
        sampled = bool(opt.render.rand_sample and training)
        if views is None:
            if idx_NN is None:
                idx_NN = self.select_neighbours(opt, var)               # [B, V]
            views = self.gather_neighbour_views(opt, var, idx_NN, sampled)
        for v, nn_in in enumerate(views):
            ray_idx = nn_in.ray_idx if sampled else None
            latent_NN = nn_in.latent_raw if "latent_raw" in nn_in else self.encoder(nn_in.rgb_input_map)
            proj_latent_rgb_NN = (nn_in.proj_latent_rgb if "proj_latent_rgb" in nn_in
                                  else self.latent_proj_rgb(latent_NN[:, opt.arch.latent_dim_shape:]))
            var.proj_latent_rgb_NN = proj_latent_rgb_NN
            pose_NN, intr_NN, scale_NN = self.pred_pose(opt, var, pred_NN=True, given_input=nn_in.rgb_input_map,
                                                        estim=nn_in.get("estim"))
            var["pose_NN_{}".format(v)], var["intr_NN_{}".format(v)], var["scale_dist_NN_{}".format(v)] = pose_NN, intr_NN, scale_NN

            # the neighbour is rendered with the INPUT image's shape code and its own colour code
            rgb, mask, _, depth, normal, _ = self.renderer(opt, pose_NN, intr_NN, scale_NN, var.proj_latent_sdf,
                                                           proj_latent_rgb_NN, ray_idx=ray_idx, training=training)
            var["rgb_recon_NN_{}".format(v)], var["mask_recon_NN_{}".format(v)] = rgb, mask
            var["depth_recon_NN_{}".format(v)], var["normal_recon_NN_{}".format(v)] = depth, normal
            if not sampled:
                as_map = lambda x, c: x.view(B, opt.H, opt.W, c).permute(0, 3, 1, 2).contiguous()
                var["rgb_recon_map_NN_{}".format(v)] = as_map(rgb, 3)
                var["mask_recon_map_NN_{}".format(v)] = as_map(mask, 1)
                var["normal_recon_map_NN_{}".format(v)] = as_map(normal, 3)

    # ------------------------------------------------------------------------------------------------
    def compute_loss(self, opt, var, training=False):
        out = edict()
        B = len(var.idx)
        w3 = var.category_weight.view(B, 1, 1) if "category_weight" in var else None
        w2 = var.category_weight.view(B, 1) if "category_weight" in var else None
        lw, fns = opt.loss_weight, self.loss_fns
        mask_gt = var.mask_gt if "mask_gt" in var else var.mask_input
        fused = (w3 is None and var.rgb_recon.is_cuda and lw.render is not None and lw.mask is not None
                 and lw.normal is not None and var.rgb_recon.dim() == 3 and opt.get("hip", {}).get("fused_loss", True))
        if fused:
            return self.compute_loss_fused(opt, var, training, mask_gt)
        if lw.render is not None:
            out.render = fns.MSE_loss(var.rgb_recon, var.rgb_gt if "rgb_gt" in var else var.rgb_input, weight=w3)
        if lw.mask is not None:
            out.mask = fns.mask_loss(var.mask_recon, mask_gt, weight=w3)
        if lw.normal is not None:
            valid = (mask_gt > 0.5) & (var.mask_recon > 0.5)
            out.normal = fns.normal_loss(var.normal_recon, var.normal_transformed, valid, weight=w3, tolerance=opt.reg.normal_tol)
        if training:
            if lw.eikonal is not None:
                out.eikonal = fns.MSE_loss(var.grad_eikonal.view(B, -1), 1, weight=w2)
            if lw.cam_margin is not None:
                out.cam_margin = fns.cam_margin_loss(opt, var)
            if lw.cam_uniform is not None:
                out.cam_uniform = fns.cam_uniform_loss(opt, var.trig_azim)
            if lw.cam_sym is not None:
                out.cam_sym = fns.cam_sym_loss(opt, var, self.estimator)
            views = [(var["input_NN_{}".format(v)], v) for v in range(opt.reg.n_views)] if \
                (lw.nearest_img is not None or lw.nearest_mask is not None or lw.nearest_normal is not None) else []
            if lw.nearest_img is not None:
                out.nearest_img = sum(fns.MSE_loss(var["rgb_recon_NN_{}".format(v)], nn_in.rgb_input, weight=w3) for nn_in, v in views)
            if lw.nearest_mask is not None:
                out.nearest_mask = sum(fns.mask_loss(var["mask_recon_NN_{}".format(v)], nn_in.mask_input, weight=w3) for nn_in, v in views)
            if lw.nearest_normal is not None:
                total = 0
                for nn_in, v in views:
                    valid = (nn_in.mask_input > 0.5) & (var["mask_recon_NN_{}".format(v)] > 0.5)
                    target = camera.transform_normal(nn_in.normal_input, var["pose_NN_{}".format(v)])
                    total = total + fns.normal_loss(var["normal_recon_NN_{}".format(v)], target, valid, weight=w3,
                                                    tolerance=opt.reg.normal_tol)
                out.nearest_normal = total
        return out

    def compute_loss_fused(self, opt, var, training, mask_gt):
        """Same dictionary as compute_loss, with the [B,R,*] reductions of each render done by ONE HIP launch
        (csrc/loss.hip) instead of ~60 small torch kernels, a sort and two boolean-index syncs."""
        from ..functional import FusedRenderLoss
        out = edict()
        B = len(var.idx)
        lw, fns = opt.loss_weight, self.loss_fns
        keep = 1 - opt.reg.normal_tol
        rgb_gt = var.rgb_gt if "rgb_gt" in var else var.rgb_input
        eik = var.grad_eikonal.view(B, -1) if (training and lw.eikonal is not None) else None
        main = FusedRenderLoss.apply(var.rgb_recon, rgb_gt, var.mask_recon, mask_gt, var.normal_recon,
                                     var.normal_transformed, eik, float(opt.reg.normal_l1), float(opt.reg.mask_mse), keep)
        out.render, out.mask, out.normal = main[0], main[1], main[2]
        if training:
            if eik is not None:
                out.eikonal = main[3]
            priors = fns.camera_prior_losses(opt, var, self.estimator)      # the three of them in one launch where possible
            for key in ("cam_margin", "cam_uniform", "cam_sym"):
                if lw[key] is not None:
                    out[key] = priors[key]()
            if lw.nearest_img is not None or lw.nearest_mask is not None or lw.nearest_normal is not None:
                tot = None
                for v in range(opt.reg.n_views):
                    nn_in = var["input_NN_{}".format(v)]
                    target = self.transform_normal(nn_in.normal_input, var["pose_NN_{}".format(v)])
                    r = FusedRenderLoss.apply(var["rgb_recon_NN_{}".format(v)], nn_in.rgb_input, var["mask_recon_NN_{}".format(v)],
                                              nn_in.mask_input, var["normal_recon_NN_{}".format(v)], target, None,
                                              float(opt.reg.normal_l1), float(opt.reg.mask_mse), keep)
                    tot = r if tot is None else tuple(a + b for a, b in zip(tot, r))
                if lw.nearest_img is not None:
                    out.nearest_img = tot[0]
                if lw.nearest_mask is not None:
                    out.nearest_mask = tot[1]
                if lw.nearest_normal is not None:
                    out.nearest_normal = tot[2]
        return out

    @staticmethod
    def transform_normal(normals, pose):
        """camera.transform_normal; [B,R,3] device normals take one HIP launch each way (csrc/camera_prior.hip)."""
        if normals.is_cuda and normals.dim() == 3 and normals.shape[-1] == 3 and tuple(pose.shape[1:]) == (3, 4) and not normals.requires_grad:
            from ..functional import TransformNormalFunction
            return TransformNormalFunction.apply(normals, pose)
        return camera.transform_normal(normals, pose)

    # ------------------------------------------------------------------------------------------------
    def pred_pose(self, opt, var, pred_NN=False, given_input=None, estim=None):
        image = given_input if given_input is not None else var.rgb_input_map
        if estim is None and not pred_NN and "_estim_input" in var:
            estim = var.pop("_estim_input")
        trig_azim, trig_elev, trig_theta, scale_focal, scale_dist = estim if estim is not None else self.estimator(image)
        if trig_azim.is_cuda:      # one launch each way instead of ~50 [B]-sized torch operators (csrc/camera.hip)
            from ..functional import PoseFromTrigFunction
            pose, intr = PoseFromTrigFunction.apply(trig_azim, trig_elev, trig_theta, scale_focal, scale_dist,
                                                    float(opt.camera.dist), float(opt.camera.focal), int(opt.W), int(opt.H))
        else:
            pose_R = camera.pose(R=rotation_from_trig(trig_azim, trig_elev, trig_theta))
            tz = scale_dist * opt.camera.dist
            pose_T = camera.pose(t=torch.stack([torch.zeros_like(tz), torch.zeros_like(tz), tz], dim=-1))
            pose = camera.pose.compose([pose_R, pose_T]).to(image.device)
            intr = camera.get_intr(opt, scale_focal)
        if not pred_NN:
            var.trig_azim, var.trig_elev, var.trig_theta = trig_azim, trig_elev, trig_theta
            var.scale_focal, var.scale_dist = scale_focal, scale_dist
        return pose, intr, scale_dist

    @torch.no_grad()
    def get_rotate_pose(self, opt, var, n_views=50):
        """[n_views,3,4] turn-table poses for visualisation (graph.py:295-322)."""
        dev = var.rgb_input_map.device
        r = opt.data[opt.data.dataset]
        azim = torch.linspace(0, 2, n_views).to(dev).view(n_views, 1) * np.pi
        elev = torch.zeros(n_views, 1).to(dev) + ((r.elev_range[1] + r.elev_range[0]) / 2 + 15) * np.pi / 180
        theta = torch.zeros(n_views, 1).to(dev) + ((r.theta_range[1] + r.theta_range[0]) / 2) * np.pi / 180
        trig = lambda a: torch.cat([torch.cos(a), torch.sin(a)], dim=1)
        pose_R = camera.pose(R=rotation_from_trig(trig(azim), trig(elev), trig(theta))).to(dev)
        pose_cam = camera.pose(t=[0, 0, opt.camera.dist]).to(dev)
        var.vis_pose = camera.pose.compose([pose_R, pose_cam]).to(dev)
        return var
