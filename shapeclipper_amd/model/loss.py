"""Loss terms with the reference's method names (model/loss.py).  The per-step hot reductions
(render MSE, mask IoU(+MSE), robust masked normal loss, eikonal MSE) are computed by the fused HIP
kernel (csrc/loss.hip) through `fused_render_losses`; the remaining camera priors are a handful
of [B]-sized torch ops."""
from __future__ import annotations

from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as torch_F


class Loss(nn.Module):

    def __init__(self, opt):
        super().__init__()
        self.opt = deepcopy(opt)

    # ---- generic reductions (reference loss.py:15-32, 69-73) ----------------------------------------
    def aggregate_loss(self, loss, weight=None):
        if weight is not None:
            loss = loss * weight
        return loss.mean()

    def L1_loss(self, pred, label=0, weight=None):
        return self.aggregate_loss((pred.contiguous() - label).abs(), weight=weight)

    def MSE_loss(self, pred, label=0, weight=None, tolerance=0.):
        loss = (pred.contiguous() - label) ** 2
        if tolerance > 1.e-5:
            assert len(pred.shape) == 3 and pred.shape[2] in [1, 3] and weight is None
            per_pixel = loss.mean(dim=2).view(-1) if pred.shape[2] == 3 else loss.view(-1)
            keep = int((1 - tolerance) * per_pixel.shape[0])
            return torch.sort(per_pixel, dim=0, descending=False)[0][:keep].contiguous().mean()
        return self.aggregate_loss(loss, weight=weight)

    def CE_loss(self, pred, label, weight=None, mask=None):
        return self.aggregate_loss(torch_F.cross_entropy(pred, label, reduction="none"), weight=weight)

    def BCE_loss(self, pred, label, weight=None, mask=None, tolerance=0.):
        """The reference's BCE_loss passes an unknown keyword to aggregate_loss and raises TypeError when
        called (loss.py:49-50; it has no caller).  Kept callable here, `mask` is ignored."""
        loss = torch_F.binary_cross_entropy(pred, label.expand_as(pred), reduction="none")
        if tolerance > 1.e-5:
            assert len(pred.shape) == 4 and pred.shape[1] == 1
            flat = loss.view(pred.shape[0], -1)
            keep = int((1 - tolerance) * flat.shape[1])
            loss = torch.sort(flat, dim=-1, descending=False)[0][:, :keep].contiguous()
        return self.aggregate_loss(loss, weight=weight)

    # ---- silhouette (loss.py:75-97) -----------------------------------------------------------------
    def iou_loss(self, inputs, targets, weight=None, tolerance=0.):
        B = inputs.shape[0]
        a = inputs.view(B, -1).contiguous()
        b = targets.view(B, -1).contiguous()
        if tolerance > 1.e-5:
            assert weight is None
            n = a.shape[1]
            diff = (a - b).abs().view(B * n)
            order = torch.sort(diff, dim=0, descending=False)[1]
            outliers = order[int((1 - tolerance) * diff.shape[0]):]
            a.view(B * n)[outliers] = b.view(B * n)[outliers]
        loss = 1 - (a * b).sum(dim=1) / (a + b - a * b + 1.e-8).sum(dim=1)
        if weight is not None:
            loss = loss * weight.squeeze(1).squeeze(1)
        return loss.mean()

    def mask_loss(self, inputs, targets, weight=None, tolerance=0.):
        return self.iou_loss(inputs, targets, weight=weight, tolerance=tolerance) + \
            self.opt.reg.mask_mse * self.MSE_loss(inputs, targets, weight=weight, tolerance=tolerance)

    # ---- surface normals (loss.py:52-67) ------------------------------------------------------------
    def normal_loss(self, normal_pred, normal_gt, mask, weight=None, tolerance=0.):
        mask = mask.squeeze(-1)
        assert normal_pred.shape == normal_gt.shape and len(normal_pred.shape) == 3 and len(mask.shape) == 2
        p, t = normal_pred[mask], normal_gt[mask]
        angular = 1 - torch.sum(p * t, dim=-1)
        loss = self.opt.reg.normal_l1 * (p - t).abs().sum(dim=-1) + angular
        keep = torch.sort(angular, dim=0, descending=False)[1][:int(loss.shape[0] * (1 - tolerance))]
        if weight is not None:
            loss = loss * weight.expand_as(normal_pred)[mask][..., 0]
        return loss[keep].mean()

    # ---- camera priors (loss.py:99-167) -------------------------------------------------------------
    def cam_margin(self, opt, trig, ranges, eps=5):
        assert ranges[0] > -180 and ranges[1] < 180
        angle = torch.atan2(trig[:, 1], trig[:, 0]) * 180 / np.pi
        return self.L1_loss((-angle + ranges[0] - eps).relu_()) + self.L1_loss((angle - ranges[1] - eps).relu_())

    def cam_margin_loss(self, opt, var):
        r = opt.data[opt.data.dataset]
        return self.cam_margin(opt, var.trig_elev, r.elev_range) + self.cam_margin(opt, var.trig_theta, r.theta_range)

    def cam_sym_loss(self, opt, var, estimator):
        # Graph.forward may already have run the mirrored images through the estimator (batched with the other views)
        fa, fe, ft = var._estim_flip[:3] if "_estim_flip" in var else estimator(var.rgb_input_map.flip(dims=[3]))[:3]
        # mirrored image: azimuth and roll change sign (sin flips), elevation is unchanged
        def sq(trig, flipped, sign):
            return (trig[:, 0] - flipped[:, 0]) ** 2 + (sign * trig[:, 1] - flipped[:, 1]) ** 2
        return sq(var.trig_azim, fa, -1).mean() + sq(var.trig_elev, fe, 1).mean() + sq(var.trig_theta, ft, -1).mean()

    def camera_prior_losses(self, opt, var, estimator):
        """{"cam_margin" | "cam_uniform" | "cam_sym": thunk}.  On the device with all three enabled they are one launch
        (csrc/camera_prior.hip, ~90 [B]-sized torch operators forward + as many backward otherwise)."""
        lw = opt.loss_weight
        trig = var.trig_azim
        if (trig.is_cuda and lw.cam_margin is not None and lw.cam_uniform is not None and lw.cam_sym is not None
                and "_estim_flip" in var and opt.get("hip", {}).get("fused_loss", True)):
            from .. import ops
            if ops.camera_prior_supported(trig.shape[0], opt.reg.emd_p):
                from ..functional import CameraPriorLossFunction
                r = opt.data[opt.data.dataset]
                assert r.elev_range[0] > -180 and r.elev_range[1] < 180 and r.theta_range[0] > -180 and r.theta_range[1] < 180
                fa, fe, ft = var._estim_flip[:3]
                m, u, s = CameraPriorLossFunction.apply(trig, var.trig_elev, var.trig_theta, fa, fe, ft,
                                                        (float(r.elev_range[0]), float(r.elev_range[1])),
                                                        (float(r.theta_range[0]), float(r.theta_range[1])), 5.0, int(opt.reg.emd_p))
                return dict(cam_margin=lambda: m, cam_uniform=lambda: u, cam_sym=lambda: s)
        return dict(cam_margin=lambda: self.cam_margin_loss(opt, var), cam_uniform=lambda: self.cam_uniform_loss(opt, trig),
                    cam_sym=lambda: self.cam_sym_loss(opt, var, estimator))

    def cam_uniform_loss(self, opt, trig):
        B = trig.shape[0]
        grid = torch.arange(1., 2 * B, 2., device=trig.device).float() * np.pi / B
        empirical = (trig[:, 0], trig[:, 1], trig[:, 0] * trig[:, 1])
        prior = (torch.cos(grid), torch.sin(grid), torch.cos(grid) * torch.sin(grid))
        dists = [p.sort(dim=0, descending=False)[0] - e.sort(dim=0, descending=False)[0] for p, e in zip(prior, empirical)]
        if opt.reg.emd_p == 1:
            return sum(d.abs().mean() for d in dists) / 3
        return sum(torch.norm(d, dim=0, p=opt.reg.emd_p) for d in dists) / (3 * B)

    def category_reg_loss(self, opt, var, shape_center):
        code = torch_F.normalize(var.proj_latent_sdf, dim=-1)
        center = torch_F.normalize(shape_center, dim=-1)
        return self.CE_loss(code @ center.permute(1, 0).contiguous() / 0.3, var.category_label)
