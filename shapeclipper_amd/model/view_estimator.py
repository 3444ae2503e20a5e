"""Viewpoint / scale estimator -- same module tree and state-dict keys as the reference's
model/view_estimator.py (ResNet-18 trunk `feature_extractor`, three bottleneck heads, extr_fc /
size_fc / perspect_fc).  Stock PyTorch-ROCm ops (out of the hand-written hot path)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as torch_F

from ..functional import bn_act, bottleneck_linear
from . import resnet

HIP_BOTTLENECK = True      # `--hip.fused_bottleneck!`: the operator-by-operator form (rocBLAS products + separate BatchNorm launches)


class Bottleneck_Linear(nn.Module):
    """1x1-conv residual bottleneck on a feature vector (reference view_estimator.py:6-33)."""

    def __init__(self, n_channels, zero_init=True):
        super().__init__()
        self.linear1 = nn.Conv2d(n_channels, n_channels, kernel_size=1, padding=0, bias=False)
        self.bn1 = nn.BatchNorm2d(n_channels)
        self.linear2 = nn.Conv2d(n_channels, n_channels, kernel_size=1, padding=0, bias=False)
        self.bn2 = nn.BatchNorm2d(n_channels)
        self.relu = nn.ReLU(inplace=True)
        if zero_init:
            nn.init.constant_(self.bn2.weight, 0)

    @staticmethod
    def _linear(conv, x):
        """A 1x1 convolution on a 1x1 map IS a matrix product: one GEMM [N, C] x [C, C] instead of MIOpen's N strided-batched
        512 x 1 x 512 products and their layout transposes.  The weight is taken as a VIEW [C_out, C_in] of the conv
        parameter (same state_dict layout as the reference): indexing it with [:, :, 0, 0] costs two select nodes whose
        backward is a fill + a copy each (~20 launches per block and step)."""
        assert conv.bias is None and conv.kernel_size == (1, 1)
        return torch_F.linear(x, conv.weight.view(conv.out_channels, conv.in_channels))

    def forward(self, x, groups=1):
        if HIP_BOTTLENECK and x.is_cuda:
            out = bottleneck_linear(x, self.linear1, self.bn1, self.linear2, self.bn2, groups=groups)      # 2 launches forward, 3 backward
            if out is not None:
                return out
        v = x[..., None, None]
        out = bn_act(self.bn1, self._linear(self.linear1, x)[..., None, None], groups=groups)        # conv1x1 -> BN -> ReLU
        out = self._linear(self.linear2, out.flatten(1))[..., None, None]
        return bn_act(self.bn2, out, residual=v, groups=groups).flatten(1)


def _random_trunk_ok(opt):
    """Explicit opt-outs from the ImageNet initialisation: --arch.enc_pretrained!, or a checkpoint that overwrites it."""
    return (not opt.arch.get("enc_pretrained", True)) or bool(opt.get("load")) or bool(opt.get("resume"))


class Estimator(nn.Module):

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.dataset = opt.data.dataset
        # always ImageNet-initialised in the reference (view_estimator.py:40); see resnet.build for what happens when
        # the weights are unavailable
        self.feature_extractor = resnet.build("resnet18", pretrained=True, allow_random=_random_trunk_ok(opt))
        n_features = self.feature_extractor.fc.in_features
        self.feature_extractor.fc = nn.Identity()
        self.extr_head = nn.Sequential(Bottleneck_Linear(n_features))
        self.size_head = nn.Sequential(Bottleneck_Linear(n_features))
        self.perspect_head = nn.Sequential(Bottleneck_Linear(n_features))
        self.extr_fc = nn.Linear(n_features, 6)
        self.size_fc = nn.Linear(n_features, 1)
        self.perspect_fc = nn.Linear(n_features, 1)
        # elevation / roll start at zero: (cos, sin) = (1, 0) after normalisation
        with torch.no_grad():
            self.extr_fc.weight[2:, :].zero_()
            self.extr_fc.bias[2:] = torch.tensor([1.0, 0.0, 1.0, 0.0])
        self.reset_scales()

    def reset_scales(self):
        for fc in (self.size_fc, self.perspect_fc):
            nn.init.constant_(fc.weight, 0.0)
            nn.init.constant_(fc.bias, 0.0)

    def forward(self, inputs, groups=1, split=False):
        """groups > 1: `inputs` stacks several image sets that the reference sends through the estimator in separate
        calls (input view, CLIP neighbour, mirrored input); BatchNorm keeps them separate, see resnet.ResNet.forward.
        split: return one 5-tuple per image set instead of 5 stacked tensors."""
        feat = self.feature_extractor(inputs, groups=groups)
        trig = self.extr_fc(self.extr_head[0](feat, groups=groups))
        size_lin = self.size_fc(self.size_head[0](feat, groups=groups))
        persp_lin = self.perspect_fc(self.perspect_head[0](feat, groups=groups))
        if trig.is_cuda:        # normalisation / tanh / range scaling of all outputs: one launch each way (csrc/camera_prior.hip)
            from ..functional import EstimatorHeadFunction
            outs = EstimatorHeadFunction.apply(trig, size_lin, persp_lin, float(self.opt.camera.size_range),
                                               float(self.opt.camera.perspect_range), groups)
            sets = [tuple(outs[5 * g:5 * g + 5]) for g in range(groups)]
            if split:
                return sets
            return sets[0] if groups == 1 else tuple(torch.cat([st[k] for st in sets], 0) for k in range(5))
        azim, elev, theta = (torch_F.normalize(trig[:, 2 * k:2 * k + 2], dim=1, p=2) for k in range(3))
        size_raw = torch.tanh(size_lin).squeeze(-1)
        persp_raw = torch.tanh(persp_lin).squeeze(-1)
        scale_size = 1 + size_raw * self.opt.camera.size_range
        scale_perspect = 1 + persp_raw * self.opt.camera.perspect_range
        out = (azim, elev, theta, scale_perspect, scale_size * scale_perspect)
        if split:
            B = inputs.shape[0] // groups
            return [tuple(t[g * B:(g + 1) * B] for t in out) for g in range(groups)]
        return out
