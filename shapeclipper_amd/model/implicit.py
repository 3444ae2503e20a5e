"""Implicit networks of the MI355X build: same classes, constructor signatures and state-dict keys
as the reference's model/implicit.py (lin{l}.weight / lin{l}.bias in torch Linear layout, `beta`),
so reference checkpoints load unchanged -- but the arithmetic runs in the hand-written HIP kernels
(csrc/sdf_fwd.hip, sdf_bwd.hip, wgrad.hip) through torch.autograd.Function wrappers.

The modules only hold parameters and do the (tiny, differentiable) weight packing; there is no
per-point torch arithmetic here and no CPU fallback: calling them with CPU tensors raises.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import packing
from ..functional import SdfFunction
from . import eager_path


class Embedder:
    """NeRF positional encoding (reference model/implicit.py:7-38); host-side helper kept for API
    compatibility -- the kernels evaluate the encoding in registers."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs["input_dims"]
        n = kwargs["num_freqs"]
        top = kwargs["max_freq_log2"]
        self.freq_bands = (2.0 ** torch.linspace(0.0, top, n)) if kwargs["log_sampling"] \
            else torch.linspace(2.0 ** 0.0, 2.0 ** top, n)
        self.out_dim = (d if kwargs["include_input"] else 0) + d * n * len(kwargs["periodic_fns"])

    def embed(self, inputs):
        parts = [inputs] if self.kwargs["include_input"] else []
        for f in self.freq_bands:
            for fn in self.kwargs["periodic_fns"]:
                parts.append(fn(inputs * f))
        return torch.cat(parts, -1)


def get_embedder(posenc_res, input_dims=3):
    eo = Embedder(include_input=True, input_dims=input_dims, max_freq_log2=posenc_res - 1, num_freqs=posenc_res,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return (lambda x, eo=eo: eo.embed(x)), eo.out_dim


class Density(nn.Module):
    def __init__(self, params_init={}):
        super().__init__()
        for name, value in params_init.items():
            setattr(self, name, nn.Parameter(torch.tensor(value)))

    def forward(self, sdf, beta=None):
        return self.density_func(sdf, beta=beta)


class LaplaceDensity(Density):
    """sigma(s) = (1/beta) * Laplace-CDF(-s)  (reference model/implicit.py:65-83).  Inside the renderer the
    density, its s- and beta-derivatives are evaluated in rgb_fwd.hip / rgb_bwd.hip from the raw `beta`
    parameter; this torch form serves external callers on small tensors."""

    def __init__(self, params_init={}, beta_min=0.0001):
        super().__init__(params_init=params_init)
        self.beta_min = torch.tensor(beta_min)

    def density_func(self, sdf, beta=None):
        if beta is None:
            beta = self.get_beta()
        half = 0.5 * torch.exp(-sdf.abs() / beta)
        return torch.where(sdf >= 0, half, 1 - half) / beta

    def get_beta(self):
        return self.beta.abs() + self.beta_min.to(self.beta.device)


def _effective(lin, name):
    """Parameter `name` of a Linear as the kernels must see it: with arch.*.weight_norm (reference model/implicit.py:131-132,213-214:
    nn.utils.weight_norm) the weight is g * v / ||v|| per output row -- computed here from the weight_g / weight_v parameters the
    reference's state dict holds (the `weight` attribute nn.utils.weight_norm leaves on the module is only refreshed by the module's
    own forward, which this path never calls); differentiable w.r.t. both."""
    if name == "weight" and hasattr(lin, "weight_g"):
        return torch._weight_norm(lin.weight_v, lin.weight_g, 0)
    return getattr(lin, name)


class SDFNetwork(nn.Module):
    """Conditional SDF MLP (reference model/implicit.py:85-189)."""

    def __init__(self, opt):
        super().__init__()
        # outside the compiled family (packing.check_arch): stock device operators, model/eager_path.py (round 5; raised before)
        self.eager = not packing.arch_supported(opt)
        if self.eager:
            eager_path.warn_once(packing.arch_summary(opt))
        a = opt.arch.impl_sdf
        self.posenc_res = a.pos_enc
        self.force_symmetry = opt.arch.force_symmetry
        self.proj_latent_dim = a.proj_latent_dim
        self.n_hidden = a.n_hidden_layers
        self.n_channel = a.n_channels
        self.skip_in = list(a.skip_connection)
        pe = 6 * a.pos_enc
        d0 = 3 + pe + self.proj_latent_dim
        dims = [d0] + [self.n_channel] * self.n_hidden + [1 + self.n_channel]
        self.num_layers = len(dims)
        # Same construction order and init calls as the reference, so torch.manual_seed(s) reproduces
        # the reference's initial weights bit for bit (nn.Linear's own init draws first).
        for l in range(self.num_layers - 1):
            in_dim = dims[l] + (dims[0] if l in self.skip_in else 0)
            out_dim = dims[l + 1]
            lin = nn.Linear(in_dim, out_dim)
            if a.geometric_init:
                std = np.sqrt(2) / np.sqrt(out_dim)
                if l == self.num_layers - 2:
                    nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(in_dim), std=0.0001)
                    nn.init.constant_(lin.bias, -a.init_sphere_radius)
                elif a.pos_enc > 0 and l == 0:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.constant_(lin.weight[:, 3:], 0.0)
                    nn.init.normal_(lin.weight[:, :3], 0.0, std)
                elif a.pos_enc > 0 and l in self.skip_in:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, std)
                    nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, std)
            if a.weight_norm:          # reparameterise AFTER the geometric initialisation, as the reference does (:130-132)
                lin = nn.utils.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)

    def weight_dict(self):
        return {"lin%d.%s" % (l, n): _effective(getattr(self, "lin%d" % l), n)
                for l in range(self.num_layers - 1) for n in ("weight", "bias")}

    def packed(self, proj_latent):
        """(w_pack, cbias [B,5,64]) for the HIP kernels; differentiable w.r.t. parameters and latent.
        The weight image does not depend on the latent: inside one Graph.forward (between begin_step() calls) it is
        built once and shared by the main render, the NN-view render and the eikonal calls (4 uses per step)."""
        if self.eager:
            raise NotImplementedError("no packed weight image for this architecture (%d x %d): it runs on model/eager_path.py" % (self.n_hidden, self.n_channel))
        cache = getattr(self, "_pack_cache", None)
        # the per-image biases too are shared inside a forward pass: the main and the neighbour-view render condition the SDF on the SAME
        # latent tensor, their two eikonal calls on the same detached one (round 5: 4 -> 2 bias launches each way per step, one gradient each)
        zkey = (proj_latent.data_ptr(), proj_latent._version, tuple(proj_latent.shape), bool(proj_latent.requires_grad))
        if cache is not None and cache[0] == torch.is_grad_enabled():
            hit = cache[3].get(zkey)
            # the key is an address: a DIFFERENT autograd tensor on the same storage (a view / alias with its own grad_fn) must not share the
            # entry, or its gradient would flow into the first tensor's graph only
            if hit is not None and hit[1] is not proj_latent and proj_latent.requires_grad:
                hit = None
            if hit is None:
                hit = cache[3][zkey] = (packing.sdf_cbias(None, proj_latent, gathered=cache[2], arch=self._arch(proj_latent.shape[1])), proj_latent)
            return cache[1], hit[0]
        w_pack, cbias, gathered = packing.pack_sdf(self.weight_dict(), proj_latent, return_gathered=True)
        if getattr(self, "_pack_cache_on", False):
            self._pack_cache = (torch.is_grad_enabled(), w_pack, gathered, {zkey: (cbias, proj_latent)})     # (the latent is kept alive: its address is the key)
        return w_pack, cbias

    def _arch(self, Z):
        return (self.n_channel, self.lin0.in_features - Z, tuple(l for l in (1, 2) if l in self.skip_in))

    def begin_step(self, enable=True):
        """Start (or stop) sharing the packed weight image; call once per forward pass, before the first render."""
        self._pack_cache_on, self._pack_cache = enable, None

    def forward(self, points_raw, proj_latent):
        """Per-point latent form of the reference ([N,3], [N,Z] -> [N,1+C]); every point is its own 'image'."""
        if self.eager:
            eager_path.require_device(points_raw)
            return eager_path.sdf_mlp(self, points_raw.unsqueeze(1), proj_latent).squeeze(1)
        w_pack, cbias = self.packed(proj_latent)
        sdf, _, feat = SdfFunction.apply(points_raw, w_pack, cbias, 1, bool(self.force_symmetry), False, True)
        return torch.cat([sdf[:, None], packing.tbl_to_rows(feat, points_raw.shape[0])[:, :self.n_channel]], dim=1)

    def get_conditional_output(self, opt, batch_size, points_flat, proj_latent, compute_grad=True):
        """-> (sdf [N,1], impl_feat [N,C], gradients [N,3] | None), N = batch_size * points-per-image.
        With compute_grad the latent is detached (reference :168-169) and `gradients` stays differentiable."""
        n = points_flat.shape[0]
        assert n % batch_size == 0 and proj_latent.shape[1] == opt.arch.impl_sdf.proj_latent_dim
        if self.eager:
            return eager_path.sdf_conditional_output(self, batch_size, points_flat, proj_latent, compute_grad)
        if compute_grad:
            proj_latent = proj_latent.detach()
        w_pack, cbias = self.packed(proj_latent)
        sdf, grad, feat = SdfFunction.apply(points_flat, w_pack, cbias, n // batch_size, bool(self.force_symmetry),
                                            bool(compute_grad), True, bool(opt.get("hip", {}).get("fused_backward", True)))
        return sdf[:, None], packing.tbl_to_rows(feat, n)[:, :self.n_channel], (grad if compute_grad else None)


class RGBNetwork(nn.Module):
    """Colour MLP (reference model/implicit.py:191-239).  Inside the renderer it runs fused with the
    compositing (rgb_fwd.hip / rgb_bwd.hip); `forward` below is the interface-compatible standalone form
    (plain device torch ops, not on the hot path)."""

    def __init__(self, opt):
        super().__init__()
        self.eager = not packing.arch_supported(opt)
        a = opt.arch.impl_rgb
        self.force_symmetry = opt.arch.force_symmetry
        self.proj_latent_dim = a.proj_latent_dim
        self.n_hidden = a.n_hidden_layers
        self.n_sdf_channel = opt.arch.impl_sdf.n_channels
        self.n_channel = a.n_channels
        dims = [3 + 6 * a.pos_enc + self.proj_latent_dim + self.n_sdf_channel] + [self.n_channel] * self.n_hidden + [3]
        self.num_layers = len(dims)
        self.posenc_res = a.pos_enc
        for l in range(self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if a.weight_norm:
                lin = nn.utils.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)

    def weight_dict(self):
        return {"lin%d.%s" % (l, n): _effective(getattr(self, "lin%d" % l), n)
                for l in range(self.num_layers - 1) for n in ("weight", "bias")}

    def packed(self, proj_latent):
        cache = getattr(self, "_pack_cache", None)
        if cache is not None and cache[0] == torch.is_grad_enabled():
            return cache[1], packing.rgb_dbias(None, proj_latent, gathered=cache[2],
                                               arch=(self.n_channel, 3 + 6 * self.posenc_res, self.n_sdf_channel))
        v_pack, dbias, gathered = packing.pack_rgb(self.weight_dict(), proj_latent, return_gathered=True, n_sdf=self.n_sdf_channel)
        if getattr(self, "_pack_cache_on", False):
            self._pack_cache = (torch.is_grad_enabled(), v_pack, gathered)
        return v_pack, dbias

    def begin_step(self, enable=True):
        self._pack_cache_on, self._pack_cache = enable, None

    def forward(self, points_raw, proj_latent, sdf_feature):
        pts = points_raw
        if self.force_symmetry:
            pts = torch.cat([pts[..., :1].abs(), pts[..., 1:]], dim=-1)
        embed, _ = get_embedder(self.posenc_res)
        x = torch.cat([embed(pts), proj_latent, sdf_feature], dim=-1)
        for l in range(self.num_layers - 1):
            x = getattr(self, "lin" + str(l))(x)
            if l < self.num_layers - 2:
                x = torch.relu(x)
        return torch.sigmoid(x)
