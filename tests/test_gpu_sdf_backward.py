"""Parity of the hand-written SDF backward (sdf_bwd.hip + wgrad.hip, incl. the second-order terms
of d/dtheta [d sdf/dx]) with CPU autograd over the oracle (double backward).

Functional: L = <sdf, c1> + <d sdf/dx, c2> + <feat, c3>.  Bar: 1e-3 relative to the largest entry of
each gradient tensor (fp32 accumulations over 100s of points in different orders)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(B, N, seed):
    from oracle import reference_ops as R
    cfg = R.Cfg()
    torch.manual_seed(seed)
    W = R.init_sdf_weights(cfg)
    W = {k: (v + 0.05 * torch.randn_like(v)) for k, v in W.items()}
    z = torch.randn(B, 64)
    pts = torch.rand(B * N, 3) * 2 - 1
    pts[1, 0] = 0.0
    c1, c2, c3 = torch.randn(B * N), torch.randn(B * N, 3), torch.randn(B * N, 64) * 0.1
    return cfg, W, z, pts, c1, c2, c3


def _oracle(cfg, W, z, pts, c1, c2, c3, B, use_grad, use_feat, detach_latent=False):
    from oracle import reference_ops as R
    Wl = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    zl = z.clone().requires_grad_(True)
    pl = pts.clone().requires_grad_(True)
    sdf, feat, grad = R.sdf_conditional(cfg, Wl, B, pl, zl, compute_grad=True)
    # sdf_conditional detaches the latent when compute_grad (implicit.py:168) -- re-run attached for z grads
    if not detach_latent:
        N = pts.shape[0] // B
        lat = zl.unsqueeze(1).repeat(1, N, 1).view(B * N, -1)
        out = R.sdf_mlp(cfg, Wl, pl, lat)
        sdf, feat = out[:, :1], out[:, 1:]
        grad = torch.autograd.grad(sdf, pl, torch.ones_like(sdf), create_graph=True)[0]
    L = (sdf[:, 0] * c1).sum()
    if use_grad:
        L = L + (grad * c2).sum()
    if use_feat:
        L = L + (feat * c3).sum()
    names = list(Wl.keys())
    gs = torch.autograd.grad(L, [Wl[k] for k in names] + [zl, pl], allow_unused=True)
    out = {k: (g if g is not None else torch.zeros_like(Wl[k])) for k, g in zip(names, gs[:len(names)])}
    out["z"] = gs[-2] if gs[-2] is not None else torch.zeros_like(z)
    out["points"] = gs[-1]
    return out


def _hip(W, z, pts, c1, c2, c3, N, use_grad, use_feat, fused=True):
    from shapeclipper_amd import packing
    from shapeclipper_amd.functional import SdfFunction
    dev = torch.device("cuda:0")
    Wd = {k: v.to(dev).requires_grad_(True) for k, v in W.items()}
    zd = z.to(dev).requires_grad_(True)
    pd = pts.to(dev).requires_grad_(True)
    pack, cb = packing.pack_sdf(Wd, zd)
    sdf, grad, feat = SdfFunction.apply(pd, pack, cb, N, True, use_grad, use_feat, fused)
    L = (sdf * c1.to(dev)).sum()
    if use_grad:
        L = L + (grad * c2.to(dev)).sum()
    if use_feat:
        L = L + (packing.tbl_to_rows(feat, pts.shape[0]) * c3.to(dev)).sum()
    names = list(Wd.keys())
    gs = torch.autograd.grad(L, [Wd[k] for k in names] + [zd, pd], allow_unused=True)
    torch.cuda.synchronize()
    out = {k: (g.cpu() if g is not None else torch.zeros_like(W[k])) for k, g in zip(names, gs[:len(names)])}
    out["z"] = gs[-2].cpu()
    out["points"] = gs[-1].cpu()
    return out


@pytest.mark.parametrize("B,N,use_grad,use_feat", [(2, 100, True, True), (1, 37, True, False), (2, 64, False, True),
                                                   (3, 17, False, False), (2, 500, True, True)])
def test_sdf_backward_vs_oracle(B, N, use_grad, use_feat):
    cfg, W, z, pts, c1, c2, c3 = _setup(B, N, 11 * B + N)
    ref = _oracle(cfg, W, z, pts, c1, c2, c3, B, use_grad, use_feat)
    got = _hip(W, z, pts, c1, c2, c3, N, use_grad, use_feat)
    for k in ref:
        scale = max(ref[k].abs().max().item(), 1e-3)
        err = (got[k] - ref[k]).abs().max().item()
        assert err <= 1e-3 * scale, (k, err, scale)


@pytest.mark.parametrize("B,N,use_feat", [(1, 16, True), (1, 48, True), (2, 96, True), (2, 512, False), (3, 1040, True),
                                          (2, 16384, True)])
def test_fused_backward_vs_oracle_and_vs_the_unfused_kernels(B, N, use_feat):
    """csrc/sdf_bwdw.hip (chain waves + weight-gradient waves exchanging operands through LDS; taken when d sdf/dx is
    differentiated and n_per_image % 16 == 0): fewer tiles than chain waves, tails, several sweeps of the persistent
    loop, image boundaries inside a workgroup's range.  Same bar as the unfused path, and the two paths agree."""
    cfg, W, z, pts, c1, c2, c3 = _setup(B, N, 7 * B + N)
    ref = _oracle(cfg, W, z, pts, c1, c2, c3, B, True, use_feat)
    got = _hip(W, z, pts, c1, c2, c3, N, True, use_feat, fused=True)
    old = _hip(W, z, pts, c1, c2, c3, N, True, use_feat, fused=False)
    worst = {}
    for k in ref:
        scale = max(ref[k].abs().max().item(), 1e-3)
        worst[k] = ((got[k] - ref[k]).abs().max().item() / scale, (got[k] - old[k]).abs().max().item() / scale)
    print("fused SDF backward B=%d N=%d: max err vs oracle / vs unfused (relative to max |ref|):" % (B, N),
          {k: "%.1e / %.1e" % v for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if v[0] > 1e-3 or v[1] > 1e-3}
    assert not bad, bad
