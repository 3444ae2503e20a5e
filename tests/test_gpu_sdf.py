"""Parity of the fused SDF-MLP forward kernel (value, feature, d sdf/dx) with the reference
(golden G2, captured from model/implicit.py) and with the CPU oracle on ragged sizes.

Bar: fp32 tolerance 2e-5 absolute on sdf/feat (values O(1)), 2e-4 relative on the gradient."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5


def hip_sdf(W, z, pts, n_per_image, want_grad=True, symmetric=True):
    from shapeclipper_amd import packing
    from shapeclipper_amd.ops import sdf_forward
    dev = torch.device("cuda:0")
    Wd = {k: v.to(dev) for k, v in W.items()}
    pack, cb = packing.pack_sdf(Wd, z.to(dev))
    out = sdf_forward(pts.to(dev).contiguous(), pack, cb, n_per_image, symmetric=symmetric,
                      want_grad=want_grad, want_feat=True)
    torch.cuda.synchronize()
    return out


def test_golden_networks(golden):
    from shapeclipper_amd import packing
    g = golden("g2_networks")
    W = {k[len("pert.sdf."):]: torch.tensor(g[k]) for k in g.files if k.startswith("pert.sdf.")}
    sdf, grad, feat = hip_sdf(W, torch.tensor(g["z_sdf"]), torch.tensor(g["pts"]), 128)
    n = g["pts"].shape[0]
    feat_rows = packing.tbl_to_rows(feat, n).cpu().numpy()
    assert np.abs(sdf.cpu().numpy() - g["sdf"][:, 0]).max() < TOL
    assert np.abs(feat_rows - g["feat"]).max() < TOL
    err = np.abs(grad.cpu().numpy() - g["grad"]).max()
    assert err < 2e-4 * max(1.0, np.abs(g["grad"]).max()), err
    # x0 == 0: torch.abs backward gives sign(0) = 0 (implicit.py:142-143)
    assert grad[0, 0].item() == 0.0


@pytest.mark.parametrize("B,N", [(1, 1), (1, 15), (2, 17), (3, 100), (2, 1000)])
def test_ragged_vs_oracle(B, N):
    from oracle import reference_ops as R
    from shapeclipper_amd import packing
    cfg = R.Cfg()
    torch.manual_seed(B * 100 + N)
    W = R.init_sdf_weights(cfg)
    W = {k: v + 0.05 * torch.randn_like(v) for k, v in W.items()}
    z = torch.randn(B, 64)
    pts = torch.rand(B * N, 3) * 2 - 1
    o_sdf, o_feat, o_grad = R.sdf_conditional(cfg, W, B, pts.clone(), z, compute_grad=True)
    sdf, grad, feat = hip_sdf(W, z, pts, N)
    assert (sdf.cpu() - o_sdf[:, 0].detach()).abs().max() < TOL
    assert (packing.tbl_to_rows(feat, B * N).cpu() - o_feat.detach()).abs().max() < TOL
    assert (grad.cpu() - o_grad.detach()).abs().max() < 2e-4 * max(1.0, o_grad.abs().max().item())


def test_value_only_kernel_matches_grad_kernel():
    from oracle import reference_ops as R
    cfg = R.Cfg()
    torch.manual_seed(3)
    W = R.init_sdf_weights(cfg)
    z = torch.randn(2, 64)
    pts = torch.rand(2 * 333, 3) * 2 - 1
    from shapeclipper_amd import ops
    s2, g2, f2 = hip_sdf(W, z, pts, 333, want_grad=False)            # value + feature: sdf_fwd.hip (fp32 MFMA)
    try:                                                              # the same kernel file with the gradient sweep: same bits
        ops.SDF_FWD_STREAM = False
        s1, _, f1 = hip_sdf(W, z, pts, 333, want_grad=True)
    finally:
        ops.SDF_FWD_STREAM = True
    assert g2 is None and torch.equal(s1, s2) and torch.equal(f1, f2)
    # round 6 default for calls with a gradient: streamed pre-split fragments (sdf_fwd_stream.hip) -- another arithmetic, fp32-accurate
    s3, _, f3 = hip_sdf(W, z, pts, 333, want_grad=True)
    assert (s3 - s2).abs().max() < 2e-6 * max(1.0, float(s2.abs().max())) and (f3 - f2).abs().max() < 2e-6 * max(1.0, float(f2.abs().max()))


def test_mirror_symmetry_known_answer():
    from oracle import reference_ops as R
    cfg = R.Cfg()
    W = R.init_sdf_weights(cfg, 0)
    z = torch.zeros(1, 64)
    pts = torch.tensor([[0.3, 0.1, -0.2], [-0.3, 0.1, -0.2], [0.0, 0.0, 0.0], [0.5, 0.0, 0.0]])
    sdf, grad, _ = hip_sdf(W, z, pts, 4)
    assert sdf[0].item() == sdf[1].item()
    assert grad[0, 0].item() == -grad[1, 0].item()
    # geometric-init known answers measured on the reference
    assert abs(sdf[2].item() - (-0.369)) < 2e-3 and abs(sdf[3].item() - 0.0067) < 2e-3


# ---- round 6: the value-only chain in the exact bf16x3 split arithmetic with pre-split weights (csrc/sdf_value_split.hip) ----
@pytest.mark.parametrize("B,N,symmetric", [(1, 1, True), (1, 15, True), (1, 16, False), (2, 17, True), (3, 100, False), (2, 1000, True), (5, 4099, True)])
def test_value_split_chain_vs_oracle_and_fp32_chain(B, N, symmetric):
    """What compute_level_grid runs by default: against the float64-free oracle at the bar of the fp32 kernel (2e-5), against the fp32-MFMA
    kernel at 2e-6 of the value range (both are fp32-accurate: they differ by summation order only), on ragged sizes (a last group with one tile, a last tile
    with one point), several images (per-image biases), mirror symmetry on and off; run twice: bit-reproducible."""
    from oracle import reference_ops as R
    from shapeclipper_amd import ops, packing
    cfg = R.Cfg()
    cfg.force_symmetry = symmetric
    torch.manual_seed(B * 1000 + N)
    W = R.init_sdf_weights(cfg)
    W = {k: v + 0.05 * torch.randn_like(v) for k, v in W.items()}
    z = torch.randn(B, 64)
    pts = torch.rand(B * N, 3) * 2 - 1
    pts[0, 0] = 0.0
    want = R.sdf_conditional(cfg, W, B, pts.clone(), z, compute_grad=False)[0][:, 0].detach()
    dev = torch.device("cuda:0")
    pack, cb = packing.pack_sdf({k: v.to(dev) for k, v in W.items()}, z.to(dev))
    p = pts.to(dev).contiguous()
    assert ops.SDF_VALUE_SPLIT
    split1 = ops.sdf_forward(p, pack, cb, N, symmetric=symmetric, want_grad=False, want_feat=False)[0]
    split2 = ops.sdf_forward(p, pack, cb, N, symmetric=symmetric, want_grad=False, want_feat=False)[0]
    try:
        ops.SDF_VALUE_SPLIT = False
        fp32 = ops.sdf_forward(p, pack, cb, N, symmetric=symmetric, want_grad=False, want_feat=False)[0]
    finally:
        ops.SDF_VALUE_SPLIT = True
    torch.cuda.synchronize()
    assert torch.equal(split1, split2)
    assert (split1.cpu() - want).abs().max() < TOL
    assert (split1 - fp32).abs().max() < 2e-6 * max(1.0, float(fp32.abs().max()))      # a few ulps of the value: summation order only
    if symmetric:       # sdf(x, y, z) == sdf(-x, y, z), bit for bit (|x| enters the encoding)
        q = p.clone()
        q[:, 0] = -q[:, 0]
        mirrored = ops.sdf_forward(q, pack, cb, N, symmetric=True, want_grad=False, want_feat=False)[0]
        assert torch.equal(mirrored, split1)


def test_value_split_chain_float64_error_is_the_fp32_chains():
    """The split arithmetic is exact up to fp32 accumulation: on 100,000 points its error against a float64 evaluation of the network is not
    larger than 1.5x the fp32-MFMA chain's (measured: 3.3e-7 against 3.1e-7 on the evaluation grid, profiles/r06_value_chain_split_ab.txt)."""
    from oracle import reference_ops as R
    from shapeclipper_amd import ops, packing
    cfg = R.Cfg()
    torch.manual_seed(11)
    W = R.init_sdf_weights(cfg)
    W = {k: v + 0.05 * torch.randn_like(v) for k, v in W.items()}
    z = torch.randn(2, 64) * 0.5
    n = 50000
    pts = torch.rand(2 * n, 3) * 1.2 - 0.6
    W64 = {k: v.double() for k, v in W.items()}
    want = R.sdf_mlp(cfg, W64, pts.double(), z.double().repeat_interleave(n, 0))[:, 0]
    dev = torch.device("cuda:0")
    pack, cb = packing.pack_sdf({k: v.to(dev) for k, v in W.items()}, z.to(dev))
    p = pts.to(dev).contiguous()
    split = ops.sdf_forward(p, pack, cb, n, want_grad=False, want_feat=False)[0]
    try:
        ops.SDF_VALUE_SPLIT = False
        fp32 = ops.sdf_forward(p, pack, cb, n, want_grad=False, want_feat=False)[0]
    finally:
        ops.SDF_VALUE_SPLIT = True
    e_split = float((split.double().cpu() - want).abs().max())
    e_fp32 = float((fp32.double().cpu() - want).abs().max())
    print("value chain vs float64 on 100,000 points: split %.3e, fp32 MFMA %.3e" % (e_split, e_fp32))
    assert e_split < 2e-6 and e_split < 1.5 * e_fp32 + 1e-7


# ---- round 6: value + feature + d sdf/dx from STREAMED pre-split fragments (csrc/sdf_fwd_stream.hip), the default of calls with a gradient ----
@pytest.mark.parametrize("B,N,symmetric,stash", [(1, 1, True, False), (1, 15, True, False), (2, 17, False, False), (3, 100, True, False),
                                                 (2, 1000, True, False), (2, 1000, False, True), (3, 1371, True, True), (1, 40000, True, True)])
def test_streamed_forward_vs_oracle_and_fp32_kernel(B, N, symmetric, stash):
    """Against the oracle at the bars of the fp32 kernel (2e-5 on sdf / feature, 2e-4 relative on the gradient) and against sdf_fwd.hip at a
    few ulps of each tensor's range (4e-6; 2e-5 for the adjoint quantities) -- sdf, d sdf/dx, feature and, in the training form, the parked activations and adjoints.  Sizes: fewer
    tiles than waves (most waves idle through every phase barrier), a ragged last tile, more rounds than one (40,000 points = 2,500 tiles
    on 2,048 waves: the last round is mostly idle waves), several images, symmetry on and off.  Twice: same bits."""
    from oracle import reference_ops as R
    from shapeclipper_amd import ops, packing
    cfg = R.Cfg()
    cfg.force_symmetry = symmetric
    torch.manual_seed(B * 1000 + N)
    W = R.init_sdf_weights(cfg)
    W = {k: v + 0.05 * torch.randn_like(v) for k, v in W.items()}
    z = torch.randn(B, 64)
    pts = torch.rand(B * N, 3) * 2 - 1
    pts[0, 0] = 0.0
    dev = torch.device("cuda:0")
    pack, cb = packing.pack_sdf({k: v.to(dev) for k, v in W.items()}, z.to(dev))
    p = pts.to(dev).contiguous()
    assert ops.SDF_FWD_STREAM
    a = ops.sdf_forward(p, pack, cb, N, symmetric=symmetric, want_grad=True, want_feat=True, stash=stash)
    a2 = ops.sdf_forward(p, pack, cb, N, symmetric=symmetric, want_grad=True, want_feat=True, stash=stash)
    try:
        ops.SDF_FWD_STREAM = False
        b = ops.sdf_forward(p, pack, cb, N, symmetric=symmetric, want_grad=True, want_feat=True, stash=stash)
    finally:
        ops.SDF_FWD_STREAM = True
    torch.cuda.synchronize()
    names = ["sdf", "grad", "feat", "stash_a", "stash_p"][:len(a)]
    for n, x, x2, y in zip(names, a, a2, b):
        assert torch.equal(x, x2), n
        scale = max(1.0, float(y.abs().max()))
        bar = 2e-5 if n in ("grad", "stash_p") else 4e-6          # the adjoint sweep chains five layers of rounding differences
        assert float((x - y).abs().max()) < bar * scale, (n, float((x - y).abs().max()), scale)
    if B * N <= 4000:
        o_sdf, o_feat, o_grad = R.sdf_conditional(cfg, W, B, pts.clone(), z, compute_grad=True)
        assert (a[0].cpu() - o_sdf[:, 0].detach()).abs().max() < TOL
        assert (packing.tbl_to_rows(a[2], B * N).cpu() - o_feat.detach()).abs().max() < TOL
        assert (a[1].cpu() - o_grad.detach()).abs().max() < 2e-4 * max(1.0, o_grad.abs().max().item())
    assert a[1][0, 0].item() == 0.0 or not symmetric      # x0 == 0: sign(0) = 0 (implicit.py:142-143)
