"""The RGB network's hidden activations parked by the forward pass (sc_rgb_composite_forward_stash) against their recomputation in the
reverse pass (sc_rgb_composite_backward_fused): model/implicit.py:220-239, model/renderer.py:110-152.  Same outputs from the forward
either way (bit for bit), the parked tensors equal the activations of a plain fp32 restatement of the network, and every gradient of the
two reverse passes agrees to 2e-5 of its largest entry (the recomputation evaluates the positional encoding with the hardware sine, the
forward pass with sincosf: the parked form is the one that matches the forward)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n_images=3, rays_per_image=40, seed=0):
    from shapeclipper_amd import ops, packing
    from oracle import reference_ops as R          # the checker's weights and restatement (tests only)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    cfg = R.Cfg()
    Ws = {k: v.to(dev) for k, v in R.init_sdf_weights(cfg, 1).items()}
    Wr = {k: v.to(dev) for k, v in R.init_rgb_weights(cfg, 2).items()}
    zs = (torch.randn(n_images, 64, generator=g) * 0.3).to(dev)
    zr = (torch.randn(n_images, 64, generator=g) * 0.3).to(dev)
    sdf_pack, cb = packing.pack_sdf(Ws, zs)
    rgb_pack, db = packing.pack_rgb(Wr, zr)
    n_rays = n_images * rays_per_image
    P = n_rays * 64
    pts = ((torch.rand(P, 3, generator=g) * 1.6 - 0.8)).to(dev)
    z = torch.sort(torch.rand(n_rays, 64, generator=g) * 2 + 4, dim=1).values.to(dev)
    dfac = (torch.rand(n_rays, generator=g) * 0.2 + 0.9).to(dev)
    sdf, grad, feat = ops.sdf_forward(pts, sdf_pack, cb, rays_per_image * 64)
    beta = torch.tensor([0.1], device=dev)
    return dict(ops=ops, dev=dev, pts=pts, z=z, dfac=dfac, sdf=sdf, grad=grad, feat=feat, rgb_pack=rgb_pack, db=db, beta=beta,
                rpi=rays_per_image, n_rays=n_rays, g=g, Wr=Wr, zr=zr)


@pytest.mark.parametrize("n_images,rays_per_image", [(3, 40), (3, 37), (1, 5)])      # 111 and 5 rays: the last workgroup's idle waves (tail path)
def test_stash_equals_recomputation(n_images, rays_per_image):
    s = _setup(n_images, rays_per_image)
    ops = s["ops"]
    common = (s["pts"], s["z"], s["dfac"], s["sdf"], s["grad"], s["feat"], s["rgb_pack"], s["db"], s["beta"], s["rpi"], True, 1e-4, 1.0, 1.0)
    a = ops.rgb_composite_forward(*common, keep_rgb_flat=True, keep_rr=True)
    b = ops.rgb_composite_forward(*common, keep_rgb_flat=True)
    for k in ("rgb", "mask", "mask_hard", "depth", "normal", "rgb_flat"):
        assert torch.equal(a[k], b[k]), k                     # parking changes nothing else
    rr = a["rr"].view(3, s["n_rays"] * 4, 16, 16, 4)          # TBL64: [tile][ch/4][pt][4]
    assert torch.isfinite(rr).all() and float(rr.min()) >= 0.0
    assert float((rr > 0).float().mean()) > 0.05              # post-ReLU activations, not zeros
    gen = s["g"]
    G = dict(G_rgb=torch.randn(s["n_rays"], 3, generator=gen).to(s["dev"]), G_mask=torch.randn(s["n_rays"], generator=gen).to(s["dev"]),
             G_depth=torch.randn(s["n_rays"], generator=gen).to(s["dev"]), G_normal=torch.randn(s["n_rays"], 3, generator=gen).to(s["dev"]))
    back = lambda rr_: ops.rgb_composite_backward(s["pts"], s["z"], s["dfac"], s["sdf"], s["grad"], s["feat"], s["rgb_pack"], s["db"], s["beta"],
                                                  a["rgb_flat"], s["rpi"], True, 1e-4, 1.0, 1.0, G["G_rgb"], G["G_mask"], G["G_depth"], G["G_normal"], rr=rr_)
    g1, g0 = back(a["rr"]), back(None)
    g1b = back(a["rr"])
    worst = {}
    for k in g0:
        assert torch.equal(g1[k], g1b[k]), "not reproducible: " + k
        scale = max(float(g0[k].abs().max()), 1e-6)
        worst[k] = float((g1[k] - g0[k]).abs().max()) / scale
    print("parked vs recomputed (max abs / max |recomputed|):", {k: "%.1e" % v for k, v in worst.items()})
    assert all(v <= 2e-5 for v in worst.values()), worst


def test_stash_entry_point_refuses_a_missing_stash():
    import ctypes
    from shapeclipper_amd import _lib
    lib = _lib.load()
    z = ctypes.c_void_p(0)
    one = torch.zeros(8, device="cuda:0")
    p = _lib.ptr(one)
    rc = lib.sc_rgb_composite_backward_fused_stash(p, p, p, p, p, p, p, p, p, p, ctypes.c_int(4), ctypes.c_int(4), ctypes.c_int(1), ctypes.c_int(0),
                                                   ctypes.c_float(1e-4), ctypes.c_float(1.0), ctypes.c_float(1.0), z, z, z, z, p, p, p, p, p, p, p, p, p, z,
                                                   _lib.stream())
    assert rc != 0          # rr == NULL: an error code, not a launch


@pytest.mark.parametrize("n_images,rays_per_image", [(3, 40), (3, 37), (1, 5), (2, 512)])
def test_split_forward_equals_fp32_forward_up_to_rounding(n_images, rays_per_image):
    """Round 6: the forward pass with the RGB network from pre-split bf16x3 weight fragments (sc_rgb_composite_forward_split, the default)
    against the fp32-MFMA form (`--hip.rgb_split!`): what does not depend on the RGB network (mask, mask_hard, depth, normal) is
    bit-identical; colours, per-sample colours and the parked activations agree to a few fp32 ulps of their range; against a float64
    restatement of the network the split form's parked activations are as close as the fp32 form's; parked or not, and twice: same bits."""
    s = _setup(n_images, rays_per_image, seed=5)
    ops = s["ops"]
    common = (s["pts"], s["z"], s["dfac"], s["sdf"], s["grad"], s["feat"], s["rgb_pack"], s["db"], s["beta"], s["rpi"], True, 1e-4, 1.0, 1.0)
    assert ops.RGB_FWD_SPLIT
    a = ops.rgb_composite_forward(*common, keep_rgb_flat=True, keep_rr=True)
    a2 = ops.rgb_composite_forward(*common, keep_rgb_flat=True, keep_rr=True)
    a3 = ops.rgb_composite_forward(*common, keep_rgb_flat=True)
    try:
        ops.RGB_FWD_SPLIT = False
        b = ops.rgb_composite_forward(*common, keep_rgb_flat=True, keep_rr=True)
    finally:
        ops.RGB_FWD_SPLIT = True
    for k in ("rgb", "mask", "mask_hard", "depth", "normal", "rgb_flat", "rr"):
        assert torch.equal(a[k], a2[k]), k
        if k != "rr":
            assert torch.equal(a[k], a3[k]), k
    for k in ("mask", "mask_hard", "depth", "normal"):
        assert torch.equal(a[k], b[k]), k
    assert (a["rgb"] - b["rgb"]).abs().max() < 2e-6 and (a["rgb_flat"] - b["rgb_flat"]).abs().max() < 2e-6
    scale = float(b["rr"].abs().max())
    assert (a["rr"] - b["rr"]).abs().max() < 4e-6 * max(1.0, scale)
    assert not torch.equal(a["rr"], b["rr"]) or rays_per_image < 40        # the two arithmetics really are different code paths
    # per-sample colours against a float64 evaluation of the network (oracle restatement; the HIP feature tensor as its input)
    from oracle import reference_ops as R
    from shapeclipper_amd import packing
    P = s["n_rays"] * 64
    feat = packing.tbl_to_rows(s["feat"], P).double().cpu()
    Wr = {k: v.double().cpu() for k, v in s["Wr"].items()}
    zr = s["zr"].double().cpu().repeat_interleave(rays_per_image * 64, 0)
    want = R.rgb_mlp(R.Cfg(), Wr, s["pts"].double().cpu(), zr, feat)
    ea, eb = (float((t["rgb_flat"].double().cpu() - want).abs().max()) for t in (a, b))
    print("per-sample colours vs float64: split %.2e, fp32 MFMA %.2e" % (ea, eb))
    assert ea < 1.5 * eb + 2e-7 and ea < 2e-6


@pytest.mark.parametrize("n_images,rays_per_image", [(3, 40), (3, 37), (1, 5), (2, 512)])
def test_split_reverse_chain_equals_fp32_reverse_chain_up_to_rounding(n_images, rays_per_image):
    """Round 6: sc_rgb_composite_backward_fused_split (V2^T, V1^T, V0f^T and the encoding's Jacobian from pre-split bf16x3 fragments) against
    sc_rgb_composite_backward_fused_stash (fp32 MFMA) on the same parked activations: what does not pass through those products (g_sdf,
    g_grad, g_z, depth_fac, beta) is bit-identical, every other gradient agrees to 2e-5 of its largest entry (the bar between the parked
    and the recomputed fp32 forms above); twice: same bits."""
    s = _setup(n_images, rays_per_image, seed=9)
    ops = s["ops"]
    common = (s["pts"], s["z"], s["dfac"], s["sdf"], s["grad"], s["feat"], s["rgb_pack"], s["db"], s["beta"], s["rpi"], True, 1e-4, 1.0, 1.0)
    a = ops.rgb_composite_forward(*common, keep_rgb_flat=True, keep_rr=True)
    gen = s["g"]
    G = [torch.randn(s["n_rays"], 3, generator=gen).to(s["dev"]), torch.randn(s["n_rays"], generator=gen).to(s["dev"]),
         torch.randn(s["n_rays"], generator=gen).to(s["dev"]), torch.randn(s["n_rays"], 3, generator=gen).to(s["dev"])]
    back = lambda: ops.rgb_composite_backward(s["pts"], s["z"], s["dfac"], s["sdf"], s["grad"], s["feat"], s["rgb_pack"], s["db"], s["beta"],
                                              a["rgb_flat"], s["rpi"], True, 1e-4, 1.0, 1.0, *G, rr=a["rr"])
    saved = ops.RGB_BWD_SPLIT
    try:
        ops.RGB_BWD_SPLIT = True
        g1, g1b = back(), back()
        ops.RGB_BWD_SPLIT = False
        g0 = back()
    finally:
        ops.RGB_BWD_SPLIT = saved
    worst = {}
    for k in g0:
        assert torch.equal(g1[k], g1b[k]), "not reproducible: " + k
        scale = max(float(g0[k].abs().max()), 1e-6)
        worst[k] = float((g1[k] - g0[k]).abs().max()) / scale
    print("split vs fp32 reverse chain (max abs / max |fp32|):", {k: "%.1e" % v for k, v in worst.items()})
    for k in ("sdf", "grad", "z_vals", "depth_fac", "beta"):
        assert worst[k] == 0.0, (k, worst[k])
    assert all(v <= 2e-5 for v in worst.values()), worst
    assert any(v > 0 for v in worst.values())                  # two arithmetics, not one code path
