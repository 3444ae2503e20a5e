"""Other members of the reference's architecture family (model/implicit.py:89-113,197-214: n_channels, pos_enc, skip_connection,
proj_latent_dim) on the HIP kernels -- VERDICT r03 missing #2.  The kernels are compiled for 64 channels, 6 octaves and skip inputs at
layers 1 and 2; packing.py embeds every SMALLER architecture exactly (zero rows / columns / octaves), nothing else changes.

  * goldens G15a / G15b, captured from the REFERENCE's own SDFNetwork / RGBNetwork built with the variant options
    (tests/golden/make_golden.py): the reference's state dict loads strictly, sdf / feature / d sdf/dx from the HIP kernels match, and so
    do the gradients of a fixed functional w.r.t. every parameter of both networks;
  * a training render (HIP sampling + SDF + RGB + compositing + fused backward) of each variant against the oracle's render of the same
    weights (the oracle is pinned to the reference for these architectures by G15): outputs and gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VARIANTS = {"a": dict(cs=48, ls=4, skip=[2], zs=32, cr=32, lr=5, zr=48), "b": dict(cs=32, ls=2, skip=[1], zs=64, cr=64, lr=0, zr=16)}


def _opt(v):
    from shapeclipper_amd.utils import options
    extra = ["--arch.impl_sdf.n_channels=%d" % v["cs"], "--arch.impl_sdf.pos_enc=%d" % v["ls"], "--arch.impl_sdf.skip_connection=%s" % v["skip"],
             "--arch.impl_sdf.proj_latent_dim=%d" % v["zs"], "--arch.impl_rgb.n_channels=%d" % v["cr"], "--arch.impl_rgb.pos_enc=%d" % v["lr"],
             "--arch.impl_rgb.proj_latent_dim=%d" % v["zr"]]
    return options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_arch", "--output_root=/tmp/sc_pytest"] + extra),
                       verbose=False)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_variant_networks_match_the_reference_capture(golden, tag):
    from shapeclipper_amd.model.implicit import RGBNetwork, SDFNetwork
    g = golden("g15%s_arch_variant" % tag)
    v = VARIANTS[tag]
    assert list(g["arch"]) == [v["cs"], v["ls"], v["zs"], v["cr"], v["lr"], v["zr"], int(1 in v["skip"]), int(2 in v["skip"])]
    dev = torch.device("cuda:0")
    opt = _opt(v)
    assert opt.arch.impl_sdf.n_channels == v["cs"] and list(opt.arch.impl_sdf.skip_connection) == v["skip"]
    sdf, rgb = SDFNetwork(opt), RGBNetwork(opt)
    sdf.load_state_dict({k[len("w.sdf."):]: torch.tensor(g[k]) for k in g.files if k.startswith("w.sdf.")}, strict=True)
    rgb.load_state_dict({k[len("w.rgb."):]: torch.tensor(g[k]) for k in g.files if k.startswith("w.rgb.")}, strict=True)
    sdf, rgb = sdf.to(dev), rgb.to(dev)
    pts, zs, zr = (torch.tensor(g[k]).to(dev) for k in ("pts", "z_sdf", "z_rgb"))
    B, N = zs.shape[0], pts.shape[0] // zs.shape[0]
    s, f, gr = sdf.get_conditional_output(opt, B, pts.clone(), zs, compute_grad=True)
    assert f.shape == (B * N, v["cs"])
    lat = zr.unsqueeze(1).repeat(1, N, 1).view(B * N, -1)
    c = rgb(pts, lat, f)
    ref = {k: torch.tensor(g[k]).to(dev) for k in ("sdf", "feat", "grad", "rgb")}
    assert (s - ref["sdf"]).abs().max() < 2e-5 and (f - ref["feat"]).abs().max() < 2e-5
    assert (gr - ref["grad"]).abs().max() < 2e-4 * max(1.0, float(ref["grad"].abs().max()))
    assert (c - ref["rgb"]).abs().max() < 2e-5
    cot = {k: torch.tensor(g["cot." + k]).to(dev) for k in ("sdf", "feat", "grad", "rgb")}
    ((s * cot["sdf"]).sum() + (f * cot["feat"]).sum() + (gr * cot["grad"]).sum() + (c * cot["rgb"]).sum()).backward()
    worst = 0.0
    for prefix, net in (("sdf.", sdf), ("rgb.", rgb)):
        for n, p in net.named_parameters():
            want = torch.tensor(g["grad." + prefix + n]).to(dev)
            assert p.grad is not None and p.grad.shape == want.shape, prefix + n
            err = float((p.grad - want).abs().max()) / max(1.0, float(want.abs().max()))
            worst = max(worst, err)
            assert err < 2e-4, (prefix + n, err)
    print("variant %s: worst relative gradient error vs the reference %.2e" % (tag, worst))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_variant_training_render_matches_the_oracle(tag):
    from oracle import reference_ops as R
    from shapeclipper_amd.model.implicit import RGBNetwork, SDFNetwork
    from shapeclipper_amd.model.renderer import Renderer
    v = VARIANTS[tag]
    dev = torch.device("cuda:0")
    opt = _opt(v)
    opt.H, opt.W = 16, 16
    cfg = R.Cfg(H=16, W=16, hidden_sdf=v["cs"], posenc_sdf=v["ls"], skip_in=tuple(v["skip"]), latent_sdf=v["zs"], hidden_rgb=v["cr"],
                posenc_rgb=v["lr"], latent_rgb=v["zr"])
    torch.manual_seed(0)
    sdf_net, rgb_net = SDFNetwork(opt), RGBNetwork(opt)
    with torch.no_grad():
        for p in list(sdf_net.parameters()) + list(rgb_net.parameters()):
            p.add_(0.03 * torch.randn_like(p))
    Ws = {k: t.detach().clone().requires_grad_(True) for k, t in sdf_net.state_dict().items()}
    Wr = {k: t.detach().clone().requires_grad_(True) for k, t in rgb_net.state_dict().items()}
    r = Renderer(opt, sdf_net, rgb_net).to(dev)
    B, Rr = 2, 64
    trig = lambda t: torch.stack([torch.cos(t), torch.sin(t)], 1)
    sd = torch.tensor([0.9, 1.1])
    pose = R.pose_from_trig(cfg, trig(torch.tensor([0.3, -1.1])), trig(torch.tensor([0.2, -0.1])), trig(torch.zeros(2)), sd)
    intr = R.get_intr(cfg, torch.ones(B))
    zs, zr = torch.randn(B, v["zs"]), torch.randn(B, v["zr"])
    ray_idx = torch.stack([torch.randperm(256)[:Rr] for _ in range(B)])
    torch.manual_seed(5)
    state = torch.get_rng_state()
    zs_d, zr_d = zs.to(dev).requires_grad_(True), zr.to(dev).requires_grad_(True)
    out = r(opt, pose.to(dev), intr.to(dev), sd.to(dev), zs_d, zr_d, ray_idx=ray_idx.to(dev), training=True)
    L = out[0].sum() + out[1].sum() + (out[4] * out[2]).sum() + ((out[5] - 1) ** 2).mean()
    L.backward()
    torch.cuda.synchronize()
    torch.set_rng_state(state)
    t_rand, eik_idx, eik_pts = R.draw_render_randoms(B * Rr, 64, True)
    zs_c, zr_c = zs.clone().requires_grad_(True), zr.clone().requires_grad_(True)
    o = R.render(cfg, Ws, Wr, torch.tensor(0.1), pose, intr, sd, zs_c, zr_c, ray_idx, True, t_rand, eik_idx, eik_pts)
    Lc = o["rgb"].sum() + o["mask"].sum() + (o["normal"] * o["mask_hard"]).sum() + ((o["grad_eikonal"] - 1) ** 2).mean()
    Lc.backward()
    for k, got in (("rgb", out[0]), ("mask", out[1]), ("depth", out[3]), ("grad_eikonal", out[5])):
        assert (got.detach().cpu() - o[k].detach()).abs().max() < 1e-4, k
    for net, W, name in ((sdf_net, Ws, "sdf"), (rgb_net, Wr, "rgb")):
        for n, p in net.named_parameters():
            want = W[n].grad
            if want is None:
                want = torch.zeros_like(W[n])
            got = p.grad.cpu() if p.grad is not None else torch.zeros_like(want)
            assert float((got - want).abs().max()) <= 5e-4 * max(float(want.abs().max()), 1e-3), (name, n, float((got - want).abs().max()))
    assert (zs_d.grad.cpu() - zs_c.grad).abs().max() <= 5e-4 * zs_c.grad.abs().max()
    assert (zr_d.grad.cpu() - zr_c.grad).abs().max() <= 5e-4 * max(float(zr_c.grad.abs().max()), 1e-6)


def test_architectures_outside_the_family_take_the_stock_operator_path():
    """Round 5: wider / deeper networks, more octaves, other skip layers no longer raise -- they are flagged `eager` and run on
    model/eager_path.py (parity: tests/test_gpu_other_architectures.py); the packed-image entry point still refuses them."""
    import warnings
    from shapeclipper_amd.model.implicit import SDFNetwork
    for extra in (["--arch.impl_sdf.n_channels=128"], ["--arch.impl_sdf.pos_enc=10"], ["--arch.impl_sdf.n_hidden_layers=8"],
                  ["--arch.impl_sdf.skip_connection=[3]"]):
        from shapeclipper_amd.utils import options
        opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_arch", "--output_root=/tmp/sc_pytest"] + extra),
                          verbose=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = SDFNetwork(opt)
        assert net.eager
        with pytest.raises(NotImplementedError):
            net.packed(torch.zeros(1, 64))


def test_variant_architecture_trains_through_the_runner():
    """The whole training step (Graph: encoders, latent projectors sized by proj_latent_dim, two renders, losses, backward, Adam) with a variant
    architecture from the command line: two finite steps, every implicit-network parameter receives a gradient of its own shape."""
    import importlib.util, os
    from shapeclipper_amd.utils.util import EasyDict as edict
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    spec = importlib.util.spec_from_file_location("sc_bench_for_arch_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    v = VARIANTS["a"]
    extra = ["--arch.impl_sdf.n_channels=%d" % v["cs"], "--arch.impl_sdf.pos_enc=%d" % v["ls"], "--arch.impl_sdf.skip_connection=%s" % v["skip"],
             "--arch.impl_sdf.proj_latent_dim=%d" % v["zs"], "--arch.impl_rgb.n_channels=%d" % v["cr"], "--arch.impl_rgb.pos_enc=%d" % v["lr"],
             "--arch.impl_rgb.proj_latent_dim=%d" % v["zr"]]
    runner, opt, batch = bench.build_runner(2, extra=extra)
    g = runner.graph.module
    assert g.sdf_network.lin2.weight.shape == (48, 48 + 3 + 24 + 32) and g.rgb_network.lin0.weight.shape == (32, 3 + 30 + 48 + 48)
    for _ in range(2):
        opt.H, opt.W = opt.image_size
        loss = runner.train_iteration(opt, edict(batch), None)
    torch.cuda.synchronize()
    assert torch.isfinite(loss.all.detach()).item()
    for net in (g.sdf_network, g.rgb_network):
        for n, p in net.named_parameters():
            assert p.grad is not None and p.grad.shape == p.shape and torch.isfinite(p.grad).all(), n
