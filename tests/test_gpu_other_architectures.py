"""Architectures and sample counts OUTSIDE the family of the HIP chain kernels (VERDICT r04 missing #2; reference model/implicit.py:93-113,
199-214, model/renderer.py:13-37): depth other than 5 / 3, more than 64 channels or 6 octaves, another skip layer, n_samples != 64.  The
product runs them on stock device operators (shapeclipper_amd/model/eager_path.py) and says so once; golden G16 is what the REFERENCE's own
Renderer / SDFNetwork computes for SDF 6 x 128 (8 octaves, skip [3], latent 48) + RGB 2 x 96 (7 octaves, latent 32) at 32 samples per ray
(tests/golden/make_golden.py): training render + every gradient, evaluation render, conditional output with d sdf / dx."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

EXTRA = ["--arch.impl_sdf.n_hidden_layers=6", "--arch.impl_sdf.n_channels=128", "--arch.impl_sdf.pos_enc=8", "--arch.impl_sdf.skip_connection=[3]",
         "--arch.impl_sdf.proj_latent_dim=48", "--arch.impl_rgb.n_hidden_layers=2", "--arch.impl_rgb.n_channels=96", "--arch.impl_rgb.pos_enc=7",
         "--arch.impl_rgb.proj_latent_dim=32", "--render.n_samples_uniform=32"]


def _opt(extra=EXTRA):
    from shapeclipper_amd.utils import options
    return options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_arch16", "--output_root=/tmp/sc_pytest"] + list(extra)),
                       verbose=False)


def _renderer(g, opt, dev):
    from shapeclipper_amd.model.implicit import RGBNetwork, SDFNetwork
    from shapeclipper_amd.model.renderer import Renderer
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        from shapeclipper_amd.model import eager_path
        eager_path._WARNED.clear()
        sdf, rgb = SDFNetwork(opt), RGBNetwork(opt)
        r = Renderer(opt, sdf, rgb)
    assert sdf.eager and rgb.eager and r.eager
    assert any("stock PyTorch-ROCm operators" in str(x.message) for x in w)                  # said once, loudly
    sdf.load_state_dict({k[len("w.sdf."):]: torch.tensor(g[k]) for k in g.files if k.startswith("w.sdf.")}, strict=True)
    rgb.load_state_dict({k[len("w.rgb."):]: torch.tensor(g[k]) for k in g.files if k.startswith("w.rgb.")}, strict=True)
    with torch.no_grad():
        r.density.beta.fill_(float(g["beta"]))
    return r.to(dev)


def test_training_render_and_all_gradients_match_the_reference(golden):
    g = golden("g16_other_architecture")
    dev = torch.device("cuda:0")
    opt = _opt()
    opt.H = opt.W = 8
    r = _renderer(g, opt, dev)
    lv = {k: torch.tensor(g[k]).to(dev).requires_grad_(True) for k in ("pose", "intr", "scale_dist", "z_sdf", "z_rgb")}
    # the reference's CPU-generator draws, replayed: rand [B R, 32] -> randint [B R] -> uniform_ [B R, 3] (renderer.py:29,33,158)
    torch.manual_seed(1617)
    out = r(opt, lv["pose"], lv["intr"], lv["scale_dist"], lv["z_sdf"], lv["z_rgb"], ray_idx=torch.tensor(g["ray_idx"]).to(dev), training=True)
    for k, got in zip(("rgb", "mask", "mask_hard", "depth", "normal", "grad_eikonal"), out):
        want = torch.tensor(g[k]).to(dev)
        if k == "mask_hard":
            guard = (torch.tensor(g["mask"]).to(dev) - 0.5).abs() > 1e-5
            assert torch.equal(got[guard], want[guard])
        else:
            tol = 2e-4 if k in ("normal", "grad_eikonal", "depth") else 2e-5        # the bars of tests/test_gpu_parity_large.py
            assert (got - want).abs().max() < tol * max(1.0, float(want.abs().max())), (k, float((got - want).abs().max()))
    cot = {k: torch.tensor(g["cot." + k]).to(dev) for k in ("rgb", "mask", "depth", "normal", "eik")}
    L = ((out[0] * cot["rgb"]).sum() + (out[1] * cot["mask"]).sum() + (out[3] * cot["depth"]).sum() + (out[4] * cot["normal"]).sum()
         + (out[5] * cot["eik"]).sum())
    params = dict(r.named_parameters())
    names = list(params) + list(lv)
    grads = torch.autograd.grad(L, [params[n] if n in params else lv[n] for n in names], allow_unused=True)
    worst = {}
    for n, gg in zip(names, grads):
        want = torch.tensor(g["grad." + n]).to(dev)
        gv = gg if gg is not None else torch.zeros_like(want)
        worst[n] = float((gv - want).abs().max()) / max(float(want.abs().max()), 1e-4)
    # camera leaves: the fp32 evaluation of these sums is itself ~7e-4 from float64 (profiles/r03_fullstep_fp64_noise.txt); measured 4.5e-4
    bad = {k: v for k, v in worst.items() if v > (1e-3 if k in ("pose", "intr", "scale_dist") else 2e-4)}
    print("other architecture, gradient errors (max abs / max |ref|): worst %.1e" % max(worst.values()))
    assert not bad, bad


def test_evaluation_render_level_grid_and_conditional_output(golden):
    from shapeclipper_amd.utils import eval_3D
    g = golden("g16_other_architecture")
    dev = torch.device("cuda:0")
    opt = _opt()
    opt.H = opt.W = 8
    r = _renderer(g, opt, dev)
    t = lambda k: torch.tensor(g[k]).to(dev)
    with torch.no_grad():
        out = r(opt, t("pose"), t("intr"), t("scale_dist"), t("z_sdf"), t("z_rgb"), ray_idx=None, training=False)
    for k, got in (("eval_rgb", out[0]), ("eval_mask", out[1]), ("eval_depth", out[3]), ("eval_normal", out[4])):
        assert (got - t(k)).abs().max() < (2e-4 if k != "eval_rgb" and k != "eval_mask" else 2e-5) * max(1.0, float(t(k).abs().max())), k
    assert out[5] is None
    s, f, gr = r.sdf_network.get_conditional_output(opt, 2, t("pts").clone(), t("z_sdf"), compute_grad=True)
    assert (s - t("pts_sdf")).abs().max() < 1e-5 and (f - t("pts_feat")).abs().max() < 1e-5 and (gr - t("pts_grad")).abs().max() < 1e-4
    # the evaluation's level grid goes through the same path (slab by slab) and agrees with point-wise evaluation
    opt.eval.vox_res = 6
    from shapeclipper_amd.utils.util import EasyDict as edict
    opt.device = "cuda:0"
    grid = eval_3D.get_dense_3D_grid(opt, edict(idx=torch.arange(2)))
    level = eval_3D.compute_level_grid(opt, r.sdf_network, t("z_sdf"), grid)
    assert level.shape == (2, 7, 7, 7)
    direct = r.sdf_network.get_conditional_output(opt, 2, grid.reshape(-1, 3).to(dev), t("z_sdf"), compute_grad=False)[0].view(2, 7, 7, 7)
    assert (level - direct).abs().max() < 1e-5


def test_no_cpu_fallback_and_family_members_stay_on_hip():
    from shapeclipper_amd.model.implicit import SDFNetwork
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = SDFNetwork(_opt())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.get_conditional_output(_opt(), 1, torch.zeros(4, 3), torch.zeros(1, 48), compute_grad=True)
    assert not SDFNetwork(_opt([])).eager                                  # the shipped architecture: HIP kernels
    assert not SDFNetwork(_opt(["--arch.impl_sdf.n_channels=48", "--arch.impl_sdf.skip_connection=[2]"])).eager


def test_other_architecture_trains_through_the_runner():
    """Two full training steps (encoders, projectors sized by proj_latent_dim, two renders, losses, backward, Adam) of a 4 x 128 / 4 x 96
    configuration: finite losses, every implicit-network parameter receives a gradient."""
    import importlib.util
    import os
    from shapeclipper_amd.utils.util import EasyDict as edict
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    spec = importlib.util.spec_from_file_location("sc_bench_for_arch16_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        runner, opt, batch = bench.build_runner(2, extra=["--arch.impl_sdf.n_hidden_layers=4", "--arch.impl_sdf.n_channels=128", "--arch.impl_sdf.pos_enc=7",
                                                          "--arch.impl_rgb.n_hidden_layers=4", "--arch.impl_rgb.n_channels=96", "--render.rand_sample=64"])
    assert runner.graph.module.renderer.eager
    for _ in range(2):
        opt.H, opt.W = opt.image_size
        loss = runner.train_iteration(opt, edict(batch), None)
    torch.cuda.synchronize()
    assert torch.isfinite(loss.all.detach()).item()
    for n, p in list(runner.graph.module.sdf_network.named_parameters()) + list(runner.graph.module.rgb_network.named_parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
