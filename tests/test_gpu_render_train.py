"""Training-mode render: outputs AND every gradient vs golden G6, captured from the reference's
Renderer.forward(training=True) + autograd (model/renderer.py:57-185), B=2, R=32 random rays of an
8x8 image, 64 samples, eikonal branch included.  The CPU generator is seeded like the capture, so
the product draws the identical stratified jitter / eikonal samples.

Bars (fp32): outputs 5e-5 abs (normals 2e-3); gradients 2e-3 relative to each tensor's max entry."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _opt(H, W):
    from shapeclipper_amd.utils import options
    o = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest",
                                             "--output_root=/tmp/sc_pytest"]), verbose=False)
    o.H, o.W = H, W
    return o


def _renderer(golden, opt, dev):
    from shapeclipper_amd.model.implicit import RGBNetwork, SDFNetwork
    from shapeclipper_amd.model.renderer import Renderer
    g = golden("g2_networks")
    sdf_net, rgb_net = SDFNetwork(opt), RGBNetwork(opt)
    sdf_net.load_state_dict({k[len("pert.sdf."):]: torch.tensor(g[k]) for k in g.files if k.startswith("pert.sdf.")})
    rgb_net.load_state_dict({k[len("pert.rgb."):]: torch.tensor(g[k]) for k in g.files if k.startswith("pert.rgb.")})
    r = Renderer(opt, sdf_net, rgb_net).to(dev)
    return r


def test_render_train_outputs_and_all_gradients(golden):
    dev = torch.device("cuda:0")
    opt = _opt(8, 8)
    g = golden("g6_render_train")
    r = _renderer(golden, opt, dev)
    with torch.no_grad():
        r.density.beta.fill_(float(g["beta"]))
    t = lambda k: torch.tensor(g[k], device=dev)
    leaves = {k: t(k).requires_grad_(True) for k in ("pose", "intr", "scale_dist", "z_sdf", "z_rgb")}
    torch.manual_seed(78)      # same CPU stream as the capture
    out = r(opt, leaves["pose"], leaves["intr"], leaves["scale_dist"], leaves["z_sdf"], leaves["z_rgb"],
            ray_idx=t("ray_idx"), training=True)
    rgb, mask, mask_hard, depth, normal, eik = out
    chk = lambda a, k, tol: np.testing.assert_allclose(a.detach().cpu().numpy(), g[k], atol=tol, rtol=0)
    chk(rgb, "rgb", 5e-5); chk(mask, "mask", 5e-5); chk(depth, "depth", 1e-4); chk(normal, "normal", 2e-3)
    chk(eik, "grad_eikonal", 2e-4)
    guard = np.abs(g["mask"] - 0.5) > 1e-5
    assert np.array_equal(mask_hard.cpu().numpy()[guard], g["mask_hard"][guard])

    L = ((rgb * t("cot.rgb")).sum() + (mask * t("cot.mask")).sum() + (depth * t("cot.depth")).sum()
         + (normal * t("cot.normal")).sum() + (eik * t("cot.eik")).sum())
    params = dict(r.named_parameters())
    names = list(params) + list(leaves)
    grads = torch.autograd.grad(L, list(params.values()) + list(leaves.values()), allow_unused=True)
    torch.cuda.synchronize()
    worst = {}
    for n, gr in zip(names, grads):
        ref = g["grad." + n]
        got = gr.cpu().numpy() if gr is not None else np.zeros_like(ref)
        scale = max(np.abs(ref).max(), 1e-4)
        worst[n] = np.abs(got - ref).max() / scale
    bad = {k: v for k, v in worst.items() if v > 2e-3}
    assert not bad, bad


def test_eval_render_through_module(golden):
    dev = torch.device("cuda:0")
    opt = _opt(8, 8)
    g = golden("g5_render_eval")
    r = _renderer(golden, opt, dev)
    with torch.no_grad():
        r.density.beta.fill_(float(g["beta"]))
        t = lambda k: torch.tensor(g[k], device=dev)
        rgb, mask, mask_hard, depth, normal, eik = r(opt, t("pose"), t("intr"), t("scale_dist"), t("z_sdf"),
                                                     t("z_rgb"), ray_idx=None, training=False)
    assert eik is None and rgb.shape == (2, 64, 3) and normal.shape == (2, 64, 3)
    np.testing.assert_allclose(rgb.cpu().numpy(), g["rgb"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(mask.cpu().numpy(), g["mask"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(normal.cpu().numpy(), g["normal"], atol=2e-3, rtol=0)
