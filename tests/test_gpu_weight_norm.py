"""arch.impl_sdf.weight_norm / arch.impl_rgb.weight_norm = true (reference model/implicit.py:130-132, 212-214: nn.utils.weight_norm on every
Linear of the two MLPs; state-dict keys lin{l}.weight_g / weight_v / bias).  The kernels see the effective weight g * v / ||v||, built
from those parameters on the device each step; gradients flow back to g and v through the packing gather.

  * golden G14 (captured from the REFERENCE's own modules with the option on, tests/golden/make_golden.py): the reference's state dict
    loads strictly; sdf / feature / d sdf/dx from the HIP kernels and the colour MLP match; the gradients of a fixed functional w.r.t.
    weight_g, weight_v and bias of both networks match the reference's;
  * a training render (HIP render + compositing + fused backward) with weight-normalised networks equals the render of plain networks
    holding the effective weights, output for output, and its g / v gradients are the chain rule of the plain weights' gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _opt(extra=()):
    from shapeclipper_amd.utils import options
    return options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_wn", "--output_root=/tmp/sc_pytest"] + list(extra)),
                       verbose=False)


WN = ["--arch.impl_sdf.weight_norm", "--arch.impl_rgb.weight_norm"]


def test_weight_norm_networks_match_the_reference_capture(golden):
    from shapeclipper_amd.model.implicit import RGBNetwork, SDFNetwork
    g = golden("g14_weight_norm")
    dev = torch.device("cuda:0")
    opt = _opt(WN)
    assert opt.arch.impl_sdf.weight_norm is True and opt.arch.impl_rgb.weight_norm is True
    sdf, rgb = SDFNetwork(opt), RGBNetwork(opt)
    sdf.load_state_dict({k[len("w.sdf."):]: torch.tensor(g[k]) for k in g.files if k.startswith("w.sdf.")}, strict=True)
    rgb.load_state_dict({k[len("w.rgb."):]: torch.tensor(g[k]) for k in g.files if k.startswith("w.rgb.")}, strict=True)
    sdf, rgb = sdf.to(dev), rgb.to(dev)
    pts, zs, zr = (torch.tensor(g[k]).to(dev) for k in ("pts", "z_sdf", "z_rgb"))
    B, N = zs.shape[0], pts.shape[0] // zs.shape[0]
    s, f, gr = sdf.get_conditional_output(opt, B, pts.clone(), zs, compute_grad=True)
    lat = zr.unsqueeze(1).repeat(1, N, 1).view(B * N, -1)
    c = rgb(pts, lat, f)
    ref = {k: torch.tensor(g[k]).to(dev) for k in ("sdf", "feat", "grad", "rgb")}
    assert (s - ref["sdf"]).abs().max() < 2e-5 and (f - ref["feat"]).abs().max() < 2e-5
    assert (gr - ref["grad"]).abs().max() < 2e-4 * max(1.0, float(ref["grad"].abs().max()))
    assert (c - ref["rgb"]).abs().max() < 2e-5
    cot = {k: torch.tensor(g["cot." + k]).to(dev) for k in ("sdf", "feat", "grad", "rgb")}
    L = (s * cot["sdf"]).sum() + (f * cot["feat"]).sum() + (gr * cot["grad"]).sum() + (c * cot["rgb"]).sum()
    L.backward()
    worst = 0.0
    for prefix, net in (("sdf.", sdf), ("rgb.", rgb)):
        for n, p in net.named_parameters():
            want = torch.tensor(g["grad." + prefix + n]).to(dev)
            assert p.grad is not None, prefix + n
            err = float((p.grad - want).abs().max()) / max(1.0, float(want.abs().max()))
            worst = max(worst, err)
            assert err < 2e-4, (prefix + n, err)
    print("weight_norm: worst relative gradient error vs the reference %.2e" % worst)


def test_weight_norm_render_equals_the_render_of_the_effective_weights():
    from shapeclipper_amd.model.implicit import RGBNetwork, SDFNetwork
    from shapeclipper_amd.model.renderer import Renderer
    from shapeclipper_amd.utils import camera
    dev = torch.device("cuda:0")
    opt_wn, opt = _opt(WN), _opt()
    torch.manual_seed(0)
    sdf_wn, rgb_wn = SDFNetwork(opt_wn), RGBNetwork(opt_wn)
    with torch.no_grad():
        for p in list(sdf_wn.parameters()) + list(rgb_wn.parameters()):
            p.add_(0.03 * torch.randn_like(p))
    sdf, rgb = SDFNetwork(opt).to(dev), RGBNetwork(opt).to(dev)
    sdf_wn, rgb_wn = sdf_wn.to(dev), rgb_wn.to(dev)          # the effective weights are formed on the device, as the forward pass forms them
    for tag, a, b in (("sdf", sdf_wn, sdf), ("rgb", rgb_wn, rgb)):
        for l in range(a.num_layers - 1):
            la, lb = getattr(a, "lin%d" % l), getattr(b, "lin%d" % l)
            with torch.no_grad():
                lb.weight.copy_(torch._weight_norm(la.weight_v, la.weight_g, 0))
                lb.bias.copy_(la.bias)
    r_wn, r = Renderer(opt_wn, sdf_wn, rgb_wn).to(dev), Renderer(opt, sdf, rgb).to(dev)
    B, R = 4, 512
    az = (torch.rand(B) * 2 - 1) * 3.14159
    trig = lambda t: torch.stack([torch.cos(t), torch.sin(t)], 1)
    Ry = camera.azim_to_rotation_matrix(trig(az), "trig"); Rx = camera.elev_to_rotation_matrix(trig(torch.zeros(B)), "trig")
    Pm = torch.tensor([[-1., 0, 0], [0, 0, -1], [0, -1, 0]])[None].expand(B, 3, 3)
    pose = camera.pose.compose([camera.pose(R=Rx @ Ry @ Pm), camera.pose(t=torch.tensor([[0., 0, 5.]]).expand(B, 3))]).to(dev)
    intr = camera.get_intr(opt, torch.ones(B)).to(dev)
    args = [pose, intr, torch.ones(B, device=dev), torch.randn(B, 64, device=dev), torch.randn(B, 64, device=dev)]
    ray_idx = torch.stack([torch.randperm(224 * 224)[:R] for _ in range(B)]).to(dev)

    def run(rr, o):
        torch.manual_seed(3)
        out = rr(o, *args, ray_idx=ray_idx, training=True)
        Lf = out[0].square().sum() + out[1].sum() + (out[4] * out[2]).sum() + out[3].sum() + ((out[5] - 1) ** 2).mean()
        Lf.backward()
        return [t.detach() for t in out if t is not None]
    o_wn, o_pl = run(r_wn, opt_wn), run(r, opt)
    for x, y in zip(o_wn, o_pl):
        assert torch.equal(x, y), "render outputs differ between weight-normalised networks and their effective weights"
    # chain rule: dL/dg, dL/dv from dL/dW of the plain network
    for tag, a, b in (("sdf", sdf_wn, sdf), ("rgb", rgb_wn, rgb)):
        for l in range(b.num_layers - 1):
            la, lb = getattr(a, "lin%d" % l), getattr(b, "lin%d" % l)
            v, gg = la.weight_v.detach().clone().requires_grad_(True), la.weight_g.detach().clone().requires_grad_(True)
            dv, dg = torch.autograd.grad(torch._weight_norm(v, gg, 0), [v, gg], lb.weight.grad)
            for name, got, want in (("weight_v", la.weight_v.grad, dv), ("weight_g", la.weight_g.grad, dg), ("bias", la.bias.grad, lb.bias.grad)):
                assert got is not None, (tag, l, name)
                assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())), (tag, l, name, float((got - want).abs().max()))
