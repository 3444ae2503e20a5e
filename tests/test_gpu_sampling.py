"""Ray-sampling kernels and the single-call C orchestrators vs torch formulations."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_sample(cam_loc, ray_dirs, scale_dist, u, R, dist):
    n = ray_dirs.shape[0]
    c = (dist * scale_dist).repeat_interleave(R).view(n, 1)
    near, far = c - 0.7, c + 0.7
    t = torch.linspace(0.0, 1.0, steps=64).to(ray_dirs.device)
    z = near * (1.0 - t) + far * t
    if u is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * u
    pts = (cam_loc.unsqueeze(1) + z.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
    return z, pts


@pytest.mark.parametrize("training", [True, False])
def test_ray_sample_matches_torch_ops_and_adjoint(training):
    from shapeclipper_amd.functional import RaySampleFunction
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, R = 3, 37
    o = torch.randn(B * R, 3, device=dev, requires_grad=True)
    d = torch.nn.functional.normalize(torch.randn(B * R, 3, device=dev), dim=-1).requires_grad_(True)
    sd = (0.8 + 0.4 * torch.rand(B, device=dev)).requires_grad_(True)
    u = torch.rand(B * R, 64, device=dev) if training else None
    z, p = RaySampleFunction.apply(o, d, sd, u, R, 5.0)
    zr, pr = _torch_sample(o, d, sd, u, R, 5.0)
    assert torch.equal(z, zr), (z - zr).abs().max()          # same fp32 op order: bit-identical
    assert torch.equal(p, pr)
    cz, cp = torch.randn_like(z), torch.randn_like(p)
    g = torch.autograd.grad((z * cz).sum() + (p * cp).sum(), [o, d, sd])
    gr = torch.autograd.grad((zr * cz).sum() + (pr * cp).sum(), [o, d, sd])
    for a, b in zip(g, gr):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)


def test_single_call_render_and_grid_equal_the_composed_path(golden):
    from shapeclipper_amd import ops, packing
    dev = torch.device("cuda:0")
    g, g2 = golden("g5_render_eval"), golden("g2_networks")
    Ws = {k[len("pert.sdf."):]: torch.tensor(g2[k], device=dev) for k in g2.files if k.startswith("pert.sdf.")}
    Wr = {k[len("pert.rgb."):]: torch.tensor(g2[k], device=dev) for k in g2.files if k.startswith("pert.rgb.")}
    t = lambda k: torch.tensor(g[k], device=dev)
    pack, cb = packing.pack_sdf(Ws, t("z_sdf")); vpack, db = packing.pack_rgb(Wr, t("z_rgb"))
    from types import SimpleNamespace as NS
    from shapeclipper_amd.utils import camera
    opt = NS(H=8, W=8, camera=NS(model="perspective", dist=5.0, focal=4.0))
    center, ray = camera.get_center_and_ray(opt, t("pose"), intr=t("intr"))
    d = torch.nn.functional.normalize(ray, dim=-1)
    depth_fac = (d.norm(dim=-1) / ray.norm(dim=-1)).reshape(-1).contiguous()
    out = ops.render_forward(center.expand(2, 64, 3).reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous(), depth_fac,
                             t("scale_dist"), None, pack, cb, vpack, db, t("beta").reshape(1), 64, True, 5.0, 1e-4, 1.0, 1.0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out["rgb"].cpu().numpy(), g["rgb"].reshape(-1, 3), atol=5e-5)
    np.testing.assert_allclose(out["mask"].cpu().numpy(), g["mask"].reshape(-1), atol=5e-5)
    np.testing.assert_allclose(out["points"].cpu().numpy(), g["points"], atol=2e-6)
    # level grid: on-the-fly grid == torch.linspace grid (bit-exact) through the same kernel
    g10 = golden("g10_eval3d")
    pack2, cb2 = packing.pack_sdf(Ws, torch.tensor(g10["z_sdf"], device=dev))
    lvl = ops.sdf_grid_forward(pack2, cb2, -0.6, 0.6, 7)
    np.testing.assert_allclose(lvl.cpu().numpy(), g10["level"], atol=2e-5)


def test_loss_backward_entry_point_scales_in_place():
    import ctypes
    from shapeclipper_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    G = torch.tensor([2.0, 3.0, 4.0, 5.0], device=dev)
    a, b, c, e, t = (torch.ones(n, device=dev) for n in (30, 10, 30, 20, 30))
    rc = lib.sc_loss_fused_backward(_lib.ptr(G), _lib.ptr(a), ctypes.c_longlong(30), _lib.ptr(b), ctypes.c_longlong(10),
                                    _lib.ptr(c), ctypes.c_longlong(30), _lib.ptr(e), ctypes.c_longlong(20), _lib.ptr(t), _lib.stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert a.unique().item() == 2 and b.unique().item() == 3 and c.unique().item() == 4 and e.unique().item() == 5
    assert t.unique().item() == 4
