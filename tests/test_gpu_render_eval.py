"""Eval-mode render (training=False): HIP sdf_fwd + rgb_composite_fwd vs golden G5 captured from the
reference's Renderer.forward (model/renderer.py:57-185), B=2, 8x8 pixels, 64 samples/ray.

Bar: 5e-5 absolute on rgb/mask/depth (values O(1)), 2e-3 on unit normals (ratio of small
numbers), mask_hard bit-exact outside a 1e-5 guard band around 0.5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _weights(golden, dev):
    g = golden("g2_networks")
    Ws = {k[len("pert.sdf."):]: torch.tensor(g[k], device=dev) for k in g.files if k.startswith("pert.sdf.")}
    Wr = {k[len("pert.rgb."):]: torch.tensor(g[k], device=dev) for k in g.files if k.startswith("pert.rgb.")}
    return Ws, Wr


def test_render_eval_golden(golden):
    from types import SimpleNamespace as NS
    from shapeclipper_amd import ops, packing
    from shapeclipper_amd.utils import camera
    dev = torch.device("cuda:0")
    g = golden("g5_render_eval")
    Ws, Wr = _weights(golden, dev)
    t = lambda k: torch.tensor(g[k], device=dev)
    opt = NS(H=8, W=8, camera=NS(model="perspective", dist=5.0, focal=4.0))
    pose, intr, sd = t("pose"), t("intr"), t("scale_dist")
    B, R, S = 2, 64, 64
    center, ray = camera.get_center_and_ray(opt, pose, intr=intr)
    d = torch.nn.functional.normalize(ray, dim=-1)
    depth_fac = (d.norm(dim=-1) / ray.norm(dim=-1)).reshape(-1).contiguous()
    tl = torch.linspace(0.0, 1.0, S, device=dev)
    near = (5.0 * sd - 0.7)[:, None, None]; far = (5.0 * sd + 0.7)[:, None, None]
    z = (near * (1 - tl) + far * tl).expand(B, R, S).reshape(B * R, S).contiguous()
    pts = (center.expand(B, R, 3).reshape(-1, 1, 3) + z[..., None] * d.reshape(-1, 1, 3)).reshape(-1, 3).contiguous()
    np.testing.assert_allclose(pts.cpu().numpy(), g["points"], atol=2e-6)

    pack, cb = packing.pack_sdf(Ws, t("z_sdf"))
    vpack, db = packing.pack_rgb(Wr, t("z_rgb"))
    sdf, grad, feat = ops.sdf_forward(pts, pack, cb, R * S)
    out = ops.rgb_composite_forward(pts, z, depth_fac, sdf, grad, feat, vpack, db, t("beta").reshape(1),
                                    R, True, 1e-4, 1.0, 1.0, keep_samples=True)
    torch.cuda.synchronize()
    c = lambda k: out[k].cpu().numpy()
    assert np.abs(sdf.cpu().numpy() - g["sdf"][:, 0]).max() < 2e-5
    assert np.abs(c("weights") - g["weights"]).max() < 5e-5
    assert np.abs(c("rgb_flat") - g["rgb_flat"]).max() < 5e-5
    assert np.abs(c("rgb") - g["rgb"].reshape(-1, 3)).max() < 5e-5
    assert np.abs(c("mask") - g["mask"].reshape(-1)).max() < 5e-5
    assert np.abs(c("depth") - g["depth"].reshape(-1)).max() < 1e-4
    assert np.abs(c("normal") - g["normal"].reshape(-1, 3)).max() < 2e-3
    guard = np.abs(g["mask"].reshape(-1) - 0.5) > 1e-5
    assert np.array_equal(c("mask_hard")[guard], g["mask_hard"].reshape(-1)[guard])
    assert set(np.unique(c("mask_hard"))) <= {0.0, 1.0}
    assert np.all(c("alpha")[:, -1] == 0.0)            # last sample: dist = 0 -> alpha = 0 (renderer.py:196)
    assert np.all(c("weights").sum(1) <= 1.0 + 1e-5)
