"""sc_render_forward (training form) + sc_render_backward -- one C call each way for a whole training render -- against
the step-by-step path the Python host drives through autograd (RaySampleFunction -> SdfFunction -> RgbCompositeFunction,
itself pinned to the reference by golden G6).  Same kernels, so the bar is tight: outputs bit-identical, gradients equal
up to atomic-accumulation order (1e-5 relative to each tensor's scale)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-5, what=""):
    scale = max(float(b.abs().max()), 1e-9)
    err = float((a - b).abs().max())
    assert err <= tol * scale, "%s: max err %.3e (scale %.3e)" % (what, err, scale)


@pytest.mark.parametrize("B,R,with_gz", [(2, 32, False), (3, 64, True)])
def test_render_backward_single_call_equals_autograd_path(golden, B, R, with_gz, monkeypatch):
    from shapeclipper_amd import _lib, ops, packing
    # sc_render_forward / sc_render_backward chain the fp32-MFMA kernels (sdf_fwd.hip, rgb_composite_fwd_kernel): the Python path is
    # compared in the same arithmetic (round 6: its default forward kernels are the split forms, `--hip.sdf_stream!` / `--hip.rgb_split!`)
    monkeypatch.setattr(ops, "SDF_FWD_STREAM", False)
    monkeypatch.setattr(ops, "RGB_FWD_SPLIT", False)
    from shapeclipper_amd.functional import RaySampleFunction, RgbCompositeFunction, SdfFunction
    from shapeclipper_amd.packing import n_tiles
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g2 = golden("g2_networks")
    Ws = {k[len("pert.sdf."):]: torch.tensor(g2[k], device=dev) for k in g2.files if k.startswith("pert.sdf.")}
    Wr = {k[len("pert.rgb."):]: torch.tensor(g2[k], device=dev) for k in g2.files if k.startswith("pert.rgb.")}
    torch.manual_seed(5)
    n_rays, P = B * R, B * R * 64
    z_sdf, z_rgb = torch.randn(B, 64, device=dev) * 0.3, torch.randn(B, 64, device=dev) * 0.3
    w_pack0, cbias0 = packing.pack_sdf(Ws, z_sdf)
    v_pack0, dbias0 = packing.pack_rgb(Wr, z_rgb)
    cam0 = (torch.tensor([0.0, 0.0, -5.0], device=dev) + 0.05 * torch.randn(B, 1, 3, device=dev)).expand(B, R, 3).reshape(-1, 3).contiguous()
    dirs0 = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev) * 0.12 + torch.tensor([0.0, 0.0, 1.0], device=dev), dim=-1)
    df0 = 1.0 + 0.01 * torch.rand(n_rays, device=dev)
    sd0 = 1.0 + 0.05 * torch.randn(B, device=dev)
    u = torch.rand(n_rays, 64, device=dev)
    beta0 = torch.tensor([0.1], device=dev)
    cots = dict(rgb=torch.randn(n_rays, 3, device=dev), mask=torch.randn(n_rays, device=dev), depth=torch.randn(n_rays, device=dev),
                normal=torch.randn(n_rays, 3, device=dev), z=torch.randn(n_rays, 64, device=dev) * 0.1 if with_gz else None)

    # ---- reference: the autograd path of the Python host ----
    leaves = [t.clone().requires_grad_(True) for t in (cam0, dirs0, df0, sd0, w_pack0, cbias0, v_pack0, dbias0, beta0)]
    cam, dirs, df, sd, w_pack, cbias, v_pack, dbias, beta = leaves
    z_vals, pts = RaySampleFunction.apply(cam, dirs, sd, u, R, 5.0)
    sdf, grad, feat = SdfFunction.apply(pts, w_pack, cbias, R * 64, True, True, True)
    outs = RgbCompositeFunction.apply(pts, z_vals, df, sdf, grad, feat, v_pack, dbias, beta, R, True, 1e-4, 1.0, 1.0, False)
    rgb, mask, mask_hard, depth, normal = outs[:5]
    loss = (rgb * cots["rgb"]).sum() + (mask * cots["mask"]).sum() + (depth * cots["depth"]).sum() + (normal * cots["normal"]).sum()
    if with_gz:
        loss = loss + (z_vals * cots["z"]).sum()
    loss.backward()
    ref = dict(cam=cam.grad, dirs=dirs.grad, df=df.grad, sd=sd.grad, w=w_pack.grad, cb=cbias.grad, v=v_pack.grad, db=dbias.grad, beta=beta.grad)

    # ---- one C call each way ----
    f32 = dict(device=dev, dtype=torch.float32)
    T = n_tiles(P) * 1024
    o = dict(rgb=torch.empty(n_rays, 3, **f32), mask=torch.empty(n_rays, **f32), mask_hard=torch.empty(n_rays, **f32),
             depth=torch.empty(n_rays, **f32), normal=torch.empty(n_rays, 3, **f32), z=torch.empty(n_rays, 64, **f32),
             pts=torch.empty(P, 3, **f32), sdf=torch.empty(P, **f32), grad=torch.empty(P, 3, **f32), feat=torch.empty(T, **f32),
             sa=torch.empty(5 * T, **f32), sp=torch.empty(4 * T, **f32), rgb_flat=torch.empty(P, 3, **f32))
    p, ci, cf = _lib.ptr, ctypes.c_int, ctypes.c_float
    rc = lib.sc_render_forward(p(cam0), p(dirs0), p(df0), p(sd0), p(u), p(w_pack0), p(cbias0), p(v_pack0), p(dbias0), p(beta0),
                               ci(n_rays), ci(R), ci(B), ci(1), cf(5.0), cf(1e-4), cf(1.0), cf(1.0), p(o["rgb"]), p(o["mask"]),
                               p(o["mask_hard"]), p(o["depth"]), p(o["normal"]), p(o["z"]), p(o["pts"]), p(o["sdf"]), p(o["grad"]),
                               p(o["feat"]), None, p(o["sa"]), p(o["sp"]), p(o["rgb_flat"]), _lib.stream())
    assert rc == 0
    assert torch.equal(o["rgb"], rgb.detach()) and torch.equal(o["mask"], mask.detach()) and torch.equal(o["normal"], normal.detach())
    assert torch.equal(o["mask_hard"], mask_hard.detach()) and torch.equal(o["z"], z_vals.detach())
    fn = lib._cdll.sc_render_backward_workspace_bytes
    fn.restype = ctypes.c_longlong
    nbytes = int(fn(ci(n_rays)))
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    from shapeclipper_amd.packing import RGB_PACK_FLOATS, SDF_PACK_FLOATS
    g = dict(w=torch.full((SDF_PACK_FLOATS,), float("nan"), **f32), cb=torch.full((5, B, 64), float("nan"), **f32),
             v=torch.full((RGB_PACK_FLOATS,), float("nan"), **f32), db=torch.full((3, B, 64), float("nan"), **f32),
             beta=torch.full((1,), float("nan"), **f32), cam=torch.empty(n_rays, 3, **f32), dirs=torch.empty(n_rays, 3, **f32),
             sd=torch.empty(n_rays, **f32), df=torch.empty(n_rays, **f32))
    rc = lib.sc_render_backward(p(dirs0), p(df0), p(w_pack0), p(v_pack0), p(dbias0), p(beta0), p(o["z"]), p(o["pts"]), p(o["sdf"]),
                                p(o["grad"]), p(o["feat"]), p(o["sa"]), p(o["sp"]), p(o["rgb_flat"]), ci(n_rays), ci(R), ci(B), ci(1),
                                cf(5.0), cf(1e-4), cf(1.0), cf(1.0), p(cots["rgb"]), p(cots["mask"]), p(cots["depth"]), p(cots["normal"]),
                                p(cots["z"]), p(g["w"]), p(g["cb"]), p(g["v"]), p(g["db"]), p(g["beta"]), p(g["cam"]), p(g["dirs"]),
                                p(g["sd"]), p(g["df"]), p(ws), ctypes.c_longlong(nbytes), _lib.stream())
    assert rc == 0
    torch.cuda.synchronize()
    _close(g["w"], ref["w"], what="g sdf_pack")
    _close(g["cb"].permute(1, 0, 2), ref["cb"], what="g cbias")
    _close(g["v"], ref["v"], what="g rgb_pack")
    _close(g["db"].permute(1, 0, 2), ref["db"], what="g dbias")
    _close(g["beta"], ref["beta"].reshape(1), what="g beta")
    _close(g["cam"], ref["cam"], what="g cam_loc")
    _close(g["dirs"], ref["dirs"], what="g ray_dirs")
    _close(g["sd"].view(B, R).sum(dim=1), ref["sd"], what="g scale_dist")
    _close(g["df"], ref["df"], what="g depth_fac")
