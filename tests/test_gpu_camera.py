"""Fused camera kernels (csrc/camera.hip) against the torch camera algebra of utils/camera.py (itself pinned to the
reference by golden G8): ray set-up of a render and the estimator-output -> pose/intrinsics map, values and gradients.
Tolerance: fp32, 1e-5 relative to each tensor's scale (different summation order only)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-5, what=""):
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max())
    assert err <= tol * scale, "%s: max err %.3e (scale %.3e)" % (what, err, scale)


def _opt(H, W):
    from shapeclipper_amd.utils.util import EasyDict as edict
    return edict(H=H, W=W, camera=edict(model="perspective", dist=5.0, focal=4.0))


def _random_pose(B, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    t = torch.randn(B, 3, 1, generator=g) * 0.3 + torch.tensor([0.0, 0.0, 5.0]).view(1, 3, 1)
    return torch.cat([q, t], dim=-1).cuda()


@pytest.mark.parametrize("B,R,H,W,sampled", [(3, 512, 224, 224, True), (2, 64, 8, 8, False), (1, 37, 16, 24, True)])
def test_camera_rays_match_torch_algebra(B, R, H, W, sampled):
    from shapeclipper_amd.functional import CameraRaysFunction
    from shapeclipper_amd.utils import camera
    opt = _opt(H, W)
    torch.manual_seed(0)
    pose0 = _random_pose(B, 1)
    f = 4.0 * (1 + 0.1 * torch.randn(B, device="cuda"))
    intr0 = camera.get_intr(opt, f / 4.0)
    intr0[:, 0, 1] = 0.01 * W                                   # general K: skew, so that the full inverse is exercised
    ray_idx = torch.stack([torch.randperm(H * W)[:R] for _ in range(B)]).cuda() if sampled else None
    cots = [torch.randn(B * R, 3, device="cuda"), torch.randn(B * R, 3, device="cuda"), torch.randn(B * R, device="cuda")]

    pa, ka = pose0.clone().requires_grad_(True), intr0.clone().requires_grad_(True)
    centre, ray = camera.get_center_and_ray(opt, pa, intr=ka, ray_idx=ray_idx)
    d = F.normalize(ray, dim=-1)
    df = d.norm(dim=-1, keepdim=True) / ray.norm(dim=-1, keepdim=True)
    ref = (centre.expand(B, R, 3).reshape(-1, 3), d.reshape(-1, 3), df.reshape(-1))
    sum((o * c).sum() for o, c in zip(ref, cots)).backward()

    pb, kb = pose0.clone().requires_grad_(True), intr0.clone().requires_grad_(True)
    got = CameraRaysFunction.apply(pb, kb, ray_idx, R, W)
    sum((o * c).sum() for o, c in zip(got, cots)).backward()
    for g_, r_, name in zip(got, ref, ("cam_loc", "ray_dirs", "depth_fac")):
        _close(g_.detach(), r_.detach(), what=name)
    _close(pb.grad, pa.grad, tol=2e-4, what="d pose")          # sums over R rays of O(1) terms with cancellation
    _close(kb.grad, ka.grad, tol=2e-4, what="d intr")


def test_pose_from_trig_matches_torch_algebra():
    from shapeclipper_amd.functional import PoseFromTrigFunction
    from shapeclipper_amd.model.graph import rotation_from_trig
    from shapeclipper_amd.utils import camera
    B = 5
    opt = _opt(64, 48)
    torch.manual_seed(1)
    raw = [torch.randn(B, 2, device="cuda") for _ in range(3)]
    sf0, sd0 = 1 + 0.2 * torch.randn(B, device="cuda"), 1 + 0.2 * torch.randn(B, device="cuda")
    cot_p, cot_k = torch.randn(B, 3, 4, device="cuda"), torch.randn(B, 3, 3, device="cuda")
    outs = []
    for fused in (False, True):
        leaves = [r.clone().requires_grad_(True) for r in raw] + [sf0.clone().requires_grad_(True), sd0.clone().requires_grad_(True)]
        az, el, th = (F.normalize(v, dim=1) for v in leaves[:3])
        sf, sd = leaves[3], leaves[4]
        if fused:
            pose, intr = PoseFromTrigFunction.apply(az, el, th, sf, sd, 5.0, 4.0, opt.W, opt.H)
        else:
            pose_R = camera.pose(R=rotation_from_trig(az, el, th))
            tz = sd * opt.camera.dist
            pose_T = camera.pose(t=torch.stack([torch.zeros_like(tz), torch.zeros_like(tz), tz], dim=-1))
            pose = camera.pose.compose([pose_R, pose_T])
            intr = camera.get_intr(opt, sf)
        ((pose * cot_p).sum() + (intr * cot_k).sum()).backward()
        outs.append((pose.detach(), intr.detach(), [l.grad for l in leaves]))
    _close(outs[1][0], outs[0][0], what="pose")
    _close(outs[1][1], outs[0][1], what="intr")
    for g_, r_, name in zip(outs[1][2], outs[0][2], ("azim", "elev", "theta", "scale_focal", "scale_dist")):
        _close(g_, r_, what="d " + name)
