"""Camera algebra for the MI355X build -- same call surface as the reference's utils/camera.py.

Everything here is small host-side torch math on [B,3,4] poses / [B,R,3] rays and stays
differentiable (the viewpoint estimator is trained through it).  The one structural difference to
the reference: rays are built only for the pixels that are actually rendered
(``get_center_and_ray(..., ray_idx=...)``) instead of materialising all H*W rays and gathering
512 of them afterwards (reference model/renderer.py:59-68, utils/camera.py:157-196).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as torch_F


class Pose:
    """[R|t] 3x4 pose helper (reference utils/camera.py:5-46)."""

    def __call__(self, R=None, t=None):
        assert R is not None or t is not None
        if R is not None and not isinstance(R, torch.Tensor):
            R = torch.tensor(R)
        if t is not None and not isinstance(t, torch.Tensor):
            t = torch.tensor(t)
        if R is None:
            R = torch.eye(3, device=t.device).repeat(*t.shape[:-1], 1, 1)
        if t is None:
            t = torch.zeros(R.shape[:-1], device=R.device)
        assert R.shape[:-1] == t.shape and R.shape[-2:] == (3, 3)
        out = torch.cat([R.float(), t.float()[..., None]], dim=-1)
        assert out.shape[-2:] == (3, 4)
        return out

    def invert(self, pose, use_inverse=False):
        R, t = pose[..., :3], pose[..., 3:]
        R_inv = inverse3x3(R) if use_inverse else R.transpose(-1, -2)
        return self(R=R_inv, t=(-R_inv @ t)[..., 0])

    def compose_pair(self, pose_a, pose_b):
        """x -> pose_b(pose_a(x))"""
        R_a, t_a = pose_a[..., :3], pose_a[..., 3:]
        R_b, t_b = pose_b[..., :3], pose_b[..., 3:]
        return self(R=R_b @ R_a, t=(R_b @ t_a + t_b)[..., 0])

    def compose(self, pose_list):
        out = pose_list[0]
        for nxt in pose_list[1:]:
            out = self.compose_pair(out, nxt)
        return out


pose = Pose()


def inverse3x3(M: torch.Tensor) -> torch.Tensor:
    """Closed-form (adjugate) inverse of [...,3,3]; avoids a LAPACK/rocSOLVER dependency for 3x3s."""
    a, b, c = M[..., 0, 0], M[..., 0, 1], M[..., 0, 2]
    d, e, f = M[..., 1, 0], M[..., 1, 1], M[..., 1, 2]
    g, h, i = M[..., 2, 0], M[..., 2, 1], M[..., 2, 2]
    A, Bc, C = e * i - f * h, -(d * i - f * g), d * h - e * g
    det = a * A + b * Bc + c * C
    adj = torch.stack([
        torch.stack([A, -(b * i - c * h), b * f - c * e], dim=-1),
        torch.stack([Bc, a * i - c * g, -(a * f - c * d)], dim=-1),
        torch.stack([C, -(a * h - b * g), a * e - b * d], dim=-1)], dim=-2)
    return adj / det[..., None, None]


def to_hom(X):
    return torch.cat([X, torch.ones_like(X[..., :1])], dim=-1)


def world2cam(X, pose_):
    return to_hom(X) @ pose_.transpose(-1, -2)


def cam2img(X, cam_intr):
    return X @ cam_intr.transpose(-1, -2)


def img2cam(X, cam_intr):
    return X @ inverse3x3(cam_intr).transpose(-1, -2)


def cam2world(X, pose_):
    return to_hom(X) @ pose.invert(pose_).transpose(-1, -2)


def get_transformed_grid(opt, points_3D, pose_, pose_gt):
    cam = world2cam(points_3D, pose_gt.unsqueeze(1).unsqueeze(1))
    return cam2world(cam, pose_.unsqueeze(1).unsqueeze(1))


def transform_normal(normals, pose_):
    """Rotate camera-frame normals into the canonical frame (reference utils/camera.py:98-103)."""
    rot = pose_[:, :, :3]
    zero_t = torch.zeros(rot.shape[0], 3, 1, device=rot.device, dtype=rot.dtype)
    return cam2world(normals, torch.cat([rot, zero_t], dim=-1))


def _cos_sin(angle, representation):
    if representation == "trig":
        return angle[:, 0], angle[:, 1]
    if representation == "angle":
        angle = angle * np.pi / 180
    elif representation != "rad":
        raise ValueError(representation)
    return torch.cos(angle), torch.sin(angle)


def _rotation_from_rows(n, device, rows):
    """Identity [n,3,3] with selected (row, col-slice) entries replaced (keeps autograd)."""
    R = torch.eye(3, device=device)[None].repeat(n, 1, 1)
    for (r, cols), vals in rows:
        R[:, r, cols] = torch.stack(vals, dim=-1)
    return R


def azim_to_rotation_matrix(azim, representation="rad"):
    """Rotation in the XZ plane (reference utils/camera.py:105-122)."""
    c, s = _cos_sin(azim, representation)
    z = torch.zeros_like(c)
    return _rotation_from_rows(len(c), c.device, [((0, slice(0, 3)), [c, z, s]), ((2, slice(0, 3)), [-s, z, c])])


def elev_to_rotation_matrix(elev, representation="rad"):
    """Rotation in the YZ plane (reference utils/camera.py:124-139)."""
    c, s = _cos_sin(elev, representation)
    return _rotation_from_rows(len(c), c.device, [((1, slice(1, 3)), [c, -s]), ((2, slice(1, 3)), [s, c])])


def roll_to_rotation_matrix(roll, representation="rad"):
    """Rotation in the XY plane (reference utils/camera.py:141-155)."""
    c, s = _cos_sin(roll, representation)
    return _rotation_from_rows(len(c), c.device, [((0, slice(0, 2)), [c, s]), ((1, slice(0, 2)), [-s, c])])


def pose_from_azim_elev(azim, elev):
    """Look-at rotation from (cos,sin) azimuth / elevation (reference utils/camera.py:53-73)."""
    cam = torch.stack([azim[:, 0] * elev[:, 0], azim[:, 1] * elev[:, 0], elev[:, 1]], dim=-1)
    fwd = -cam
    down = torch.tensor([[0.0, 0.0, -1.0]], device=azim.device).expand_as(fwd)
    right = torch_F.normalize(torch.cross(down, fwd, dim=-1), dim=-1, p=2)
    up = torch_F.normalize(torch.cross(fwd, right, dim=-1), dim=-1, p=2)
    return torch.stack([right, up, fwd], dim=-1).permute(0, 2, 1).contiguous()


def pixel_centers(opt, batch_size, device, ray_idx=None):
    """Image-plane coordinates [B,R,2] of the rendered pixels (centre = index + 0.5)."""
    if ray_idx is None:
        idx = torch.arange(opt.H * opt.W, device=device)[None].expand(batch_size, -1)
    else:
        idx = ray_idx
    if opt.camera.model == "perspective":
        x = (idx % opt.W).float() + 0.5
        y = torch.div(idx, opt.W, rounding_mode="floor").float() + 0.5
    elif opt.camera.model == "orthographic":
        assert opt.H == opt.W
        lin = torch.linspace(-1, 1, opt.H, device=device)
        x = lin[idx % opt.W]
        y = lin[torch.div(idx, opt.W, rounding_mode="floor")]
    else:
        raise NotImplementedError(opt.camera.model)
    return torch.stack([x, y], dim=-1)


def get_camera_grid(opt, batch_size, device, intr=None, ray_idx=None):
    xy = pixel_centers(opt, batch_size, device, ray_idx)
    grid = img2cam(to_hom(xy), intr) if opt.camera.model == "perspective" else to_hom(xy)
    return xy, grid


def get_center_and_ray(opt, pose_, intr=None, offset=None, device=None, ray_idx=None):
    """Camera centre [B,1|R,3] and un-normalised ray [B,R,3] in world space
    (reference utils/camera.py:177-196), restricted to `ray_idx` when given."""
    if device is None:
        device = pose_.device
    B = len(pose_)
    xy, grid = get_camera_grid(opt, B, device, intr=intr, ray_idx=ray_idx)
    if opt.camera.model == "perspective":
        if offset is not None:
            grid = torch.cat([grid[..., :2] + offset, grid[..., 2:]], dim=-1)
        center = torch.zeros(B, 1, 3, device=device)
    else:
        center = torch.cat([xy, torch.zeros_like(xy[..., :1])], dim=-1)
    grid_w = cam2world(grid, pose_)
    center_w = cam2world(center, pose_)
    return center_w, grid_w - center_w


def get_intr(opt, scale_focal):
    """K = [[fW,0,W/2],[0,fH,H/2],[0,0,1]], f = camera.focal * scale (reference utils/camera.py:198-211)."""
    z, o = torch.zeros_like(scale_focal), torch.ones_like(scale_focal)
    f = opt.camera.focal * scale_focal
    rows = [f * opt.W, z, o * opt.W / 2, z, f * opt.H, o * opt.H / 2, z, z, o]
    return torch.stack(rows, dim=-1).view(scale_focal.shape[0], 3, 3).contiguous()
