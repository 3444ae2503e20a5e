"""Synthetic Pix3D-shaped batches: the dataset, its CLIP-NN CSV and the pretrained
encoders are not available offline, so benchmarks / smoke tests / multi-process tests use random
images with disc masks in exactly the batch-dict schema of the reference's data/pix3d.py:110-228."""
from __future__ import annotations

import threading

import numpy as np
import torch

from .utils.util import EasyDict as edict


def _disc_masks(n, H, W, gen):
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    rad = 40 + 50 * torch.rand(n, generator=gen)
    rad = rad * (H / 224.0)
    d2 = (yy - H / 2) ** 2 + (xx - W / 2) ** 2
    return (d2[None] <= (rad ** 2)[:, None, None]).float().unsqueeze(1)        # [n,1,H,W]


def _views(n, H, W, R, gen, opt=None, importance=False):
    rgb = torch.rand(n, 3, H, W, generator=gen)
    mask = _disc_masks(n, H, W, gen)
    normal = torch.nn.functional.normalize(torch.randn(n, 3, H, W, generator=gen), dim=1) * mask
    out = dict(rgb_input_map=rgb, mask_input_map=mask, normal_input_map=normal)
    if R:
        if importance:
            # the reference's training loader draws the rays around the silhouette (data/pix3d.py:230-240 ->
            # utils/util.py:237-248; numpy global RNG): same call, same arguments
            from .utils.util import compute_sampling_prob
            ray_idx = torch.stack([compute_sampling_prob(opt, mask[i, 0], opt.render.ray_uniform_fac) for i in range(n)], 0).long()
        else:
            ray_idx = torch.stack([torch.randperm(H * W, generator=gen)[:R] for _ in range(n)], 0)
        take = lambda m: m.flatten(2).permute(0, 2, 1).gather(1, ray_idx[..., None].expand(-1, -1, m.shape[1]))
        out.update(ray_idx=ray_idx, rgb_input=take(rgb), mask_input=take(mask), normal_input=take(normal))
    else:
        flat = lambda m: m.flatten(2).permute(0, 2, 1).contiguous()
        out.update(rgb_input=flat(rgb), mask_input=flat(mask), normal_input=flat(normal))
    return out


_CAP_LOCK = threading.Lock()


def cap_host_threads(limit=8):
    """Cap torch's intra-op CPU thread pool once, from the main thread of an entry point (values never depend on the thread count:
    element-wise operators and generator draws only).  Returns the thread count in effect."""
    with _CAP_LOCK:
        if threading.current_thread() is threading.main_thread() and torch.get_num_threads() > limit:
            torch.set_num_threads(limit)
    return torch.get_num_threads()


def make_batch(opt, batch_size, seed=0, training=True, n_gt_points=2048, importance=False):
    """One batch with the reference's keys; neighbour stacks carry a trailing K dimension.
    importance=True: ray_idx from the reference's silhouette importance sampler instead of a uniform permutation
    (needs opt.H == image height, consumes numpy's global RNG like the reference's loader)."""
    # The generator is a few dozen CPU operators on image-sized tensors; with the 256 host threads of a GPU box every one of them pays
    # the wake-up of an idle thread pool (24 ms per `norm`: 0.33 s per evaluation sample, tools/prof_eval_host.py).  The intra-op
    # thread count is process-global, so it is capped ONCE per process by the entry points (`cap_host_threads()`: cli.py, bench.py),
    # never switched back and forth here (a DataLoader / autograd thread would see the change, racing restores could stick).
    gen = torch.Generator().manual_seed(seed)
    H, W = opt.image_size
    R = opt.render.rand_sample if training else 0
    K = opt.data.k_nearest
    views = lambda: _views(batch_size, H, W, R, gen, opt, importance)
    v = views()
    batch = edict(idx=torch.arange(batch_size), category_label=torch.zeros(batch_size, dtype=torch.long), **v)
    azim = (torch.rand(batch_size, generator=gen) * 2 - 1) * np.pi
    elev = (torch.rand(batch_size, generator=gen) * 2 - 1) * np.pi / 6
    ca, sa, ce, se = torch.cos(azim), torch.sin(azim), torch.cos(elev), torch.sin(elev)
    Rm = torch.zeros(batch_size, 3, 3)
    Rm[:, 0, 0], Rm[:, 0, 2], Rm[:, 1, 1], Rm[:, 2, 0], Rm[:, 2, 2] = ca, sa, 1.0, -sa, ca
    Rx = torch.zeros(batch_size, 3, 3)
    Rx[:, 0, 0], Rx[:, 1, 1], Rx[:, 1, 2], Rx[:, 2, 1], Rx[:, 2, 2] = 1.0, ce, -se, se, ce
    pose = torch.cat([Rx @ Rm, torch.tensor([0.0, 0.0, 5.0]).expand(batch_size, 3)[..., None]], dim=-1)
    f = 4.0
    intr = torch.tensor([[f * W, 0, W / 2], [0, f * H, H / 2], [0, 0, 1]]).expand(batch_size, 3, 3).contiguous()
    batch.pose_gt, batch.intr = pose, intr
    pts = torch.rand(batch_size, n_gt_points, 3, generator=gen) - 0.5
    batch.dpc = edict(points=pts, normals=torch.nn.functional.normalize(torch.randn(batch_size, n_gt_points, 3, generator=gen), dim=-1))
    if training:
        stacks = [views() for _ in range(K)]
        for key in ("rgb_input", "mask_input", "normal_input", "rgb_input_map", "mask_input_map", "normal_input_map", "ray_idx"):
            batch[key + "_NN"] = torch.stack([s[key] for s in stacks], dim=-1)
        batch.pose_gt_NN = pose[..., None].expand(-1, -1, -1, K).contiguous()
    return batch
