"""Minimal image dumps (PNG via PIL when available).  Visualisation is out of the hot path; the
reference's TensorBoard grids / GIF / PLY writers (utils/util_vis.py) are not reproduced."""
import os

import numpy as np
import torch

try:
    from PIL import Image
except Exception:  # pragma: no cover
    Image = None


@torch.no_grad()
def dump_images(opt, idx, name, images, masks=None, from_range=(0, 1), poses=None, folder="dump"):
    if Image is None:
        return
    lo, hi = from_range
    imgs = ((images - lo) / (hi - lo)).clamp(0, 1)
    if masks is not None:
        imgs = imgs * masks + (1 - masks)
    imgs = (imgs.cpu().permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)
    for i, img in zip(idx, imgs):
        arr = img[..., 0] if img.shape[-1] == 1 else img
        Image.fromarray(arr).save("{}/{}/{}_{}.png".format(opt.output_path, folder, int(i), name))
