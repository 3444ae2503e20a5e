"""G11: architecture vectors for the CLIP ViT image tower.

The reference calls the un-vendored, un-pinned openai/CLIP package (CLIP_anno.py:16,166) and no weights are available
offline -> parity with the reference is UNPINNED.  What can be frozen is the architecture: a shrunken
transformers.CLIPVisionModelWithProjection (same block structure: pre-LN, quick_gelu, class token, ln_post + projection)
with seeded random weights, one input batch and its embedding.  Run in the build container:
    python tests/golden/make_golden_clip.py
"""
import os

import numpy as np
import torch
from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(width=64, layers=2, heads=1, mlp=128, patch=16, image=64, proj=32)


def main():
    torch.manual_seed(11)
    cfg = CLIPVisionConfig(hidden_size=CFG["width"], intermediate_size=CFG["mlp"], num_hidden_layers=CFG["layers"],
                           num_attention_heads=CFG["heads"], patch_size=CFG["patch"], image_size=CFG["image"],
                           projection_dim=CFG["proj"], hidden_act="quick_gelu")
    m = CLIPVisionModelWithProjection(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():      # non-trivial biases / LayerNorm parameters, O(1) activations
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            else:
                p.mul_(3.0)
        x = torch.randn(2, 3, CFG["image"], CFG["image"])
        y = m(pixel_values=x).image_embeds
    out = {"input": x.numpy(), "embedding": y.numpy(), "cfg": np.array([CFG[k] for k in ("width", "layers", "heads", "mlp", "patch", "image", "proj")])}
    for k, v in m.state_dict().items():
        if v.dtype.is_floating_point:
            out["w." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "g11_clip_arch.npz"), **out)
    print("wrote g11_clip_arch.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB; |embedding| max", float(y.abs().max()))


if __name__ == "__main__":
    main()
