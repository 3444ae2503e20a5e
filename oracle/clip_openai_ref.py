"""TEST INFRASTRUCTURE ONLY (tests/ import this; the product never does).

Architecture oracle of the image tower of openai/CLIP in the ORIGINAL package's own parameter naming -- the dependency the
reference loads with `clip.load("ViT-L/14", device=device)` and calls as `clip_encoder.encode_image(image)` (CLIP_anno.py:16,
166-167).  openai/CLIP is un-vendored and un-pinned (`pip install git+https://github.com/openai/CLIP.git`, README.md:14,24):
its code is not in /root/reference and no weights exist offline, so **parity unpinned**.  What this file restates is the
published architecture of `clip/model.py` (VisionTransformer / ResidualAttentionBlock / QuickGELU) in plain torch modules
whose `state_dict()` carries exactly the key names of `clip.load(...).state_dict()`:

    visual.conv1.weight  visual.class_embedding  visual.positional_embedding  visual.ln_pre.{weight,bias}
    visual.transformer.resblocks.N.{attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.{weight,bias},
                                    ln_1.{weight,bias}, ln_2.{weight,bias}, mlp.c_fc.{weight,bias}, mlp.c_proj.{weight,bias}}
    visual.ln_post.{weight,bias}  visual.proj

The attention is torch's own nn.MultiheadAttention (as in the package), so the packed [q; k; v] layout of `in_proj_weight` that
`ClipVisionTower.from_openai_state_dict` has to split is torch's, not this file's.  Used by tests/test_clip_openai_mapping.py
(CPU: mapping == transformers' tower) and tests/test_gpu_clip.py (HIP tower loaded through from_openai_state_dict).
"""
from collections import OrderedDict

import torch
import torch.nn as nn


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)            # sequence-first, packed in_proj_weight [3 d, d]
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = nn.LayerNorm(d_model)

    def forward(self, x):
        y = self.ln_1(x)
        x = x + self.attn(y, y, y, need_weights=False)[0]
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))          # [width, output_dim]: applied as x @ proj

    def forward(self, x):
        x = self.conv1(x)                                                # [B, width, g, g]
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)       # [B, g*g, width]
        cls = self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(x)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)        # sequence-first inside the transformer
        return self.ln_post(x[:, 0, :]) @ self.proj


class ClipImageSide(nn.Module):
    """`visual` + the `encode_image` call; state_dict() keys start with `visual.` as the original package's do."""

    def __init__(self, input_resolution=224, patch_size=32, width=768, layers=12, output_dim=512):
        super().__init__()
        self.visual = VisionTransformer(input_resolution, patch_size, width, layers, width // 64, output_dim)

    def encode_image(self, image):
        return self.visual(image)


def seeded(input_resolution, patch_size, width, layers, output_dim, seed, fp16_storage=False):
    """A randomly initialised tower with non-trivial biases / LayerNorm parameters.  fp16_storage: weights rounded to fp16 the
    way `clip.load(..., device='cuda')` hands them out (convert_weights: conv / linear / in_proj / proj in fp16, LayerNorm fp32)."""
    torch.manual_seed(seed)
    m = ClipImageSide(input_resolution, patch_size, width, layers, output_dim).eval()
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            else:
                p.mul_(1.5)
    sd = m.state_dict()
    if fp16_storage:
        sd = {k: (v.half() if (v.dim() > 1 or "in_proj_bias" in k or ".attn." in k or ".mlp." in k or k.endswith("class_embedding")) and ".ln_" not in k
                  else v) for k, v in sd.items()}
    return m, sd
