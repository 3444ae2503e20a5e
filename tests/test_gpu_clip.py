"""CLIP ViT image tower (csrc/clip_vit.hip) vs transformers.CLIPVisionModelWithProjection (fp32, CPU) with
seeded random weights -- architecture-level oracle; parity with the reference is unpinned (openai/CLIP is an
un-vendored dependency of CLIP_anno.py:16 and no weights are available offline).

Bar: bf16 GEMM inputs, fp32 accumulate: cosine similarity of embeddings > 0.999, max abs error < 3 % of max |ref|."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hf(width, layers, heads, mlp, patch, image, proj, seed):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(seed)
    cfg = CLIPVisionConfig(hidden_size=width, intermediate_size=mlp, num_hidden_layers=layers, num_attention_heads=heads,
                           patch_size=patch, image_size=image, projection_dim=proj, hidden_act="quick_gelu")
    m = CLIPVisionModelWithProjection(cfg).eval()
    with torch.no_grad():      # make biases / LN params non-trivial
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            else:
                p.mul_(3.0)
    return m


@pytest.mark.parametrize("width,layers,heads,mlp,patch,image,proj,B", [
    (128, 2, 2, 256, 32, 64, 64, 3),             # 5 tokens
    (768, 12, 12, 3072, 32, 224, 512, 4),        # ViT-B/32: 50 tokens
    (768, 12, 12, 3072, 32, 224, 512, 32),       # ViT-B/32 at the BASELINE config[2] batch (1600 token rows: split-K proj / fc2)
    (128, 2, 2, 256, 14, 112, 64, 2),            # 65 tokens (> one key chunk... and 2 query blocks + 1), patch K = 588 (padded)
    (192, 2, 3, 384, 8, 112, 96, 2),             # 197 tokens: 7 key chunks, query blocks wrap over the 4 waves
    (1024, 2, 16, 4096, 14, 224, 768, 2),        # ViT-L/14 geometry (257 tokens), 2 layers
])
def test_clip_tower_vs_transformers(width, layers, heads, mlp, patch, image, proj, B):
    from shapeclipper_amd.model.clip_vit import ClipVisionTower
    hf = _hf(width, layers, heads, mlp, patch, image, proj, seed=width)
    x = torch.randn(B, 3, image, image)
    with torch.no_grad():
        ref = hf(pixel_values=x).image_embeds
    tower = ClipVisionTower(image_size=image, patch=patch, width=width, layers=layers, heads=heads, mlp=mlp, proj=proj)
    tower.load_state_dict(hf.state_dict())
    tower = tower.cuda()
    got = tower.encode_image(x.cuda()).cpu()
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=-1)
    assert cos.min().item() > 0.999, cos
    assert (got - ref).abs().max().item() < 0.03 * ref.abs().max().item()


def test_gemm_bf16_transpose_detecting():
    """C = A W^T with asymmetric operands (catches swapped row/col in the MFMA C/D mapping)."""
    import ctypes
    from shapeclipper_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    M, N, K = 200, 192, 128
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) + torch.arange(N, device=dev)[:, None] * 0.01).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev)
    rc = lib.sc_gemm_bf16(ctypes.c_int(0), _lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), ctypes.c_int(M),
                          ctypes.c_int(N), ctypes.c_int(K), _lib.stream())
    assert rc == 0
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t() + bias
    assert (out - ref).abs().max().item() < 2e-3 * ref.abs().max().item()


def test_clip_tower_golden_fixture():
    """G11 (tests/golden/make_golden_clip.py): committed weights / input / embedding of a shrunken tower."""
    import os
    from shapeclipper_amd.model.clip_vit import ClipVisionTower
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_clip_arch.npz"))
    width, layers, heads, mlp, patch, image, proj = (int(v) for v in g["cfg"])
    tower = ClipVisionTower(image_size=image, patch=patch, width=width, layers=layers, heads=heads, mlp=mlp, proj=proj)
    tower.load_state_dict({k[2:]: torch.tensor(g[k]) for k in g.files if k.startswith("w.")})
    got = tower.cuda().encode_image(torch.tensor(g["input"]).cuda()).cpu()
    ref = torch.tensor(g["embedding"])
    assert torch.nn.functional.cosine_similarity(got, ref, dim=-1).min().item() > 0.999
    assert (got - ref).abs().max().item() < 0.03 * ref.abs().max().item()
