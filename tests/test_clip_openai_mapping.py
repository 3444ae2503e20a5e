"""`ClipVisionTower.from_openai_state_dict` (the entry point for REAL openai/CLIP checkpoints: CLIP_anno.py:16 `clip.load`) on CPU:
a state dict in the original package's naming (oracle/clip_openai_ref.py: torch's own nn.MultiheadAttention with its packed
in_proj_weight) is mapped onto the tower's transformers-style names; loaded into transformers.CLIPVisionModelWithProjection those
parameters must reproduce the original-architecture forward.  Pins the q / k / v split, `visual.proj.t()`, the class / position
embeddings and the geometry inference without a GPU.  (The HIP tower on the same state dict: tests/test_gpu_clip.py.)"""
import pytest
import torch

from oracle import clip_openai_ref as O
from shapeclipper_amd.model.clip_vit import ClipVisionTower, geometry_from_openai_state_dict


@pytest.mark.parametrize("res,patch,width,layers,proj", [(64, 32, 128, 2, 64), (112, 14, 192, 3, 96)])
def test_openai_names_map_onto_transformers_tower(res, patch, width, layers, proj):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    ref, sd = O.seeded(res, patch, width, layers, proj, seed=3)
    geo = geometry_from_openai_state_dict(sd)
    assert geo == dict(image_size=res, patch=patch, width=width, layers=layers, heads=width // 64, mlp=4 * width, proj=proj)
    tower = ClipVisionTower.from_openai_state_dict(sd)                   # geometry inferred, as clip.build_model does
    assert {k: tower.cfg[k] for k in geo} == geo
    hf = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers,
                                                        num_attention_heads=width // 64, patch_size=patch, image_size=res,
                                                        projection_dim=proj, hidden_act="quick_gelu")).eval()
    missing, unexpected = hf.load_state_dict(tower.state_dict(), strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    x = torch.randn(3, 3, res, res)
    with torch.no_grad():
        a, b = ref.encode_image(x), hf(pixel_values=x).image_embeds
    assert (a - b).abs().max() < 2e-5 * a.abs().max(), (a - b).abs().max()


def test_fp16_checkpoint_is_widened_exactly():
    """clip.load(..., device='cuda') returns fp16 weights: the mapping widens them to fp32 without touching a bit."""
    _, sd = O.seeded(64, 32, 128, 1, 64, seed=5, fp16_storage=True)
    assert sd["visual.proj"].dtype == torch.float16 and sd["visual.ln_pre.weight"].dtype == torch.float32
    t = ClipVisionTower.from_openai_state_dict(sd).state_dict()
    assert t["visual_projection.weight"].dtype == torch.float32
    assert torch.equal(t["visual_projection.weight"], sd["visual.proj"].float().t())
    w = sd["visual.transformer.resblocks.0.attn.in_proj_weight"].float()
    assert torch.equal(t["vision_model.encoder.layers.0.self_attn.k_proj.weight"], w[128:256])
    assert torch.equal(t["vision_model.encoder.layers.0.self_attn.v_proj.bias"], sd["visual.transformer.resblocks.0.attn.in_proj_bias"].float()[256:])


def test_mapping_rejects_a_text_only_or_resnet_checkpoint():
    with pytest.raises(KeyError):
        geometry_from_openai_state_dict({"visual.layer1.0.conv1.weight": torch.zeros(1)})
