"""CLIP ViT image tower on the HIP kernels (csrc/clip_vit.hip).

Drop-in for the one call the reference makes into the un-vendored openai/CLIP package:
`clip_encoder.encode_image(image).float()` (CLIP_anno.py:166).  Parameters carry the
transformers.CLIPVisionModelWithProjection names so a converted checkpoint loads with
load_state_dict; `from_openai_state_dict` maps the original openai/CLIP `visual.*` names.
Compute: 16-bit MFMA GEMMs with fp32 accumulation, fp32 LayerNorm / softmax statistics / residual stream.
`dtype="fp16"` (default): IEEE fp16 operands, the arithmetic openai/CLIP itself uses on a GPU (fp16 weights and
activations with fp32 LayerNorm, CLIP_anno.py:16); `dtype="bf16"`: bf16 operands (8-bit mantissa, fp32's range)."""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from .. import _lib

VIT_B32 = dict(image_size=224, patch=32, width=768, layers=12, heads=12, mlp=3072, proj=512)
VIT_L14 = dict(image_size=224, patch=14, width=1024, layers=24, heads=16, mlp=4096, proj=768)   # the reference's model (CLIP_anno.py:16)


class ClipVisionTower(nn.Module):

    def __init__(self, image_size=224, patch=32, width=768, layers=12, heads=12, mlp=3072, proj=512, channels=3, dtype="fp16"):
        super().__init__()
        if dtype not in ("fp16", "bf16"):
            raise ValueError("ClipVisionTower dtype must be 'fp16' or 'bf16', got %r" % (dtype,))
        self.dtype16 = dtype
        self.cfg = dict(image_size=image_size, patch=patch, width=width, layers=layers, heads=heads, mlp=mlp, proj=proj,
                        channels=channels)
        T = (image_size // patch) ** 2 + 1
        assert width // heads == 64 and width % 64 == 0 and mlp % 64 == 0, "kernels need head_dim 64"
        P = lambda *s: nn.Parameter(torch.randn(*s) * 0.02)
        vm = "vision_model."
        names = {vm + "embeddings.class_embedding": P(width),
                 vm + "embeddings.patch_embedding.weight": P(width, channels, patch, patch),
                 vm + "embeddings.position_embedding.weight": P(T, width),
                 vm + "pre_layrnorm.weight": nn.Parameter(torch.ones(width)), vm + "pre_layrnorm.bias": nn.Parameter(torch.zeros(width)),
                 vm + "post_layernorm.weight": nn.Parameter(torch.ones(width)), vm + "post_layernorm.bias": nn.Parameter(torch.zeros(width)),
                 "visual_projection.weight": P(proj, width)}
        for l in range(layers):
            pre = vm + "encoder.layers.%d." % l
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                names[pre + "self_attn.%s.weight" % n] = P(width, width)
                names[pre + "self_attn.%s.bias" % n] = nn.Parameter(torch.zeros(width))
            for n in ("layer_norm1", "layer_norm2"):
                names[pre + n + ".weight"] = nn.Parameter(torch.ones(width))
                names[pre + n + ".bias"] = nn.Parameter(torch.zeros(width))
            names[pre + "mlp.fc1.weight"], names[pre + "mlp.fc1.bias"] = P(mlp, width), nn.Parameter(torch.zeros(mlp))
            names[pre + "mlp.fc2.weight"], names[pre + "mlp.fc2.bias"] = P(width, mlp), nn.Parameter(torch.zeros(width))
        self._names = list(names)
        self.params = nn.ParameterDict({k.replace(".", "/"): v for k, v in names.items()})
        self._packed = None

    # state-dict ABI = transformers' names
    def state_dict(self, *a, **k):
        return {n: self.params[n.replace(".", "/")].detach() for n in self._names}

    def load_state_dict(self, sd, strict=True):
        missing = [n for n in self._names if n not in sd]
        if strict and missing:
            raise KeyError("missing keys: %s" % missing[:5])
        with torch.no_grad():
            for n in self._names:
                if n in sd:
                    self.params[n.replace(".", "/")].copy_(sd[n])
        self._packed = None
        return missing

    def _pack(self):
        c = self.cfg
        g = lambda n: self.params[n.replace(".", "/")].detach()
        vm = "vision_model."
        w_patch = g(vm + "embeddings.patch_embedding.weight").reshape(c["width"], -1)
        pad = (-w_patch.shape[1]) % 64             # the GEMM wants K % 64 == 0 (ViT-L/14: 588 -> 640, zero columns)
        mats = [torch.nn.functional.pad(w_patch, (0, pad))]
        vecs = [g(vm + "embeddings.class_embedding"), g(vm + "embeddings.position_embedding.weight").reshape(-1),
                g(vm + "pre_layrnorm.weight"), g(vm + "pre_layrnorm.bias")]
        for l in range(c["layers"]):
            pre = vm + "encoder.layers.%d." % l
            mats += [torch.cat([g(pre + "self_attn.%s.weight" % n) for n in ("q_proj", "k_proj", "v_proj")], 0),
                     g(pre + "self_attn.out_proj.weight"), g(pre + "mlp.fc1.weight"), g(pre + "mlp.fc2.weight")]
            vecs += [g(pre + "layer_norm1.weight"), g(pre + "layer_norm1.bias"),
                     torch.cat([g(pre + "self_attn.%s.bias" % n) for n in ("q_proj", "k_proj", "v_proj")], 0),
                     g(pre + "self_attn.out_proj.bias"), g(pre + "layer_norm2.weight"), g(pre + "layer_norm2.bias"),
                     g(pre + "mlp.fc1.bias"), g(pre + "mlp.fc2.bias")]
        mats.append(g("visual_projection.weight"))
        vecs += [g(vm + "post_layernorm.weight"), g(vm + "post_layernorm.bias")]
        w_bf16 = torch.cat([m.reshape(-1) for m in mats]).to(torch.float16 if self.dtype16 == "fp16" else torch.bfloat16).contiguous()
        w_f32 = torch.cat([v.reshape(-1).float() for v in vecs]).contiguous()
        # ViT-B geometry at <= 64 tokens: the layer matrices once more in the order the small-batch kernel consumes them (csrc/clip_cluster.hpp)
        w_cluster = None
        lib = _lib.load()
        T = (c["image_size"] // c["patch"]) ** 2 + 1
        if w_bf16.is_cuda and lib.sc_clip_cluster_supported(ctypes.c_int(c["width"]), ctypes.c_int(c["mlp"]), ctypes.c_int(c["heads"]), ctypes.c_int(T)):
            lib.sc_clip_cluster_pack_elems.restype = ctypes.c_longlong
            n = int(lib.sc_clip_cluster_pack_elems(ctypes.c_int(c["layers"])))
            w_cluster = torch.empty(n, device=w_bf16.device, dtype=w_bf16.dtype)
            with torch.cuda.device(w_bf16.device):
                _lib.check(lib.sc_clip_cluster_pack(_lib.ptr(w_bf16), ctypes.c_int(mats[0].shape[1]), ctypes.c_int(c["layers"]), _lib.ptr(w_cluster),
                                                    _lib.stream()), "sc_clip_cluster_pack")
        return w_bf16, w_f32, w_cluster

    def workspace_bytes(self, B):
        c = self.cfg
        return int(_lib.load().sc_clip_vit_workspace_bytes(ctypes.c_int(B), ctypes.c_int(c["channels"]), ctypes.c_int(c["image_size"]),
                                                           ctypes.c_int(c["image_size"]), ctypes.c_int(c["patch"]),
                                                           ctypes.c_int(c["width"]), ctypes.c_int(c["mlp"])))

    @torch.no_grad()
    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        """image [B,3,H,W] (CLIP-normalised) -> embedding [B, proj] fp32 (un-normalised, as CLIP's encode_image)."""
        lib = _lib.load()
        c = self.cfg
        image = image.float().contiguous()
        B = image.shape[0]
        assert image.shape[1:] == (c["channels"], c["image_size"], c["image_size"])
        if self._packed is None or self._packed[0].device != image.device:
            self._packed = self._pack()
        w_bf16, w_f32, w_cluster = self._packed
        out = torch.empty(B, c["proj"], device=image.device, dtype=torch.float32)
        nbytes = self.workspace_bytes(B)
        ws = torch.empty(nbytes, device=image.device, dtype=torch.uint8)
        code = lib.sc_clip_vit_forward_packed(_lib.ptr(image), ctypes.c_int(B), ctypes.c_int(c["channels"]),
                                              ctypes.c_int(c["image_size"]), ctypes.c_int(c["image_size"]), ctypes.c_int(c["patch"]),
                                              ctypes.c_int(c["width"]), ctypes.c_int(c["mlp"]), ctypes.c_int(c["layers"]),
                                              ctypes.c_int(c["heads"]), ctypes.c_int(c["proj"]), _lib.ptr(w_bf16), _lib.ptr(w_f32),
                                              _lib.ptr(w_cluster) if w_cluster is not None else ctypes.c_void_p(0),
                                              ctypes.c_int(1 if self.dtype16 == "fp16" else 0),
                                              ctypes.c_float(1e-5), _lib.ptr(out), _lib.ptr(ws), ctypes.c_longlong(nbytes), _lib.stream())
        _lib.check(code, "sc_clip_vit_forward")
        return out

    forward = encode_image

    @staticmethod
    def from_openai_state_dict(sd, dtype="fp16", **cfg):
        """Map openai/CLIP `visual.*` keys (clip.load(...).state_dict(), CLIP_anno.py:16) onto this module.  Geometry is read off the
        tensor shapes (as the package's own build_model does) unless given; fp16 checkpoints are widened to fp32 parameters exactly."""
        m = ClipVisionTower(dtype=dtype, **(cfg or geometry_from_openai_state_dict(sd)))
        W = m.cfg["width"]
        out = {"vision_model.embeddings.class_embedding": sd["visual.class_embedding"],
               "vision_model.embeddings.patch_embedding.weight": sd["visual.conv1.weight"],
               "vision_model.embeddings.position_embedding.weight": sd["visual.positional_embedding"],
               "vision_model.pre_layrnorm.weight": sd["visual.ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd["visual.ln_pre.bias"],
               "vision_model.post_layernorm.weight": sd["visual.ln_post.weight"], "vision_model.post_layernorm.bias": sd["visual.ln_post.bias"],
               "visual_projection.weight": sd["visual.proj"].t()}
        for l in range(m.cfg["layers"]):
            src, dst = "visual.transformer.resblocks.%d." % l, "vision_model.encoder.layers.%d." % l
            w, b = sd[src + "attn.in_proj_weight"], sd[src + "attn.in_proj_bias"]
            for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                out[dst + "self_attn.%s.weight" % n], out[dst + "self_attn.%s.bias" % n] = w[i * W:(i + 1) * W], b[i * W:(i + 1) * W]
            out[dst + "self_attn.out_proj.weight"], out[dst + "self_attn.out_proj.bias"] = sd[src + "attn.out_proj.weight"], sd[src + "attn.out_proj.bias"]
            out[dst + "layer_norm1.weight"], out[dst + "layer_norm1.bias"] = sd[src + "ln_1.weight"], sd[src + "ln_1.bias"]
            out[dst + "layer_norm2.weight"], out[dst + "layer_norm2.bias"] = sd[src + "ln_2.weight"], sd[src + "ln_2.bias"]
            out[dst + "mlp.fc1.weight"], out[dst + "mlp.fc1.bias"] = sd[src + "mlp.c_fc.weight"], sd[src + "mlp.c_fc.bias"]
            out[dst + "mlp.fc2.weight"], out[dst + "mlp.fc2.bias"] = sd[src + "mlp.c_proj.weight"], sd[src + "mlp.c_proj.bias"]
        m.load_state_dict({k: v.float() for k, v in out.items()})
        return m


def geometry_from_openai_state_dict(sd):
    """ClipVisionTower(**geometry) of an openai/CLIP checkpoint, from its tensor shapes (ViT towers only; the package's ResNet towers
    have no `visual.proj`): width / patch from `visual.conv1.weight`, depth from the resblock count, tokens from
    `visual.positional_embedding`, heads = width / 64, the projection width from `visual.proj`."""
    if "visual.proj" not in sd or "visual.conv1.weight" not in sd:
        raise KeyError("not an openai/CLIP ViT checkpoint: no visual.proj / visual.conv1.weight")
    width, _, patch, _ = sd["visual.conv1.weight"].shape
    layers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    mlp = sd["visual.transformer.resblocks.0.mlp.c_fc.weight"].shape[0]
    return dict(image_size=int(grid * patch), patch=int(patch), width=int(width), layers=int(layers), heads=int(width // 64), mlp=int(mlp),
                proj=int(sd["visual.proj"].shape[1]))
