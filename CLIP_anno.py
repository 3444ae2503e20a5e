"""python CLIP_anno.py --yaml=options/clip/pix3d.yaml [--clip_ckpt=<state dict>] [--data.dataset=synthetic]

Offline CLIP nearest-neighbour annotation (reference CLIP_anno.py): embed every image of a split with
the CLIP image tower, L2-normalise, cosine k-NN, write `<anno_root>/<cat>_<split>.csv` with the header
Query,Top_1..Top_{k-1},Top_1_score..  sorted by query (same on-disk format; data/pix3d.py:95-108 reads it).

MI355X build: the tower runs on the HIP kernels (shapeclipper_amd/model/clip_vit.py; --clip_model=ViT-B/32
(default, BASELINE config[2]) or ViT-L/14 (the model the reference loads, CLIP_anno.py:16)); the O(N^2)
similarity is ONE GEMM + top-k instead of a Python loop (reference :29-57).  Weights: pass --clip_ckpt (a transformers CLIPVisionModelWithProjection or
openai/CLIP state dict); without it the tower is randomly initialised (no network here)."""
import csv
import importlib
import os
import sys

import torch
import torch.nn.functional as torch_F

import utils.options as options
from utils.util import log
from shapeclipper_amd.model.clip_vit import VIT_B32, VIT_L14, ClipVisionTower


class NN_annotator:
    def __init__(self, opt):
        name = str(opt.get("clip_model", "ViT-B/32"))
        if name not in ("ViT-B/32", "ViT-L/14"):
            raise NotImplementedError("clip_model '%s' (available: ViT-B/32, ViT-L/14)" % name)
        cfg = VIT_L14 if name == "ViT-L/14" else VIT_B32
        self.tower = ClipVisionTower(**cfg)
        ckpt = opt.get("clip_ckpt", None)
        if ckpt:
            sd = torch.load(ckpt, map_location="cpu")
            if any(k.startswith("visual.") for k in sd):
                self.tower = ClipVisionTower.from_openai_state_dict(sd, **cfg)
            else:
                self.tower.load_state_dict(sd)
        self.tower = self.tower.to(opt.device)
        self.clip_dim = cfg["proj"]

    @torch.no_grad()
    def calc_matches(self, opt, features, k_nearest=6):
        """features [N,D] L2-normalised -> (indices [N,k], values [N,k]); row i starts with i itself."""
        sim = features @ features.t()
        if opt.thres is None:
            values, indices = sim.topk(k_nearest, dim=1, largest=True)
            return indices, values
        # thresholded random neighbours (reference :43-53); falls back to top-k when too few pass
        N = features.shape[0]
        values, indices = sim.topk(k_nearest, dim=1, largest=True)
        for i in range(N):
            ok = ((sim[i] >= opt.thres) & (sim[i] < 1.)).nonzero().squeeze(1)
            if len(ok) >= k_nearest - 1:
                pick = ok[torch.randperm(len(ok), device=ok.device)[:k_nearest - 1]]
                idx = torch.cat([torch.tensor([i], device=ok.device), pick])
                indices[i], values[i] = idx, sim[i][idx]
        return indices, values

    def save_anno(self, opt, labels, index_topk, value_topk, split, k_nearest=6):
        cat = opt.data[opt.data.dataset].cat.replace(", ", "_") if opt.data.dataset in opt.data else "all"
        os.makedirs(opt.anno_root, exist_ok=True)
        path = os.path.join(opt.anno_root, "{}_{}.csv".format(cat, split))
        rows = []
        for i, label in enumerate(labels):
            rows.append([label] + [labels[j] for j in index_topk[i][1:].tolist()]
                        + ["{:.4f}".format(v) for v in value_topk[i][1:].tolist()])
        header = ["Query"] + ["Top_{}".format(i) for i in range(1, k_nearest)] + ["Top_{}_score".format(i) for i in range(1, k_nearest)]
        with open(path, "w") as f:
            w = csv.writer(f)
            w.writerow(header)
            w.writerows(sorted(rows, key=lambda r: r[0]))
        return path

    @torch.no_grad()
    def embed_split(self, opt, images):
        feats = []
        for i in range(0, len(images), opt.batch_size):
            e = self.tower.encode_image(images[i:i + opt.batch_size].to(opt.device)).float()
            feats.append(torch_F.normalize(e, dim=-1))
        return torch.cat(feats, dim=0)


def main():
    log.process(os.getpid())
    log.title("[{}] (compute CLIP-NN)".format(sys.argv[0]))
    opt = options.set(opt_cmd=options.parse_arguments(sys.argv[1:]))
    ann = NN_annotator(opt)
    if opt.data.dataset == "synthetic":
        n = int(opt.data.get("synthetic_len", 256))
        gen = torch.Generator().manual_seed(0)
        images = torch.randn(n, 3, 224, 224, generator=gen)
        labels = ["synthetic/img_{:05d}.png".format(i) for i in range(n)]
        splits = {"train": (images, labels)}
    else:
        raise NotImplementedError("the Pix3D image loader is out of scope of this build (SURVEY 2.1); use --data.dataset=synthetic")
    for split, (images, labels) in splits.items():
        feats = ann.embed_split(opt, images)
        idx, val = ann.calc_matches(opt, feats, k_nearest=opt.k_nearest)
        print("wrote", ann.save_anno(opt, labels, idx.cpu(), val.cpu(), split, k_nearest=opt.k_nearest))


if __name__ == "__main__":
    main()
