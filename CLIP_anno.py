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
        cfg = dict(cfg, dtype=str(opt.get("clip_dtype", "fp16")))       # fp16: the reference's arithmetic on a GPU (:16); or bf16
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
        """features [N,D] L2-normalised -> (indices [N,k], values [N,k]); row i starts with i itself
        (reference CLIP_anno.py:29-57).

        Default (opt.thres is None): ONE similarity GEMM + top-k instead of the reference's per-query Python loop.
        Thresholded branch (:43-53, off in options/clip/pix3d.yaml): k-1 random neighbours among cos >= thres; it keeps
        the reference's per-query operator sequence -- (query * features).sum(1), the `< 1.` test that is meant to drop
        the query itself (a query whose self-similarity rounds to 0.99999994 stays a candidate, as in the reference) and
        one CPU-generator randperm per qualifying query, in query order -- so a seeded run picks the same neighbours."""
        if opt.thres is None:
            values, indices = (features @ features.t()).topk(k_nearest, dim=1, largest=True)
            return indices, values
        indices, values = [], []
        for i in range(features.shape[0]):
            cos_sim = (features[i].unsqueeze(0) * features).sum(dim=1)
            index = ((cos_sim >= opt.thres) & (cos_sim < 1.)).nonzero()
            n_valid = len(index)
            if n_valid < k_nearest - 1:
                top_k_val, top_k_ind = cos_sim.topk(k_nearest, largest=True)
                indices.append(top_k_ind)
                values.append(top_k_val)
            else:
                sampled = index[torch.randperm(n_valid)[:k_nearest - 1].to(index.device)].squeeze(1)
                chosen = torch.cat([torch.tensor([i], device=features.device), sampled], dim=0)
                indices.append(chosen)
                values.append(cos_sim[chosen])
        return torch.stack(indices, dim=0), torch.stack(values, dim=0)

    def label2path(self, root, label):
        return os.path.join(root, label), None

    def save_anno(self, opt, label2path, labels, index_topk, value_topk, k_nearest=6, category_set="all"):
        """`<anno_root>/<category>_<split>.csv`, header Query,Top_1..,Top_1_score.., rows sorted by query
        (reference CLIP_anno.py:98-127; read back by data/pix3d.py:95-108).  self.split names the split."""
        category = opt.data[opt.data.dataset].cat.replace(", ", "_") if category_set == "custom" else category_set
        os.makedirs(opt.anno_root, exist_ok=True)
        path = os.path.join(opt.anno_root, "{}_{}.csv".format(category, self.split))
        rows = []
        for i, label in enumerate(labels):
            rows.append([label2path("", label)[0]] + [label2path("", labels[int(j)])[0] for j in index_topk[i][1:]]
                        + ["{:.4f}".format(v) for v in value_topk[i][1:]])
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
        feats = torch.cat(feats, dim=0)
        # the small-batch tower reports a cluster-barrier time-out (its members were not co-resident: another kernel held CUs) by poisoning
        # the class-token row, i.e. as NaN embeddings (csrc/clip_cluster.hpp); a nearest-neighbour table must never be built from those
        if not bool(torch.isfinite(feats).all()):
            bad = (~torch.isfinite(feats).all(dim=1)).nonzero().flatten().tolist()
            raise RuntimeError("CLIP tower produced non-finite embeddings for %d image(s) (first: %s): a cluster-barrier time-out of the small-batch "
                               "kernel or non-finite inputs; re-run with SC_CLIP_CLUSTER_MAX_B=0 to take the launch-per-operation form" % (len(bad), bad[:5]))
        return feats


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
        ann.split = split
        custom = "custom" if opt.data.dataset in opt.data else "all"
        print("wrote", ann.save_anno(opt, ann.label2path, labels, idx.cpu(), val.cpu(), k_nearest=opt.k_nearest, category_set=custom))


if __name__ == "__main__":
    main()
