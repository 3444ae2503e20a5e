#!/usr/bin/env python
"""Generate tests/golden/g13_clip_nn.npz + g13_chair_train.csv from the REFERENCE's CLIP_anno.py (build container only).

The reference module is imported unmodified with its absent third-party imports stubbed (`clip` -- the openai package is
not vendored --, vigra, termcolor); only NN_annotator.calc_matches (CLIP_anno.py:29-57) and save_anno (:98-127) are
exercised, on seeded L2-normalised features: the default top-k branch, the thresholded-random-neighbour branch (CPU
generator seeded, so the randperm stream is part of the vector) and the CSV the loader reads back.  Fixtures are data.

    python tests/golden/make_golden_anno.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.environ.get("GOLDEN_OUT", HERE)

for name in ("termcolor", "vigra", "clip"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["termcolor"].colored = lambda s, **k: s
for pkg in ("model", "utils", "data"):
    for k in [k for k in sys.modules if k == pkg or k.startswith(pkg + ".")]:
        del sys.modules[k]
    m = types.ModuleType(pkg)
    m.__path__ = [os.path.join(REF, pkg)]
    sys.modules[pkg] = m
import matplotlib
matplotlib.use("Agg")
spec = importlib.util.spec_from_file_location("ref_CLIP_anno", os.path.join(REF, "CLIP_anno.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
assert os.path.realpath(ref.__file__).startswith(REF + os.sep)
from utils.util import EasyDict                                      # noqa: E402  (reference's)
assert sys.modules["utils.util"].__file__.startswith(REF + os.sep)


def main():
    ann = object.__new__(ref.Pix3D_annotator)            # no clip.load: only the matching / CSV methods are used
    gen = torch.Generator().manual_seed(2024)
    N, D, K = 48, 32, 6
    centers = torch.randn(10, D, generator=gen)
    # 6 clusters of 6 images (enough neighbours above the threshold) + 4 clusters of 3 (too few: top-k fallback)
    cluster = torch.cat([torch.arange(36) % 6, 6 + torch.arange(12) % 4])[torch.randperm(N, generator=gen)]
    feats = torch.nn.functional.normalize(centers[cluster] + 0.35 * torch.randn(N, D, generator=gen), dim=-1)
    labels = ["img/chair/%04d.png" % ((i * 37) % 1000) for i in range(N)]       # not sorted: the CSV gets sorted by query
    opt = EasyDict(device="cpu", thres=None, anno_root=OUT, data=EasyDict(dataset="pix3d", pix3d=EasyDict(cat="chair")))
    idx, val = ann.calc_matches(opt, feats, k_nearest=K)
    idx = torch.stack(idx, 0)
    # the thresholded branch: random neighbours among cos >= thres (CPU randperm stream), top-k fallback for sparse rows
    opt_t = EasyDict(opt); opt_t.thres = 0.6
    sim = feats @ feats.t()
    n_valid = ((sim >= 0.6) & (sim < 1.)).sum(1)
    assert (n_valid >= K - 1).any() and (n_valid < K - 1).any(), "want both sub-branches"
    assert ((sim - 0.6).abs() > 1e-4).all()              # no decision sits on a rounding edge
    torch.manual_seed(5)
    idx_t, val_t = ann.calc_matches(opt_t, feats, k_nearest=K)
    idx_t = torch.stack(idx_t, 0)
    ann.split = "train"
    ann.save_anno(opt, ann.label2path, labels, idx, val, k_nearest=K, category_set="custom")
    csv_path = os.path.join(OUT, "chair_train.csv")
    os.replace(csv_path, os.path.join(OUT, "g13_chair_train.csv"))
    np.savez_compressed(os.path.join(OUT, "g13_clip_nn.npz"), feats=feats.numpy(), labels=np.array(labels), k=np.int32(K),
                        idx=idx.numpy(), val=val.numpy(), thres=np.float32(0.6), seed_t=np.int32(5),
                        idx_t=idx_t.numpy(), val_t=val_t.numpy(), n_valid=n_valid.numpy())
    print("wrote g13_clip_nn.npz, g13_chair_train.csv to", OUT)


if __name__ == "__main__":
    main()
