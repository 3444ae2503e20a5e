"""CLIP-NN annotation host logic against golden G13, captured from the reference's own CLIP_anno.py
(tests/golden/make_golden_anno.py): top-k matching, the thresholded random-neighbour branch with its CPU randperm
stream, and the CSV the data loader reads back -- byte for byte.  CPU only (the tower itself is covered by
tests/test_gpu_clip.py)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _annotator():
    import CLIP_anno
    return object.__new__(CLIP_anno.NN_annotator)        # matching / CSV methods do not touch the tower


def _opt(tmp, thres=None):
    from shapeclipper_amd.utils.util import EasyDict as edict
    return edict(device="cpu", thres=thres, anno_root=str(tmp), data=edict(dataset="pix3d", pix3d=edict(cat="chair")))


def test_topk_matches_reference(golden):
    g = golden("g13_clip_nn")
    idx, val = _annotator().calc_matches(_opt("/tmp"), torch.tensor(g["feats"]), k_nearest=int(g["k"]))
    assert np.array_equal(idx.numpy(), g["idx"])
    assert np.array_equal(idx[:, 0].numpy(), np.arange(len(idx)))            # every row starts with the query itself
    np.testing.assert_allclose(val.numpy(), g["val"], atol=2e-6)             # one GEMM vs per-query dot products


def test_thresholded_branch_reproduces_reference_stream(golden):
    g = golden("g13_clip_nn")
    K = int(g["k"])
    assert (g["n_valid"] >= K - 1).any() and (g["n_valid"] < K - 1).any()    # both sub-branches are exercised
    torch.manual_seed(int(g["seed_t"]))
    idx, val = _annotator().calc_matches(_opt("/tmp", float(g["thres"])), torch.tensor(g["feats"]), k_nearest=K)
    assert np.array_equal(idx.numpy(), g["idx_t"]) and np.array_equal(val.numpy(), g["val_t"])


def test_csv_is_byte_identical_to_the_reference(golden, tmp_path):
    g = golden("g13_clip_nn")
    ann = _annotator()
    ann.split = "train"
    path = ann.save_anno(_opt(tmp_path), ann.label2path, [str(x) for x in g["labels"]], torch.tensor(g["idx"]), torch.tensor(g["val"]),
                         k_nearest=int(g["k"]), category_set="custom")
    assert os.path.basename(path) == "chair_train.csv"
    assert open(path, "rb").read() == open(os.path.join(GOLDEN_DIR, "g13_chair_train.csv"), "rb").read()
    ann.split = "val"
    assert os.path.basename(ann.save_anno(_opt(tmp_path), ann.label2path, ["a"], torch.zeros(1, 6, dtype=torch.long), torch.ones(1, 6),
                                          k_nearest=6)) == "all_val.csv"
