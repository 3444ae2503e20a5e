"""End-to-end run of the offline CLIP-NN annotator on the GPU (reference CLIP_anno.py:129-182 `main()`): synthetic image set ->
HIP ViT tower (fp16, as the reference runs CLIP on a GPU) -> L2-normalise -> cosine top-k -> `<anno_root>/<cat>_<split>.csv`.
The matching / CSV code is pinned byte for byte by the CPU tests (tests/test_clip_anno.py, golden G13); this test covers the
pipeline wiring on the device: CSV header, row count, sort order, score format, and every query's own neighbour list."""
import csv
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("model", ["ViT-B/32"])
def test_annotator_main_on_synthetic_set(tmp_path, monkeypatch, model):
    import CLIP_anno
    n, k = 96, 6
    argv = ["CLIP_anno.py", "--yaml=%s/options/clip/pix3d.yaml" % ROOT, "--data.dataset=synthetic", "--data.synthetic_len=%d" % n,
            "--anno_root=%s" % tmp_path, "--output_root=%s/out" % tmp_path, "--clip_model=%s" % model, "--batch_size=32"]
    monkeypatch.setattr(sys, "argv", argv)
    torch.manual_seed(0)
    CLIP_anno.main()
    files = os.listdir(tmp_path)
    csvs = [f for f in files if f.endswith(".csv")]
    assert csvs == ["chair_train.csv"], files      # `<category>_<split>.csv`: the synthetic set reuses the Pix3D category block of the yaml
    rows = list(csv.reader(open(os.path.join(tmp_path, csvs[0]))))
    header, body = rows[0], rows[1:]
    assert header == ["Query"] + ["Top_%d" % i for i in range(1, k)] + ["Top_%d_score" % i for i in range(1, k)]
    assert len(body) == n and [r[0] for r in body] == sorted("synthetic/img_%05d.png" % i for i in range(n))
    for r in body:
        assert len(r) == 1 + 2 * (k - 1)
        assert r[0] not in r[1:k]                                   # the query itself (top-1 of the similarity) is not listed
        assert len(set(r[1:k])) == k - 1                            # distinct neighbours
        scores = [float(v) for v in r[k:]]
        assert all(-1.0001 <= v <= 1.0001 for v in scores) and scores == sorted(scores, reverse=True)
        assert all(len(v.split(".")[1]) == 4 for v in r[k:])        # "{:.4f}"
    # the annotator's embeddings are what the tower returns for the same images (wiring: batching, normalisation, device moves)
    opt = CLIP_anno.options.set(opt_cmd=CLIP_anno.options.parse_arguments(argv[1:]), verbose=False)
    ann = CLIP_anno.NN_annotator(opt)
    gen = torch.Generator().manual_seed(0)
    images = torch.randn(n, 3, 224, 224, generator=gen)
    feats = ann.embed_split(opt, images)
    assert feats.shape == (n, ann.clip_dim) and torch.allclose(feats.norm(dim=-1), torch.ones(n, device=feats.device), atol=1e-5)
    idx, val = ann.calc_matches(opt, feats, k_nearest=k)
    assert torch.equal(idx[:, 0].cpu(), torch.arange(n))           # self-match in column 0 of the similarity ranking
