"""Cluster form of the CLIP layers (csrc/clip_cluster.hpp) against the launch-per-operation form: same weights and images in two processes
(SC_CLIP_CLUSTER_MAX_B=0 switches the cluster form off), embeddings compared, both timed.  python tools/dbg_clip_cluster.py [batches...]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
OUT = os.environ.get("SC_DBG_OUT", "/tmp")


def child(tag, batches):
    import torch
    from shapeclipper_amd.model.clip_vit import ClipVisionTower, VIT_B32
    torch.manual_seed(0)
    t = ClipVisionTower(**VIT_B32).cuda()
    res = {}
    for B in batches:
        g = torch.Generator(device="cuda").manual_seed(100 + B)
        x = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
        y = t.encode_image(x)
        torch.cuda.synchronize()
        y2 = t.encode_image(x)
        torch.cuda.synchronize()
        for _ in range(5): t.encode_image(x)
        torch.cuda.synchronize(); t0 = time.time()
        n = 50
        for _ in range(n): t.encode_image(x)
        torch.cuda.synchronize(); dt = (time.time() - t0) / n
        res[B] = y.float().cpu()
        print("%s B=%d: %.3f ms  finite=%s  rerun identical=%s" % (tag, B, dt * 1e3, bool(torch.isfinite(y).all()), bool((y == y2).all())), flush=True)
    torch.save(res, os.path.join(OUT, "clip_%s.pt" % tag))


def stress(n=300, B=32):
    """n calls of the cluster form on one input: every result bit-identical to the first (python tools/dbg_clip_cluster.py stress)."""
    import torch
    from shapeclipper_amd.model.clip_vit import ClipVisionTower, VIT_B32
    torch.manual_seed(0)
    t = ClipVisionTower(**VIT_B32).cuda()
    x = torch.randn(B, 3, 224, 224, device="cuda")
    first = t.encode_image(x)
    bad = 0
    for i in range(n):
        if i % 50 == 0:
            junk = torch.randn(64, 1024, 1024, device="cuda").sum()       # something else on the chip in between: other cache contents, other clocks
        bad += int(not torch.equal(t.encode_image(x), first))
    print("stress: %d calls at B=%d, %d differ from the first" % (n, B, bad))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "stress":
        os.environ.setdefault("SC_CLIP_CLUSTER_MIN_B", "1"); os.environ.setdefault("SC_CLIP_CLUSTER_MAX_B", "64")
        globals()["stress"]()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("cluster", "launches"):
        child(sys.argv[1], [int(b) for b in sys.argv[2:]])
        sys.exit(0)
    batches = sys.argv[1:] or ["1", "5", "8", "32", "33", "64"]
    for tag, env in (("launches", {"SC_CLIP_CLUSTER_MAX_B": "0"}), ("cluster", {"SC_CLIP_CLUSTER_MIN_B": "1", "SC_CLIP_CLUSTER_MAX_B": "64"})):
        e = dict(os.environ); e.update(env)
        rc = subprocess.call(["timeout", "300", sys.executable, os.path.abspath(__file__), tag] + batches, env=e)
        if rc: print("%s: exit code %d" % (tag, rc))
    import torch
    a = torch.load(os.path.join(OUT, "clip_launches.pt")); b = torch.load(os.path.join(OUT, "clip_cluster.pt"))
    for B in a:
        d = (a[B] - b[B]).abs().max().item(); s = a[B].abs().max().item()
        cos = torch.nn.functional.cosine_similarity(a[B], b[B], dim=1).min().item()
        print("B=%d: max |diff| %.3e (scale %.3e)  min cos %.7f  identical=%s" % (B, d, s, cos, bool((a[B] == b[B]).all())))

