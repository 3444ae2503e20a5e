"""Chamfer grid search vs all pairs on SURFACE clouds (what the evaluation compares: 100,000 samples of the predicted iso-surface against
100,000 ground-truth surface points, batch 1), as a function of how far apart the two surfaces are.
    python tools/perf_chamfer_surface.py [B [delta]]      (delta: only that row, e.g. under rocprofv3 --kernel-trace --stats)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chamfer_3D  # noqa: E402


def sphere(B, n, r, gen, bumps=0.0):
    v = torch.randn(B, n, 3, device="cuda", generator=gen)
    v = v / v.norm(dim=-1, keepdim=True)
    rad = r * (1 + bumps * torch.sin(7 * v[..., :1]) * torch.cos(5 * v[..., 1:2]))
    return (v * rad).contiguous()


def run(x1, x2, mode, iters=5):
    B, N, M = x1.shape[0], x1.shape[1], x2.shape[1]
    d1, d2 = torch.zeros(B, N, device="cuda"), torch.zeros(B, M, device="cuda")
    i1, i2 = torch.zeros(B, N, dtype=torch.int32, device="cuda"), torch.zeros(B, M, dtype=torch.int32, device="cuda")
    old, chamfer_3D.SEARCH = chamfer_3D.SEARCH, mode
    try:
        chamfer_3D.forward(x1, x2, d1, d2, i1, i2)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(iters):
            chamfer_3D.forward(x1, x2, d1, d2, i1, i2)
        torch.cuda.synchronize()
    finally:
        chamfer_3D.SEARCH = old
    return (time.time() - t0) / iters * 1e3, (d1, d2, i1, i2)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    only = float(sys.argv[2]) if len(sys.argv) > 2 else None
    gen = torch.Generator(device="cuda").manual_seed(0)
    for delta, bumps in ((0.0, 0.0), (0.005, 0.0), (0.02, 0.05), (0.05, 0.1), (0.1, 0.1), (0.2, 0.2), (0.4, 0.3)):
        if only is not None and abs(delta - only) > 1e-9:
            continue
        a, b = sphere(B, 100000, 0.4, gen), sphere(B, 100000, 0.4 + delta, gen, bumps)
        tg, og = run(a, b, "grid")
        tb, ob = run(a, b, "brute")
        same = all(torch.equal(x, y) for x, y in zip(og, ob))
        print("B=%d radius 0.4 vs %.3f (bumps %.2f): grid %.3f ms, all pairs %.3f ms, same results %s, mean nn dist %.4f" % (
            B, 0.4 + delta, bumps, tg, tb, same, float(og[0].sqrt().mean())))


if __name__ == "__main__":
    main()
