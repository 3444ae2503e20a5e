"""BatchNorm(+residual+ReLU) kernels of csrc/bn_act.hip per ResNet layer shape, in isolation: us per call and the HBM rate of the algorithmic
bytes (forward: statistics 1R, apply 1R [+1R residual] + 1W; backward: statistics 2R [+1W dres], apply 2R + 1W).
    python tools/perf_bn.py            # shapes of the bs32 step: 64 images x 2 groups (ResNet-34), 96 images x 3 groups (ResNet-18)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shapeclipper_amd import ops  # noqa: E402


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    dev = torch.device("cuda")
    tot = {}
    for N, G, counts in ((64, 2, (6, 8, 12, 6)), (96, 3, (4, 4, 4, 4))):
        for (C, H), cnt in zip(((64, 56), (128, 28), (256, 14), (512, 7)), counts):
            x = torch.randn(N, C, H, H, device=dev)
            res = torch.randn_like(x)
            dy = torch.randn_like(x)
            gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
            rm, rv, nt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.long, device=dev)
            nbytes = x.numel() * 4
            for with_res in (False, True):
                r = res if with_res else None
                y, stats = ops.bn_act_forward(x, r, gamma, beta, rm, rv, nt, True, 0.1, 1e-5, True, G)
                t_f = timed(lambda: ops.bn_act_forward(x, r, gamma, beta, rm, rv, nt, True, 0.1, 1e-5, True, G))
                t_b = timed(lambda: ops.bn_act_backward(dy, x, y if with_res else None, gamma, beta, stats, True, True, True, with_res, G))
                bf = (3 + with_res) * nbytes
                bb = (5 + 2 * with_res) * nbytes          # stats: dy, x (+ y, + dres written); apply: dy (or dres), x, dx
                print("N=%3d G=%d C=%3d %2dx%-2d res=%d | fwd %6.1f us %5.2f TB/s | bwd %6.1f us %5.2f TB/s | x%d per pass" % (
                    N, G, C, H, H, with_res, t_f, bf / t_f * 1e-6, t_b, bb / t_b * 1e-6, cnt // 2))
                tot["fwd"] = tot.get("fwd", 0) + t_f * cnt / 2
                tot["bwd"] = tot.get("bwd", 0) + t_b * cnt / 2
    print("per step (both trunks, alone on the chip): forward %.2f ms, backward %.2f ms" % (tot["fwd"] * 1e-3, tot["bwd"] * 1e-3))


if __name__ == "__main__":
    main()
