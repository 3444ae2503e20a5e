"""Do the two ResNet trunks of a step share the chip better when their persistent convolution grids are sized for a PART of it?
Encoder (ResNet-34, 2 x 32 images) and view estimator (ResNet-18, 3 x 32), forward + backward, both on two streams, at several CU
reservations (SHAPECLIPPER_RESERVE_CUS is read once per process: one process per setting).
    for r in 0 64 96 128; do SHAPECLIPPER_RESERVE_CUS=$r python tools/probe_trunks_split.py; done"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapeclipper_amd.model import resnet
from shapeclipper_amd import _lib

torch.manual_seed(0)
enc = resnet.build("resnet34").cuda().train()
est = resnet.build("resnet18").cuda().train()
x_enc = torch.randn(64, 3, 224, 224, device="cuda")
x_est = torch.randn(96, 3, 224, 224, device="cuda")
side = torch.cuda.Stream()


def run_enc():
    enc(x_enc, groups=2).square().mean().backward()


def run_est():
    est(x_est, groups=3).square().mean().backward()


def run_both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        y2 = est(x_est, groups=3)
    y1 = enc(x_enc, groups=2)
    main.wait_stream(side)
    (y1.square().mean() + y2.square().mean()).backward()


def timeit(fn, n=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


a, b, c = timeit(run_enc), timeit(run_est), timeit(run_both)
print("reserved CUs %s (persistent grids: %d): encoder alone %.2f ms, estimator alone %.2f ms, sum %.2f; both on two streams %.2f ms"
      % (os.environ.get("SHAPECLIPPER_RESERVE_CUS", "0"), _lib.load().sc_grid_cus(), a, b, a + b, c), flush=True)
