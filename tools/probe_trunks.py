"""How well do the two ResNet trunks of a step overlap on two streams?  Forward + backward of the ResNet-34 encoder (2 x 32 images) and
of the ResNet-18 view estimator (3 x 32), each alone and both at once (side stream, as Graph.encode_all_views does), with the 3x3
convolutions on csrc/conv3x3*.hip and on MIOpen.   python tools/probe_trunks.py"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapeclipper_amd.model import resnet

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


for hip_enc, hip_est in ((True, True), (False, False), (True, False), (False, True), (True, True), (False, False)):
    enc.hip_conv3x3, est.hip_conv3x3 = hip_enc, hip_est
    a, b, c = timeit(run_enc), timeit(run_est), timeit(run_both)
    print("3x3 convolutions encoder %-6s estimator %-6s: encoder alone %.2f ms, estimator alone %.2f ms, sum %.2f; both on two streams %.2f ms (%.2f ms hidden)"
          % ("HIP" if hip_enc else "MIOpen", "HIP" if hip_est else "MIOpen", a, b, a + b, c, a + b - c), flush=True)
