"""Stem tail (BatchNorm + ReLU + 3x3/2 max-pool on [N, 64, 112, 112]) forward + backward: run under rocprofv3 --kernel-trace, read tools/trace_by_grid.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeclipper_amd import ops
for N, G in ((64, 2), (96, 3)):
    x = torch.randn(N, 64, 112, 112, device="cuda")
    g = torch.ones(64, device="cuda"); b = torch.zeros(64, device="cuda")
    rm = torch.zeros(64, device="cuda"); rv = torch.ones(64, device="cuda"); nt = torch.zeros((), dtype=torch.int64, device="cuda")
    dy = torch.randn(N, 64, 56, 56, device="cuda")
    for _ in range(10):
        y, idx, st = ops.bn_relu_pool_forward(x, g, b, rm, rv, nt, True, 0.1, 1e-5, G)
        ops.bn_relu_pool_backward(dy, idx, x, g, b, st, True, G)
torch.cuda.synchronize()
