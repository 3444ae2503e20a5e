"""Chamfer3D forward throughput (both directions), BASELINE config[2]/[4] sizes."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import chamfer_3D
for B, N in ((1, 100000), (8, 100000), (32, 100000)):
    a = torch.rand(B, N, 3, device="cuda") - 0.5; b = torch.rand(B, N, 3, device="cuda") - 0.5
    d1 = torch.zeros(B, N, device="cuda"); d2 = torch.zeros(B, N, device="cuda")
    i1 = torch.zeros(B, N, dtype=torch.int32, device="cuda"); i2 = torch.zeros(B, N, dtype=torch.int32, device="cuda")
    chamfer_3D.forward(a, b, d1, d2, i1, i2); torch.cuda.synchronize()
    t0 = time.time(); n = 5
    for _ in range(n): chamfer_3D.forward(a, b, d1, d2, i1, i2)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    pairs = 2.0 * B * N * N
    print("chamfer B=%d N=M=%d: %.2f ms  %.2f Tpairs/s  %.1f TFLOP/s algorithmic (8 FLOP/pair) = %.1f%% of fp32 VALU peak" % (B, N, dt * 1e3, pairs / dt / 1e12, 8 * pairs / dt / 1e12, 100 * 8 * pairs / dt / 157.3e12))
