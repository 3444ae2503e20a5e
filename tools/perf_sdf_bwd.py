"""Timing of the SDF backward alone at the bench size (B=32 x 512 rays x 64 samples = 1,048,576 points):
python tools/perf_sdf_bwd.py [fused|unfused] [n_images=32]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import ops, packing
from oracle import reference_ops as R          # weights only (tuning tool, not product)
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda")
torch.manual_seed(0)
W = {k: v.to(dev) for k, v in R.init_sdf_weights(R.Cfg(), 0).items()}
z = torch.randn(B, 64, device=dev) * 0.3
pack, cb = packing.pack_sdf(W, z)
npi = 512 * 64
n = B * npi
pts = torch.rand(n, 3, device=dev) * 1.2 - 0.6
sdf, grad, feat, sa, sp = ops.sdf_forward(pts, pack, cb, npi, stash=True)
g_sdf, g_grad, g_feat = torch.randn(n, device=dev), torch.randn(n, 3, device=dev), torch.randn_like(feat) * 0.1
run = lambda: ops.sdf_backward(pts, pack, npi, B, True, sa, sp, g_sdf, g_grad, g_feat, fused=(mode == "fused"))
for _ in range(3): run()
torch.cuda.synchronize()
evs = []
for _ in range(10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); run(); e.record(); evs.append((s, e))
torch.cuda.synchronize()
ms = sorted(s.elapsed_time(e) for s, e in evs)
flop = (161280 + 864 * 2048 // 16) * n
print("%s%s: %.3f ms (best %.3f) per SDF backward of %d points -> %.1f TFLOP/s algorithmic (input-gradient + weight-gradient)"
      % (mode, "", sum(ms) / len(ms), ms[0], n, flop / (sum(ms) / len(ms) * 1e-3) / 1e12))
