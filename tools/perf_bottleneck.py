"""Timing of the fused 1x1 bottleneck launches and of the latent-bias launches (round 5): python tools/perf_bottleneck.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import ops
from shapeclipper_amd.model.view_estimator import Bottleneck_Linear
dev = torch.device("cuda")

def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for N, G in ((32, 1), (64, 2), (96, 3)):
    C = 512
    x = torch.randn(N, C, device=dev); w = torch.randn(C, C, device=dev) * 0.05; gm = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev)
    rm, rv, nt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    out, y, st = ops.linear_bn_forward(x, w, gm, bt, x, rm, rv, nt, True, 0.1, 1e-5, True, G)
    g = torch.randn(N, C, device=dev)
    t_f = timed(lambda: ops.linear_bn_forward(x, w, gm, bt, x, rm, rv, nt, True, 0.1, 1e-5, True, G))
    t_b = timed(lambda: ops.linear_bn_backward(g, None, None, None, out, y, st, gm, x, True, True, True, G))
    gy = ops.linear_bn_backward(g, None, None, None, out, y, st, gm, x, True, True, True, G)[0]
    t_b2 = timed(lambda: ops.linear_bn_backward(None, gy, w, None, out, y, st, gm, x, False, True, True, G))
    t_d = timed(lambda: ops.linear_backward_data(gy, w, g))
    t_mm = timed(lambda: x @ w.t())
    print("N=%3d groups=%d C=512: linear_bn forward %.1f us, backward (gradient given) %.1f us, backward (gradient = gy W of the next layer) %.1f us, "
          "backward data %.1f us; a stock [N,512]x[512,512] product alone %.1f us (events over 50 back-to-back launches)" % (N, G, t_f, t_b, t_b2, t_d, t_mm))
B, Z = 32, 64
z = torch.randn(B, Z, device=dev); lat = torch.randn(192, Z, device=dev); bias = torch.randn(5, 64, device=dev); post = torch.tensor([1.0, 0.7, 0.7], device=dev)
g = torch.randn(B, 5, 64, device=dev)
print("latent bias B=32: forward %.1f us, backward %.1f us" % (timed(lambda: ops.latent_bias_forward(z, lat, bias, post)), timed(lambda: ops.latent_bias_backward(g, z, lat, post, 5))))
