"""3x3 / stride-1 weight gradient per trunk layer shape: fp32-MFMA kernel vs the bf16x3-split kernel (tuning tool).
    python tools/perf_wgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeclipper_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

tot = [0.0, 0.0]
for net, B, layers in (("resnet34", 64, ((56, 64, 6), (28, 128, 7), (14, 256, 11), (7, 512, 5))), ("resnet18", 96, ((56, 64, 4), (28, 128, 3), (14, 256, 3), (7, 512, 3)))):
    for side, c, count in layers:
        x = torch.randn(B, c, side, side, device="cuda"); gy = torch.randn(B, c, side, side, device="cuda")
        a, b = t(lambda: ops.conv3x3_backward_weight(gy, x)), t(lambda: ops.conv3x3_backward_weight(gy, x, split=True))
        fl = 2.0 * 9 * c * c * B * side * side
        print("%s B=%d %3dx%-3d %3d ch x%2d: fp32 %6.1f us (%5.1f TF/s)  split %6.1f us (%5.1f TF/s)  x%.2f" % (net, B, side, side, c, count, a, fl / a / 1e6, b, fl / b / 1e6, a / b))
        tot[0] += a * count; tot[1] += b * count
print("per step: fp32 %.2f ms, split %.2f ms" % (tot[0] / 1e3, tot[1] / 1e3))
# digest of the split results (same-bits check across library variants)
torch.manual_seed(0)
x = torch.randn(8, 128, 28, 28, device="cuda"); gy = torch.randn(8, 128, 28, 28, device="cuda")
dw = ops.conv3x3_backward_weight(gy, x, split=True)
ref = torch.nn.grad.conv2d_weight(x.double(), (128, 128, 3, 3), gy.double(), padding=1)
print("split digest %.10e   max |err| vs float64 / max |ref| = %.2e" % (float(dw.double().sum()), float((dw.double() - ref).abs().max() / ref.abs().max())))
