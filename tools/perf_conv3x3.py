"""One shape of csrc/conv3x3.hip in a loop (for rocprofv3 --kernel-trace / --pmc):  python tools/perf_conv3x3.py SIDE CH BATCH [iters]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapeclipper_amd import ops

side, ch, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
x = torch.randn(batch, ch, side, side, device="cuda:0")
w = torch.randn(ch, ch, 3, 3, device="cuda:0") * 0.05
wp = ops.conv3x3_pack(w, side)
for _ in range(3):
    ops.conv3x3_apply(x, wp, ch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.conv3x3_apply(x, wp, ch)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
flop = 2.0 * batch * ch * side * side * ch * 9
print("conv3x3 %dx%d C=%d B=%d: %.3f ms  %.1f TFLOP/s (%.2f of 157.3)" % (side, side, ch, batch, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3))
