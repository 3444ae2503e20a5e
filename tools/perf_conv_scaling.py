"""Fixed cost of a conv3x3 launch: time vs batch for one layer shape (T = a + b B).  python tools/perf_conv_scaling.py SIDE CH [split]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapeclipper_amd import ops

side, ch = int(sys.argv[1]), int(sys.argv[2])
split = len(sys.argv) > 3 and sys.argv[3] == "split"
w = torch.randn(ch, ch, 3, 3, device="cuda:0") * 0.05
wp = ops.conv3x3_pack(w, side, False, split)
rows = []
for batch in (8, 16, 32, 64, 96, 128, 192, 256):
    x = torch.randn(batch, ch, side, side, device="cuda:0")
    for _ in range(3):
        ops.conv3x3_apply(x, wp, ch, split)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv3x3_apply(x, wp, ch, split)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    flop = 2.0 * batch * ch * side * side * ch * 9
    rows.append((batch, us, flop / us / 1e6))
print("conv3x3 %dx%d C=%d %s: " % (side, side, ch, "split" if split else "fp32") + "  ".join("B=%d %.0f us (%.0f TF/s)" % r for r in rows))
