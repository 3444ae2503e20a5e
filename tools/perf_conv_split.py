"""3x3 / stride-1 forward (= backward-data) in split arithmetic per trunk layer shape (tuning tool; also the target of PMC passes):
    python tools/perf_conv_split.py [iters=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeclipper_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tot = 0.0
for net, B, layers in (("resnet34", 64, ((56, 64, 6), (28, 128, 7), (14, 256, 11), (7, 512, 5))), ("resnet18", 96, ((56, 64, 4), (28, 128, 3), (14, 256, 3), (7, 512, 3)))):
    for side, c, count in layers:
        x = torch.randn(B, c, side, side, device="cuda"); w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
        pk = ops.conv3x3_pack(w, side, False, True)
        for _ in range(2): ops.conv3x3_apply(x, pk, c, True)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): ops.conv3x3_apply(x, pk, c, True)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / iters * 1e3
        fl = 2.0 * 9 * c * c * B * side * side
        print("%s B=%d %3dx%-3d %3d ch x%2d: %6.1f us (%5.1f TF/s)" % (net, B, side, side, c, count, us, fl / us / 1e6))
        tot += 2 * us * count
print("forward + backward-data per step: %.2f ms" % (tot / 1e3))
