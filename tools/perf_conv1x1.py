"""The 1x1 / stride-2 shortcut convolutions per trunk shape (forward, backward-data, backward-weight): run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeclipper_amd import ops
for B in (64, 96):
    for hin, cin in ((56, 64), (28, 128), (14, 256)):
        x = torch.randn(B, cin, hin, hin, device="cuda"); w = torch.randn(2 * cin, cin, device="cuda") * 0.05
        gy = torch.randn(B, 2 * cin, hin // 2, hin // 2, device="cuda")
        for _ in range(10):
            ops.conv1x1s2_forward(x, w); ops.conv1x1s2_backward_data(gy, w); ops.conv1x1s2_backward_weight(gy, x)
torch.cuda.synchronize()
