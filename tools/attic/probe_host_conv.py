"""Host-side cost of the MIOpen convolution calls: ResNet-18 fwd+bwd at B=2 (GPU idle most of the time, so wall =
host) with cudnn.benchmark off/on.  Usage: python tools/probe_host_conv.py"""
import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model import resnet
x = torch.rand(2, 3, 224, 224, device="cuda")
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    net = resnet.build("resnet18").cuda()
    for i in range(13):
        if i == 3:
            torch.cuda.synchronize(); t0 = time.time()
        net.zero_grad(set_to_none=True)
        net(x).sum().backward()
    torch.cuda.synchronize()
    print("benchmark=%s: %.2f ms per fwd+bwd (20 convs)" % (bench, (time.time() - t0) / 10 * 1e3), flush=True)
