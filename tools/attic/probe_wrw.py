"""Time one ResNet-34 fwd+bwd (B=32, fp32, NCHW) under the MIOpen settings given in the environment and list the
convolution kernels it ran (A/B probe for the weight-gradient solver choice).  Usage: ENVVAR=.. python tools/probe_wrw.py"""
import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model import resnet
x = torch.rand(32, 3, 224, 224, device="cuda")
net = resnet.build(os.environ.get("NET", "resnet34")).cuda()
for i in range(7):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.time()
    net.zero_grad(set_to_none=True)
    net(x).sum().backward()
torch.cuda.synchronize()
print(os.environ.get("TAG", "run"), "%.2f ms" % ((time.time() - t0) / 4 * 1e3), flush=True)
if os.environ.get("KERNELS"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        net.zero_grad(set_to_none=True)
        net(x).sum().backward()
        torch.cuda.synchronize()
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:12]:
        print("   %8.1f us x%4d  %s" % (e.device_time_total, e.count, e.key[:90]))
