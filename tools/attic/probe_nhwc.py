"""A/B probe: ResNet fwd+bwd (B=32, fp32) NCHW vs channels_last under PYTORCH_MIOPEN_SUGGEST_NHWC, stock BN operators."""
import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model import resnet
resnet.FUSED_BN = False
cl = os.environ.get("CL", "0") == "1"
x = torch.rand(32, 3, 224, 224, device="cuda")
net = resnet.build(os.environ.get("NET", "resnet34")).cuda()
if cl:
    net = net.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
for i in range(7):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.time()
    net.zero_grad(set_to_none=True)
    net(x).sum().backward()
torch.cuda.synchronize()
print(os.environ.get("TAG", "run"), "%.2f ms" % ((time.time() - t0) / 4 * 1e3), flush=True)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    net.zero_grad(set_to_none=True)
    net(x).sum().backward()
    torch.cuda.synchronize()
ev = prof.key_averages()
print("   total device time %.2f ms" % (sum(e.device_time_total for e in ev) / 1e3))
for e in sorted(ev, key=lambda e: -e.device_time_total)[:14]:
    print("   %8.1f us x%4d  %s" % (e.device_time_total, e.count, e.key[:100]))
