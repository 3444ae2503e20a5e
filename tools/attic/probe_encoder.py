import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model import resnet
def run(tag, net, x, opt=None, n=6):
    for i in range(n):
        if i == 2: torch.cuda.synchronize(); t0 = time.time()
        if opt: opt.zero_grad()
        net(x).sum().backward()
        if opt: opt.step()
    torch.cuda.synchronize(); print(tag, "%.2f ms" % ((time.time() - t0) / (n - 2) * 1e3), flush=True)
x = torch.rand(32, 3, 224, 224, device="cuda")
for name in ("resnet34", "resnet18"):
    net = resnet.build(name).cuda()
    run(name + " nchw fp32", net, x)
    net_cl = resnet.build(name).cuda().to(memory_format=torch.channels_last)
    run(name + " channels_last fp32", net_cl, x.to(memory_format=torch.channels_last))
    torch.backends.cudnn.benchmark = True
    run(name + " nchw benchmark=True", net, x, n=8)
    torch.backends.cudnn.benchmark = False
net = resnet.build("resnet34").cuda()
ps = list(net.parameters())
run("resnet34 + Adam(foreach default)", net, x, torch.optim.Adam(ps, lr=1e-4))
run("resnet34 + Adam(fused)", net, x, torch.optim.Adam(ps, lr=1e-4, fused=True))
