"""Feasibility probe: the product ResNet-34 pass (grouped BN, MIOpen / HIP convolutions) under torch.cuda.make_graphed_callables."""
import copy
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapeclipper_amd.model import resnet

name, B, groups = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
net = resnet.build(name).cuda().train()
x = torch.randn(B, 3, 224, 224, device="cuda")


class Wrap(torch.nn.Module):
    def __init__(self, net, groups):
        super().__init__()
        self.net, self.groups = net, groups

    def forward(self, x):
        return self.net(x, groups=self.groups)


def step(m, xin):
    y = m(xin)
    y.square().mean().backward()
    return y


def timeit(m, n=20):
    xin = x.clone().requires_grad_(True)
    for _ in range(3):
        step(m, xin)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        step(m, xin)
    t_host = time.time() - t0
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3, t_host / n * 1e3


eager = Wrap(copy.deepcopy(net), groups)
ms_e, host_e = timeit(eager)
print("%s B=%d groups=%d eager: %.2f ms per fwd+bwd (host %.2f ms)" % (name, B, groups, ms_e, host_e), flush=True)
g_mod = Wrap(copy.deepcopy(net), groups)
graphed = torch.cuda.make_graphed_callables(g_mod, (x.clone().requires_grad_(True),))
ms_g, host_g = timeit(graphed)
print("%s B=%d groups=%d graphed: %.2f ms per fwd+bwd (host %.2f ms)" % (name, B, groups, ms_g, host_g), flush=True)
# same numbers?
a, b = Wrap(copy.deepcopy(net), groups), Wrap(copy.deepcopy(net), groups)
gb = torch.cuda.make_graphed_callables(b, (x.clone().requires_grad_(True),))
xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
ya, yb = step(a, xa), step(gb, xb)
print("logits diff %.2e, d input diff %.2e, d conv1.weight diff %.2e, running_var diff %.2e" % (
    float((ya - yb).abs().max()), float((xa.grad - xb.grad).abs().max() / xa.grad.abs().max()),
    float((a.net.conv1.weight.grad - b.net.conv1.weight.grad).abs().max() / a.net.conv1.weight.grad.abs().max()),
    float((a.net.layer1[0].bn1.running_var - b.net.layer1[0].bn1.running_var).abs().max())))
