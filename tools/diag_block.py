import copy, torch, sys
sys.path.insert(0, '.')
from shapeclipper_amd.model import resnet
torch.manual_seed(0); dev = torch.device("cuda:0")
resnet.HIP_CONV3X3_SPLIT = False
net = resnet.ResNet([2, 2, 2, 2]).to(dev).train()
x = torch.randn(8, 3, 224, 224, device=dev)
def run(fused, groups):
    resnet.FUSED_BLOCK = fused
    n = copy.deepcopy(net)
    xi = x.clone().requires_grad_(True)
    y = n(xi, groups=groups)
    (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
    return {k: p.grad.clone() for k, p in n.named_parameters()}
for groups in (1, 2):
    a, b, c = run(False, groups), run(False, groups), run(True, groups)
    bad_ab = [k for k in a if not torch.equal(a[k], b[k])]
    bad_ac = [k for k in a if not torch.equal(a[k], c[k])]
    print("groups", groups, "old vs old differing:", len(bad_ab), bad_ab[:4], "| old vs fused differing:", len(bad_ac), bad_ac[:6])
