import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from shapeclipper_amd.functional import bn_act
torch.manual_seed(0)
shape=(32,512,7,7); N,C,H,W=shape; dev="cuda"
x0=(torch.randn(shape,device=dev)*1.7+0.4); r0=torch.randn(shape,device=dev); cot=torch.randn(shape,device=dev)
bn_a=nn.BatchNorm2d(C).to(dev)
with torch.no_grad():
    bn_a.weight.copy_(torch.randn(C)*0.5+1.0); bn_a.bias.copy_(torch.randn(C)*0.3)
    bn_a.running_mean.copy_(torch.randn(C)*0.2); bn_a.running_var.copy_(torch.rand(C)+0.5)
bn_b=copy.deepcopy(bn_a); bn_a.eval(); bn_b.eval()
x=x0.clone().requires_grad_(True); r=r0.clone().requires_grad_(True)
y=torch.relu(bn_a(x)+r); (y*cot).sum().backward()
x2=x0.clone().requires_grad_(True); r2=r0.clone().requires_grad_(True)
y2=bn_act(bn_b,x2,residual=r2,relu=True); (y2*cot).sum().backward()
d=(x2.grad-x.grad).abs()
print("y err", float((y2-y).abs().max()), "dres err", float((r2.grad-r.grad).abs().max()), "dx err", float(d.max()))
idx=(d>1e-3).nonzero()
print(len(idx), idx[:10])
i=idx[0]; n,c,h,w=[int(v) for v in i]
print("ref", float(x.grad[n,c,h,w]), "got", float(x2.grad[n,c,h,w]), "g", float(r.grad[n,c,h,w]), "scale", float(bn_a.weight[c]*torch.rsqrt(bn_a.running_var[c]+bn_a.eps)))
print("per-channel bad count", torch.bincount(idx[:,1], minlength=C).nonzero().flatten()[:20])
