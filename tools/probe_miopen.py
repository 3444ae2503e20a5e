import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model import resnet
mode = sys.argv[1]
if mode == "nocudnn":
    torch.backends.cudnn.enabled = False
net = resnet.build("resnet34").cuda()
x = torch.rand(32, 3, 224, 224, device="cuda")
for i in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    net(x).sum().backward()
    torch.cuda.synchronize(); print(mode, "iter", i, "%.3f s" % (time.time() - t0), flush=True)
