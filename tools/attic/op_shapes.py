"""Which aten operators (with which shapes) launch the small element-wise kernels of a training step?  torch.profiler over one step,
operators grouped by (name, input shapes), sorted by call count.   B=32 python tools/op_shapes.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = ["bench.py"]
import bench
from shapeclipper_amd.utils.util import EasyDict as edict
runner, opt, batch = bench.build_runner(int(os.environ.get("B", "32")), 0, 0, 1, [])
def step():
    opt.H, opt.W = opt.image_size
    return runner.train_iteration(opt, edict(batch), None)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::mul", "aten::sum", "aten::empty_like", "aten::clone", "aten::contiguous"):
        cnt[(ev.name, str(ev.input_shapes)[:90])] += 1
for (name, shp), c in cnt.most_common(40):
    print("%4d  %-18s %s" % (c, name, shp))
