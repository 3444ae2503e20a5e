"""Which stock torch operators still launch device kernels in a bs32 training step?  (name, input shapes) -> calls and device time.
    B=32 python tools/aten_census.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from shapeclipper_amd.utils.util import EasyDict as edict

runner, opt, batch = bench.build_runner(int(os.environ.get("B", "32")))
def step():
    opt.H, opt.W = opt.image_size
    return runner.train_iteration(opt, edict(batch), None)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dev = getattr(e, "self_device_time_total", None)
    if dev is None: dev = getattr(e, "self_cuda_time_total", 0)
    if dev > 0 and (e.key.startswith("aten::") or "Backward" in e.key or e.key.startswith("Optimizer")):
        rows.append((dev, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("operators with device time: %d kinds, %.2f ms of device time, %d calls" % (len(rows), tot / 1e3, sum(r[1] for r in rows)))
rows2 = sorted([r for r in rows if r[2].startswith("aten::")], key=lambda r: -r[1])
print("stock aten operators: %d kinds, %d calls, %.2f ms of device time" % (len(rows2), sum(r[1] for r in rows2), sum(r[0] for r in rows2) / 1e3))
print("--- stock operators by call count")
for dev, n, k, sh in rows2[:int(os.environ.get('TOP', '45'))]:
    print("%8.1f us %4d x  %-28s %s" % (dev, n, k, sh))
print("--- by device time")
for dev, n, k, sh in rows[:25]:
    print("%8.1f us %4d x  %-28s %s" % (dev, n, k, sh))
