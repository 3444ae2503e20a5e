import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1:2]
sys.argv = ["bench.py"]
import bench
from shapeclipper_amd.utils.util import EasyDict as edict
runner, opt, batch = bench.build_runner(32, 0, 0, 1, [])
def step():
    opt.H, opt.W = opt.image_size
    return runner.train_iteration(opt, edict(batch), None)
import gc
if mode == ['nogc']:
    gc.collect(); gc.freeze(); gc.disable()
ts = []
mem = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.time(); step(); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    st = torch.cuda.memory_stats()
    mem.append((st["reserved_bytes.all.current"] >> 20, st["segment.all.allocated"], st["num_device_alloc"] if "num_device_alloc" in st else -1))
print(" ".join("%.1f" % t for t in ts))
print("reserved MiB / segments allocated / device allocs per step:", " ".join("%d/%d/%d" % m for m in mem))
