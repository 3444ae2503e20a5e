"""A/B: full training step with MIOpen immediate mode (default) vs cudnn.benchmark=True + the given MIOPEN_FIND_MODE."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
import bench
from shapeclipper_amd.utils.util import EasyDict as edict
runner, opt, batch = bench.build_runner(int(os.environ.get("B", "32")))
torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK", "0") == "1"
def step():
    opt.H, opt.W = opt.image_size
    return runner.train_iteration(opt, edict(batch), None)
t0 = time.time()
for _ in range(4):
    step()
torch.cuda.synchronize()
print("warmup %.1f s" % (time.time() - t0))
t0 = time.time()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("benchmark=%s find=%s: %.2f ms/step" % (torch.backends.cudnn.benchmark, os.environ["MIOPEN_FIND_MODE"], (time.time() - t0) / 10 * 1e3))
