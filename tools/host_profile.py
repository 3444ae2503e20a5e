"""Where does the HOST time of a training step go?  cProfile over a few small-batch steps (the GPU is not the limit
at B=8, so wall time = host time).  Usage (GPU box): B=8 python tools/host_profile.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
import bench
from shapeclipper_amd.utils.util import EasyDict as edict

runner, opt, batch = bench.build_runner(int(os.environ.get("B", "8")))
def step():
    opt.H, opt.W = opt.image_size
    return runner.train_iteration(opt, edict(batch), None)
for _ in range(3):
    step()
torch.cuda.synchronize()
# phase timing (host): forward / backward / optimizer
import types
g = runner.graph
t = dict(fwd=0.0, bwd=0.0, opt=0.0, n=0)
orig_fwd = g.forward
def timed_fwd(*a, **k):
    t0 = time.time(); r = orig_fwd(*a, **k); t["fwd"] += time.time() - t0; return r
g.forward = timed_fwd
orig_bw = torch.Tensor.backward
def timed_bw(self, *a, **k):
    t0 = time.time(); r = orig_bw(self, *a, **k); t["bwd"] += time.time() - t0; return r
torch.Tensor.backward = timed_bw
N = 5
t0 = time.time()
for _ in range(N):
    step()
torch.cuda.synchronize()
tot = (time.time() - t0) / N * 1e3
print("step %.1f ms: forward %.1f ms, backward %.1f ms, rest (optimizer, checks) %.1f ms" %
      (tot, t["fwd"] / N * 1e3, t["bwd"] / N * 1e3, tot - (t["fwd"] + t["bwd"]) / N * 1e3))
torch.Tensor.backward = orig_bw
g.forward = orig_fwd
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "45")))

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
