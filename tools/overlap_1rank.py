"""What the gradient-exchange schedules cost on ONE rank (VERDICT r03 next #4a): the bs32 training step with a 1-rank RCCL communicator and
`always_communicate`, so that every collective call of the N > 1 path is issued -- (a) no reducer, (b) one flat all-reduce after backward,
(c) [early | late] with the early segment issued from inside backward on a side stream.  A 1-rank RCCL all-reduce launches NO collective
kernel (in place it is a no-op), so this measures what surrounds the collective -- hooks, the multi-tensor pack of 147 MB into the flat
buffer, the 1/world scale, the buffer broadcast, the stream fork / join -- and NOT the contention between RCCL's kernels and the
chip-filling persistent grids, which no one-GPU run can show (RCCL refuses several ranks per device).
    python tools/overlap_1rank.py [steps=100] [batch=32]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
from shapeclipper_amd.parallel import FlatGradAllReduce
from shapeclipper_amd.utils.util import EasyDict as edict
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
batch_n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
runner, opt, batch = bench.build_runner(batch_n)


def run(label, reducer):
    runner.reducer = reducer
    def step():
        opt.H, opt.W = opt.image_size
        return runner.train_iteration(opt, edict(batch), None)
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(steps): step()
    host = time.time() - t0
    torch.cuda.synchronize(); dt = time.time() - t0
    n = reducer.collectives if reducer is not None else 0
    print("%-58s %.3f ms / step   host enqueue %.3f ms   data-path collectives issued %d" % (label, dt / steps * 1e3, host / steps * 1e3, n))
    if reducer is not None:
        reducer.close()


run("(a) no reducer (the 1-GPU path)", None)
run("(b) one flat all-reduce after backward (overlap off)", FlatGradAllReduce(runner.graph.module, 1, always_communicate=True, overlap=False))
run("(c) [early 94.5 % | late] early from inside backward", FlatGradAllReduce(runner.graph.module, 1, always_communicate=True, overlap=True))
run("(a) again", None)
dist.destroy_process_group()
