"""How many device kernels does each part of Graph.forward launch (forward only)?  Wraps the parts in record_function
ranges and counts the CUDA kernels whose launch falls inside each range.  Usage (GPU box): B=8 python tools/op_census.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
from torch.profiler import profile, ProfilerActivity, record_function
import bench
from shapeclipper_amd.utils.util import EasyDict as edict
from shapeclipper_amd.utils import camera
from shapeclipper_amd.model import graph as G, renderer as R, loss as L
from shapeclipper_amd import packing

def wrap(obj, name, tag):
    fn = getattr(obj, name)
    def inner(*a, **k):
        with record_function("REGION:" + tag):
            return fn(*a, **k)
    setattr(obj, name, inner)

runner, opt, batch = bench.build_runner(int(os.environ.get("B", "8")))
g = runner.graph.module
wrap(g, "select_neighbours", "select_neighbours")
wrap(g, "gather_neighbour_views", "gather_views")
wrap(g, "encode_all_views", "encoders")
wrap(g, "pred_pose", "pred_pose")
wrap(g.renderer, "forward", "renderer(all)")
wrap(camera, "get_center_and_ray", "renderer/rays")
wrap(g.sdf_network, "packed", "renderer/pack_sdf")
wrap(g.rgb_network, "packed", "renderer/pack_rgb")
wrap(g, "compute_loss", "loss")
wrap(g.latent_proj_shape, "forward", "latent_proj")
wrap(g.latent_proj_rgb, "forward", "latent_proj")
wrap(camera, "transform_normal", "transform_normal")

def step():
    opt.H, opt.W = opt.image_size
    return runner.train_iteration(opt, edict(batch), None)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
regions = [(e.name[7:], e.time_range.start, e.time_range.end) for e in ev if e.name.startswith("REGION:")]
launch = [e for e in ev if e.name in ("hipLaunchKernel", "hipExtLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel", "hipLaunchKernelGGL", "cudaLaunchKernel")]
print("regions:", len(regions), "launch events:", len(launch))
cnt = collections.Counter(); tot = 0
for e in launch:
    t = e.time_range.start
    inside = [r for r in regions if r[1] <= t <= r[2]]
    tot += 1
    if not inside:
        cnt["(outside: backward / optimizer / other)"] += 1
    else:
        cnt[min(inside, key=lambda r: r[2] - r[1])[0]] += 1      # innermost region
for k, v in cnt.most_common():
    print("%6d  %s" % (v, k))
print("total launches", tot)
# backward: autograd node census
bw = collections.Counter(e.name.split(":")[-1].strip() for e in ev if e.name.startswith("autograd::engine::evaluate_function"))
print("backward nodes:", sum(bw.values()))
for k, v in bw.most_common(25):
    print("%6d  %s" % (v, k))

# which autograd node / optimizer range does each backward-side launch belong to, and what are the aten kernels it launches?
nodes = [(e.name.split("evaluate_function:")[-1].strip(), e.time_range.start, e.time_range.end) for e in ev
         if e.name.startswith("autograd::engine::evaluate_function")]
nodes += [(e.name, e.time_range.start, e.time_range.end) for e in ev if e.name.startswith("Optimizer.step")]
nodes.sort(key=lambda r: r[1])
import bisect
starts = [r[1] for r in nodes]
aten = [e for e in ev if e.name.startswith("aten::") and e.cpu_parent is not None]
per_node = collections.Counter(); per_node_ops = collections.defaultdict(collections.Counter)
mem = [e for e in ev if e.name in ("hipMemcpyAsync", "hipMemsetAsync", "hipMemcpyWithStream")]
for e in launch + mem:
    t = e.time_range.start
    if any(r[1] <= t <= r[2] for r in regions):
        continue
    i = bisect.bisect_right(starts, t) - 1
    name = nodes[i][0] if i >= 0 and nodes[i][2] >= t else "(no node)"
    per_node[name] += 1
    p = e.cpu_parent
    while p is not None and not p.name.startswith("aten::"):
        p = p.cpu_parent
    per_node_ops[name][p.name if p is not None else e.name] += 1
print("\nbackward / optimizer launches by autograd node (aten parents listed):")
for k, v in per_node.most_common(40):
    print("%6d  %-45s %s" % (v, k, ", ".join("%s x%d" % kv for kv in per_node_ops[k].most_common(6))))

# where in the forward were the select / slice views made (their backward is fill + copy + add_ each)?
fw = collections.defaultdict(collections.Counter)
for e in ev:
    if e.name in ("aten::select", "aten::slice", "aten::unsqueeze", "aten::index_select", "aten::gather") and e.sequence_nr >= 0 and e.cpu_parent is not None:
        t = e.time_range.start
        inside = [r for r in regions if r[1] <= t <= r[2]]
        if inside:
            fw[e.name][min(inside, key=lambda r: r[2] - r[1])[0]] += 1
        elif not (nodes and nodes[0][1] <= t):
            fw[e.name]["(forward, outside regions)"] += 1
for k, c in fw.items():
    print(k, dict(c))
