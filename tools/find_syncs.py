"""Run two training iterations with torch's sync debug mode on and print every synchronising call site.
Usage (GPU box): python tools/find_syncs.py"""
import os, sys, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
import bench

def main():
    from shapeclipper_amd.utils.util import EasyDict as edict
    runner, opt, batch = bench.build_runner(int(os.environ.get("B", "8")))
    def var_fn():
        opt.H, opt.W = opt.image_size
        return edict(batch)
    for _ in range(2):
        runner.train_iteration(opt, var_fn())
    torch.cuda.synchronize()
    seen = {}
    def hook(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" in str(message):
            st = [f for f in traceback.extract_stack() if "/repo/" in f.filename and "find_syncs" not in f.filename]
            key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-3:]))
            if not key:
                key = str(message)[:80] + " @ " + " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno)
                                                               for f in reversed(traceback.extract_stack()[-8:-1]))
            seen[key] = seen.get(key, 0) + 1
    warnings.showwarning = hook
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    runner.train_iteration(opt, var_fn())
    torch.cuda.set_sync_debug_mode("default")
    for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
        print("%3d  %s" % (v, k))
    print("total synchronising calls in one iteration:", sum(seen.values()))

main()
