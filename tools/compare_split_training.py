"""Does --hip.conv3x3_split change training?  The same bs32 run (same seed, same synthetic batches, same host RNG stream) three times:
fp32-MFMA convolutions, the same again (run-to-run noise of the step itself: atomics in the Chamfer-free train step are deterministic,
so this should be exactly 0), and the bf16-split convolutions.  Prints loss.all of the first steps side by side and the relative
deviation after N steps.   python tools/compare_split_training.py [steps]"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sys.argv = ["bench.py"]
import bench
from shapeclipper_amd.model import resnet
from shapeclipper_amd.utils.util import EasyDict as edict


def run(split):
    torch.manual_seed(0)
    np.random.seed(0)
    runner, opt, batch = bench.build_runner(32, 0, 0, 1, [] if split else ["--hip.conv3x3_split!"])     # (options.set applies the switch)
    assert bool(resnet.HIP_CONV3X3_SPLIT) == bool(split)
    losses = []
    for _ in range(steps):
        opt.H, opt.W = opt.image_size
        loss = runner.train_iteration(opt, edict(batch), None)
        losses.append(float(loss.all))
    return np.array(losses)


a, a2, b = run(False), run(False), run(True)
print("step   fp32-MFMA      fp32-MFMA again   bf16-split     rel. deviation (again / split)")
for i in list(range(0, min(steps, 10))) + list(range(10, steps, max(1, steps // 10))):
    print("%4d   %.8f   %.8f   %.8f   %.2e / %.2e" % (i, a[i], a2[i], b[i], abs(a2[i] - a[i]) / abs(a[i]), abs(b[i] - a[i]) / abs(a[i])))
print("mean loss over the last 10 steps: fp32-MFMA %.6f, again %.6f, bf16-split %.6f" % (a[-10:].mean(), a2[-10:].mean(), b[-10:].mean()))
