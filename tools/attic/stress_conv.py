"""Determinism stress test of csrc/conv3x3.hip (stream-K tail: partial tiles + fix-up launch): many shapes, repeated launches on two
streams at once, every result compared bit for bit with the first one and against float64."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapeclipper_amd import ops

torch.manual_seed(0)
dev = torch.device("cuda:0")
side_stream = torch.cuda.Stream()
bad = 0
for split in (False, True):
    for side, ch, batch in ((56, 64, 64), (56, 64, 96), (28, 128, 64), (28, 128, 96), (14, 256, 64), (14, 256, 96), (7, 512, 64), (7, 512, 96),
                            (7, 512, 3), (14, 256, 5), (28, 128, 7), (56, 64, 9)):
        x = torch.randn(batch, ch, side, side, device=dev)
        w = torch.randn(ch, ch, 3, 3, device=dev) * 0.05
        wp = ops.conv3x3_pack(w, side, False, split)
        ref = ops.conv3x3_apply(x, wp, ch, split)
        y64 = torch.nn.functional.conv2d(x[:2].double(), w.double(), None, 1, 1)
        err = float((ref[:2].double() - y64).abs().max() / y64.abs().max())
        torch.cuda.synchronize()
        outs = []
        for it in range(30):
            outs.append(ops.conv3x3_apply(x, wp, ch, split))
            with torch.cuda.stream(side_stream):            # a second convolution kernel competing for the CUs
                outs.append(ops.conv3x3_apply(x, wp, ch, split))
        torch.cuda.synchronize()
        n_bad = sum(0 if torch.equal(o, ref) else 1 for o in outs)
        bad += n_bad + (err > 2e-5)
        print("side %2d ch %3d batch %2d split %d: error vs float64 %.2e, %d of %d repeated launches differ" % (side, ch, batch, split, err, n_bad, len(outs)), flush=True)
print("STRESS", "FAILED" if bad else "OK")
