"""Where do the parked activations of the split RGB forward differ from the fp32 form's?  (debug, round 6)"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import torch
from test_gpu_rgb_stash import _setup
s = _setup(3, 40, seed=5)
ops = s["ops"]
common = (s["pts"], s["z"], s["dfac"], s["sdf"], s["grad"], s["feat"], s["rgb_pack"], s["db"], s["beta"], s["rpi"], True, 1e-4, 1.0, 1.0)
ops.RGB_FWD_SPLIT = False
b = ops.rgb_composite_forward(*common, keep_rgb_flat=True, keep_rr=True)
ops.RGB_FWD_SPLIT = True
for rep in range(3):
    a = ops.rgb_composite_forward(*common, keep_rgb_flat=True, keep_rr=True)
    ra, rb = (t["rr"].view(3, s["n_rays"], 4, 16, 16, 4) for t in (a, b))          # [layer][ray][tile k][ch/4][pt][4]
    d = (ra - rb).abs()
    bad = d > 1e-5
    print("rep", rep, "bad elements", int(bad.sum()), "of", bad.numel(), " negatives", int((ra < 0).sum()), " max diff %.3e" % float(d.max()),
          " rgb diff %.2e" % float((a["rgb"] - b["rgb"]).abs().max()))
    for L in range(3):
        bl = bad[L]
        if int(bl.sum()) == 0:
            continue
        print("  layer", L, "bad", int(bl.sum()), " by tile k:", [int(bl[:, k].sum()) for k in range(4)], " by ch/4 block:", [int(bl[:, :, c].sum()) for c in range(16)],
              " by lane-group reg r:", [int(bl[..., r].sum()) for r in range(4)], " rays with bad:", int(bl.flatten(1).any(1).sum()))
        idx = bl.nonzero()[:6]
        for i in idx:
            i = tuple(int(v) for v in i)
            print("     ", i, "split %.6f fp32 %.6f" % (float(ra[L][i]), float(rb[L][i])))
