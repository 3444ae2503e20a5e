"""Round-6 go / no-go of the streamed pre-split SDF forward (csrc/sdf_fwd_stream.hip) against sdf_fwd.hip (fp32 MFMA): evaluation form
(value + feature + d sdf/dx, per-wave scratch) and training form (stashes), values compared.   python tools/perf_sdf_stream.py [points=8388608]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import ops, packing
from oracle import reference_ops as R          # weights only (measurement tool)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8 * 128 * 128 * 64
B = 8
dev = torch.device("cuda:0")
cfg = R.Cfg()
torch.manual_seed(0)
Ws = {k: (v + 0.02 * torch.randn_like(v)).to(dev) for k, v in R.init_sdf_weights(cfg, 1).items()}
zs = (torch.randn(B, 64) * 0.3).to(dev)
pack, cb = packing.pack_sdf(Ws, zs)
pts = (torch.rand(N, 3, device=dev) * 1.6 - 0.8)
npi = N // B


def run(stream, stash):
    ops.SDF_FWD_STREAM = stream
    out = ops.sdf_forward(pts, pack, cb, npi, want_grad=True, want_feat=True, stash=stash)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out = ops.sdf_forward(pts, pack, cb, npi, want_grad=True, want_feat=True, stash=stash); e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best, out


for stash in (False, True):
    for rep in range(2):
        ta, a = run(False, stash)
        tb, b = run(True, stash)
        names = ["sdf", "grad", "feat", "stash_a", "stash_p"][:len(a)]
        diffs = {n: "%.2e" % float((x - y).abs().max() / max(float(x.abs().max()), 1e-9)) for n, x, y in zip(names, a, b)}
        print("%s, %d points rep %d: fp32 MFMA %.3f ms   streamed pre-split %.3f ms   ratio %.2f   rel diffs %s"
              % ("training (stashes)" if stash else "evaluation", N, rep, ta, tb, ta / tb, diffs), flush=True)
ops.SDF_FWD_STREAM = True
