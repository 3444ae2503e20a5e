"""Stem tail (BatchNorm + ReLU + 3x3/2 max-pool on [N, 64, 112, 112]) forward / backward timed with events, and a digest of every output
(compare across libraries selected with SHAPECLIPPER_HIP_LIB: the variants must be bit-identical)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeclipper_amd import ops
torch.manual_seed(0)
for N, G in ((64, 2), (96, 3)):
    x = torch.randn(N, 64, 112, 112, device="cuda")
    x[0, 0, 0, :4] = float("nan"); x[1, 1, 5, 7:11] = 0.25                                  # a NaN window and exact ties
    g = torch.rand(64, device="cuda") - 0.3; b = torch.randn(64, device="cuda") * 0.1     # some negative scales
    dy = torch.randn(N, 64, 56, 56, device="cuda")
    def mk():
        return torch.zeros(64, device="cuda"), torch.ones(64, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")
    def fwd():
        return ops.bn_relu_pool_forward(x, g, b, *mk(), True, 0.1, 1e-5, G)
    y, idx, st = fwd()
    def bwd():
        return ops.bn_relu_pool_backward(dy, idx, x, g, b, st, True, G)
    outs = [y, idx] + [t for t in bwd()]
    h = hashlib.sha256()
    for t in outs:
        h.update(t.detach().cpu().numpy().tobytes())
    ref = os.environ.get("SC_POOL_REF")            # first run writes the outputs there, later runs compare with them
    if ref:
        f = "%s_%d.pt" % (ref, N)
        if os.path.exists(f):
            old = torch.load(f)
            print("  vs %s: " % f + ", ".join("%s %.3g (scale %.3g)" % (n, float((o.float() - t.cpu().float()).abs().nan_to_num().max()), float(o.float().abs().nan_to_num().max()))
                                            for n, o, t in zip(("y", "idx", "dx", "dgamma", "dbeta"), old, outs)))
        else:
            torch.save([t.cpu() for t in outs], f)
    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    print("N=%d G=%d: forward (stats + pool) %.1f us, backward (2 gathers) %.1f us, digest %s" % (N, G, timeit(fwd), timeit(bwd), h.hexdigest()[:16]), flush=True)
