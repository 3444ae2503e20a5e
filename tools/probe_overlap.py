"""Do an MFMA-bound convolution kernel and an HBM-bound BatchNorm kernel of two HIP streams run concurrently on this chip?
Times N launches of each alone and the same launches enqueued on two streams: perfect overlap = max of the two, none = their sum."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeclipper_amd import ops

def run(side, c, B, Bbn, cbn, sbn, n_conv=20, bn_per_conv=2):
    x = torch.randn(B, c, side, side, device="cuda"); w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
    pk = ops.conv3x3_pack(w, side, False, True)
    xb = torch.randn(Bbn, cbn, sbn, sbn, device="cuda")
    g = torch.ones(cbn, device="cuda"); b = torch.zeros(cbn, device="cuda")
    rm = torch.zeros(cbn, device="cuda"); rv = torch.ones(cbn, device="cuda"); nt = torch.zeros((), dtype=torch.int64, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def conv():
        for _ in range(n_conv): ops.conv3x3_apply(x, pk, c, True)
    def bn():
        for _ in range(n_conv * bn_per_conv): ops.bn_act_forward(xb, None, g, b, rm, rv, nt, True, 0.1, 1e-5, True, 2)
    def timed(fa, fb):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if fa:
            with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
    for _ in range(2): timed(conv, bn)
    tc = min(timed(conv, None) for _ in range(3)); tb = min(timed(None, bn) for _ in range(3)); tt = min(timed(conv, bn) for _ in range(3))
    print("conv %dx%d %dch B=%d x%d: %.2f ms | bn %dx%d %dch B=%d x%d: %.2f ms | both streams: %.2f ms (sum %.2f, max %.2f)" % (
        side, side, c, B, n_conv, tc, sbn, sbn, cbn, Bbn, n_conv * bn_per_conv, tb, tt, tc + tb, max(tc, tb)))

run(14, 256, 64, 96, 64, 56)
run(56, 64, 64, 96, 128, 28)
run(28, 128, 96, 64, 64, 56)
run(7, 512, 64, 96, 256, 14, bn_per_conv=4)
