"""Round-6 A/B of the RGB reverse pass: transposed products from pre-split bf16x3 fragments (sc_rgb_composite_backward_fused_split) against
fp32 MFMAs (sc_rgb_composite_backward_fused_stash), training shape (B x 512 rays).   python tools/perf_rgb_bwd_split.py [B=32]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import ops, packing
from oracle import reference_ops as R          # weights only (measurement tool)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = R.Cfg()
Ws = {k: v.to(dev) for k, v in R.init_sdf_weights(cfg, 1).items()}
Wr = {k: v.to(dev) for k, v in R.init_rgb_weights(cfg, 2).items()}
g = torch.Generator().manual_seed(0)
zs, zr = (torch.randn(B, 64, generator=g) * 0.3).to(dev), (torch.randn(B, 64, generator=g) * 0.3).to(dev)
sdf_pack, cb = packing.pack_sdf(Ws, zs)
rgb_pack, db = packing.pack_rgb(Wr, zr)
beta = torch.tensor([0.1], device=dev)
rpi = 512
n_rays = B * rpi
pts = (torch.rand(n_rays * 64, 3, device=dev) * 1.6 - 0.8)
z = torch.sort(torch.rand(n_rays, 64, device=dev) * 2 + 4, dim=1).values
dfac = torch.rand(n_rays, device=dev) * 0.2 + 0.9
sdf, grad, feat = ops.sdf_forward(pts, sdf_pack, cb, rpi * 64)
common = (pts, z, dfac, sdf, grad, feat, rgb_pack, db, beta, rpi, True, 1e-4, 1.0, 1.0)
a = ops.rgb_composite_forward(*common, keep_rgb_flat=True, keep_rr=True)
G = [torch.randn(n_rays, 3, device=dev), torch.randn(n_rays, device=dev), torch.randn(n_rays, device=dev), torch.randn(n_rays, 3, device=dev)]
back = lambda: ops.rgb_composite_backward(pts, z, dfac, sdf, grad, feat, rgb_pack, db, beta, a["rgb_flat"], rpi, True, 1e-4, 1.0, 1.0, *G, rr=a["rr"])
res = {}
for rep in range(3):
    for split in (False, True):
        ops.RGB_BWD_SPLIT = split
        for _ in range(2):
            out = back()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(10):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); out = back(); e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e))
        res[split] = (best, out)
    d = max(float((res[True][1][k] - res[False][1][k]).abs().max()) / max(float(res[False][1][k].abs().max()), 1e-9) for k in res[False][1])
    print("B=%d (%d rays) rep %d: fp32 MFMA %.3f ms   pre-split reverse chain %.3f ms   ratio %.2f   worst rel diff %.1e"
          % (B, n_rays, rep, res[False][0], res[True][0], res[False][0] / res[True][0], d), flush=True)
