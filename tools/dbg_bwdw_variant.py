"""Debug aid: outputs of sc_sdf_backward_fused of the library selected by SHAPECLIPPER_HIP_LIB, saved for comparison.
python tools/dbg_bwdw_variant.py <out.pt> [B]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import ops, packing
from oracle import reference_ops as R
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda")
torch.manual_seed(0)
W = {k: v.to(dev) for k, v in R.init_sdf_weights(R.Cfg(), 0).items()}
z = torch.randn(B, 64, device=dev) * 0.3
pack, cb = packing.pack_sdf(W, z)
npi = int(os.environ.get("NPI", 512 * 64))
n = B * npi
pts = torch.rand(n, 3, device=dev) * 1.2 - 0.6
sdf, grad, feat, sa, sp = ops.sdf_forward(pts, pack, cb, npi, stash=True)
g_sdf, g_grad, g_feat = torch.randn(n, device=dev), torch.randn(n, 3, device=dev), torch.randn_like(feat) * 0.1
if os.environ.get("EIK"):
    g_sdf, g_feat = None, None
outs = []
for rep in range(2):
    gp, gw, gc = ops.sdf_backward(pts, pack, npi, B, True, sa, sp, g_sdf, g_grad, g_feat, fused=True)
    outs.append([t.clone().cpu() for t in (gp, gw, gc)])
torch.cuda.synchronize()
print("run-to-run identical:", [bool(torch.equal(a, b)) for a, b in zip(outs[0], outs[1])])
print("finite:", [bool(torch.isfinite(t).all()) for t in outs[0]], "fwd finite", bool(torch.isfinite(sdf).all()), bool(torch.isfinite(sa).all()), bool(torch.isfinite(sp).all()))
torch.save(dict(fwd=[t.cpu() for t in (sdf, grad, feat, sa, sp)], bwd=outs[0]), sys.argv[1])
if len(sys.argv) > 3:
    ref = torch.load(sys.argv[3])
    for name, a, b in zip(("sdf", "grad", "feat", "stash_a", "stash_p"), ref["fwd"], (sdf, grad, feat, sa, sp)):
        print("fwd %-8s max|diff| %.3g of max %.3g" % (name, (a - b.cpu()).abs().max(), a.abs().max()))
    for name, a, b in zip(("g_points", "g_w", "g_cbias"), ref["bwd"], outs[0]):
        d = (a - b).abs()
        print("bwd %-8s max|diff| %.3g of max %.3g; worst index %d of %d" % (name, d.max(), a.abs().max(), int(d.flatten().argmax()), d.numel()))
    gw_r, gw = ref["bwd"][1], outs[0][1]
    from shapeclipper_amd.packing import SDF_OFF
    for k, o in SDF_OFF.items():
        pass
    print("SDF_OFF", SDF_OFF)
    d = (gw_r - gw).abs()
    keys = sorted(SDF_OFF.items(), key=lambda kv: kv[1])
    for i, (k, o) in enumerate(keys):
        e = keys[i + 1][1] if i + 1 < len(keys) else d.numel()
        print("  %-4s max|diff| %.3g of max %.3g" % (k, d[o:e].max(), gw_r[o:e].abs().max()))
