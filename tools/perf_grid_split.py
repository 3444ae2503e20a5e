"""Round-6 go / no-go of the pre-split bf16x3 value chain (csrc/sdf_value_split.hip) against the fp32-MFMA chain (sdf_fwd.hip, value only)
on the evaluation grid (vox_res = 100: 1,030,301 points per image): time of both, largest difference between them, and both against a
float64 evaluation of the same network on a sample of the grid.   python tools/perf_grid_split.py [n_images=1]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import ops, packing
from shapeclipper_amd.model.implicit import SDFNetwork
from shapeclipper_amd.utils import options

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt = options.set(options.parse_arguments(["--yaml=%s/options/pix3d/config.yaml" % ROOT, "--name=perf", "--output_root=/tmp/sc_perf", "--tb!",
                                           "--arch.enc_pretrained!"]), verbose=False)
torch.manual_seed(0)
net = SDFNetwork(opt).cuda()
z = torch.randn(B, 64, device="cuda") * 0.5
with torch.no_grad():
    w_pack, cbias = net.packed(z)
w_pack, cbias = w_pack.contiguous(), cbias.contiguous()
N = 101


def timed(split, iters=20):
    for _ in range(3):
        out = ops.sdf_grid_forward(w_pack, cbias, -0.6, 0.6, N, True, split=split)
    torch.cuda.synchronize()
    best, tot = 1e9, 0.0
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = ops.sdf_grid_forward(w_pack, cbias, -0.6, 0.6, N, True, split=split)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        best, tot = min(best, ms), tot + ms
    return out, tot / iters, best


for rep in range(2):
    a, ms_a, best_a = timed(False)
    b, ms_b, best_b = timed(True)
    print("rep %d, %d image(s) x %d points: fp32 MFMA chain %.3f ms (best %.3f)   pre-split bf16x3 chain %.3f ms (best %.3f)   ratio %.2f"
          % (rep, B, N ** 3, ms_a, best_a, ms_b, best_b, ms_a / ms_b), flush=True)
print("max |split - fp32| = %.3e   (max |level| %.3f)" % (float((a - b).abs().max()), float(a.abs().max())))
# float64 evaluation of the same network (stock operators) on a sample of the grid
ax = torch.linspace(-0.6, 0.6, N, device="cuda")
idx = torch.randint(0, N, (20000, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
pts = torch.stack([ax[idx[:, 0]], ax[idx[:, 1]], ax[idx[:, 2]]], 1)
from oracle import reference_ops as R            # checker only (this is a measurement tool, not the product path)
Ws = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
ref = R.sdf_mlp(R.Cfg(), Ws, pts.double().cpu(), z[:1].double().cpu().repeat(pts.shape[0], 1))[:, 0]
for name, lv in (("fp32 MFMA", a), ("split", b)):
    got = lv[0][idx[:, 0], idx[:, 1], idx[:, 2]].double().cpu()
    print("%-10s vs float64 oracle on 20,000 grid points: max abs err %.3e" % (name, float((got - ref).abs().max())))
