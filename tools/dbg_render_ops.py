"""Debug aid: one training render (B=4, R=512) twice; every ops.* call's tensor outputs are recorded and compared run to run, and saved
(python tools/dbg_render_ops.py out.pt [ref.pt]) for comparison with another library build (SHAPECLIPPER_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import ops
from shapeclipper_amd.utils import options, camera
from shapeclipper_amd.model.implicit import SDFNetwork, RGBNetwork
from shapeclipper_amd.model.renderer import Renderer
dev = torch.device("cuda:0")
opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=dbg", "--output_root=/tmp/sc_dbg"]), verbose=False)
torch.manual_seed(0)
sdf, rgb = SDFNetwork(opt), RGBNetwork(opt)
r = Renderer(opt, sdf, rgb).to(dev)
B, R = 4, 512
az = (torch.rand(B) * 2 - 1) * 3.14159
trig = lambda t: torch.stack([torch.cos(t), torch.sin(t)], 1)
Ry = camera.azim_to_rotation_matrix(trig(az), "trig"); Rx = camera.elev_to_rotation_matrix(trig(torch.zeros(B)), "trig")
P = torch.tensor([[-1., 0, 0], [0, 0, -1], [0, -1, 0]])[None].expand(B, 3, 3)
pose = camera.pose.compose([camera.pose(R=Rx @ Ry @ P), camera.pose(t=torch.tensor([[0., 0, 5.]]).expand(B, 3))]).to(dev).requires_grad_(True)
intr = camera.get_intr(opt, torch.ones(B)).to(dev)
sd = torch.ones(B, device=dev, requires_grad=True)
zs = torch.randn(B, 64, device=dev, requires_grad=True); zr = torch.randn(B, 64, device=dev, requires_grad=True)
ray_idx = torch.stack([torch.randperm(224 * 224)[:R] for _ in range(B)]).to(dev)
log = []
def wrap(name):
    fn = getattr(ops, name)
    def f(*a, **k):
        out = fn(*a, **k)
        flat = out.values() if isinstance(out, dict) else (out if isinstance(out, (tuple, list)) else [out])
        log.append((name, [t.detach().clone().cpu() if torch.is_tensor(t) else None for t in flat]))
        return out
    setattr(ops, name, f)
for n in ("sdf_forward", "rgb_composite_forward", "rgb_composite_backward", "sdf_backward", "sdf_backward_fused"):
    if hasattr(ops, n):
        wrap(n)
import shapeclipper_amd.functional as F
state = torch.get_rng_state()
runs = []
for rep in range(2):
    torch.set_rng_state(state)
    log.clear()
    out = r(opt, pose, intr, sd, zs, zr, ray_idx=ray_idx, training=True)
    L = out[0].sum() + out[1].sum() + (out[4] * out[2]).sum() + ((out[5] - 1) ** 2).mean()
    for p_ in list(r.parameters()) + [pose, sd, zs, zr]:
        p_.grad = None
    L.backward()
    torch.cuda.synchronize()
    runs.append(list(log))
for (n0, t0), (n1, t1) in zip(*runs):
    print("%-26s run-to-run identical: %s" % (n0, [None if a is None else bool(torch.equal(a, b)) for a, b in zip(t0, t1)]))
torch.save(runs[0], sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for (n0, t0), (n1, t1) in zip(ref, runs[0]):
        print("%-26s vs ref: %s" % (n0, ["-" if a is None else "%.1e/%.1e" % (float((a - b).abs().max()), float(a.abs().max())) for a, b in zip(t0, t1)]))
