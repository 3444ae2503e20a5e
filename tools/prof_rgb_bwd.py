"""Phase profile of rgb_composite_bwd_kernel<FUSED> (tuning tool): s_memtime stamps of one iteration of one workgroup, chain wave 0 and
weight-gradient wave 4, averaged over a few (workgroup, iteration) samples of a bs32 training render.
    bash tools/build_variants.sh rgb_bwd.hip SC_RGBB_PROFILE 1
    SHAPECLIPPER_HIP_LIB=$PWD/shapeclipper_amd/lib/variants/lib_SC_RGBB_PROFILE_1.so python tools/prof_rgb_bwd.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import _lib
from shapeclipper_amd.utils import options, camera
from shapeclipper_amd.model.implicit import SDFNetwork, RGBNetwork
from shapeclipper_amd.model.renderer import Renderer

dev = torch.device("cuda:0")
yaml = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "options/pix3d/config.yaml")
opt = options.set(options.parse_arguments(["--yaml=" + yaml, "--name=perf", "--output_root=/tmp/sc_perf"]), verbose=False)
torch.manual_seed(0)
r = Renderer(opt, SDFNetwork(opt), RGBNetwork(opt)).to(dev)
B, R = 32, 512
az = (torch.rand(B) * 2 - 1) * 3.14159
trig = lambda t: torch.stack([torch.cos(t), torch.sin(t)], 1)
Ry = camera.azim_to_rotation_matrix(trig(az), "trig"); Rx = camera.elev_to_rotation_matrix(trig(torch.zeros(B)), "trig")
P = torch.tensor([[-1., 0, 0], [0, 0, -1], [0, -1, 0]])[None].expand(B, 3, 3)
pose = camera.pose.compose([camera.pose(R=Rx @ Ry @ P), camera.pose(t=torch.tensor([[0., 0, 5.]]).expand(B, 3))]).to(dev).requires_grad_(True)
intr = camera.get_intr(opt, torch.ones(B)).to(dev)
sd = torch.ones(B, device=dev, requires_grad=True)
zs = torch.randn(B, 64, device=dev, requires_grad=True); zr = torch.randn(B, 64, device=dev, requires_grad=True)
ray_idx = torch.stack([torch.randperm(224 * 224)[:R] for _ in range(B)]).to(dev)
lib = _lib.load()._cdll
prof = torch.zeros(8 * 64, dtype=torch.int64, device=dev)


def step(block, it):
    assert lib.sc_rgbb_set_prof(ctypes.c_void_p(prof.data_ptr()), ctypes.c_int(block), ctypes.c_int(it)) == 0
    out = r(opt, pose, intr, sd, zs, zr, ray_idx=ray_idx, training=True)
    (out[0].sum() + out[1].sum() + out[4].sum() + ((out[5] - 1) ** 2).mean()).backward()
    torch.cuda.synchronize()


for _ in range(2):
    step(-1, -1)
names = ["inputs + PE", "forward chain (240 MFMA)", "output layer, dV3, Gy2", "x1: wait B1", "x1: write", "x1: wait B2", "V2^T (64) + mask",
         "x2: wait B1", "x2: write", "x2: wait B2", "V1^T (64) + mask", "x3: wait B1", "x3: write", "x3: wait B2", "V0f^T (64) + G feat store",
         "Jacobian (48) + G point"]
acc, n = None, 0
for blk, it in [(37, 3), (100, 7), (201, 11), (5, 14), (77, 1), (150, 5), (250, 9), (128, 12)]:
    prof.zero_()
    step(blk, it)
    t = prof.cpu().view(8, 64)
    c = t[0]
    if c[0] == 0 or c[62] == 0:
        continue
    row = [int(c[1] - c[0])]                                     # phase 1
    for k in range(4):
        s = [int(c[2 + 15 * k + j]) for j in range(15)]
        end = int(c[2 + 15 * (k + 1)]) if k < 3 else int(c[62])
        # +0 is taken after the tile's inputs were requested and its PE evaluated; for k > 0 that interval is inside the previous tile's
        # last segment (Jacobian + G point + next tile's inputs), so "inputs + PE" is only separate for tile 0
        seg = [s[0] - int(c[1]) if k == 0 else 0] + [s[j] - s[j - 1] for j in range(1, 15)] + [end - s[14]]
        row.extend(seg)
    row.append(int(c[62] - c[0]))
    w = t[4]
    wrow = []
    for q in range(12):
        wrow.append(int(w[2 + 2 * q] - w[1 + 2 * q]))             # MFMA part of a consume
    wrow.append(int(w[24] - w[0]) if w[24] and w[0] else 0)
    acc = [a + b for a, b in zip(acc, row + wrow)] if acc else row + wrow
    n += 1
acc = [a / n for a in acc]
print("samples: %d   (s_memtime ticks)" % n)
print("chain wave 0, one iteration (one ray = 4 tiles): total %.0f" % acc[1 + 64])
print("  phase 1 (compositing backward, lane = sample): %.0f" % acc[0])
for j, nm in enumerate(names):
    v = [acc[1 + 16 * k + j] for k in range(4)]
    print("  %-28s per tile %s   sum %.0f" % (nm, " ".join("%6.0f" % x for x in v), sum(v)))
wr = acc[1 + 64 + 1:]
print("weight-gradient wave 4: MFMA part of its 12 consumes: %s; first barrier to last consume %.0f" % (" ".join("%.0f" % x for x in wr[:12]), wr[12]))
