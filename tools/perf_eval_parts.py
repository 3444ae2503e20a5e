import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from shapeclipper_amd.utils import eval_3D
dev = "cuda"
def timed(name, fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print("%8.3f ms  %s" % ((time.time() - t0) / n * 1e3, name)); return r
B, N = 1, 100000
P = torch.randn(B, N, 3, device=dev); R = torch.linalg.qr(torch.randn(B, 3, 3, device=dev))[0]
rot = lambda Rm, P: (Rm @ P.permute(0, 2, 1)).permute(0, 2, 1).contiguous()
timed("rot (Rm @ P^T)^T", lambda: rot(R, P))
timed("rot as P @ Rm^T", lambda: (P @ R.transpose(1, 2)).contiguous())
timed("rot elementwise", lambda: (P[:, :, None, :] * R[:, None, :, :]).sum(-1))
timed("normalize_pc", lambda: eval_3D.normalize_pc(P))
class O: pass
X1, X2 = eval_3D.normalize_pc(P), eval_3D.normalize_pc(torch.randn(B, N, 3, device=dev))
d = timed("chamfer_distance", lambda: eval_3D.chamfer_distance(None, X1, X2))
timed("compute_fscore", lambda: eval_3D.compute_fscore(d[0], d[1], [0.005, 0.01, 0.02, 0.05, 0.1, 0.2]))
timed("tensor(_FLIP).to(dev)", lambda: torch.tensor(eval_3D._FLIP_PRED).float().to(dev).unsqueeze(0).expand(B, 3, 3))
