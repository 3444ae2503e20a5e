"""Host cost of the small torch GEMMs (heads / latent projectors: ~83 aten::mm per step) by BLAS backend: B=8 step time (host-paced)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
import bench
from shapeclipper_amd.utils.util import EasyDict as edict
runner, opt, batch = bench.build_runner(int(os.environ.get("B", "8")))
def step():
    opt.H, opt.W = opt.image_size
    return runner.train_iteration(opt, edict(batch), None)
def timed(n=30):
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
print("default backend (%s): %.2f ms / step" % (torch.backends.cuda.preferred_blas_library(), timed()))
for lib in ("cublas", "cublaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
        print("preferred_blas_library(%s): %.2f ms / step" % (lib, timed()))
    except Exception as e:
        print(lib, "failed:", e)
x = torch.randn(96, 512, device="cuda"); w = torch.randn(512, 512, device="cuda")
for lib in ("cublas", "cublaslt"):
    torch.backends.cuda.preferred_blas_library(lib)
    for _ in range(10): torch.nn.functional.linear(x, w)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(2000): torch.nn.functional.linear(x, w)
    h = time.time() - t0; torch.cuda.synchronize(); d = time.time() - t0
    print("%s: linear 96x512x512 host %.1f us / call, with GPU %.1f us / call" % (lib, h / 2000 * 1e6, d / 2000 * 1e6))
