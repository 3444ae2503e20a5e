import ctypes, os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/shapeclipper_amd") else os.getcwd())
import torch
from shapeclipper_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
def t(epi, M, N, K):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (0, 1) else torch.bfloat16)
    run = lambda: lib.sc_gemm_bf16(ctypes.c_int(epi), _lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), ctypes.c_int(M), ctypes.c_int(N), ctypes.c_int(K), _lib.stream())
    for _ in range(3): run()
    torch.cuda.synchronize()
    evs = []
    for _ in range(15):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[7] * 1e3
for epi in (3, 0):
    for (M, N) in ((12800, 2304), (4096, 4096), (32768, 1024)):
        row = []
        for K in (64, 768, 1536, 3072, 6144):
            us = t(epi, M, N, K)
            row.append("K=%d: %.1f us (%.0f TF)" % (K, us, 2.0 * M * N * K / us / 1e6))
        print("epi %d M=%d N=%d  " % (epi, M, N) + "  ".join(row))
