"""bf16 GEMM of the CLIP tower (sc_gemm_bf16) at the four layer shapes of ViT-B/32: python tools/perf_gemm.py [tokens=12800]"""
import ctypes, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import _lib
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
dev = torch.device("cuda")
for name, epi, N, K in (("qkv  (bf16 out)", 3, 2304, 768), ("proj (resid)", 1, 768, 768), ("fc1  (gelu)", 2, 3072, 768), ("fc2  (resid)", 1, 768, 3072)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (0, 1) else torch.bfloat16)
    run = lambda: lib.sc_gemm_bf16(ctypes.c_int(epi), _lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), ctypes.c_int(M), ctypes.c_int(N), ctypes.c_int(K), _lib.stream())
    for _ in range(3): run()
    torch.cuda.synchronize()
    evs = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)[len(evs) // 2]
    print("%-16s M=%d N=%d K=%d: %.1f us  %.0f TFLOP/s (%.1f%% of 2.5 PF)" % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, 2.0 * M * N * K / (ms * 1e-3) / 2.5e13))
