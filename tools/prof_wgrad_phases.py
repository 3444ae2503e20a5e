"""Where a launch of the split weight-gradient kernel spends its time: s_memrealtime stamps of every workgroup (profile build:
tools/build_variants.sh conv3x3_wgrad.hip SC_WGRAD_PROFILE 1).
    SHAPECLIPPER_HIP_LIB=shapeclipper_amd/lib/variants/lib_SC_WGRAD_PROFILE_1.so python tools/prof_wgrad_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeclipper_amd import ops, _lib

lib = _lib.load()._cdll
buf = torch.zeros(16 * 1024, dtype=torch.int64, device="cuda")
for B, side, c in ((64, 14, 256), (64, 7, 512), (64, 28, 128), (64, 56, 64), (96, 14, 256)):
    x = torch.randn(B, c, side, side, device="cuda"); gy = torch.randn(B, c, side, side, device="cuda")
    for _ in range(3): ops.conv3x3_backward_weight(gy, x, split=True)
    torch.cuda.synchronize()
    buf.zero_()
    assert lib.sc_wgrad_debug_set_prof(ctypes.c_void_p(buf.data_ptr())) == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.conv3x3_backward_weight(gy, x, split=True); e.record()
    torch.cuda.synchronize()
    lib.sc_wgrad_debug_set_prof(ctypes.c_void_p(0))
    t = buf.cpu().numpy().astype("int64").reshape(-1, 16)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    def stat(name, a):
        a = a.astype("float64")
        print("   %-40s n=%3d  min %7.1f  mean %7.1f  max %7.1f" % (name, len(a), a.min(), a.mean(), a.max()))
    print("B=%d %dx%d %d ch: event time (kernel + reduce) %.1f us, workgroups %d" % (B, side, side, c, s.elapsed_time(e) * 1e3, len(t)))
    stat("start (us after first wg)", (t[:, 0] - t0) / 100.0)
    stat("zero fill + first loads issued", (t[:, 1] - t[:, 0]) / 100.0)
    stat("K-steps", t[:, 6])
    stat("staging per K-step (wait, split, ds_write, barrier)", t[:, 4] / 100.0 / t[:, 6])
    stat("MFMA loop per K-step (+ load issue, barrier)", t[:, 5] / 100.0 / t[:, 6])
    stat("K loop total", (t[:, 2] - t[:, 1]) / 100.0)
    stat("partial store", (t[:, 3] - t[:, 2]) / 100.0)
    stat("workgroup ends at", (t[:, 3] - t0) / 100.0)
