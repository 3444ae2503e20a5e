"""Where a stream-K convolution launch spends its time: s_memrealtime stamps of every workgroup of conv3x3_kernel / conv3x3_fixup_kernel
(profile build of conv3x3.hip: tools/build_variants.sh conv3x3.hip SC_CONV_PROFILE 1).
    SHAPECLIPPER_HIP_LIB=shapeclipper_amd/lib/variants/lib_SC_CONV_PROFILE_1.so python tools/prof_conv_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeclipper_amd import ops, _lib

lib = _lib.load()._cdll
buf = torch.zeros(65536, dtype=torch.int64, device="cuda")
for B, side, c in ((64, 14, 256), (64, 7, 512), (64, 28, 128), (64, 56, 64), (96, 14, 256)):
    x = torch.randn(B, c, side, side, device="cuda"); w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
    pk = ops.conv3x3_pack(w, side, False, True)
    for _ in range(3): ops.conv3x3_apply(x, pk, c, True)
    torch.cuda.synchronize()
    buf.zero_()
    assert lib.sc_conv_debug_set_prof(ctypes.c_void_p(buf.data_ptr())) == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.conv3x3_apply(x, pk, c, True); e.record()
    torch.cuda.synchronize()
    lib.sc_conv_debug_set_prof(ctypes.c_void_p(0))
    t = buf.cpu().numpy().astype("int64")
    main = t[:16384].reshape(512, 32)
    main = main[main[:, 0] > 0]
    t0 = main[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    def stat(name, a):
        a = a.astype("float64")
        print("   %-34s n=%3d  min %6.1f  mean %6.1f  max %6.1f" % (name, len(a), a.min(), a.mean(), a.max()))
    print("B=%d %dx%d %d ch: event time %.1f us, workgroups %d" % (B, side, side, c, s.elapsed_time(e) * 1e3, len(main)))
    stat("start skew (us after first wg)", us(main[:, 0]))
    for k, nm in ((1, "whole tile"), (9, "shared tile 1"), (17, "shared tile 2")):
        m = main[main[:, k] > 0]
        if not len(m): continue
        print("  item: %s" % nm)
        stat("begins at", us(m[:, k]))
        stat("prologue (first stage staged)", (m[:, k + 1] - m[:, k]) / 100.0)
        if (m[:, k + 5] > 0).any():
            mm = m[m[:, k + 5] > 0]
            stat("first K-step", (mm[:, k + 5] - mm[:, k + 1]) / 100.0)
        stat("K loop", (m[:, k + 2] - m[:, k + 1]) / 100.0)
        stat("K-steps", m[:, k + 4])
        stat("us per K-step", (m[:, k + 2] - m[:, k + 1]) / 100.0 / m[:, k + 4])
        stat("store (to vmcnt 0)", (m[:, k + 3] - m[:, k + 2]) / 100.0)
    stat("workgroup ends at", us(main[:, 25]))
    fx = t[16384:16384 + 2 * 4096].reshape(-1, 2)
    fx = fx[fx[:, 0] > 0]
    if len(fx):
        stat("fixup wg begins at", us(fx[:, 0]))
        stat("fixup wg duration", (fx[:, 1] - fx[:, 0]) / 100.0)
        stat("fixup wg ends at", us(fx[:, 1]))
