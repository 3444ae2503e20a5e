"""Phase profile of sdf_bwdw_kernel (tuning tool): s_memtime stamps of one workgroup iteration, chain wave 0 and wgrad wave 4.
Needs tools/micro/libbwdw_prof.so (sdf_bwdw.hip built with -DSC_BWDW_PROFILE, see tools/micro/Makefile).
    python tools/prof_bwdw.py [n_images=32]"""
import ctypes, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import ops, packing, _lib
from oracle import reference_ops as R          # weights only (tuning tool, not product)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
torch.manual_seed(0)
W = {k: v.to(dev) for k, v in R.init_sdf_weights(R.Cfg(), 0).items()}
z = torch.randn(B, 64, device=dev) * 0.3
pack, cb = packing.pack_sdf(W, z)
npi = 512 * 64
n = B * npi
pts = torch.rand(n, 3, device=dev) * 1.2 - 0.6
sdf, grad, feat, sa, sp = ops.sdf_forward(pts, pack, cb, npi, stash=True)
g_sdf, g_grad, g_feat = torch.randn(n, device=dev), torch.randn(n, 3, device=dev), torch.randn_like(feat) * 0.1

lib = ctypes.CDLL(os.environ.get("SC_BWDW_PROF_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libbwdw_prof.so"))
parts = lib.sc_sdf_backward_fused_parts(ctypes.c_int(n))
park = torch.empty(256 * 4 * 4 * 1024, device=dev)
partial = torch.empty(parts * lib.sc_sdf_backward_fused_partial_floats(ctypes.c_int(B)), device=dev)
g_c = torch.zeros(B, 5, 64, device=dev)
g_points = torch.empty(n, 3, device=dev)
prof = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
P = _lib.ptr


def launch(block, it):
    lib.sc_bwdw_set_prof(ctypes.c_void_p(prof.data_ptr()), ctypes.c_int(block), ctypes.c_int(it))
    code = lib.sc_sdf_backward_fused(P(pts), P(pack), ctypes.c_int(n), ctypes.c_int(npi), ctypes.c_int(B), ctypes.c_int(1), P(sa), P(sp),
                                     P(g_sdf), P(g_grad), P(g_feat), P(g_points), P(park), P(partial), P(g_c), _lib.stream())
    assert code == 0, code


for _ in range(2):
    launch(-1, -1)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); launch(-1, -1); e.record(); torch.cuda.synchronize()
print("launch (stamps off): %.3f ms" % s.elapsed_time(e))

samples = [(37, 10), (100, 20), (201, 33), (5, 40), (77, 50), (150, 5), (250, 25), (128, 60)]
acc = None
for blk, it in samples:
    prof.zero_()
    launch(blk, it)
    torch.cuda.synchronize()
    t = prof.cpu().view(8, 64).double()
    acc = t if acc is None else acc + 0  # keep last for absolute; deltas averaged below
    rows = []
    c = t[0]                                   # chain wave 0
    t0 = c[60]
    names = ["mm", "elem", "B1wait", "write+B2"]
    prev = t0
    chain = []
    for k in range(11):
        for j in range(4):
            chain.append(float(c[4 * k + j] - prev)); prev = c[4 * k + j]
    w = t[4]
    wg = []
    prevw = w[60]
    for k in range(23):
        wg.append(float(w[k] - prevw)); prevw = w[k]
    rows = (chain, wg, float(c[43] - t0), float(w[22] - w[60]), float(w[60] - t0))
    if 'tot' not in globals():
        tot = [list(rows[0]), list(rows[1]), rows[2], rows[3]]
        cnt = 1
    else:
        tot[0] = [a + b for a, b in zip(tot[0], rows[0])]
        tot[1] = [a + b for a, b in zip(tot[1], rows[1])]
        tot[2] += rows[2]; tot[3] += rows[3]; cnt += 1
    print("block %3d iter %2d: chain iteration %.0f cycles, wgrad iteration %.0f, wgrad start - chain start %.0f" % (blk, it, rows[2], rows[3], rows[4]))

print("\nchain wave 0, mean cycles per phase over %d samples (s_memtime ticks = shader cycles):" % cnt)
print("step    mm(+loads)   elem     B1-wait   write+B2   | step total")
for k in range(11):
    v = [x / cnt for x in tot[0][4 * k:4 * k + 4]]
    print("%4d  %9.0f %9.0f %9.0f %9.0f   | %9.0f" % (k, v[0], v[1], v[2], v[3], sum(v)))
sums = [sum(tot[0][j::4]) / cnt for j in range(4)]
print(" sum  %9.0f %9.0f %9.0f %9.0f   | %9.0f" % (sums[0], sums[1], sums[2], sums[3], sum(sums)))
print("\nwgrad wave 4: per step (wait at B1+B2, consume [+ point terms])")
for k in range(11):
    print("%4d  wait %8.0f   work %8.0f" % (k, tot[1][2 * k + 1] / cnt, (tot[1][2 * k + 2] if 2 * k + 2 < 23 else 0) / cnt))
print(" wait total %.0f   work total %.0f  (work of step k is measured from after B2 of step k to before B1 of step k+1)"
      % (sum(tot[1][1::2]) / cnt, (sum(tot[1][2::2]) + tot[1][0]) / cnt))
