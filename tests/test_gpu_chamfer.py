"""Parity of the HIP Chamfer3D kernels (through the C ABI / chamfer_3D module) with the oracle.

Bar: idx and dist bit-exact against oracle/chamfer_ref.c, which is itself pinned bit for bit against the reference's own
extension built for this GPU (oracle/_ref, tests/test_gpu_chamfer_ref.py): d = fmaf(dy, dy, dx*dx) + dz*dz."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run_hip(a, b):
    import chamfer_3D
    dev = torch.device("cuda:0")
    x1, x2 = torch.tensor(a, device=dev), torch.tensor(b, device=dev)
    B, N, M = x1.shape[0], x1.shape[1], x2.shape[1]
    d1 = torch.zeros(B, N, device=dev); d2 = torch.zeros(B, M, device=dev)
    i1 = torch.zeros(B, N, dtype=torch.int32, device=dev); i2 = torch.zeros(B, M, dtype=torch.int32, device=dev)
    assert chamfer_3D.forward(x1, x2, d1, d2, i1, i2) == 1
    torch.cuda.synchronize()
    return d1.cpu().numpy(), d2.cpu().numpy(), i1.cpu().numpy(), i2.cpu().numpy()


def test_golden_with_ties(golden):
    g = golden("g9_chamfer")
    d1, d2, i1, i2 = run_hip(g["xyz1"], g["xyz2"])
    assert np.array_equal(i1, g["idx1"]) and np.array_equal(i2, g["idx2"])
    assert np.array_equal(d1, g["dist1"]) and np.array_equal(d2, g["dist2"])


@pytest.mark.parametrize("B,N,M", [(1, 1, 1), (2, 17, 5), (3, 1000, 2049), (1, 4097, 1023), (2, 2048, 2048)])
def test_ragged_sizes_vs_oracle(B, N, M):
    from oracle import chamfer_ref
    rng = np.random.RandomState(N * 7 + M)
    a = rng.uniform(-0.5, 0.5, (B, N, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (B, M, 3)).astype(np.float32)
    if M > 40:
        b[:, 33] = b[:, 3]  # duplicate targets: lowest index must win
    r = chamfer_ref.chamfer_forward(a, b)
    h = run_hip(a, b)
    for x, y in zip(h, r):
        assert np.array_equal(x, y)


def test_self_distance_is_zero_and_identity_index():
    rng = np.random.RandomState(0)
    a = rng.uniform(-0.5, 0.5, (1, 5000, 3)).astype(np.float32)
    d1, d2, i1, i2 = run_hip(a, a)
    assert np.all(d1 == 0) and np.all(d2 == 0)
    assert np.array_equal(i1[0], np.arange(5000)) and np.array_equal(i2[0], np.arange(5000))


def test_empty_leaves_outputs_untouched():
    import chamfer_3D
    dev = torch.device("cuda:0")
    x1 = torch.zeros(1, 0, 3, device=dev); x2 = torch.rand(1, 8, 3, device=dev)
    d1 = torch.zeros(1, 0, device=dev); d2 = torch.full((1, 8), 7.0, device=dev)
    i1 = torch.zeros(1, 0, dtype=torch.int32, device=dev); i2 = torch.full((1, 8), 5, dtype=torch.int32, device=dev)
    chamfer_3D.forward(x1, x2, d1, d2, i1, i2)
    torch.cuda.synchronize()
    assert torch.all(d2 == 7.0) and torch.all(i2 == 5)


def test_backward_vs_oracle(golden):
    import chamfer_3D
    g = golden("g9_chamfer")
    dev = torch.device("cuda:0")
    t = lambda k, dt=None: torch.tensor(g[k], device=dev)
    g1 = torch.zeros(g["xyz1"].shape, device=dev); g2 = torch.zeros(g["xyz2"].shape, device=dev)
    assert chamfer_3D.backward(t("xyz1"), t("xyz2"), g1, g2, t("gd1"), t("gd2"), t("idx1"), t("idx2")) == 1
    torch.cuda.synchronize()
    # atomics reorder the fp32 sums: tolerance instead of bit equality
    np.testing.assert_allclose(g1.cpu().numpy(), g["g1"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g2.cpu().numpy(), g["g2"], rtol=1e-5, atol=1e-5)


def test_full_size_properties():
    """BASELINE size (N=M=100k): size-independent properties instead of the O(N*M) oracle."""
    rng = np.random.RandomState(1)
    a = rng.uniform(-0.5, 0.5, (1, 100000, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (1, 100000, 3)).astype(np.float32)
    d1, d2, i1, i2 = run_hip(a, b)
    # reported distance equals the distance to the reported index (same fma chain, in fp32)
    diff = b[0][i1[0]] - a[0]
    dd = np.float32(diff[:, 0]) * np.float32(diff[:, 0])                                   # fma(dy,dy,dx*dx) + dz*dz
    dd = np.float32(np.float64(diff[:, 1]) * np.float64(diff[:, 1]) + np.float64(dd))      # fma emulation
    dd = np.float32(dd + np.float32(diff[:, 2]) * np.float32(diff[:, 2]))
    assert np.array_equal(dd, d1[0])
    # no sampled target is closer than the reported minimum
    sub = rng.choice(100000, 2000, replace=False)
    D = ((a[0][:512, None, :].astype(np.float64) - b[0][None, sub, :]) ** 2).sum(-1)
    assert np.all(D.min(1) >= d1[0][:512] - 1e-7)
    # symmetry: chamfer(a,b).dist1 == chamfer(b,a).dist2
    e1, e2, j1, j2 = run_hip(b, a)
    assert np.array_equal(e2, d1) and np.array_equal(j2, i1)


@pytest.mark.parametrize("B,N,M,nsplit", [(1, 5000, 7001, 8), (2, 4099, 4100, 3), (1, 100, 50, 4), (1, 5000, 70001, 0), (3, 9000, 4100, 0)])
def test_split_entry_point_is_bit_identical(B, N, M, nsplit):
    import ctypes
    from oracle import chamfer_ref
    from shapeclipper_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(N)
    a = rng.uniform(-0.5, 0.5, (B, N, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (B, M, 3)).astype(np.float32)
    b[:, M // 2] = b[:, 3]; b[:, M - 1] = b[:, 3]                     # duplicates in different slices
    x1, x2 = torch.tensor(a, device=dev), torch.tensor(b, device=dev)
    d1 = torch.zeros(B, N, device=dev); d2 = torch.zeros(B, M, device=dev)
    i1 = torch.zeros(B, N, dtype=torch.int32, device=dev); i2 = torch.zeros(B, M, dtype=torch.int32, device=dev)
    ws = torch.empty(B * (N + M), dtype=torch.int64, device=dev)
    rc = lib.sc_chamfer3d_forward_split(_lib.ptr(x1), _lib.ptr(x2), _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(i1), _lib.ptr(i2),
                                        ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M), ctypes.c_int(nsplit), _lib.ptr(ws), _lib.stream())
    assert rc == 0
    torch.cuda.synchronize()
    r = chamfer_ref.chamfer_forward(a, b)
    for x, y in zip((d1, d2, i1, i2), r):
        assert np.array_equal(x.cpu().numpy(), y)
