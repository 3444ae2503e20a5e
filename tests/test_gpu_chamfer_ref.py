"""Chamfer3D against the REFERENCE ITSELF: external/chamfer3D (chamfer_cuda.cpp + chamfer3D.cu) built for gfx950 from
/root/reference by oracle/build_chamfer_ref.py into oracle/_ref/chamfer_3D_ref.so ("outputs of the reference run here").
This pins, on this hardware and compiler, what the restated oracle (oracle/chamfer_ref.c) and the HIP kernels are compared with:
squared distances and nearest-neighbour indices bit for bit -- including which products of chamfer3D.cu:35 the compiler fuses and
which index wins an exact tie -- and the gradients of chamfer3D.cu:155-195 (float atomics: order-dependent rounding, so 1e-6)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    # the strongest Chamfer evidence must not evaporate silently.  The binary is git-ignored and only exists where
    # __graft_entry__.build() ran next to /root/reference; a GPU box without it FAILS here (built on demand when the reference is present).
    from oracle import build_chamfer_ref
    mod = build_chamfer_ref.load_or_build()
    if mod is None:
        pytest.fail("oracle/_ref/chamfer_3D_ref.so is missing and /root/reference is not here to build it from: the reference-kernel "
                    "pin of Chamfer3D did not run (run __graft_entry__.build() in the build container before shipping the tree)")
    return mod


def _clouds(B, N, M, seed, kind):
    rng = np.random.RandomState(seed)
    if kind == "grid":        # points on a coarse lattice: thousands of exact ties and exact zero distances
        a = rng.randint(-4, 5, (B, N, 3)).astype(np.float32) * 0.125
        b = rng.randint(-4, 5, (B, M, 3)).astype(np.float32) * 0.125
    else:
        a = rng.uniform(-0.5, 0.5, (B, N, 3)).astype(np.float32)
        b = rng.uniform(-0.5, 0.5, (B, M, 3)).astype(np.float32)
        if M > 40:
            b[:, 33] = b[:, 3]
    return a, b


def _forward(mod, a, b):
    dev = torch.device("cuda:0")
    x1, x2 = torch.tensor(a, device=dev), torch.tensor(b, device=dev)
    B, N, M = x1.shape[0], x1.shape[1], x2.shape[1]
    d1, d2 = torch.zeros(B, N, device=dev), torch.zeros(B, M, device=dev)
    i1, i2 = torch.zeros(B, N, dtype=torch.int32, device=dev), torch.zeros(B, M, dtype=torch.int32, device=dev)
    assert mod.forward(x1, x2, d1, d2, i1, i2) == 1
    torch.cuda.synchronize()
    return x1, x2, d1, d2, i1, i2


@pytest.mark.parametrize("B,N,M,kind", [(2, 1000, 700, "uniform"), (1, 7001, 3000, "uniform"), (4, 512, 2048, "uniform"), (3, 17, 5, "uniform"),
                                        (2, 3000, 2500, "grid"), (1, 100000, 100000, "uniform")])
def test_forward_bit_exact_with_the_reference_build(ref, B, N, M, kind):
    import chamfer_3D
    a, b = _clouds(B, N, M, N + M, kind)
    r = _forward(ref, a, b)
    h = _forward(chamfer_3D, a, b)
    for name, x, y in zip(("dist1", "dist2", "idx1", "idx2"), h[2:], r[2:]):
        same = torch.equal(x, y)
        if not same and x.dtype == torch.float32:
            print(name, "max ulp-scale difference", float(((x - y).abs() / y.abs().clamp_min(1e-30)).max()))
        assert same, "%s differs from the reference build (%d of %d entries)" % (name, int((x != y).sum()), x.numel())
    if N * M <= 3000 * 7001:          # and the restated CPU oracle is what the reference computes
        from oracle import chamfer_ref
        o = chamfer_ref.chamfer_forward(a, b)
        for name, x, y in zip(("dist1", "dist2", "idx1", "idx2"), o, r[2:]):
            assert np.array_equal(x, y.cpu().numpy()), "oracle %s differs from the reference build" % name


@pytest.mark.parametrize("B,N,M", [(2, 1000, 700), (1, 20000, 3000)])
def test_backward_matches_the_reference_build(ref, B, N, M):
    import chamfer_3D
    a, b = _clouds(B, N, M, 3 * N + M, "uniform")
    x1, x2, d1, d2, i1, i2 = _forward(ref, a, b)
    g = torch.Generator(device="cpu").manual_seed(N)
    gd1, gd2 = torch.randn(B, N, generator=g).cuda(), torch.randn(B, M, generator=g).cuda()
    out = []
    for mod in (ref, chamfer_3D):
        g1, g2 = torch.zeros_like(x1), torch.zeros_like(x2)
        assert mod.backward(x1, x2, g1, g2, gd1, gd2, i1, i2) == 1
        torch.cuda.synchronize()
        out.append((g1, g2))
    for name, x, y in zip(("gradxyz1", "gradxyz2"), out[1], out[0]):
        err = float((x - y).abs().max() / y.abs().max())
        print("chamfer backward vs the reference build, %s: %.2e of max" % (name, err))
        assert err < 1e-6
