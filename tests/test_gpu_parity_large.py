"""Oracle parity at sizes where the kernels' persistent loops wrap (VERDICT r1 "weak #1").

The chain kernels launch at most 256 workgroups x 8 waves x 16 points = 32,768 points per sweep of the grid, so
every test here runs >= 4 sweeps with n_per_image = 32,768 (per-image bias / latent-gradient boundaries inside a
sweep), against the CPU oracle (oracle/reference_ops.py, pinned to the reference by tests/golden/make_golden.py) on
the same inputs and the same CPU-RNG draws.  The shape is G12's (rays really hit it).  Every test prints its errors.

  * training render B=4 x R=512 (131,072 points + 4,096 eikonal points): outputs + all 28 gradient tensors
  * compute_level_grid at vox_res=100 for one image (1,030,301 points)           utils/eval_3D.py:21-38
  * 128x128 full-frame evaluation render of one image (1,048,576 points)         model/renderer.py:57-152
  * Chamfer 20,000 x 20,000 through both entry points (direct and target-split)  chamfer3D.cu:12-154
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _opt(H, W, extra=()):
    from shapeclipper_amd.utils import options
    o = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest", "--output_root=/tmp/sc_pytest"]
                                            + list(extra)), verbose=False)
    o.H, o.W = H, W
    return o


def _weights(golden):
    g = golden("g12_render_hits")
    Ws = {k[len("w.sdf."):]: torch.tensor(g[k]) for k in g.files if k.startswith("w.sdf.")}
    Wr = {k[len("w.rgb."):]: torch.tensor(g[k]) for k in g.files if k.startswith("w.rgb.")}
    return Ws, Wr


def _renderer(Ws, Wr, opt, dev, beta):
    from shapeclipper_amd.model.implicit import RGBNetwork, SDFNetwork
    from shapeclipper_amd.model.renderer import Renderer
    sdf_net, rgb_net = SDFNetwork(opt), RGBNetwork(opt)
    sdf_net.load_state_dict(Ws)
    rgb_net.load_state_dict(Wr)
    r = Renderer(opt, sdf_net, rgb_net).to(dev)
    with torch.no_grad():
        r.density.beta.fill_(beta)
    return r


def _cameras(cfg, B, seed):
    from oracle import reference_ops as R
    g = torch.Generator().manual_seed(seed)
    az = (torch.rand(B, generator=g) * 2 - 1) * np.pi
    el = (torch.rand(B, generator=g) - 0.5) * np.pi / 3
    trig = lambda t: torch.stack([torch.cos(t), torch.sin(t)], 1)
    sd = 0.9 + 0.2 * torch.rand(B, generator=g)
    pose = R.pose_from_trig(cfg, trig(az), trig(el), trig(torch.zeros(B)), sd)
    intr = R.get_intr(cfg, torch.ones(B))
    zs, zr = torch.randn(B, 64, generator=g) * 0.3, torch.randn(B, 64, generator=g) * 0.3
    return pose, intr, sd, zs, zr


def test_training_render_b4_r512_vs_oracle_all_gradients(golden):
    from oracle import reference_ops as R
    dev = torch.device("cuda:0")
    B, Rr, beta = 4, 512, 0.05
    opt, cfg = _opt(224, 224), R.Cfg(H=224, W=224)
    Ws, Wr = _weights(golden)
    pose, intr, sd, zs, zr = _cameras(cfg, B, seed=7)
    gen = torch.Generator().manual_seed(8)
    ray_idx = torch.stack([torch.randperm(224 * 224, generator=gen)[:Rr] for _ in range(B)])
    # ---- oracle (CPU), explicit RNG draws in the reference's order
    torch.manual_seed(1234)
    state = torch.get_rng_state()
    t_rand, eik_idx, eik_pts = R.draw_render_randoms(B * Rr, 64, True)
    oWs = {k: v.clone().requires_grad_(True) for k, v in Ws.items()}
    oWr = {k: v.clone().requires_grad_(True) for k, v in Wr.items()}
    ob = torch.tensor(beta).requires_grad_(True)
    ol = dict(pose=pose.clone().requires_grad_(True), intr=intr.clone().requires_grad_(True), scale_dist=sd.clone().requires_grad_(True),
              z_sdf=zs.clone().requires_grad_(True), z_rgb=zr.clone().requires_grad_(True))
    o = R.render(cfg, oWs, oWr, ob, ol["pose"], ol["intr"], ol["scale_dist"], ol["z_sdf"], ol["z_rgb"], ray_idx, True,
                 t_rand, eik_idx, eik_pts)
    hit = o["mask_hard"].detach()
    cot = dict(rgb=torch.randn(B, Rr, 3, generator=gen), mask=torch.randn(B, Rr, 1, generator=gen),
               depth=torch.randn(B, Rr, 1, generator=gen), normal=torch.randn(B, Rr, 3, generator=gen) * hit,
               eik=torch.randn(2 * B * Rr, generator=gen))
    fun = lambda rgb, mask, depth, normal, eik, c: ((rgb * c["rgb"]).sum() + (mask * c["mask"]).sum() + (depth * c["depth"]).sum()
                                                    + (normal * c["normal"]).sum() + (eik * c["eik"]).sum())
    names = (["sdf_network." + k for k in oWs] + ["rgb_network." + k for k in oWr] + ["density.beta"] + list(ol))
    og = torch.autograd.grad(fun(o["rgb"], o["mask"], o["depth"], o["normal"], o["grad_eikonal"], cot),
                             list(oWs.values()) + list(oWr.values()) + [ob] + list(ol.values()), allow_unused=True)
    ref = {n: (g_ if g_ is not None else None) for n, g_ in zip(names, og)}
    # ---- HIP path, same CPU generator state
    r = _renderer(Ws, Wr, opt, dev, beta)
    lv = {k: v.detach().to(dev).requires_grad_(True) for k, v in ol.items()}
    torch.set_rng_state(state)
    rgb, mask, mask_hard, depth, normal, eik = r(opt, lv["pose"], lv["intr"], lv["scale_dist"], lv["z_sdf"], lv["z_rgb"],
                                                 ray_idx=ray_idx.to(dev), training=True)
    err = lambda a, k: float((a.detach().cpu() - o[k].detach()).abs().max())
    hm = hit[..., 0] > 0
    e = dict(rgb=err(rgb, "rgb"), mask=err(mask, "mask"), depth=err(depth, "depth"), eik=err(eik, "grad_eikonal"),
             normal_hit=float((normal.detach().cpu() - o["normal"].detach())[hm].abs().max()))
    print("B=4 R=512 train render, max abs err:", {k: "%.2e" % v for k, v in e.items()}, "hit fraction %.2f" % float(hm.float().mean()))
    assert 0.2 < float(hm.float().mean()) < 0.9
    assert e["rgb"] < 5e-5 and e["mask"] < 5e-5 and e["depth"] < 2e-4 and e["eik"] < 2e-4 and e["normal_hit"] < 2e-4
    guard = (o["mask"].detach() - 0.5).abs() > 1e-5
    assert torch.equal(mask_hard.cpu()[guard], o["mask_hard"].detach()[guard])
    cd = {k: v.to(dev) for k, v in cot.items()}
    params = dict(r.named_parameters())
    got = torch.autograd.grad(fun(rgb, mask, depth, normal, eik, cd), [params[n] if n in params else lv[n] for n in names],
                              allow_unused=True)
    torch.cuda.synchronize()
    worst = {}
    for n, gg in zip(names, got):
        rf = ref[n] if ref[n] is not None else torch.zeros_like(params[n] if n in params else lv[n]).cpu()
        gv = gg.cpu() if gg is not None else torch.zeros_like(rf)
        worst[n] = float((gv - rf).abs().max() / max(float(rf.abs().max()), 1e-4))
    print("B=4 R=512 gradient errors (max abs / max |ref|):", {k: "%.1e" % v for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if v > 2e-4}       # measured: <= 3.2e-5
    assert not bad, bad


@pytest.mark.parametrize("value_split", [True, False])
def test_level_grid_vox100_vs_oracle(golden, value_split):
    """value_split: the default pre-split bf16x3 value chain (csrc/sdf_value_split.hip, round 6) / the fp32-MFMA chain (`--hip.value_split!`)."""
    from oracle import reference_ops as R
    from shapeclipper_amd import ops
    from shapeclipper_amd.model.implicit import SDFNetwork
    from shapeclipper_amd.utils import eval_3D
    from shapeclipper_amd.utils.util import EasyDict as edict
    o = _opt(224, 224, ["--eval.vox_res=100"])
    o.device = "cuda:0"
    Ws, _ = _weights(golden)
    net = SDFNetwork(o)
    net.load_state_dict(Ws)
    net = net.cuda()
    z = torch.randn(1, 64, generator=torch.Generator().manual_seed(3)) * 0.3
    grid = eval_3D.get_dense_3D_grid(o, edict(idx=torch.arange(1)))
    assert grid.shape == (1, 101, 101, 101, 3)
    try:
        ops.SDF_VALUE_SPLIT = value_split
        lvl = eval_3D.compute_level_grid(o, net, z.cuda(), grid).cpu()
    finally:
        ops.SDF_VALUE_SPLIT = True
    ref = R.level_grid(R.Cfg(), Ws, z, R.dense_grid(-0.6, 0.6, 100, 1))
    e = float((lvl - ref).abs().max())
    sign_flips = int(((lvl > 0) != (ref > 0))[ref.abs() > 1e-5].sum())
    print("level grid vox_res=100 (1,030,301 points): max abs err %.2e, |level| max %.2f, sign flips outside 1e-5: %d"
          % (e, float(ref.abs().max()), sign_flips))
    assert lvl.shape == ref.shape == (1, 101, 101, 101)
    assert e < 2e-5 and sign_flips == 0
    assert float((ref < 0).float().mean()) > 0.01          # the grid really contains the shape


def test_full_frame_eval_render_128_one_image_vs_oracle(golden):
    from oracle import reference_ops as R
    dev = torch.device("cuda:0")
    beta = 0.05
    opt, cfg = _opt(128, 128), R.Cfg(H=128, W=128)
    Ws, Wr = _weights(golden)
    pose, intr, sd, zs, zr = _cameras(cfg, 1, seed=11)
    r = _renderer(Ws, Wr, opt, dev, beta)
    with torch.no_grad():
        rgb, mask, mask_hard, depth, normal, eik = r(opt, pose.to(dev), intr.to(dev), sd.to(dev), zs.to(dev), zr.to(dev),
                                                     ray_idx=None, training=False)
    assert rgb.shape == (1, 128 * 128, 3) and eik is None
    # the oracle materialises ~10 KB per sample point: 4 chunks of 4096 rays (rays are independent)
    outs = {k: [] for k in ("rgb", "mask", "mask_hard", "depth", "normal")}
    for c in range(4):
        idx = torch.arange(c * 4096, (c + 1) * 4096).view(1, -1)
        _, eik_idx, _ = R.draw_render_randoms(4096, 64, False)
        with torch.no_grad():
            o = R.render(cfg, Ws, Wr, torch.tensor(beta), pose, intr, sd, zs, zr, idx, False, None, eik_idx, None)
        for k in outs:
            outs[k].append(o[k].detach())
    ref = {k: torch.cat(v, 1) for k, v in outs.items()}
    hm = ref["mask_hard"][..., 0] > 0
    err = lambda a, k: float((a.cpu() - ref[k]).abs().max())
    e = dict(rgb=err(rgb, "rgb"), mask=err(mask, "mask"), depth=err(depth, "depth"),
             normal_hit=float((normal.cpu() - ref["normal"])[hm].abs().max()))
    print("128x128 eval render (1,048,576 points), max abs err:", {k: "%.2e" % v for k, v in e.items()},
          "hit fraction %.2f" % float(hm.float().mean()))
    assert 0.1 < float(hm.float().mean()) < 0.9
    assert e["rgb"] < 5e-5 and e["mask"] < 5e-5 and e["depth"] < 2e-4 and e["normal_hit"] < 2e-4
    guard = (ref["mask"] - 0.5).abs() > 1e-5
    assert torch.equal(mask_hard.cpu()[guard], ref["mask_hard"][guard])


def _run_chamfer(a, b, split):
    import ctypes
    from shapeclipper_amd import _lib
    import chamfer_3D
    dev = torch.device("cuda:0")
    x1, x2 = torch.tensor(a, device=dev), torch.tensor(b, device=dev)
    B, N, M = x1.shape[0], x1.shape[1], x2.shape[1]
    d1 = torch.zeros(B, N, device=dev); d2 = torch.zeros(B, M, device=dev)
    i1 = torch.zeros(B, N, dtype=torch.int32, device=dev); i2 = torch.zeros(B, M, dtype=torch.int32, device=dev)
    lib = _lib.load()
    if split:
        ws = torch.empty(B * (N + M), dtype=torch.int64, device=dev)
        rc = lib.sc_chamfer3d_forward_split(_lib.ptr(x1), _lib.ptr(x2), _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(i1), _lib.ptr(i2),
                                            ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M), ctypes.c_int(16), _lib.ptr(ws), _lib.stream())
    else:
        rc = lib.sc_chamfer3d_forward(_lib.ptr(x1), _lib.ptr(x2), _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(i1), _lib.ptr(i2),
                                      ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M), _lib.stream())
    assert rc == 0
    torch.cuda.synchronize()
    # module entry point (chooses a variant itself) must agree with the explicit one
    d1m, d2m, i1m, i2m = torch.zeros_like(d1), torch.zeros_like(d2), torch.zeros_like(i1), torch.zeros_like(i2)
    assert chamfer_3D.forward(x1, x2, d1m, d2m, i1m, i2m) == 1
    assert torch.equal(d1m, d1) and torch.equal(i1m, i1) and torch.equal(d2m, d2) and torch.equal(i2m, i2)
    return x1, x2, d1, d2, i1, i2


@pytest.mark.parametrize("split", [False, True])
def test_chamfer_20000_both_entry_points(split):
    from oracle import chamfer_ref
    rng = np.random.RandomState(20)
    a = rng.uniform(-0.5, 0.5, (1, 20000, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (1, 20000, 3)).astype(np.float32)
    b[0, 15000] = b[0, 33]; a[0, 7] = b[0, 33]                  # exact duplicate target + zero distance
    x1, x2, d1, d2, i1, i2 = _run_chamfer(a, b, split)
    r1, r2, j1, j2 = chamfer_ref.chamfer_forward(a, b)          # same fma chain: bit-exact
    assert np.array_equal(d1.cpu().numpy(), r1) and np.array_equal(d2.cpu().numpy(), r2)
    assert np.array_equal(i1.cpu().numpy(), j1) and np.array_equal(i2.cpu().numpy(), j2)
    assert int(i1[0, 7]) == 33 and float(d1[0, 7]) == 0.0
    # against a float64 brute force: a few ulp on dist (three roundings + the rounded differences), and the index is a true
    # minimiser up to that gap
    worst_ulp, worst_gap = 0.0, 0.0
    for (q, t, d, i) in ((x1, x2, d1, i1), (x2, x1, d2, i2)):
        q64, t64 = q[0].double(), t[0].double()
        for s in range(0, q64.shape[0], 2000):
            D = (q64[s:s + 2000, None, :] - t64[None, :, :]).square().sum(-1)
            dmin = D.min(1).values
            dd = d[0, s:s + 2000].double()
            ulp = torch.maximum(dd.float(), torch.tensor(1e-30, device=dd.device)).double() * 2.0 ** -23
            worst_ulp = max(worst_ulp, float(((dd - dmin).abs() / ulp).max()))
            chosen = D.gather(1, i[0, s:s + 2000].long()[:, None])[:, 0]
            worst_gap = max(worst_gap, float(((chosen - dmin) / ulp).max()))
    print("Chamfer 20000x20000 (split=%s): dist vs float64 brute force %.2f ulp, chosen index within %.2f ulp of the minimum"
          % (split, worst_ulp, worst_gap))
    assert worst_ulp <= 4.0 and worst_gap <= 8.0     # dx = t - q is itself rounded: a few ulp against exact arithmetic


def test_config2_chamfer_b32_100k_equals_single_item_runs():
    """BASELINE config[2]: Chamfer at B=32, N=M=100,000 (the batched kernel, 32 x 98 workgroups per direction).  Every
    item must equal the B=1 call on the same clouds bit for bit (that call takes the target-split / atomicMin variant, which
    is itself compared with the oracle at 20,000^2 above), and swapping the clouds swaps the outputs."""
    import chamfer_3D
    dev = torch.device("cuda:0")
    B, N = 32, 100000
    g = torch.Generator().manual_seed(9)
    a = (torch.rand(B, N, 3, generator=g) - 0.5).to(dev)
    b = (torch.rand(B, N, 3, generator=g) - 0.5).to(dev)

    def run(x, y):
        n = x.shape[0]
        d1 = torch.zeros(n, N, device=dev); d2 = torch.zeros(n, N, device=dev)
        i1 = torch.zeros(n, N, dtype=torch.int32, device=dev); i2 = torch.zeros(n, N, dtype=torch.int32, device=dev)
        assert chamfer_3D.forward(x, y, d1, d2, i1, i2) == 1
        return d1, d2, i1, i2
    full = run(a, b)
    for item in (0, 17, 31):
        one = run(a[item:item + 1].contiguous(), b[item:item + 1].contiguous())
        for x, y in zip(full, one):
            assert torch.equal(x[item], y[0]), item
    swapped = run(b, a)
    assert torch.equal(swapped[0], full[1]) and torch.equal(swapped[3], full[2])
    assert int(full[2].min()) >= 0 and int(full[2].max()) < N
    print("Chamfer B=32 N=M=100000: items 0/17/31 bit-identical to their B=1 runs; mean sqrt distance %.5f" % float(full[0].sqrt().mean()))


def test_config2_render_128_b32_rows_equal_single_image_renders(golden):
    """BASELINE config[2]: full-frame 128x128 evaluation render at B=32 (524,288 rays, 33.5 M sample points in one call).
    Images are independent: rows of the batched render equal the single-image renders (compared with the oracle above)."""
    from oracle import reference_ops as R
    dev = torch.device("cuda:0")
    B, beta = 32, 0.05
    opt, cfg = _opt(128, 128), R.Cfg(H=128, W=128)
    Ws, Wr = _weights(golden)
    pose, intr, sd, zs, zr = _cameras(cfg, B, seed=21)
    r = _renderer(Ws, Wr, opt, dev, beta)
    to = lambda t: t.to(dev)
    with torch.no_grad():
        full = r(opt, to(pose), to(intr), to(sd), to(zs), to(zr), ray_idx=None, training=False)
        assert full[0].shape == (B, 128 * 128, 3) and full[5] is None
        worst = 0.0
        for item in (0, 13, 31):
            s = slice(item, item + 1)
            one = r(opt, to(pose[s]), to(intr[s]), to(sd[s]), to(zs[s]), to(zr[s]), ray_idx=None, training=False)
            for k in (0, 1, 3):
                worst = max(worst, float((full[k][item] - one[k][0]).abs().max()))
            assert torch.equal(full[2][item], one[2][0]) or float((full[1][item] - 0.5).abs().min()) < 1e-5
    hit = float(full[2].mean())
    print("128x128 render B=32: rows vs single-image renders max abs diff %.2e, hit fraction %.2f" % (worst, hit))
    # Round 5: the per-image latent biases come from csrc/latent_bias.hip -- one fixed-order sum per element, whatever the batch
    # size -- so a row of the batched render IS the single-image render, bit for bit (with the rocBLAS product of round 4: 2.2e-6).
    assert worst == 0.0 and 0.1 < hit < 0.9


def test_chamfer_backward_20000_vs_oracle():
    """NmDistanceGradKernel (chamfer3D.cu:155-174) at 20,000 x 20,000: scatter-add gradients vs the C restatement (atomic
    accumulation order differs: 1e-5 relative) -- many queries share a nearest target here, so the atomics really collide."""
    import chamfer_3D
    from oracle import chamfer_ref
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(3)
    a = rng.uniform(-0.5, 0.5, (2, 20000, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (2, 3000, 3)).astype(np.float32)          # 7 queries per target on average
    d1, d2, i1, i2 = chamfer_ref.chamfer_forward(a, b)
    gd1, gd2 = rng.randn(2, 20000).astype(np.float32), rng.randn(2, 3000).astype(np.float32)
    r1, r2 = chamfer_ref.chamfer_backward(a, b, gd1, gd2, i1, i2)
    t = lambda x: torch.tensor(x, device=dev)
    g1, g2 = torch.zeros(2, 20000, 3, device=dev), torch.zeros(2, 3000, 3, device=dev)
    assert chamfer_3D.backward(t(a), t(b), g1, g2, t(gd1), t(gd2), t(i1), t(i2)) == 1
    torch.cuda.synchronize()
    e1 = float(np.abs(g1.cpu().numpy() - r1).max() / np.abs(r1).max()); e2 = float(np.abs(g2.cpu().numpy() - r2).max() / np.abs(r2).max())
    print("Chamfer backward 20000 x 3000 (B=2): max err / max |ref| = %.1e (xyz1), %.1e (xyz2)" % (e1, e2))
    assert e1 < 1e-5 and e2 < 1e-5
