"""The SURVEY 8(d) workloads besides the train step, each timed with events on the launch stream and with a bounded
CPU leg beside it (bench.py prints them in its `workloads` object; `python bench.py --workloads-only` runs only these,
which is the command the rocprofv3 summaries under profiles/ are taken from).

  chamfer_b1 / chamfer_b32   Chamfer3D forward, N = M = 100,000, both directions   utils/eval_3D.py:155-165, chamfer3D.cu:142-143
  clip_vit_b32               CLIP ViT-B/32 image tower, batch 32                   CLIP_anno.py:166-167
  render_eval_128            full-frame evaluation render 128x128, batch 32        model/renderer.py:57-152
  level_grid_100             SDF level grid, vox_res = 100, one image              utils/eval_3D.py:21-38
  resnet_conv3x3             the 3x3 / stride-1 convolutions of one bs32 step's trunks  model/graph.py:50-54, view_estimator.py:40-42
                             (ResNet-34 encoder at 64 images + ResNet-18 estimator at 96: forward, backward-data, backward-weight of 42 layers)

Algorithmic work per unit is SURVEY 8(d)'s: 8 FLOP per ordered pair (Chamfer), 8.725 GFLOP per image (ViT-B/32),
12,763,136 FLOP per ray (evaluation render), 80,640 FLOP per grid point.  `frac` = achieved / peak of the unit that
bounds the kernel (fp32 VALU / bf16 MFMA / fp32 MFMA, MI355X_MICROARCH.md).  CPU legs ("cpu"): the oracle
(oracle/reference_ops.py), a chunked torch.cdist brute force, and transformers' CLIP vision model, on a bounded sample
of the same workload with min(os.cpu_count(), 32) threads -- reported baselines, not targets.  Chamfer also carries
`reference_gpu`: the reference's own CUDA extension built for gfx950 (oracle/_ref), timed on the same tensors.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32 = 157.3       # TFLOP/s, fp32 vector = fp32 MFMA dense peak
PEAK_BF16 = 2500.0      # TFLOP/s, bf16 MFMA dense peak
PEAK_HBM_GBPS = 8000.0  # GB/s, HBM3E (MI355X_MICROARCH.md)
PEAK_SPLIT = PEAK_BF16 / 6   # fp32-equivalent roof of the exact bf16x3 split: six bf16 MFMA products per fp32 product (416.7 TFLOP/s)
CHAMFER_FLOP_PER_PAIR = 8
VIT_B32_GFLOP = 8.725
VIT_L14_GFLOP = 155.5   # the model the reference loads (CLIP_anno.py:16)
RENDER_EVAL_FLOP_PER_RAY = 64 * 199424
GRID_FLOP_PER_POINT = 80640


# share of the reference-dense count the kernels execute (latent columns folded into per-image biases: SURVEY section 7 hard part 2)
EXECUTED_SHARE_RENDER = (2 * 28032 + 14976) / (2 * 40320 + 19072)
EXECUTED_SHARE_GRID = 28032 / 40320


def _gpu_ms(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    d = sorted(s.elapsed_time(e) for s, e in evs)
    return sum(d) / len(d), d[0]


def _cpu_time(fn, budget_s=6.0, max_n=5):
    fn()                                     # warm-up (thread pool, allocator)
    t0, n = time.time(), 0
    while n < 1 or (time.time() - t0 < budget_s and n < max_n):
        fn(); n += 1
    return (time.time() - t0) / n, n


def cpu_threads():
    return min(os.cpu_count() or 1, 32)


def chamfer(B, N=100000, with_cpu=True, with_reference_gpu=False):
    import chamfer_3D
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(B)
    a = (torch.rand(B, N, 3, generator=g) - 0.5).to(dev); b = (torch.rand(B, N, 3, generator=g) - 0.5).to(dev)
    d1 = torch.zeros(B, N, device=dev); d2 = torch.zeros(B, N, device=dev)
    i1 = torch.zeros(B, N, dtype=torch.int32, device=dev); i2 = torch.zeros(B, N, dtype=torch.int32, device=dev)
    def run(search, x=None, y=None, outs=None):
        x, y = (a, b) if x is None else (x, y)
        o = (d1, d2, i1, i2) if outs is None else outs
        old, chamfer_3D.SEARCH = chamfer_3D.SEARCH, search
        try:
            return _gpu_ms(lambda: chamfer_3D.forward(x, y, *o), iters=5 if B > 1 else 20)
        finally:
            chamfer_3D.SEARCH = old
    # all pairs (csrc/chamfer.hip): the reference's algorithm at the packed-fp32 issue floor -- the roofline line
    bms, bbest = run("brute")
    keep = [t.clone() for t in (d1, d2, i1, i2)]
    # what chamfer_3D.forward runs by default: the exact uniform-grid search (csrc/chamfer_grid.hip), same bits
    ms, best = run("grid")
    same = all(torch.equal(x, y) for x, y in zip(keep, (d1, d2, i1, i2)))
    pairs = 2.0 * B * N * N
    tf = CHAMFER_FLOP_PER_PAIR * pairs / (bms * 1e-3) / 1e12
    out = dict(workload="Chamfer3D forward B=%d, N=M=%d, both directions" % (B, N), ms=round(ms, 3), ms_best=round(best, 3),
               search="exact uniform-grid ring search (default of chamfer_3D.forward from 2048 points up); results bit-identical to all pairs",
               same_results_as_all_pairs=bool(same), speedup_vs_all_pairs=round(bms / ms, 2),
               bound="latency / L2 gathers (tens of candidates per query instead of M)", mqueries_per_s=round(2.0 * B * N / (ms * 1e-3) / 1e6, 1),
               equivalent_tpairs_per_s=round(pairs / (ms * 1e-3) / 1e12, 2), algorithmic_bytes=B * 2 * N * (12 + 8),
               all_pairs=dict(ms=round(bms, 3), ms_best=round(bbest, 3), algorithmic_flop=CHAMFER_FLOP_PER_PAIR * pairs,
                              achieved=round(tf, 2), peak=PEAK_FP32, unit="TFLOP/s", bound="fp32 VALU (4 packed vector instructions per pair: the issue floor of the reference's expression)",
                              frac=round(tf / PEAK_FP32, 4), tpairs_per_s=round(pairs / (bms * 1e-3) / 1e12, 3),
                              ceiling_frac=round(8.0 / 15.0, 4), frac_of_ceiling=round(tf / PEAK_FP32 / (8.0 / 15.0), 4),
                              ceiling="the reference's expression costs 7 packed fp32 instructions + 1 v_min3 per TWO pairs = 7.5 issue slots per pair "
                                      "for 8 algorithmic FLOP, against 2 FLOP per slot at the FMA peak: 8 / 15 of 157.3 TFLOP/s is the most any all-pairs "
                                      "kernel with these bits can reach"))
    # the evaluation's clouds are SURFACES (100,000 samples of the predicted iso-surface against 100,000 ground-truth surface points,
    # utils/eval_3D.py:205), not volumes, and they do not coincide: the same search on two bumpy spheres a mean distance 0.047 apart
    # (tools/perf_chamfer_surface.py sweeps the distance: the grid search wins up to ~0.1, costs up to 1.6x all pairs beyond at B=1)
    gd = torch.Generator(device="cuda").manual_seed(B)
    def _sphere(r, bumps):
        v = torch.randn(B, N, 3, device=dev, generator=gd)
        v = v / v.norm(dim=-1, keepdim=True)
        return (v * r * (1 + bumps * torch.sin(7 * v[..., :1]) * torch.cos(5 * v[..., 1:2]))).contiguous()
    # the surface leg has its OWN clouds and result buffers: d1, d2, i1, i2 keep the uniform clouds' results, which the
    # reference_gpu leg below is compared with (round 3 compared it with the surface results: a false `same_results: false`)
    sa, sb = _sphere(0.4, 0.0), _sphere(0.45, 0.1)
    so = (torch.zeros_like(d1), torch.zeros_like(d2), torch.zeros_like(i1), torch.zeros_like(i2))
    sb_ms, _ = run("brute", sa, sb, so)
    skeep = [t.clone() for t in so]
    sg_ms, _ = run("grid", sa, sb, so)
    out["surface_clouds"] = dict(workload="two bumpy spheres (radius 0.4 / 0.45), mean nearest-neighbour distance %.3f" % float(so[0].sqrt().mean()),
                                 ms=round(sg_ms, 3), all_pairs_ms=round(sb_ms, 3), speedup_vs_all_pairs=round(sb_ms / sg_ms, 2),
                                 same_results_as_all_pairs=bool(all(torch.equal(x, y) for x, y in zip(skeep, so))))
    del sa, sb, so, skeep
    if with_cpu:
        n = N                                   # B = 1, N = M = 100,000
        x, y = a[0, :n].cpu(), b[0, :n].cpu()
        torch.set_num_threads(cpu_threads())

        def brute():
            for q, t in ((x, y), (y, x)):
                for s in range(0, n, 4000):
                    D = torch.cdist(q[s:s + 4000], t)
                    D.min(dim=1)
        t0 = time.time()
        brute()                                 # one run, no warm-up: 2e10 pair evaluations are tens of seconds of host time
        dt = time.time() - t0
        out["cpu"] = dict(value=round(2.0 * n * n / dt / 1e9, 3), unit="Gpairs/s", cores=cpu_threads(), kind="port", seconds=round(dt, 1),
                          sample="chunked torch.cdist + min/argmin, B=1, N=M=%d, both directions (BASELINE.md section 3), 1 timed run without warm-up" % n,
                          gpu_value=round(pairs / (ms * 1e-3) / 1e9, 1))
    # baseline leg, OPT-IN (`bench.py --with-reference-gpu`): the reference's OWN extension built for this GPU (oracle/_ref, checker of
    # tests/test_gpu_chamfer_ref.py) on the same tensors and device -- what a user of the reference would get here without this
    # build.  It maps a binary compiled from the reference's (untrusted) sources into the bench process, so it never runs by default.
    ref = None
    if with_reference_gpu:
        try:
            from oracle import build_chamfer_ref
            ref = build_chamfer_ref.load_module()
        except Exception:        # noqa: BLE001
            ref = None
    if ref is not None:
        assert all(torch.equal(x, y) for x, y in zip(keep, (d1, d2, i1, i2))), "d1/d2/i1/i2 must still hold the uniform clouds' results"
        e1, e2, j1, j2 = torch.zeros_like(d1), torch.zeros_like(d2), torch.zeros_like(i1), torch.zeros_like(i2)
        rms, _ = _gpu_ms(lambda: ref.forward(a, b, e1, e2, j1, j2), iters=3, warm=1)
        out["reference_gpu"] = dict(ms=round(rms, 3), kind="reference", speedup=round(rms / ms, 2), same_results=bool(
            torch.equal(e1, d1) and torch.equal(e2, d2) and torch.equal(j1, i1) and torch.equal(j2, i2)),
            sample="external/chamfer3D of the reference built for gfx950 (oracle/build_chamfer_ref.py), same uniform clouds on the same GPU; "
                   "same_results compares dist1, dist2, idx1, idx2 bit for bit with chamfer_3D.forward's")
    return out


def clip_vit(B=32, with_cpu=True, model="ViT-B/32"):
    """The tower in its default arithmetic: fp16 operands (what openai/CLIP runs on a GPU), fp32 accumulate; same 2.5 PFLOP/s dense
    MFMA peak as bf16."""
    from shapeclipper_amd.model.clip_vit import VIT_B32, VIT_L14, ClipVisionTower
    torch.manual_seed(0)
    cfg, gflop = (VIT_L14, VIT_L14_GFLOP) if model == "ViT-L/14" else (VIT_B32, VIT_B32_GFLOP)
    tower = ClipVisionTower(**cfg).cuda()
    x = torch.randn(B, 3, 224, 224, device="cuda")
    ms, best = _gpu_ms(lambda: tower.encode_image(x), iters=10 if model == "ViT-L/14" else 20, warm=3)
    tf = B * gflop * 1e9 / (ms * 1e-3) / 1e12
    out = dict(workload="CLIP %s image tower forward, B=%d, 224x224" % (model, B), ms=round(ms, 3), ms_best=round(best, 3), dtype=tower.dtype16,
               algorithmic_flop=B * gflop * 1e9, achieved=round(tf, 1), peak=PEAK_BF16, unit="TFLOP/s", bound="fp16 MFMA (dense peak = bf16's)",
               frac=round(tf / PEAK_BF16, 4), images_per_s=round(B / (ms * 1e-3), 1),
               parity="unpinned against the reference (openai/CLIP is un-vendored, no weights offline: SURVEY 8c); checked against transformers' "
                      "fp32 architecture with seeded random weights (tests/test_gpu_clip.py)")
    if with_cpu and model != "ViT-B/32":
        with_cpu = False
    if with_cpu:
        try:
            from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
            torch.set_num_threads(cpu_threads())
            cfg = CLIPVisionConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                                   patch_size=32, image_size=224, projection_dim=512, hidden_act="quick_gelu")
            m = CLIPVisionModelWithProjection(cfg).eval()
            xc = x[:32].cpu()
            with torch.no_grad():
                dt, k = _cpu_time(lambda: m(pixel_values=xc))
            out["cpu"] = dict(value=round(xc.shape[0] / dt, 2), unit="images/s", cores=cpu_threads(), kind="port",
                              sample="transformers CLIPVisionModelWithProjection (ViT-B/32 geometry, fp32), batch %d (BASELINE.md section 3), %d timed runs" % (xc.shape[0], k),
                              gpu_value=out["images_per_s"])
        except Exception as e:      # transformers missing on the box: say so instead of failing the bench
            out["cpu"] = dict(value=None, error="%s: %s" % (type(e).__name__, e))
    return out


def _renderer(opt):
    from shapeclipper_amd.model.implicit import RGBNetwork, SDFNetwork
    from shapeclipper_amd.model.renderer import Renderer
    torch.manual_seed(0)
    sdf, rgb = SDFNetwork(opt), RGBNetwork(opt)
    return Renderer(opt, sdf, rgb).cuda(), sdf, rgb


def _options(extra=()):
    from shapeclipper_amd.utils import options
    return options.set(options.parse_arguments(["--yaml=%s/options/pix3d/config.yaml" % ROOT, "--name=bench_workloads",
                                                "--output_root=/tmp/sc_bench_w"] + list(extra)), verbose=False)


def _cameras(opt, B):
    """Random turn-table cameras through the PRODUCT's camera algebra (the oracle only ever appears in the CPU legs)."""
    from shapeclipper_amd.model.graph import rotation_from_trig
    from shapeclipper_amd.utils import camera
    g = torch.Generator().manual_seed(1)
    az = (torch.rand(B, generator=g) * 2 - 1) * np.pi
    el = (torch.rand(B, generator=g) - 0.5) * np.pi / 3
    trig = lambda t: torch.stack([torch.cos(t), torch.sin(t)], 1)
    sd = 0.8 + 0.4 * torch.rand(B, generator=g)
    Rm = rotation_from_trig(trig(az), trig(el), trig(torch.zeros(B)))
    tz = sd * opt.camera.dist
    pose = camera.pose.compose([camera.pose(R=Rm), camera.pose(t=torch.stack([torch.zeros(B), torch.zeros(B), tz], -1))])
    return pose, camera.get_intr(opt, torch.ones(B)), sd, torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)


def render_eval_128(B=32, with_cpu=True):
    opt = _options()
    opt.H = opt.W = 128
    r, sdf_net, rgb_net = _renderer(opt)
    pose, intr, sd, zs, zr = _cameras(opt, B)
    dev = torch.device("cuda")
    args = [t.to(dev) for t in (pose, intr, sd, zs, zr)]

    def run():
        with torch.no_grad():
            return r(opt, *args, ray_idx=None, training=False)
    ms, best = _gpu_ms(run, iters=5)
    rays = B * 128 * 128
    tf = RENDER_EVAL_FLOP_PER_RAY * rays / (ms * 1e-3) / 1e12
    from shapeclipper_amd import ops as _ops
    split = bool(_ops.SDF_FWD_STREAM and _ops.RGB_FWD_SPLIT)
    # round 6: both networks run in the exact three-piece bf16 split arithmetic (csrc/sdf_fwd_stream.hip, rgb_fwd.hip): the roof is the bf16
    # matrix pipe / 6 piece products per fp32 product (the trunk convolutions' roof); `frac_of_fp32_mfma_peak` keeps the old yardstick
    peak = PEAK_SPLIT if split else PEAK_FP32
    ms32 = None
    if split:       # the fp32-MFMA kernels on the same inputs, same box (`--hip.sdf_stream! --hip.rgb_split!`)
        try:
            _ops.SDF_FWD_STREAM = _ops.RGB_FWD_SPLIT = False
            ms32, _ = _gpu_ms(run, iters=3)
        finally:
            _ops.SDF_FWD_STREAM = _ops.RGB_FWD_SPLIT = True
    out = dict(workload="full-frame evaluation render 128x128, B=%d (%d rays x 64 samples)" % (B, rays), ms=round(ms, 3),
               ms_best=round(best, 3), algorithmic_flop=RENDER_EVAL_FLOP_PER_RAY * rays, algorithmic_bytes=rays * 44,
               achieved=round(tf, 2), peak=peak, unit="TFLOP/s", bound="bf16 MFMA / 6 (exact 3-piece split, fp32 accumulate)" if split else "fp32 MFMA",
               frac=round(tf / peak, 4), frac_of_fp32_mfma_peak=round(tf / PEAK_FP32, 4), fp32_mfma_kernels_ms=round(ms32, 3) if ms32 else None,
               # `frac` counts the reference-dense FLOPs of SURVEY 8(d); the kernels fold the 64 latent columns of the
               # conditioned layers into per-image biases (28,032 of 40,320 SDF MACs -- value and d/dx sweep alike -- and 14,976 of 19,072 RGB
               # MACs per point are executed), so the matrix pipe does EXECUTED_SHARE of the counted work: the real utilisation is the second figure
               executed_flop=int(RENDER_EVAL_FLOP_PER_RAY * rays * EXECUTED_SHARE_RENDER), frac_executed=round(tf * EXECUTED_SHARE_RENDER / peak, 4),
               frac_note="frac = reference-dense FLOPs (SURVEY 8d) / time / peak; frac_executed = the MACs the kernels execute after folding the "
                         "latent columns into per-image biases (%.3f of the dense count) / time / peak = the matrix pipe's real utilisation" % EXECUTED_SHARE_RENDER,
               mrays_per_s=round(rays / (ms * 1e-3) / 1e6, 2))
    if with_cpu:
        from oracle import reference_ops as R
        cfg = R.Cfg(H=128, W=128)
        torch.set_num_threads(cpu_threads())
        Ws = {k: v.detach().cpu() for k, v in sdf_net.state_dict().items()}
        Wr = {k: v.detach().cpu() for k, v in rgb_net.state_dict().items()}
        beta = r.density.beta.detach().cpu().reshape(())
        idx = torch.arange(4096).view(1, -1)
        _, eik_idx, _ = R.draw_render_randoms(4096, 64, False)

        def oracle():
            with torch.no_grad():
                R.render(cfg, Ws, Wr, beta, pose[:1], intr[:1], sd[:1], zs[:1], zr[:1], idx, False, None, eik_idx, None)
        dt, k = _cpu_time(oracle)
        out["cpu"] = dict(value=round(4096 / dt / 1e6, 4), unit="Mrays/s", cores=cpu_threads(), kind="port",
                          sample="oracle evaluation render of 4096 rays (a quarter frame of one image), %d timed runs" % k,
                          gpu_value=out["mrays_per_s"])
    return out


def level_grid_100(with_cpu=True):
    from shapeclipper_amd.utils import eval_3D
    from shapeclipper_amd.utils.util import EasyDict as edict
    opt = _options(["--eval.vox_res=100"])
    opt.device = "cuda:0"
    r, sdf_net, _ = _renderer(opt)
    z = torch.randn(1, 64, generator=torch.Generator().manual_seed(2)).cuda()
    grid = eval_3D.get_dense_3D_grid(opt, edict(idx=torch.arange(1)))
    ms, best = _gpu_ms(lambda: eval_3D.compute_level_grid(opt, sdf_net, z, grid), iters=20)
    n = 101 ** 3
    tf = GRID_FLOP_PER_POINT * n / (ms * 1e-3) / 1e12
    from shapeclipper_amd import ops as _ops
    split = bool(_ops.SDF_VALUE_SPLIT)
    # round 6: the value chain runs in the exact three-piece bf16 split arithmetic (csrc/sdf_value_split.hip): its roof is the bf16 matrix
    # pipe / 6 piece products per fp32 product (the trunk convolutions' roof), no longer the fp32 pipe
    peak = PEAK_SPLIT if split else PEAK_FP32
    out = dict(workload="SDF level grid, vox_res=100, one image (%d points)" % n, ms=round(ms, 3), ms_best=round(best, 3),
               algorithmic_flop=GRID_FLOP_PER_POINT * n, algorithmic_bytes=4 * n, achieved=round(tf, 2), peak=peak, unit="TFLOP/s",
               bound="bf16 MFMA / 6 (exact 3-piece split, fp32 accumulate)" if split else "fp32 MFMA", frac=round(tf / peak, 4),
               frac_of_fp32_mfma_peak=round(tf / PEAK_FP32, 4),
               executed_flop=int(GRID_FLOP_PER_POINT * n * EXECUTED_SHARE_GRID), frac_executed=round(tf * EXECUTED_SHARE_GRID / peak, 4),
               frac_note="frac = reference-dense 80,640 FLOP per point / time / peak; frac_executed = the 28,032 of 40,320 MACs per point left after "
                         "folding the latent columns into per-image biases (%.3f) = the matrix pipe's real utilisation; round 5 ran this on the "
                         "fp32 pipe (0.52 ms, 0.97 of 157.3 dense / 0.68 executed)" % EXECUTED_SHARE_GRID,
               mpoints_per_s=round(n / (ms * 1e-3) / 1e6, 1))
    if split:        # the fp32-MFMA chain on the same grid, same box (`--hip.value_split!`)
        try:
            _ops.SDF_VALUE_SPLIT = False
            ms32, _ = _gpu_ms(lambda: eval_3D.compute_level_grid(opt, sdf_net, z, grid), iters=20)
        finally:
            _ops.SDF_VALUE_SPLIT = True
        out["fp32_mfma_chain_ms"] = round(ms32, 3)
    if with_cpu:
        from oracle import reference_ops as R
        torch.set_num_threads(cpu_threads())
        Ws = {k: v.detach().cpu() for k, v in sdf_net.state_dict().items()}
        k_pts = 10 * 101 * 101
        flat = grid[0, :10].reshape(-1, 3).cpu()
        lat = z.cpu().repeat(k_pts, 1)

        def oracle():
            with torch.no_grad():
                R.sdf_mlp(R.Cfg(), Ws, flat, lat)
        dt, k = _cpu_time(oracle)
        out["cpu"] = dict(value=round(k_pts / dt / 1e6, 3), unit="Mpoints/s", cores=cpu_threads(), kind="port",
                          sample="oracle SDF MLP on 10 of the 101 x-slabs (%d points), %d timed runs" % (k_pts, k),
                          gpu_value=out["mpoints_per_s"])
    return out


def marching_cubes_100(B=32):
    """Evaluation meshing on the device (utils/eval_3D.py:123-153 in the reference: PyMCubes + trimesh on CPU threads): marching cubes of
    B level grids at vox_res = 100 (count per block, scan over blocks, emit) and 100,000 area-uniform surface samples per image."""
    from shapeclipper_amd import ops
    from shapeclipper_amd.utils import eval_3D
    S = 101
    ax = torch.linspace(-0.6, 0.6, S, device="cuda")
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    gen = torch.Generator(device="cuda").manual_seed(0)
    r = 0.25 + 0.2 * torch.rand(B, 1, 1, 1, device="cuda", generator=gen)
    level = (torch.sqrt(X * X + Y * Y + 1.3 * Z * Z)[None] - r + 0.04 * torch.sin(9 * X)[None] * torch.cos(7 * Y)[None]).contiguous()
    tris, per = ops.isosurface_triangles(level, 0.0)
    ms, best = _gpu_ms(lambda: ops.isosurface_triangles(level, 0.0), iters=10)
    ms_s, _ = _gpu_ms(lambda: eval_3D.surface_points_device(level, -0.6, 0.6, 100000, seed=1), iters=3, warm=1)
    nbytes = 2 * 4.0 * B * S ** 3 + 36.0 * tris.shape[0]                          # grid read by both passes, triangles written
    return dict(workload="marching cubes of %d level grids at vox_res=100 (%d triangles) + 100,000 surface samples per image" % (B, tris.shape[0]),
                ms=round(ms, 3), ms_best=round(best, 3), algorithmic_bytes=nbytes, achieved=round(nbytes / (ms * 1e-3) / 1e9, 1), peak=PEAK_HBM_GBPS,
                unit="GB/s", bound="hbm / latency (count + case index per cube, one scan launch over the 1,024-cube blocks, one host read of the per-image triangle counts, "
                      "emit dealt out by vertex; round 5: 0.52 ms)",
                frac=round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4), mcubes_per_s=round(B * (S - 1) ** 3 / (ms * 1e-3) / 1e6, 1),
                with_sampling_ms=round(ms_s, 3), triangles_per_image=int(tris.shape[0] // B),
                parity="triangulation unpinned against PyMCubes / trimesh (absent: SURVEY 8c); vertex set and triangles checked against "
                       "oracle/isosurface_ref.py (tests/test_gpu_isosurface.py)")


def resnet_conv3x3(with_cpu=True):
    """csrc/conv3x3.hip + conv3x3_wgrad.hip on every 3x3 / stride-1 layer shape of the two trunks, weighted by how often a step runs
    it, with MIOpen (torch's operators) on the same tensors beside it.  Default arithmetic: forward / backward-data from exact
    three-piece bf16 operand splits (roof: bf16 dense peak / 6), weight gradient on fp32 MFMA (roof 157.3); the fp32-MFMA
    forward / backward-data kernels (`--hip.conv3x3_split!`) are timed beside them."""
    from shapeclipper_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(0)
    shapes = []       # (channels, side, batch, layers)
    for layers, batch in (([2, 2, 2, 2], 96), ([3, 4, 6, 3], 64)):       # view estimator: 3 x 32 images, encoder: 2 x 32
        for li, (c, side) in enumerate(((64, 56), (128, 28), (256, 14), (512, 7))):
            shapes.append((c, side, batch, 2 * layers[li] - (1 if li else 0)))
    ms_split = ms_f32 = ms_wg = ms_wgs = ms_lib = flop1 = 0.0
    rows = []
    for c, side, batch, count in shapes:
        x = torch.randn(batch, c, side, side, device=dev)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.05
        gy = torch.randn(batch, c, side, side, device=dev)
        wp_f, wp_b = ops.conv3x3_pack(w, side), ops.conv3x3_pack(w, side, True)
        ws_f, ws_b = ops.conv3x3_pack(w, side, False, True), ops.conv3x3_pack(w, side, True, True)
        t = [_gpu_ms(f, iters=10)[0] for f in (lambda: ops.conv3x3_apply(x, wp_f, c), lambda: ops.conv3x3_apply(gy, wp_b, c),
                                               lambda: ops.conv3x3_backward_weight(gy, x))]
        ts = [_gpu_ms(f, iters=10)[0] for f in (lambda: ops.conv3x3_apply(x, ws_f, c, True), lambda: ops.conv3x3_apply(gy, ws_b, c, True),
                                                lambda: ops.conv3x3_backward_weight(gy, x, split=True))]
        bw = lambda m: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, m)
        tl = [_gpu_ms(f, iters=10)[0] for f in (lambda: torch.nn.functional.conv2d(x, w, None, 1, 1), lambda: bw([True, False, False]),
                                                lambda: bw([False, True, False]))]
        f1 = 2.0 * batch * side * side * c * c * 9
        ms_split += count * sum(ts)
        ms_f32 += count * (t[0] + t[1])
        ms_wg += count * t[2]
        ms_wgs += count * ts[2]
        ms_lib += count * sum(tl)
        flop1 += count * f1
        rows.append(dict(channels=c, side=side, batch=batch, layers=count, split_ms=[round(v, 3) for v in ts], fp32_mfma_ms=[round(v, 3) for v in t],
                         miopen_ms=[round(v, 3) for v in tl], split_tflops=[round(f1 / v / 1e9, 1) for v in ts],
                         fp32_mfma_tflops=[round(f1 / v / 1e9, 1) for v in t]))
        del x, w, gy
    ms_hip = ms_split                      # the product path: all three products in split arithmetic
    tf_s, tf_ws = 3 * flop1 / (ms_split * 1e-3) / 1e12, flop1 / (ms_wgs * 1e-3) / 1e12
    tf_w, tf_f = flop1 / (ms_wg * 1e-3) / 1e12, 2 * flop1 / (ms_f32 * 1e-3) / 1e12
    out = dict(workload="3x3 stride-1 convolutions of one bs32 step (ResNet-34 encoder x 64 images, ResNet-18 estimator x 96): fwd + bwd-data + bwd-weight of 42 layers",
               ms=round(ms_hip, 3), ms_miopen=round(ms_lib, 3),
               ms_miopen_note="torch / MIOpen in IMMEDIATE mode (MIOPEN_FIND_MODE=FAST: no tuning search) and WITHOUT a workspace (its log says "
                              "'workspace required ... provided ptr: 0', so it falls back to fp32 Winograd F(2,3) / direct solvers): an untuned, "
                              "workspace-less MIOpen, not MIOpen at its best -- context, not a claim",
               algorithmic_flop=3 * flop1, achieved=round(tf_s, 2), peak=round(PEAK_SPLIT, 1),
               unit="TFLOP/s", dtype="f32 (bf16x3-split MFMA, fp32 accumulate, all three products)",
               bound="bf16 MFMA / 6 (exact 3-piece split: six bf16 products per fp32 product)", frac=round(tf_s / PEAK_SPLIT, 4),
               backward_weight=dict(ms=round(ms_wgs, 3), achieved=round(tf_ws, 2), peak=round(PEAK_SPLIT, 1), unit="TFLOP/s",
                                    bound="bf16 MFMA / 6", frac=round(tf_ws / PEAK_SPLIT, 4)),
               fp32_mfma_backward_weight=dict(ms=round(ms_wg, 3), achieved=round(tf_w, 2), peak=PEAK_FP32, unit="TFLOP/s", bound="fp32 MFMA",
                                              frac=round(tf_w / PEAK_FP32, 4), note="--hip.conv3x3_split!"),
               fp32_mfma_forward_backward_data=dict(ms=round(ms_f32, 3), achieved=round(tf_f, 2), peak=PEAK_FP32, unit="TFLOP/s", bound="fp32 MFMA",
                                                    frac=round(tf_f / PEAK_FP32, 4), note="--hip.conv3x3_split!"),
               layers=rows)
    if with_cpu:
        torch.set_num_threads(cpu_threads())
        c, side, batch = 128, 28, 8
        xc, wc = torch.randn(batch, c, side, side), torch.randn(c, c, 3, 3) * 0.05
        dt, k = _cpu_time(lambda: torch.nn.functional.conv2d(xc, wc, None, 1, 1), budget_s=4.0)
        out["cpu"] = dict(value=round(2.0 * batch * side * side * c * c * 9 / dt / 1e12, 4), unit="TFLOP/s", cores=cpu_threads(), kind="port",
                          sample="torch conv2d forward, 128 channels 28x28, 8 images, %d timed runs" % k, gpu_value=out["achieved"])
    return out


def run_all(with_cpu=True, with_reference_gpu=False):
    out = {}
    for name, fn in (("chamfer_b1", lambda: chamfer(1, with_cpu=with_cpu, with_reference_gpu=with_reference_gpu)),
                     ("chamfer_b32", lambda: chamfer(32, with_cpu=False, with_reference_gpu=with_reference_gpu)),
                     ("clip_vit_b32", lambda: clip_vit(32, with_cpu=with_cpu)), ("clip_vit_b32_batch256", lambda: clip_vit(256, with_cpu=False)),
                     ("clip_vit_l14_b32", lambda: clip_vit(32, with_cpu=False, model="ViT-L/14")), ("render_eval_128", lambda: render_eval_128(32, with_cpu=with_cpu)),
                     ("level_grid_100", lambda: level_grid_100(with_cpu=with_cpu)), ("marching_cubes_100", lambda: marching_cubes_100(32)),
                     ("resnet_conv3x3", lambda: resnet_conv3x3(with_cpu=with_cpu))):
        out[name] = fn()
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(run_all("--no-cpu" not in sys.argv, "--with-reference-gpu" in sys.argv)))
