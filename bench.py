#!/usr/bin/env python
"""bench.py -- train-step images/sec of the ShapeClipper hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1, either way:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's command)
                        python bench.py --gpus N ...   (no launcher environment: bench.py starts the N ranks itself, self_launch())

A "step" is one full training iteration on a synthetic Pix3D-shaped batch that is already resident in
HBM: Graph.forward(training=True) (ResNet-34 encoder, ResNet-18 estimator on stock PyTorch-ROCm; TWO
HIP training renders of 512 rays x 64 samples per image incl. the eikonal branch) -> losses ->
backward (hand-written HIP backward kernels) -> one flat RCCL all-reduce of all gradients (N > 1) ->
Adam.  Batch = 32 images per GPU (BASELINE.json configs[1]/[2]/[3]; weak scaling).

One JSON line on rank 0; see the repository prompt for the contract.  Extras:
  roofline      dominant hand-written kernel: achieved = algorithmic FLOPs / mean
                launch duration measured with events on the launch stream inside the timed region
  cpu_baseline  the CPU oracle (pure PyTorch restatement of the reference, oracle/reference_ops.py) timed
                on this box's host cores on a bounded sample (hot path only)
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # immediate-mode conv selection: no exhaustive MIOpen search on a fresh box
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic (un-padded, reference-dense) FLOPs per sample point of each entry point -- DESIGN.md "kernels"
SDF_VALUE = 2 * 40320          # SDFNetwork.forward incl. latent columns (80,640 FLOP/pt)
SDF_GRAD = 2 * 40320           # d(sdf)/dx reverse sweep through the same layers
RGB_VALUE = 2 * 19072
FLOPS_PER_POINT = {
    "sc_sdf_forward": SDF_VALUE + SDF_GRAD,
    "sc_sdf_backward": 2 * (SDF_VALUE + SDF_GRAD) // 2,     # input-gradient half of the double backward
    # fused backward: input-gradient half + the weight-gradient GEMMs of the SDF network (864 MFMAs per 16 points)
    "sc_sdf_backward_fused": 2 * (SDF_VALUE + SDF_GRAD) // 2 + 864 * 2048 // 16,
    "sc_rgb_composite_forward": RGB_VALUE,
    "sc_rgb_composite_backward": RGB_VALUE + RGB_VALUE,     # recompute + input-gradient sweep
}
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2 dense peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=0, help="images of the CPU baseline step (0: 8 of the batch: 1 warm-up + 3 timed steps inside ~30 s)")
    ap.add_argument("--sustained", type=int, default=200, help="extra steps timed after the K-step region (0: off)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the other SURVEY 8(d) workloads (tools/workloads.py)")
    ap.add_argument("--with-reference-gpu", action="store_true", help="Chamfer workloads: also time the reference's own extension built for "
                    "gfx950 (oracle/_ref; maps a binary built from the reference's sources into this process -- off by default)")
    ap.add_argument("--workloads-only", action="store_true", help="only those workloads (the rocprofv3 command of profiles/)")
    ap.add_argument("--no-alt", action="store_true", help="skip the second measurement with --hip.conv3x3_split! (fp32-MFMA convolutions)")
    ap.add_argument("--alt-steps", type=int, default=60)
    ap.add_argument("--opt", action="append", default=[], help="extra option override(s), e.g. --opt=--hip.fused_backward!")
    return ap.parse_args()


def _host_cpu():
    """(physical cores, logical cpus, model name) of this host: BASELINE.md section 3 asks for the core count and the lscpu model."""
    logical = os.cpu_count() or 1
    physical, model = None, "unknown"
    try:
        import psutil
        physical = psutil.cpu_count(logical=False)
    except Exception:
        pass
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return int(physical or logical), logical, model


# entry points that are one kernel under several names (a compile-time form per configuration): ONE name in the line
STABLE_NAME = {
    "sc_rgb_composite_backward_v3": "sc_rgb_composite_backward",              # output layer's gradient folded in
    "sc_rgb_composite_backward_fused": "sc_rgb_composite_backward",           # RGB weight gradients formed in the kernel (round 5)
    "sc_rgb_composite_backward_fused_stash": "sc_rgb_composite_backward",     # ... reading the activations the forward parked (round 5)
    "sc_rgb_composite_backward_fused_split": "sc_rgb_composite_backward",     # ... with the reverse chain from pre-split bf16x3 fragments (round 6)
    "sc_rgb_composite_forward_stash": "sc_rgb_composite_forward",             # the forward that parks them
    "sc_sdf_forward_stream": "sc_sdf_forward",                                # value + feature + d sdf/dx from streamed pre-split fragments (round 6)
    "sc_rgb_composite_forward_split": "sc_rgb_composite_forward",             # ... with the RGB network from pre-split bf16x3 fragments (round 6)
}
# what holds each kernel below its roof (phase profiles under profiles/): a statement about THAT kernel only
LIMITER = {
    "sc_sdf_backward_fused": "issue: 87 k cycles per 4 tiles against 60 k of MFMAs; both roles of the workgroup are busy (chain waves wait 11 k, "
                             "weight-gradient waves 23 k: profiles/r05_bwdw_phase_profile_final.txt); restructurings that shift work between the "
                             "roles were built and are slower (DESIGN.md 4.1)",
    "sc_sdf_forward": "round 6: exact bf16x3 split arithmetic from pre-split weight fragments streamed through LDS (csrc/sdf_fwd_stream.hip), 1.2x the "
                      "fp32-MFMA kernel; what is left is vector work (softplus, the activation splits, the accurate sincosf of the value path) and "
                      "one tile per wave at 256 registers (profiles/r06_sdf_stream_ablation.txt)",
    "sc_sdf_backward": "issue: same chain as the fused form without the weight-gradient role (DESIGN.md 4.1)",
    "sc_rgb_composite_forward": "round 6: RGB network from pre-split bf16x3 fragments resident in LDS (1.3x at the training shape, 1.46x at the "
                                "evaluation shape); vector work + the activation stash it writes (768 B per point, profiles/r06_traffic.json)",
    "sc_rgb_composite_backward": "issue: reverse sweep (round 6: its transposed products from pre-split bf16x3 fragments) + fp32 weight-gradient MFMAs "
                                 "in one workgroup; reads the parked activations (DESIGN.md 4.1, 4.1.2)",
}


def timing_report(durations, steps, n_pts_main, batch, profiles_dir=None):
    """Per-entry-point GPU time and the roofline object(s) of the bench line from `durations` = {C-ABI entry point: [ms of every launch in the
    timed region]} (shapeclipper_amd._lib.TIMING, events on the launch stream).  Pure host arithmetic: tests/test_host_logic.py feeds it a
    synthetic dict in which every entry point dominates in turn (VERDICT r05 #1b: the round-5 line died on a renamed key).

    All bookkeeping happens in ONE namespace: the raw names are merged into their stable names first, then everything reads `merged`."""
    merged = {}
    for name, durs in durations.items():
        if durs:
            merged.setdefault(STABLE_NAME.get(name, name), []).extend(durs)
    per = {n: dict(calls=len(d), total_ms=sum(d), mean_ms=sum(d) / len(d), max_ms=max(d)) for n, d in merged.items()}
    scored = [n for n in per if n in FLOPS_PER_POINT]
    if not scored:
        return per, dict(error="no scored entry point in the timed region: %s" % sorted(per)), None
    dom = max(scored, key=lambda n: per[n]["total_ms"])
    # the big launches are the two main renders of a step (eikonal launches are 32x smaller): use them
    big = sorted(merged[dom], reverse=True)[:2 * steps]
    mean_big = sum(big) / len(big)
    achieved = FLOPS_PER_POINT[dom] * n_pts_main / (mean_big * 1e-3) / 1e12
    traffic, traffic_source = None, None
    pdir = profiles_dir or os.path.join(ROOT, "profiles")
    for prof in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:   # HBM-side bytes per launch measured with rocprofv3 --pmc on this workload (profiles/, see its _comment)
            with open(os.path.join(pdir, prof)) as f:
                entry = json.load(f)["kernels"].get(dom)
            if entry is None:
                continue
            traffic = entry["traffic_bytes"] if batch == 32 else None       # the committed figure is the B=32 main render's
            traffic_source = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, corrected per MI355X_MICROARCH.md; a " \
                             "committed constant, NOT measured in this run)" % prof
            break
        except Exception:
            pass
    # Which roof: the kernel's algorithmic bytes are ~0 (a fused backward would need none of its hand-off tensors), so
    # its roof is the fp32 matrix pipe; but when the bytes it actually moves run at >= 75 % of the achievable HBM rate
    # (6.3 TB/s measured float4 copy, MI355X_MICROARCH.md) the HBM traffic it creates is what sets its time: say so.
    hbm_rate = traffic / (mean_big * 1e-3) / 1e9 if traffic else None
    bound = "hbm" if (hbm_rate and hbm_rate >= 0.75 * 6300.0) else "mfma"
    roofline = dict(kernel=dom, bound=bound, achieved=round(achieved, 2), peak=PEAK_FP32_MFMA_TFLOPS,
                    unit="TFLOP/s", frac=round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), traffic=traffic, traffic_source=traffic_source,
                    limiter=LIMITER.get(dom), launch_ms=round(mean_big, 4), points_per_launch=n_pts_main,
                    flops_per_point=FLOPS_PER_POINT[dom],
                    hbm_GBps_of_measured_traffic=round(hbm_rate, 1) if hbm_rate else None)
    # second roofline: the separate weight-gradient GEMM launches (wgrad.hip).  With the fused SDF and RGB backward kernels (default) none
    # remain; with --hip.fused_backward! the SDF network's 9 and the RGB network's 3 are there (1064 MFMAs per 16 points).  Scored on
    # FLOPs: the operand bytes they stream exist only because the backward kernels materialise them.
    roofline_wgrad = None
    if "sc_wgrad" in merged:
        fused = "sc_sdf_backward_fused" in merged
        per_render, mfmas = (3, 240) if fused else (12, 1064)
        wd = sorted(merged["sc_wgrad"], reverse=True)[:2 * per_render * steps]
        per_render_ms = sum(wd) / (2 * steps)
        flops_pt = mfmas * 2048 // 16
        wg_tf = flops_pt * n_pts_main / (per_render_ms * 1e-3) / 1e12
        roofline_wgrad = dict(kernel="sc_wgrad (%d launches of a main render: %s)" % (per_render, "RGB network" if fused else "SDF + RGB networks"),
                              bound="mfma", achieved=round(wg_tf, 2), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                              frac=round(wg_tf / PEAK_FP32_MFMA_TFLOPS, 4), ms_per_render=round(per_render_ms, 4),
                              flops_per_point=flops_pt)
    return per, roofline, roofline_wgrad


def cpu_baseline(batch, rays=512, explicit=False, budget_s=75.0):
    """Oracle hot path on the host cores, SURVEY 8(d) / BASELINE.md section 3: (i) ONE render call fwd + bwd and (ii) the TWO render calls of a
    training step (incl. eikonal) on the same synthetic images, torch.set_num_threads(all PHYSICAL cores), 1 warm-up + up to 3 timed
    iterations each, images/s = B / best time.  Bounded sample: 8 images of the batch (the oracle's cost is linear in the images -- every
    image is an independent set of 512 rays) and a wall-time budget; when the all-cores pool is the slow one (many small operators and a
    double backward scale badly past a few dozen threads: 256 threads on this pool's hosts were 200x slower than 8), a second leg
    on 32 threads is reported beside it -- `value` stays the all-physical-cores figure the contract names, `best` says which pool won."""
    from oracle import reference_ops as R
    physical, logical, model = _host_cpu()
    try:
        import psutil
        free_gb = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        free_gb = 16.0
    B = batch if explicit else min(batch, 8)
    while B > 1 and 1.5 * B > 0.5 * free_gb:          # ~0.6 GB of autograd state per image and render measured, x2.5 head room, half of what is free
        B //= 2
    cfg = R.Cfg()
    torch.manual_seed(0)
    Ws = {k: v.requires_grad_(True) for k, v in R.init_sdf_weights(cfg).items()}
    Wr = {k: v.requires_grad_(True) for k, v in R.init_rgb_weights(cfg).items()}
    beta = torch.tensor(0.1, requires_grad=True)
    az = (torch.rand(B) * 2 - 1) * 3.14159
    trig = lambda t: torch.stack([torch.cos(t), torch.sin(t)], 1)
    sd = (0.8 + 0.4 * torch.rand(B)).requires_grad_(True)
    intr = R.get_intr(cfg, torch.ones(B))
    zs, zr = torch.randn(B, 64, requires_grad=True), torch.randn(B, 64, requires_grad=True)
    ray_idx = torch.stack([torch.randperm(cfg.H * cfg.W)[:rays] for _ in range(B)])

    def render_call():
        pose = R.pose_from_trig(cfg, trig(az), trig(torch.zeros(B)), trig(torch.zeros(B)), sd)
        t_rand, eik_idx, eik_pts = R.draw_render_randoms(B * rays, 64, True)
        o = R.render(cfg, Ws, Wr, beta, pose, intr, sd, zs, zr, ray_idx, True, t_rand, eik_idx, eik_pts)
        (o["rgb"].sum() + o["mask"].sum() + o["normal"].sum() + ((o["grad_eikonal"] - 1) ** 2).mean()).backward()

    def leg(threads, budget):
        """best-of times of one render call and of a two-render step on `threads` threads, inside `budget` seconds of wall time"""
        torch.set_num_threads(threads)
        t0 = time.time()
        render_call()                              # warm-up (allocator, thread pool)
        warm = time.time() - t0
        one, two = [], []
        while len(one) < 3 and (not one or time.time() - t0 + warm < 0.4 * budget):
            t1 = time.time(); render_call(); one.append(time.time() - t1)
        while len(two) < 3 and (not two or time.time() - t0 + 2 * warm < budget):
            t1 = time.time(); render_call(); render_call(); two.append(time.time() - t1)
        return dict(threads=threads, one_render_images_per_s=round(B / min(one), 3), two_render_step_images_per_s=round(B / min(two), 3),
                    one_render_s=round(min(one), 3), two_render_step_s=round(min(two), 3), timed=[len(one), len(two)], warmup_s=round(warm, 2))

    t_all = time.time()
    main = leg(physical, 0.6 * budget_s if physical > 32 else budget_s)
    legs = [main]
    if physical > 32 and time.time() - t_all < budget_s:
        legs.append(leg(32, budget_s - (time.time() - t_all)))
    best = max(legs, key=lambda l: l["two_render_step_images_per_s"])
    return dict(value=main["two_render_step_images_per_s"],
                unit="images/s (oracle: the 2 training renders of a step only, fwd+bwd; the GPU `value` is a WHOLE step)",
                cores=physical, host_cpu_count=logical, cpu_model=model, kind="port",
                one_render_images_per_s=main["one_render_images_per_s"], s_per_step=main["two_render_step_s"],
                legs=legs, best=dict(threads=best["threads"], two_render_step_images_per_s=best["two_render_step_images_per_s"],
                                     one_render_images_per_s=best["one_render_images_per_s"]),
                sample="oracle hot path only (no encoders / optimizer): (i) one render call fwd+bwd and (ii) the 2 render calls of a training step "
                       "(512 rays x 64 samples, eikonal incl.), B=%d of the batch's images (host memory free %.0f GB), 1 warm-up + up to 3 timed "
                       "iterations each, best time, torch.set_num_threads(%d = all physical cores of %s)%s"
                       % (B, free_gb, physical, model, "; second leg on 32 threads" if len(legs) > 1 else ""))


def _guarded(fn):
    """A secondary object of the line (CPU baseline, other workloads) must not cost the headline: its failure is reported in its place."""
    try:
        return fn()
    except Exception as exc:
        import traceback
        sys.stderr.write(traceback.format_exc())
        return dict(error="%s: %s" % (type(exc).__name__, exc))


def _workloads():
    import importlib.util
    spec = importlib.util.spec_from_file_location("sc_bench_workloads", os.path.join(ROOT, "tools", "workloads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_runner(batch_per_gpu, rank=0, local=0, world=1, extra=()):
    """Runner + options + HBM-resident synthetic batch for the Pix3D training configuration."""
    from shapeclipper_amd import synthetic
    from shapeclipper_amd.model.runner import Runner
    from shapeclipper_amd.utils import options, util
    from shapeclipper_amd.utils.util import EasyDict as edict

    opt = options.set(options.parse_arguments([
        "--yaml=%s/options/pix3d/config.yaml" % ROOT, "--name=bench", "--output_root=/tmp/sc_bench_%d" % rank,
        "--batch_size=%d" % (batch_per_gpu * world), "--tb!", "--arch.enc_pretrained!"] + list(extra)), verbose=False)
    opt.device, opt.world_size, opt.port = local, world, 0
    opt.freq.scalar, opt.freq.ckpt_latest = 0, 10 ** 9
    torch.manual_seed(rank)
    runner = Runner(opt)                       # divides batch_size by world_size
    runner.build_networks(opt)
    runner.setup_optimizer(opt)
    runner.graph.train()
    runner.it, runner.ep, runner.best_val = 1, 0, 0.0
    runner.timer = edict(start=time.time(), it_mean=None)
    synthetic.cap_host_threads()
    batch = util.move_to_device(synthetic.make_batch(opt, batch_per_gpu, seed=rank, training=True), "cuda:%d" % local)
    return runner, opt, batch


def self_launch(a):
    """`python bench.py --gpus N` (N > 1) WITHOUT a launcher environment: start the N ranks ourselves -- re-exec this file under
    `python -m torch.distributed.run` exactly as the driver's N > 1 command does (one rank per GPU, rendezvous on 127.0.0.1, a free
    port), pass every argument through, forward the ranks' output and return their exit code.  Rank 0 prints the one JSON line."""
    import socket
    import subprocess
    backend = os.environ.get("SC_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < a.gpus:
        sys.exit("bench.py --gpus %d: this node exposes %d GPU(s); RCCL takes one rank per device (SC_BENCH_BACKEND=gloo runs the same "
                 "code path with the ranks sharing the GPUs that exist -- a functional check, not a measurement)" % (a.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver only supports dmabuf IPC (RCCL needs it across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: no launcher environment, starting %d ranks: %s\n" % (a.gpus, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        sys.exit("bench.py --gpus %d started with WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or with no "
                 "launcher at all (bench.py starts its own ranks)" % (a.gpus, world, a.gpus))
    # One rank per GPU over RCCL.  SC_BENCH_BACKEND=gloo (tests/test_gpu_bench_contract.py) runs the same N > 1 code path with
    # several ranks sharing the GPUs that exist -- the build box has one MI355X and RCCL refuses two ranks on one device.
    backend = os.environ.get("SC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend)

    from shapeclipper_amd import _lib
    from shapeclipper_amd.utils.util import EasyDict as edict
    if a.workloads_only:
        print(json.dumps(dict(workloads=_workloads().run_all(with_cpu=not a.no_cpu_baseline, with_reference_gpu=a.with_reference_gpu))))
        return
    runner, opt, batch = build_runner(a.batch, rank, local, world, a.opt)

    def step():
        opt.H, opt.W = opt.image_size
        return runner.train_iteration(opt, edict(batch), None)

    for _ in range(a.warmup):
        step()
    _lib.TIMING = {}
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        loss = step()
    host_dt = time.time() - t0          # host time to enqueue the K steps (the GPU is still working)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.time() - t0
    timing, _lib.TIMING = _lib.TIMING, None
    ranks = None
    if world > 1:
        # every rank's own clock over the K steps (its barrier wait included) and its host enqueue time: the line prints min / max so that
        # a straggler (a slower box, a starved RCCL kernel) is visible beside the MAX the contract asks for
        mine = torch.tensor([dt, host_dt], device="cuda")
        every = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        per_rank = [t[0].item() / a.steps * 1e3 for t in every]
        ranks = dict(ms_per_step_min=round(min(per_rank), 3), ms_per_step_max=round(max(per_rank), 3),
                     host_enqueue_ms_per_step_max=round(max(t[1].item() for t in every) / a.steps * 1e3, 3),
                     backend=torch.distributed.get_backend(), backend_world_size=torch.distributed.get_world_size(),
                     devices_visible=torch.cuda.device_count())
        t = torch.tensor([dt], device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    assert torch.isfinite(loss.all.detach()).item(), "non-finite loss in the timed region"
    # the K-step region above is a burst (clocks have not settled): time a longer run as well
    sustained = None
    if a.sustained > 0:
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(a.sustained):
            step()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        sdt = time.time() - t1
        if world > 1:
            t = torch.tensor([sdt], device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            sdt = t.item()
        sustained = dict(steps=a.sustained, ms_per_step=round(sdt / a.sustained * 1e3, 3),
                         value=round(a.batch * world / (sdt / a.sustained), 2))
    # `value` above is measured in the default arithmetic (round 3, VERDICT r02 ruling): every product of the step is an fp32 product
    # with fp32 accumulation; the forward / backward-data / backward-weight products of the 3x3 stride-1 convolutions are evaluated on the bf16 matrix
    # pipe from exact three-piece operand splits (x = p0 + p1 + p2, six exact piece products summed in fp32: error against float64 no
    # larger than the fp32-MFMA kernels', tests/test_gpu_conv.py), everything else on the fp32 matrix pipe.  Second measurement: the
    # same step with `--hip.conv3x3_split!`, i.e. fp32 MFMA instructions throughout.
    alt = None
    from shapeclipper_amd.model import resnet
    split_default = bool(resnet.HIP_CONV3X3_SPLIT)
    if world == 1 and not a.no_alt and split_default:
        resnet.HIP_CONV3X3_SPLIT = False
        try:
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            t1 = time.time()
            for _ in range(a.alt_steps):
                step()
            torch.cuda.synchronize()
            adt = (time.time() - t1) / a.alt_steps
            alt = dict(steps=a.alt_steps, ms_per_step=round(adt * 1e3, 3), value=round(a.batch / adt, 2), dtype="f32 (fp32 MFMA throughout)",
                       note="same step with --hip.conv3x3_split!: the 3x3 convolution forward / backward-data / backward-weight products on v_mfma_f32_32x32x2_f32 "
                            "instead of the exact bf16x3 split; not the headline")
        finally:
            resnet.HIP_CONV3X3_SPLIT = True
    # BASELINE config[1] ("Pix3D train.py bs16, 1 x MI355X") beside the headline: the same step at 16 images per GPU on its own runner.  The
    # step is GPU-bound at this size too (2 ms less host work per step did not move it: DESIGN.md section 6); what does not shrink with the
    # batch is the fixed part of every launch (~33 us per convolution outside its K loop), which is what the object says.
    config1 = None
    if world == 1 and not a.no_alt and a.batch == 32:
        import gc
        r16, o16, b16 = build_runner(16, rank, local, world, a.opt)
        from shapeclipper_amd.utils.util import EasyDict as _ed

        def step16():
            o16.H, o16.W = o16.image_size
            return r16.train_iteration(o16, _ed(b16), None)
        for _ in range(5):
            step16()
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(a.alt_steps):
            step16()
        torch.cuda.synchronize()
        cdt = (time.time() - t1) / a.alt_steps
        config1 = dict(workload="BASELINE config[1]: the same training step at bs16 on one GPU", steps=a.alt_steps, ms_per_step=round(cdt * 1e3, 3),
                       value=round(16 / cdt, 2), unit="images/s", note="GPU-bound like bs32 (A/B with 2 ms less host work per step: no change); the per-launch fixed costs of ~960 kernels "
                       "do not shrink with the batch (DESIGN.md 4.2); not the headline")
        del r16, o16, b16, step16
        gc.collect()
        torch.cuda.empty_cache()
    allreduce = None
    if world > 1 and runner.reducer is not None:      # the step's only exchange: one flat all-reduce
        flat = runner.reducer.flat
        for _ in range(2):
            torch.distributed.all_reduce(flat)
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(10):
            torch.distributed.all_reduce(flat)
        torch.cuda.synchronize()
        ar = (time.time() - t1) / 10
        nbytes = flat.numel() * 4
        t = torch.tensor([ar], device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        ar = t.item()
        from shapeclipper_amd import ops as _ops
        allreduce = dict(ms=round(ar * 1e3, 3), payload_bytes=nbytes, algo_GBps=round(nbytes / ar / 1e9, 1),
                         bus_GBps=round(2 * (world - 1) / world * nbytes / ar / 1e9, 1), peak_per_link_GBps=153.0,
                         note="the step's only exchange, timed ALONE (10 back-to-back collectives of the flat gradient buffer, MAX over ranks); "
                              "bus_GBps = 2 (w-1)/w x payload / time, the ring figure to hold against 7 xGMI links x 153 GB/s per GPU",
                         schedule="overlapped [early | late]" if runner.reducer.overlap else "single flat all-reduce after backward",
                         reserved_cus=int(opt.get("hip", {}).get("reserve_cus", 0) or 0), persistent_grid_cus=_ops.set_reserved_cus(
                             int(opt.get("hip", {}).get("reserve_cus", 0) or 0)))

    if rank == 0:
        ms = dt / a.steps * 1e3
        # a reporting bug must never cost the line: every N > 1 run prints its value even when the per-kernel bookkeeping throws
        try:
            durations = {name: [s.elapsed_time(e) for s, e, _ in evs] for name, evs in timing.items()}
            per, roofline, roofline_wgrad = timing_report(durations, a.steps, a.batch * opt.render.rand_sample * 64, a.batch)
        except Exception as exc:                              # pragma: no cover (tests/test_host_logic.py feeds timing_report every dominant kernel)
            per, roofline_wgrad = {}, None
            roofline = dict(error="%s: %s" % (type(exc).__name__, exc))
        out = dict(metric="train-step images/sec (Pix3D cfg, bs32/GPU)", value=round(a.batch * world / (dt / a.steps), 2),
                   unit="images/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="f32 (bf16x3-split MFMA, fp32 accumulate)" if split_default else "f32", data="synthetic",
                   config=dict(workload="Pix3D train step: bs32/GPU, 224x224 inputs, 512 rays x 64 samples, 2 renders "
                                        "(input + CLIP-NN view) + eikonal, ResNet-34 encoder + ResNet-18 estimator, Adam; fp32 arithmetic "
                                        "throughout" + ("; the 3x3 stride-1 convolution products (forward, backward-data, backward-weight) run on the bf16 matrix pipe "
                                                        "from exact 3-piece operand splits with fp32 accumulation (fp32-accurate), all other "
                                                        "products on fp32 MFMA" if split_default else " (fp32 MFMA)"),
                               global_batch=a.batch * world, rays_per_image=opt.render.rand_sample, samples_per_ray=64,
                               parallelism="dp%d" % world),
                   roofline=roofline, roofline_wgrad=roofline_wgrad, host_enqueue_ms_per_step=round(host_dt / a.steps * 1e3, 3),
                   hip_ms_per_step={k: round(v["total_ms"] / a.steps, 3) for k, v in sorted(per.items())},
                   sustained=sustained)
        if allreduce is not None:
            out["allreduce"] = allreduce
        if ranks is not None:
            out["ranks"] = ranks
        if alt is not None:
            out["fp32_mfma_convolutions"] = alt
        if config1 is not None:
            out["config1_bs16"] = config1
        if not a.no_cpu_baseline and world == 1:      # reported at N = 1 only (the other ranks must not wait for rank 0)
            out["cpu_baseline"] = _guarded(lambda: cpu_baseline(a.cpu_batch or a.batch, explicit=bool(a.cpu_batch)))
        if not a.no_workloads and world == 1:
            out["workloads"] = _guarded(lambda: _workloads().run_all(with_cpu=not a.no_cpu_baseline, with_reference_gpu=a.with_reference_gpu))
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
