"""The data-parallel training step of the product Graph on TWO ranks with real kernels (model/runner.py:121,255 of the reference:
DistributedDataParallel over NCCL).  The GPU box has one MI355X, so both ranks share cuda:0 and the collectives travel over gloo
(RCCL refuses two ranks on one device); everything else -- the flat gradient buffer, the early segment issued from inside backward on
a side stream, the persistent BatchNorm buffer broadcast, fused Adam on the flat views -- is the code path `bench.py --gpus N` runs.

Checked per schedule (overlapped / one all-reduce after backward):
  * ranks start from different initialisations and different batches and end two steps later with bit-identical parameters (the
    data-parallel invariant); the BatchNorm running statistics follow rank 0 at the next step's broadcast;
  * the exchanged gradient equals the mean of the ranks' local gradients, element for element (non-overlapped schedule, where the
    local gradient is still observable before the exchange);
  * the collective count per step (2 overlapped, 1 otherwise);
and across schedules: the overlapped exchange ends in the SAME parameters, bit for bit, as the single all-reduce."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, overlap, outdir, out):
    import numpy as np
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    failed = []
    try:
        torch.cuda.set_device(0)
        from shapeclipper_amd import synthetic
        from shapeclipper_amd.model.runner import Runner
        from shapeclipper_amd.utils import options, util
        from shapeclipper_amd.utils.util import EasyDict as edict
        per_rank = 2
        extra = ["--hip.overlap_allreduce"] if overlap else ["--hip.overlap_allreduce!"]      # the default is the single all-reduce (round 4)
        opt = options.set(options.parse_arguments([
            "--yaml=%s/options/pix3d/config.yaml" % ROOT, "--name=pytest_two_ranks", "--output_root=/tmp/sc_pytest_%d" % rank,
            "--batch_size=%d" % (per_rank * world), "--tb!", "--arch.enc_pretrained!"] + extra), verbose=False)
        opt.device, opt.world_size, opt.port = 0, world, 0
        opt.freq.scalar, opt.freq.ckpt_latest = 0, 10 ** 9
        torch.manual_seed(100 + rank); np.random.seed(0)      # different weights per rank: the reducer's broadcast must fix that
        runner = Runner(opt)
        if opt.batch_size != per_rank: failed.append("batch_size %d" % opt.batch_size)
        runner.build_networks(opt)
        runner.setup_optimizer(opt)
        runner.graph.train()
        runner.it, runner.ep, runner.best_val = 1, 0, 0.0
        runner.timer = edict(start=time.time(), it_mean=None)
        red = runner.reducer
        if red is None or red.overlap != overlap: failed.append("reducer / schedule")
        batch = util.move_to_device(synthetic.make_batch(opt, per_rank, seed=rank, training=True), "cuda:0")
        if not overlap:                                      # the local gradient is observable just before the one all-reduce
            orig = red.all_reduce
            seen = {}

            def checked():
                local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in red.params]).cpu()
                orig()
                gathered = [torch.zeros_like(local) for _ in range(world)]
                dist.all_gather(gathered, local)
                want = (gathered[0] + gathered[1]) * (1.0 / world)
                seen["diff"] = float((red.flat.cpu() - want).abs().max())
                seen["differ"] = float((gathered[0] - gathered[1]).abs().max())
            red.all_reduce = checked
        for _ in range(2):
            opt.H, opt.W = opt.image_size
            runner.train_iteration(opt, edict(batch), None)
        runner.check_finite()
        torch.cuda.synchronize()
        if not overlap:
            if seen.get("diff", 1.0) != 0.0: failed.append("exchanged gradient != mean of the local ones (%r)" % seen.get("diff"))
            if not seen.get("differ", 0.0) > 0.0: failed.append("the ranks' local gradients do not differ: the check is vacuous")
        if red.collectives != (4 if overlap else 2): failed.append("collectives %d" % red.collectives)
        g = runner.graph.module
        if not all(p.grad is None or p.grad.data_ptr() >= red.flat.data_ptr() for p in g.parameters()): failed.append("grads outside the flat buffer")
        params = torch.cat([p.detach().reshape(-1) for p in g.parameters()]).cpu()      # module order (the flat buffer's order depends on the schedule)
        # running statistics follow rank 0 at the START of every step (DDP's broadcast_buffers=True), so after the last step each rank
        # holds its own batch's update: they differ now and agree after the next step's broadcast
        mine = red.buf_flat.detach().cpu().clone()
        theirs = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(theirs, mine)
        if torch.equal(theirs[0], theirs[1]): failed.append("running statistics of the two ranks' different batches are equal: vacuous")
        red.broadcast_buffers()
        bufs = red.buf_flat.detach().cpu().clone()
        if rank == 0 and not torch.equal(bufs, mine): failed.append("the broadcast changed rank 0's buffers")
        for name, t in (("parameters", params), ("BatchNorm buffers", bufs)):
            both = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(both, t)
            if not torch.equal(both[0], both[1]): failed.append("%s differ between the ranks (max %g)" % (name, float((both[0] - both[1]).abs().max())))
        if not bool(torch.isfinite(params).all()): failed.append("non-finite parameters")
        if rank == 0:
            torch.save(params, os.path.join(outdir, "params_%d.pt" % int(overlap)))
    except Exception as e:      # noqa: BLE001
        import traceback
        failed.append("%s: %s\n%s" % (type(e).__name__, e, traceback.format_exc()))
    out[rank] = failed
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_training_step_on_one_gpu(tmp_path):
    outdir = str(tmp_path)
    for overlap in (False, True):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), overlap, outdir, out), nprocs=2, join=True)
        assert out[0] == [] and out[1] == [], "overlap=%s: rank 0 %s, rank 1 %s" % (overlap, out[0], out[1])
    a, b = torch.load(os.path.join(outdir, "params_0.pt")), torch.load(os.path.join(outdir, "params_1.pt"))
    assert torch.equal(a, b), "overlapped exchange and single all-reduce end in different parameters: max %g" % float((a - b).abs().max())


def _eval_worker(rank, world, port, outdir, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    failed = []
    try:
        torch.cuda.set_device(0)
        from shapeclipper_amd.model.runner import Runner
        from shapeclipper_amd.utils import options
        o = options.set(options.parse_arguments(["--yaml=%s/options/pix3d/config.yaml" % ROOT, "--name=pytest_eval_rank%d" % rank,
                                                 "--output_root=%s" % outdir, "--arch.enc_pretrained!", "--data.dataset=synthetic",
                                                 "--eval.vox_res=16", "--eval.num_points=1000", "--tb!"]), verbose=False)
        o.device, o.world_size, o.port = 0, 1, 0
        torch.manual_seed(0)                                  # no checkpoint: the same random weights on both ranks
        r = Runner(o)
        r.load_dataset(o, eval_split="test")
        r.build_networks(o)
        r.evaluate(o, ep=0)                                   # the reference's single-process flow, whole test set, on every rank
        single = open(os.path.join(o.output_path, "chamfer.txt")).read()
        single_f = open(os.path.join(o.output_path, "f_score.txt")).read()
        os.remove(os.path.join(o.output_path, "chamfer.txt"))
        o.world_size = world
        r.evaluate_sharded(o, ep=0)                           # samples idx % 2 == rank, one gather, rank 0 writes
        if rank == 0:
            sharded = open(os.path.join(o.output_path, "chamfer.txt")).read()
            va = torch.tensor([[float(x) for x in l.split()] for l in single.strip().splitlines()])
            vb = torch.tensor([[float(x) for x in l.split()] for l in sharded.strip().splitlines()])
            if va.shape != vb.shape or va.shape[0] != len(r.test_data): failed.append("records %s vs %s" % (tuple(va.shape), tuple(vb.shape)))
            elif not torch.allclose(va, vb, atol=1e-6): failed.append("chamfer.txt differs: max %g" % float((va - vb).abs().max()))
            if open(os.path.join(o.output_path, "f_score.txt")).read() != single_f: failed.append("f_score.txt differs")
        elif os.path.exists(os.path.join(o.output_path, "chamfer.txt")):
            failed.append("rank 1 wrote chamfer.txt")
    except Exception as e:      # noqa: BLE001
        import traceback
        failed.append("%s: %s\n%s" % (type(e).__name__, e, traceback.format_exc()))
    out[rank] = failed
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_sharded_evaluation_on_one_gpu(tmp_path):
    """BASELINE config[4] (evaluation sharded over the ranks) with two ranks on the box's one GPU: each rank evaluates every second sample
    (level grid, marching cubes, surface samples, Chamfer, F-score -- all on the device), the records are gathered once, and rank 0's
    chamfer.txt / f_score.txt equal the single-process evaluation's (utils/eval_3D.py, model/runner.py:evaluate of the reference)."""
    out = mp.Manager().dict()
    mp.spawn(_eval_worker, args=(2, _free_port(), str(tmp_path), out), nprocs=2, join=True)
    assert out[0] == [] and out[1] == [], "rank 0 %s, rank 1 %s" % (out[0], out[1])


def _run(cmd, env, timeout=900):
    import subprocess
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    if r.returncode != 0:
        print("---- failed command: %s\n---- stdout tail:\n%s\n---- stderr tail:\n%s" % (" ".join(cmd), r.stdout[-3000:], r.stderr[-6000:]))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout + r.stderr


@pytest.mark.timeout(2400)
def test_train_and_evaluate_scripts_under_the_launcher(tmp_path):
    """The drop-in scripts end to end (train.py:37-41 / evaluate.py:16-18 of the reference; BASELINE config[3] and config[4] in small):
    `python -m torch.distributed.run --nproc-per-node 2 train.py ...` on the synthetic dataset -- two ranks on the box's one GPU over gloo
    (SHAPECLIPPER_DIST_BACKEND) -- trains two epochs with the sharded sampler, evaluates and checkpoints on rank 0; evaluate.py then restores
    the best checkpoint (a) in one process and (b) sharded over two ranks: same chamfer.txt, and the CD the training run reported as best; then (c) at
    config[4]'s own resolution, `--eval.vox_res=100`, in one process and sharded over eight ranks."""
    import re
    import sys
    common = ["--yaml=%s/options/pix3d/config.yaml" % ROOT, "--name=e2e", "--output_root=%s" % tmp_path, "--data.dataset=synthetic",
              "--data.synthetic_len=8", "--batch_size=4", "--max_epoch=2", "--freq.eval=1", "--eval.vox_res=16", "--eval.num_points=1000",
              "--tb!", "--arch.enc_pretrained!"]
    env = dict(os.environ, MIOPEN_LOG_LEVEL="1", MIOPEN_FIND_MODE="FAST", SHAPECLIPPER_DIST_BACKEND="gloo")
    launch = lambda: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", str(_free_port())]
    log = _run(launch() + [os.path.join(ROOT, "train.py")] + common, env)
    assert log.count("TRAINING DONE") == 1, log[-2000:]                  # rank 0 alone reports
    best = float(re.search(r"Best CD: ([0-9.]+)", log).group(1))
    out = os.path.join(str(tmp_path), "pix3d_output", "e2e")
    for f in ("best.ckpt", "latest.ckpt", "options.yaml"):
        assert os.path.exists(os.path.join(out, f)), f
    read = lambda: torch.tensor([[float(x) for x in l.split()] for l in open(os.path.join(out, "chamfer.txt")).read().strip().splitlines()])
    _run([sys.executable, os.path.join(ROOT, "evaluate.py")] + common + ["--resume"], dict(env, SHAPECLIPPER_DIST_BACKEND="nccl"))
    single = read()
    os.remove(os.path.join(out, "chamfer.txt"))
    _run(launch() + [os.path.join(ROOT, "evaluate.py")] + common + ["--resume"], env)
    sharded = read()
    assert single.shape == sharded.shape == (8, 3) and torch.allclose(single, sharded, atol=1e-6), (single, sharded)
    cd = float((single[:, 1].mean() + single[:, 2].mean()) / 2)
    assert abs(cd - best) < 2e-4, (cd, best)                             # the checkpoint evaluate.py restores is the one training called best
    # BASELINE config[4] at its own sizes: `evaluate.py --eval.vox_res=100` (1,030,301 grid points per sample through the pre-split value
    # chain, marching cubes, 100k x 100k Chamfer) in one process and sharded over EIGHT ranks (one sample each; the ranks share the box's
    # one GPU over gloo) -- same records
    big = [a for a in common if not a.startswith(("--eval.vox_res", "--batch_size"))] + ["--eval.vox_res=100", "--batch_size=8", "--resume"]   # (the training batch is divided by the ranks)
    launch8 = lambda: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                       "--master-port", str(_free_port())]
    os.remove(os.path.join(out, "chamfer.txt"))
    _run([sys.executable, os.path.join(ROOT, "evaluate.py")] + big, dict(env, SHAPECLIPPER_DIST_BACKEND="nccl"))
    single100 = read()
    os.remove(os.path.join(out, "chamfer.txt"))
    _run(launch8() + [os.path.join(ROOT, "evaluate.py")] + big, dict(env, OMP_NUM_THREADS="4"), timeout=1200)
    sharded100 = read()
    assert single100.shape == sharded100.shape == (8, 3) and torch.allclose(single100, sharded100, atol=1e-6), (single100, sharded100)
    assert not torch.allclose(single100[:, 1:], single[:, 1:], atol=1e-6)        # the finer grid really was evaluated
