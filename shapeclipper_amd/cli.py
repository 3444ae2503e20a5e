"""Entry points behind the drop-in scripts `train.py`, `pretrain.py`, `evaluate.py` (same command lines as the reference's
scripts of those names).  One place decides how processes map to GPUs:

  * launched by `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment): one process per
    GPU already exists -> join the RCCL group and run rank-local work;
  * launched plainly on a multi-GPU node: spawn one worker per visible GPU (what the reference's train.py does);
  * one GPU / no GPU: run in-process.
"""
from __future__ import annotations

import contextlib
import os
import sys

import torch

from .utils import options
from .utils.util import is_port_in_use, log


def _free_port(start=34567):
    port = start
    while is_port_in_use(port):
        port += 1
    return port


def _announce(what, argv0):
    log.process(os.getpid())
    log.title("[{}] ({})".format(argv0, what))


def _under_torchrun():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def _device_scope(opt):
    on_gpu = torch.cuda.is_available() and (isinstance(opt.device, int) or str(opt.device).startswith("cuda"))
    return torch.cuda.device(opt.device) if on_gpu else contextlib.nullcontext()


# ---- training ------------------------------------------------------------------------------------------------------
def _launcher_ranks():
    """(rank, device index, world size) of a process started by torch.distributed.run: one rank per GPU over RCCL; with
    SHAPECLIPPER_DIST_BACKEND=gloo the ranks share the GPUs that exist (tests/test_gpu_two_ranks.py on a one-GPU box)."""
    from .utils import util
    rank, local, world = int(os.environ.get("RANK", os.environ["LOCAL_RANK"])), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    if util.dist_backend() != "nccl" and torch.cuda.is_available():
        local %= torch.cuda.device_count()
    return rank, local, world


def _train_worker(rank, world_size, port, opt, device=None):
    from .model.runner import Runner
    opt.device, opt.world_size, opt.port = (rank if device is None else device), world_size, port
    if device is not None:
        opt.rank = rank
    if torch.cuda.is_available():
        torch.cuda.set_device(opt.device)
    runner = Runner(opt)
    for stage in ("load_dataset", "build_networks", "setup_optimizer", "restore_checkpoint", "setup_visualizer", "train"):
        getattr(runner, stage)(opt)


def train_main(argv=None):
    argv = sys.argv if argv is None else argv
    _announce("training", argv[0])
    opt = options.set(opt_cmd=options.parse_arguments(argv[1:]))
    options.save_options_file(opt)
    port = _free_port()
    if _under_torchrun():
        import torch.distributed as dist
        from .utils import util
        rank, local, world = _launcher_ranks()
        torch.cuda.set_device(local)
        dist.init_process_group(util.dist_backend())
        _train_worker(rank, world, port, opt, device=local)
        return
    n_gpus = max(torch.cuda.device_count(), 1)
    if n_gpus == 1:
        _train_worker(0, 1, port, opt)
    else:
        torch.multiprocessing.spawn(_train_worker, nprocs=n_gpus, args=(n_gpus, port, opt))


# ---- sphere pre-training -------------------------------------------------------------------------------------------------
def pretrain_main(argv=None):
    from .model import pretrainer
    argv = sys.argv if argv is None else argv
    _announce("training", argv[0])
    opt = options.set(opt_cmd=options.parse_arguments(argv[1:]))
    options.save_options_file(opt)
    with _device_scope(opt):      # (the reference enters torch.cuda.device unconditionally and cannot run with --cpu)
        runner = pretrainer.Runner(opt)
        for stage in ("load_dataset", "build_networks", "setup_optimizer", "train"):
            getattr(runner, stage)(opt)


# ---- evaluation ----------------------------------------------------------------------------------------------------------
def evaluate_main(argv=None):
    """Single process: the reference's flow.  Under torchrun: the test set is sharded over the ranks and the per-sample
    metrics are gathered once (Runner.evaluate_sharded, BASELINE config[4])."""
    from .model.runner import Runner
    argv = sys.argv if argv is None else argv
    _announce("evaluating", argv[0])
    opt = options.set(opt_cmd=options.parse_arguments(argv[1:]))
    opt.device, opt.world_size, opt.port = 0, 1, _free_port()
    sharded = _under_torchrun() and int(os.environ["WORLD_SIZE"]) > 1
    if sharded:
        import torch.distributed as dist
        from .utils import util
        opt.rank, opt.device, opt.world_size = _launcher_ranks()
        torch.cuda.set_device(opt.device)
        dist.init_process_group(util.dist_backend())
    with _device_scope(opt):
        runner = Runner(opt)
        runner.load_dataset(opt, eval_split="test")
        runner.test_data.id_filename_mapping(opt, os.path.join(opt.output_path, "data_list.txt"))
        runner.build_networks(opt)
        runner.restore_checkpoint(opt, best=True, evaluate=True)
        runner.setup_visualizer(opt)
        if sharded:
            runner.reducer = None
            runner.evaluate_sharded(opt, ep=0)
            torch.distributed.destroy_process_group()
        else:
            runner.evaluate(opt, ep=0)
