"""python train.py --yaml=options/pix3d/config.yaml --name=<run> [--load=<pretrain ckpt>] [--a.b=v ...]

One process per visible GPU (torch.multiprocessing.spawn or torchrun: RANK/WORLD_SIZE in the
environment are honoured); gradients are combined with a single RCCL all-reduce per step."""
import os
import sys

import torch
import torch.multiprocessing as mp

import utils.options as options
from utils.util import is_port_in_use, log
import model.runner


def main_worker(rank, world_size, port, opt):
    opt.device = rank
    opt.world_size = world_size
    opt.port = port
    if torch.cuda.is_available():
        torch.cuda.set_device(rank)
    trainer = model.runner.Runner(opt)
    trainer.load_dataset(opt)
    trainer.build_networks(opt)
    trainer.setup_optimizer(opt)
    trainer.restore_checkpoint(opt)
    trainer.setup_visualizer(opt)
    trainer.train(opt)


def main():
    log.process(os.getpid())
    log.title("[{}] (training)".format(sys.argv[0]))
    opt = options.set(opt_cmd=options.parse_arguments(sys.argv[1:]))
    options.save_options_file(opt)
    port = 34567
    while is_port_in_use(port):
        port += 1
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:      # launched by torchrun
        import torch.distributed as dist
        rank, world = int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl")
        main_worker(rank, world, port, opt)
        return
    world_size = max(torch.cuda.device_count(), 1)
    if world_size == 1:
        main_worker(0, world_size, port, opt)
    else:
        mp.spawn(main_worker, nprocs=world_size, args=(world_size, port, opt))


if __name__ == "__main__":
    main()
