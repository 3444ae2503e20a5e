"""python train.py --yaml=options/pix3d/config.yaml --name=<run> [--load=<pretrain ckpt>] [--a.b=v ...]
(also under `python -m torch.distributed.run --nproc-per-node N`; see shapeclipper_amd/cli.py)"""
from shapeclipper_amd.cli import train_main

if __name__ == "__main__":
    train_main()
