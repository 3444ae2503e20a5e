"""python evaluate.py --yaml=options/pix3d/config.yaml --name=<run> --resume --eval.vox_res=100
(under `python -m torch.distributed.run --nproc-per-node N` the test set is sharded; see shapeclipper_amd/cli.py)"""
from shapeclipper_amd.cli import evaluate_main

if __name__ == "__main__":
    evaluate_main()
