"""python pretrain.py --yaml=options/pix3d/config.yaml --name=pretrain [--pre.viewpoint!] [--data.dataset=synthetic]"""
from shapeclipper_amd.cli import pretrain_main

if __name__ == "__main__":
    pretrain_main()
