"""python pretrain.py --yaml=options/pix3d/config.yaml --name=pretrain [--pre.viewpoint!] [--data.dataset=synthetic]"""
import contextlib
import os
import sys

import torch

import utils.options as options
from utils.util import log
import model.pretrainer

log.process(os.getpid())
log.title("[{}] (training)".format(sys.argv[0]))
opt = options.set(opt_cmd=options.parse_arguments(sys.argv[1:]))
options.save_options_file(opt)

# the reference enters torch.cuda.device(opt.device) unconditionally and crashes with --cpu (pretrain.py:15)
ctx = torch.cuda.device(opt.device) if str(opt.device).startswith("cuda") else contextlib.nullcontext()
with ctx:
    trainer = model.pretrainer.Runner(opt)
    trainer.load_dataset(opt)
    trainer.build_networks(opt)
    trainer.setup_optimizer(opt)
    trainer.train(opt)
