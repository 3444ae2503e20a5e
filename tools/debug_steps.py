import os, sys, time, traceback
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import synthetic
from shapeclipper_amd.model.runner import Runner
from shapeclipper_amd.utils import options, util
from shapeclipper_amd.utils.util import EasyDict as edict
opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml","--name=b","--output_root=/tmp/sc_b","--batch_size=2","--tb!","--arch.enc_pretrained!"]), verbose=False)
opt.device, opt.world_size, opt.port = 0, 1, 0
opt.freq.scalar=0; opt.freq.ckpt_latest=10**9
runner=Runner(opt); runner.build_networks(opt); runner.setup_optimizer(opt); runner.graph.train()
runner.it, runner.ep, runner.best_val = 1,0,0.0
runner.timer=edict(start=time.time(), it_mean=None)
batch=util.move_to_device(synthetic.make_batch(opt,2,seed=0),"cuda:0")
for i in range(3):
    opt.H,opt.W=opt.image_size
    try:
        l=runner.train_iteration(opt, edict(batch), None); print(i, float(l.all), {k: float(v) for k,v in l.items() if k!='all'})
    except Exception as e:
        traceback.print_exc(limit=30); break
