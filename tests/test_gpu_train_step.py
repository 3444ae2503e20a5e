"""Full training iterations through Runner.train_iteration on the GPU (synthetic batch): every loss key the
reference produces is present and finite, parameters of every trainable child move, consecutive steps work."""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


def test_two_training_iterations():
    from shapeclipper_amd import synthetic
    from shapeclipper_amd.model.runner import Runner
    from shapeclipper_amd.utils import options, util
    from shapeclipper_amd.utils.util import EasyDict as edict
    opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_step", "--output_root=/tmp/sc_pytest",
                                               "--batch_size=2", "--tb!", "--arch.enc_pretrained!"]), verbose=False)
    opt.device, opt.world_size, opt.port = 0, 1, 0
    opt.freq.scalar, opt.freq.ckpt_latest = 0, 10 ** 9
    torch.manual_seed(0)
    runner = Runner(opt)
    runner.build_networks(opt)
    runner.setup_optimizer(opt)
    runner.graph.train()
    runner.it, runner.ep, runner.best_val = 1, 0, 0.0
    runner.timer = edict(start=time.time(), it_mean=None)
    batch = util.move_to_device(synthetic.make_batch(opt, 2, seed=0), "cuda:0")
    g = runner.graph.module
    before = {n: p.detach().clone() for n, p in g.named_parameters()}
    keys = None
    for _ in range(2):
        opt.H, opt.W = opt.image_size
        loss = runner.train_iteration(opt, edict(batch), None)
        keys = set(loss.keys())
        assert all(torch.isfinite(torch.as_tensor(float(v))) for v in loss.values())
    assert keys == {"render", "mask", "normal", "eikonal", "cam_margin", "cam_uniform", "cam_sym", "nearest_img", "nearest_mask",
                    "nearest_normal", "all"}
    moved = {n.split(".")[0] for n, p in g.named_parameters() if not torch.equal(p.detach(), before[n])}
    assert {"estimator", "sdf_network", "rgb_network", "renderer", "encoder", "latent_proj_shape", "latent_proj_rgb"} <= moved
    assert not torch.equal(g.renderer.density.beta.detach(), before["renderer.density.beta"])     # beta is trained by the HIP backward
