"""Run-to-run reproducibility of the hand-written path (the reference asks for it: utils/options.py:14 sets cudnn.deterministic).

Every sum whose order used to depend on timing -- float atomicAdds for the loss values, the per-image bias gradients of both MLPs,
d/d beta, the weight-gradient combines, the partial-image reductions -- is now evaluated in a fixed order (per-workgroup / per-wave
partial images + an ordered sum; csrc/loss.hip, sdf_bwdw.hip, rgb_bwd.hip, wgrad.hip).  Tests:
  * the same training render (forward + backward) twice: every output and every gradient bit-identical;
  * the same TRAINING STEPS twice from the same state: every parameter bit-identical after two optimizer steps."""
import copy
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


def _render_once(r, opt, args, ray_idx, seed):
    torch.manual_seed(seed)
    leaves = [a.clone().requires_grad_(True) for a in args]
    for p in r.parameters():
        p.grad = None
    out = r(opt, *leaves, ray_idx=ray_idx, training=True)
    L = out[0].square().sum() + out[1].sum() + (out[4] * out[2]).sum() + out[3].sum() + ((out[5] - 1) ** 2).mean()
    L.backward()
    torch.cuda.synchronize()
    return ([o.detach().clone() for o in out if o is not None], [l.grad.clone() for l in leaves],
            {n: p.grad.clone() for n, p in r.named_parameters()})


def test_training_render_is_bit_reproducible():
    from shapeclipper_amd.model.implicit import RGBNetwork, SDFNetwork
    from shapeclipper_amd.model.renderer import Renderer
    from shapeclipper_amd.utils import camera, options
    dev = torch.device("cuda:0")
    opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_det", "--output_root=/tmp/sc_pytest"]), verbose=False)
    torch.manual_seed(0)
    sdf, rgb = SDFNetwork(opt), RGBNetwork(opt)
    with torch.no_grad():
        for p in list(sdf.parameters()) + list(rgb.parameters()):
            p.add_(0.03 * torch.randn_like(p))
    r = Renderer(opt, sdf, rgb).to(dev)
    B, R = 8, 512
    az = (torch.rand(B) * 2 - 1) * 3.14159
    trig = lambda t: torch.stack([torch.cos(t), torch.sin(t)], 1)
    Ry = camera.azim_to_rotation_matrix(trig(az), "trig"); Rx = camera.elev_to_rotation_matrix(trig(torch.zeros(B)), "trig")
    Pm = torch.tensor([[-1., 0, 0], [0, 0, -1], [0, -1, 0]])[None].expand(B, 3, 3)
    pose = camera.pose.compose([camera.pose(R=Rx @ Ry @ Pm), camera.pose(t=torch.tensor([[0., 0, 5.]]).expand(B, 3))]).to(dev)
    intr = camera.get_intr(opt, torch.ones(B)).to(dev)
    args = [pose, intr, torch.ones(B, device=dev), torch.randn(B, 64, device=dev), torch.randn(B, 64, device=dev)]
    ray_idx = torch.stack([torch.randperm(224 * 224)[:R] for _ in range(B)]).to(dev)
    a = _render_once(r, opt, args, ray_idx, seed=3)
    for rep in range(3):
        b = _render_once(r, opt, args, ray_idx, seed=3)
        for x, y in zip(a[0], b[0]):
            assert torch.equal(x, y), "render output differs between identical runs"
        for k, (x, y) in enumerate(zip(a[1], b[1])):
            assert torch.equal(x, y), "input gradient %d differs between identical runs" % k
        for n in a[2]:
            assert torch.equal(a[2][n], b[2][n]), "gradient of %s differs between identical runs" % n


def test_two_training_steps_are_bit_reproducible():
    from shapeclipper_amd import synthetic
    from shapeclipper_amd.model.runner import Runner
    from shapeclipper_amd.utils import options, util
    from shapeclipper_amd.utils.util import EasyDict as edict
    states = []
    for run in range(2):
        # (the six stride-2 3x3 layers' gradients are still MIOpen kernels: --hip.deterministic_conv selects its deterministic algorithms)
        opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_det2", "--output_root=/tmp/sc_pytest",
                                                   "--batch_size=4", "--tb!", "--arch.enc_pretrained!", "--hip.deterministic_conv"]), verbose=False)
        opt.device, opt.world_size, opt.port = 0, 1, 0
        opt.freq.scalar, opt.freq.ckpt_latest = 0, 10 ** 9
        torch.manual_seed(0); np.random.seed(0)
        runner = Runner(opt)
        runner.build_networks(opt)
        runner.setup_optimizer(opt)
        runner.graph.train()
        runner.it, runner.ep, runner.best_val = opt.optim.iter_camera + 1, 0, 0.0      # full optimizer: every network trains
        runner.timer = edict(start=time.time(), it_mean=None)
        batch = util.move_to_device(synthetic.make_batch(opt, 4, seed=1), "cuda:0")
        losses = []
        for _ in range(2):
            opt.H, opt.W = opt.image_size
            loss = runner.train_iteration(opt, edict(batch), None)
            losses.append(loss.all.detach().clone())
        torch.cuda.synchronize()
        states.append(({k: v.detach().clone() for k, v in runner.graph.state_dict().items()}, losses))
    (s0, l0), (s1, l1) = states
    for a, b in zip(l0, l1):
        assert torch.equal(a, b), "loss.all differs between identical runs: %r vs %r" % (float(a), float(b))
    bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
    assert not bad, "parameters / buffers differ between two identical 2-step runs: %s" % bad[:8]
