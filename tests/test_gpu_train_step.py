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
    import copy, pickle
    clone = copy.deepcopy(runner.graph)            # no autograd history / stream objects left hanging on the modules
    assert set(clone.state_dict().keys()) == set(runner.graph.state_dict().keys())
    pickle.dumps(runner.graph.state_dict())


@pytest.mark.parametrize("guarded", [True, False], ids=["guarded_step", "host_wait"])
def test_non_finite_loss_never_reaches_the_weights(guarded):
    """The reference asserts on NaN / Inf losses in summarize_loss, before backward (runner.py:296-302).  Here the assertion names the loss
    and neither the weights nor the Adam state (moments, step counters) of the poisoned step change:
      * hip.guarded_step (default): the flag is the fused optimizer's `found_inf`, the update is skipped on the device; the host raises
        one step later (before the NEXT update is applied) or at the next flush (`check_finite()`: end of epoch, checkpoint);
      * --hip.guarded_step!: the host waits for the flag between the queued backward pass and optim.step() and raises in the same step.
    The next clean step trains normally."""
    from shapeclipper_amd import synthetic
    from shapeclipper_amd.model.runner import Runner
    from shapeclipper_amd.utils import options, util
    from shapeclipper_amd.utils.util import EasyDict as edict
    opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_nan", "--output_root=/tmp/sc_pytest",
                                               "--batch_size=2", "--tb!", "--arch.enc_pretrained!"]
                                              + ([] if guarded else ["--hip.guarded_step!"])), verbose=False)
    opt.device, opt.world_size, opt.port = 0, 1, 0
    opt.freq.scalar, opt.freq.ckpt_latest = 0, 10 ** 9
    torch.manual_seed(0)
    runner = Runner(opt)
    runner.build_networks(opt)
    runner.setup_optimizer(opt)
    assert runner._guarded_step == guarded
    runner.graph.train()
    runner.it, runner.ep, runner.best_val = 1, 0, 0.0
    runner.timer = edict(start=time.time(), it_mean=None)
    batch = util.move_to_device(synthetic.make_batch(opt, 2, seed=0), "cuda:0")
    opt.H, opt.W = opt.image_size
    runner.train_iteration(opt, edict(batch), None)                    # one clean step: the Adam state exists
    g = runner.graph.module
    state = lambda: ({n: p.detach().clone() for n, p in g.named_parameters()},
                     [v["exp_avg"].clone() for v in runner.optim_full.state.values() if "exp_avg" in v],
                     [float(v["step"]) for v in runner.optim_full.state.values() if "step" in v])
    same = lambda a, b: (all(torch.equal(a[0][n], b[0][n]) for n in b[0]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
                         and a[2] == b[2])
    poisoned = edict(batch)
    poisoned.rgb_input_map = batch["rgb_input_map"].clone()
    poisoned.rgb_input_map[0, 0, 5, 5] = float("nan")                 # through the encoders into every loss
    before = state()
    if guarded:
        runner.train_iteration(opt, poisoned, None)                    # returns: the device skipped the update, the host does not know yet
        assert same(state(), before), "a non-finite step changed the weights or the Adam state"
        with pytest.raises(AssertionError, match="is NaN|is Inf"):
            runner.train_iteration(opt, edict(batch), None)            # raised before this (clean) step's update
        assert same(state(), before), "the step behind a non-finite one was applied before the assertion"
        runner.train_iteration(opt, poisoned, None)
        with pytest.raises(AssertionError, match="is NaN|is Inf"):
            runner.check_finite()                                       # the flush of train_epoch / the checkpoint path
    else:
        with pytest.raises(AssertionError, match="is NaN|is Inf"):
            runner.train_iteration(opt, poisoned, None)
    assert same(state(), before), "weights / Adam state were updated by a non-finite step"
    loss = runner.train_iteration(opt, edict(batch), None)             # BatchNorm running statistics are poisoned like in the reference;
    assert set(loss.keys()) >= {"render", "all"}                       # the step itself runs (training-mode BN uses batch statistics)
    runner.check_finite()
    after = state()
    assert after[2] == [s + 1 for s in before[2]] and not same(after, before)


def test_batched_encoder_passes_equal_sequential_passes():
    """hip.batched_encoders (one grouped encoder pass + one grouped estimator pass per step) gives the losses,
    gradients and BatchNorm running statistics of the reference's five separate passes."""
    import copy
    import numpy as np
    from shapeclipper_amd import synthetic
    from shapeclipper_amd.model.graph import Graph
    from shapeclipper_amd.utils import util
    from shapeclipper_amd.utils.util import EasyDict as edict
    from shapeclipper_amd.utils import options
    o = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_batched", "--output_root=/tmp/sc_pytest",
                                             "--batch_size=4", "--tb!", "--arch.enc_pretrained!"]), verbose=False)
    o.device = 0
    torch.manual_seed(0)
    g0 = Graph(o).cuda().train()
    batch = util.move_to_device(synthetic.make_batch(o, 4, seed=5, training=True), "cuda:0")
    res = []
    for batched in (False, True):
        o.hip.batched_encoders = batched
        g = copy.deepcopy(g0)
        torch.manual_seed(11); np.random.seed(11)
        o.H, o.W = o.image_size
        var, loss = g(o, edict(batch), training=True, get_loss=True)
        total = sum(float(o.loss_weight[k]) * loss[k].mean() for k in loss if o.loss_weight[k] is not None)
        total.backward()
        res.append((float(total), {k: float(v.mean()) for k, v in loss.items()},
                    g.estimator.feature_extractor.conv1.weight.grad.clone(), g.encoder.layer2[0].conv1.weight.grad.clone(),
                    g.sdf_network.lin3.weight.grad.clone(), g.encoder.bn1.running_var.clone(),
                    g.estimator.feature_extractor.layer4[1].bn2.running_mean.clone(),
                    int(g.estimator.feature_extractor.bn1.num_batches_tracked), int(g.encoder.bn1.num_batches_tracked)))
    o.hip.batched_encoders = True
    seq, bat = res
    assert abs(bat[0] - seq[0]) < 2e-3 * abs(seq[0])
    for k in seq[1]:
        assert abs(bat[1][k] - seq[1][k]) < 2e-3 * abs(seq[1][k]) + 1e-6, k
    rel = lambda u, v: float((u - v).norm() / v.norm().clamp_min(1e-12))
    for i in (2, 3, 4):
        assert rel(bat[i], seq[i]) < 2e-2, i
    assert rel(bat[5], seq[5]) < 1e-4 and rel(bat[6], seq[6]) < 1e-3
    assert bat[7] == seq[7] == 3 and bat[8] == seq[8] == 2


def test_flat_reducer_path_on_rccl_single_rank():
    """The N > 1 code path (flat gradient buffer, one RCCL all-reduce, buffer broadcast, fused Adam on the flat views) run
    on one rank with the collectives forced on: parameters after two iterations equal the reducer-less run."""
    import copy
    import numpy as np
    import torch.distributed as dist
    from shapeclipper_amd import synthetic
    from shapeclipper_amd.model.runner import Runner
    from shapeclipper_amd.parallel import FlatGradAllReduce
    from shapeclipper_amd.utils import options, util
    from shapeclipper_amd.utils.util import EasyDict as edict
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        results = []
        for use_reducer in (False, True):
            opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_reducer", "--output_root=/tmp/sc_pytest",
                                                       "--batch_size=2", "--tb!", "--arch.enc_pretrained!"]), verbose=False)
            opt.device, opt.world_size, opt.port = 0, 1, 0
            opt.freq.scalar, opt.freq.ckpt_latest = 0, 10 ** 9
            torch.manual_seed(0); np.random.seed(0)
            runner = Runner(opt)
            runner.build_networks(opt)
            if use_reducer:
                runner.reducer = FlatGradAllReduce(runner.graph.module, 1, always_communicate=True)
            runner.setup_optimizer(opt)
            runner.graph.train()
            runner.it, runner.ep, runner.best_val = 1, 0, 0.0
            runner.timer = edict(start=time.time(), it_mean=None)
            batch = util.move_to_device(synthetic.make_batch(opt, 2, seed=0), "cuda:0")
            for _ in range(2):
                opt.H, opt.W = opt.image_size
                runner.train_iteration(opt, edict(batch), None)
            g = runner.graph.module
            results.append({n: p.detach().clone() for n, p in g.named_parameters()})
            if use_reducer:
                assert all(p.grad.data_ptr() >= runner.reducer.flat.data_ptr() for p in g.parameters() if p.grad is not None)
                # overlapped schedule on the GPU: the early segment from inside backward (side stream) + the late one after it
                assert runner.reducer.overlap and runner.reducer.collectives == 2 * 2, runner.reducer.collectives
        # Every sum of the step has a fixed order (tests/test_gpu_determinism.py) and the exchange of one rank is the identity, so the
        # two paths agree bit for bit -- since the early segment's side stream waits for every stream of the step (before that fix the
        # pack could read unfinished gradients and only a 2-lr bound held here).
        for n in results[0]:
            a, b = results[0][n], results[1][n]
            assert torch.equal(a, b), "%s: max %g" % (n, float((a - b).abs().max()))
        assert torch.allclose(results[0]["renderer.density.beta"], results[1]["renderer.density.beta"], atol=1e-6)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("flag", ["--hip.two_streams!", "--hip.batched_encoders!", "--hip.fused_loss!", "--hip.fused_adam!", "--hip.device_rng",
                                  "--hip.fused_backward!", "--hip.fused_rgb_wgrad!", "--hip.rgb_stash!", "--hip.upload_stream!", "--hip.device_choice!", "--hip.conv3x3!", "--hip.conv3x3_split!", "--hip.conv_stem!", "--hip.conv1x1!", "--hip.conv3x3s2!",
                                  "--hip.overlap_allreduce", "--hip.sdf_stream!", "--hip.rgb_split!", "--hip.rgb_bwd_split!", "--hip.value_split!", "--hip.fused_block!", "--hip.fused_bottleneck!", "--hip.rocblas!", "--arch.impl_sdf.weight_norm", "--arch.impl_rgb.weight_norm"])
def test_every_hip_option_has_a_working_alternate_path(flag):
    """Each fast path of this build can be switched off (README): the step still runs and gives the same loss
    (device_rng draws different jitter, so only finiteness is compared there).  The two weight_norm switches of the reference's
    config ride along: the reparameterisation is applied after the initialisation, so the first step's loss is the same."""
    from shapeclipper_amd import synthetic
    from shapeclipper_amd.model.runner import Runner
    from shapeclipper_amd.utils import options, util
    from shapeclipper_amd.utils.util import EasyDict as edict
    losses = []
    for extra in ([], [flag]):
        opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_opts", "--output_root=/tmp/sc_pytest",
                                                   "--batch_size=2", "--tb!", "--arch.enc_pretrained!"] + extra), verbose=False)
        opt.device, opt.world_size, opt.port = 0, 1, 0
        opt.freq.scalar, opt.freq.ckpt_latest = 0, 10 ** 9
        torch.manual_seed(0)
        import numpy as np
        np.random.seed(0)
        runner = Runner(opt)
        runner.build_networks(opt)
        runner.setup_optimizer(opt)
        runner.graph.train()
        runner.it, runner.ep, runner.best_val = 1, 0, 0.0
        runner.timer = edict(start=time.time(), it_mean=None)
        batch = util.move_to_device(synthetic.make_batch(opt, 2, seed=0), "cuda:0")
        opt.H, opt.W = opt.image_size
        losses.append(float(runner.train_iteration(opt, edict(batch), None).all.detach()))
    assert all(torch.isfinite(torch.tensor(losses)))
    if flag != "--hip.device_rng":
        assert abs(losses[1] - losses[0]) < 2e-3 * abs(losses[0]), losses


def test_fused_loss_carries_the_normal_target_gradient_into_the_estimator():
    """The normal-loss target is transform_normal(input normal, predicted pose) (reference model/graph.py:85,260), so the
    reference's normal_loss trains the view estimator through its TARGET as well.  The fused HIP loss must give the same
    parameter gradients as the unfused torch restatement of model/loss.py for the normal terms alone."""
    import copy
    import numpy as np
    from shapeclipper_amd import synthetic
    from shapeclipper_amd.model.graph import Graph
    from shapeclipper_amd.utils import options, util
    from shapeclipper_amd.utils.util import EasyDict as edict
    o = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_fusedloss", "--output_root=/tmp/sc_pytest",
                                             "--batch_size=4", "--tb!", "--arch.enc_pretrained!"]), verbose=False)
    o.device = 0
    torch.manual_seed(0)
    g0 = Graph(o).cuda().train()
    batch = util.move_to_device(synthetic.make_batch(o, 4, seed=5, training=True), "cuda:0")
    grads = []
    for fused in (False, True):
        o.hip.fused_loss = fused
        g = copy.deepcopy(g0)
        torch.manual_seed(11); np.random.seed(11)
        o.H, o.W = o.image_size
        var, loss = g(o, edict(batch), training=True, get_loss=True)
        (loss.normal.mean() + loss.nearest_normal.mean()).backward()
        grads.append({n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None})
    o.hip.fused_loss = True
    ref, got = grads
    est = [n for n in ref if n.startswith("estimator.")]
    assert est and all(n in got for n in est)
    num = sum(float((got[n] - ref[n]).pow(2).sum()) for n in est)
    den = sum(float(ref[n].pow(2).sum()) for n in est)
    assert den > 0
    print("estimator gradient of the normal losses, fused vs unfused: rel L2 err %.3e" % (num / den) ** 0.5)
    assert (num / den) ** 0.5 < 2e-3
    for n in ("sdf_network.lin3.weight", "renderer.density.beta"):
        assert float((got[n] - ref[n]).norm() / ref[n].norm().clamp_min(1e-20)) < 2e-3, n
