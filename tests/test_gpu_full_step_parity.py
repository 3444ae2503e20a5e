"""Numeric parity of a FULL training step's graph wiring at BASELINE config[1] (bs16) and at the headline batch (bs32): 512 rays x 64 samples,
2 renders + eikonal.

The reference's model/graph.py cannot be imported (torchvision), so the step is pinned at the level SURVEY 8c prescribes:
the product's own encoder / estimator outputs (both latent projections, the colour code of the neighbour view, poses, intrinsics,
distance scales, the trigonometric camera outputs incl. those of the mirrored image) are injected into the oracle's restatement of
model/renderer.py + model/loss.py (oracle/reference_ops.py, itself pinned against the imported reference: G1-G12), which then
recomputes what model/graph.py:68-112,220-265 and model/runner.py:294-305 compute from them:

  * all 10 loss values of the step                                  bar 2e-5 relative (achieved: printed)
  * d loss.all / d {z_sdf, z_rgb, z_rgb_NN, SDF / RGB weights, beta}   bar 2e-4 of each tensor's max entry at bs16 (both renders accumulated;
                                                                       measured <= 1e-4), 3e-4 at bs32 (measured <= 2.1e-4)
  * d loss.all / d {pose, pose_NN}                                     bar 1e-3.  Not the product's noise: on this very batch the fp32
        ORACLE is 7.4e-4 (pose) / 2.4e-4 (pose_NN) of the max entry away from the same oracle evaluated in float64, the product 6.7e-4 /
        2.0e-4 (i.e. as close to exact arithmetic as the reference's own arithmetic, and 2.7e-4 / 2.0e-4 from the fp32 oracle) --
        tools/oracle_fp64_noise.py, profiles/r03_fullstep_fp64_noise.txt.  The whole error sits in d loss / d R[2,0] of one image: the camera
        centre is -R^T t with t = (0, 0, t_z), so that entry is -t_z times the sum of d loss / d x over all 32,768 sample points of the
        image, and with the x-mirror symmetry of the SDF (abs(x0), implicit.py:139-145) those terms almost cancel.

The implicit networks are perturbed away from their initialisation first: the reference's geometric initialisation zeroes every
latent column of the SDF network (model/implicit.py:86-136), which makes d loss / d z_sdf exactly zero -- a vacuous check.

Both sides draw the stratified jitter / eikonal samples from the same CPU generator state in the reference's call order (main render,
then the neighbour render).  Two quantities are discrete functions of a rendered value and are therefore taken from the product and
checked separately, instead of being recomputed from the oracle's own (1e-5-different) masks:
  * the ray set of the normal losses, `mask_gt & (mask_recon > 0.5)` (graph.py:230-232): the two sides' sets may differ only on rays
    whose rendered mask lies within 1e-4 of 0.5 (asserted);
  * inside loss.normal_loss the 80 % smallest angular errors are kept: a near-tie at the cut may be resolved differently, which moves
    the loss by O(1/n); the normal losses therefore get a 2e-4 bar (the others 2e-5)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


@pytest.mark.parametrize("B", [16, 32], ids=["bs16", "bs32"])
def test_full_step_losses_and_gradients(B):
    from oracle import reference_ops as R
    from shapeclipper_amd import synthetic
    from shapeclipper_amd.model.graph import Graph
    from shapeclipper_amd.utils import options, util
    from shapeclipper_amd.utils.util import EasyDict as edict

    opt = options.set(options.parse_arguments(["--yaml=options/pix3d/config.yaml", "--name=pytest_fullstep", "--output_root=/tmp/sc_pytest",
                                               "--batch_size=%d" % B, "--tb!", "--arch.enc_pretrained!"]
                                              + os.environ.get("SC_FULLSTEP_OPTS", "").split()), verbose=False)
    opt.device = 0
    torch.manual_seed(0)
    g = Graph(opt).cuda().train()
    with torch.no_grad():       # off the geometric initialisation (its latent columns are zero), as smoke() and G12 do
        for p in list(g.sdf_network.parameters()) + list(g.rgb_network.parameters()):
            p.add_(0.02 * torch.randn_like(p))
    batch = util.move_to_device(synthetic.make_batch(opt, B, seed=3, training=True), "cuda:0")
    opt.H, opt.W = opt.image_size

    # ---- product: one Graph.forward(training=True) + loss.all.backward() -------------------------------------------------
    torch.manual_seed(21); np.random.seed(21)
    state = torch.get_rng_state()
    var, loss = g(opt, edict(batch), training=True, get_loss=True)
    weights = {k: float(opt.loss_weight[k]) for k in loss}
    total = sum(weights[k] * loss[k].mean() for k in loss)
    keep = dict(z_sdf=var.proj_latent_sdf, z_rgb=var.proj_latent_rgb, z_rgb_NN=var.proj_latent_rgb_NN, pose=var.pose, pose_NN=var.pose_NN_0)
    for t in keep.values():
        t.retain_grad()
    total.backward()
    torch.cuda.synchronize()
    got_loss = {k: float(v.mean()) for k, v in loss.items()}
    assert set(got_loss) == {"render", "mask", "normal", "eikonal", "cam_margin", "cam_uniform", "cam_sym", "nearest_img", "nearest_mask",
                             "nearest_normal"}

    # ---- oracle on the host: same injected encoder outputs, same random draws ------------------------------------------------
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    c = lambda t: t.detach().float().cpu().clone()
    cfg = R.Cfg(H=opt.H, W=opt.W, cam_dist=float(opt.camera.dist), cam_focal=float(opt.camera.focal), normal_pow=float(opt.reg.normal_pow),
                normal_l1=float(opt.reg.normal_l1), mask_mse=float(opt.reg.mask_mse), emd_p=int(opt.reg.emd_p), bgcolor=float(opt.data.bgcolor))
    Ws = {k: c(v).requires_grad_(True) for k, v in g.sdf_network.state_dict().items()}
    Wr = {k: c(v).requires_grad_(True) for k, v in g.rgb_network.state_dict().items()}
    beta = c(g.renderer.density.beta).reshape(()).requires_grad_(True)
    leaves = {k: c(v).requires_grad_(True) for k, v in keep.items()}
    nn_in = var.input_NN_0
    Rr = opt.render.rand_sample
    tol = float(opt.reg.normal_tol)
    ref_loss = {}

    def render_losses(pose, intr, sd, z_rgb, ray_idx, rgb_t, mask_t, normal_t, mask_prod, names, with_eik):
        t_rand, eik_idx, eik_pts = R.draw_render_randoms(B * Rr, 64, True)
        o = R.render(cfg, Ws, Wr, beta, pose, c(intr), c(sd), leaves["z_sdf"], z_rgb, c(ray_idx).long(), True, t_rand, eik_idx, eik_pts)
        # the discrete ray set of the normal loss comes from the product's mask; the two sides may only differ next to 0.5
        valid_prod = (c(mask_t) > 0.5) & (c(mask_prod) > 0.5)
        valid_ref = (c(mask_t) > 0.5) & (o["mask"].detach() > 0.5)
        differ = valid_prod != valid_ref
        assert not differ.any() or float((o["mask"].detach()[differ] - 0.5).abs().max()) < 1e-4
        target = R.transform_normal(c(normal_t), pose)
        out = {names[0]: R.mse_loss(o["rgb"], c(rgb_t)), names[1]: R.mask_loss(cfg, o["mask"], c(mask_t)),
               names[2]: R.normal_loss(cfg, o["normal"], target, valid_prod, tolerance=tol)}
        if with_eik:
            out["eikonal"] = R.mse_loss(o["grad_eikonal"].view(B, -1), 1)
        part = sum(weights[k] * v for k, v in out.items())
        part.backward()                       # one render at a time: the oracle keeps ~0.6 GB of autograd state per image
        ref_loss.update({k: float(v) for k, v in out.items()})
        err_out = {k: float((o[k].detach() - c(p)).abs().max()) for k, p in (("rgb", None), ("mask", mask_prod)) if p is not None}
        return err_out

    torch.set_rng_state(state)
    e1 = render_losses(leaves["pose"], var.intr, var.scale_dist, leaves["z_rgb"], var.ray_idx, var.rgb_input, var.mask_input, var.normal_input,
                       var.mask_recon, ("render", "mask", "normal"), True)
    e2 = render_losses(leaves["pose_NN"], var.intr_NN_0, var.scale_dist_NN_0, leaves["z_rgb_NN"], nn_in.ray_idx, nn_in.rgb_input, nn_in.mask_input,
                       nn_in.normal_input, var.mask_recon_NN_0, ("nearest_img", "nearest_mask", "nearest_normal"), False)
    print("rendered mask, product vs oracle: max |diff| %.2e (main) %.2e (neighbour view)" % (e1["mask"], e2["mask"]))
    r = opt.data[opt.data.dataset]
    ta, te, tt = c(var.trig_azim), c(var.trig_elev), c(var.trig_theta)
    ref_loss["cam_margin"] = float(R.cam_margin(te, r.elev_range) + R.cam_margin(tt, r.theta_range))
    ref_loss["cam_uniform"] = float(R.cam_uniform_loss(cfg, ta))
    ref_loss["cam_sym"] = float(R.cam_sym_terms(ta, te, tt, tuple(c(t) for t in var._estim_flip[:3])))

    # ---- the 10 loss values and loss.all ---------------------------------------------------------------------------------------
    worst = {}
    for k in sorted(got_loss):
        worst[k] = abs(got_loss[k] - ref_loss[k]) / max(abs(ref_loss[k]), 1e-12)
    ref_total = sum(weights[k] * ref_loss[k] for k in ref_loss)
    worst["all"] = abs(float(total) - ref_total) / abs(ref_total)
    print("bs%d step, loss values" % B, " product vs oracle (relative):", {k: "%.1e" % v for k, v in worst.items()})
    print("   values:", {k: "%.6f" % v for k, v in ref_loss.items()})
    for k, v in worst.items():
        assert v < (2e-4 if "normal" in k else 2e-5), (k, v, got_loss.get(k), ref_loss.get(k))

    # ---- gradients of loss.all ---------------------------------------------------------------------------------------------
    gerr = {}
    for k, leaf in leaves.items():
        ref = leaf.grad
        assert float(ref.abs().max()) > 0, k
        gerr[k] = float((keep[k].grad.cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
    if os.environ.get("SC_FULLSTEP_DUMP"):          # debugging aid: everything an offline (e.g. float64) oracle run needs
        torch.save({"got": {k: keep[k].grad.cpu() for k in leaves}, "ref": {k: v.grad for k, v in leaves.items()},
                    "leaves": {k: v.detach() for k, v in leaves.items()}, "Ws": {k: v.detach() for k, v in Ws.items()},
                    "Wr": {k: v.detach() for k, v in Wr.items()}, "beta": beta.detach(), "state": state, "weights": weights,
                    "main": [c(t) for t in (var.intr, var.scale_dist, var.ray_idx, var.rgb_input, var.mask_input, var.normal_input, var.mask_recon)],
                    "nn": [c(t) for t in (var.intr_NN_0, var.scale_dist_NN_0, nn_in.ray_idx, nn_in.rgb_input, nn_in.mask_input, nn_in.normal_input,
                                          var.mask_recon_NN_0)]}, os.environ["SC_FULLSTEP_DUMP"])
    print("   max |d loss.all / d .|:", {k: "%.2e" % float(v.grad.abs().max()) for k, v in leaves.items()})
    for net, W in ((g.sdf_network, Ws), (g.rgb_network, Wr)):
        for k, p in net.named_parameters():
            ref = W[k].grad
            gerr["%s.%s" % ("sdf" if net is g.sdf_network else "rgb", k)] = float((p.grad.cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
    gerr["beta"] = float((g.renderer.density.beta.grad.cpu().reshape(()) - beta.grad).abs() / beta.grad.abs().clamp_min(1e-12))
    print("bs%d step," % B, "gradients of loss.all product vs oracle (max abs / max |ref|):", {k: "%.1e" % v for k, v in gerr.items()})
    # bs32: twice as many terms per weight gradient, in a different order on the two sides -- the fp32 summation noise grows with the batch
    # (measured on rgb.lin3.weight: 2.1e-4 at bs32 against <= 1e-4 at bs16); the bar follows as sqrt(2) x, rounded up
    bar = 2e-4 if B <= 16 else 3e-4
    for k, v in gerr.items():
        assert v < (1e-3 if k.startswith("pose") else bar), (k, v)
