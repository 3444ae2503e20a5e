"""torch.autograd.Function wrappers around the HIP kernels (forward AND hand-written backward)."""
from __future__ import annotations

import torch

from . import ops


class SdfFunction(torch.autograd.Function):
    """points [N,3], w_pack, cbias [B,5,64] -> sdf [N], grad [N,3] (d sdf/d point), feat (TBL64).

    Mirrors SDFNetwork.get_conditional_output (reference model/implicit.py:163-189): `grad` is the
    create_graph=True gradient, and it is itself differentiable here (double backward is the
    hand-written sdf_bwd.hip / wgrad.hip path)."""

    @staticmethod
    def forward(ctx, points, w_pack, cbias, n_per_image, symmetric, want_grad, want_feat, fused_backward=True):
        ctx.set_materialize_grads(False)       # the kernels take null for an output nobody differentiated: no zero fills
        points = points.contiguous()
        need = any(ctx.needs_input_grad[:3])
        res = ops.sdf_forward(points, w_pack, cbias, n_per_image, symmetric=symmetric, want_grad=want_grad,
                              want_feat=want_feat, stash=need)
        sdf, grad, feat = res[0], res[1], res[2]
        ctx.meta = (n_per_image, cbias.shape[0], symmetric, want_grad, want_feat)
        ctx.fused_backward = fused_backward
        if need:
            ctx.save_for_backward(points, w_pack, res[3], res[4] if want_grad else None)
        empty = points.new_empty(0)
        outs = (sdf, grad if want_grad else empty, feat if want_feat else empty)
        nd = [o for o, w in zip(outs[1:], (want_grad, want_feat)) if not w]
        if nd:
            ctx.mark_non_differentiable(*nd)
        return outs

    @staticmethod
    def backward(ctx, g_sdf, g_grad, g_feat):
        n_per_image, n_images, symmetric, want_grad, want_feat = ctx.meta
        points, w_pack, stash_a, stash_p = ctx.saved_tensors
        g_sdf = g_sdf.contiguous() if g_sdf is not None else None
        g_grad = g_grad.contiguous() if (want_grad and g_grad is not None) else None
        g_feat = g_feat.contiguous() if (want_feat and g_feat is not None) else None
        gp, gw, gc = ops.sdf_backward(points, w_pack, n_per_image, n_images, symmetric, stash_a, stash_p,
                                      g_sdf, g_grad, g_feat, want_points_grad=ctx.needs_input_grad[0],
                                      fused=ctx.fused_backward)
        return gp, gw, gc, None, None, None, None, None


class RgbCompositeFunction(torch.autograd.Function):
    """Per-point SDF results -> per-ray render outputs (reference model/renderer.py:110-152,187-209).

    inputs : points [P,3], z_vals [n_rays,64], depth_fac [n_rays], sdf [P], grad [P,3], feat TBL64,
             v_pack, dbias [B,3,64], beta (raw parameter, shape [1] or [])
    outputs: rgb [n_rays,3], mask [n_rays], mask_hard [n_rays], depth [n_rays], normal [n_rays,3]
             (+ weights, alpha [n_rays,64], rgb_flat [P,3] when keep_samples; non-differentiable)"""

    @staticmethod
    def forward(ctx, points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta, rays_per_image, symmetric,
                beta_min, bgcolor, normal_pow, keep_samples):
        ctx.set_materialize_grads(False)
        need = any(ctx.needs_input_grad[:9])
        beta1 = beta.reshape(1).contiguous()
        stash = need and ops.RGB_STASH and ops.FUSED_RGB_WGRAD and dbias.shape[0] <= 256      # what the fused backward takes
        out = ops.rgb_composite_forward(points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta1,
                                        rays_per_image, symmetric, beta_min, bgcolor, normal_pow,
                                        keep_samples=keep_samples, keep_rgb_flat=need, keep_rr=stash)
        ctx.meta = (rays_per_image, symmetric, beta_min, bgcolor, normal_pow, beta.shape)
        if need:
            ctx.save_for_backward(points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta1, out["rgb_flat"], *((out["rr"],) if stash else ()))
        ctx.mark_non_differentiable(out["mask_hard"])
        extra = ()
        if keep_samples:
            extra = (out["weights"], out["alpha"], out["rgb_flat"])
            ctx.mark_non_differentiable(*extra)
        return (out["rgb"], out["mask"], out["mask_hard"], out["depth"], out["normal"]) + extra

    @staticmethod
    def backward(ctx, G_rgb, G_mask, G_mask_hard, G_depth, G_normal, *unused):
        rays_per_image, symmetric, beta_min, bgcolor, normal_pow, beta_shape = ctx.meta
        points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta1, rgb_flat, *rr = ctx.saved_tensors
        c = lambda t: t.contiguous() if t is not None else None
        g = ops.rgb_composite_backward(points, z_vals, depth_fac, sdf, grad, feat, v_pack, dbias, beta1, rgb_flat,
                                       rays_per_image, symmetric, beta_min, bgcolor, normal_pow,
                                       c(G_rgb), c(G_mask), c(G_depth), c(G_normal), rr=rr[0] if rr else None)
        return (g["points"], g["z_vals"], g["depth_fac"], g["sdf"], g["grad"], g["feat"], g["v_pack"], g["dbias"],
                g["beta"].reshape(beta_shape), None, None, None, None, None, None)


class FusedRenderLoss(torch.autograd.Function):
    """render MSE, mask (IoU + mask_mse*MSE), robust masked normal loss and eikonal MSE of one render in a
    single HIP launch (csrc/loss.hip).  Returns four scalars (render, mask, normal, eikonal): separate outputs rather
    than one [4] tensor, whose indexing would cost a select node (fill + copy + add in the backward) per loss term."""

    @staticmethod
    def forward(ctx, rgb, rgb_t, mask, mask_t, normal, normal_t, eik, normal_l1, mask_mse, keep_frac):
        # The normal target is transform_normal(input normal, predicted pose): the reference's normal_loss
        # back-propagates through it into the view estimator (model/loss.py:52-67, model/graph.py:85,260).
        ctx.set_materialize_grads(False)
        want_t = ctx.needs_input_grad[5]
        out, grads = ops.loss_fused_forward(rgb, rgb_t, mask, mask_t, normal, normal_t, eik, normal_l1, mask_mse, keep_frac,
                                            want_target_grad=want_t)
        ctx.shapes = (rgb.shape, mask.shape, normal.shape, eik.shape if eik is not None else None)
        ctx.save_for_backward(*[g for g in grads if g is not None])
        ctx.has_eik, ctx.has_t = eik is not None, want_t
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, G_render, G_mask, G_normal, G_eik):
        saved = list(ctx.saved_tensors)
        g_rgb, g_mask, g_normal = saved[0], saved[1], saved[2]
        s_rgb, s_mask, s_normal, s_eik = ctx.shapes
        scale = lambda g, G, shape: (g * G).view(shape) if G is not None else None
        g_eik = scale(saved[3], G_eik, s_eik) if ctx.has_eik else None
        g_t = scale(saved[-1], G_normal, s_normal) if ctx.has_t else None
        return (scale(g_rgb, G_render, s_rgb), None, scale(g_mask, G_mask, s_mask), None, scale(g_normal, G_normal, s_normal),
                g_t, g_eik, None, None, None)


class RaySampleFunction(torch.autograd.Function):
    """cam_loc, ray_dirs [n_rays,3], scale_dist [B], u [n_rays,64] | None -> z_vals [n_rays,64], points [n_rays*64,3]
    (UniformSampler.get_z_vals + point generation, reference model/renderer.py:13-37,84-86) with a hand-written adjoint."""

    @staticmethod
    def forward(ctx, cam_loc, ray_dirs, scale_dist, u, rays_per_image, cam_dist):
        ctx.set_materialize_grads(False)
        cam_loc, ray_dirs, scale_dist = cam_loc.contiguous(), ray_dirs.contiguous(), scale_dist.contiguous()
        z, pts = ops.ray_sample_forward(cam_loc, ray_dirs, scale_dist, u, rays_per_image, cam_dist)
        ctx.save_for_backward(ray_dirs, z)
        ctx.meta = (rays_per_image, scale_dist.shape[0], cam_dist)
        return z, pts

    @staticmethod
    def backward(ctx, g_z, g_points):
        ray_dirs, z = ctx.saved_tensors
        rpi, n_images, cam_dist = ctx.meta
        if g_points is None:
            g_points = torch.zeros(z.numel(), 3, device=z.device)
        g_o, g_d, g_sd = ops.ray_sample_backward(ray_dirs, z, g_points.contiguous(), g_z.contiguous() if g_z is not None else None,
                                                 rpi, n_images, cam_dist)
        return g_o, g_d, g_sd, None, None, None


class RaySampleEikFunction(torch.autograd.Function):
    """RaySampleFunction + the eikonal sample points of a training render (reference model/renderer.py:154-165) in the same two launches:
    -> z_vals [n_rays,64], points [n_rays*64,3], eik_points [B, 2 R, 3] (uniform block | near-surface block).  The near point of a ray is its
    sample eik_idx -- no gather of z, no second evaluation of cam_loc + z * ray_dir, and in backward no scatter into a dense z gradient
    (round 5: ~15 stock launches per render)."""

    @staticmethod
    def forward(ctx, cam_loc, ray_dirs, scale_dist, u, eik_idx, eik_uniform, rays_per_image, cam_dist):
        ctx.set_materialize_grads(False)
        cam_loc, ray_dirs, scale_dist = cam_loc.contiguous(), ray_dirs.contiguous(), scale_dist.contiguous()
        eik_idx, eik_uniform = eik_idx.contiguous(), eik_uniform.contiguous()
        z, pts, eik = ops.ray_sample_forward_eik(cam_loc, ray_dirs, scale_dist, u, eik_idx, eik_uniform, rays_per_image, cam_dist)
        ctx.save_for_backward(ray_dirs, z, eik_idx)
        ctx.meta = (rays_per_image, scale_dist.shape[0], cam_dist)
        return z, pts, eik

    @staticmethod
    def backward(ctx, g_z, g_points, g_eik):
        ray_dirs, z, eik_idx = ctx.saved_tensors
        rpi, n_images, cam_dist = ctx.meta
        g_o, g_d, g_sd = ops.ray_sample_backward_eik(ray_dirs, z, g_points.contiguous() if g_points is not None else None,
                                                     g_z.contiguous() if g_z is not None else None, eik_idx,
                                                     g_eik.contiguous() if g_eik is not None else None, rpi, n_images, cam_dist)
        return g_o, g_d, g_sd, None, None, None, None, None


class BnActFunction(torch.autograd.Function):
    """y = [relu]( batch_norm(x) [+ res] ) in two HIP launches (csrc/bn_act.hip); nn.BatchNorm2d semantics for the
    running statistics (updated in place; num_batches_tracked incremented by the kernel)."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, running_mean, running_var, n_tracked, training, momentum, eps, relu, groups):
        x = ops._aligned(x)
        if res is not None:
            res = ops._aligned(res)
        y, stats = ops.bn_act_forward(x, res, weight, bias, running_mean, running_var, n_tracked, training,
                                      momentum, eps, relu, groups)
        ctx.meta = (training, relu, res is not None, groups)
        ctx.save_for_backward(x, weight, bias, stats, y if (res is not None and relu) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        training, relu, has_res, groups = ctx.meta
        x, weight, bias, stats, y = ctx.saved_tensors
        dx, dres, dg, db = ops.bn_act_backward(ops._aligned(dy), x, y, weight, bias, stats, training, relu,
                                               ctx.needs_input_grad[0], has_res and ctx.needs_input_grad[1], groups)
        return dx, dres, dg, db, None, None, None, None, None, None, None, None


class BottleneckLinearFunction(torch.autograd.Function):
    """out = relu(bn2(relu(bn1(x W1^T)) W2^T) + x): a 1x1 Bottleneck_Linear block (reference view_estimator.py:6-33, graph.py:16-40) as two
    launches forward and three backward (csrc/bottleneck.hip) instead of ~14 stock launches.  bn* = (running_mean, running_var,
    num_batches_tracked, training, momentum, eps) as BnActFunction takes them."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, bn1, bn2, groups):
        x, w1, w2 = ops._aligned(x), ops._aligned(w1), ops._aligned(w2)
        rm1, rv1, nt1, training, mom1, eps1 = bn1
        rm2, rv2, nt2, _, mom2, eps2 = bn2
        a1, y1, st1 = ops.linear_bn_forward(x, w1, g1, b1, None, rm1, rv1, nt1, training, mom1, eps1, True, groups)
        out, y2, st2 = ops.linear_bn_forward(a1, w2, g2, b2, x, rm2, rv2, nt2, training, mom2, eps2, True, groups)
        ctx.meta = (training, groups)
        ctx.save_for_backward(x, w1, g1, w2, g2, y1, a1, st1, y2, out, st2)
        return out

    @staticmethod
    def backward(ctx, g_out):
        training, groups = ctx.meta
        x, w1, g1, w2, g2, y1, a1, st1, y2, out, st2 = ctx.saved_tensors
        gy2, g_res, dw2, dg2, db2 = ops.linear_bn_backward(ops._aligned(g_out), None, None, None, out, y2, st2, g2, a1, True, training, True, groups)
        gy1, _, dw1, dg1, db1 = ops.linear_bn_backward(None, gy2, w2, None, a1, y1, st1, g1, x, False, training, True, groups)
        dx = ops.linear_backward_data(gy1, w1, g_res) if ctx.needs_input_grad[0] else None
        return dx, dw1, dg1, db1, dw2, dg2, db2, None, None, None


class BnReluPoolFunction(torch.autograd.Function):
    """ResNet stem tail: maxpool3x3/2( relu( batch_norm(x) ) ) without materialising the full-resolution BN output."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, n_tracked, training, momentum, eps, groups):
        x = ops._aligned(x)
        y, idx, stats = ops.bn_relu_pool_forward(x, weight, bias, running_mean, running_var, n_tracked, training,
                                                 momentum, eps, groups)
        ctx.training, ctx.groups = training, groups
        ctx.save_for_backward(x, weight, bias, stats, idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, stats, idx = ctx.saved_tensors
        dx, dg, db = ops.bn_relu_pool_backward(ops._aligned(dy), idx, x, weight, bias, stats, ctx.training, ctx.groups)
        return dx, dg, db, None, None, None, None, None, None, None


def _bn_fusable(bn, x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and bn.affine and bn.momentum is not None
            and (bn.training or bn.track_running_stats))


def bn_act(bn, x, residual=None, relu=True, groups=1):
    """nn.BatchNorm2d `bn` applied to x, then `+ residual`, then ReLU.  Device tensors take the fused HIP path (raises
    if the library is missing); host tensors use the stock torch operators (pretrain plumbing on CPU, config[0]).
    groups > 1: x holds `groups` independent sub-batches stacked along dim 0 (statistics per sub-batch, running
    statistics updated once per sub-batch in order -- the result of `groups` consecutive calls)."""
    if not _bn_fusable(bn, x):
        out = bn(x) if groups == 1 else torch.cat([bn(c) for c in x.chunk(groups)], 0)
        if residual is not None:
            out = out + residual
        return torch.relu_(out) if relu else out
    training = bn.training or not bn.track_running_stats
    track = bn.track_running_stats
    return BnActFunction.apply(x, residual, bn.weight, bn.bias, bn.running_mean if track else None,
                               bn.running_var if track else None,
                               bn.num_batches_tracked if (track and training) else None,
                               training, float(bn.momentum), float(bn.eps), relu, groups)


def _bn_state(bn):
    training = bn.training or not bn.track_running_stats
    track = bn.track_running_stats
    return (bn.running_mean if track else None, bn.running_var if track else None,
            bn.num_batches_tracked if (track and training) else None, training, float(bn.momentum), float(bn.eps))


def bottleneck_linear(x, conv1, bn1, conv2, bn2, groups=1):
    """relu(bn2(conv2(relu(bn1(conv1(v))))) + v) for a feature VECTOR x [N, C] (1x1 map), conv* = nn.Conv2d(C, C, 1, bias=False): the
    fused HIP block when the shape is taken (N <= 128, groups <= 4, C % 64 == 0), else None (the caller keeps its operator-by-operator form)."""
    C = x.shape[1]
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and _bn_fusable(bn1, x[..., None, None]) and _bn_fusable(bn2, x[..., None, None])
            and conv1.bias is None and conv2.bias is None and conv1.weight.shape == (C, C, 1, 1) and conv2.weight.shape == (C, C, 1, 1)
            and x.shape[0] % groups == 0 and ops.linear_bn_supported(x.shape[0], C, C, groups)):
        return None
    if (bn1.training or bn2.training) and x.shape[0] // groups < 2:
        return None     # one row per statistics group in training: the operator-by-operator form raises torch's own error, as the reference does
    return BottleneckLinearFunction.apply(x, conv1.weight.view(C, C), bn1.weight, bn1.bias, conv2.weight.view(C, C), bn2.weight, bn2.bias,
                                          _bn_state(bn1), _bn_state(bn2), groups)


def bn_relu_maxpool(bn, x, groups=1):
    """maxpool3x3/2/1(relu(bn(x))) (ResNet stem)."""
    if not _bn_fusable(bn, x):
        out = bn(x) if groups == 1 else torch.cat([bn(c) for c in x.chunk(groups)], 0)
        return torch.nn.functional.max_pool2d(torch.relu_(out), 3, 2, 1)
    training = bn.training or not bn.track_running_stats
    track = bn.track_running_stats
    return BnReluPoolFunction.apply(x, bn.weight, bn.bias, bn.running_mean if track else None,
                                    bn.running_var if track else None,
                                    bn.num_batches_tracked if (track and training) else None,
                                    training, float(bn.momentum), float(bn.eps), groups)


class Conv3x3Function(torch.autograd.Function):
    """F.conv2d(x, w, None, 1, 1) for the 3x3 / stride-1 layers of the ResNet trunks on csrc/conv3x3.hip (fp32 matrix pipe): forward
    and backward-data are the same kernel (the latter with the transposed, flipped filter); the weight gradient is conv3x3_wgrad.hip.
    pack_f / pack_b: the kernel-ready images of w from a Conv3x3PackSet (None: packed here, two extra launches).
    split: forward / backward-data with fp32-accurate products on the bf16 matrix pipe (sc_conv3x3_forward_split)."""

    @staticmethod
    def forward(ctx, x, w, pack_f, pack_b, split):
        x = ops._aligned(x)
        # pack_b is a view of the pack set's buffer, which every forward pass of the network rewrites in place through a raw pointer.
        # That is safe: the rewrite reproduces the same bytes unless w changed, and w is saved here too, so a backward pass after an
        # in-place update of w (forward A; optimizer.step(); forward B; backward A) fails in ctx.saved_tensors with autograd's own
        # "modified by an inplace operation" error before the stale image could be used (tests/test_gpu_conv.py).
        ctx.save_for_backward(x, w, pack_b)
        ctx.split = split
        if pack_f is None:
            pack_f = ops.conv3x3_pack(w, x.shape[2], False, split)
        return ops.conv3x3_apply(x, pack_f, w.shape[0], split)

    @staticmethod
    def backward(ctx, gy):
        x, w, pack_b = ctx.saved_tensors
        gy = ops._aligned(gy)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = ops.conv3x3_apply(gy, pack_b if pack_b is not None else ops.conv3x3_pack(w, x.shape[2], True, ctx.split), w.shape[1], ctx.split)
        gw = None
        if ctx.needs_input_grad[1]:
            if ops.conv3x3_wgrad_supported(x.shape, w.shape):
                gw = ops.conv3x3_backward_weight(gy, x, split=ctx.split)
            else:           # channel counts that are not multiples of 64: MIOpen
                gw = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return gx, gw, None, None, None


class ConvStemFunction(torch.autograd.Function):
    """ResNet.conv1 (7x7 / 2, 3 -> 64, 224 x 224 inputs) on csrc/conv_stem.hip: forward and weight gradient; the input is data, a
    requested input gradient falls back to MIOpen's backward-data."""

    @staticmethod
    def forward(ctx, x, w):
        x = ops._aligned(x)
        ctx.save_for_backward(x, w)
        return ops.conv_stem_forward(x, w)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = ops._aligned(gy)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        gw = ops.conv_stem_backward_weight(gy, x) if ctx.needs_input_grad[1] else None
        return gx, gw


def conv_stem(conv, x):
    """nn.Conv2d `conv` (the ResNet stem) applied to x through the HIP kernel when the shapes are the ones it takes."""
    if (x.is_cuda and x.dtype == torch.float32 and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1)
            and ops.conv_stem_supported(x.shape, conv.weight.shape, conv.stride, conv.padding)):
        return ConvStemFunction.apply(x, conv.weight)
    return conv(x)


class Conv1x1S2Function(torch.autograd.Function):
    """BasicBlock.downsample[0] (1x1 / stride 2) on csrc/conv1x1s2.hip: forward, backward-data (writes the zeros itself), backward-weight."""

    @staticmethod
    def forward(ctx, x, w):
        x = ops._aligned(x)
        ctx.save_for_backward(x, w)
        return ops.conv1x1s2_forward(x, w)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = ops._aligned(gy)
        gx = ops.conv1x1s2_backward_data(gy, w) if ctx.needs_input_grad[0] else None
        gw = ops.conv1x1s2_backward_weight(gy, x) if ctx.needs_input_grad[1] else None
        return gx, gw


def conv1x1s2(conv, x):
    """nn.Conv2d `conv` (a 1x1 / stride-2 shortcut) applied to x through the HIP kernels when the shapes are the ones they take."""
    if (x.is_cuda and x.dtype == torch.float32 and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1)
            and ops.conv1x1s2_supported(x.shape, conv.weight.shape, conv.stride, conv.padding)):
        return Conv1x1S2Function.apply(x, conv.weight)
    return conv(x)


class Conv3x3S2Function(torch.autograd.Function):
    """BasicBlock.conv1 of layer2-4 (3x3 / stride 2): forward on csrc/conv3x3.hip (stride-2 instance); backward-data as four parity
    sub-convolutions on the same kernel family (exact bf16x3 split, the default arithmetic), backward-weight on csrc/conv3x3_wgrad.hip with
    stride-2 patch addressing.  With `--hip.conv3x3_split!` (fp32 MFMA everywhere) or channel counts that are not multiples of 64 the
    two gradients go to MIOpen."""

    @staticmethod
    def forward(ctx, x, w):
        x = ops._aligned(x)
        ctx.save_for_backward(x, w)
        return ops.conv3x3s2_forward(x, w)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = ops._aligned(gy)
        from .model import resnet
        if resnet.HIP_CONV3X3_S2_GRADS and ops.conv3x3s2_grads_supported(x.shape, w.shape):
            gx = gw = None
            if ctx.needs_input_grad[0]:
                if resnet.HIP_CONV3X3_SPLIT:
                    gx = ops.conv3x3s2_backward_data(gy, w, x.shape[2])
                else:
                    gx = torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
            if ctx.needs_input_grad[1]:
                gw = ops.conv3x3s2_backward_weight(gy, x)
            return gx, gw
        gx, gw, _ = torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                        [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        return gx, gw


def conv3x3s2(conv, x):
    if (x.is_cuda and x.dtype == torch.float32 and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1)
            and ops.conv3x3s2_supported(x.shape, conv.weight.shape, conv.stride, conv.padding)):
        return Conv3x3S2Function.apply(x, conv.weight)
    return conv(x)


def conv3x3_takes(conv, x):
    return (x.is_cuda and x.dtype == torch.float32 and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1)
            and ops.conv3x3_supported(x.shape, conv.weight.shape, conv.stride, conv.padding))


def conv3x3(conv, x, packs=None):
    """nn.Conv2d `conv` applied to x: shapes the HIP kernel takes go through it (device tensors only), the rest through torch.
    packs: the Conv3x3PackSet holding conv.weight's images (refreshed by the caller), or None."""
    if not conv3x3_takes(conv, x):
        return conv(x)
    if packs is not None and id(conv.weight) in packs.index:
        return Conv3x3Function.apply(x, conv.weight, packs.get(conv.weight, 0), packs.get(conv.weight, 1), packs.split)
    from .model import resnet          # the arithmetic switch lives with the trunks (`--hip.conv3x3_split`)
    return Conv3x3Function.apply(x, conv.weight, None, None, bool(resnet.HIP_CONV3X3_SPLIT))


class BasicBlockFunction(torch.autograd.Function):
    """One autograd node for a whole torchvision BasicBlock with stride 1 and no shortcut convolution:
        out = relu( bn2( conv2( relu( bn1( conv1(x) ) ) ) ) + x )
    Round 4: the step was host-paced -- 34.6 ms of enqueue work for a 34.8 ms step, ~150 autograd.Function applies
    at 25-33 us each and as many backward-node dispatches.  This node runs the SAME eight ops launches in the SAME order as the four
    nodes it replaces (Conv3x3Function, BnActFunction, Conv3x3Function, BnActFunction) plus the gradient addition autograd did for x
    (convolution input and residual), so values and gradients are bit-identical; what goes away is three applies, four backward
    dispatches and one AccumulateGrad-style add node per block (18 of the 24 blocks of the two trunks take this path)."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, pf1, pb1, pf2, pb2, split, bn1_state, bn2_state, groups):
        # (round 4, second half: the eight Python operator wrappers of a block -- 61 us forward, 104 us backward of host time each -- became
        #  ONE C call each way, csrc/block.hip: the same launches in the same order)
        x = ops._aligned(x)
        out, saved = ops.basic_block_forward(x, pf1, pf2, g1, b1, g2, b2, bn1_state, bn2_state, split, groups)
        ctx.meta = (split, bn1_state[3], groups)
        # w1 / w2 are saved for the reason Conv3x3Function gives: a backward pass after an in-place update of the filters must fail in
        # ctx.saved_tensors instead of using the pack set's rewritten images
        ctx.save_for_backward(x, out, *saved, g1, b1, g2, b2, pb1, pb2, w1, w2)
        return out

    @staticmethod
    def backward(ctx, d_out):
        split, training, groups = ctx.meta
        x, out, y1, a1, y2, st1, st2, g1, b1, g2, b2, pb1, pb2, w1, w2 = ctx.saved_tensors
        dx, gw1, dg1, db1, gw2, dg2, db2 = ops.basic_block_backward(
            ops._aligned(d_out), x, (y1, a1, y2, st1, st2), out, pb1, pb2, g1, b1, g2, b2, training, split, groups,
            ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[4])
        return dx, gw1, dg1, db1, gw2, dg2, db2, None, None, None, None, None, None, None, None


def _bn_state(bn):
    training = bn.training or not bn.track_running_stats
    track = bn.track_running_stats
    return (bn.running_mean if track else None, bn.running_var if track else None,
            bn.num_batches_tracked if (track and training) else None, training, float(bn.momentum), float(bn.eps))


def basic_block_takes(block, x, packs):
    """Stride-1 block without shortcut convolution whose two filters are in the pack set and whose BatchNorms take the fused path."""
    if packs is None or block.downsample is not None or block.conv1.stride != (1, 1):
        return False
    if id(block.conv1.weight) not in packs.index or id(block.conv2.weight) not in packs.index:
        return False
    if not (conv3x3_takes(block.conv1, x) and conv3x3_takes(block.conv2, x) and _bn_fusable(block.bn1, x) and _bn_fusable(block.bn2, x)):
        return False
    if not (ops.conv3x3_wgrad_supported(x.shape, block.conv1.weight.shape) and ops.conv3x3_wgrad_supported(x.shape, block.conv2.weight.shape)):
        return False
    return block.bn1.training == block.bn2.training


def basic_block(block, x, packs, groups=1):
    c1, c2 = block.conv1.weight, block.conv2.weight
    return BasicBlockFunction.apply(x, c1, block.bn1.weight, block.bn1.bias, c2, block.bn2.weight, block.bn2.bias,
                                    packs.get(c1, 0), packs.get(c1, 1), packs.get(c2, 0), packs.get(c2, 1), packs.split,
                                    _bn_state(block.bn1), _bn_state(block.bn2), groups)


class CameraRaysFunction(torch.autograd.Function):
    """pose [B,3,4], intr [B,3,3], ray_idx [B,R] | None -> cam_loc [B*R,3], ray_dirs [B*R,3] (unit), depth_fac [B*R]
    (reference utils/camera.py:157-196 + model/renderer.py:69-76, perspective camera) in one launch each way."""

    @staticmethod
    def forward(ctx, pose, intr, ray_idx, n_rays, width):
        ctx.set_materialize_grads(False)
        pose, intr = pose.contiguous().float(), intr.contiguous().float()
        if ray_idx is not None:
            ray_idx = ray_idx.contiguous().long()
        out = ops.camera_rays_forward(pose, intr, ray_idx, n_rays, width)
        ctx.save_for_backward(pose, intr, ray_idx)
        ctx.meta = (n_rays, width)
        return out

    @staticmethod
    def backward(ctx, g_loc, g_dirs, g_df):
        pose, intr, ray_idx = ctx.saved_tensors
        n_rays, width = ctx.meta
        c = lambda t: t.contiguous() if t is not None else None
        g_pose, g_intr = ops.camera_rays_backward(pose, intr, ray_idx, n_rays, width, c(g_loc), c(g_dirs), c(g_df))
        return g_pose, g_intr, None, None, None


class PoseFromTrigFunction(torch.autograd.Function):
    """Estimator outputs -> (pose [B,3,4], intr [B,3,3]) (reference model/graph.py:272-293)."""

    @staticmethod
    def forward(ctx, azim, elev, theta, scale_focal, scale_dist, cam_dist, focal, width, height):
        ctx.set_materialize_grads(False)
        args = [t.contiguous().float() for t in (azim, elev, theta, scale_focal, scale_dist)]
        ctx.save_for_backward(*args)
        ctx.meta = (cam_dist, focal, width, height)
        return ops.pose_from_trig_forward(*args, cam_dist, focal, width, height)

    @staticmethod
    def backward(ctx, g_pose, g_intr):
        args = ctx.saved_tensors
        B = args[0].shape[0]
        if g_pose is None:
            g_pose = torch.zeros(B, 3, 4, device=args[0].device)
        if g_intr is None:
            g_intr = torch.zeros(B, 3, 3, device=args[0].device)
        ga, ge, gt, gsf, gsd = ops.pose_from_trig_backward(*args, *ctx.meta, g_pose.contiguous(), g_intr.contiguous())
        return ga, ge, gt, gsf, gsd, None, None, None, None


class EstimatorHeadFunction(torch.autograd.Function):
    """extr_fc / size_fc / perspect_fc outputs -> the estimator's five results (reference model/view_estimator.py:62-75), one launch
    each way.  The rows are `groups` stacked image sets: the results come back PER SET, 5 * groups tensors (azim, elev, theta
    [B,2], scale_focal, scale_dist [B] of set 0, then of set 1, ...), so that no slice node (fill + copy + add in its backward)
    sits between the estimator and its consumers."""

    @staticmethod
    def forward(ctx, trig, size_lin, persp_lin, size_range, persp_range, groups):
        ctx.set_materialize_grads(False)
        trig, size_lin, persp_lin = ops._f32c(trig), ops._f32c(size_lin.reshape(-1)), ops._f32c(persp_lin.reshape(-1))
        N = trig.shape[0]
        assert N % groups == 0
        B = N // groups
        o = ops.estimator_head_forward(trig, size_lin, persp_lin, size_range, persp_range)
        ctx.save_for_backward(trig, size_lin, persp_lin)
        ctx.meta = (size_range, persp_range, groups)
        outs = []
        for g in range(groups):
            for k in range(3):
                outs.append(o[2 * N * k + 2 * B * g:2 * N * k + 2 * B * (g + 1)].view(B, 2))
            outs.append(o[6 * N + B * g:6 * N + B * (g + 1)])
            outs.append(o[7 * N + B * g:7 * N + B * (g + 1)])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        trig, size_lin, persp_lin = ctx.saved_tensors
        size_range, persp_range, groups = ctx.meta
        g_trig, g_size, g_persp = ops.estimator_head_backward(trig, size_lin, persp_lin, size_range, persp_range, list(grads), groups)
        return g_trig, g_size.view(-1, 1), g_persp.view(-1, 1), None, None, None


class CameraPriorLossFunction(torch.autograd.Function):
    """cam_margin_loss, cam_uniform_loss and cam_sym_loss (reference model/loss.py:99-167) as three scalars from one launch; the
    gradients w.r.t. the six [B,2] estimator outputs are produced alongside and scaled in the backward (one launch)."""

    @staticmethod
    def forward(ctx, azim, elev, theta, f_azim, f_elev, f_theta, elev_range, theta_range, margin_eps, emd_p):
        ctx.set_materialize_grads(False)
        args = [ops._f32c(t) for t in (azim, elev, theta, f_azim, f_elev, f_theta)]
        out, grads = ops.camera_prior_forward(*args, elev_range, theta_range, margin_eps, emd_p)
        ctx.save_for_backward(grads)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, G_margin, G_uniform, G_sym):
        grads, = ctx.saved_tensors
        g = ops.camera_prior_backward(grads, G_margin, G_uniform, G_sym)
        return g[0], g[1], g[2], g[3], g[4], g[5], None, None, None, None


class TransformNormalFunction(torch.autograd.Function):
    """camera.transform_normal (reference utils/camera.py:98-103): [B,R,3] camera-frame normals (data) rotated by the predicted
    pose; the backward is the per-image 3x3 sum that reaches the view estimator through the normal loss."""

    @staticmethod
    def forward(ctx, normals, pose):
        normals, pose = ops._f32c(normals), ops._f32c(pose)
        ctx.save_for_backward(normals)
        return ops.transform_normal_forward(normals, pose)

    @staticmethod
    def backward(ctx, g_out):
        normals, = ctx.saved_tensors
        return None, ops.transform_normal_backward(normals, ops._f32c(g_out))


class LossTotalFunction(torch.autograd.Function):
    """loss.all = sum_k float(w_k) * loss_k in key order (reference model/runner.py:294-305) and the NaN/Inf flag of all terms: one
    launch instead of ~8 per key.  apply(weights tuple, *scalars) -> (total, bad)."""

    @staticmethod
    def forward(ctx, weights, *values):
        values = [ops._f32c(v) for v in values]
        total, bad = ops.loss_total_forward(values, weights)
        ctx.weights = weights
        ctx.mark_non_differentiable(bad)
        return total, bad

    @staticmethod
    def backward(ctx, G, _):
        g = ops.loss_total_backward(ctx.weights, G)
        return (None,) + tuple(g[k] for k in range(len(ctx.weights)))
