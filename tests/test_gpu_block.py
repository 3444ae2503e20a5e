"""sc_basic_block_forward / _backward (csrc/block.hip): ONE C call per torchvision BasicBlock (stride 1, no shortcut convolution; reference
model/graph.py:50-54, model/view_estimator.py:40-42) must equal, bit for bit, the operator sequence it replaces -- the entry points
sc_conv3x3_forward_split, sc_bn_act_forward / _backward, sc_conv3x3_wgrad_split called one at a time -- and the stock torch operators to the
convolution kernels' own tolerance."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _block_inputs(B, C, S, groups, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s, scale=1.0: torch.randn(*s, device="cuda", generator=g) * scale
    x = r(B, C, S, S)
    w1, w2 = r(C, C, 3, 3, scale=(2.0 / (9 * C)) ** 0.5), r(C, C, 3, 3, scale=(2.0 / (9 * C)) ** 0.5)
    g1, b1, g2, b2 = 1 + 0.1 * r(C), 0.1 * r(C), 1 + 0.1 * r(C), 0.1 * r(C)
    d_out = r(B, C, S, S)
    return x, w1, w2, g1, b1, g2, b2, d_out


def _bn_state(C, training=True):
    return (torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda"), training, 0.1, 1e-5)


@pytest.mark.parametrize("B,C,S,groups", [(4, 64, 56, 2), (6, 128, 28, 3), (8, 256, 14, 2), (6, 512, 7, 3), (2, 64, 14, 1)])
@pytest.mark.parametrize("split", [True, False])
def test_basic_block_call_equals_operator_sequence(B, C, S, groups, split):
    from shapeclipper_amd import ops
    x, w1, w2, g1, b1, g2, b2, d_out = _block_inputs(B, C, S, groups)
    pf1, pb1 = ops.conv3x3_pack(w1, S, False, split), ops.conv3x3_pack(w1, S, True, split)
    pf2, pb2 = ops.conv3x3_pack(w2, S, False, split), ops.conv3x3_pack(w2, S, True, split)

    # the operator sequence (what functional.BasicBlockFunction issued before the C call existed)
    s1, s2 = _bn_state(C), _bn_state(C)
    y1 = ops.conv3x3_apply(x, pf1, C, split)
    a1, st1 = ops.bn_act_forward(y1, None, g1, b1, s1[0], s1[1], s1[2], True, 0.1, 1e-5, True, groups)
    y2 = ops.conv3x3_apply(a1, pf2, C, split)
    out, st2 = ops.bn_act_forward(y2, x, g2, b2, s2[0], s2[1], s2[2], True, 0.1, 1e-5, True, groups)
    dy2, dres, dg2, db2 = ops.bn_act_backward(d_out, y2, out, g2, b2, st2, True, True, True, True, groups)
    da1 = ops.conv3x3_apply(dy2, pb2, C, split)
    gw2 = ops.conv3x3_backward_weight(dy2, a1, split=split)
    dy1, _, dg1, db1 = ops.bn_act_backward(da1, y1, None, g1, b1, st1, True, True, True, False, groups)
    dx = ops.conv3x3_apply(dy1, pb1, C, split) + dres
    gw1 = ops.conv3x3_backward_weight(dy1, x, split=split)

    # one call each way
    t1, t2 = _bn_state(C), _bn_state(C)
    out_c, saved = ops.basic_block_forward(x, pf1, pf2, g1, b1, g2, b2, t1, t2, split, groups)
    got = ops.basic_block_backward(d_out, x, saved, out_c, pb1, pb2, g1, b1, g2, b2, True, split, groups, True, True, True)
    assert torch.equal(out_c, out)
    for name, a, b in zip(("y1", "a1", "y2", "st1", "st2"), saved, (y1, a1, y2, st1, st2)):
        assert torch.equal(a, b), name
    for name, a, b in zip(("dx", "gw1", "dgamma1", "dbeta1", "gw2", "dgamma2", "dbeta2"), got, (dx, gw1, dg1, db1, gw2, dg2, db2)):
        assert torch.equal(a, b), name
    for a, b in zip(t1[:3] + t2[:3], s1[:3] + s2[:3]):           # running statistics and num_batches_tracked
        assert torch.equal(a, b)
    assert int(t1[2]) == groups

    # without the input gradient / the filter gradients
    part = ops.basic_block_backward(d_out, x, saved, out_c, pb1, pb2, g1, b1, g2, b2, True, split, groups, False, False, True)
    assert part[0] is None and part[1] is None and torch.equal(part[4], gw2) and torch.equal(part[2], dg1)


def test_basic_block_call_matches_stock_operators():
    """... and the stock torch operators (float64 reference) at the convolution tests' own bar."""
    from shapeclipper_amd import ops
    B, C, S, groups = 4, 64, 28, 1
    x, w1, w2, g1, b1, g2, b2, d_out = _block_inputs(B, C, S, groups, seed=3)
    pf1, pb1, pf2, pb2 = (ops.conv3x3_pack(w, S, f, True) for w in (w1, w2) for f in (False, True))
    out, saved = ops.basic_block_forward(x, pf1, pf2, g1, b1, g2, b2, _bn_state(C), _bn_state(C), True, groups)
    dx, gw1, dg1, db1, gw2, dg2, db2 = ops.basic_block_backward(d_out, x, saved, out, pb1, pb2, g1, b1, g2, b2, True, True, groups, True, True, True)
    xd = x.double().requires_grad_(True)
    p = [t.double().requires_grad_(True) for t in (w1, g1, b1, w2, g2, b2)]
    F = torch.nn.functional
    h = F.relu(F.batch_norm(F.conv2d(xd, p[0], None, 1, 1), None, None, p[1], p[2], True, 0.1, 1e-5))
    ref = F.relu(F.batch_norm(F.conv2d(h, p[3], None, 1, 1), None, None, p[4], p[5], True, 0.1, 1e-5) + xd)
    grads = torch.autograd.grad(ref, [xd] + p, d_out.double())
    assert (out.double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    for name, a, b in zip(("dx", "gw1", "dg1", "db1", "gw2", "dg2", "db2"), (dx, gw1, dg1, db1, gw2, dg2, db2), grads):
        assert (a.double() - b).abs().max() <= 1e-4 * b.abs().max() + 1e-6, name
