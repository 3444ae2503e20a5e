"""Fused 1x1 Bottleneck_Linear blocks (csrc/bottleneck.hip) against the stock torch modules of the same block -- two nn.Conv2d(C, C, 1,
bias=False) + two nn.BatchNorm2d on 1x1 maps, residual, ReLU (reference model/view_estimator.py:6-33, model/graph.py:16-40): outputs,
running statistics, every gradient, in training and evaluation mode, with 1..3 stacked sub-batches; bit-reproducible; and the two block
kinds of the product (estimator head with zero-initialised bn2.weight, latent projector) inside their modules."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


class StockBlock(nn.Module):
    def __init__(self, C):
        super().__init__()
        self.linear1, self.bn1 = nn.Conv2d(C, C, 1, bias=False), nn.BatchNorm2d(C)
        self.linear2, self.bn2 = nn.Conv2d(C, C, 1, bias=False), nn.BatchNorm2d(C)

    def forward(self, x, groups=1):
        outs = []
        for v in x.chunk(groups):                      # `groups` consecutive calls, as the reference makes them
            v4 = v[..., None, None]
            o = torch.relu(self.bn1(self.linear1(v4)))
            outs.append(torch.relu(self.bn2(self.linear2(o)) + v4).flatten(1))
        return torch.cat(outs, 0)


def _pair(C, seed):
    from shapeclipper_amd.model.view_estimator import Bottleneck_Linear
    torch.manual_seed(seed)
    ref = StockBlock(C).double()
    with torch.no_grad():
        for bn in (ref.bn1, ref.bn2):
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
            bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
    mine = Bottleneck_Linear(C, zero_init=False)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    return ref, mine.cuda()


@pytest.mark.parametrize("N,C,groups,training", [(32, 512, 1, True), (96, 512, 3, True), (64, 512, 2, True), (48, 128, 3, True), (20, 64, 1, True),
                                                  (96, 512, 3, False), (128, 256, 4, True)])
def test_block_matches_the_stock_modules(N, C, groups, training):
    ref, mine = _pair(C, seed=N + C)
    ref.train(training); mine.train(training)
    x = torch.randn(N, C, dtype=torch.float64) * 1.5
    xr = x.clone().requires_grad_(True)
    xm = x.float().cuda().requires_grad_(True)
    cot = torch.randn(N, C, dtype=torch.float64)
    yr = ref(xr, groups)
    ym = mine(xm, groups=groups)
    assert (ym.double().cpu() - yr).abs().max() < 2e-5 * max(1.0, float(yr.abs().max()))
    (yr * cot).sum().backward()
    (ym * cot.float().cuda()).sum().backward()
    rel = lambda a, b: float((a.double().cpu() - b).abs().max()) / max(float(b.abs().max()), 1e-6)
    assert rel(xm.grad, xr.grad) < 5e-5, rel(xm.grad, xr.grad)
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        assert pm.grad is not None and rel(pm.grad.view(pr.grad.shape), pr.grad) < 5e-5, (n, rel(pm.grad.view(pr.grad.shape), pr.grad))
    for (n, br), (_, bm) in zip(ref.named_buffers(), mine.named_buffers()):
        if br.dtype.is_floating_point:
            assert rel(bm, br) < 1e-5, n
        else:
            assert int(bm) == int(br), n                 # num_batches_tracked: + groups in training, unchanged in evaluation


def test_block_is_bit_reproducible_and_takes_the_fused_path():
    from shapeclipper_amd import _lib
    _, mine = _pair(512, seed=1)
    x = torch.randn(96, 512, device="cuda")
    outs = []
    for _ in range(2):
        xi = x.clone().requires_grad_(True)
        _lib.TIMING = {}
        try:
            y = mine(xi, groups=3)
            y.square().sum().backward()
            torch.cuda.synchronize()
            calls = {k: len(v) for k, v in _lib.TIMING.items()}
        finally:
            _lib.TIMING = None
        outs.append((y.detach().clone(), xi.grad.clone(), mine.linear1.weight.grad.clone(), mine.bn2.weight.grad.clone()))
        mine.zero_grad()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert "sc_bn_act_forward" not in calls                          # no separate BatchNorm launches: the block is 2 + 3 fused launches


def test_shapes_outside_the_kernel_keep_the_operator_form():
    from shapeclipper_amd import ops
    assert not ops.linear_bn_supported(200, 512, 512, 1) and not ops.linear_bn_supported(96, 512, 512, 5) and not ops.linear_bn_supported(32, 96, 96, 1)
    ref, mine = _pair(96, seed=3)
    x = torch.randn(16, 96, dtype=torch.float64)
    y = mine(x.float().cuda())
    assert (y.double().cpu() - ref(x)).abs().max() < 2e-5
