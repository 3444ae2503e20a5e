"""functional.BasicBlockFunction (one autograd node per stride-1 BasicBlock, round 4) against the four nodes it replaces: same kernels in the
same order, so the output, all parameter gradients and the BatchNorm running statistics are BIT-identical in the default (split) arithmetic."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("groups,split", [(1, True), (3, True), (2, False)])
def test_fused_block_equals_the_four_node_path_bit_for_bit(groups, split):
    from shapeclipper_amd.model import resnet
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    old = (resnet.FUSED_BLOCK, resnet.HIP_CONV3X3_SPLIT)
    try:
        resnet.HIP_CONV3X3_SPLIT = split
        net_a = resnet.ResNet([2, 2, 2, 2]).to(dev).train()
        with torch.no_grad():
            for m in net_a.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
        net_b = copy.deepcopy(net_a)
        x = torch.randn(4 * groups, 3, 224, 224, device=dev)
        outs = []
        for net, fused in ((net_a, True), (net_b, False)):
            resnet.FUSED_BLOCK = fused
            xi = x.clone().requires_grad_(True)
            y = net(xi, groups=groups)
            (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
            outs.append((y.detach(), {n: p.grad.clone() for n, p in net.named_parameters()},
                         {n: b.clone() for n, b in net.named_buffers()}))
        (ya, ga, ba), (yb, gb, bb) = outs
        assert torch.equal(ya, yb)
        for n in ga:
            if split:        # the default arithmetic: every sum of the pass has a fixed order (tests/test_gpu_determinism.py)
                assert torch.equal(ga[n], gb[n]), "%s: max %g" % (n, float((ga[n] - gb[n]).abs().max()))
            else:            # `--hip.conv3x3_split!`: the four-node path itself differs from run to run in the last bits (tools/diag_block.py)
                assert float((ga[n] - gb[n]).abs().max()) <= 1e-4 * float(gb[n].abs().max()) + 1e-6, n
        for n in ba:
            assert torch.equal(ba[n], bb[n]), n
        # the fused node really was used: layer1's blocks and the second block of layers 2-4 are stride 1 without shortcut
        resnet.FUSED_BLOCK = True
        from shapeclipper_amd import functional
        calls = []
        orig = functional.BasicBlockFunction.apply
        functional.basic_block.__globals__["BasicBlockFunction"] = type("Spy", (), {"apply": staticmethod(lambda *a: (calls.append(1), orig(*a))[1])})
        try:
            net_a(x, groups=groups)
        finally:
            functional.basic_block.__globals__["BasicBlockFunction"] = functional.BasicBlockFunction
        assert len(calls) == 5                       # ResNet-18: 2 + 1 + 1 + 1
    finally:
        resnet.FUSED_BLOCK, resnet.HIP_CONV3X3_SPLIT = old


def test_fused_block_eval_mode_and_frozen_input():
    """Evaluation mode (running statistics) and an input that needs no gradient take the same node."""
    from shapeclipper_amd.model import resnet
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    old = resnet.FUSED_BLOCK
    try:
        net = resnet.ResNet([2, 2, 2, 2]).to(dev).eval()
        x = torch.randn(2, 3, 224, 224, device=dev)
        resnet.FUSED_BLOCK = True
        with torch.no_grad():
            ya = net(x)
        resnet.FUSED_BLOCK = False
        with torch.no_grad():
            yb = net(x)
        assert torch.equal(ya, yb)
    finally:
        resnet.FUSED_BLOCK = old
