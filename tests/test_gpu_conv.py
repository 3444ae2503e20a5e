"""3x3 stride-1 convolutions of the ResNet trunks (csrc/conv3x3.hip) against torch's operators (MIOpen, fp32) and, for one small
case, against a float64 convolution: SURVEY 8f-1 -- the reference runs torchvision ResNets (model/graph.py:50-54,
model/view_estimator.py:40-42), so torch's conv2d IS the reference arithmetic.  Tolerance: fp32 accumulation over K = 9 Cin
products in a different order than MIOpen's Winograd / implicit GEMM -- 2e-5 of the output scale (achieved: <= 3e-6, one
sequential fp32 accumulator per output over all 9 Cin products; MIOpen's blocked sums reach <= 7e-7); errors are printed."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("side,cin,cout,batch", [(56, 64, 64, 3), (28, 128, 128, 5), (14, 256, 256, 7), (7, 512, 512, 13),
                                                  (7, 512, 512, 1), (14, 64, 128, 2), (28, 64, 72, 1), (56, 8, 64, 1)])
def test_conv3x3_forward_and_backward_data(side, cin, cout, batch):
    from shapeclipper_amd import ops
    torch.manual_seed(side + cin + batch)
    dev = torch.device("cuda:0")
    x = torch.randn(batch, cin, side, side, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    assert ops.conv3x3_supported(x.shape, w.shape)
    y = ops.conv3x3_forward(x, w)
    y64 = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    y_t = torch.nn.functional.conv2d(x, w, None, 1, 1)
    e_hip, e_torch = _rel(y.double(), y64), _rel(y_t.double(), y64)
    gy = torch.randn_like(y)
    gx = ops.conv3x3_backward_data(gy, w)
    gx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), 1, 1)
    e_bwd = _rel(gx.double(), gx64)
    print("conv3x3 %dx%d %d>%d B=%d: forward %.2e of max (torch/MIOpen: %.2e), backward-data %.2e" % (side, side, cin, cout, batch, e_hip, e_torch, e_bwd))
    assert e_hip < 2e-5 and e_bwd < 2e-5


def test_conv3x3_rejects_other_shapes():
    from shapeclipper_amd import ops
    assert not ops.conv3x3_supported((2, 64, 32, 32), (64, 64, 3, 3))
    assert not ops.conv3x3_supported((2, 64, 56, 56), (64, 64, 1, 1))
    assert not ops.conv3x3_supported((2, 64, 56, 56), (128, 64, 3, 3), stride=2)
    x = torch.randn(1, 64, 32, 32, device="cuda:0")
    with pytest.raises(RuntimeError):
        ops.conv3x3_forward(x, torch.randn(64, 64, 3, 3, device="cuda:0"))


@pytest.mark.parametrize("split", [False, True])
def test_resnet34_hip_convolutions_equal_miopen(split):
    """The whole trunk, forward and backward, with the 3x3 / stride-1 layers on csrc/conv3x3.hip vs on MIOpen (resnet.HIP_CONV3X3);
    split: their forward / backward-data products on the bf16 matrix pipe (resnet.HIP_CONV3X3_SPLIT)."""
    import copy
    from shapeclipper_amd.model import resnet
    torch.manual_seed(5)
    net = resnet.build("resnet34").cuda().train()
    x = torch.randn(6, 3, 224, 224, device="cuda")
    res = []
    saved = (resnet.HIP_CONV3X3, resnet.HIP_CONV3X3_SPLIT)
    for hip in (False, True):
        resnet.HIP_CONV3X3, resnet.HIP_CONV3X3_SPLIT = hip, hip and split
        try:
            m = copy.deepcopy(net)
            xi = x.clone().requires_grad_(True)
            y = m(xi, groups=3)
            y.square().mean().backward()
            res.append((y.detach(), xi.grad.clone(), m.conv1.weight.grad.clone(), m.layer1[0].conv1.weight.grad.clone(),
                        m.layer3[2].conv2.weight.grad.clone(), m.layer4[2].conv2.weight.grad.clone()))
        finally:
            resnet.HIP_CONV3X3, resnet.HIP_CONV3X3_SPLIT = saved
    for a, b, name in zip(res[1], res[0], ("logits", "d input", "d conv1.weight", "d layer1.0.conv1.weight", "d layer3.2.conv2.weight",
                                           "d layer4.2.conv2.weight")):
        rel = float((a - b).norm() / b.norm().clamp_min(1e-12))
        print("resnet34 hip-conv%s vs MIOpen, %s: relative L2 difference %.2e" % (" (bf16 split)" if split else "", name, rel))
        # 33 BN layers deep with 2 images per BN group: rounding differences (and the odd ReLU-kink flip) are amplified on the way back
        assert rel < (1e-4 if name == "logits" else 3e-2), name


def test_pack_set_equals_single_packs():
    """One launch packs every filter of a trunk (forward + backward-data images): same bytes as the per-filter entry point."""
    from shapeclipper_amd import ops
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    items = [(torch.randn(64, 64, 3, 3, device=dev), 56), (torch.randn(128, 128, 3, 3, device=dev), 28), (torch.randn(72, 64, 3, 3, device=dev), 28),
             (torch.randn(256, 256, 3, 3, device=dev), 14), (torch.randn(512, 512, 3, 3, device=dev), 7)]
    for split in (False, True):         # split: the unit-per-workgroup kernel (sc_conv3x3_pack_multi_units), also on a 72-channel filter
        packs = ops.Conv3x3PackSet(items, split=split)
        assert packs.units == split
        packs.buf.fill_(float("nan"))
        packs.refresh()
        for w, side in items:
            for flip in (False, True):
                a, b = packs.get(w, flip), ops.conv3x3_pack(w, side, flip, split)
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (tuple(w.shape), side, flip, split)
        assert not packs.stale()
    import time
    big = [(torch.randn(c, c, 3, 3, device=dev), s) for c, s, k in ((64, 56, 6), (128, 28, 7), (256, 14, 11), (512, 7, 5)) for _ in range(k)]
    packs = ops.Conv3x3PackSet(big, split=True)
    for fn in ("units", "words"):
        packs.units = fn == "units"
        packs.refresh(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            packs.refresh()
        e.record(); torch.cuda.synchronize()
        print("ResNet-34 filter pack (%s kernel): %.1f us" % (fn, s.elapsed_time(e) / 5 * 1e3))


@pytest.mark.parametrize("side,cin,cout,batch", [(56, 64, 64, 3), (28, 128, 128, 5), (14, 256, 256, 7), (7, 512, 512, 13), (7, 512, 512, 1),
                                                  (14, 64, 128, 2), (28, 128, 64, 1), (56, 64, 64, 1), (7, 64, 64, 2)])
def test_conv3x3_backward_weight(side, cin, cout, batch):
    from shapeclipper_amd import ops
    torch.manual_seed(side + cin + batch)
    dev = torch.device("cuda:0")
    x = torch.randn(batch, cin, side, side, device=dev)
    gy = torch.randn(batch, cout, side, side, device=dev)
    assert ops.conv3x3_wgrad_supported(x.shape, (cout, cin, 3, 3))
    dw = ops.conv3x3_backward_weight(gy, x)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), gy.double(), 1, 1)
    dw_t = torch.ops.aten.convolution_backward(gy, x, torch.zeros(cout, cin, 3, 3, device=dev), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                               [False, True, False])[1]
    e_hip, e_torch = _rel(dw.double(), dw64), _rel(dw_t.double(), dw64)
    print("conv3x3 wgrad %dx%d %d>%d B=%d: %.2e of max (torch/MIOpen: %.2e)" % (side, side, cin, cout, batch, e_hip, e_torch))
    assert e_hip < 2e-5
    assert torch.equal(dw, ops.conv3x3_backward_weight(gy, x))          # fixed summation order


@pytest.mark.parametrize("side,cin,cout,batch", [(56, 64, 64, 3), (28, 128, 128, 5), (14, 256, 256, 7), (7, 512, 512, 13), (7, 512, 512, 1),
                                                  (14, 64, 128, 2), (28, 128, 64, 1), (56, 64, 64, 1), (7, 64, 64, 2), (14, 128, 64, 96), (7, 64, 64, 3)])
def test_conv3x3_backward_weight_split_is_fp32_accurate(side, cin, cout, batch):
    """sc_conv3x3_wgrad_split (both operands as exact three-piece bf16 splits, six products, fp32 accumulate): the fp32 kernel's bar
    (2e-5 of the output scale against float64) on operands of mixed magnitudes, no worse than 2x the fp32-MFMA kernel on the same
    tensors, bit-identical run to run, and a single hot pixel lands in exactly the nine taps it belongs to."""
    from shapeclipper_amd import ops
    torch.manual_seed(side + cin + batch)
    dev = torch.device("cuda:0")
    x = torch.randn(batch, cin, side, side, device=dev) * torch.rand(batch, cin, 1, 1, device=dev) * 4
    gy = torch.randn(batch, cout, side, side, device=dev) * torch.rand(batch, cout, 1, 1, device=dev) * 1e-2
    dw = ops.conv3x3_backward_weight(gy, x, split=True)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), gy.double(), 1, 1)
    e_split, e_fp32 = _rel(dw.double(), dw64), _rel(ops.conv3x3_backward_weight(gy, x).double(), dw64)
    print("conv3x3 wgrad split %dx%d %d>%d B=%d: %.2e of max (fp32 MFMA kernel: %.2e)" % (side, side, cin, cout, batch, e_split, e_fp32))
    assert e_split < 2e-5 and e_split < 2 * e_fp32 + 1e-7
    assert torch.equal(dw, ops.conv3x3_backward_weight(gy, x, split=True))
    # one non-zero pixel in each operand: exact products, every (ky, kx) addressed on its own, borders and corners included
    for (yy, xx) in ((0, 0), (side - 1, side - 1), (side // 2, side - 1), (side - 1, 0), (3, 2)):
        x1, g1 = torch.zeros_like(x), torch.zeros_like(gy)
        b = batch - 1
        x1[b, 5, yy, xx] = 1.5
        for ky in range(3):
            for kx in range(3):
                oy, ox = yy - ky + 1, xx - kx + 1          # output pixel that meets input (yy, xx) through tap (ky, kx)
                if 0 <= oy < side and 0 <= ox < side:
                    g1[b, 7, oy, ox] = float(1 + ky * 3 + kx)
        d1 = ops.conv3x3_backward_weight(g1, x1, split=True)
        ref = torch.nn.grad.conv2d_weight(x1.double(), (cout, cin, 3, 3), g1.double(), 1, 1).float()
        assert torch.equal(d1, ref), (yy, xx)


@pytest.mark.parametrize("side,cin,cout,batch", [(56, 64, 64, 3), (28, 128, 128, 5), (14, 256, 256, 7), (7, 512, 512, 13), (7, 512, 512, 1),
                                                  (14, 64, 128, 2), (28, 64, 72, 1), (56, 8, 64, 1)])
def test_conv3x3_split_bf16_is_fp32_accurate(side, cin, cout, batch):
    """--hip.conv3x3_split: three-piece bf16 operands, six products, fp32 accumulate.  The bar is the fp32 kernels' own: 2e-5 of the
    output scale against float64, and no worse than 2x the fp32-MFMA kernel's error on the same tensors (printed)."""
    from shapeclipper_amd import ops
    torch.manual_seed(side + cin + batch)
    dev = torch.device("cuda:0")
    x = torch.randn(batch, cin, side, side, device=dev) * torch.rand(batch, cin, 1, 1, device=dev) * 4       # mixed magnitudes
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    y64 = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    e_fp32 = _rel(ops.conv3x3_forward(x, w).double(), y64)
    e_split = _rel(ops.conv3x3_forward(x, w, split=True).double(), y64)
    gy = torch.randn_like(y64, dtype=torch.float32)
    gx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), 1, 1)
    e_bwd = _rel(ops.conv3x3_backward_data(gy, w, split=True).double(), gx64)
    print("conv3x3 split %dx%d %d>%d B=%d: forward %.2e of max (fp32 MFMA kernel %.2e), backward-data %.2e" % (side, side, cin, cout, batch, e_split, e_fp32, e_bwd))
    assert e_split < 2e-5 and e_bwd < 2e-5 and e_split < 2 * e_fp32 + 2e-7


def test_split_is_the_default_arithmetic():
    """VERDICT r02 ruling: the three-piece split is the default of the 3x3 / stride-1 forward and backward-data products."""
    from shapeclipper_amd.model import resnet
    from shapeclipper_amd.utils import options
    assert resnet.HIP_CONV3X3_SPLIT is True and options.HIP_DEFAULTS["hip"]["conv3x3_split"] is True


@pytest.mark.parametrize("case", ["tiny", "range30", "huge"])
def test_conv3x3_split_edge_magnitudes(case):
    """Where the three-piece split could differ from fp32 arithmetic in kind, not in the last bits (conditions of the VERDICT r02 ruling):
      tiny    operands around the smallest normal fp32 number (1.2e-38): the second and third pieces of such a number are below the bf16 /
              fp32 normal range, and every product is subnormal in fp32 -- the fp32-MFMA kernel keeps subnormals, the split may flush them.
              Bar: absolute error <= 2^-126 x the number of summed products (each lost piece is below the smallest normal) on top of the
              usual relative bar, i.e. the result is right to within the subnormal range;
      range30 one reduction mixes input channels whose magnitudes span 2^30: the low pieces of the small channels vanish against the large
              ones exactly as their low mantissa bits do in fp32 -- the usual bar (2e-5 of the output scale) must hold;
      huge    operands of 1e18: products 1e36..1e38 stay finite in both kernels (no spurious overflow from the piece products)."""
    from shapeclipper_amd import ops
    torch.manual_seed(11)
    dev = torch.device("cuda:0")
    B, C, S = 2, 64, 14
    x = torch.randn(B, C, S, S, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
    abs_tol = 0.0
    gy = torch.randn(B, C, S, S, device=dev)
    if case == "tiny":
        x, gy = x * 2.0 ** -120, gy * 2.0 ** -120
        w = w * 2.0 ** -4                    # products around 2^-126: the fp32 result itself is subnormal
        abs_tol = 9 * C * 2.0 ** -126
    elif case == "range30":
        x = x * (2.0 ** torch.linspace(0, 30, C, device=dev)).view(1, C, 1, 1)
        gy = gy * (2.0 ** torch.linspace(0, 30, C, device=dev)).view(1, C, 1, 1)
    else:
        x, w, gy = x * 1e18, w * 1e18, gy * 1e18
    y64 = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    gx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), 1, 1)
    scale, scale_b = float(y64.abs().max()), float(gx64.abs().max())
    for split in (False, True):
        y = ops.conv3x3_forward(x, w, split=split)
        gx = ops.conv3x3_backward_data(gy, w, split=split)
        err = float((y.double() - y64).abs().max())
        err_b = float((gx.double() - gx64).abs().max())
        print("conv3x3 %s, %s: forward |error| %.3e (output scale %.3e), backward-data %.3e (scale %.3e)"
              % (case, "bf16x3 split" if split else "fp32 MFMA", err, scale, err_b, scale_b))
        assert torch.isfinite(y).all() and torch.isfinite(gx).all()
        assert err <= 2e-5 * scale + abs_tol and err_b <= 2e-5 * scale_b + abs_tol


def test_conv3x3_split_propagates_non_finite_values():
    """An Inf or NaN input must poison exactly the outputs whose receptive field contains it, in both arithmetics (the training loop's
    finite check relies on it).  With split operands Inf - Inf of the residual makes the poisoned outputs NaN where fp32 arithmetic
    gives +-Inf: `not finite` is the contract, the flavour is not."""
    from shapeclipper_amd import ops
    torch.manual_seed(12)
    dev = torch.device("cuda:0")
    x = torch.randn(2, 64, 14, 14, device=dev)
    w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    x[0, 3, 5, 7] = float("inf")
    x[1, 60, 0, 13] = float("nan")
    x[1, 7, 9, 2] = float("-inf")
    ref = ~torch.isfinite(torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1))
    assert ref.any() and not ref.all()
    for split in (False, True):
        y = ops.conv3x3_forward(x, w, split=split)
        assert torch.equal(~torch.isfinite(y).cpu(), ref), "split=%s" % split


def test_backward_after_inplace_filter_update_raises():
    """forward A; optimizer.step(); forward B; backward A: the packed filter image of A has been overwritten by B's refresh.  The filter
    itself is saved for backward too, so autograd's version check refuses the stale backward."""
    from shapeclipper_amd.model import resnet
    torch.manual_seed(3)
    net = resnet.build("resnet18").cuda().train()
    x = torch.randn(2, 3, 224, 224, device="cuda")
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    ya = net(x).square().mean()
    net(x).square().mean().backward()
    opt.step()
    net(x)                                   # forward B: refreshes the pack set from the updated filters
    with pytest.raises(RuntimeError, match="inplace"):
        ya.backward()


@pytest.mark.parametrize("batch", [1, 3, 64])
def test_conv_stem_forward_and_weight_gradient(batch):
    """ResNet.conv1 (7x7 / 2, 3 -> 64) on csrc/conv_stem.hip against float64 and against torch / MIOpen."""
    from shapeclipper_amd import ops
    torch.manual_seed(batch)
    dev = torch.device("cuda:0")
    x = torch.randn(batch, 3, 224, 224, device=dev)
    w = torch.randn(64, 3, 7, 7, device=dev) * 0.1
    assert ops.conv_stem_supported(x.shape, w.shape)
    nref = min(batch, 4)
    y = ops.conv_stem_forward(x, w)
    y64 = torch.nn.functional.conv2d(x[:nref].double(), w.double(), None, 2, 3)
    y_t = torch.nn.functional.conv2d(x[:nref], w, None, 2, 3)
    e_f, e_ft = _rel(y[:nref].double(), y64), _rel(y_t.double(), y64)
    gy = torch.randn_like(y)
    dw = ops.conv_stem_backward_weight(gy, x)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), 2, 3)
    dw_t = torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    e_w, e_wt = _rel(dw.double(), dw64), _rel(dw_t.double(), dw64)
    print("conv stem B=%d: forward %.2e of max (torch/MIOpen %.2e), weight gradient %.2e (torch/MIOpen %.2e)" % (batch, e_f, e_ft, e_w, e_wt))
    assert e_f < 2e-5 and e_w < 2e-5
    assert torch.equal(dw, ops.conv_stem_backward_weight(gy, x))


@pytest.mark.parametrize("hin,cin,cout,batch", [(56, 64, 128, 3), (28, 128, 256, 5), (14, 256, 512, 7), (14, 256, 512, 1), (56, 64, 128, 64), (6, 64, 64, 2)])
def test_conv1x1_stride2_forward_backward(hin, cin, cout, batch):
    """BasicBlock.downsample[0] on csrc/conv1x1s2.hip against float64 (and torch / MIOpen beside it)."""
    from shapeclipper_amd import ops
    torch.manual_seed(hin + cin + batch)
    dev = torch.device("cuda:0")
    x = torch.randn(batch, cin, hin, hin, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) * (1.0 / cin) ** 0.5
    assert ops.conv1x1s2_supported(x.shape, w.shape)
    y = ops.conv1x1s2_forward(x, w)
    y64 = torch.nn.functional.conv2d(x.double(), w.double(), None, 2, 0)
    gy = torch.randn_like(y)
    gx = ops.conv1x1s2_backward_data(gy, w)
    gx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), 2, 0)
    dw = ops.conv1x1s2_backward_weight(gy, x)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), 2, 0)
    e = (_rel(y.double(), y64), _rel(gx.double(), gx64), _rel(dw.double(), dw64))
    print("conv1x1/2 %dx%d %d>%d B=%d: forward %.2e, backward-data %.2e, backward-weight %.2e of max" % ((hin, hin, cin, cout, batch) + e))
    assert max(e) < 2e-5
    assert torch.equal(dw, ops.conv1x1s2_backward_weight(gy, x))


@pytest.mark.parametrize("hin,cin,cout,batch", [(56, 64, 128, 3), (28, 128, 256, 5), (14, 256, 512, 7), (14, 256, 512, 1), (56, 64, 128, 33), (28, 64, 72, 2)])
def test_conv3x3_stride2_forward(hin, cin, cout, batch):
    """BasicBlock.conv1 of layer2-4 (3x3 / stride 2 / pad 1): the stride-2 instance of csrc/conv3x3.hip against float64."""
    from shapeclipper_amd import ops
    torch.manual_seed(hin + cin + batch)
    dev = torch.device("cuda:0")
    x = torch.randn(batch, cin, hin, hin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    assert ops.conv3x3s2_supported(x.shape, w.shape)
    y = ops.conv3x3s2_forward(x, w)
    y64 = torch.nn.functional.conv2d(x.double(), w.double(), None, 2, 1)
    e, e_t = _rel(y.double(), y64), _rel(torch.nn.functional.conv2d(x, w, None, 2, 1).double(), y64)
    print("conv3x3/2 %dx%d %d>%d B=%d: forward %.2e of max (torch/MIOpen %.2e)" % (hin, hin, cin, cout, batch, e, e_t))
    assert y.shape == y64.shape and e < 2e-5


@pytest.mark.parametrize("hin,cin,cout,batch", [(56, 64, 128, 3), (28, 128, 256, 5), (14, 256, 512, 7), (14, 256, 512, 1), (56, 64, 128, 33), (28, 64, 64, 2),
                                                (14, 256, 512, 64), (56, 64, 128, 96)])
def test_conv3x3_stride2_gradients(hin, cin, cout, batch):
    """The two gradients of BasicBlock.conv1 of layer2-4 (3x3 / stride 2 / pad 1; torchvision layer{2,3,4}.0.conv1): backward-data as four
    parity sub-convolutions in the exact bf16x3 split arithmetic (csrc/conv3x3.hip, BD2), backward-weight with stride-2 patch addressing
    (csrc/conv3x3_wgrad.hip), both against float64 with torch / MIOpen beside them.  Bar: 2e-5 of the output scale, as for every
    convolution kernel; the weight gradient has a fixed summation order (two runs bit-identical)."""
    from shapeclipper_amd import ops
    torch.manual_seed(hin + cin + batch)
    dev = torch.device("cuda:0")
    x = torch.randn(batch, cin, hin, hin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    gy = torch.randn(batch, cout, hin // 2, hin // 2, device=dev)
    assert ops.conv3x3s2_grads_supported(x.shape, w.shape)
    nref = min(batch, 8)
    gx = ops.conv3x3s2_backward_data(gy, w, hin)
    gx64 = torch.nn.grad.conv2d_input((nref, cin, hin, hin), w.double(), gy[:nref].double(), 2, 1)
    dw = ops.conv3x3s2_backward_weight(gy, x)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), 2, 1)
    gx_t, dw_t, _ = torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, True, False])
    e = (_rel(gx[:nref].double(), gx64), _rel(dw.double(), dw64), _rel(gx_t[:nref].double(), gx64), _rel(dw_t.double(), dw64))
    print("conv3x3/2 gradients %dx%d %d>%d B=%d: backward-data %.2e, backward-weight %.2e of max (torch/MIOpen %.2e, %.2e)"
          % ((hin, hin, cin, cout, batch) + e))
    assert gx.shape == x.shape and dw.shape == w.shape
    assert e[0] < 2e-5 and e[1] < 2e-5
    if batch > nref:        # the images beyond the float64 sample against torch's fp32 operator
        assert _rel(gx.double(), gx_t.double()) < 2e-5
    assert torch.equal(dw, ops.conv3x3s2_backward_weight(gy, x)) and torch.equal(gx, ops.conv3x3s2_backward_data(gy, w, hin))


@pytest.mark.parametrize("kind,cin,cout,side,k,stride,pad", [("conv3x3", 64, 64, 56, 3, 1, 1), ("conv3x3", 256, 256, 14, 3, 1, 1),
                                                               ("conv3x3s2", 64, 128, 56, 3, 2, 1), ("conv3x3s2", 256, 512, 14, 3, 2, 1),
                                                               ("conv1x1s2", 128, 256, 28, 1, 2, 0), ("conv_stem", 3, 64, 224, 7, 2, 3)])
def test_autograd_wrappers_match_torch(kind, cin, cout, side, k, stride, pad):
    """functional.conv3x3 / conv3x3s2 / conv1x1s2 / conv_stem on an nn.Conv2d: output, input gradient and weight gradient against the
    same module through torch's own operator (wiring of the autograd Functions: argument order, which gradients are requested)."""
    from shapeclipper_amd import functional as F_
    torch.manual_seed(cin + side)
    conv = torch.nn.Conv2d(cin, cout, k, stride, pad, bias=False).cuda()
    x = torch.randn(4, cin, side, side, device="cuda")
    res = []
    for hip in (False, True):
        conv.weight.grad = None
        xi = x.clone().requires_grad_(True)
        y = getattr(F_, kind)(conv, xi) if hip else conv(xi)
        g = torch.Generator(device="cpu").manual_seed(1)
        (y * torch.randn(y.shape, generator=g).cuda()).sum().backward()
        res.append((y.detach(), xi.grad.clone(), conv.weight.grad.clone()))
    for a, b, name in zip(res[1], res[0], ("output", "input gradient", "weight gradient")):
        e = _rel(a.double(), b.double())
        print("%s %d>%d %dx%d: %s differs from torch's by %.2e of max" % (kind, cin, cout, side, side, name, e))
        assert e < 2e-5, name
