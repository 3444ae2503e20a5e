"""Per-layer table of the ResNet-18/34 convolutions of one bs32 training step: every distinct
(Cin, Cout, H, kernel, stride) shape at the batch the step runs it with (ResNet-34 encoder: 2 passes x 32 = 64 images in one grouped
pass, ResNet-18 view estimator: 3 x 32 = 96), forward / backward-data / backward-weight timed separately through torch (= MIOpen's pick), with the
effective TFLOP/s against the 157.3 TFLOP/s fp32 MFMA peak and the share of the step each shape carries.

    python tools/perf_conv.py [--hip]      --hip: time shapeclipper_amd's own convolution kernels beside MIOpen
"""
import argparse
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK = 157.3e12


def shapes(layers, batch):
    """(name, Cin, Cout, H_in, k, stride, count) for a torchvision-style BasicBlock ResNet at 224x224."""
    out = [("stem", 3, 64, 224, 7, 2, 1)]
    inp, h = 64, 56
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
        stride = 1 if li == 0 else 2
        same = 0
        if stride != 1:
            out.append(("l%d.0.conv1" % (li + 1), inp, planes, h, 3, 2, 1))
            out.append(("l%d.0.down" % (li + 1), inp, planes, h, 1, 2, 1))
            h //= 2
            same = 2 * n - 1
        else:
            same = 2 * n
        out.append(("l%d.3x3" % (li + 1), planes, planes, h, 3, 1, same))
        inp = planes
    return [(n, ci, co, hh, k, s, c, batch) for (n, ci, co, hh, k, s, c) in out]


def time_fn(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hip", action="store_true")
    ap.add_argument("--split", action="store_true", help="also time the bf16-split forward kernel")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    rows = []
    total = {"fwd": 0.0, "bwd_data": 0.0, "bwd_weight": 0.0}
    hip_conv = None
    if args.hip:
        from shapeclipper_amd import ops
        hip_conv = ops
    print("%-12s %-22s %5s | %8s %6s | %8s %6s | %8s %6s | %7s" % ("net", "layer Cin>Cout HxH k/s", "count", "fwd ms", "TF/s", "bwdD ms", "TF/s", "bwdW ms", "TF/s", "ms/step"))
    for net, layers, batch in (("resnet18 B=96", [2, 2, 2, 2], 96), ("resnet34 B=64", [3, 4, 6, 3], 64)):
        for (name, ci, co, h, k, s, count, b) in shapes(layers, batch):
            x = torch.randn(b, ci, h, h, device=dev, requires_grad=True)
            w = torch.randn(co, ci, k, k, device=dev, requires_grad=True) * 0.05
            pad = k // 2
            y = torch.nn.functional.conv2d(x, w, None, s, pad)
            gy = torch.randn_like(y)
            flop = 2.0 * y.numel() * ci * k * k
            t_f = time_fn(lambda: torch.nn.functional.conv2d(x, w, None, s, pad))
            t_d = time_fn(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]))
            t_w = time_fn(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))
            need_d = name != "stem"
            per_step = count * (t_f + (t_d if need_d else 0.0) + t_w)
            total["fwd"] += count * t_f
            total["bwd_data"] += count * t_d if need_d else 0.0
            total["bwd_weight"] += count * t_w
            line = "%-12s %-22s %5d | %8.3f %6.1f | %8.3f %6.1f | %8.3f %6.1f | %7.2f" % (
                net, "%s %d>%d %d k%d/s%d" % (name, ci, co, h, k, s), count, t_f, flop / t_f / 1e9, t_d, flop / t_d / 1e9, t_w, flop / t_w / 1e9, per_step)
            if hip_conv is not None and name == "stem":
                xd, wd = x.detach(), w.detach()
                t_hf = time_fn(lambda: hip_conv.conv_stem_forward(xd, wd))
                t_hw = time_fn(lambda: hip_conv.conv_stem_backward_weight(gy, xd))
                line += " | hip fwd %7.3f (%5.1f TF/s) bwdW %7.3f (%5.1f)" % (t_hf, flop / t_hf / 1e9, t_hw, flop / t_hw / 1e9)
            if hip_conv is not None and k == 3 and s == 2:
                xd, wd = x.detach(), w.detach()
                t_hf = time_fn(lambda: hip_conv.conv3x3s2_forward(xd, wd))
                line += " | hip fwd (incl. pack) %7.3f (%5.1f TF/s)" % (t_hf, flop / t_hf / 1e9)
            if hip_conv is not None and k == 1 and s == 2:
                xd, wd = x.detach(), w.detach()
                t_hf = time_fn(lambda: hip_conv.conv1x1s2_forward(xd, wd))
                t_hd = time_fn(lambda: hip_conv.conv1x1s2_backward_data(gy, wd))
                t_hw = time_fn(lambda: hip_conv.conv1x1s2_backward_weight(gy, xd))
                line += " | hip fwd %7.3f bwdD %7.3f bwdW %7.3f" % (t_hf, t_hd, t_hw)
            if hip_conv is not None and k == 3 and s == 1:
                xd, wd = x.detach(), w.detach()
                wp_f, wp_b = hip_conv.conv3x3_pack(wd, h), hip_conv.conv3x3_pack(wd, h, True)
                t_pk = time_fn(lambda: hip_conv.conv3x3_pack(wd, h))
                t_hf = time_fn(lambda: hip_conv.conv3x3_apply(xd, wp_f, co))
                t_hd = time_fn(lambda: hip_conv.conv3x3_apply(gy, wp_b, ci))
                line += " | hip pack %6.3f fwd %7.3f (%5.1f TF/s) bwdD %7.3f (%5.1f)" % (t_pk, t_hf, flop / t_hf / 1e9, t_hd, flop / t_hd / 1e9)
                if args.split:
                    ws_f = hip_conv.conv3x3_pack(wd, h, False, True)
                    t_sf = time_fn(lambda: hip_conv.conv3x3_apply(xd, ws_f, co, True))
                    line += " split fwd %7.3f (%5.1f)" % (t_sf, flop / t_sf / 1e9)
                if hip_conv.conv3x3_wgrad_supported(x.shape, w.shape):
                    t_hw = time_fn(lambda: hip_conv.conv3x3_backward_weight(gy, xd))
                    line += " bwdW %7.3f (%5.1f)" % (t_hw, flop / t_hw / 1e9)
            print(line, flush=True)
            del x, w, y, gy
    print("per step (one grouped encoder pass + one grouped estimator pass): fwd %.2f ms, backward-data %.2f ms, backward-weight %.2f ms, sum %.2f ms"
          % (total["fwd"], total["bwd_data"], total["bwd_weight"], sum(total.values())))


if __name__ == "__main__":
    main()
