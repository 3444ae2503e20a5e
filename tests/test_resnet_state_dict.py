"""The ResNet-18 / ResNet-34 trunks against torchvision's published layout (the reference builds `torchvision.models.resnet34(pretrained=...)`
and `resnet18(pretrained=True)`: model/graph.py:50-55, model/view_estimator.py:40-42; torchvision itself is absent here).
A state dict with torchvision's key names and shapes -- generated below from the architecture's published rules, NOT from resnet.py -- must
load strictly, and the loaded trunk must compute what a plain functional restatement of torchvision's forward computes from those keys."""
import pytest
import torch
import torch.nn.functional as F

from shapeclipper_amd.model import resnet

LAYERS = {"resnet18": [2, 2, 2, 2], "resnet34": [3, 4, 6, 3]}


def torchvision_layout(name):
    """(key, shape) of torchvision.models.<name>().state_dict(), in its order (BasicBlock nets, 1000 classes)."""
    out = []

    def bn(prefix, c):
        out.extend([(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,)),
                    (prefix + ".num_batches_tracked", ())])
    out.append(("conv1.weight", (64, 3, 7, 7)))
    bn("bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), LAYERS[name]), start=1):
        for b in range(blocks):
            stride = 2 if (b == 0 and li > 1) else 1
            p = "layer%d.%d" % (li, b)
            out.append((p + ".conv1.weight", (planes, inplanes, 3, 3)))
            bn(p + ".bn1", planes)
            out.append((p + ".conv2.weight", (planes, planes, 3, 3)))
            bn(p + ".bn2", planes)
            if stride != 1 or inplanes != planes:
                out.append((p + ".downsample.0.weight", (planes, inplanes, 1, 1)))
                bn(p + ".downsample.1", planes)
            inplanes = planes
    out.extend([("fc.weight", (1000, 512)), ("fc.bias", (1000,))])
    return out


def random_state(name, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in torchvision_layout(name):
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(7, dtype=torch.long)
        elif k.endswith("running_var"):
            sd[k] = torch.rand(shape, generator=g) + 0.5
        elif len(shape) == 4:
            sd[k] = torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
        else:
            sd[k] = torch.randn(shape, generator=g) * 0.1 + (1.0 if k.endswith("bn1.weight") or k.endswith("bn2.weight") or k.endswith(".1.weight") else 0.0)
    return sd


def functional_forward(name, sd, x):
    """torchvision ResNet.forward in evaluation mode, written out over the state dict."""
    def bn(t, p):
        return F.batch_norm(t, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.1, 1e-5)
    t = F.max_pool2d(F.relu(bn(F.conv2d(x, sd["conv1.weight"], None, 2, 3), "bn1")), 3, 2, 1)
    for li, blocks in enumerate(LAYERS[name], start=1):
        for b in range(blocks):
            p = "layer%d.%d" % (li, b)
            stride = 2 if (b == 0 and li > 1) else 1
            idt = t
            if p + ".downsample.0.weight" in sd:
                idt = bn(F.conv2d(t, sd[p + ".downsample.0.weight"], None, stride, 0), p + ".downsample.1")
            o = F.relu(bn(F.conv2d(t, sd[p + ".conv1.weight"], None, stride, 1), p + ".bn1"))
            o = bn(F.conv2d(o, sd[p + ".conv2.weight"], None, 1, 1), p + ".bn2")
            t = F.relu(o + idt)
    return F.linear(torch.flatten(F.adaptive_avg_pool2d(t, 1), 1), sd["fc.weight"], sd["fc.bias"])


@pytest.mark.parametrize("name", ["resnet18", "resnet34"])
def test_torchvision_state_dict_loads_strictly_and_means_the_same(name):
    net = resnet.build(name)
    layout = torchvision_layout(name)
    own = net.state_dict()
    assert list(own.keys()) == [k for k, _ in layout]                       # same keys, same order
    assert [tuple(v.shape) for v in own.values()] == [s for _, s in layout]
    n_params = sum(v.numel() for k, v in own.items() if "running" not in k and "num_batches" not in k)
    assert n_params == {"resnet18": 11689512, "resnet34": 21797672}[name]   # torchvision's published parameter counts
    sd = random_state(name, 3)
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    net.eval()
    with torch.no_grad():
        got, want = net(x), functional_forward(name, sd, x)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), float((got - want).abs().max())


def test_reference_head_replacement_keeps_the_trunk_keys():
    """model/graph.py:52-55 replaces `encoder.fc`, model/view_estimator.py:42 sets `feature_extractor.fc = nn.Identity()`: the trunk keys of a
    reference checkpoint are torchvision's minus / with a resized fc -- loading them must not depend on the head."""
    net = resnet.build("resnet18")
    net.fc = torch.nn.Identity()
    sd = {k: v for k, v in random_state("resnet18", 1).items() if not k.startswith("fc.")}
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
