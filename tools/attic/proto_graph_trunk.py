"""Prototype: capture a ResNet trunk's forward AND backward as two hipGraphs (torch.cuda.make_graphed_callables) and compare
outputs / gradients / host time with the eager path.   python tools/proto_graph_trunk.py [batch]"""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Fixed(torch.nn.Module):
    def __init__(self, net, groups):
        super().__init__()
        self.net, self.groups = net, groups

    def forward(self, x):
        return self.net(x, groups=self.groups)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    runner, opt, batch = bench.build_runner(B)
    enc = runner.graph.module.encoder
    ref = copy.deepcopy(enc)
    G = 2
    x = torch.randn(G * B, 3, 224, 224, device="cuda")
    w = torch.randn(G * B, enc.fc.out_features, device="cuda")

    def run(mod, x):
        for p in mod.parameters():
            p.grad = None
        y = mod(x)
        (y * w).sum().backward()
        return y.detach().clone(), [p.grad.clone() for p in mod.parameters()]

    wrapped = Fixed(enc, G)
    graphed = torch.cuda.make_graphed_callables(wrapped, (x.clone(),), num_warmup_iters=3)
    # the warm-up / capture passes moved BatchNorm's running statistics of `enc`; bring the eager copy to the same state
    ref.load_state_dict(enc.state_dict())
    y0, g0 = run(Fixed(ref, G), x)
    y1, g1 = run(graphed, x)
    print("output equal:", torch.equal(y0, y1), float((y0 - y1).abs().max()))
    bad = [i for i, (a, b) in enumerate(zip(g0, g1)) if not torch.equal(a, b)]
    print("grads differing:", len(bad), "of", len(g0), "max abs", max([float((g0[i] - g1[i]).abs().max()) for i in bad] or [0.0]))
    sd0, sd1 = ref.state_dict(), enc.state_dict()
    print("buffers differing:", [k for k in sd0 if not torch.equal(sd0[k], sd1[k])][:5])
    for name, mod in (("eager", Fixed(ref, G)), ("graphed", graphed)):
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                y = mod(x)
                (y * w).sum().backward()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print("%s: host %.2f ms / pass, wall %.2f ms / pass" % (name, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))


if __name__ == "__main__":
    main()
