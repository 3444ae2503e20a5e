"""The CLIP tower as S independent chains on S streams (batch split S ways) against one chain: python tools/proto_streams_clip.py [batch=32]
At ViT-B/32, batch 32 the tower is a chain of ~90 dependent launches (0.65 ms of the 1.0 ms is the chain's latency, tools/proto_graph_clip.py at
batch 8): do independent chains fill each other's gaps?"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model.clip_vit import ClipVisionTower, VIT_B32, VIT_L14
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "B/32"
t = ClipVisionTower(**(VIT_L14 if name == "L/14" else VIT_B32)).cuda()
x = torch.randn(B, 3, 224, 224, device="cuda")
ref = t.encode_image(x).clone()


def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3


print("ViT-%s B=%d one chain: %.3f ms" % (name, B, timed(lambda: t.encode_image(x))))
for S in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    chunks = list(x.chunk(S))
    outs = [None] * S

    def run():
        main = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs[i] = t.encode_image(chunks[i])
        for s in streams:
            main.wait_stream(s)
    ms = timed(run)
    y = torch.cat(outs)
    print("  %d chains: %.3f ms   max |diff| vs one chain %.3g" % (S, ms, float((y - ref).abs().max())))
