"""The CLIP tower as S independent half-batch chains on S streams INSIDE one captured hipGraph (no host pacing), against one chain:
python tools/proto_graph_streams_clip.py [batch=32] [B/32|L/14].  tools/proto_streams_clip.py (eager, host-paced) found 2 chains slower."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model.clip_vit import ClipVisionTower, VIT_B32, VIT_L14
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "B/32"
t = ClipVisionTower(**(VIT_L14 if name == "L/14" else VIT_B32)).cuda()
x = torch.randn(B, 3, 224, 224, device="cuda")
ref = t.encode_image(x).clone()


def timed(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3


print("ViT-%s B=%d one chain, eager: %.3f ms" % (name, B, timed(lambda: t.encode_image(x))))
for S in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    chunks = list(x.chunk(S))
    outs = [None] * S
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for c in chunks: t.encode_image(c)              # warm-up on the capture stream (allocator, attribute calls)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            main = torch.cuda.current_stream()
            for i, s in enumerate(streams):
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    outs[i] = t.encode_image(chunks[i])
            for s in streams:
                main.wait_stream(s)
    torch.cuda.current_stream().wait_stream(cap)
    ms = timed(g.replay)
    y = torch.cat(outs)
    print("  graph, %d chain(s): %.3f ms   max |diff| vs eager one chain %.3g" % (S, ms, float((y - ref).abs().max())))
