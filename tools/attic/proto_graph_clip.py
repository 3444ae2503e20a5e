"""hipGraph replay of the CLIP image tower against stream launches: python tools/proto_graph_clip.py [batch=32] [B/32 | L/14].
The tower is ~90 dependent launches of 5-19 us at ViT-B/32, batch 32 (docs/LAB_NOTEBOOK.md section 7): does a captured graph close the gaps between them?"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model.clip_vit import ClipVisionTower, VIT_B32, VIT_L14
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "B/32"
t = ClipVisionTower(**(VIT_L14 if name == "L/14" else VIT_B32)).cuda()
x = torch.randn(B, 3, 224, 224, device="cuda")


def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3


eager = t.encode_image(x).clone()
ms_eager = timed(lambda: t.encode_image(x))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): t.encode_image(x)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = t.encode_image(x)
g.replay(); torch.cuda.synchronize()
print("ViT-%s B=%d: stream launches %.3f ms, graph replay %.3f ms, identical output: %s" % (name, B, ms_eager, timed(g.replay), torch.equal(y, eager)))
