"""CLIP ViT image tower throughput: python tools/perf_clip.py [batch=32] [B/32 | L/14] (BASELINE config[2]: B/32, batch 32)."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model.clip_vit import ClipVisionTower, VIT_B32, VIT_L14
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "B/32"
cfg, gflop = (VIT_L14, 155.5) if name == "L/14" else (VIT_B32, 8.725)
t = ClipVisionTower(**cfg).cuda()
x = torch.randn(B, 3, 224, 224, device="cuda")
for _ in range(3): t.encode_image(x)
torch.cuda.synchronize(); t0 = time.time()
n = 20
for _ in range(n): t.encode_image(x)
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print("ViT-%s B=%d: %.3f ms/batch  %.0f img/s  %.1f TFLOP/s (%.3f GFLOP/img)" % (name, B, dt * 1e3, B / dt, B * gflop * 1e9 / dt / 1e12, gflop))
