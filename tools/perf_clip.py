"""CLIP ViT-B/32 image tower throughput at batch 32 (BASELINE config[2])."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model.clip_vit import ClipVisionTower, VIT_B32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t = ClipVisionTower(**VIT_B32).cuda()
x = torch.randn(B, 3, 224, 224, device="cuda")
for _ in range(3): t.encode_image(x)
torch.cuda.synchronize(); t0 = time.time()
n = 20
for _ in range(n): t.encode_image(x)
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print("ViT-B/32 B=%d: %.3f ms/batch  %.0f img/s  %.1f TFLOP/s (8.725 GFLOP/img)" % (B, dt * 1e3, B / dt, B * 8.725e9 / dt / 1e12))
