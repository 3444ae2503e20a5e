"""Phase profile of the cluster form of the CLIP layers (library built with SC_CL_PROF=1: tools/build_variants.sh clip_vit.hip SC_CL_PROF 1,
SHAPECLIPPER_HIP_LIB=shapeclipper_amd/lib/variants/lib_SC_CL_PROF_1.so).  python tools/prof_clip_cluster.py [batch=32]"""
import ctypes, os, sys
os.environ.setdefault("SC_CLIP_CLUSTER_MAX_B", "64")
os.environ.setdefault("SC_CLIP_CLUSTER_MIN_B", "1")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd import _lib
from shapeclipper_amd.model.clip_vit import ClipVisionTower, VIT_B32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t = ClipVisionTower(**VIT_B32).cuda()
x = torch.randn(B, 3, 224, 224, device="cuda")
for _ in range(5): t.encode_image(x)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 256)()
assert lib.sc_clip_cluster_prof_read(buf) == 0
names = ["ln1+band", "qkv", "attention", "barrier 1", "proj", "barrier 2", "ln2+band", "fc1", "barrier 3", "fc2", "barrier 4"]
tot = [0.0] * 11
for l in range(12):
    st = [buf[l * 16 + i] for i in range(12)]
    d = [(st[i + 1] - st[i]) / 100.0 for i in range(11)]
    tot = [a + b for a, b in zip(tot, d)]
    print("layer %2d: " % l + "  ".join("%s %.1f" % (n, v) for n, v in zip(names, d)) + "   | sum %.1f us" % sum(d))
print("mean per layer (us): " + "  ".join("%s %.2f" % (n, v / 12) for n, v in zip(names, tot)) + "   | %.1f" % (sum(tot) / 12))
