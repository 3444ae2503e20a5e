"""Same-box A/B of the CLIP image tower between two builds of the C ABI (the driver's B=256 / ViT-L/14 lines read 3.354 ->
3.579 ms and 7.754 -> 8.050 ms from round 4 to round 5 on two different boxes): the round-4 library (shapeclipper_amd/lib/variants/lib_r04.so,
built from commit 9c0e07c) against the tree's, through the entry point both export (sc_clip_vit_forward_f16: the launch-per-operation
tower), alternating, same weights, same input.   python tools/r6_clip_ab.py
(the variant library is not kept in the tree: `git worktree add /tmp/r04 9c0e07c && make -C /tmp/r04/shapeclipper_amd/csrc && mkdir -p
shapeclipper_amd/lib/variants && cp /tmp/r04/shapeclipper_amd/lib/libshapeclipper_hip.so shapeclipper_amd/lib/variants/lib_r04.so`)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from shapeclipper_amd.model.clip_vit import ClipVisionTower, VIT_B32, VIT_L14
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
libs = {"r04": ctypes.CDLL(os.path.join(ROOT, "shapeclipper_amd/lib/variants/lib_r04.so")),
        "r06": ctypes.CDLL(os.path.join(ROOT, "shapeclipper_amd/lib/libshapeclipper_hip.so"))}
P = lambda t: ctypes.c_void_p(t.data_ptr())
for name, cfg, B in (("B/32", VIT_B32, 256), ("L/14", VIT_L14, 32), ("B/32", VIT_B32, 32)):
    torch.manual_seed(0)
    t = ClipVisionTower(**cfg).cuda()
    x = torch.randn(B, 3, 224, 224, device="cuda")
    t.encode_image(x)
    w16, wf, _ = t._packed
    c = t.cfg
    nbytes = t.workspace_bytes(B)
    ws = torch.empty(nbytes, device="cuda", dtype=torch.uint8)
    outs = {}
    def run(lib, out):
        lib.sc_clip_vit_workspace_bytes.restype = ctypes.c_longlong
        rc = lib.sc_clip_vit_forward_f16(P(x), B, c["channels"], c["image_size"], c["image_size"], c["patch"], c["width"], c["mlp"], c["layers"], c["heads"],
                                         c["proj"], P(w16), P(wf), ctypes.c_float(1e-5), P(out), P(ws), ctypes.c_longlong(nbytes),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    for rep in range(3):
        for k, lib in libs.items():
            out = torch.empty(B, c["proj"], device="cuda")
            for _ in range(3): run(lib, out)
            torch.cuda.synchronize(); t0 = time.time()
            n = 20
            for _ in range(n): run(lib, out)
            torch.cuda.synchronize(); dt = (time.time() - t0) / n
            outs[k] = out
            print("ViT-%s B=%d %s rep %d: %.3f ms" % (name, B, k, rep, dt * 1e3), flush=True)
    print("   same values: %s (max |diff| %.3g)" % (bool(torch.equal(outs["r04"], outs["r06"])), float((outs["r04"] - outs["r06"]).abs().max())))
