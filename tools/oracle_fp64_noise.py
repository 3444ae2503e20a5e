"""How far is the fp32 ORACLE from exact arithmetic on the bs16 full-step parity case?  (justifies the bars of
tests/test_gpu_full_step_parity.py; host-only, ~2 minutes on 8 cores)

    SC_FULLSTEP_DUMP=gpurun_out/fullstep.pt python -m pytest tests/test_gpu_full_step_parity.py -s      # on the GPU box: dumps the case
    python tools/oracle_fp64_noise.py gpurun_out/fullstep.pt                                             # anywhere

Runs oracle/reference_ops.py twice on the dumped inputs -- as it is (fp32) and through a dtype-generic copy of its source (its
`.float()` / `dtype=torch.float32` follow torch's default dtype; written to a temporary file, nothing changes in oracle/) in float64 --
and prints, per differentiated leaf, max |fp32 - fp64| / max |fp64| next to the product's distance from both.  Round-3 result
(profiles/r03_fullstep_fp64_noise.txt): pose 7.4e-4, pose_NN 2.4e-4, z_rgb 6.4e-5, z_sdf 1.5e-5 for the fp32 oracle; the product
is as close or closer to float64 on every leaf (the x-mirror symmetry `abs(x0)` makes d loss / d R[2,0] = -t_z * sum of d loss / d x
over a whole image a sum of 32,768 terms that almost cancel)."""
import importlib.util
import os
import sys
import tempfile

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def load_generic():
    src = open(os.path.join(ROOT, "oracle", "reference_ops.py")).read()
    src = src.replace("torch.cat([R.float(), t.float()[..., None]], dim=-1)", "torch.cat([R, t[..., None]], dim=-1)")
    src = src.replace("dtype=torch.float32", "dtype=torch.get_default_dtype()").replace(".float()", ".to(torch.get_default_dtype())")
    f = tempfile.NamedTemporaryFile("w", suffix="_ref_generic.py", delete=False)
    f.write(src); f.close()
    spec = importlib.util.spec_from_file_location("ref_generic", f.name)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_generic"] = mod          # dataclasses looks the module up
    spec.loader.exec_module(mod)
    os.unlink(f.name)
    return mod


def main(path):
    R = load_generic()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    d = torch.load(path)
    B, Rr = d["leaves"]["z_sdf"].shape[0], d["main"][2].shape[1]
    out = {}
    for dt in (torch.float32, torch.float64):
        torch.set_default_dtype(dt)
        c = lambda t: t.to(dt).clone()
        cfg = R.Cfg()
        Ws = {k: c(v).requires_grad_(True) for k, v in d["Ws"].items()}
        Wr = {k: c(v).requires_grad_(True) for k, v in d["Wr"].items()}
        beta = c(d["beta"]).requires_grad_(True)
        lv = {k: c(v).requires_grad_(True) for k, v in d["leaves"].items()}
        w = d["weights"]
        torch.set_rng_state(d["state"])

        def one(pose, pack, z_rgb, names, with_eik):
            intr, sd, ray_idx, rgb_t, mask_t, normal_t, mask_prod = pack
            torch.set_default_dtype(torch.float32)          # the draws are float32 draws of the CPU generator in both passes
            t_rand, eik_idx, eik_pts = R.draw_render_randoms(B * Rr, 64, True)
            torch.set_default_dtype(dt)
            o = R.render(cfg, Ws, Wr, beta, pose, c(intr), c(sd), lv["z_sdf"], z_rgb, ray_idx.long(), True, c(t_rand), eik_idx, c(eik_pts))
            valid = (mask_t > 0.5) & (mask_prod > 0.5)
            tgt = R.transform_normal(c(normal_t), pose)
            L = {names[0]: R.mse_loss(o["rgb"], c(rgb_t)), names[1]: R.mask_loss(cfg, o["mask"], c(mask_t)),
                 names[2]: R.normal_loss(cfg, o["normal"], tgt, valid, tolerance=0.2)}
            if with_eik:
                L["eikonal"] = R.mse_loss(o["grad_eikonal"].view(B, -1), 1)
            sum(w[k] * v for k, v in L.items()).backward()
        one(lv["pose"], d["main"], lv["z_rgb"], ("render", "mask", "normal"), True)
        one(lv["pose_NN"], d["nn"], lv["z_rgb_NN"], ("nearest_img", "nearest_mask", "nearest_normal"), False)
        out[dt] = {k: v.grad.double() for k, v in lv.items()}
    torch.set_default_dtype(torch.float32)
    print("max |a - b| / max |fp64|, gradients of loss.all at bs%d (both renders):" % B)
    for k in ("z_sdf", "z_rgb", "z_rgb_NN", "pose", "pose_NN"):
        r64, r32, got = out[torch.float64][k], out[torch.float32][k], d["got"][k].double()
        m = r64.abs().max()
        print("  %-9s fp32 oracle vs fp64 oracle %.2e | product vs fp64 oracle %.2e | product vs fp32 oracle %.2e"
              % (k, float((r32 - r64).abs().max() / m), float((got - r64).abs().max() / m), float((got - r32).abs().max() / m)))


if __name__ == "__main__":
    main(sys.argv[1])
