"""Fused loss kernel (csrc/loss.hip) vs golden G7 captured from the reference's model/loss.py: values and
the gradients of  MSE + 0.5*mask + 0.01*normal(tol 0.2) + 0.03*MSE(eik,1).  Bar 1e-5 abs on values, 1e-6 on grads."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_losses_golden(golden):
    from shapeclipper_amd.functional import FusedRenderLoss
    g = golden("g7_losses")
    dev = torch.device("cuda:0")
    t = lambda k: torch.tensor(g[k], device=dev)
    rgb, pm, npred, eik = (t(k).requires_grad_(True) for k in ("pred3", "pm", "npred", "eik"))
    ngt = t("ngt").requires_grad_(True)      # transform_normal(input normal, predicted pose): differentiable target
    out = FusedRenderLoss.apply(rgb, t("tgt3"), pm, t("tm"), npred, ngt, eik, 5.0, 0.0, 1 - 0.2)
    vals = torch.stack([o.detach() for o in out]).cpu().numpy()
    assert abs(vals[0] - g["val.mse"]) < 1e-6 and abs(vals[1] - g["val.mask"]) < 1e-6
    assert abs(vals[2] - g["val.normal"]) < 1e-5 and abs(vals[3] - g["val.mse_eik"]) < 1e-6
    (out[0] + 0.5 * out[1] + 0.01 * out[2] + 0.03 * out[3]).backward()
    torch.cuda.synchronize()
    for got, key in ((rgb.grad, "g_pred3"), (pm.grad, "g_pm"), (npred.grad, "g_npred"), (eik.grad, "g_eik"),
                     (ngt.grad, "g_ngt")):
        err = np.abs(got.cpu().numpy() - g[key]).max()
        print("G7 %-8s max abs err %.3e (max |ref| %.3e)" % (key, err, np.abs(g[key]).max()))
        np.testing.assert_allclose(got.cpu().numpy(), g[key], atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("tol", [0.0, 0.2, 0.5])
def test_fused_losses_vs_oracle_with_exact_ties(tol):
    from oracle import reference_ops as R
    from shapeclipper_amd.functional import FusedRenderLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    B, Rr = 3, 100
    rgb, tgt = torch.rand(B, Rr, 3), torch.rand(B, Rr, 3)
    pm, tm = torch.rand(B, Rr, 1), (torch.rand(B, Rr, 1) > 0.4).float()
    npred = torch.nn.functional.normalize(torch.randn(B, Rr, 3), dim=-1)
    ngt = torch.nn.functional.normalize(torch.randn(B, Rr, 3), dim=-1)
    npred[0, :40] = npred[0, 0]; ngt[0, :40] = ngt[0, 0]          # 40 exact duplicates of the angular error
    cfg = R.Cfg(mask_mse=0.3)
    mask = (tm > 0.5) & (pm > 0.5)
    ref = (R.mse_loss(rgb, tgt), R.mask_loss(cfg, pm, tm), R.normal_loss(cfg, npred, ngt, mask, tolerance=tol))
    out = FusedRenderLoss.apply(rgb.to(dev), tgt.to(dev), pm.to(dev), tm.to(dev), npred.to(dev), ngt.to(dev), None, 5.0, 0.3, 1 - tol)
    for i in range(3):
        assert abs(out[i].item() - ref[i].item()) < 2e-5, (i, out[i].item(), ref[i].item())
    assert out[3].item() == 0.0


def test_empty_mask_gives_nan_like_torch():
    from shapeclipper_amd.functional import FusedRenderLoss
    dev = torch.device("cuda:0")
    z = torch.zeros(2, 16, 3, device=dev)
    m = torch.zeros(2, 16, 1, device=dev)
    out = FusedRenderLoss.apply(z, z, m, m, z, z, None, 5.0, 0.0, 0.8)
    assert torch.isnan(out[2]) and out[0].item() == 0.0


@pytest.mark.parametrize("B,Rr", [(32, 512), (40, 512), (33, 500)])
def test_robust_normal_selection_at_and_past_the_register_resident_size(B, Rr):
    """Round 5: the selection keeps the first 16 elements of each of its 1,024 threads in registers (all 16,384 at the training size) and
    reads the rest from the workspace: the training size, a larger batch that uses both paths, and a ragged size -- values and the
    gradient of the normal loss against the oracle, with exact ties across the register / workspace boundary."""
    from oracle import reference_ops as R
    from shapeclipper_amd.functional import FusedRenderLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(B + Rr)
    rgb, tgt = torch.rand(B, Rr, 3), torch.rand(B, Rr, 3)
    pm, tm = torch.rand(B, Rr, 1), (torch.rand(B, Rr, 1) > 0.4).float()
    npred = torch.nn.functional.normalize(torch.randn(B, Rr, 3), dim=-1)
    ngt = torch.nn.functional.normalize(torch.randn(B, Rr, 3), dim=-1)
    npred[0, :60] = npred[0, 0]; ngt[0, :60] = ngt[0, 0]                      # exact duplicates at the start ...
    npred[-1, -60:] = npred[0, 0]; ngt[-1, -60:] = ngt[0, 0]                  # ... and at the very end (past element 16,384 when B*R is larger)
    cfg = R.Cfg()
    mask = (tm > 0.5) & (pm > 0.5)
    np_c = npred.clone().requires_grad_(True)
    ref = R.normal_loss(cfg, np_c, ngt, mask, tolerance=0.2)
    ref.backward()
    np_d = npred.to(dev).requires_grad_(True)
    out = FusedRenderLoss.apply(rgb.to(dev), tgt.to(dev), pm.to(dev), tm.to(dev), np_d, ngt.to(dev), None, 5.0, 0.0, 0.8)
    assert abs(out[2].item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
    out[2].backward()
    # ties are exact duplicates: which of them are kept does not change the value, but it does change WHICH rows get a gradient; the
    # reference keeps the first ones of a stable sort, i.e. the lowest indices -- as the kernel does
    assert (np_d.grad.cpu() - np_c.grad).abs().max() < 1e-6
