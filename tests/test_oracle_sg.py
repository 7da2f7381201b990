"""CPU: C oracle of sgutils (sg.cu:27-175) vs torch autograd.  PARITY UNPINNED (no reference test)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import cref, torch_ref
from scenes import rel_l2


def sg_inputs(N=2, D=500, L=7, seed=11, sigma_lo=0.05, sigma_hi=0.6):
    g = torch.Generator().manual_seed(seed)
    dirs = F.normalize(torch.randn(N, D, 3, generator=g), dim=-1)
    sig = sigma_lo + (sigma_hi - sigma_lo) * torch.rand(N, D, generator=g)
    lv = torch.rand(N, L, 3, generator=g)
    lp = F.normalize(torch.randn(N, L, 3, generator=g), dim=-1) * 1100.0
    pp = torch.randn(N, D, 3, generator=g) * 60.0
    nl = torch.tensor([L, max(1, L - 2)][:N], dtype=torch.int32)
    return dirs, sig, lv, lp, pp, nl


@pytest.mark.parametrize("w_type", [0, 1, 2, 3])
def test_sg_forward_backward_vs_autograd(w_type):
    dirs, sig, lv, lp, pp, nl = sg_inputs()
    out = cref.evaluate_gaussian_fwd(dirs, sig, lv, lp, pp, nl, w_type)
    td, ts, tl = dirs.clone().requires_grad_(True), sig.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    ref = torch_ref.evaluate_gaussian(td, ts, tl, lp, pp, nl, w_type)
    assert rel_l2(out, ref) < 1e-5
    g = torch.Generator().manual_seed(2)
    go = torch.randn(out.shape, generator=g)
    (ref * go).sum().backward()
    gd, gs, gl = cref.evaluate_gaussian_bwd(dirs, sig, lv, lp, pp, nl, go, w_type, want_light_grad=True)
    assert rel_l2(gd, td.grad) < 1e-4
    assert rel_l2(gs, ts.grad) < 1e-4
    assert rel_l2(gl, tl.grad) < 1e-4


def test_sg_edge_derivative_is_minus_20():
    # lobe exactly at the light: |cos| == 1 -> sg.cu:129 substitutes -20 for d acos/dc, but
    # dL/dangle is 0 there (angle == 0) so the direction gradient is exactly 0 and finite.
    dirs = torch.tensor([[[0.0, 0.0, 1.0]]])
    sig = torch.tensor([[0.2]])
    lv = torch.ones(1, 1, 3)
    lp = torch.tensor([[[0.0, 0.0, 10.0]]])
    pp = torch.zeros(1, 1, 3)
    nl = torch.tensor([1], dtype=torch.int32)
    gd, gs, _ = cref.evaluate_gaussian_bwd(dirs, sig, lv, lp, pp, nl, torch.ones(1, 1, 3), 0)
    assert torch.isfinite(gd).all() and float(gd.abs().max()) == 0.0
    assert torch.isfinite(gs).all()


def test_sg_respects_n_lights():
    dirs, sig, lv, lp, pp, nl = sg_inputs()
    nl2 = torch.tensor([3, 0], dtype=torch.int32)
    out = cref.evaluate_gaussian_fwd(dirs, sig, lv, lp, pp, nl2, 0)
    assert float(out[1].abs().max()) == 0.0
    ref = cref.evaluate_gaussian_fwd(dirs[:, :, :], sig, lv[:, :3].contiguous(), lp[:, :3].contiguous(), pp,
                                     torch.tensor([3, 0], dtype=torch.int32), 0)
    assert torch.equal(out[0], ref[0])
