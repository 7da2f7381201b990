"""GPU parity: HIP sgutils kernels (through the C ABI) vs the CPU oracle of sg.cu:27-175.
Tolerance: rel-L2 <= 1e-4 (north_star); the reference itself runs with -use_fast_math."""
import pytest
import torch

from scenes import rel_l2
from test_oracle_sg import sg_inputs

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("w_type", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(2, 1000, 7), (1, 257, 1), (3, 64, 33)])
def test_sg_fwd_bwd_matches_oracle(w_type, shape):
    from goliath_amd import sg
    from oracle import cref

    N, D, L = shape
    dirs, sig, lv, lp, pp, nl = sg_inputs(N=min(N, 2), D=D, L=L)
    if N > 2:
        dirs, sig, lv, lp, pp = (torch.cat([t, t[:1]], 0) for t in (dirs, sig, lv, lp, pp))
        nl = torch.cat([nl, torch.tensor([1], dtype=torch.int32)])
    ref = cref.evaluate_gaussian_fwd(dirs, sig, lv, lp, pp, nl, w_type)
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(4))
    gd_ref, gs_ref, gl_ref = cref.evaluate_gaussian_bwd(dirs, sig, lv, lp, pp, nl, go, w_type, want_light_grad=True)

    d = dirs.cuda().requires_grad_(True)
    s = sig.cuda().requires_grad_(True)
    v = lv.cuda().requires_grad_(True)
    out = sg.evaluate_gaussian(d, s, v, lp.cuda(), pp.cuda(), nl.cuda(), w_type=w_type, normalize_lobe_dirs=False)
    assert rel_l2(out, ref) < TOL
    out.backward(go.cuda())
    assert rel_l2(d.grad, gd_ref) < TOL
    assert rel_l2(s.grad, gs_ref) < TOL
    assert rel_l2(v.grad, gl_ref) < TOL


def test_sg_lib_contract_and_errors():
    from goliath_amd import sg

    dirs, sig, lv, lp, pp, nl = (t.cuda() for t in sg_inputs())
    out = torch.empty(2, dirs.shape[1], 3, device="cuda")
    assert sg.sgutilslib.evaluate_gaussian_fwd(dirs, sig, lv, lp, pp, nl, out, 0) == []
    assert torch.isfinite(out).all()
    with pytest.raises(RuntimeError):  # non-contiguous input (CHECK_INPUT in the reference)
        sg.sgutilslib.evaluate_gaussian_fwd(dirs.transpose(0, 1), sig, lv, lp, pp, nl, out, 0)
    with pytest.raises(RuntimeError):  # batch mismatch
        sg.sgutilslib.evaluate_gaussian_fwd(dirs, sig[:1], lv, lp, pp, nl, out, 0)
    with pytest.raises(RuntimeError):  # CPU tensor
        sg.sgutilslib.evaluate_gaussian_fwd(dirs.cpu(), sig, lv, lp, pp, nl, out, 0)
    # no light grad requested -> backward returns None for light_values
    d = dirs.clone().requires_grad_(True)
    o = sg.evaluate_gaussian(d, sig, lv, lp, pp, nl)
    o.sum().backward()
    assert d.grad is not None and torch.isfinite(d.grad).all()


def test_sg_empty_and_zero_lights():
    from goliath_amd import sg

    dirs, sig, lv, lp, pp, nl = (t.cuda() for t in sg_inputs())
    nl0 = torch.zeros_like(nl)
    out = sg.evaluate_gaussian(dirs, sig, lv, lp, pp, nl0)
    assert float(out.abs().max()) == 0.0
    e = sg.evaluate_gaussian(dirs[:, :0], sig[:, :0], lv, lp, pp[:, :0], nl)
    assert e.shape == (2, 0, 3)


def test_sg_full_size_linearity():
    """BASELINE size (250k Gaussians): linear in light_values, additive over light subsets."""
    from goliath_amd import sg

    dirs, sig, lv, lp, pp, nl = (t.cuda() for t in sg_inputs(N=1, D=250_000, L=8, seed=5))
    a = sg.evaluate_gaussian(dirs, sig, lv, lp, pp, nl)
    b = sg.evaluate_gaussian(dirs, sig, 2.5 * lv, lp, pp, nl)
    assert rel_l2(b, 2.5 * a) < 1e-6
    first = sg.evaluate_gaussian(dirs, sig, lv[:, :3].contiguous(), lp[:, :3].contiguous(), pp, torch.full_like(nl, 3))
    rest = sg.evaluate_gaussian(dirs, sig, lv[:, 3:].contiguous(), lp[:, 3:].contiguous(), pp, torch.full_like(nl, 5))
    assert rel_l2(first + rest, a) < 1e-5


@pytest.mark.parametrize("w_type", [0, 1, 2, 3])
def test_sg_matches_reference_kernels_compiled_for_the_host(w_type):
    """R6 pinned: the HIP kernels against the reference's OWN sg.cu kernels (oracle/_ref/libref.so, built from
    /root/reference/extensions/sgutils/sg.cu by oracle/Makefile), incl. the clamped-cosine / -20 branch."""
    from goliath_amd import sg
    from oracle import refso

    if not refso.available():
        pytest.skip("oracle/_ref/libref.so not built")
    dirs, sig, lv, lp, pp, nl = sg_inputs(N=2, D=2000, L=9, seed=21)
    dirs = dirs * (1.0 + 0.3 * torch.rand(2, 2000, 1, generator=torch.Generator().manual_seed(2)))  # |cos| can exceed 1
    ref = refso.evaluate_gaussian_fwd(dirs, sig, lv, lp, pp, nl, w_type)
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(4))
    gd_ref, gs_ref, gl_ref = refso.evaluate_gaussian_bwd(dirs, sig, lv, lp, pp, nl, go, w_type, want_light_grad=True)
    d, s, v = (t.cuda().requires_grad_(True) for t in (dirs, sig, lv))
    out = sg.evaluate_gaussian(d, s, v, lp.cuda(), pp.cuda(), nl.cuda(), w_type=w_type, normalize_lobe_dirs=False)
    assert rel_l2(out, ref) < TOL
    out.backward(go.cuda())
    assert rel_l2(d.grad, gd_ref) < TOL and rel_l2(s.grad, gs_ref) < TOL and rel_l2(v.grad, gl_ref) < TOL
