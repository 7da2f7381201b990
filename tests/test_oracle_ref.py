"""CPU: the reference's OWN sgutils / compute_raydirs kernels, compiled for the host (oracle/_ref/libref.so, see
oracle/Makefile and oracle/ref_shim/), pin the restatements:
    oracle/sg_oracle.c            <- extensions/sgutils/sg.cu:27-175      (all four w_types, fwd + bwd + light grad,
                                                                           the |cos| >= 1 edge with its -20 substitute)
    oracle/mvp_oracle.c:orc_raydirs <- extensions/utils/utils_kernel.cu:11-51
The two implementations evaluate the same fp32 expressions in the same order, so the bar is far below the HIP
parity tolerance (1e-6 rel-L2; light gradients are float-atomic sums)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import cref, refso
from scenes import rel_l2
from test_oracle_sg import sg_inputs

pytestmark = pytest.mark.skipif(not refso.available(), reason="oracle/_ref/libref.so not built (needs /root/reference)")


@pytest.mark.parametrize("w_type", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(2, 500, 7), (1, 129, 1), (2, 64, 33)])
def test_sg_restatement_equals_reference_kernels(w_type, shape):
    N, D, L = shape
    dirs, sig, lv, lp, pp, nl = sg_inputs(N=N, D=D, L=L)
    a = cref.evaluate_gaussian_fwd(dirs, sig, lv, lp, pp, nl, w_type)
    b = refso.evaluate_gaussian_fwd(dirs, sig, lv, lp, pp, nl, w_type)
    assert rel_l2(a, b) < 1e-6
    go = torch.randn(a.shape, generator=torch.Generator().manual_seed(4))
    ga = cref.evaluate_gaussian_bwd(dirs, sig, lv, lp, pp, nl, go, w_type, want_light_grad=True)
    gb = refso.evaluate_gaussian_bwd(dirs, sig, lv, lp, pp, nl, go, w_type, want_light_grad=True)
    for x, y, name in zip(ga, gb, ("dirs", "sigmas", "light_values")):
        assert rel_l2(x, y) < (1e-5 if name == "light_values" else 1e-6), (name, rel_l2(x, y))
    # without the optional light gradient
    gc = refso.evaluate_gaussian_bwd(dirs, sig, lv, lp, pp, nl, go, w_type)
    assert gc[2] is None and torch.equal(gc[0], gb[0]) and torch.equal(gc[1], gb[1])


@pytest.mark.parametrize("w_type", [0, 1, 2, 3])
def test_sg_clamped_cosine_branch(w_type):
    """Un-normalised lobe directions push dot(ldir, dir) beyond +-1: the forward clamps (sg.cu:53), the backward
    substitutes -20 for d acos / dc (sg.cu:129,139) or gates the clamp (w_type 2/3)."""
    dirs, sig, lv, lp, pp, nl = sg_inputs(N=2, D=300, L=5, seed=3)
    dirs = dirs * (1.0 + 0.5 * torch.rand(2, 300, 1, generator=torch.Generator().manual_seed(9)))  # |dir| in [1, 1.5]
    dirs[0, :50] = F.normalize(lp[0, 0][None] - pp[0, :50], dim=-1) * 1.2  # pointing exactly at light 0: cos = 1.2
    a = cref.evaluate_gaussian_fwd(dirs, sig, lv, lp, pp, nl, w_type)
    b = refso.evaluate_gaussian_fwd(dirs, sig, lv, lp, pp, nl, w_type)
    assert rel_l2(a, b) < 1e-6
    go = torch.randn(a.shape, generator=torch.Generator().manual_seed(5))
    ga = cref.evaluate_gaussian_bwd(dirs, sig, lv, lp, pp, nl, go, w_type, want_light_grad=True)
    gb = refso.evaluate_gaussian_bwd(dirs, sig, lv, lp, pp, nl, go, w_type, want_light_grad=True)
    for x, y in zip(ga, gb):
        assert torch.isfinite(y).all()
        assert rel_l2(x, y) < 1e-5


def test_raydirs_restatement_equals_reference_kernel():
    g = torch.Generator().manual_seed(1)
    N, H, W = 2, 37, 53
    viewpos = torch.tensor([[0.1, -0.2, -3.0], [1.5, 0.3, -2.0]])
    q = F.normalize(torch.randn(N, 4, generator=g) * 0.2 + torch.tensor([1.0, 0, 0, 0]), dim=-1)
    w, x, y, z = q.unbind(-1)
    viewrot = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                           1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                           1 - 2 * (x * x + y * y)], -1).reshape(N, 3, 3)
    focal, princpt = torch.full((N, 2), 80.0), torch.tensor([[W / 2.0, H / 2.0]] * N)
    pix = torch.stack(torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")[::-1], -1)
    pix = (pix[None] + torch.rand(N, H, W, 2, generator=g)).contiguous()
    for pc in (pix, (W, H)):
        a = cref.compute_raydirs(viewpos, viewrot, focal, princpt, pc, 1.5)
        b = refso.compute_raydirs(viewpos, viewrot, focal, princpt, pc, 1.5)
        for u, v in zip(a, b):
            assert torch.isfinite(v).all() and rel_l2(u, v) < 1e-6
