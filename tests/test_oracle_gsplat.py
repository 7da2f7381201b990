"""CPU: the C oracle of the gsplat-0.1.11 path vs an independent torch/autograd restatement.

PARITY UNPINNED: the reference has no golden vectors for this path; these tests pin
the oracle's forward to a second restatement and its hand-written backward to autograd.
"""
import torch

from oracle import cref, torch_ref
from scenes import head_scene, rel_l2


def _scene(N=600, H=96, W=80, **kw):
    return head_scene(N, H, W, seed=7, scale_range=(2.0, 12.0), focal=260.0, max_opacity=0.9, **kw)


def test_project_forward_matches_torch():
    s = _scene()
    a = cref.project_gaussians(s["means"], s["scales"], 1.0, s["quats"], s["viewmat"], s["fx"], s["fy"],
                               s["cx"], s["cy"], s["H"], s["W"], 16, 0.1)
    b = torch_ref.project_gaussians(s["means"], s["scales"], 1.0, s["quats"], s["viewmat"], s["fx"], s["fy"],
                                    s["cx"], s["cy"], s["H"], s["W"], 16, 0.1)
    names = ["xys", "depths", "radii", "conics", "comp", "tiles", "cov3d"]
    assert int(a[5].sum()) > 0
    for n, x, y in zip(names, a, b):
        if x.dtype == torch.int32:
            assert (x != y).float().mean() < 0.01, n  # ceil()/int() may flip on an ulp
        else:
            assert rel_l2(x, y) < 1e-5, (n, rel_l2(x, y))


def test_bin_sort_is_sorted_and_complete():
    s = _scene()
    xys, depths, radii, conics, comp, nth, _ = cref.project_gaussians(
        s["means"], s["scales"], 1.0, s["quats"], s["viewmat"], s["fx"], s["fy"], s["cx"], s["cy"],
        s["H"], s["W"], 16, 0.1)
    keys, ids, bins = cref.bin_and_sort(xys, depths, radii, nth, s["H"], s["W"], 16)
    assert keys.numel() == int(nth.sum())
    assert bool((keys[1:] >= keys[:-1]).all())
    assert int((bins[:, 1] - bins[:, 0]).sum()) == keys.numel()
    # every gaussian appears exactly num_tiles_hit times
    assert torch.equal(torch.bincount(ids.long(), minlength=nth.numel()).int(), nth)


def test_raster_forward_and_backward_match_autograd():
    s = _scene()
    H, W = s["H"], s["W"]
    xys, depths, radii, conics, comp, nth, _ = cref.project_gaussians(
        s["means"], s["scales"], 1.0, s["quats"], s["viewmat"], s["fx"], s["fy"], s["cx"], s["cy"], H, W, 16, 0.1)
    _, ids, bins = cref.bin_and_sort(xys, depths, radii, nth, H, W, 16)
    opac = (s["opacity"][:, 0] * comp).contiguous()
    bg = torch.tensor([0.1, 0.2, 0.3])
    out, Ts, idx = cref.rasterize_forward(ids, bins, xys, conics, s["colors"], opac, H, W, 16, bg)

    tx = xys.clone().requires_grad_(True)
    tc = conics.clone().requires_grad_(True)
    tcol = s["colors"].clone().requires_grad_(True)
    to = opac.clone().requires_grad_(True)
    out_t, Ts_t = torch_ref.rasterize(ids, bins, tx, tc, tcol, to, H, W, 16, bg)
    assert rel_l2(out, out_t) < 1e-5
    assert rel_l2(Ts, Ts_t) < 1e-5
    assert float((1 - Ts).max()) > 0.5  # the scene is not empty

    g = torch.Generator().manual_seed(3)
    v_out = torch.randn(H, W, 3, generator=g)
    v_alpha = torch.randn(H, W, generator=g)
    ((out_t * v_out).sum() + ((1 - Ts_t) * v_alpha).sum()).backward()
    # alpha never reaches either cap here (max_opacity=0.9), so autograd == explicit backward
    v_xy, v_conic, v_col, v_op = cref.rasterize_backward(ids, bins, xys, conics, s["colors"], opac, H, W, 16, bg,
                                                         Ts, idx, v_out, v_alpha)
    assert rel_l2(v_col, tcol.grad) < 1e-4
    assert rel_l2(v_op[:, 0], to.grad) < 1e-4
    assert rel_l2(v_xy, tx.grad) < 1e-4
    assert rel_l2(v_conic, tc.grad) < 1e-4


def test_project_backward_matches_autograd():
    s = _scene(N=300)
    H, W = s["H"], s["W"]
    m = s["means"].clone().requires_grad_(True)
    sc = s["scales"].clone().requires_grad_(True)
    q = s["quats"].clone().requires_grad_(True)
    xys, depths, radii, conics, comp, nth, cov3d = torch_ref.project_gaussians(
        m, sc, 1.0, q, s["viewmat"], s["fx"], s["fy"], s["cx"], s["cy"], H, W, 16, 0.1)
    g = torch.Generator().manual_seed(5)
    v_xy, v_d = torch.randn(xys.shape, generator=g), torch.randn(depths.shape, generator=g)
    v_con, v_cmp = torch.randn(conics.shape, generator=g), torch.randn(comp.shape, generator=g)
    ((xys * v_xy).sum() + (depths * v_d).sum() + (conics * v_con).sum() + (comp * v_cmp).sum()).backward()

    fw = cref.project_gaussians(s["means"], s["scales"], 1.0, s["quats"], s["viewmat"], s["fx"], s["fy"],
                                s["cx"], s["cy"], H, W, 16, 0.1)
    _, _, v_mean, v_scale, v_quat = cref.project_gaussians_backward(
        s["means"], s["scales"], 1.0, s["quats"], s["viewmat"], s["fx"], s["fy"], fw[6], fw[2], fw[3], fw[4],
        v_xy, v_d, v_con, v_cmp)
    # the fov clamp is inactive for this camera, so the upstream vjp (which ignores it) is exact
    assert rel_l2(v_mean, m.grad) < 2e-4, rel_l2(v_mean, m.grad)
    assert rel_l2(v_scale, sc.grad) < 2e-4, rel_l2(v_scale, sc.grad)
    # upstream's quaternion vjp does not chain through the in-kernel normalisation: for unit
    # quaternions autograd's gradient is the tangential projection of it
    qn = s["quats"]
    proj = v_quat - qn * (qn * v_quat).sum(-1, keepdim=True)
    assert rel_l2(proj, q.grad) < 2e-4, rel_l2(proj, q.grad)
