"""CPU: the hand-written backward of the gsplat restatement IS the derivative of its forward (VERDICT r1 "next" #1d).

gsplat 0.1.11's source is absent, so the restatement (oracle/gsplat_oracle.c) stays "parity unpinned"; what CAN be
shown is internal consistency at a precision fp32 cannot give: the same C code compiled in fp64 (orc64_*), the forward
differentiated numerically (central differences, h = 1e-6) along random directions of every input -- means, scales,
quaternions, opacities, colours -- against <analytic gradient, direction> from rasterize_backward + project_backward.
The tile lists are frozen at the unperturbed state (they are piecewise constant in the inputs).  The scene stays away
from the documented upstream quirks, which are NOT derivatives of the forward by design:
  * alpha cap 0.999 forward / 0.99 backward        -> opacities <= 0.9
  * the EWA vjp ignores the 1.3 * tan(fov) clamp (A.5)          -> everything well inside the frustum
  * v_quat is the vjp w.r.t. the NORMALISED quaternion (A.5)    -> chained here by hand: (I - q q^T) / |q|
  * 0.5 / (comp + 1e-6) in the compensation vjp                 -> a 1e-6 relative deviation, below the bar
"""
import math

import pytest
import torch

from oracle import cref, cref64
from scenes import head_scene

H = W = 64
BG = torch.tensor([0.3, 0.5, 0.2], dtype=torch.float64)


def _scene(seed):
    s = head_scene(90, H, W, seed=seed, focal=150.0, max_opacity=0.9, scale_range=(4.0, 20.0), cam_dist=700.0)
    return {k: (v.double() if torch.is_tensor(v) else v) for k, v in s.items()}


def _forward(s, x, lists, w_img, w_alpha):
    means, scales, quats, opacity, colors = x
    xys, depths, radii, conics, comp, nth, cov3d = cref64.project_fwd(
        means, scales, 1.0, quats, s["viewmat"], s["fx"], s["fy"], s["cx"], s["cy"], H, W, 16, 0.1)
    ids, bins = lists
    img, Ts, idx = cref64.rasterize_fwd(ids, bins, xys, conics, colors, opacity[:, 0] * comp, H, W, 16, BG)
    loss = float((img * w_img).sum() + ((1.0 - Ts) * w_alpha).sum())
    return loss, (xys, depths, radii, conics, comp, cov3d, Ts, idx)


def _analytic(s, x, lists, w_img, w_alpha, aux):
    means, scales, quats, opacity, colors = x
    xys, depths, radii, conics, comp, cov3d, Ts, idx = aux
    ids, bins = lists
    opac_eff = opacity[:, 0] * comp
    v_xy, v_conic, v_col, v_op = cref64.rasterize_bwd(ids, bins, xys, conics, colors, opac_eff, H, W, 16, BG, Ts, idx,
                                                      w_img, w_alpha)
    v_opacity = (v_op * comp)[:, None]
    v_comp = v_op * opacity[:, 0]
    v_mean, v_scale, v_quat = cref64.project_bwd(means, scales, 1.0, quats, s["viewmat"], s["fx"], s["fy"], cov3d, radii,
                                                 conics, comp, v_xy, torch.zeros_like(depths), v_conic, v_comp)
    # chain the quaternion vjp through the in-kernel normalisation (upstream leaves this to autograd of F.normalize)
    n = quats.norm(dim=-1, keepdim=True)
    qh = quats / n
    v_quat = (v_quat - qh * (qh * v_quat).sum(-1, keepdim=True)) / n
    return v_mean, v_scale, v_quat, v_opacity, v_col


@pytest.mark.parametrize("seed", [3, 4])
def test_backward_is_the_derivative_of_the_forward_fp64(seed):
    s = _scene(seed)
    x0 = [s["means"], s["scales"], s["quats"] * 1.3, s["opacity"], s["colors"]]  # |q| != 1: normalisation is exercised
    g = torch.Generator().manual_seed(seed)
    w_img = torch.randn(H, W, 3, generator=g, dtype=torch.float64)
    w_alpha = torch.randn(H, W, generator=g, dtype=torch.float64)
    # frozen lists from the fp32 oracle at the unperturbed state
    xys, depths, radii, conics, comp, nth, _ = cref.project_gaussians(x0[0], x0[1], 1.0, x0[2], s["viewmat"], s["fx"], s["fy"],
                                                                       s["cx"], s["cy"], H, W, 16, 0.1)
    assert int((radii > 0).sum()) > 60 and int(nth.sum()) > 300
    _, ids, bins = cref.bin_and_sort(xys, depths, radii, nth, H, W, 16)
    lists = (ids, bins)
    L0, aux = _forward(s, x0, lists, w_img, w_alpha)
    grads = _analytic(s, x0, lists, w_img, w_alpha, aux)
    assert float((1.0 - aux[6]).max()) > 0.5  # real occlusion: the transmittance chain matters
    names = ("means", "scales", "quats", "opacity", "colors")
    steps = (1e-4, 1e-5, 1e-6, 1e-6, 1e-6)  # means are in mm (|x| ~ 100), the rest O(1)
    worst = {}
    for k, (name, h) in enumerate(zip(names, steps)):
        errs = []
        for trial in range(6):
            d = torch.randn(x0[k].shape, generator=g, dtype=torch.float64)
            xp = [t.clone() for t in x0]
            xm = [t.clone() for t in x0]
            xp[k] = x0[k] + h * d
            xm[k] = x0[k] - h * d
            fd = (_forward(s, xp, lists, w_img, w_alpha)[0] - _forward(s, xm, lists, w_img, w_alpha)[0]) / (2 * h)
            an = float((grads[k] * d).sum())
            errs.append(abs(fd - an) / max(abs(an), 1e-12))
        errs.sort()
        worst[name] = errs
        # a perturbation can push a (pixel, Gaussian) pair across alpha = 1/255 or T = 1e-4, where the forward jumps:
        # such a trial is off by orders of magnitude, the others agree to ~1e-7.  At most one of six may do so.
        assert errs[-2] < 2e-5, (name, errs)
    print("\nGSPLAT_FD", {k: ["%.1e" % e for e in v] for k, v in worst.items()})
