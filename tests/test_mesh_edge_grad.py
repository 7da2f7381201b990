"""CPU: the differentiable part of goliath_amd.meshraster (round 4) -- `render` (depth / barycentrics as functions of the
projected vertices, drtk.render) and `edge_grad_estimator` (drtk.edge_grad_estimator) -- on top of index images from the
numpy z-buffer of oracle/mesh_ref.py (the HIP rasterizer produces the same images on the GPU: tests/test_gpu_meshraster.py).

drtk is absent (requirements.txt:6, unpinned), so these are not parity tests against it:
  * interior: the gradient of a smooth function of depth / barycentrics equals its finite-difference derivative (exact
    arithmetic statement, 1e-6);
  * discontinuities: the estimator's gradient of <g, image> w.r.t. rigid moves and single-vertex moves of a flat-coloured
    occluder matches the finite-difference derivative of the same functional on a 16x SUPERSAMPLED (box-filtered) render --
    the quantity an edge-gradient estimator approximates -- within 15 % for a silhouette and for an occlusion boundary.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W = 40, 48


def _raster(v_pix, vi, h=H, w=W):
    from oracle import mesh_ref

    idx, depth, bary = mesh_ref.rasterize(v_pix.detach().numpy(), vi.numpy(), h, w)
    return torch.from_numpy(idx), torch.from_numpy(depth), torch.from_numpy(bary)


def _scene():
    # face 0: a large triangle at depth 5; face 1: a smaller, nearer one (depth 3) overlapping its right part
    v = torch.tensor([[[6.3, 5.2, 5.0], [41.7, 9.4, 5.0], [17.9, 35.6, 5.0],
                       [24.4, 12.3, 3.0], [44.2, 20.8, 3.0], [27.6, 33.1, 3.0]]], dtype=torch.float64)
    vi = torch.tensor([[0, 1, 2], [3, 4, 5]])
    return v, vi


def _smooth_field(h, w, c, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float64) + 0.5, torch.arange(w, dtype=torch.float64) + 0.5, indexing="ij")
    out = torch.zeros(1, c, h, w, dtype=torch.float64)
    for k in range(c):
        for _ in range(3):
            fx, fy, ph = (torch.rand(3, generator=g, dtype=torch.float64) * torch.tensor([0.25, 0.25, 6.28])).tolist()
            out[0, k] += torch.cos(fx * xx + fy * yy + ph)
    return out


def test_render_matches_the_rasterizer_and_its_gradient_is_exact_in_the_interior():
    from goliath_amd import meshraster

    v, vi = _scene()
    idx, depth, bary = _raster(v, vi)
    leaf = v.clone().requires_grad_(True)
    d, b = meshraster.render(leaf, vi, idx)
    assert torch.allclose(d, depth, atol=1e-12) and torch.allclose(b, bary, atol=1e-12)
    # a smooth functional of depth and barycentrics over the pixels that keep their face under small moves
    gd, gb = _smooth_field(H, W, 1, 1)[0, 0], _smooth_field(H, W, 3, 2)[0]
    interior = torch.from_numpy(np.minimum.reduce([np.roll(idx.numpy(), s, a) == idx.numpy() for s in (1, -1) for a in (1, 2)]))
    interior &= idx >= 0

    def functional(vv, index):
        dd, bb = meshraster.render(vv, vi, index)
        return ((dd * gd)[interior]).sum() + ((bb[0] * gb)[:, interior[0]]).sum()

    functional(leaf, idx).backward()
    g = torch.Generator().manual_seed(3)
    for _ in range(4):
        dv = torch.randn(v.shape, generator=g, dtype=torch.float64)
        eps = 1e-6
        fd = (functional(v + eps * dv, idx) - functional(v - eps * dv, idx)) / (2 * eps)
        an = (leaf.grad * dv).sum()
        assert abs(float(fd - an)) < 1e-6 * max(1.0, abs(float(fd))), (float(fd), float(an))


def _flat_image(idx, colors):
    """[1,3,H,W] flat colour per face, black background."""
    img = torch.zeros(1, 3, *idx.shape[1:], dtype=torch.float64)
    for f, c in enumerate(colors):
        img += (idx == f)[:, None].double() * torch.tensor(c, dtype=torch.float64)[None, :, None, None]
    return img


def _aa_functional(v, vi, colors, g, S=16):
    """<g, box-filtered S x supersampled flat-colour render>."""
    vs = v.clone()
    vs[..., :2] *= S
    idx, _, _ = _raster(vs, vi, H * S, W * S)
    img = _flat_image(idx, colors)
    img = torch.nn.functional.avg_pool2d(img, S)
    return float((img * g).sum())


@pytest.mark.parametrize("move", ["translate_far_face", "translate_near_face", "one_vertex_of_the_near_face"])
def test_edge_gradient_matches_the_derivative_of_the_supersampled_render(move):
    from goliath_amd import meshraster

    v, vi = _scene()
    colors = [(0.9, 0.3, 0.1), (0.1, 0.5, 0.8)]
    g = _smooth_field(H, W, 3, 7)
    dv = torch.zeros_like(v)
    if move == "translate_far_face":
        dv[0, 0:3, 0], dv[0, 0:3, 1] = 0.8, 0.6          # silhouette of face 0 moves; where face 1 covers it nothing changes
    elif move == "translate_near_face":
        dv[0, 3:6, 0], dv[0, 3:6, 1] = -0.6, 0.8         # occlusion boundary (over face 0) and silhouette (over the background)
    else:
        dv[0, 4, 0], dv[0, 4, 1] = 0.7, -0.7
    eps = 0.25
    fd = (_aa_functional(v + eps * dv, vi, colors, g) - _aa_functional(v - eps * dv, vi, colors, g)) / (2 * eps)
    idx, depth, bary = _raster(v, vi)
    leaf = v.clone().requires_grad_(True)
    img = meshraster.edge_grad_estimator(leaf, vi, bary, _flat_image(idx, colors), idx, depth)
    (img * g).sum().backward()
    est = float((leaf.grad * dv).sum())
    assert abs(fd) > 1.0, fd                              # the move does change the image
    assert abs(est - fd) < 0.15 * abs(fd), (move, est, fd)
    assert float(leaf.grad[..., 2].abs().max()) == 0.0   # image-plane estimator: nothing to z


def test_faces_that_share_a_mesh_edge_are_not_a_discontinuity():
    """Two coplanar faces of one surface, different flat colours: the shared edge is a texture seam, not a silhouette --
    the estimator leaves it to the interior (barycentric) gradient and only reports the outer boundary."""
    from goliath_amd import meshraster

    v = torch.tensor([[[8.2, 6.1, 4.0], [38.3, 8.7, 4.0], [36.9, 31.2, 4.0], [9.6, 30.4, 4.0]]], dtype=torch.float64)
    vi = torch.tensor([[0, 1, 2], [0, 2, 3]])
    idx, depth, bary = _raster(v, vi)
    g = _smooth_field(H, W, 3, 9)
    leaf = v.clone().requires_grad_(True)
    same = meshraster.edge_grad_estimator(leaf, vi, bary, _flat_image(idx, [(0.5, 0.5, 0.5)] * 2), idx, depth)
    (same * g).sum().backward()
    g_same = leaf.grad.clone()
    leaf.grad = None
    diff = meshraster.edge_grad_estimator(leaf, vi, bary, _flat_image(idx, [(0.5, 0.5, 0.5), (0.5, 0.5, 0.5)]), idx, depth)
    (diff * g).sum().backward()
    assert torch.equal(g_same, leaf.grad)
    # the diagonal of a single-coloured quad is no discontinuity: triangulated the other way the quad gets the same gradient,
    # up to the pixel pairs next to a corner (there the segment between the two centres crosses the diagonal AND the outline,
    # and the estimator attributes the pair to one crossing of the occluding face: measured 2-3 % of the gradient's norm)
    vi2 = torch.tensor([[0, 1, 3], [1, 2, 3]])
    idx2, depth2, bary2 = _raster(v, vi2)
    leaf2 = v.clone().requires_grad_(True)
    other = meshraster.edge_grad_estimator(leaf2, vi2, bary2, _flat_image(idx2, [(0.5, 0.5, 0.5)] * 2), idx2, depth2)
    (other * g).sum().backward()
    assert float((leaf2.grad - g_same).norm() / g_same.norm()) < 0.05
    # ... and the quad's gradient is the derivative of the supersampled render as well
    dv = torch.zeros_like(v)
    dv[0, :, 0], dv[0, :, 1] = 0.5, -0.8
    fd = (_aa_functional(v + 0.25 * dv, vi, [(0.5, 0.5, 0.5)] * 2, g) - _aa_functional(v - 0.25 * dv, vi, [(0.5, 0.5, 0.5)] * 2, g)) / 0.5
    assert abs(float((g_same * dv).sum()) - fd) < 0.15 * abs(fd)
