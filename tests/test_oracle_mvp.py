"""CPU: the C oracle of the MVP ray marcher / raydirs vs golden vectors produced by the reference's OWN
in-tree PyTorch restatements (tests/golden/make_mvp_golden.py).  PINNED."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cref
from scenes import rel_l2

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GDIR, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def mvp_case(G, tag):
    """Leaf parameters -> kernel inputs exactly as the reference's gradcheck feeds its CUDA op
    (mvpraymarch.py:682-689)."""
    leaf = {k: G[f"{tag}/leaf_{k}"].clone().requires_grad_(True) for k in ("template", "primpos", "primrot", "primscale")}
    template = F.softplus(leaf["template"] * 1.5).permute(0, 1, 3, 4, 5, 2).contiguous()
    primpos = leaf["primpos"] * 0.3
    primscale = torch.exp(0.1 * leaf["primscale"])
    return leaf, template, primpos, leaf["primrot"], primscale


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mvp_oracle_reproduces_reference(tag):
    G = load("mvp_golden.npz")
    leaf, template, primpos, primrot, primscale = mvp_case(G, tag)
    fs, fe = (float(v) for v in G[f"{tag}/fade"])
    step = float(G[f"{tag}/stepsize"])
    rgba, raysat, _ = cref.mvp_forward(G[f"{tag}/raypos"], G[f"{tag}/raydir"], step, G[f"{tag}/tminmax"], primpos,
                                       primrot, primscale, template, fs, fe)
    assert rel_l2(rgba, G[f"{tag}/rayrgba"]) < 1e-5, rel_l2(rgba, G[f"{tag}/rayrgba"])
    assert float((raysat[..., 0] > -1).float().mean()) > 0.3  # the recipe saturates many rays
    gp, gr, gs, gt = cref.mvp_backward(G[f"{tag}/raypos"], G[f"{tag}/raydir"], step, G[f"{tag}/tminmax"], primpos,
                                       primrot, primscale, template, raysat, torch.ones_like(rgba), fs, fe)
    # chain the kernel-input gradients back to the reference's leaf parameters with autograd
    (template * gt).sum().add((primpos * gp).sum()).add((primrot * gr).sum()).add((primscale * gs).sum()).backward()
    for k in ("template", "primpos", "primrot", "primscale"):
        e = rel_l2(leaf[k].grad, G[f"{tag}/grad_{k}"])
        assert e < 2e-4, (k, e)


def test_mvp_oracle_with_warp_field_reproduces_reference():
    """algo 1 (mvpraymarch.py:790-803): the reference's PyTorch fixture with dowarp=True -- rgba and all five leaf gradients."""
    G = load("mvp_golden.npz")
    leaf, template, primpos, primrot, primscale = mvp_case(G, "w")
    warp_leaf = G["w/leaf_warp"].clone().requires_grad_(True)
    warp = warp_leaf.permute(0, 1, 3, 4, 5, 2).contiguous()  # chlast, mvpraymarch.py:686
    fs, fe = (float(v) for v in G["w/fade"])
    step = float(G["w/stepsize"])
    rgba, raysat, _ = cref.mvp_forward(G["w/raypos"], G["w/raydir"], step, G["w/tminmax"], primpos, primrot, primscale,
                                       template, fs, fe, warp=warp)
    assert rel_l2(rgba, G["w/rayrgba"]) < 1e-5, rel_l2(rgba, G["w/rayrgba"])
    # the warp matters: without it the image differs
    rgba0, _, _ = cref.mvp_forward(G["w/raypos"], G["w/raydir"], step, G["w/tminmax"], primpos, primrot, primscale,
                                   template, fs, fe)
    assert rel_l2(rgba0, G["w/rayrgba"]) > 1e-4
    gp, gr, gs, gt, gw = cref.mvp_backward(G["w/raypos"], G["w/raydir"], step, G["w/tminmax"], primpos, primrot, primscale,
                                           template, raysat, torch.ones_like(rgba), fs, fe, warp=warp)
    ((template * gt).sum() + (primpos * gp).sum() + (primrot * gr).sum() + (primscale * gs).sum()
     + (warp * gw).sum()).backward()
    for k in ("template", "primpos", "primrot", "primscale"):
        e = rel_l2(leaf[k].grad, G[f"w/grad_{k}"])
        assert e < 2e-4, (k, e)
    e = rel_l2(warp_leaf.grad, G["w/grad_warp"])
    assert e < 2e-4, ("warp", e)


def test_raydirs_oracle_reproduces_reference():
    G = load("raydirs_golden.npz")
    rp, rd, tm = cref.compute_raydirs(G["viewpos"], G["viewrot"], G["focal"], G["princpt"], G["pixelcoords"], 1.0)
    assert rel_l2(rd, G["raydir"]) < 1e-6
    assert rel_l2(tm, G["tminmax"]) < 1e-5
    assert rel_l2(rp, G["viewpos"][:, None, None, :].expand_as(rp)) < 1e-7
    # implicit pixel grid == explicit one
    H, W = G["pixelcoords"].shape[1:3]
    rp2, rd2, tm2 = cref.compute_raydirs(G["viewpos"], G["viewrot"], G["focal"], G["princpt"], (W, H), 1.0)
    assert torch.equal(rd2, rd) and torch.equal(tm2, tm)


def test_aabb_contains_box_corners():
    G = load("mvp_golden.npz")
    _, _, primpos, primrot, primscale = mvp_case(G, "a")
    primpos, primscale = primpos.detach(), primscale.detach()
    A = cref.mvp_aabb(primpos, primrot.detach(), primscale)
    K = primpos.shape[1]
    assert A.shape == (2, 2 * K - 1, 2, 3)
    assert bool((A[:, 0, 0] <= A[:, K - 1:, 0].min(1).values + 1e-6).all())  # root encloses every leaf
    assert bool((A[:, 0, 1] >= A[:, K - 1:, 1].max(1).values - 1e-6).all())
    # leaf box contains its centre
    assert bool(((A[:, K - 1:, 0] <= primpos) & (A[:, K - 1:, 1] >= primpos)).all())
