"""GPU: the EWA projection fused into the shading kernels (gol_shade_project_fwd / bwd + gol_render_fwd / bwd_projected,
the path AutoEncoder.forward takes: rgca.py:505-588 -> render_gsplat.py:49-63 without the attribute round trip) against
the same chain with the projection as kernels of its own (gol_shade_fwd -> gol_render_fwd, which the chain tests pin to
the oracle).  One arithmetic (csrc/gol_project.h) on the same register values: tile lists and projection outputs must be
IDENTICAL, images and gradients equal to float-accumulation-order noise."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

from scenes import rel_l2  # noqa: E402


def _inputs(B, S, H, W, env, seed=5):
    import bench

    cfg = dict(bench.CFG, gaussians=S * S, slab=S, height=H, width=W, views_per_gpu=B, focal=3000.0 * W / 1334.0, seed=seed)
    t = bench.make_inputs(cfg, torch.device("cuda"))
    if not env:
        g = torch.Generator().manual_seed(seed)
        t["light_intensity"] = (0.5 + torch.rand(B, 3, 1, generator=g)).cuda()
        t["light_pos"] = (400.0 * torch.randn(B, 3, 3, generator=g)).cuda()
        t["n_lights"] = torch.tensor([3, 2, 1, 3][:B] + [3] * max(B - 4, 0), dtype=torch.int32).cuda()
    return t


LEAVES = ("f_vn", "f_vc", "postex", "tn", "albedo")


def _run(t, H, W, env, fused, extra_consumers, rand=False):
    from goliath_amd import render_gs, shade

    for k in LEAVES:
        t[k].grad = None
    vs = render_gs.view_set(t["K"], t["Rt"], H, W) if fused else None
    kw = dict(preconv_envmap=t["mips"], lightrot=t["lightrot"]) if env else dict(
        light_intensity=t["light_intensity"], headrel_light_pos=t["light_pos"], n_lights=t["n_lights"])
    if rand:
        g = torch.Generator().manual_seed(3)
        kw["light_sh_rand"] = torch.randn(t["light_sh"].shape, generator=g).cuda()
    preds = shade.shading_tail(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"], views=vs,
                               **kw)
    assert ("projected" in preds) == fused
    rgb, alpha, depth, l1 = render_gs.render_batch(t["K"], t["Rt"], preds, H, W, l1_target=t["target"])
    loss = l1 + 1e-3 * depth.mean()
    if extra_consumers:
        # regularisers on the attributes the renderer also consumes: their gradients and the projection's ADD UP in the kernel
        loss = loss + 1e-4 * preds["primscale"].square().mean() + 1e-3 * preds["opacity"].mean() + \
            1e-5 * preds["primpos"].square().mean() + 1e-3 * (preds["primqvec"][..., 0]).mean() + 1e-3 * preds["color"].mean()
    loss.backward()
    grads = {k: t[k].grad.clone() for k in LEAVES}
    return preds, rgb.detach(), alpha.detach(), depth.detach(), float(l1), grads


@pytest.mark.parametrize("env,extra,rand", [(True, False, False), (True, True, True), (False, True, False)])
def test_fused_projection_equals_the_separate_kernels(env, extra, rand):
    from goliath_amd import splat

    B, S, H, W = 3, 64, 288, 208
    t = _inputs(B, S, H, W, env)
    p0, rgb0, a0, d0, l0, g0 = _run(t, H, W, env, False, extra, rand)
    p1, rgb1, a1, d1, l1, g1 = _run(t, H, W, env, True, extra, rand)
    # the projection outputs the shading kernel wrote == what gol_project_fwd computes from the returned attributes
    intr = torch.stack([t["K"][:, 0, 0], t["K"][:, 1, 1], t["K"][:, 0, 2], t["K"][:, 1, 2]], -1).contiguous()
    N = S * S
    with torch.no_grad():
        ref = splat._project_fwd(B, N, p0["primpos"].contiguous(), p0["primscale"].contiguous(), 1.0,
                                 p0["primqvec"].contiguous(), t["Rt"].reshape(B, 12).contiguous(), intr, H, W, 0.1,
                                 p0["opacity"].reshape(B, N).contiguous(), p0["color"].contiguous())
    cov3d, xys, depths, radii, conics, comp, nth, opac_eff, records = ref
    pr = p1["projected"]
    assert torch.equal(pr.field("radii"), radii) and int((radii > 0).sum()) > N  # most of the head is on screen
    for name, want in (("xys", xys), ("depths", depths), ("conics", conics), ("comp", comp), ("opac_eff", opac_eff)):
        assert torch.equal(pr.field(name), want), name
    assert torch.equal(pr.records, records)
    # same lists -> same images up to nothing at all
    assert torch.equal(rgb0, rgb1) and torch.equal(a0, a1) and torch.equal(d0, d1) and l0 == l1
    for k in LEAVES:
        assert rel_l2(g1[k], g0[k]) < 2e-6, k


def test_fused_projection_with_a_gaussian_count_that_is_not_a_multiple_of_four():
    """N = 33 x 33: the one-Gaussian-per-lane instantiation of the fused forward (the 4-per-lane one needs N % 4 == 0)."""
    B, S, H, W = 2, 33, 160, 112
    t = _inputs(B, S, H, W, True)
    p0, rgb0, a0, d0, l0, g0 = _run(t, H, W, True, False, True)
    p1, rgb1, a1, d1, l1, g1 = _run(t, H, W, True, True, True)
    assert torch.equal(rgb0, rgb1) and torch.equal(a0, a1) and torch.equal(d0, d1) and l0 == l1
    for k in LEAVES:
        assert rel_l2(g1[k], g0[k]) < 2e-6, k


def test_fused_projection_at_the_benchmarked_size():
    """BASELINE config 2, one micro-batch of bench.py (4 views, 250k Gaussians, 2048x1334): the chain test
    (test_gpu_fullsize.py) pins the separate kernels to the oracle; here the fused path must reproduce them -- identical
    tile lists and images, leaf gradients to accumulation-order noise."""
    import bench

    cfg = dict(bench.CFG, views_per_gpu=4)
    t = bench.make_inputs(cfg, torch.device("cuda"))
    H, W = cfg["height"], cfg["width"]
    p0, rgb0, a0, d0, l0, g0 = _run(t, H, W, True, False, False)
    lists0 = None
    p1, rgb1, a1, d1, l1, g1 = _run(t, H, W, True, True, False)
    assert torch.equal(rgb0, rgb1) and torch.equal(a0, a1) and torch.equal(d0, d1) and l0 == l1
    worst = max(rel_l2(g1[k], g0[k]) for k in LEAVES)
    assert worst < 2e-6, worst


def test_a_render_with_other_attributes_does_not_use_stale_records():
    """The diffuse / specular breakdown renders (rgca.py:232-245) swap preds["color"]: the records describe the ORIGINAL
    colours, so the renderer must project again."""
    from goliath_amd import render_gs, shade

    B, S, H, W = 2, 32, 160, 112
    t = _inputs(B, S, H, W, True)
    with torch.no_grad():
        vs = render_gs.view_set(t["K"], t["Rt"], H, W)
        preds = shade.shading_tail(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                                   preconv_envmap=t["mips"], lightrot=t["lightrot"], views=vs)
        full = render_gs.render_batch(t["K"], t["Rt"], preds, H, W)[0]
        preds["color"] = preds["diff_color"].clamp(min=0.0)
        part = render_gs.render_batch(t["K"], t["Rt"], preds, H, W)[0]
        want = render_gs.render_batch(t["K"], t["Rt"], {k: v for k, v in preds.items() if k != "projected"}, H, W)[0]
    assert torch.equal(part, want) and not torch.equal(part, full)
    # other cameras than the ones the shading call was given: projected again, too
    K2 = t["K"].clone()
    with torch.no_grad():
        preds["color"] = preds["projected"].sources["color"]
        again = render_gs.render_batch(K2, t["Rt"], preds, H, W)[0]
    assert torch.equal(again, full)


def test_projection_only_backward_when_the_image_is_unused():
    """No gradient reaches the records (the loss only touches a regulariser): the plain shading backward runs."""
    from goliath_amd import render_gs, shade

    B, S, H, W = 2, 32, 160, 112
    t = _inputs(B, S, H, W, True)
    vs = render_gs.view_set(t["K"], t["Rt"], H, W)
    preds = shade.shading_tail(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                               preconv_envmap=t["mips"], lightrot=t["lightrot"], views=vs)
    preds["primscale"].square().mean().backward()
    g1 = t["f_vn"].grad.clone()
    t["f_vn"].grad = None
    preds = shade.shading_tail(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                               preconv_envmap=t["mips"], lightrot=t["lightrot"])
    preds["primscale"].square().mean().backward()
    assert torch.equal(g1, t["f_vn"].grad)
