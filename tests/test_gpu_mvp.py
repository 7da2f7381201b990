"""GPU parity: HIP MVP ray marcher / raydirs / AABBs (C ABI) vs
  (a) golden vectors from the reference's own in-tree PyTorch ray marcher (tests/golden/mvp_golden.npz,
      raydirs_golden.npz) and (b) the C oracle (oracle/mvp_oracle.c, itself pinned to (a)).
Tolerance rel-L2 <= 1e-4 forward; gradients 3e-4 (float atomics + __powf/__expf like the reference's
-use_fast_math build)."""
import pytest
import torch
import torch.nn.functional as F

from scenes import rel_l2
from test_oracle_mvp import load, mvp_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mvp_matches_reference_golden(tag):
    from goliath_amd import mvp

    G = load("mvp_golden.npz")
    leaf = {k: G[f"{tag}/leaf_{k}"].cuda().requires_grad_(True) for k in ("template", "primpos", "primrot", "primscale")}
    template = F.softplus(leaf["template"] * 1.5).permute(0, 1, 3, 4, 5, 2).contiguous()
    fs, fe = (float(v) for v in G[f"{tag}/fade"])
    out = mvp.mvpraymarch(G[f"{tag}/raypos"].cuda(), G[f"{tag}/raydir"].cuda(), float(G[f"{tag}/stepsize"]),
                          G[f"{tag}/tminmax"].cuda(), (leaf["primpos"] * 0.3, leaf["primrot"].contiguous(),
                                                       torch.exp(0.1 * leaf["primscale"])),
                          template, None, fadescale=fs, fadeexp=fe, accum=0)
    assert rel_l2(out, G[f"{tag}/rayrgba"]) < 2e-6, rel_l2(out, G[f"{tag}/rayrgba"])   # measured 1.7e-7
    out.backward(torch.ones_like(out))
    for k in leaf:
        e = rel_l2(leaf[k].grad, G[f"{tag}/grad_{k}"])
        assert e < 2e-5, (k, e)   # measured 1.8e-6 (sums of float atomics)


def _random_case(N, H, W, K, T, seed, step=0.05):
    g = torch.Generator().manual_seed(seed)
    k3 = round(K ** (1 / 3))
    lin = torch.linspace(-0.75, 0.75, k3)
    centres = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)[:K]
    primpos = (centres[None] + 0.05 * torch.randn(N, K, 3, generator=g)).contiguous()
    q = F.normalize(torch.randn(N, K, 4, generator=g), dim=-1)
    w, x, y, z = q.unbind(-1)
    primrot = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                           1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                           1 - 2 * (x * x + y * y)], -1).reshape(N, K, 3, 3).contiguous()
    primscale = (k3 * (0.8 + 0.4 * torch.rand(N, K, 3, generator=g))).contiguous()
    template = F.softplus(1.5 * torch.randn(N, K, T[0], T[1], T[2], 4, generator=g))
    template[..., 3] = F.softplus(1.5 * torch.randn(N, K, *T, generator=g) - 2.0)
    viewpos = torch.tensor([[0.1, -0.2, -3.0]] * N) + 0.2 * torch.randn(N, 3, generator=g)
    viewrot = torch.eye(3)[None].repeat(N, 1, 1).contiguous()
    focal = torch.full((N, 2), 1.6 * W)
    princpt = torch.tensor([[W * 0.5, H * 0.5]] * N)
    return dict(primpos=primpos, primrot=primrot, primscale=primscale, template=template.contiguous(), viewpos=viewpos,
                viewrot=viewrot, focal=focal, princpt=princpt, step=step)


def test_raydirs_and_aabb_vs_golden_and_oracle():
    from goliath_amd import mvp
    from oracle import cref

    G = load("raydirs_golden.npz")
    c = lambda t: t.cuda().contiguous()
    rp, rd, tm = mvp.compute_raydirs(c(G["viewpos"]), c(G["viewrot"]), c(G["focal"]), c(G["princpt"]),
                                     c(G["pixelcoords"]), 1.0)
    assert rel_l2(rd, G["raydir"]) < 1e-6 and rel_l2(tm, G["tminmax"]) < 1e-5
    H, W = G["pixelcoords"].shape[1:3]
    rp2, rd2, tm2 = mvp.compute_raydirs(c(G["viewpos"]), c(G["viewrot"]), c(G["focal"]), c(G["princpt"]), (W, H), 1.0)
    assert torch.equal(rd2, rd) and torch.equal(tm2, tm) and torch.equal(rp2, rp)
    from oracle import refso

    if refso.available():  # the reference's own compute_raydirs kernel compiled for the host (oracle/_ref)
        for pc_gpu, pc_cpu in ((c(G["pixelcoords"]), G["pixelcoords"]), ((W, H), (W, H))):
            got = mvp.compute_raydirs(c(G["viewpos"]), c(G["viewrot"]), c(G["focal"]), c(G["princpt"]), pc_gpu, 1.0)
            want = refso.compute_raydirs(G["viewpos"], G["viewrot"], G["focal"], G["princpt"], pc_cpu, 1.0)
            assert rel_l2(got[1], want[1]) < 1e-6 and rel_l2(got[2], want[2]) < 1e-5
    case = _random_case(2, 8, 8, 27, (4, 4, 4), 3)
    _, _, aabb = mvp.build_accel((c(case["primpos"]), c(case["primrot"]), c(case["primscale"])))
    assert rel_l2(aabb, cref.mvp_aabb(case["primpos"], case["primrot"], case["primscale"])) < 1e-6


@pytest.mark.parametrize("N,H,W,K,T,fe", [(2, 70, 50, 64, (4, 8, 8), 7.5), (1, 33, 17, 27, (3, 5, 6), 7.5),
                                          (1, 40, 40, 1, (8, 8, 8), 7.5), (2, 70, 50, 64, (4, 8, 8), 8.0)])
def test_mvp_vs_oracle_with_shadow(N, H, W, K, T, fe):
    """Non power-of-two K (heap leaves on two levels), ragged image sizes, K = 1, shadow splatting; fadeexp = 8 (the
    value the reference's models use) takes the squaring fast path of the fade term, 7.5 the generic one."""
    from goliath_amd import mvp
    from oracle import cref

    case = _random_case(N, H, W, K, T, seed=K)
    c = lambda t: t.cuda().contiguous()
    rp, rd, tm = mvp.compute_raydirs(c(case["viewpos"]), c(case["viewrot"]), c(case["focal"]), c(case["princpt"]),
                                     (W, H), 1.0)
    leaf = {k: c(case[k]).requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
    out, shadow = mvp.mvpraymarch(rp, rd, case["step"], tm, (leaf["primpos"], leaf["primrot"], leaf["primscale"]),
                                  leaf["template"], None, fadescale=6.5, fadeexp=fe, with_shadow=True)
    ref, raysat, ref_shadow = cref.mvp_forward(rp.cpu(), rd.cpu(), case["step"], tm.cpu(), case["primpos"],
                                               case["primrot"], case["primscale"], case["template"], 6.5, fe,
                                               with_shadow=True)
    assert float(ref[..., 3].max()) > 0.2
    assert rel_l2(out, ref) < 3e-6, rel_l2(out, ref)   # measured 3.1e-7
    ref_sh = ref_shadow[..., 0:1] / (ref_shadow[..., 1:] + 1e-5)
    assert rel_l2(shadow, ref_sh) < 4e-5   # measured 3.7e-6
    gen = torch.Generator().manual_seed(1)
    go = torch.randn(out.shape, generator=gen)
    out.backward(go.cuda())
    gp, gr, gs, gt = cref.mvp_backward(rp.cpu(), rd.cpu(), case["step"], tm.cpu(), case["primpos"], case["primrot"],
                                       case["primscale"], case["template"], raysat, go, 6.5, fe)
    for k, g in (("primpos", gp), ("primrot", gr), ("primscale", gs), ("template", gt)):
        e = rel_l2(leaf[k].grad, g)
        assert e < 2e-5, (k, e)   # measured 1.4e-6


def test_mvp_far_from_the_origin_with_a_small_step():
    """|pos| / stepsize 150x the BASELINE ratio (the scene 60 units away from the origin, step 0.004): the accumulated
    float error of the marched position grows with that ratio, and so must the margin of the per-box iteration windows
    -- a sample the reference evaluates may not fall outside its window (the margin was calibrated for the
    BASELINE ratio only).  Rays are given explicitly (the unit-cube clipping of compute_raydirs does not apply)."""
    from goliath_amd import mvp
    from oracle import cref

    N, H, W, K, T = 1, 48, 40, 27, (4, 6, 6)
    case = _random_case(N, H, W, K, T, seed=5, step=0.004)
    off = torch.tensor([60.0, -45.0, 52.0])
    c = lambda t: t.cuda().contiguous()
    rp0, rd, tm = mvp.compute_raydirs(c(case["viewpos"]), c(case["viewrot"]), c(case["focal"]), c(case["princpt"]),
                                      (W, H), 1.0)
    rp = (rp0 + off.cuda()).contiguous()
    primpos = (case["primpos"] + off).contiguous()
    leaf = {"primpos": c(primpos).requires_grad_(True), "primrot": c(case["primrot"]).requires_grad_(True),
            "primscale": c(case["primscale"]).requires_grad_(True), "template": c(case["template"]).requires_grad_(True)}
    out = mvp.mvpraymarch(rp, rd, case["step"], tm, (leaf["primpos"], leaf["primrot"], leaf["primscale"]),
                          leaf["template"], None, fadescale=6.5, fadeexp=8.0)
    ref, raysat, _ = cref.mvp_forward(rp.cpu(), rd.cpu(), case["step"], tm.cpu(), primpos, case["primrot"],
                                      case["primscale"], case["template"], 6.5, 8.0)
    assert float(ref[..., 3].max()) > 0.2
    # positions carry ~1e-5 of absolute rounding here (ulp of 60), i.e. ~3e-4 of a voxel: looser than at the origin
    assert rel_l2(out, ref) < 4e-5, rel_l2(out, ref)   # measured 3.3e-6
    gen = torch.Generator().manual_seed(1)
    go = torch.randn(out.shape, generator=gen)
    out.backward(go.cuda())
    gp, gr, gs, gt = cref.mvp_backward(rp.cpu(), rd.cpu(), case["step"], tm.cpu(), primpos, case["primrot"],
                                       case["primscale"], case["template"], raysat, go, 6.5, 8.0)
    for k, g in (("primscale", gs), ("template", gt)):
        e = rel_l2(leaf[k].grad, g)
        assert e < 1e-4, (k, e)   # measured 8.4e-6


def test_mvp_warp_fields_match_reference_golden():
    """algo 1: the reference fixture with dowarp=True (mvpraymarch.py:790-803) -- image and all five leaf gradients."""
    from goliath_amd import mvp

    G = load("mvp_golden.npz")
    leaf = {k: G[f"w/leaf_{k}"].cuda().requires_grad_(True) for k in ("template", "warp", "primpos", "primrot", "primscale")}
    template = F.softplus(leaf["template"] * 1.5).permute(0, 1, 3, 4, 5, 2).contiguous()
    warp = leaf["warp"].permute(0, 1, 3, 4, 5, 2).contiguous()
    fs, fe = (float(v) for v in G["w/fade"])
    out = mvp.mvpraymarch(G["w/raypos"].cuda(), G["w/raydir"].cuda(), float(G["w/stepsize"]), G["w/tminmax"].cuda(),
                          (leaf["primpos"] * 0.3, leaf["primrot"].contiguous(), torch.exp(0.1 * leaf["primscale"])),
                          template, warp, algo=1, fadescale=fs, fadeexp=fe, accum=0)
    assert rel_l2(out, G["w/rayrgba"]) < 2e-6, rel_l2(out, G["w/rayrgba"])   # measured 1.8e-7
    out.backward(torch.ones_like(out))
    for k in leaf:
        e = rel_l2(leaf[k].grad, G[f"w/grad_{k}"])
        assert e < 2e-5, (k, e)   # measured 1.2e-6


@pytest.mark.parametrize("N,H,W,K,T,WT,amp", [(2, 70, 50, 64, (4, 8, 8), (3, 4, 5), 0.05), (1, 33, 17, 27, (3, 5, 6), (2, 2, 2), 0.6)])
def test_mvp_warp_fields_vs_oracle_with_shadow(N, H, W, K, T, WT, amp):
    """Warp fields far from the identity (amp 0.6: warped positions leave the box, corners drop out one by one, some samples
    have no corner at all), non-cubic warp grids, shadow splatting at the warped position."""
    from goliath_amd import mvp
    from oracle import cref

    case = _random_case(N, H, W, K, T, seed=K + 7)
    g = torch.Generator().manual_seed(99)
    lin = [torch.linspace(-1, 1, n) for n in WT]
    grid = torch.stack(torch.meshgrid(*lin, indexing="ij")[::-1], -1)  # [WD,WH,WW,3] = (x, y, z) of the cell
    warp = (grid[None, None] + amp * torch.randn(N, K, *WT, 3, generator=g)).contiguous()
    c = lambda t: t.cuda().contiguous()
    rp, rd, tm = mvp.compute_raydirs(c(case["viewpos"]), c(case["viewrot"]), c(case["focal"]), c(case["princpt"]),
                                     (W, H), 1.0)
    leaf = {k: c(case[k]).requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
    leaf["warp"] = c(warp).requires_grad_(True)
    out, shadow = mvp.mvpraymarch(rp, rd, case["step"], tm, (leaf["primpos"], leaf["primrot"], leaf["primscale"]),
                                  leaf["template"], leaf["warp"], algo=1, fadescale=6.5, fadeexp=7.5, with_shadow=True)
    ref, raysat, ref_shadow = cref.mvp_forward(rp.cpu(), rd.cpu(), case["step"], tm.cpu(), case["primpos"],
                                               case["primrot"], case["primscale"], case["template"], 6.5, 7.5,
                                               with_shadow=True, warp=warp)
    assert float(ref[..., 3].max()) > 0.2
    assert rel_l2(out, ref) < 3e-6, rel_l2(out, ref)   # measured 3.1e-7
    ref_sh = ref_shadow[..., 0:1] / (ref_shadow[..., 1:] + 1e-5)
    assert rel_l2(shadow, ref_sh) < 4e-5   # measured 3.7e-6
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
    out.backward(go.cuda())
    gp, gr, gs, gt, gw = cref.mvp_backward(rp.cpu(), rd.cpu(), case["step"], tm.cpu(), case["primpos"], case["primrot"],
                                           case["primscale"], case["template"], raysat, go, 6.5, 7.5, warp=warp)
    for k, gref in (("primpos", gp), ("primrot", gr), ("primscale", gs), ("template", gt), ("warp", gw)):
        e = rel_l2(leaf[k].grad, gref)
        assert e < 2e-5, (k, e)   # measured 1.4e-6


def test_raymarcher_wrapper_and_errors():
    from goliath_amd import mvp

    case = _random_case(1, 32, 32, 8, (4, 4, 4), 5)
    c = lambda t: t.cuda().contiguous()
    rp, rd, tm = mvp.compute_raydirs(c(case["viewpos"]), c(case["viewrot"]), c(case["focal"]), c(case["princpt"]),
                                     (32, 32), 1.0)
    rm = mvp.Raymarcher(volradius=1.0, dt=0.05)
    dec = dict(primpos=c(case["primpos"]), primrot=c(case["primrot"]), primscale=c(case["primscale"]),
               primrgba=c(case["template"]), valid_prims=(torch.arange(8) % 2 == 0).cuda())
    rgb, alpha, rgba, shadow = rm(rp, rd, tm, dec, renderoptions={"fadescale": 6.5, "fadeexp": 7.5, "bogus": 1})
    assert rgb.shape == (1, 3, 32, 32) and alpha.shape == (1, 1, 32, 32) and shadow is None
    with pytest.raises(NotImplementedError):  # a warp field needs algo=1 (and algo 1 a warp field)
        mvp.mvpraymarch(rp, rd, 0.05, tm, (dec["primpos"], dec["primrot"], dec["primscale"]), dec["primrgba"],
                        torch.zeros(1, 8, 2, 2, 2, 3).cuda())
    with pytest.raises(NotImplementedError):
        mvp.mvpraymarch(rp, rd, 0.05, tm, (dec["primpos"], dec["primrot"], dec["primscale"]), dec["primrgba"], None, algo=1)
    with pytest.raises(RuntimeError):  # CPU tensors
        mvp.mvpraymarch(rp.cpu(), rd.cpu(), 0.05, tm.cpu(), (case["primpos"], case["primrot"], case["primscale"]),
                        case["template"], None)


def test_light_batched_shadow_march_equals_per_light_copies():
    """hand_teacher_mvp.py:271-358: L lights per frame.  The reference expands primitives and template L times and pads
    the opacity template with a constant colour; shadow_march shares them (alpha-only template).  Same shadow grid."""
    from goliath_amd import mvp

    B, L, H, W, K, T = 2, 3, 40, 36, 27, (4, 6, 5)
    case = _random_case(B, H, W, K, T, seed=9)
    c = lambda t: t.cuda().contiguous()
    g = torch.Generator().manual_seed(3)
    # light cameras: per (frame, light) a viewpoint looking at the volume
    viewpos = (torch.tensor([0.1, -0.2, -3.0]) + 0.6 * torch.randn(B * L, 3, generator=g)).contiguous()
    viewrot = torch.eye(3)[None].repeat(B * L, 1, 1).contiguous()
    focal, princpt = torch.full((B * L, 2), 1.6 * W), torch.tensor([[W * 0.5, H * 0.5]] * (B * L))
    rp, rd, tm = mvp.compute_raydirs(c(viewpos), c(viewrot), c(focal), c(princpt), (W, H), 1.0)
    alpha = case["template"][..., 3:4].contiguous()                                   # [B,K,TD,TH,TW,1]
    prims = (c(case["primpos"]), c(case["primrot"]), c(case["primscale"]))
    shadow, img = mvp.shadow_march(rp, rd, case["step"], tm, prims, c(alpha), L, fadescale=6.5, fadeexp=8.0,
                                   return_image=True)
    # the reference's formulation: everything repeated per light, rgb = 255, through the ordinary march
    rep = lambda t: c(t[:, None].expand(-1, L, *t.shape[1:]).reshape(B * L, *t.shape[1:]))
    tpl = torch.cat([torch.full_like(alpha.expand(-1, -1, -1, -1, -1, 3), 255.0), alpha], -1)
    with torch.no_grad():
        ref_img, ref_shadow = mvp.mvpraymarch(rp, rd, case["step"], tm, (rep(case["primpos"]), rep(case["primrot"]),
                                                                         rep(case["primscale"])), rep(tpl), None,
                                              fadescale=6.5, fadeexp=8.0, with_shadow=True)
    assert shadow.shape == ref_shadow.shape == (B * L, K, *T, 1)
    assert float(ref_shadow.max()) > 0.1
    assert rel_l2(shadow, ref_shadow) < 1e-6, rel_l2(shadow, ref_shadow)   # measured 7e-8
    assert rel_l2(img[..., 3], ref_img[..., 3]) < 1e-6
    with pytest.raises(ValueError):
        mvp.shadow_march(rp, rd, case["step"], tm, prims, c(alpha), L + 1)
