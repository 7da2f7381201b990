"""GPU parity: fused HIP shading tail (C ABI gol_shade_fwd/bwd) vs
  (a) golden vectors produced by the reference's own PyTorch code (tests/golden/shade_golden.npz), and
  (b) the torch oracle (oracle/shade_ref.py, itself pinned to (a)) at a larger size.
Tolerance rel-L2 <= 1e-4 on every output and every gradient."""
import pytest
import torch
import torch.nn.functional as F

from scenes import rel_l2
from test_oracle_shade import OracleSG, load_golden, run_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _hip_from_golden(G, tag, n_keep=None):
    from goliath_amd import shade

    env, train = tag.startswith("env"), tag.endswith("train")
    cut = (lambda t: t.reshape(*t.shape[:2], -1)[:, :, :n_keep, None]) if n_keep else (lambda t: t)
    leaf = {n: cut(G[f"in/{n}"]).cuda().contiguous().requires_grad_(True) for n in ("f_vnocond", "f_vcond", "postex", "tn_raw")}
    alb = G["in/albedo"][:, :n_keep] if n_keep else G["in/albedo"]
    leaf["albedo"] = alb.cuda().contiguous().requires_grad_(True)
    kw = {}
    if env:
        kw.update(preconv_envmap=[G[f"in/mip{i}"].cuda() for i in range(4)], lightrot=G["in/lightrot"].cuda())
    else:
        kw.update(light_intensity=G["in/light_intensity"].cuda(), headrel_light_pos=G["in/light_pos"].cuda(),
                  n_lights=G["in/n_lights"].cuda())
    if train:
        kw["light_sh_rand"] = G[f"{tag}/in/light_sh_rand"][:, None, :].expand(-1, 3, -1).contiguous().cuda()
    out = shade.shading_tail(leaf["f_vnocond"], leaf["f_vcond"], leaf["postex"], F.normalize(leaf["tn_raw"], dim=1),
                             leaf["albedo"], G["in/light_sh"].cuda(), G["in/campos"].cuda(), **kw)
    return leaf, out


@pytest.mark.parametrize("tag", ["sg_eval", "sg_train", "env_eval"])
@pytest.mark.parametrize("n_keep", [None, 62, 61])  # 64 Gaussians -> 16-byte plane loads (N % 4 == 0); 62, 61 -> the scalar-plane path, one / three Gaussians short of a lane's four
def test_shade_matches_reference_golden(tag, n_keep):
    G = load_golden()
    leaf, out = _hip_from_golden(G, tag, n_keep)
    loss = 0.0
    for k, v in out.items():
        ref = G[f"{tag}/out/{k}"]
        if n_keep:
            ref = ref[:, :n_keep]
        assert rel_l2(v.reshape(ref.shape), ref) < TOL, (k, rel_l2(v.reshape(ref.shape), ref))
        w = G[f"w/{k}"][:, :n_keep] if n_keep else G[f"w/{k}"]
        loss = loss + (v.reshape(ref.shape) * w.cuda()).sum()
    loss.backward()
    for n, t in leaf.items():
        ref = G[f"{tag}/grad/{n}"]
        if n_keep:
            ref = ref.reshape(*ref.shape[:2], -1)[:, :, :n_keep, None] if n != "albedo" else ref[:, :n_keep]
        assert rel_l2(t.grad.reshape(ref.shape), ref) < TOL, (n, rel_l2(t.grad.reshape(ref.shape), ref))


@pytest.mark.parametrize("env", [False, True])
def test_shade_vs_oracle_4096(env):
    from goliath_amd import shade
    from oracle import shade_ref

    gen = torch.Generator().manual_seed(31 + env)
    B, S, L = 2, 64, 6
    N = S * S
    r = lambda *s: torch.randn(*s, generator=gen)
    inp = dict(f_vnocond=0.3 * r(B, 125, S, S), f_vcond=0.3 * r(B, 4, S, S), postex=60 * r(B, 3, S, S),
               tn=F.normalize(r(B, 3, S, S), dim=1), albedo=0.2 + 0.6 * torch.rand(1, N, 3, generator=gen))
    inp["f_vnocond"][:, 113 + 7: 113 + 10] *= 8  # exercise the primscale clamp and the softplus tail
    light_sh, light_sh_rand = r(B, 3, 81) * 0.3, r(B, 3, 81) * 0.3
    campos = torch.tensor([[30.0, -40.0, -900.0], [-500.0, 100.0, -700.0]])
    li, lp = torch.rand(B, L, 3, generator=gen), F.normalize(r(B, L, 3), dim=-1) * 1100
    nl = torch.tensor([L, 2])
    mips = [torch.rand(B, 3, 64 >> i, 128 >> i, generator=gen) * 1.5 for i in range(4)]
    rot = torch.linalg.qr(r(B, 3, 3))[0]

    cpu = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    gpu = {k: v.clone().cuda().requires_grad_(True) for k, v in inp.items()}
    if env:
        ref = shade_ref.shade(cpu["f_vnocond"], cpu["f_vcond"], cpu["postex"], cpu["tn"], cpu["albedo"], light_sh,
                              campos, envmips=mips, lightrot=rot, light_sh_rand=light_sh_rand)
        out = shade.shading_tail(gpu["f_vnocond"], gpu["f_vcond"], gpu["postex"], gpu["tn"], gpu["albedo"],
                                 light_sh.cuda(), campos.cuda(), preconv_envmap=[m.cuda() for m in mips],
                                 lightrot=rot.cuda(), light_sh_rand=light_sh_rand.cuda())
    else:
        ref = shade_ref.shade(cpu["f_vnocond"], cpu["f_vcond"], cpu["postex"], cpu["tn"], cpu["albedo"], light_sh,
                              campos, light_intensity=li, light_pos=lp, n_lights=nl, light_sh_rand=light_sh_rand,
                              sg_eval=OracleSG.apply)
        out = shade.shading_tail(gpu["f_vnocond"], gpu["f_vcond"], gpu["postex"], gpu["tn"], gpu["albedo"],
                                 light_sh.cuda(), campos.cuda(), light_intensity=li.cuda(),
                                 headrel_light_pos=lp.cuda(), n_lights=nl.cuda(), light_sh_rand=light_sh_rand.cuda())
    lr, lo = 0.0, 0.0
    for k in ref:
        w = torch.randn(ref[k].shape, generator=gen)
        assert rel_l2(out[k].reshape(ref[k].shape), ref[k]) < TOL, (k, rel_l2(out[k].reshape(ref[k].shape), ref[k]))
        if ref[k].requires_grad:
            lr = lr + (ref[k] * w).sum()
            lo = lo + (out[k].reshape(ref[k].shape) * w.cuda()).sum()
    lr.backward()
    lo.backward()
    for k in cpu:
        assert rel_l2(gpu[k].grad, cpu[k].grad) < TOL, (k, rel_l2(gpu[k].grad, cpu[k].grad))   # measured 4.3e-5


def test_shade_full_size_linearity_in_light():
    """BASELINE size (B=2 of 250k): diffuse is linear in the light SH; zero light -> colour == specular."""
    from goliath_amd import shade

    gen = torch.Generator().manual_seed(5)
    B, S = 2, 500
    N = S * S
    f_vn = (0.3 * torch.randn(B, 125, S, S, generator=gen)).cuda()
    f_vc = (0.3 * torch.randn(B, 4, S, S, generator=gen)).cuda()
    postex = (60 * torch.randn(B, 3, S, S, generator=gen)).cuda()
    tn = F.normalize(torch.randn(B, 3, S, S, generator=gen), dim=1).cuda()
    alb = (0.2 + 0.6 * torch.rand(1, N, 3, generator=gen)).cuda()
    lsh = (0.3 * torch.randn(B, 3, 81, generator=gen)).cuda()
    cam = torch.tensor([[0.0, 0.0, -900.0]] * B).cuda()
    li, lp, nl = torch.rand(B, 4, 3).cuda(), (F.normalize(torch.randn(B, 4, 3), dim=-1) * 1100).cuda(), torch.tensor([4, 4]).cuda()
    a = shade.shading_tail(f_vn, f_vc, postex, tn, alb, lsh, cam, light_intensity=li, headrel_light_pos=lp, n_lights=nl)
    b = shade.shading_tail(f_vn, f_vc, postex, tn, alb, 3 * lsh, cam, light_intensity=li, headrel_light_pos=lp, n_lights=nl)
    z = shade.shading_tail(f_vn, f_vc, postex, tn, alb, 0 * lsh, cam, light_intensity=li, headrel_light_pos=lp, n_lights=nl)
    assert rel_l2(b["diff_color"], 3 * a["diff_color"]) < 1e-6
    assert torch.equal(b["primpos"], a["primpos"]) and torch.equal(b["spec_color"], a["spec_color"])
    assert rel_l2(z["color"], z["spec_color"].clamp(min=0)) < 1e-6
    assert rel_l2(a["primqvec"].norm(dim=-1), torch.ones(B, N)) < 1e-6


def test_packed_envmap_cache_and_its_refresh():
    """shade.pack_envmap caches the footprint records of a level on (address, shape, version counter): a write through
    `.data` is not seen (documented caveat) until refresh=True / invalidate_envmap_cache()."""
    from goliath_amd import shade

    m = [torch.rand(1, 3, 8, 16, device="cuda")]
    a = shade.pack_envmap(m)[0].clone()
    m[0].data.mul_(2.0)                                   # the version counter of m[0] does not move
    assert torch.equal(shade.pack_envmap(m)[0], a)        # stale: the caveat
    fresh = shade.pack_envmap(m, refresh=True)[0]
    assert torch.allclose(fresh, 2 * a)
    assert torch.equal(shade.pack_envmap(m)[0], fresh)    # the refreshed copy replaced the cache entry
    m[0].mul_(0.5)                                        # an ordinary in-place op IS seen
    assert torch.allclose(shade.pack_envmap(m)[0], a)
    shade.invalidate_envmap_cache()
    assert torch.allclose(shade.pack_envmap(m)[0], a)


@pytest.mark.parametrize("form", ["one_map", "expanded_view"])
@pytest.mark.parametrize("fused_projection", [False, True])
def test_shared_env_pyramid_equals_per_view_copies(form, fused_projection):
    """BASELINE config 2 / light_decorator.py:96-100: ONE pyramid lights every view (gol_shade_in.mips_shared).  Handing the
    levels over as [1,3,h,w], or as the stride-0 `expand(B, ...)` views dropin.patch_light_decorator returns, must give
    bit-identical outputs and gradients to B materialised copies, with and without the projection fused in; env + the
    training-only random light (shade_fwd_kernel<.., ENV, RAND, ..>) included."""
    from goliath_amd import render_gs, shade

    B, S = 3, 24
    N = S * S
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g)
    base = dict(f_vn=0.3 * r(B, 125, S, S), f_vc=0.3 * r(B, 4, S, S), postex=60.0 * r(B, 3, S, S), tn=F.normalize(r(B, 3, S, S), dim=1),
                albedo=0.2 + 0.6 * torch.rand(1, N, 3, generator=g), light_sh=0.3 * r(B, 3, 81), campos=torch.tensor([[30.0, -40.0, -900.0]]).repeat(B, 1),
                lightrot=torch.linalg.qr(r(B, 3, 3))[0], rand=0.3 * r(B, 3, 81))
    one = [torch.rand(1, 3, 16 >> i, 32 >> i, generator=g) * 1.6 for i in range(3)]
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = 300.0
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = 48.0, 64.0, 1.0
    Rt = torch.eye(3, 4)[None].repeat(B, 1, 1)
    Rt[:, 2, 3] = 900.0

    def run(mips):
        leaf = {k: base[k].clone().cuda().requires_grad_(True) for k in ("f_vn", "f_vc", "postex", "tn", "albedo")}
        vs = render_gs.view_set(K.cuda(), Rt.cuda(), 128, 96) if fused_projection else None
        out = shade.shading_tail(leaf["f_vn"], leaf["f_vc"], leaf["postex"], leaf["tn"], leaf["albedo"], base["light_sh"].cuda(),
                                 base["campos"].cuda(), preconv_envmap=mips, lightrot=base["lightrot"].cuda(),
                                 light_sh_rand=base["rand"].cuda(), views=vs)
        w = torch.Generator().manual_seed(9)
        loss = sum((v * torch.randn(v.shape, generator=w).cuda()).sum() for k, v in sorted(out.items()) if torch.is_tensor(v) and v.requires_grad)
        if fused_projection:
            loss = loss + (out["projected"].records * torch.randn(out["projected"].records.shape, generator=w).cuda()).sum()
        loss.backward()
        return {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}, {k: v.grad for k, v in leaf.items()}

    copies = [m.cuda().expand(B, -1, -1, -1).contiguous() for m in one]
    shared = [m.cuda() for m in one] if form == "one_map" else [m.cuda().expand(B, -1, -1, -1) for m in one]
    o1, g1 = run(copies)
    o2, g2 = run(shared)
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    with pytest.raises(ValueError):      # neither one map nor one per view
        run([m.cuda().expand(2, -1, -1, -1).contiguous() for m in one])
    with pytest.raises(ValueError):      # a stride-0 view expanded over ANOTHER batch size is not "one map for all"
        run([m.cuda().expand(2, -1, -1, -1) for m in one])


def test_frame_scale_goes_to_the_kernel_and_nothing_is_repacked():
    """ADVICE r5 / light_decorator.py:147-149: the relight driver scales the registered pyramid by 2 pi norm_scale[0] of the
    FRAME.  dropin._shared_mipmap hands each level over with its unscaled buffer and the scale; shade packs the buffer's
    footprint records once (cache keyed on the buffer) and passes the scale to the kernel (gol_shade_in.mips_scale): the
    results equal those of the scaled copies (interpolation is linear: rounding only), for every frame, with ONE packed
    pyramid in the cache."""
    from goliath_amd import dropin, shade

    B, S = 2, 16
    N = S * S
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g)
    t = dict(f_vn=0.3 * r(B, 125, S, S), f_vc=0.3 * r(B, 4, S, S), postex=60.0 * r(B, 3, S, S), tn=F.normalize(r(B, 3, S, S), dim=1),
             albedo=0.2 + 0.6 * torch.rand(1, N, 3, generator=g), light_sh=0.3 * r(B, 3, 81),
             campos=torch.tensor([[30.0, -40.0, -900.0]]).repeat(B, 1), lightrot=torch.linalg.qr(r(B, 3, 3))[0])

    class Deco:                                          # the attributes EnvSpinDecorator.mipmap reads
        miplevel = 3

    d = Deco()
    for i in range(3):
        setattr(d, f"mipmap_{i}", (torch.rand(1, 3, 16 >> i, 32 >> i, generator=g) * 0.4).cuda())

    def run(mips):
        leaf = {k: t[k].clone().cuda().requires_grad_(True) for k in ("f_vn", "f_vc", "postex", "tn", "albedo")}
        out = shade.shading_tail(leaf["f_vn"], leaf["f_vc"], leaf["postex"], leaf["tn"], leaf["albedo"], t["light_sh"].cuda(),
                                 t["campos"].cuda(), preconv_envmap=mips, lightrot=t["lightrot"].cuda())
        w = torch.Generator().manual_seed(9)
        sum((v * torch.randn(v.shape, generator=w).cuda()).sum() for k, v in sorted(out.items()) if torch.is_tensor(v) and v.requires_grad).backward()
        return {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}, {k: v.grad for k, v in leaf.items()}

    shade.invalidate_envmap_cache()
    for scale in (1.0, 2.37, 5.1):
        levels = dropin._shared_mipmap(d, B, torch.device("cuda"), scale)
        assert all(m.stride(0) == 0 and m._gol_base is getattr(d, f"mipmap_{i}") for i, m in enumerate(levels))
        o1, g1 = run(levels)
        assert len(shade._PACKED) == 3, len(shade._PACKED)              # the three levels of the unscaled buffer, once
        o2, g2 = run([(getattr(d, f"mipmap_{i}") * scale).expand(B, -1, -1, -1).contiguous() for i in range(3)])
        shade._PACKED = {k: v for k, v in shade._PACKED.items() if k[1][0] == 1}   # forget the B-copy pyramids just packed
        for k in o1:
            assert rel_l2(o1[k], o2[k]) < 1e-6, (scale, k, rel_l2(o1[k], o2[k]))
        for k in g1:
            assert rel_l2(g1[k], g2[k]) < 2e-6, (scale, k, rel_l2(g1[k], g2[k]))
    with pytest.raises(ValueError):      # lightrot must be a rotation (gol_shade_in.lightrot precondition)
        t["lightrot"] = 1.3 * t["lightrot"]
        run([getattr(d, f"mipmap_{i}") for i in range(3)])
