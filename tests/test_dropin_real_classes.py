"""CPU: the drop-in against the REAL reference classes (VERDICT r1 "next" #10).

With stand-ins for the third-party imports this image lacks (tests/golden/ref_stubs.py), `ca_code.models.rgca`,
`ca_code.utils.shadowmap` and `ca_code.loss` import unchanged from /root/reference.  Checked here:
  * dropin.patch_rgca() / patch_urhand() / patch_losses() replace PrimDecoder.forward, AutoEncoder.render,
    get_shadow_map, rgb_l1, rgb_ssim by callables with the SAME parameter lists (names, order, defaults, kinds) --
    `filter_inputs` (ca_code/utils/train.py:99-116) selects batch entries by introspection, so names matter;
  * a real `PrimDecoder` (reference architecture, 162.8 M parameters) qualifies for the fused decoder tail, its
    state_dict loads strictly into goliath_amd.decoder.PrimDecoderConvs and back, and the light-contracted tail
    (goliath_amd/tail.py) applied to the REFERENCE layer's parameters reproduces `layer(x)` contracted with the light
    at the native 1024^2 slab;
  * rgb_l1(mask_erode=...) == the reference's rgb_l1 on CPU-computable pieces (the erosion)."""
import inspect
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))


def _params(fn):
    return [(p.name, p.kind, p.default) for p in inspect.signature(fn).parameters.values()]


@pytest.fixture(scope="module")
def ref():
    import types

    import ref_stubs

    ref_stubs.install()
    sys.modules.setdefault("sgutilslib", types.ModuleType("sgutilslib"))
    import ca_code.loss as L
    import ca_code.models.rgca as R
    import ca_code.utils.shadowmap as SM

    return types.SimpleNamespace(R=R, SM=SM, L=L)


def test_patched_entry_points_keep_the_reference_signatures(ref):
    from goliath_amd import dropin, losses, shadowmap

    R = ref.R
    orig_fwd, orig_render, orig_ae = R.PrimDecoder.forward, R.AutoEncoder.render, R.AutoEncoder.forward
    try:
        assert dropin.patch_rgca(R) is R
        assert R.PrimDecoder.forward is not orig_fwd and R.AutoEncoder.render is not orig_render
        assert R.AutoEncoder.forward is not orig_ae
        assert _params(R.PrimDecoder.forward) == _params(orig_fwd)
        assert _params(R.AutoEncoder.render) == _params(orig_render)
        assert _params(R.AutoEncoder.forward) == _params(orig_ae)  # what filter_inputs introspects
        assert inspect.signature(R.AutoEncoder.forward).return_annotation == inspect.signature(orig_ae).return_annotation
    finally:
        R.PrimDecoder.forward, R.AutoEncoder.render, R.AutoEncoder.forward = orig_fwd, orig_render, orig_ae
    assert _params(shadowmap.get_shadow_map) == _params(ref.SM.get_shadow_map)
    assert _params(losses.rgb_l1) == _params(ref.L.rgb_l1)
    assert _params(losses.rgb_ssim) == _params(ref.L.rgb_ssim)


def test_loss_registry_patch_on_the_real_registry(ref):
    from goliath_amd import dropin, losses

    import ca_code.loss.registry as reg

    saved = dict(reg.loss_registry)
    try:
        dropin.patch_losses(reg)
        mod = reg.loss_registry["rgb_l1"](None, src_key="rgb", mask_erode=5)
        assert isinstance(mod, reg.FnLoss) and mod.fn is losses.rgb_l1 and mod.extra_args["mask_erode"] == 5
        assert reg.loss_registry["rgb_ssim"](None).fn is losses.rgb_ssim
        assert set(reg.loss_registry) == set(saved)  # nothing else touched
    finally:
        reg.loss_registry.clear()
        reg.loss_registry.update(saved)


def test_mask_erode_matches_reference_erode(ref):
    from goliath_amd import losses

    from ca_code.utils.image import erode

    g = torch.Generator().manual_seed(0)
    mask = (torch.rand(2, 1, 40, 33, generator=g) > 0.15).float()
    for ks in (3, 5, 9):
        assert torch.equal(losses.erode_mask(mask, ks), erode(mask, ks))
    assert torch.equal(losses.erode_mask(mask[:, 0] > 0, 3)[:, 0] > 0, erode(mask[:, 0] > 0, 3)[:, 0])


def test_real_prim_decoder_takes_the_fused_tail_branch(ref):
    from goliath_amd import decoder, rgca, tail

    torch.manual_seed(0)
    dec = ref.R.PrimDecoder(256, None, torch.rand(3, 1024, 1024) * 255.0)
    with torch.no_grad():  # a "trained" state: non-trivial untied bias and magnitudes
        for m in (dec.vnocond_mod[-1], dec.vcond_mod[-1]):
            m.bias.normal_(0.0, 0.1)
            m.weight_g.mul_(1.0 + 0.2 * torch.rand_like(m.weight_g))
    assert rgca._can_fuse_tail(dec)
    # state_dict: reference -> ours -> reference, strict both ways
    sd = dec.state_dict()
    mine = decoder.PrimDecoderConvs()
    missing = mine.load_state_dict({k: v for k, v in sd.items() if k != "albedo"}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    back = dict(mine.state_dict(), albedo=sd["albedo"])
    dec.load_state_dict(back, strict=True)
    # the light-contracted tail on the reference layer's own parameters == the reference layer, then the contraction
    last = dec.vnocond_mod[-1]
    ncol, nmono = dec.n_color_sh_coeffs, dec.n_mono_sh_coeffs
    nd = 3 * ncol + nmono
    x = torch.randn(1, 16, 512, 512)
    L = torch.randn(1, 3, ncol + nmono)
    with torch.no_grad():
        f = last(x)                                                   # [1, 125, 1024, 1024], the reference's layer
        assert torch.allclose(mine.vnocond_mod[-1](x), f, atol=1e-5)  # our module with the loaded state: same output
        Lc = tail.light_matrix(L, ncol, nmono)                       # rgca.py:506-514 + :528-530 as one matrix
        want = torch.cat([torch.einsum("bek,bkn->ben", Lc, f[:, :nd].reshape(1, nd, -1)), f[:, nd:].reshape(1, 12, -1)], 1)
        got, E = tail.contracted_vnocond_torch(x, tail.wn_weight(last), last.bias, L, None, ncol, nmono)
    assert E == 3
    err = float((got.reshape(1, 15, -1) - want).norm() / want.norm())
    assert err < 1e-5, err


@pytest.fixture(scope="module")
def ref_hands():
    """ca_code.models.urhand / hand_teacher_mvp, imported unchanged (their native / third-party imports stubbed)."""
    import types

    import ref_stubs

    ref_stubs.install()
    for n in ("sgutilslib", "utilslib", "mvpraymarchlib"):
        sys.modules.setdefault(n, types.ModuleType(n))
    import ca_code.models.hand_teacher_mvp as T
    import ca_code.models.urhand as U

    return types.SimpleNamespace(U=U, T=T)


def test_urhand_and_teacher_patches_keep_the_reference_signatures(ref_hands):
    """BASELINE configs 4 / 5 as drop-ins: ConvTeacherDecoder.forward (urhand.py:349-630) and OLATRGBDecoder.forward_rgb
    (hand_teacher_mvp.py:253-494) are replaced by callables with the same parameter lists; the module-level names the
    model constructors resolve: get_shadow_map points at the HIP-backed version, RenderLayer only on request."""
    from goliath_amd import dropin, meshraster, shadowmap, urhand

    U, T = ref_hands.U, ref_hands.T
    saved = (U.ConvTeacherDecoder.forward, U.get_shadow_map, U.RenderLayer, T.OLATRGBDecoder.forward_rgb)
    try:
        assert dropin.patch_urhand(U) is U and dropin.patch_hand_teacher(T) is T
        assert U.ConvTeacherDecoder.forward is urhand.conv_teacher_decoder_forward
        assert T.OLATRGBDecoder.forward_rgb is urhand.olat_rgb_decoder_forward_rgb
        assert _params(U.ConvTeacherDecoder.forward) == _params(saved[0])
        assert _params(T.OLATRGBDecoder.forward_rgb) == _params(saved[3])
        # the name AutoEncoder.__init__ resolves for its final textured render (self.renderer, urhand.py:684) is untouched:
        # training through the drop-in keeps drtk's edge gradients; the opt-in swap is forward-only
        assert U.get_shadow_map is shadowmap.get_shadow_map and U.RenderLayer is saved[2]
        src = inspect.getsource(U.AutoEncoder.__init__)
        assert "self.renderer = RenderLayer(" in src    # ... and that IS the name the constructor uses
        assert dropin.patch_urhand(U, mesh_render_layer=True).RenderLayer is meshraster.RenderLayer
        assert _params(meshraster.RenderLayer.__init__) == _params(saved[2].__init__)
        assert _params(meshraster.RenderLayer.forward) == _params(saved[2].forward)
    finally:
        U.ConvTeacherDecoder.forward, U.get_shadow_map, U.RenderLayer, T.OLATRGBDecoder.forward_rgb = saved


def test_helper_restatements_of_the_shaped_stand_ins_equal_the_reference_helpers(ref_hands):
    """tests/urhand_shaped.py restates the geometry helpers the reference forwards call, so that the GPU tests can run the
    drop-ins where /root/reference does not exist; each restatement must reproduce the reference function."""
    sys.path.insert(0, os.path.dirname(__file__))
    import urhand_shaped as S

    U, T = ref_hands.U, ref_hands.T
    g = torch.Generator().manual_seed(0)
    geo = S.FakeGeo(32, 4)
    v = torch.randn(2, 25, 3, generator=g)
    assert torch.allclose(S.vert_normals(v, geo.vi), U.vert_normals(v, geo.vi), atol=1e-6)
    idx = torch.randint(0, 25, (7, 3), generator=g)
    assert torch.equal(S.index(v[0], idx, 0), U.index(v[0], idx, 0))
    tri_xyz, tri_uv = torch.randn(2, 9, 3, 3, generator=g), torch.rand(9, 3, 2, generator=g)
    n = torch.nn.functional.normalize(torch.randn(2, 9, 3, generator=g), dim=-1)
    for a, b in zip(S.compute_tbn_uv_given_normal(tri_xyz, tri_uv, n), U.compute_tbn_uv_given_normal(tri_xyz, tri_uv, n)):
        assert torch.allclose(a, b, atol=1e-6)
    xyz = torch.randn(2, 3, 12, 9, generator=g)
    assert torch.allclose(S.xyz2normals(xyz), U.xyz2normals(xyz), atol=1e-6)
    x = torch.randn(3, 5, generator=g)
    assert torch.equal(S.tile2d(x, 4), U.tile2d(x, 4))
    cam, ctr = torch.randn(4, 3, generator=g) * 500, torch.randn(4, 3, generator=g) * 10
    for ref_fn in (U.build_cam_rot_mat, T.build_cam_rot_mat):
        assert torch.allclose(S.build_cam_rot_mat(cam.clone(), ctr), ref_fn(cam.clone(), ctr), atol=1e-6)
