"""GPU: the model-level bindings of BASELINE configs 4 / 5 against goldens produced by the REFERENCE's own model code.

tests/golden/urhand_model_golden.npz holds what `ConvTeacherDecoder.forward` (ca_code/models/urhand.py:349-630) and
`OLATRGBDecoder.forward_rgb` (ca_code/models/hand_teacher_mvp.py:253-494) return -- the reference methods themselves,
run on the CPU on the seeded stand-in modules of tests/urhand_shaped.py (tests/golden/make_urhand_model_golden.py).  Here the
drop-ins of goliath_amd/urhand.py run on identical modules on the GPU: gol_mesh_raster + gol_shadow_pcf +
gol_uvlight_phong/ggx for URHand, gol_raydirs + gol_mvp_shadow_march for the teacher's deep shadow -- which is thereby
pinned against oracle/mvp_oracle.c in shadow mode (the golden's march), not against another launch of itself."""
import os

import numpy as np
import pytest
import torch

import urhand_shaped as S
from scenes import rel_l2

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "urhand_model_golden.npz"))


def _cuda(d):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}


@pytest.mark.parametrize("tag,depth", [("eval", "golden"), ("train", "golden"), ("eval", "hip")])
def test_conv_teacher_decoder_forward_matches_the_reference_forward(tag, depth):
    """depth = "golden": the light cameras' depth images are the golden's (the numpy z-buffer of oracle/mesh_ref.py), so
    everything downstream -- PCF, both light loops, the binding -- is compared at rounding level.  depth = "hip": the
    whole path incl. gol_mesh_raster, whose depth images are compared with the golden's first (same coverage, depth to
    2e-7 since the barycentrics are evaluated relative to a vertex of the face; in absolute pixel coordinates they were
    1.7e-4 off, 0.4 % of the shadow map)."""
    from goliath_amd import meshraster, urhand

    dec = S.ShapedConvTeacherDecoder(seed=0)
    gf = dec.geo_fn
    if depth == "hip":
        # a foreign render layer with the reference's attributes (what drtk's RenderLayer looks like to the binding): the
        # depth images must come from gol_mesh_raster off its topology, the layer itself must never be called
        class ForeignLayer(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.h = self.w = 512
                self.register_buffer("vi", gf.vi.int(), persistent=False)

            def forward(self, *a, **k):
                raise AssertionError("the shadow path called the model's render layer")

        dec.rl = ForeignLayer()
    dec = dec.cuda().train(tag == "train")
    if depth == "golden":
        dec.rl = S.ReplayRenderLayer(512, 512, [torch.from_numpy(GOLD[f"urhand_{tag}_depth{i}"]) for i in range(2)])
    loose = 1.0
    inp = _cuda(S.urhand_inputs(B=1, L=2))
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("verts_rec", "tex_mean")}
    rendered, orig_rasterize = [], meshraster.rasterize
    meshraster.rasterize = lambda *a, **k: (lambda out: (rendered.append(out[1]), out)[1])(orig_rasterize(*a, **k))
    try:
        res = urhand.conv_teacher_decoder_forward(dec, **{**inp, **leaves})
    finally:
        meshraster.rasterize = orig_rasterize
    if depth == "hip":   # the depth renders themselves: same coverage except at silhouettes, same depth where both cover
        assert len(rendered) == 2
        for i, d in enumerate(rendered):
            want = torch.from_numpy(GOLD[f"urhand_{tag}_depth{i}"])
            d = d.cpu()
            cov_h, cov_o = d > 0, want > 0
            mismatch = float((cov_h != cov_o).float().mean())
            both = cov_h & cov_o
            depth_err = float(((d - want).abs() / want.clamp(min=1e-6))[both].max())
            print(f"\nDEPTH_RENDER {i}: coverage mismatch {mismatch:.2e} of the pixels, covered {float(cov_o.float().mean()):.3f}, "
                  f"max relative depth difference where both cover {depth_err:.2e}")
            assert mismatch < 2e-4 and depth_err < 2e-6
    tol = {}
    worst = {}
    for k in ("tex", "phys_tex", "diff_feature", "spec_feature", "diff_feature_raw", "spec_feature_raw", "shadow",
              "shadow_raw", "feature_normal", "verts_displaced", "displacement", "roughness", "id_pose_conv"):
        want = torch.from_numpy(GOLD[f"urhand_{tag}_{k}"])
        assert tuple(res[k].shape) == tuple(want.shape), (k, res[k].shape, want.shape)
        worst[k] = rel_l2(res[k].detach().cpu(), want)
    print("\nURHAND_MODEL outputs", tag, depth, {k: float("%.2e" % v) for k, v in worst.items()})
    for k in list(worst):
        assert worst[k] < loose * tol.get(k, 3e-5), (k, worst[k])   # measured <= 3.3e-6 (profiles/r04_parity_ledger.json)
    g = torch.Generator().manual_seed(5)
    loss = sum((res[k] * torch.randn(res[k].shape, generator=g).cuda()).sum() for k in ("tex", "phys_tex", "diff_feature_raw"))
    loss.backward()
    for k, v in leaves.items():
        e = rel_l2(v.grad.cpu(), torch.from_numpy(GOLD[f"urhand_{tag}_grad_{k}"]))
        worst["grad_" + k] = e
        assert e < loose * 3e-5, (k, e)   # measured <= 3.4e-6
    params = dict(dec.named_parameters())
    for n in ("global_scale", "global_albedo_scale", "geo_refiner.geo.weight", "texmod1.0.weight"):
        e = rel_l2(params[n].grad.cpu(), torch.from_numpy(GOLD[f"urhand_{tag}_grad_{n}"]))
        worst["grad_" + n] = e
        assert e < loose * 3e-5, (n, e)   # measured <= 2.6e-6
    print("\nURHAND_MODEL", tag, depth, {k: float("%.2e" % v) for k, v in worst.items()})


def test_olat_rgb_decoder_forward_rgb_matches_the_reference_forward_rgb():
    from goliath_amd import mvp, urhand

    dec = S.ShapedOLATRGBDecoder(mvp.Raymarcher(200.0), 200.0, seed=0).cuda().eval()
    captured = {}
    dec.enc_layers[0].register_forward_pre_hook(lambda m, a: captured.__setitem__("x", a[0].detach().clone()))
    inp = _cuda(S.teacher_inputs(B=1, L=2))
    with torch.no_grad():
        res = urhand.olat_rgb_decoder_forward_rgb(dec, **inp)
    x, want = captured["x"].cpu(), torch.from_numpy(GOLD["teacher_unet_input"])
    assert tuple(x.shape) == tuple(want.shape)
    # channels: light directions [0, 3Z), view directions [3Z, 6Z), 1 - deep shadow [6Z, 7Z)
    Z = 2
    e_dirs = rel_l2(x[:, :6 * Z], want[:, :6 * Z])
    e_shadow = rel_l2(x[:, 6 * Z:], want[:, 6 * Z:])
    e_rgb = rel_l2(res["primrgb"].cpu(), torch.from_numpy(GOLD["teacher_primrgb"]))
    e_ps = rel_l2(res["primshadow"].cpu(), torch.from_numpy(GOLD["teacher_primshadow"]))
    print("\nTEACHER_MODEL", dict(dirs=e_dirs, deep_shadow=e_shadow, primrgb=e_rgb, primshadow=e_ps))
    # measured: 3.9e-8 / 3.7e-5 (the deep-shadow march accumulates over ~100 samples of exp()) / 8.3e-8 / 2.7e-6
    assert e_dirs < 1e-6 and e_shadow < 1e-4 and e_rgb < 2e-6 and e_ps < 3e-5
