"""GPU: the model boundary north_star names -- `AutoEncoder.forward` / `AutoEncoder.render` / `PrimDecoder.forward`
(ca_code/models/rgca.py:112-253, 466-620) and the env-relight driver's hand-over (ca_code/utils/light_decorator.py:96-164) --
against a fixture produced by the REFERENCE's own code.

tests/golden/rgca_model_golden.npz holds what the reference's methods return (and the gradients of a fixed random scalar)
when run unbound on the seeded stand-in of tests/rgca_shaped.py on the CPU, with gsplat served by the C oracle and sgutilslib
by the reference's sg.cu compiled for the host (tests/golden/make_rgca_model_golden.py).  Here the functions
`goliath_amd.dropin.patch_rgca()` installs (tests/test_dropin_real_classes.py checks on the real classes that these are
the ones) run on an identical stand-in on the GPU: head-relative transforms, fused decoder tail + shading tail with the
projection fused in, ONE batched render, fused image tail -- every returned key and every recorded gradient is compared.

Reference Python that the model code calls but that is absent on the GPU box is REPLAYED from the fixture, not restated:
`sh.dir2sh_torch` (the recorded coefficients of the recorded directions), the device's `torch.rand` for the training-only
random light (the recorded draw), and `compose_envmap` (the affine map it is, from three recorded probe calls).
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

import rgca_shaped as S
from scenes import rel_l2

pytestmark = pytest.mark.gpu
GOLD_PATH = os.path.join(os.path.dirname(__file__), "golden", "rgca_model_golden.npz")

# rel-L2 bars (north_star: 1e-4).  Measured (profiles/r05_rgca_model_parity.json): per-Gaussian outputs <= 9.9e-6 (SG specular
# of train_point; everything else <= 1.2e-6), images <= 1.2e-5, every gradient <= 7.7e-5.
# The fixture is built so that the comparison measures arithmetic, not which side of a discontinuity a rounding lands on
# (tests/golden/make_rgca_model_golden.py; each was found by bisecting a first, failing version of this test):
#   * no two Gaussians sharing a tile within 32 ulps in depth (their compositing order hinged on the last bit of
#     Rt @ head_pose: one swapped pair = 7e-4 on rgb, 4e-4 on every gradient);
#   * no LeakyReLU pre-activation of the decoder within 4e-6 of zero (one of 32768 at 4e-7 = 1.6e-3 on a weight gradient);
#   * SG lobes of sigma >= 0.05 (at the 0.01 floor two fp32 evaluations of exp(-angle^2 / 2 sigma^2) differ by 1e-3);
#   * the weight-norm denominator of the generator in fp64 (torch's CPU fp32 norm of 4 M elements is 9.5e-5 off; the GPU's is
#     not), and the camera / light tensors that come out of LAPACK / BLAS stored in the fixture.
BAR_PER_GAUSSIAN, BAR_IMAGE, BAR_GRAD = 3e-5, 5e-5, 1e-4  # gradients: north_star's bar (measured 7.7e-5)


def _gold():
    return np.load(GOLD_PATH)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class _Replay:
    """Context: `ca_code.utils.sh.dir2sh_torch`, `ca_code.utils.envmap.compose_envmap` and `torch.rand` replayed from the
    fixture for one case (`tag`)."""

    def __init__(self, G, tag, compose=False):
        self.G, self.tag = G, tag
        self.sh = []
        i = 0
        while f"{tag}/sh{i}/dirs" in G:
            self.sh.append((_t(G[f"{tag}/sh{i}/dirs"]), _t(G[f"{tag}/sh{i}/coeffs"])))
            i += 1
        self.rand = []
        i = 0
        while f"{tag}/rand{i}" in G:
            self.rand.append(_t(G[f"{tag}/rand{i}"]))
            i += 1
        self.compose = None
        if compose:
            self.compose = tuple(_t(G[f"{tag}/compose/{k}"]).cuda() for k in ("one_minus_ma", "bg_term", "mirror_term"))
        self.sh_calls = self.rand_calls = 0

    def dir2sh_torch(self, n, d):
        for dirs, coeffs in self.sh:
            if tuple(dirs.shape) == tuple(d.shape) and float((dirs - d.detach().cpu()).abs().max()) < 2e-5:
                assert coeffs.shape[-1] == (n + 1) ** 2
                self.sh_calls += 1
                return coeffs.to(d.device)
        raise AssertionError(f"dir2sh_torch called with directions the reference never passed (shape {tuple(d.shape)})")

    def compose_envmap(self, render, alpha, envbg, K, Rt):
        one_minus_ma, bg_term, mirror_term = self.compose
        return one_minus_ma * render + (1.0 - alpha) * bg_term + mirror_term

    def __enter__(self):
        self._saved = {k: sys.modules.get(k) for k in ("ca_code", "ca_code.utils", "ca_code.utils.sh", "ca_code.utils.envmap")}
        for name in ("ca_code", "ca_code.utils"):
            sys.modules[name] = types.ModuleType(name)
        sh = types.ModuleType("ca_code.utils.sh")
        sh.dir2sh_torch = self.dir2sh_torch
        env = types.ModuleType("ca_code.utils.envmap")
        env.compose_envmap = self.compose_envmap
        sys.modules["ca_code.utils.sh"], sys.modules["ca_code.utils.envmap"] = sh, env
        sys.modules["ca_code.utils"].sh, sys.modules["ca_code.utils"].envmap = sh, env
        sys.modules["ca_code"].utils = sys.modules["ca_code.utils"]
        self._rand = torch.rand
        queue = list(self.rand)

        def rand(*a, **k):
            if not queue:
                return self._rand(*a, **k)
            r = queue.pop(0)
            shape = tuple(a[0]) if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else tuple(a)
            assert tuple(r.shape) == shape, (r.shape, shape)
            self.rand_calls += 1
            return r.to(device=k.get("device", "cpu"), dtype=k.get("dtype", torch.float32))

        torch.rand = rand
        return self

    def __exit__(self, *exc):
        torch.rand = self._rand
        for k, v in self._saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _stored(G, tag):
    """The fixture's copies of the inputs whose last bit depends on the host's BLAS / LAPACK / vector libm."""
    return {k.split("/stored/")[1]: G[k] for k in G.files if k.startswith(f"{tag}/stored/")}


def _model(G, embs, geom, cal=True, blur=True):
    """The stand-in on the GPU with what dropin.patch_rgca() installs on the reference classes bound to it."""
    from goliath_amd import rgca

    m = S.ShapedAutoEncoder(embs, geom, 0, cal=cal, blur=blur, nudges=(G["nudges/index"], G["nudges/dz"])).cuda()
    m.decoder.forward = types.MethodType(rgca.prim_decoder_forward, m.decoder)
    m.render = types.MethodType(rgca.autoencoder_render, m)
    m.forward = types.MethodType(rgca.autoencoder_forward, m)
    return m


def _cuda(d):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}


def _param(m, name):
    obj = m
    for part in name.split("."):
        obj = getattr(obj, part) if not part.isdigit() else obj[int(part)]
    return obj


def _compare_outputs(G, tag, preds, report, keys=None):
    want_keys = [k.split("/out/")[1] for k in G.files if k.startswith(f"{tag}/out/")]
    assert want_keys, tag
    for k in want_keys:
        if keys is not None and k not in keys:
            continue
        assert k in preds, f"{tag}: the reference returns `{k}`, the drop-in does not"
        want = _t(G[f"{tag}/out/{k}"])
        got = preds[k].detach().float().cpu()
        assert tuple(got.shape) == tuple(want.shape), (tag, k, tuple(got.shape), tuple(want.shape))
        report[f"out/{k}"] = rel_l2(got, want)
    extra = set(k for k, v in preds.items() if torch.is_tensor(v)) - set(want_keys)
    assert not extra or keys is not None, f"{tag}: keys the reference does not return: {sorted(extra)}"


def _backprop(preds):
    wf = S.loss_weights(0)   # the generator's cotangents, drawn in the same order on the CPU
    loss = 0.0
    for k in ("rgb", "depth", "primscale_preclip", "spec_nml", "color_rand"):
        if k in preds:
            loss = loss + (preds[k] * wf[k](preds[k]).to(preds[k].device)).sum()
    loss.backward()
    return float(loss)


def _compare_grads(G, tag, m, embs, geom, report):
    report["grad/embs"] = rel_l2(embs.grad, _t(G[f"{tag}/grad/embs"]))
    report["grad/geom"] = rel_l2(geom.grad, _t(G[f"{tag}/grad/geom"]))
    for name in S.GRAD_PARAMS:
        key = f"{tag}/grad/{name}"
        if key not in G.files:
            continue
        g = _param(m, name).grad
        assert g is not None, name
        report[f"grad/{name}"] = rel_l2(g, _t(G[key]))


def _judge(tag, report):
    print(f"\nRGCA_MODEL_GOLDEN {tag} " + " ".join(f"{k}={v:.2e}" for k, v in sorted(report.items())))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        import json

        path = os.path.join(out_dir, "rgca_model_parity.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[tag] = report
        json.dump(data, open(path, "w"), indent=1)
    for k, v in report.items():
        assert v == v, (tag, k, "NaN")
        if k.startswith("grad/"):
            bar = BAR_GRAD
        elif k in ("out/rgb", "out/alpha", "out/depth"):
            bar = BAR_IMAGE
        else:
            bar = BAR_PER_GAUSSIAN
        assert v < bar, (tag, k, v, bar)


def test_train_point_lights_every_key_and_gradient():
    """Training mode, point lights (n_lights 3 and 2), is_fully_lit_frame mixed, CalV5 + background + LearnableBlur on,
    the training-only random light (color_rand, cos_weight)."""
    G = _gold()
    st = _stored(G, "train_point")
    embs, geom = (t.detach().cuda().requires_grad_(True) for t in S.leaves(2, 0, st))
    m = _model(G, embs, geom).train()
    batch = _cuda(S.batch_inputs(2, 0, stored=st))
    with _Replay(G, "train_point") as rp:
        preds = m.forward(**batch)
        assert rp.sh_calls == 2 and rp.rand_calls == 1
    report = {}
    _compare_outputs(G, "train_point", preds, report)
    assert not preds["alpha"].requires_grad and preds["depth"].requires_grad       # rgca.py:137, 144-145
    _backprop(preds)
    _compare_grads(G, "train_point", m, embs, geom, report)
    _judge("train_point", report)


def _env_batch(G, tag, B, seed, with_envbg):
    batch = S.batch_inputs(B, seed, stored=_stored(G, tag))
    for k in ("light_intensity", "light_pos", "n_lights"):
        batch.pop(k)
    for k in ("light_intensity", "light_pos", "lightrot", "n_lights"):
        batch[k] = _t(G[f"{tag}/in/{k}"])
    if with_envbg:
        batch["envbg"] = _t(G[f"{tag}/in/envbg"])
    # what dropin.patch_light_decorator makes EnvSpinDecorator.mipmap return: stride-0 batch views of ONE scaled pyramid
    batch = _cuda(batch)
    batch["preconv_envmap"] = [_t(G[f"{tag}/in/preconv_envmap_{i}"]).cuda().expand(B, -1, -1, -1) for i in range(4)]
    # the rest of what EnvSpinDecorator.forward puts into the call (light_decorator.py:151-162); swallowed by **kwargs
    batch.update(sigma_step=0.2, light_type="envmap", is_fullylit_frame=torch.zeros(1).cuda(), index=[0] * B)
    return batch


def test_eval_env_relight_driver_inputs_every_key_and_gradient():
    """Eval, the batch EnvSpinDecorator.forward hands over (its own mipmap() pyramid x scale, lightrot, the 512 env lights for
    the SH diffuse term), calibration / blur off as run_vis_relight.py:83-84 sets them; the shared pyramid reaches the kernel
    as ONE map (gol_shade_in.mips_shared)."""
    from goliath_amd import shade

    G = _gold()
    embs, geom = (t.detach().cuda().requires_grad_(True) for t in S.leaves(2, 100, _stored(G, "eval_env")))
    m = _model(G, embs, geom).eval()
    m.learn_blur_enabled = m.cal_enabled = False
    batch = _env_batch(G, "eval_env", 2, 100, with_envbg=False)
    seen = []
    make_in = shade._make_in
    shade._make_in = lambda *a, **k: (lambda s: (seen.append(int(s.mips_shared)), s)[1])(make_in(*a, **k))
    try:
        with _Replay(G, "eval_env"):
            preds = m.forward(**batch)
            report = {}
            _compare_outputs(G, "eval_env", preds, report)
            _backprop(preds)
    finally:
        shade._make_in = make_in
    assert seen and all(seen), "the expanded pyramid was not handed to the kernel as one shared map"
    _compare_grads(G, "eval_env", m, embs, geom, report)
    _judge("eval_env", report)


def test_vis_env_run_vis_relight_call():
    """run_vis_relight.py:110-122: no_grad, `envbg` present -> env background composite + diffuse / specular breakdown renders
    concatenated along the width (rgca.py:232-245)."""
    G = _gold()
    embs, geom = (t.detach().cuda() for t in S.leaves(1, 200, _stored(G, "vis_env")))
    m = _model(G, embs, geom).eval()
    m.learn_blur_enabled = m.cal_enabled = False
    batch = _env_batch(G, "vis_env", 1, 200, with_envbg=True)
    with torch.no_grad(), _Replay(G, "vis_env", compose=True):
        preds = m.forward(**batch)
    report = {}
    _compare_outputs(G, "vis_env", preds, report, keys=("rgb", "alpha", "depth", "color", "headrel_light_sh", "spec_color",
                                                       "diff_color"))
    assert preds["rgb"].shape[-1] == 3 * S.W
    _judge("vis_env", report)


# ---- the fourth case: an UN-manicured scene (VERDICT r5 next #3e) ----------------------------------------------------------
RAW_PATH = os.path.join(os.path.dirname(__file__), "golden", "rgca_model_raw_golden.npz")
RAW_SEED = 300
# parameter / leaf gradients of the raw scene: every discontinuity the three fixtures above keep out is in (LeakyReLU kinks of
# the decoder ladder, flip pixels, equal-depth pairs), so these are REPORTED against a coarse bar; the 1e-4 bar is applied
# where the discontinuities can be attributed -- per Gaussian, at the decoder outputs, by the W-protocol of
# tests/test_gpu_fullsize.py with predicates evaluated on the fixture's (the reference's) data
RAW_BAR_PARAM_GRAD = 5e-3


def _raw_predicates(G, rgb_hip, alpha_hip):
    """Per view: the Gaussians that reach a FLAGGED pixel with alpha >= half the 1/255 cut.  Flagged = (a) flip pixels: the
    final transmittance differs by more than rounding -- one pipeline composites a Gaussian the other skips at the alpha =
    1/255 / T = 1e-4 cuts; (b) tie pixels: pixels that BOTH members of a same-tile pair closer than 32 ulps in depth reach
    (their compositing order hinges on the last bit of Rt @ head_pose) and where the colour actually differs.  Oracle-side
    data throughout: the reference's preds of the fixture, projected by the CPU oracle."""
    from oracle import cref
    from test_gpu_fullsize import _flip_touched

    B = G["raw_point/out/rgb"].shape[0]
    hp, Rt, K = _t(G["raw_point/stored/head_pose"]), _t(G["raw_point/stored/Rt"]), S.cameras(B)[0]
    hp4 = torch.cat([hp, torch.zeros_like(hp[:, :1])], 1)
    hp4[:, 3, 3] = 1.0
    hRt = Rt @ hp4
    out, stats = [], {"flip_pixels": 0, "tie_pairs": 0, "tie_pixels_that_differ": 0}
    for b in range(B):
        pr = {k: _t(G[f"raw_point/out/{k}"])[b] for k in ("primpos", "primscale", "primqvec", "opacity")}
        xys, depths, radii, conics, comp, nth, _ = cref.project_gaussians(
            pr["primpos"], pr["primscale"], 1.0, pr["primqvec"], hRt[b], float(K[b, 0, 0]), float(K[b, 1, 1]),
            float(K[b, 0, 2]), float(K[b, 1, 2]), S.H, S.W, 16, 0.1)
        o = {"xys": xys, "conics": conics, "opac_eff": pr["opacity"][:, 0] * comp, "radii": radii}
        T_ref = 1.0 - _t(G["raw_point/out/alpha"])[b, 0]
        T_hip = 1.0 - alpha_hip[b, 0].cpu()
        flip = (T_hip - T_ref).abs() > 1e-3 * T_ref.clamp(min=1e-4) + 3e-7     # (+ the resolution of alpha = 1 - T in fp32)
        stats["flip_pixels"] += int(flip.sum())
        keys, ids, bins = cref.bin_and_sort(xys, depths, radii, nth, S.H, S.W, 16)
        dbits = keys & 0xFFFFFFFF
        close = ((keys[1:] >> 32) == (keys[:-1] >> 32)) & ((dbits[1:] - dbits[:-1]) < 32)
        pairs = {(int(ids[k]), int(ids[k + 1])) for k in torch.nonzero(close).flatten().tolist()}
        stats["tie_pairs"] += len(pairs)
        differs = (rgb_hip[b].cpu() - _t(G["raw_point/out/rgb"])[b]).abs().amax(0) > 2e-5
        yy, xx = torch.meshgrid(torch.arange(S.H) + 0.5, torch.arange(S.W) + 0.5, indexing="ij")

        def reach(i):
            dx, dy = xys[i, 0] - xx, xys[i, 1] - yy
            sig = 0.5 * (conics[i, 0] * dx * dx + conics[i, 2] * dy * dy) + conics[i, 1] * dx * dy
            return (o["opac_eff"][i] * torch.exp(-sig) >= 0.5 / 255.0) & (sig >= 0)

        tie = torch.zeros(S.H, S.W, dtype=torch.bool)
        for i, j in pairs:
            tie |= reach(i) & reach(j)
        tie &= differs
        stats["tie_pixels_that_differ"] += int(tie.sum())
        out.append(_flip_touched(o, torch.nonzero(flip | tie)))
    return torch.stack(out), stats


@pytest.mark.parametrize("fused_tail", [False, True])
def test_raw_scene_point_lights_w_protocol(fused_tail, monkeypatch):
    """train_point's call on the scene nobody screened: plain seeded leaves, no depth nudges, no roughness conditioning
    (tests/golden/make_rgca_model_golden.py: raw_case).  Every returned key, images, parameter gradients against a coarse
    bar; and -- with the decoder's last layers un-fused (GOLIATH_FUSED_TAIL=0: the product's default contracts the light
    into them and never materialises the 125-channel activation, goliath_amd/tail.py) -- the per-Gaussian gradients at the
    decoder outputs by the W-protocol."""
    from scenes import worst_set

    monkeypatch.setenv("GOLIATH_FUSED_TAIL", "1" if fused_tail else "0")
    tag = "raw_point" + ("_fused_tail" if fused_tail else "")

    G = np.load(RAW_PATH)
    st = _stored(G, "raw_point")
    embs, geom = (t.detach().cuda().requires_grad_(True) for t in S.leaves(2, RAW_SEED, st))
    from goliath_amd import rgca

    m = S.ShapedAutoEncoder(embs, geom, 0, cal=True, blur=True, nudges=None, raw=True).cuda()
    m.decoder.forward = types.MethodType(rgca.prim_decoder_forward, m.decoder)
    m.render = types.MethodType(rgca.autoencoder_render, m)
    m.forward = types.MethodType(rgca.autoencoder_forward, m)
    m.train()
    kept = {}

    def keep(name):
        def hook(mod, inp, o):
            o.retain_grad()
            kept[name] = o
        return hook

    hooks = [m.decoder.vnocond_mod.register_forward_hook(keep("f_vnocond")),
             m.decoder.vcond_mod.register_forward_hook(keep("f_vcond"))]
    batch = _cuda(S.batch_inputs(2, RAW_SEED, stored=st))
    with _Replay(G, "raw_point") as rp:
        preds = m.forward(**batch)
        assert rp.sh_calls == 2 and rp.rand_calls == 1
    for h in hooks:
        h.remove()
    report = {}
    _compare_outputs(G, "raw_point", preds, report)
    assert bool(kept) == (not fused_tail)    # the fused tail never calls the two stacks as a whole
    for k in kept:                           # the decoder ladder itself (MIOpen / ATen on both sides of the boundary)
        report[f"mid/{k}"] = rel_l2(kept[k].detach().cpu(), _t(G[f"raw_point/mid/{k}"]))
    flagged, stats = _raw_predicates(G, preds["rgb"].detach(), preds["alpha"].detach())     # [B, N] bool
    # images: the pixels outside the flagged set must agree to the image bar; the flagged set must be small
    B, N = flagged.shape
    npix = B * S.H * S.W
    _backprop(preds)
    _compare_grads(G, "raw_point", m, embs, geom, report)
    w = {}
    for k in kept:
        a = kept[k].grad.reshape(B, -1, N)
        b = _t(G[f"raw_point/grad/{k}"]).reshape(B, -1, N)
        W_idx, all_rel, rest_rel = worst_set(a, b, 1e-4)
        fl = flagged.flatten()
        e2 = (a.double().cpu() - b.double()).pow(2).sum(1).flatten()
        r2 = b.double().pow(2).sum(1).flatten()
        w[k] = {"rel_l2_all_gaussians": all_rel, "rel_l2_without_flagged": float((e2[~fl].sum() / r2[~fl].sum()).sqrt()),
                "flagged_fraction": float(fl.float().mean()), "rel_l2_without_W": rest_rel, "W_size": int(W_idx.numel()),
                "W_unexplained": int((~fl[W_idx]).sum())}
    print(f"\nRGCA_MODEL_GOLDEN {tag} " + " ".join(f"{k}={v:.2e}" for k, v in sorted(report.items())))
    print(f"RGCA_MODEL_GOLDEN {tag} predicates", stats, "decoder-output gradients", w)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        import json

        path = os.path.join(out_dir, "rgca_model_parity.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[tag] = dict(report, predicates=stats, decoder_output_gradients=w)
        json.dump(data, open(path, "w"), indent=1)
    # measured (round 6, profiles/r06_rgca_model_parity.json): 262 same-tile pairs within 32 depth-ulps, NONE composited in a
    # different order on the GPU; 1 flip pixel; every parameter / leaf gradient <= 5.9e-5, every decoder-output gradient
    # <= 5.7e-5 with W empty.  While the scene shows (almost) no realised discontinuity the parameter gradients are held to
    # 1.5e-4; a box on which a tied pair swaps falls back to the coarse bar and must explain itself per Gaussian below.
    quiet = stats["tie_pixels_that_differ"] == 0 and stats["flip_pixels"] <= 4
    for k, v in report.items():
        assert v == v, (k, "NaN")
        if k.startswith("grad/"):
            assert v < (1.5e-4 if quiet else RAW_BAR_PARAM_GRAD), (k, v, stats)
        elif k in ("out/rgb", "out/depth"):
            assert v < 2e-3, (k, v)                   # whole image incl. its flip / tie pixels
        elif k == "out/alpha":
            assert v < BAR_IMAGE, (k, v)              # (the order of a tied pair does not change T)
        else:
            assert v < BAR_PER_GAUSSIAN, (k, v)
    assert stats["flip_pixels"] + stats["tie_pixels_that_differ"] < 2e-3 * npix, stats
    for k, r in w.items():
        assert r["W_unexplained"] == 0 and r["rel_l2_without_W"] <= 1e-4 and r["W_size"] <= 0.02 * B * N, (k, r)
        assert r["rel_l2_without_flagged"] < 2e-4, (k, r)
