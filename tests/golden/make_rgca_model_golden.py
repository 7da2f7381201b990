"""Generates tests/golden/rgca_model_golden.npz by running the REFERENCE's own model code on the CPU:
    ca_code.models.rgca.AutoEncoder.forward / .render        (rgca.py:112-151, 153-253)
    ca_code.models.rgca.PrimDecoder.forward                  (rgca.py:466-620)
    ca_code.utils.render_gsplat.render                       (render_gsplat.py:13-108)
    ca_code.utils.light_decorator.EnvSpinDecorator           (__init__ incl. the SG-prefiltered pyramid, mipmap(), forward: :18-164)
    ca_code.nn.color_cal.CalV5, ca_code.nn.dof_cal.LearnableBlur, ca_code.nn.layers (the weight-normalised decoder layers)
as unbound methods / real classes on the seeded stand-in of tests/rgca_shaped.py (same attribute layout, an 8 -> 64 decoder
ladder, 4096 Gaussians, 256 x 208 images).  What the reference obtains from third-party / CUDA code is served by the CPU oracles:
    gsplat.project_gaussians / rasterize_gaussians  -> oracle/gsplat_module.py over oracle/gsplat_oracle.c (PARITY UNPINNED inside:
                                                       gsplat's sources are absent; the autograd wiring follows the call sites)
    sgutilslib.evaluate_gaussian_fwd / _bwd         -> the reference's OWN sg.cu compiled for the host (oracle/_ref/libref.so)
    cv2.imread / cv2.resize                         -> a synthetic log-normal HDR of 64 x 128 texels (no image file, no OpenCV here)
Cases:
    train_point   training mode, point lights (n_lights mixed), is_fully_lit_frame mixed, calibration + learnable blur on;
                  outputs, and the gradients of a fixed random scalar w.r.t. embs, geom and a set of parameters
    eval_env      eval, the batch EnvSpinDecorator.forward hands to the model (preconv_envmap from mipmap(), lightrot, the 512
                  env lights), WITHOUT `envbg` (the differentiable env path): outputs + gradients
    vis_env       what run_vis_relight.py runs (run_vis_relight.py:110-122): decorated model, no_grad, `envbg` present -> the
                  env background composite and the diffuse / specular breakdown renders (rgca.py:232-245); B = 1
    raw_point     (its own file, rgca_model_raw_golden.npz; `--raw-only` writes just that one) train_point's call on an
                  UN-manicured scene: plain seeded leaves (no LeakyReLU screening), no depth nudges, no roughness conditioning.
                  Besides outputs and the parameter gradients it stores the gradients at the DECODER OUTPUTS (f_vnocond,
                  f_vcond: forward hooks on the two stacks) -- per-Gaussian quantities the GPU test judges by the W-protocol
                  with the flip / depth-tie predicates (tests/test_gpu_rgca_model_golden.py)
Run in the build container (needs /root/reference):   python tests/golden/make_rgca_model_golden.py [--raw-only]
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import ref_stubs  # noqa: E402
from oracle import cref, gsplat_module, refso  # noqa: E402

_sg = refso if refso.available() else cref


class _SgLib:
    @staticmethod
    def evaluate_gaussian_fwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, integral, w_type):
        integral.copy_(_sg.evaluate_gaussian_fwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, w_type))
        return []

    @staticmethod
    def evaluate_gaussian_bwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, grad_integral, grad_dirs,
                              grad_sigmas, grad_light_values, w_type):
        gd, gs, _ = _sg.evaluate_gaussian_bwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights,
                                              grad_integral, w_type)
        grad_dirs.copy_(gd)
        grad_sigmas.copy_(gs)
        return []


ref_stubs.install(sgutilslib=_SgLib)
sys.modules["gsplat"] = gsplat_module.as_module()      # before ca_code.utils.render_gsplat binds the two names
import ca_code.models.rgca as R  # noqa: E402
import ca_code.nn.layers as la  # noqa: E402
import ca_code.utils.light_decorator as LD  # noqa: E402
import ca_code.utils.sh as ref_sh  # noqa: E402
from ca_code.nn.color_cal import CalV5  # noqa: E402
from ca_code.nn.dof_cal import LearnableBlur  # noqa: E402
from ca_code.utils import envmap as ref_envmap  # noqa: E402

import rgca_shaped as S  # noqa: E402
import importlib  # noqa: E402

_wn_mod = importlib.import_module("torch.nn.utils.weight_norm")   # (the attribute of that name on torch.nn.utils is the function)

# torch's fp32 CPU norm accumulates sequentially: for encmod's 4 M-element direction tensor (rgca.py:399-401: 256 -> 256 x 8 x 8,
# hard-wired by the .view of :494) it is 9.5e-5 off, which the weight norm (layers.py:157-244 -> torch._weight_norm) turns
# into a 9.5e-5 relative error of every activation behind it -- an artefact of the HOST's reduction order (the GPU's tree
# reduction, where the reference runs, is 3e-7 from fp64; tools/diag_model_golden.py).  The generator therefore forms the
# whole-tensor norm in fp64; everything else of the reference's layers runs as is, in fp32.
_orig_weight_norm = _wn_mod._weight_norm


def _accurate_weight_norm(v, g, dim=0):
    if dim == -1:
        return v * (g / v.double().pow(2).sum().sqrt().to(v.dtype))
    return _orig_weight_norm(v, g, dim)


_wn_mod._weight_norm = _accurate_weight_norm

SEED = 0
CASES = (("train_point", 2, SEED), ("eval_env", 2, SEED + 100), ("vis_env", 1, SEED + 200))
from goliath_amd import decoder as _D  # noqa: E402

_D._wn = lambda v, g: v * (g / v.double().pow(2).sum().sqrt().to(v.dtype))   # the stand-in's layers: same fp64 denominator
NUDGES = None       # (flat Gaussian indices, dz): set by depth_nudges() before the cases run
LEAVES = {}         # tag -> (embs, geom): chosen by pick_leaves() first
MIN_DEPTH_GAP_ULPS = 32


def depth_nudges():
    """Per-Gaussian z offsets (through the last layer's untied bias) such that, in every view of every case, no two
    Gaussians that share a tile are closer than MIN_DEPTH_GAP_ULPS in depth.  Two fp32 pipelines whose camera matrices
    differ in the last bit (Rt @ head_pose on another BLAS, or on the GPU) would otherwise composite such a pair in
    opposite orders: an O(1) change of the pixels under it and of every gradient, which says nothing about either
    pipeline.  Stored in the fixture (`nudges/index`, `nudges/dz`); tests/rgca_shaped.py applies them."""
    from oracle import shade_ref

    idx_all, dz_all = [], []
    for it in range(12):
        nud = (np.array(idx_all, np.int64), np.array(dz_all, np.float32)) if idx_all else None
        found = 0
        for tag, B, seed in CASES:
            embs, geom = (t.detach() for t in LEAVES[tag])
            dec = S.ShapedPrimDecoder(SEED, nud).eval()
            batch = S.batch_inputs(B, seed)
            hp = batch["head_pose"]
            rot, trans = hp[:, :3, :3], hp[:, :3, 3]
            campos = ((batch["campos"] - trans)[:, None] @ rot)[:, 0]
            hp4 = torch.cat([hp, torch.zeros_like(hp[:, :1])], 1)
            hp4[:, 3, 3] = 1.0
            hRt = batch["Rt"] @ hp4
            with torch.no_grad():
                postex = dec.geo_fn.to_uv(geom)
                tn = torch.nn.functional.normalize(dec.geo_fn.to_uv(dec.geo_fn.vn(geom)), dim=1)
                z = dec.encmod(embs).view(-1, 256, 8, 8)
                view = dec.viewmod(torch.nn.functional.normalize(campos, dim=1))[:, :, None, None].expand(-1, -1, 8, 8)
                f_vn, f_vc = dec.vnocond_mod(z), dec.vcond_mod(torch.cat([z, view], 1))
                pr = shade_ref.shade(f_vn, f_vc, postex, tn, dec.albedo, torch.zeros(B, 3, 81), campos,
                                     light_intensity=torch.ones(B, 1, 3), light_pos=torch.ones(B, 1, 3), n_lights=torch.ones(B))
            K = batch["K"]
            for b in range(B):
                xys, depths, radii, conics, comp, nth, _ = cref.project_gaussians(
                    pr["primpos"][b], pr["primscale"][b], 1.0, pr["primqvec"][b], hRt[b], float(K[b, 0, 0]), float(K[b, 1, 1]),
                    float(K[b, 0, 2]), float(K[b, 1, 2]), S.H, S.W, 16, 0.1)
                keys, ids, bins = cref.bin_and_sort(xys, depths, radii, nth, S.H, S.W, 16)
                dbits = (keys & 0xFFFFFFFF)
                same_tile = (keys[1:] >> 32) == (keys[:-1] >> 32)
                close = same_tile & ((dbits[1:] - dbits[:-1]) < MIN_DEPTH_GAP_ULPS)
                for k in torch.nonzero(close).flatten().tolist():
                    gidx = int(ids[k + 1])
                    if gidx not in idx_all:
                        idx_all.append(gidx)
                        dz_all.append(0.013 * (1 + len(idx_all) % 7))       # 0.013 .. 0.09 mm away from the camera side
                        found += 1
        print(f"depth separation pass {it}: {found} new nudges ({len(idx_all)} in total)")
        if not found:
            return (np.array(idx_all, np.int64), np.array(dz_all, np.float32))
    raise RuntimeError("depth separation did not converge")


def reference_model(embs, geom, cal=True, blur=True, raw=False):
    """The stand-in with every sub-module that has a reference class replaced by that class, loaded with the stand-in's seeded
    parameters; the reference's methods bound on it."""
    m = S.ShapedAutoEncoder(embs, geom, SEED, cal=cal, blur=blur, nudges=None if raw else NUDGES, raw=raw)
    dec = m.decoder
    sd = dec.state_dict()
    lrelu = lambda: torch.nn.LeakyReLU(0.2, inplace=True)
    dec.viewmod = torch.nn.Sequential(*la.make_linear(3, 8, "wn", lrelu()))
    dec.encmod = torch.nn.Sequential(*la.make_linear(256, 256 * 8 * 8, "wn", lrelu()))

    def stack(n_in, n_out):
        layers, c, s = [], n_in, 8
        for co in S.HIDDEN:
            s *= 2
            layers += la.make_conv_trans(c, co, 4, 2, 1, "wn", lrelu(), ub=(s, s))
            c = co
        return torch.nn.Sequential(*layers, *la.make_conv_trans(c, n_out, 4, 2, 1, "wn", ub=(2 * s, 2 * s)))

    dec.vnocond_mod, dec.vcond_mod = stack(256, dec.n_diff_coeffs + 12), stack(256 + 8, 4)
    missing = dec.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    dec.forward = types.MethodType(R.PrimDecoder.forward, dec)
    if cal:
        shaped = m.cal
        m.cal = CalV5(cameras=list(S.CAMERAS), identity_camera=S.IDENTITY)
        assert m.cal.identity_idx == shaped.identity_idx and list(m.cal.grey_idxs) == list(shaped.grey_idxs)
        assert (m.cal.gs_lrscale, m.cal.col_lrscale) == (shaped.gs_lrscale, shaped.col_lrscale)
        with torch.no_grad():
            m.cal.holder.params.copy_(shaped.params)
    if blur:
        shaped = m.learn_blur
        m.learn_blur = LearnableBlur(list(S.CAMERAS))
        with torch.no_grad():
            m.learn_blur.weights_raw.copy_(shaped.weights_raw)
    m.render = types.MethodType(R.AutoEncoder.render, m)
    m.forward = types.MethodType(R.AutoEncoder.forward, m)
    return m


def grad_of(m, name):
    if name == "cal.params":
        return m.cal.holder.params.grad
    obj = m
    for part in name.split("."):
        obj = getattr(obj, part) if not part.isdigit() else obj[int(part)]
    return obj.grad


class Recorder:
    """Records every call of the reference's sh.dir2sh_torch and torch.rand made during a forward: the GPU test replays them
    (the SH basis is reference Python that is absent on the GPU box; the random training light is drawn on the device there)."""

    def __init__(self, out, tag):
        self.out, self.tag, self.n_sh, self.n_rand = out, tag, 0, 0

    def __enter__(self):
        self._sh, self._rand = ref_sh.dir2sh_torch, torch.rand

        def dir2sh(n, d):
            r = self._sh(n, d)
            self.out[f"{self.tag}/sh{self.n_sh}/dirs"] = d.detach().numpy().copy()
            self.out[f"{self.tag}/sh{self.n_sh}/coeffs"] = r.detach().numpy().copy()
            self.n_sh += 1
            return r

        def rand(*a, **k):
            r = self._rand(*a, **k)
            self.out[f"{self.tag}/rand{self.n_rand}"] = r.numpy().copy()
            self.n_rand += 1
            return r

        ref_sh.dir2sh_torch, torch.rand = dir2sh, rand
        return self

    def __exit__(self, *exc):
        ref_sh.dir2sh_torch, torch.rand = self._sh, self._rand


def store(out, tag, preds, keys=None):
    for k, v in preds.items():
        if torch.is_tensor(v) and (keys is None or k in keys):
            out[f"{tag}/out/{k}"] = v.detach().numpy().astype(np.float32)


def backprop(out, tag, m, preds, embs, geom):
    wf = S.loss_weights(SEED)
    loss = sum((preds[k] * wf[k](preds[k])).sum() for k in ("rgb", "depth", "primscale_preclip", "spec_nml", "color_rand")
               if k in preds)
    loss.backward()
    out[f"{tag}/loss"] = np.float64(loss.item())
    out[f"{tag}/grad/embs"], out[f"{tag}/grad/geom"] = embs.grad.numpy().copy(), geom.grad.numpy().copy()
    for name in S.GRAD_PARAMS:
        if name.split(".")[0] in ("cal", "learn_blur") and not hasattr(m, name.split(".")[0]):
            continue
        g = grad_of(m, name)
        if g is not None:
            out[f"{tag}/grad/{name}"] = g.numpy().copy()


def synthetic_hdr():
    g = torch.Generator().manual_seed(SEED + 21)
    # log-normal radiance with a few bright blobs: values above 1 exist (the specular clamp(max=1) of rgca.py:556 is active)
    img = torch.exp(0.8 * torch.randn(64, 128, 3, generator=g)) * 0.25
    yy, xx = torch.meshgrid(torch.arange(64.0), torch.arange(128.0), indexing="ij")
    for cy, cx, amp in ((20.0, 30.0, 80.0), (40.0, 90.0, 40.0)):
        img += amp * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 30.0)[..., None]
    return img.numpy().astype(np.float32)


MIN_PREACT = 4e-6


def pick_leaves(B, seed, batch_of):
    """embs / geom for a case such that no LeakyReLU of the decoder sees a pre-activation within MIN_PREACT of zero: there the
    derivative jumps from 0.2 to 1, and the CPU's and the GPU's value (5e-7 apart) may land on opposite sides -- an O(1)
    change of that element's gradient (measured: one of 32768 pre-activations at 4.0e-7 put 1.6e-3 on a weight gradient) that
    says nothing about either pipeline.  Tries seed, seed + 1000, ... ; the chosen tensors are stored in the fixture."""
    for attempt in range(40):
        embs, geom = S.leaves(B, seed + 1000 * attempt)
        dec = S.ShapedPrimDecoder(SEED, NUDGES).eval()
        lows = []
        hooks = [mod.register_forward_pre_hook(lambda mod, inp: lows.append(float(inp[0].detach().abs().min())))
                 for mod in dec.modules() if isinstance(mod, torch.nn.LeakyReLU)]
        batch = batch_of(B, seed)
        hp = batch["head_pose"]
        campos = ((batch["campos"] - hp[:, :3, 3])[:, None] @ hp[:, :3, :3])[:, 0]
        with torch.no_grad():
            z = dec.encmod(embs).view(-1, 256, 8, 8)
            view = dec.viewmod(torch.nn.functional.normalize(campos, dim=1))[:, :, None, None].expand(-1, -1, 8, 8)
            dec.vnocond_mod(z), dec.vcond_mod(torch.cat([z, view], 1))
        for h in hooks:
            h.remove()
        if min(lows) >= MIN_PREACT:
            print(f"leaves for seed {seed}: attempt {attempt}, smallest |pre-activation| {min(lows):.2e}")
            return embs, geom
    raise RuntimeError("no leaves without a LeakyReLU kink found")


def store_inputs(out, tag, batch, embs, geom):
    for k in ("head_pose", "Rt", "campos", "light_pos"):
        if k in batch:
            out[f"{tag}/stored/{k}"] = batch[k].numpy().copy()
    out[f"{tag}/stored/embs"], out[f"{tag}/stored/geom"] = embs.detach().numpy().copy(), geom.detach().numpy().copy()


RAW_SEED = SEED + 300


def raw_case():
    """The fourth case: nothing of the scene is screened or nudged (see the module docstring)."""
    out = {}
    B = 2
    embs, geom = S.leaves(B, RAW_SEED)
    m = reference_model(embs, geom, raw=True).train()
    kept = {}
    def keep(name):
        def hook(mod, inp, o):
            o.retain_grad()
            kept[name] = o          # (returns None: the output itself goes on)
        return hook

    hooks = [m.decoder.vnocond_mod.register_forward_hook(keep("f_vnocond")),
             m.decoder.vcond_mod.register_forward_hook(keep("f_vcond"))]
    batch = S.batch_inputs(B, RAW_SEED)
    store_inputs(out, "raw_point", batch, embs, geom)
    torch.manual_seed(99)
    with Recorder(out, "raw_point"):
        preds = m.forward(**batch)
    for h in hooks:
        h.remove()
    store(out, "raw_point", preds)
    backprop(out, "raw_point", m, preds, embs, geom)
    for k, v in kept.items():
        out[f"raw_point/grad/{k}"] = v.grad.numpy().copy()
        out[f"raw_point/mid/{k}"] = v.detach().numpy().copy()
    print("raw_point: alpha mean", float(preds["alpha"].mean()), "sigma min", float(preds["sigma"].min()), "max",
          float(preds["sigma"].max()), "rgb mean", float(preds["rgb"].mean()))
    path = os.path.join(HERE, "rgca_model_raw_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


def main():
    global NUDGES
    if "--raw-only" in sys.argv:
        return raw_case()
    out = {}
    for tag, B, seed in CASES:
        LEAVES[tag] = pick_leaves(B, seed, S.batch_inputs)
    NUDGES = depth_nudges()
    out["nudges/index"], out["nudges/dz"] = NUDGES
    # ------------------------------------------------------------------------------------------ train_point
    B = 2
    embs, geom = LEAVES["train_point"]
    m = reference_model(embs, geom).train()
    batch = S.batch_inputs(B, SEED)
    store_inputs(out, "train_point", batch, embs, geom)
    torch.manual_seed(99)
    with Recorder(out, "train_point"):
        preds = m.forward(**batch)
    store(out, "train_point", preds)
    backprop(out, "train_point", m, preds, embs, geom)
    print("train_point: alpha mean", float(preds["alpha"].mean()), "max", float(preds["alpha"].max()), "rgb mean",
          float(preds["rgb"].mean()), "keys", sorted(preds))

    # ------------------------------------------------------------------------------------------ the env-map driver
    cv2 = sys.modules["cv2"]
    hdr = synthetic_hdr()
    cv2.imread = lambda path, flag=None: hdr[:, :, ::-1]           # (BGR on disk; _set_lightmap flips it back)
    cv2.resize = lambda img, size, interpolation=None: img         # the synthetic map already has the size we want
    cv2.INTER_AREA = 3
    torch.Tensor.cuda = lambda self, *a, **k: self                 # _set_lightmap prefilters on "the GPU" (:76, :95)
    torch.manual_seed(7)
    embs, geom = LEAVES["eval_env"]
    m = reference_model(embs, geom)
    m.eval()
    m.learn_blur_enabled = m.cal_enabled = False                   # run_vis_relight.py:83-84
    handed = {}

    def model_call(**data):
        handed.update(data)
        return m.forward(**data)

    deco = LD.EnvSpinDecorator(model_call, envmap_path="synthetic.hdr", ydown=True, env_scale=8.0)   # run_vis_relight.py:110-115
    out["env/image"] = deco.image.numpy().copy()
    for i in range(deco.miplevel):
        out[f"env/mipmap_{i}"] = getattr(deco, f"mipmap_{i}").numpy().copy()
    batch = S.batch_inputs(B, SEED + 100)
    store_inputs(out, "eval_env", batch, embs, geom)
    for k in ("light_intensity", "light_pos", "n_lights"):          # the decorator supplies them
        batch.pop(k)

    # eval_env: the decorator's batch without `envbg` -> the differentiable env path of AutoEncoder.forward
    class NoBg(dict):
        pass

    def model_call_nobg(**data):
        handed.update(data)
        data = {k: v for k, v in data.items() if k != "envbg"}
        return m.forward(**data)

    deco.mod = model_call_nobg
    with Recorder(out, "eval_env"):
        preds = deco(**batch, index=[37, 120])
    for k in ("light_intensity", "light_pos", "lightrot", "envbg", "n_lights"):
        out[f"eval_env/in/{k}"] = handed[k].numpy().copy()
    out["eval_env/in/preconv_scale"] = np.float32(float(handed["preconv_envmap"][0][0, 0, 0, 0] / deco.mipmap_0[0, 0, 0, 0]))
    for i, lvl in enumerate(handed["preconv_envmap"]):
        assert lvl.shape[0] == B and torch.equal(lvl[0], lvl[1])    # ONE pyramid expanded over the batch (:96-100)
        out[f"eval_env/in/preconv_envmap_{i}"] = lvl[:1].numpy().copy()
    store(out, "eval_env", preds)
    backprop(out, "eval_env", m, preds, embs, geom)
    print("eval_env: alpha mean", float(preds["alpha"].mean()), "spec max", float(preds["spec_color"].max()),
          "spec clamp active on", int((preds["spec_color"] / preds["spec_vis"].clamp(min=1e-6) > 0.999).sum()), "values")

    # vis_env: run_vis_relight's call (no_grad, envbg present), B = 1
    embs1, geom1 = LEAVES["vis_env"]
    m = reference_model(embs1, geom1)
    m.eval()
    m.learn_blur_enabled = m.cal_enabled = False
    deco.mod = model_call
    batch = S.batch_inputs(1, SEED + 200)
    store_inputs(out, "vis_env", batch, embs1, geom1)
    for k in ("light_intensity", "light_pos", "n_lights"):
        batch.pop(k)
    handed.clear()
    with torch.no_grad(), Recorder(out, "vis_env"):
        preds = deco(**batch, index=[200])
    for k in ("light_intensity", "light_pos", "lightrot", "envbg", "n_lights"):
        out[f"vis_env/in/{k}"] = handed[k].numpy().copy()
    for i, lvl in enumerate(handed["preconv_envmap"]):
        out[f"vis_env/in/preconv_envmap_{i}"] = lvl[:1].numpy().copy()
    store(out, "vis_env", preds, keys=("rgb", "alpha", "depth", "color", "headrel_light_sh", "spec_color", "diff_color"))
    # compose_envmap (envmap.py:325-345) is an affine map of (render, alpha) whose coefficients depend on envbg, K, Rt only:
    #   out = (1 - ma) * (render + (1 - alpha) * bg) + ma * mi.   Recorded through three probe calls of the reference function
    # so that the GPU test (no ca_code there) can stand in for it.
    K, Rt, envbg = batch["K"], batch["Rt"], handed["envbg"]
    z3, z1 = torch.zeros(1, 3, S.H, S.W), torch.zeros(1, 1, S.H, S.W)
    x00 = ref_envmap.compose_envmap(z3.clone(), z1.clone(), envbg, K, Rt)          # (1 - ma) * bg + ma * mi
    x01 = ref_envmap.compose_envmap(z3.clone(), z1.clone() + 1.0, envbg, K, Rt)    # ma * mi
    x11 = ref_envmap.compose_envmap(z3.clone() + 1.0, z1.clone() + 1.0, envbg, K, Rt)   # (1 - ma) + ma * mi
    out["vis_env/compose/one_minus_ma"] = (x11 - x01).numpy().copy()
    out["vis_env/compose/bg_term"] = (x00 - x01).numpy().copy()                    # (1 - ma) * bg
    out["vis_env/compose/mirror_term"] = x01.numpy().copy()                        # ma * mi
    path = os.path.join(HERE, "rgca_model_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    raw_case()


if __name__ == "__main__":
    main()
