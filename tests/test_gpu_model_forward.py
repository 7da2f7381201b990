"""GPU: the model-level entry point `goliath_amd.rgca.autoencoder_forward` (drop-in for AutoEncoder.forward,
ca_code/models/rgca.py:153-253) driven end to end on a module shaped like the reference's AutoEncoder: reference-size
decoder (1024^2 slab, 162.8 M parameters, goliath_amd.decoder), head-relative transforms, fused decoder tail + shading
tail, batched render, fused image tail (calibration + background + learnable blur).  Checked: the output keys of
rgca.py:574-618 / :247-251, shapes, and the image against the same steps composed from separately tested pieces
(render_batch + plain-torch calibration / composite / gaussian blur) -- a CONSISTENCY test at the reference's decoder size
(1,048,576 Gaussians), not a parity test: PARITY of these entry points against the reference's own AutoEncoder.forward /
render / PrimDecoder.forward is tests/test_gpu_rgca_model_golden.py (round 5)."""
import math
import os
import sys
import types

import pytest
import torch
import torch.nn.functional as F

from scenes import rel_l2

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))


def _install_sh_stand_in():
    """ca_code.utils.sh is reference code (absent on the GPU box); autoencoder_forward only needs SOME fixed map from
    light directions to (n+1)^2 coefficients -- the same map feeds both sides of the comparison."""
    if "ca_code.utils.sh" in sys.modules:
        return

    def dir2sh_torch(n, d):
        x, y, z = d.unbind(-1)
        feats = [torch.ones_like(x)]
        k = 1
        while len(feats) < (n + 1) ** 2:
            feats.append(torch.cos(k * x + 0.5 * k * y) * torch.sin(0.7 * k * z + k))
            k += 1
        return torch.stack(feats, -1)

    for name in ("ca_code", "ca_code.utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    m = types.ModuleType("ca_code.utils.sh")
    m.dir2sh_torch = dir2sh_torch
    sys.modules["ca_code.utils.sh"] = m
    sys.modules["ca_code.utils"].sh = m


class _Geo:
    def __init__(self, postex, tn_raw):
        self.postex, self.tn_raw, self.n = postex, tn_raw, 0

    def to_uv(self, x):
        self.n += 1
        return self.postex if self.n % 2 == 1 else self.tn_raw

    def vn(self, g):
        return g


class _Cal(torch.nn.Module):
    def __init__(self, cams):
        super().__init__()
        self.cams = sorted(cams)
        p = torch.tensor([1.0, 1.0, 1.0, 0.0, 0.0, 0.0]).repeat(len(cams), 1)
        self.params = torch.nn.Parameter(p + 0.1 * torch.randn(len(cams), 6, generator=torch.Generator().manual_seed(5)))
        self.identity_idx, self.grey_idxs = 0, [i for i, c in enumerate(self.cams) if c.startswith("41")]
        self.gs_lrscale, self.col_lrscale = 1.0, 0.1

    def holder(self, idxs):
        return self.params[idxs]

    def name_to_idx(self, names):
        return torch.tensor([self.cams.index(n) for n in names], device=self.params.device)


def test_autoencoder_forward_dropin_end_to_end():
    import ref_stubs
    from goliath_amd import decoder, imgtail, render_gs, rgca

    _install_sh_stand_in()
    torch.manual_seed(0)
    dev = torch.device("cuda")
    B, S, H, W = 2, 1024, 256, 192
    cams = ["400002", "400004", "410011"]
    g = torch.Generator(device=dev).manual_seed(1)
    dec = decoder.PrimDecoderConvs(base=8).to(dev)
    with torch.no_grad():  # a decoder whose Gaussians have sensible sizes / opacities: bias the parameter channels
        dec.vnocond_mod[-1].bias[113 + 7:113 + 10] += 0.5
    d = F.normalize(torch.randn(S * S, 3, device=dev, generator=g), dim=-1)
    pos = d * torch.rand(S * S, 1, device=dev, generator=g) ** (1 / 3) * torch.tensor([90.0, 120.0, 100.0], device=dev)
    postex = pos.t().reshape(1, 3, S, S).expand(B, -1, -1, -1).contiguous()
    dec.geo_fn = _Geo(postex, postex)
    dec.albedo = torch.nn.Parameter(0.2 + 0.6 * torch.rand(1, S * S, 3, device=dev, generator=g))
    dec.color_sh_degree, dec.diff_sh_degree = 3, 8
    dec.forward = types.MethodType(rgca.prim_decoder_forward, dec)

    model = torch.nn.Module()
    model.height, model.width, model.n_diff_sh = H, W, 8
    embs = torch.randn(B, 256, device=dev, generator=g)
    model.encoder = lambda verts, color: {"embs": embs, "embs_mu": embs, "embs_logvar": torch.zeros_like(embs)}
    model.geomdecoder = lambda e: {"face_geom": torch.zeros(B, 10, 3, device=dev)}
    model.decoder = dec
    model.render = types.MethodType(rgca.autoencoder_render, model)
    model.cal_enabled = model.learn_blur_enabled = True
    model.cal = _Cal(cams).to(dev)
    raw = torch.nn.Parameter(torch.randn(len(cams), 3, device=dev, generator=g))
    model.learn_blur = types.SimpleNamespace(reg=lambda names: raw[torch.tensor([cams.index(n) for n in names], device=dev)])
    model.train()
    model.cal.train()
    dec.eval()  # no random training light: the comparison below re-runs the decoder

    head_pose = torch.cat([torch.eye(3), torch.tensor([[5.0], [-3.0], [10.0]])], 1)[None].repeat(B, 1, 1).to(dev)
    K = torch.zeros(B, 3, 3, device=dev)
    K[:, 0, 0] = K[:, 1, 1] = 3000.0 * W / 1334.0
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = W / 2.0, H / 2.0, 1.0
    Rt, campos = [], []
    for b in range(B):
        ang = 0.4 * b - 0.2
        eye = torch.tensor([700.0 * math.sin(ang), 0.0, -700.0 * math.cos(ang)])
        fwd = -eye / eye.norm()
        right = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), fwd)
        right = right / right.norm()
        R = torch.stack([right, torch.linalg.cross(fwd, right), fwd])
        Rt.append(torch.cat([R, (-R @ eye)[:, None]], 1))
        campos.append(eye)
    Rt, campos = torch.stack(Rt).to(dev), torch.stack(campos).to(dev)
    batch = dict(head_pose=head_pose, campos=campos, registration_vertices=torch.zeros(B, 10, 3, device=dev),
                 color=torch.zeros(B, 3, 8, 8, device=dev), light_intensity=torch.rand(B, 4, 1, device=dev, generator=g),
                 light_pos=1100.0 * F.normalize(torch.randn(B, 4, 3, device=dev, generator=g), dim=-1),
                 n_lights=torch.full((B,), 4, dtype=torch.int32, device=dev), K=K, Rt=Rt,
                 background=torch.rand(B, 3, H, W, device=dev, generator=g),
                 is_fully_lit_frame=torch.tensor([True, False], device=dev), camera_id=["400004", "410011"],
                 frame_id=torch.arange(B), iteration=0)
    preds = rgca.autoencoder_forward(model, **batch)
    want_keys = {"geom", "headrel_light_sh", "embs", "embs_mu", "embs_logvar", "color", "opacity", "primpos", "primqvec",
                 "primscale", "primscale_preclip", "sigma", "spec_vis", "spec_nml", "spec_dnml", "diff_color",
                 "spec_color", "primnmlbase", "rgb", "alpha", "depth", "learn_blur_weights"}
    assert want_keys <= set(preds), want_keys - set(preds)
    assert preds["rgb"].shape == (B, 3, H, W) and preds["alpha"].shape == (B, 1, H, W) and preds["depth"].shape == (B, 1, H, W)
    assert preds["primpos"].shape == (B, S * S, 3) and preds["learn_blur_weights"].shape == (B, 3)
    assert torch.isfinite(preds["rgb"]).all() and float(preds["alpha"].max()) > 0.9
    preds["rgb"].mean().backward()
    assert torch.isfinite(dec.albedo.grad).all() and float(dec.albedo.grad.abs().max()) > 0
    assert raw.grad is not None and model.cal.params.grad is not None

    # the same image from separately tested pieces
    with torch.no_grad():
        dec.geo_fn.n = 0
        headrel_Rt = Rt @ torch.cat([head_pose, torch.tensor([[[0.0, 0.0, 0.0, 1.0]]], device=dev).expand(B, -1, -1)], 1)
        dp = {k: preds[k] for k in ("primpos", "primqvec", "primscale", "opacity", "color")}
        rgb0, alpha0, _ = render_gs.render_batch(K, headrel_Rt, dp, H, W)
        M, bias = imgtail.cal_v5_matrix(model.cal, model.cal.name_to_idx(batch["camera_id"]))
        x = torch.einsum("bcj,bjhw->bchw", M, rgb0) + bias[:, :, None, None]
        bg = batch["background"].clone()
        bg[~batch["is_fully_lit_frame"]] *= 0.0
        x = x + (1.0 - alpha0) * bg
        wts = torch.softmax(model.learn_blur.reg(batch["camera_id"]), -1).reshape(B, 3, 1, 1, 1)
        want = wts[:, 0] * x + wts[:, 1] * ref_stubs.gaussian_blur(x, [3, 3]) + wts[:, 2] * ref_stubs.gaussian_blur(x, [7, 7])
    assert rel_l2(preds["rgb"], want) < 1e-5, rel_l2(preds["rgb"], want)
    assert torch.equal(preds["alpha"], alpha0)
