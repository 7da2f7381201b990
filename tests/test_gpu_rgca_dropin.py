"""GPU: the model-level drop-in entry points (goliath_amd.rgca) reproduce the reference's
PrimDecoder.forward outputs (golden) when installed on a stand-in decoder, and AutoEncoder.render
semantics (alpha detached, depth normalised).  (The render test below checks the binding against render_batch, i.e.
consistency; parity of AutoEncoder.render / forward against the reference's own code: tests/test_gpu_rgca_model_golden.py.)"""
import types

import pytest
import torch
import torch.nn.functional as F

from scenes import head_scene, rel_l2
from test_oracle_shade import load_golden

pytestmark = pytest.mark.gpu


class _Const(torch.nn.Module):
    def __init__(self, v):
        super().__init__()
        self.v = v

    def forward(self, *_):
        return self.v


class _Geo:
    def __init__(self, postex, tn_raw):
        self.postex, self.tn_raw, self.n = postex, tn_raw, 0

    def to_uv(self, x):
        self.n += 1
        return self.postex if self.n % 2 == 1 else self.tn_raw

    def vn(self, g):
        return g


@pytest.mark.parametrize("tag", ["sg_eval", "env_eval"])
def test_prim_decoder_forward_dropin(tag):
    from goliath_amd import rgca

    G = load_golden()
    c = lambda k: G[f"in/{k}"].cuda()
    B = c("f_vnocond").shape[0]
    dec = torch.nn.Module()
    dec.encmod, dec.viewmod = _Const(torch.zeros(B, 256 * 8 * 8).cuda()), _Const(torch.zeros(B, 8).cuda())
    dec.vnocond_mod, dec.vcond_mod = _Const(c("f_vnocond")), _Const(c("f_vcond"))
    dec.geo_fn = _Geo(c("postex"), c("tn_raw"))
    dec.albedo = torch.nn.Parameter(c("albedo"))
    dec.color_sh_degree, dec.diff_sh_degree = 3, 8
    dec.eval()
    env = tag.startswith("env")
    preds = rgca.prim_decoder_forward(
        dec, torch.zeros(B, 256).cuda(), torch.zeros(B, 1, 3).cuda(), c("campos"), c("light_intensity"), c("light_pos"),
        c("light_sh"), c("n_lights"), [c(f"mip{i}") for i in range(4)] if env else None, c("lightrot") if env else None)
    for k, v in preds.items():
        ref = G[f"{tag}/out/{k}"]
        assert rel_l2(v.reshape(ref.shape), ref) < 6e-5, k   # measured 6.1e-6


def test_autoencoder_render_dropin_and_random_light():
    from goliath_amd import render_gs, rgca

    H, W, N, B = 96, 80, 2000, 2
    views = [head_scene(N, H, W, seed=20 + b, cam_angle=0.3 * b) for b in range(B)]
    preds = {k2: torch.stack([v[k] for v in views]).cuda() for k, k2 in
             (("means", "primpos"), ("quats", "primqvec"), ("scales", "primscale"), ("opacity", "opacity"), ("colors", "color"))}
    preds["primpos"].requires_grad_(True)
    K = torch.zeros(B, 3, 3)
    for b, v in enumerate(views):
        K[b, 0, 0], K[b, 1, 1], K[b, 0, 2], K[b, 1, 2] = v["fx"], v["fy"], v["cx"], v["cy"]
    Rt = torch.stack([v["viewmat"] for v in views])
    me = types.SimpleNamespace(height=H, width=W)
    rgb, alpha, depth = rgca.autoencoder_render(me, K.cuda(), Rt.cuda(), preds)
    assert rgb.shape == (B, 3, H, W) and not alpha.requires_grad and depth.requires_grad
    r2, a2, d2 = render_gs.render_batch(K.cuda(), Rt.cuda(), preds, H, W)
    assert torch.equal(rgb, r2) and torch.equal(alpha, a2)
    # training-only random light: unit direction, SH of that direction broadcast to 3 channels
    ld, lsh = rgca.random_light_sh(lambda deg, d: torch.cat([torch.ones_like(d[..., :1]), d], -1), 8, B, "cuda", torch.float32)
    assert ld.shape == (B, 1, 3) and lsh.shape == (B, 3, 4)
    assert torch.allclose(ld.norm(dim=-1), torch.ones(B, 1, device="cuda"), atol=1e-6)
    assert torch.allclose(lsh[:, 0, 1:], ld[:, 0])
