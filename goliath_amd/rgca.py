"""Model-level entry points of the RGCA render path with the reference's signatures.

  autoencoder_render(self, K, Rt, preds)     <- AutoEncoder.render, ca_code/models/rgca.py:112-151
  prim_decoder_forward(self, embs, geom, ...) <- PrimDecoder.forward, ca_code/models/rgca.py:466-620
Both are written as unbound methods so goliath_amd.dropin.patch_rgca() can install them on the
reference classes; the decoder convolutions (`encmod`, `viewmod`, `vnocond_mod`, `vcond_mod`) and the
geometry module stay the reference's PyTorch modules with their parameter names (checkpoints load
unchanged) -- only what follows them is replaced by the fused HIP shading tail.
"""
from typing import Any, Dict, List, Optional

import torch as th
import torch.nn.functional as F

import os

from .render_gs import render_batch
from .shade import shading_tail
from .tail import fused_tail
from .views import ViewSet


def _fuse_projection() -> bool:
    """GOLIATH_FUSED_PROJECTION=0: shade and project as separate kernels (rounds 1-3)."""
    return os.environ.get("GOLIATH_FUSED_PROJECTION", "1") != "0"


def autoencoder_forward(
    self,
    head_pose: th.Tensor,
    campos: th.Tensor,
    registration_vertices: th.Tensor,
    color: th.Tensor,
    light_intensity: th.Tensor,
    light_pos: th.Tensor,
    n_lights: th.Tensor,
    K: th.Tensor,
    Rt: th.Tensor,
    background: Optional[th.Tensor] = None,
    is_fully_lit_frame: Optional[th.Tensor] = None,
    camera_id: Optional[List[str]] = None,
    frame_id: Optional[th.Tensor] = None,
    iteration: Optional[int] = None,
    preconv_envmap: Optional[th.Tensor] = None,
    lightrot: Optional[th.Tensor] = None,
    **kwargs,
) -> Dict[str, Any]:
    """AutoEncoder.forward (ca_code/models/rgca.py:153-253) with the same parameters (train.py's `filter_inputs`
    introspects them) and the same output keys.  The sub-modules (`encoder`, `geomdecoder`, `decoder`, `cal`,
    `learn_blur`) are the reference's own with their parameters; what changes is how the steps run: head-relative
    transforms as in :175-195, decode (the fused tail when `PrimDecoder.forward` is patched), ONE batched render, and
    ONE fused image pass for calibration + background composite + learnable blur (goliath_amd/imgtail.py) instead of a
    per-view Python loop, three elementwise kernels and two padded depthwise convolutions."""
    import ca_code.utils.sh as sh  # the reference's SH basis (a handful of lights per view)

    from .imgtail import autoencoder_image_tail

    light_intensity = light_intensity.expand(-1, -1, 3)
    rot, trans = head_pose[:, :3, :3], head_pose[:, :3, 3]
    bottom = th.zeros_like(head_pose[:, :1, :])
    bottom[:, 0, 3] = 1.0
    headrel_Rt = Rt @ th.cat([head_pose, bottom], dim=1)
    headrel_campos = ((campos - trans)[:, None] @ rot)[:, 0]
    headrel_light_pos = (light_pos - trans[:, None]) @ rot
    sh_coeffs = sh.dir2sh_torch(self.n_diff_sh, F.normalize(headrel_light_pos, p=2, dim=-1))
    headrel_light_sh = (sh_coeffs[:, :, None] * light_intensity[..., None]).sum(dim=1)
    if lightrot is not None:
        lightrot = lightrot @ rot
    enc_preds = self.encoder(registration_vertices, color)
    embs = enc_preds["embs"]
    geom = self.geomdecoder(embs)["face_geom"]
    # the cameras are known before the decoder runs: its shading kernel projects the Gaussians it produces (views.py), and
    # self.render below starts at the tile count.  Handed over as an attribute: PrimDecoder.forward keeps its signature.
    self.decoder._goliath_view_set = ViewSet(K, headrel_Rt, self.height, self.width) if _fuse_projection() else None
    try:
        dec_preds = self.decoder(embs, geom, headrel_campos, light_intensity, headrel_light_pos, headrel_light_sh,
                                 n_lights, preconv_envmap, lightrot)
    finally:
        self.decoder._goliath_view_set = None
    preds = {"geom": geom, "headrel_light_sh": headrel_light_sh, **enc_preds, **dec_preds}
    rgb, alpha, depth = self.render(K, headrel_Rt, preds)
    preds.pop("projected", None)   # (consumed by the render; the returned dict has the reference's keys)
    if preconv_envmap is not None and "envbg" in kwargs:
        # visualisation-only branch (run_vis_relight.py): calibrate / composite like the reference, then the env-map
        # background and the diffuse / specular breakdown renders (rgca.py:232-245) with the reference's own helper
        from ca_code.utils.envmap import compose_envmap

        rgb, _ = autoencoder_image_tail(_NoBlur(self), rgb, alpha, camera_id, background, is_fully_lit_frame)
        rgbs = [compose_envmap(rgb, alpha, kwargs["envbg"], K, Rt)]
        for key in ("diff_color", "spec_color"):
            preds["color"] = preds[key].clamp(min=0.0)
            part, alpha, depth = self.render(K, headrel_Rt, preds)  # (alpha / depth of the last render are returned)
            rgbs.append(part)
        rgb = th.cat(rgbs, -1)
        preds.update(rgb=rgb, alpha=alpha, depth=depth)
        if getattr(self, "learn_blur_enabled", False):
            preds["rgb"] = autoencoder_image_tail(_OnlyBlur(self), rgb, alpha, camera_id)[0]
            preds["learn_blur_weights"] = self.learn_blur.reg(camera_id)
        return preds
    rgb, blur_reg = autoencoder_image_tail(self, rgb, alpha, camera_id, background, is_fully_lit_frame)
    preds.update(rgb=rgb, alpha=alpha, depth=depth)
    if blur_reg is not None:
        preds["learn_blur_weights"] = blur_reg
    return preds


class _View:
    """The attributes autoencoder_image_tail reads, with one stage switched off."""

    def __init__(self, model, cal, blur):
        self.training = model.training
        self.cal_enabled = cal and getattr(model, "cal_enabled", False)
        self.learn_blur_enabled = blur and getattr(model, "learn_blur_enabled", False)
        self.cal = getattr(model, "cal", None)
        self.learn_blur = getattr(model, "learn_blur", None)


def _NoBlur(model):
    return _View(model, cal=True, blur=False)


def _OnlyBlur(model):
    v = _View(model, cal=False, blur=True)
    v.training = False  # no second background composite
    return v


def autoencoder_render(self, K: th.Tensor, Rt: th.Tensor, preds: Dict[str, Any]):
    """All B views in one launch sequence; K stays on the device (no .item())."""
    return render_batch(K, Rt, preds, self.height, self.width)


def random_light_sh(sh_fn, n_diff_sh: int, batch: int, device, dtype):
    """The training-only random light of rgca.py:590-612 (no_grad): returns (light_dir[B,1,3],
    light_sh[B,3,(n+1)^2]) for unit intensity.  `sh_fn` is the reference's sh.dir2sh_torch."""
    with th.no_grad():
        light_dir = F.normalize(th.rand(batch, 1, 3, device=device, dtype=dtype) - 0.5, p=2, dim=-1)
        coeffs = sh_fn(n_diff_sh, light_dir)                      # [B,1,C]
        light_sh = coeffs.sum(dim=1)[:, None, :].expand(-1, 3, -1).contiguous()
    return light_dir, light_sh


def _can_fuse_tail(dec) -> bool:
    """The last layers are un-fused weight-normalised transposed convs (parameters weight_v / weight_g / bias)
    inside nn.Sequential stacks, and GOLIATH_FUSED_TAIL is not 0."""
    if os.environ.get("GOLIATH_FUSED_TAIL", "1") == "0":
        return False
    mods = (dec.vnocond_mod, dec.vcond_mod)
    if not all(isinstance(m, th.nn.Sequential) for m in mods):
        return False
    return all(all(hasattr(m[-1], a) for a in ("weight_v", "weight_g", "bias")) and
               tuple(m[-1].weight_v.shape[2:]) == (4, 4) for m in mods)


def prim_decoder_forward(self, embs: th.Tensor, geom: th.Tensor, headrel_campos: th.Tensor,
                         light_intensity: th.Tensor, headrel_light_pos: th.Tensor, headrel_light_sh: th.Tensor,
                         n_lights: th.Tensor, preconv_envmap: Optional[th.Tensor] = None,
                         lightrot: Optional[th.Tensor] = None):
    B = embs.shape[0]
    # uv position / normal maps and the two decoders: unchanged PyTorch (rgca.py:483-503)
    postex = self.geo_fn.to_uv(geom)
    tn = F.normalize(self.geo_fn.to_uv(self.geo_fn.vn(geom)), dim=1)
    z = self.encmod(embs).view(-1, 256, 8, 8)
    view = self.viewmod(F.normalize(headrel_campos, dim=1))[:, :, None, None].expand(-1, -1, 8, 8)
    zv = th.cat([z, view], dim=1)
    fuse = _can_fuse_tail(self)
    if not fuse:
        f_vnocond, f_vcond = self.vnocond_mod(z), self.vcond_mod(zv)

    light_sh_rand, light_dir = None, None
    if self.training:
        import ca_code.utils.sh as sh  # the reference's SH basis (per-light, tiny)

        light_dir, light_sh_rand = random_light_sh(sh.dir2sh_torch, self.diff_sh_degree, B, embs.device,
                                                   headrel_light_pos.dtype)
    kw = dict(light_intensity=light_intensity, headrel_light_pos=headrel_light_pos, n_lights=n_lights,
              preconv_envmap=preconv_envmap, lightrot=lightrot, light_sh_rand=light_sh_rand,
              n_color_sh=self.color_sh_degree, n_diff_sh=self.diff_sh_degree,
              views=getattr(self, "_goliath_view_set", None))
    if fuse:  # the 125-channel activation is never materialised (goliath_amd/tail.py)
        preds = fused_tail(self.vnocond_mod[-1], self.vcond_mod[-1], self.vnocond_mod[:-1](z), self.vcond_mod[:-1](zv),
                           postex, tn, self.albedo, headrel_light_sh, headrel_campos, **kw)
    else:
        preds = shading_tail(f_vnocond, f_vcond, postex, tn, self.albedo, headrel_light_sh, headrel_campos, **kw)
    if self.training:
        with th.no_grad():
            preds["cos_weight"] = (light_dir * preds["spec_nml"]).sum(dim=-1, keepdim=True)
    return preds
