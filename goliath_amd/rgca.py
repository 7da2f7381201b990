"""Model-level entry points of the RGCA render path with the reference's signatures.

  autoencoder_render(self, K, Rt, preds)     <- AutoEncoder.render, ca_code/models/rgca.py:112-151
  prim_decoder_forward(self, embs, geom, ...) <- PrimDecoder.forward, ca_code/models/rgca.py:466-620
Both are written as unbound methods so goliath_amd.dropin.patch_rgca() can install them on the
reference classes; the decoder convolutions (`encmod`, `viewmod`, `vnocond_mod`, `vcond_mod`) and the
geometry module stay the reference's PyTorch modules with their parameter names (checkpoints load
unchanged) -- only what follows them is replaced by the fused HIP shading tail.
"""
from typing import Any, Dict, Optional

import torch as th
import torch.nn.functional as F

import os

from .render_gs import render_batch
from .shade import shading_tail
from .tail import fused_tail


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
              n_color_sh=self.color_sh_degree, n_diff_sh=self.diff_sh_degree)
    if fuse:  # the 125-channel activation is never materialised (goliath_amd/tail.py)
        preds = fused_tail(self.vnocond_mod[-1], self.vcond_mod[-1], self.vnocond_mod[:-1](z), self.vcond_mod[:-1](zv),
                           postex, tn, self.albedo, headrel_light_sh, headrel_campos, **kw)
    else:
        preds = shading_tail(f_vnocond, f_vcond, postex, tn, self.albedo, headrel_light_sh, headrel_campos, **kw)
    if self.training:
        with th.no_grad():
            preds["cos_weight"] = (light_dir * preds["spec_nml"]).sum(dim=-1, keepdim=True)
    return preds
