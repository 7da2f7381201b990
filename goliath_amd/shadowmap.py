"""URHand shadow-map lookup (host side) on top of gol_shadow_pcf.

  get_shadow_map(rl, Rt, K, verts, postex, nml=None)   <- ca_code/utils/shadowmap.py:17-96, same signature:
      the depth render stays the caller's `rl` (drtk RenderLayer in the reference -- third-party), everything per
      texel (projection, 3x3 PCF over 18 grid_samples, back-face blend) is one HIP kernel.
  shadow_pcf(depth, Rt, postex, nml, exp_scale)         the native form: postex / nml are [B,3,H,W] (NOT repeated per
      light), depth / Rt carry B*L light cameras, and exp(-x/8) of urhand.py:416 can be fused (exp_scale=8).
Forward only: the reference evaluates it under torch.no_grad() (ca_code/models/urhand.py:404,492).
"""
import torch

from . import _lib
from ._lib import c_int, fptr, stream_ptr



def shadow_pcf(depth, Rt, postex, nml=None, exp_scale=0.0, focal=1000.0):
    """depth[B*L,h,w], Rt[B*L,3,4], postex[B,3,H,W], nml[B,3,H,W]|None -> [B*L,1,H,W]."""
    import ctypes

    if not postex.is_cuda:
        raise _lib.GoliathHipError("shadow_pcf needs CUDA(HIP) tensors; there is no CPU path")
    B, _, H, W = postex.shape
    BL, dh, dw = depth.shape
    if BL % max(B, 1) != 0:
        raise ValueError("depth must hold B*L images")
    L = BL // B if B else 0
    c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    depth, Rt, postex, nml = c(depth), c(Rt), c(postex), c(nml)
    out = torch.empty(BL, 1, H, W, device=postex.device)
    f = ctypes.c_float
    with _lib.device_guard(postex.device):
        _lib.call("gol_shadow_pcf", c_int(B), c_int(L), c_int(H), c_int(W), c_int(dh), c_int(dw), fptr(depth), fptr(Rt),
                  f(focal), f(focal), f(dw / 2), f(dh / 2), fptr(postex), fptr(nml), f(exp_scale), fptr(out), stream_ptr())
    return out


def get_shadow_map(rl, Rt, K, verts, postex, nml=None):
    """Drop-in for ca_code.utils.shadowmap.get_shadow_map (K is ignored there too: it is overwritten, :21-26)."""
    batch = postex.shape[0]
    Kl = torch.eye(3, device=Rt.device)[None].repeat(Rt.shape[0], 1, 1)
    Kl[:, 0, 0] = Kl[:, 1, 1] = 1000.0
    Kl[:, 0, 2], Kl[:, 1, 2] = rl.w / 2, rl.h / 2
    tex = torch.empty(batch, 1, 1024, 1024, device=Rt.device)          # shadowmap.py:39-43: dummy texture for the renderer
    if isinstance(verts, (list, tuple)):
        z = torch.empty(batch, 1, 256, 256, device=Rt.device)
        tex = [z, z, tex]
    depth = rl(verts, tex, Kl, Rt)["depth_img"]                        # [B, h, w]
    return shadow_pcf(depth, Rt, postex, nml)                          # one light camera per batch element (L = 1)
