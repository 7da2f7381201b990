"""Render wrappers with the reference's signatures, running on the fused HIP path.

  render(...)        <- ca_code/utils/render_gsplat.py:13-108 (one view; same arguments, same dict)
  render_batch(...)  <- ca_code/models/rgca.py:112-151 AutoEncoder.render (all B views at once,
                        no Python view loop, no K[b,.,.].item() host syncs)
"""
from typing import Any, Dict, Optional

import torch as th

from .splat import render_views
from .views import ViewSet


def view_set(K: th.Tensor, Rt: th.Tensor, height: int, width: int) -> ViewSet:
    """The cameras AutoEncoder.render will be called with, for shading_tail(..., views=...): the projection then runs
    inside the shading kernel (K[B,3,3], Rt[B,3,4]; pass the SAME tensors to render_batch)."""
    return ViewSet(K, Rt, height, width)


def render(cam_img_w: int, cam_img_h: int, fx: float, fy: float, cx: float, cy: float, Rt: th.Tensor,
           primpos: th.Tensor, primqvec: th.Tensor, primscale: th.Tensor, opacity: th.Tensor,
           colors: th.Tensor, return_depth: bool = True, bg_color: Optional[th.Tensor] = None,
           block_width: int = 16, global_scale: float = 1.0, z_near: float = 0.1):
    """Single-view render; returns {"render"[3,H,W], "final_T"[1,H,W], "alpha"[1,H,W], "radii"[N],
    "depth"[1,H,W]} exactly like the reference wrapper, from ONE fused colour+depth raster pass."""
    if block_width != 16:
        raise NotImplementedError("block_width must be 16")
    dev = Rt.device
    intr = th.tensor([[fx, fy, cx, cy]], dtype=th.float32).to(dev, non_blocking=True)
    out = render_views(primpos.reshape(1, -1, 3), primscale.reshape(1, -1, 3), primqvec.reshape(1, -1, 4),
                       opacity.reshape(1, -1), colors.reshape(1, -1, 3), Rt.reshape(1, -1)[:, :12], intr,
                       cam_img_h, cam_img_w, background=bg_color, glob_scale=global_scale, clip_thresh=z_near,
                       with_depth=return_depth)
    res = {"render": out["render"][0], "final_T": out["final_T"][0], "alpha": out["alpha"][0],
           "radii": out["radii"][0]}
    if return_depth:
        res["depth"] = out["depth"][0]
    return res


def render_batch(K: th.Tensor, Rt: th.Tensor, preds: Dict[str, Any], height: int, width: int,
                 l1_target: Optional[th.Tensor] = None, l1_mask: Optional[th.Tensor] = None):
    """AutoEncoder.render semantics (rgca.py:112-151): rgb[B,3,H,W], alpha = 1 - T.detach(),
    depth / alpha.clamp(0.05, 1).  K[B,3,3] and Rt[B,3,4] stay on the device.
    With l1_target (and optionally l1_mask) a fourth value is returned: the masked L1 loss of rgb against it
    (rgb_l1, ca_code/loss/__init__.py:391-411), fused into the raster passes."""
    pr = preds.get("projected")
    if pr is not None and pr.valid_for(preds, K, Rt, height, width) and (pr.views.glob_scale, pr.views.clip_thresh) == (1.0, 0.1):
        # the shading kernel already projected these Gaussians onto these cameras (shading_tail(..., views=...)): start at
        # the tile count; the backward hands its gradient records to the shading backward
        out = render_views(None, None, None, None, None, None, None, height, width, with_depth=True, l1_target=l1_target,
                           l1_mask=l1_mask, raw_depth=False, projected=pr)
    else:
        intr = th.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], dim=-1)
        out = render_views(preds["primpos"], preds["primscale"], preds["primqvec"], preds["opacity"],
                           preds["color"], Rt, intr, height, width, with_depth=True, l1_target=l1_target, l1_mask=l1_mask,
                           raw_depth=False)  # only depth / alpha.clamp(0.05, 1) leaves AutoEncoder.render
    # alpha = 1 - T.detach() and depth / alpha.clamp(0.05, 1) are written by the raster kernel's epilogue
    if l1_target is not None:
        return out["render"], out["alpha"].detach(), out["depth_norm"], out["l1_loss"]
    return out["render"], out["alpha"].detach(), out["depth_norm"]
