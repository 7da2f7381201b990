"""The image tail of `AutoEncoder.forward` between the rasterizer and the losses, fused.

  image_tail(rgb, ...)                 one HIP pass forward / one backward (csrc/imgtail.hip) for
      colour calibration  CalV5.forward          ca_code/nn/color_cal.py:211-241
      background          rgb + (1 - alpha) * bg  ca_code/models/rgca.py:226-230
      LearnableBlur.forward                       ca_code/nn/dof_cal.py:44-56
  cal_v5_matrix(cal, cam_idxs)         CalV5's per-view gain / bias as [B,3,3] + [B,3] tensors, built on the device
                                       (the reference loops over the views on the host and compares camera indices
                                       there -- one sync per view)
  autoencoder_image_tail(self, ...)    what rgca.py:223-231 + :249-251 do, on the reference module's own parameters
"""
import ctypes
from typing import List, Optional

import torch

from . import _lib
from ._lib import c_int, fptr, stream_ptr


class _ImageTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, alpha, bg, bg_scale, cal_M, cal_b, blur_w):
        B, _, H, W = rgb.shape
        out = torch.empty_like(rgb)
        with _lib.device_guard(rgb.device):
            _lib.call("gol_imgtail_fwd", c_int(B), c_int(H), c_int(W), fptr(rgb), fptr(alpha), fptr(bg), fptr(bg_scale),
                      fptr(cal_M), fptr(cal_b), fptr(blur_w), fptr(out), stream_ptr())
        ctx.save_for_backward(rgb, alpha, bg, bg_scale, cal_M, cal_b, blur_w)
        return out

    @staticmethod
    def backward(ctx, g):
        rgb, alpha, bg, bg_scale, cal_M, cal_b, blur_w = ctx.saved_tensors
        B, _, H, W = rgb.shape
        g = g.to(torch.float32).contiguous()
        g_rgb = torch.empty_like(rgb)
        fn = _lib.load().gol_imgtail_partial_floats
        fn.restype = ctypes.c_int64
        n = int(fn(c_int(B), c_int(H), c_int(W)))
        partials = torch.empty(B, max(n // max(B, 1) // 16, 1), 16, device=rgb.device)
        with _lib.device_guard(rgb.device):
            _lib.call("gol_imgtail_bwd", c_int(B), c_int(H), c_int(W), fptr(rgb), fptr(alpha), fptr(bg), fptr(bg_scale),
                      fptr(cal_M), fptr(cal_b), fptr(blur_w), fptr(g), fptr(g_rgb), fptr(partials), stream_ptr())
        s = partials.sum(1)  # [B,16]: blur weights 0..2 | bias 3..5 | M 6..14 (per-workgroup partial sums, no atomics)
        need = ctx.needs_input_grad
        return (g_rgb if need[0] else None, None, None, None,
                s[:, 6:15].reshape(B, 3, 3) if cal_M is not None and need[4] else None,
                s[:, 3:6] if cal_b is not None and need[5] else None,
                s[:, 0:3] if blur_w is not None and need[6] else None)


def image_tail(rgb: torch.Tensor, alpha: Optional[torch.Tensor] = None, bg: Optional[torch.Tensor] = None,
               bg_scale: Optional[torch.Tensor] = None, cal_M: Optional[torch.Tensor] = None,
               cal_b: Optional[torch.Tensor] = None, blur_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = blur(cal(rgb) + (1 - alpha) * bg * bg_scale) for rgb[B,3,H,W]:
    cal(rgb)[c] = sum_j cal_M[b,c,j] rgb[j] + cal_b[b,c] (omitted: identity); alpha[B,1,H,W] (no gradient, like the
    detached alpha of rgca.py:137), bg[B,3,H,W], bg_scale[B] (omitted: 1); blur_weights[B,3] = the softmaxed
    LearnableBlur weights (identity, 3x3, 7x7; omitted: no blur).  Gradients reach rgb, cal_M, cal_b, blur_weights."""
    if not rgb.is_cuda:
        raise _lib.GoliathHipError("image_tail needs CUDA(HIP) tensors; there is no CPU path")
    c = lambda t: None if t is None else t.to(torch.float32).contiguous()
    B, _, H, W = rgb.shape
    if alpha is not None:
        alpha = c(alpha.detach()).reshape(B, H, W)
        bg = c(bg.detach())
    if (cal_M is None) != (cal_b is None):
        raise ValueError("cal_M and cal_b go together")
    return _ImageTail.apply(c(rgb), alpha, bg, None if bg_scale is None else c(bg_scale.detach()), c(cal_M), c(cal_b),
                            c(blur_weights))


def cal_v5_matrix(cal, cam_idxs: torch.Tensor):
    """CalV5.forward (color_cal.py:211-241) as data: per view a 3x3 gain matrix and a bias.  identity camera -> (I, 0)
    without gradient; grey cameras -> three identical rows w and bias sum(b); others -> diag(w), b.  The reference's
    gradient hook (params grads scaled by gs_lrscale / col_lrscale in training) is reproduced by scaling the gradient
    path, not the value."""
    params = cal.holder(cam_idxs)                                       # [B,6], ParamHolder (torchutils.py:85-127)
    dev = params.device
    is_id = cam_idxs == int(cal.identity_idx)
    grey = torch.as_tensor(list(cal.grey_idxs), device=dev, dtype=cam_idxs.dtype)
    is_grey = torch.isin(cam_idxs, grey) if grey.numel() else torch.zeros_like(is_id)
    if cal.training and params.requires_grad:
        s = torch.where(is_grey, float(cal.gs_lrscale), float(cal.col_lrscale))[:, None].to(params.dtype)
        params = params * s + (params * (1.0 - s)).detach()             # same value, gradient scaled by s
    w, b = params[:, :3], params[:, 3:]
    M = torch.where(is_grey[:, None, None], w[:, None, :].expand(-1, 3, -1), torch.diag_embed(w))
    bias = torch.where(is_grey[:, None], b.sum(-1, keepdim=True).expand(-1, 3), b)
    eye = torch.eye(3, device=dev, dtype=params.dtype).expand_as(M)
    M = torch.where(is_id[:, None, None], eye, M)
    bias = torch.where(is_id[:, None], torch.zeros_like(bias), bias)
    return M, bias


def autoencoder_image_tail(self, rgb: torch.Tensor, alpha: torch.Tensor, camera_id: Optional[List[str]],
                           background: Optional[torch.Tensor] = None,
                           is_fully_lit_frame: Optional[torch.Tensor] = None):
    """rgca.py:223-231 and :249-251 on the reference module `self` (an AutoEncoder): calibration (if cal_enabled),
    the training background composite, LearnableBlur (if learn_blur_enabled).  Returns (rgb, learn_blur_weights or None)."""
    M = b = bw = bg = bgs = a = None
    if getattr(self, "cal_enabled", False):
        M, b = cal_v5_matrix(self.cal, self.cal.name_to_idx(camera_id))
    if self.training and background is not None:
        a, bg = alpha, background[:, :3]
        if is_fully_lit_frame is not None:
            bgs = is_fully_lit_frame.reshape(-1).to(torch.float32)
    reg = None
    if getattr(self, "learn_blur_enabled", False):
        reg = self.learn_blur.reg(camera_id)                            # weights_raw[idxs] (dof_cal.py:37-42)
        bw = torch.softmax(reg, dim=-1)
    if M is None and bg is None and bw is None:
        return rgb, reg
    return image_tail(rgb, a, bg, bgs, M, b, bw), reg
