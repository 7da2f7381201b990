"""Image losses on top of the C ABI ("next" row).

  rgb_l1(preds, targets, ...)   <- ca_code/loss/__init__.py:391-411  (same signature / keys)
  l1_image(pred, target, mask)  the fused op: one read pass forward, one read + one write pass backward
"""
from typing import Optional

import torch

from . import _lib
from ._lib import c_int, fptr, stream_ptr


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask):
        B, C = pred.shape[:2]
        HW = pred[0, 0].numel()
        mask_c = 0 if mask is None else mask.shape[1]
        nb = _lib.load().gol_l1_blocks(HW)
        partial = torch.empty(B * C * nb, device=pred.device)
        with _lib.device_guard(pred.device):
            _lib.call("gol_l1_fwd", c_int(B), c_int(C), c_int(HW), c_int(mask_c), fptr(pred), fptr(target), fptr(mask),
                      fptr(partial), stream_ptr())
        ctx.save_for_backward(pred, target, mask)
        return partial.sum() / (B * C * HW)

    @staticmethod
    def backward(ctx, g):
        pred, target, mask = ctx.saved_tensors
        B, C = pred.shape[:2]
        HW = pred[0, 0].numel()
        mask_c = 0 if mask is None else mask.shape[1]
        out = torch.empty_like(pred)
        g = g.to(torch.float32).reshape(1).contiguous()
        with _lib.device_guard(pred.device):
            _lib.call("gol_l1_bwd", c_int(B), c_int(C), c_int(HW), c_int(mask_c), fptr(pred), fptr(target), fptr(mask),
                      fptr(g), fptr(out), stream_ptr())
        return out, None, None


def l1_image(pred: torch.Tensor, target: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mean(|(pred - target) * mask|) for [B,C,H,W] images; mask [B,1,H,W] or [B,C,H,W] or None."""
    if not pred.is_cuda:
        raise _lib.GoliathHipError("l1_image needs CUDA(HIP) tensors; there is no CPU path")
    c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    return _L1.apply(pred.to(torch.float32).contiguous(), c(target), c(mask))


def erode_mask(mask: torch.Tensor, ks: int) -> torch.Tensor:
    """ca_code/utils/image.py:393-422 erode(): a pixel survives iff every pixel of its ks x ks window (zero-padded
    complement) is set.  Returns a float mask of 0 / 1.  (A ks x ks box filter of the complement on a tiny tensor:
    plain torch, it has no gradient and is not on the per-pixel hot path.)"""
    assert ks % 2 == 1
    x = mask.to(torch.float32)
    if x.dim() == 3:
        x = x[:, None]
    flip = 1.0 - x
    w = torch.ones(1, 1, ks, ks, device=x.device)
    hit = torch.nn.functional.conv2d(flip.reshape(-1, 1, *x.shape[-2:]), w, padding=ks // 2) > 0
    return 1.0 - hit.reshape(x.shape).to(torch.float32)


def rgb_l1(preds, targets, src_key: str = "rendered_rgb", tgt_key: str = "image", mask_key: str = "image_mask",
           ddisc_key: str = "depth_disc_mask", mask_erode: Optional[int] = None):
    """Same semantics as the reference's rgb_l1 (ca_code/loss/__init__.py:391-411), incl. `mask_erode`."""
    mask = targets.get(mask_key, preds.get(mask_key, None))
    if mask_erode is not None:
        if mask is None:
            mask = torch.ones_like(preds[src_key])
        mask = (erode_mask(mask, mask_erode) > 0).to(torch.float32)  # .to(th.bool) in the reference
    if ddisc_key in preds:
        d = preds[ddisc_key]
        inv = (~d).float() if d.dtype == torch.bool else (1 - d)
        mask = inv if mask is None else mask * inv
    return l1_image(preds[src_key], targets[tgt_key], None if mask is None else mask.float())


class _Ssim(torch.autograd.Function):
    """Masked-mean SSIM of (target, pred); differentiable in pred (ca_code/utils/ssim.py:25-65)."""

    @staticmethod
    def forward(ctx, pred, target, mask):
        B, C, H, W = pred.shape
        mask_c = 0 if mask is None else mask.shape[1]
        nb = _lib.load().gol_ssim_blocks(H, W)
        partial = torch.empty(B * C * nb, device=pred.device)
        dmap = torch.empty(3, B, C, H, W, device=pred.device) if ctx.needs_input_grad[0] else None
        with _lib.device_guard(pred.device):
            _lib.call("gol_ssim_fwd", c_int(B), c_int(C), c_int(H), c_int(W), c_int(mask_c), fptr(target), fptr(pred),
                      fptr(mask), fptr(partial), fptr(dmap), stream_ptr())
        if mask is None:
            denom = torch.full((), float(B * C * H * W), device=pred.device)
        else:  # ssim.py:44-49: the mask is expanded to the image's channels before it is summed
            denom = (mask.sum() * (C // mask_c)).clamp(min=1)
        ctx.save_for_backward(pred, target, dmap, denom)
        return partial.sum() / denom

    @staticmethod
    def backward(ctx, g):
        pred, target, dmap, denom = ctx.saved_tensors
        B, C, H, W = pred.shape
        out = torch.empty_like(pred)
        gs = (g.to(torch.float32) / denom).reshape(1).contiguous()
        with _lib.device_guard(pred.device):
            _lib.call("gol_ssim_bwd", c_int(B), c_int(C), c_int(H), c_int(W), fptr(target), fptr(pred), fptr(dmap),
                      fptr(gs), fptr(out), stream_ptr())
        return out, None, None


def ssim_image(pred: torch.Tensor, target: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ssim(target, pred, mask=mask) of ca_code/utils/ssim.py for [B,C,H,W] images (window 11, size_average)."""
    if not pred.is_cuda:
        raise _lib.GoliathHipError("ssim_image needs CUDA(HIP) tensors; there is no CPU path")
    c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    return _Ssim.apply(pred.to(torch.float32).contiguous(), c(target), c(mask))


def rgb_ssim(preds, targets, src_key: str = "rendered_rgb", tgt_key: str = "image", mask_key: str = "image_mask",
             normalize_mask: bool = True):
    """Same semantics as the reference's rgb_ssim (ca_code/loss/__init__.py:478-494)."""
    mask = targets.get(mask_key, preds.get(mask_key, None))
    if mask is None or normalize_mask:
        return 1.0 - ssim_image(preds[src_key], targets[tgt_key], None if mask is None else mask.float())
    return 1.0 - ssim_image(mask * preds[src_key], mask * targets[tgt_key])
