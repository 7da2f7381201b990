#!/usr/bin/env python
"""Time gol_ssim_fwd / gol_ssim_bwd at the bench image size (8 views x 3 x 2048 x 1334)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from goliath_amd import _lib, losses

B, C, H, W = 8, 3, 2048, 1334
tgt = torch.rand(B, C, H, W, device="cuda")
pred = (tgt + 0.1 * torch.randn_like(tgt)).requires_grad_(True)
mask = (torch.rand(B, 1, H, W, device="cuda") > 0.2).float()
for _ in range(2):
    losses.ssim_image(pred, tgt, mask).backward()
torch.cuda.synchronize()
_lib.TIMING = []
for _ in range(10):
    pred.grad = None
    losses.ssim_image(pred, tgt, mask).backward()
torch.cuda.synchronize()
per = {}
for name, e0, e1 in _lib.TIMING:
    per.setdefault(name, []).append(e0.elapsed_time(e1))
n = B * C * H * W
alg = {"gol_ssim_fwd": (8 + 12) * n + 4 * B * H * W, "gol_ssim_bwd": (12 + 8 + 4) * n}
for k, v in per.items():
    ms = sum(v) / len(v)
    print(f"{k}: {ms:.3f} ms, algorithmic {alg[k] / 1e6:.0f} MB -> {alg[k] / ms / 1e6:.0f} GB/s")
