"""CPU oracle for the SSIM loss -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Restates /root/reference/ca_code/utils/ssim.py:15-65 (gaussian window 11 / sigma 1.5, five depthwise
convolutions with zero padding, C1 = 0.01^2, C2 = 0.03^2, masked mean) and rgb_ssim
(/root/reference/ca_code/loss/__init__.py:478-494) in plain PyTorch; gradients by autograd.
Pinned: tests/golden/ssim_golden.npz was produced by the reference's own ssim() (tests/golden/make_ssim_golden.py).
"""
import math

import torch
import torch.nn.functional as F


def window(channel, size=11, sigma=1.5):
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)])  # ssim.py:15-17
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float()[None, None].expand(channel, 1, size, size).contiguous()                # ssim.py:19-23


def ssim(img1, img2, mask=None, size=11):
    C = img1.shape[-3]
    w = window(C, size).to(img1)
    conv = lambda t: F.conv2d(t, w, padding=size // 2, groups=C)
    mu1, mu2 = conv(img1), conv(img2)                                                                 # ssim.py:26-27
    s11, s22, s12 = conv(img1 * img1) - mu1 * mu1, conv(img2 * img2) - mu2 * mu2, conv(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))    # ssim.py:40
    if mask is None:
        return m.mean()
    mask = mask.expand(-1, m.shape[1], -1, -1)                                                        # ssim.py:43
    return (m * mask).sum() / mask.sum().clamp(min=1)                                                 # ssim.py:48


def rgb_ssim(pred, target, mask=None, normalize_mask=True):
    if mask is None or normalize_mask:
        return 1.0 - ssim(target, pred, mask)
    return 1.0 - ssim(mask * target, mask * pred)
