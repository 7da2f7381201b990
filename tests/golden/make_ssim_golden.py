"""Generate tests/golden/ssim_golden.npz with the REFERENCE's own ssim() (/root/reference/ca_code/utils/ssim.py,
pure PyTorch, runs on CPU) and autograd for the gradient.  Build container only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from ca_code.utils.ssim import ssim  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    for tag, (B, C, H, W, mc) in {"a": (2, 3, 45, 70, 1), "b": (1, 3, 33, 32, 0), "c": (2, 1, 20, 97, 1),
                                  "d": (1, 3, 64, 40, 3)}.items():
        g = torch.Generator().manual_seed(ord(tag))
        target = torch.rand(B, C, H, W, generator=g)
        pred = (target + 0.2 * torch.randn(B, C, H, W, generator=g)).clamp(0, 1.2).requires_grad_(True)
        mask = (torch.rand(B, mc, H, W, generator=g) > 0.3).float() if mc else None
        val = ssim(target, pred, mask=mask)          # argument order of rgb_ssim (loss/__init__.py:492)
        (grad,) = torch.autograd.grad(val, pred)
        out[f"{tag}/target"], out[f"{tag}/pred"] = target.numpy(), pred.detach().numpy()
        if mask is not None:
            out[f"{tag}/mask"] = mask.numpy()
        out[f"{tag}/value"], out[f"{tag}/grad"] = val.detach().numpy(), grad.numpy()
    np.savez_compressed(os.path.join(HERE, "ssim_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
