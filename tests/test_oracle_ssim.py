"""CPU: the SSIM oracle (oracle/ssim_ref.py) reproduces the golden vectors made by the reference's own ssim()."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ssim_golden.npz")


def load_cases():
    G = np.load(GOLD)
    tags = sorted({k.split("/")[0] for k in G.files})
    return {t: {k.split("/")[1]: torch.from_numpy(G[k]) for k in G.files if k.startswith(t + "/")} for t in tags}


def test_ssim_oracle_matches_reference_golden():
    from oracle import ssim_ref

    for tag, c in load_cases().items():
        pred = c["pred"].clone().requires_grad_(True)
        val = ssim_ref.ssim(c["target"], pred, c.get("mask"))
        (grad,) = torch.autograd.grad(val, pred)
        assert abs(float(val) - float(c["value"])) < 1e-6, tag
        assert float((grad - c["grad"]).norm() / c["grad"].norm()) < 1e-5, tag
        assert abs(float(ssim_ref.rgb_ssim(c["pred"], c["target"], c.get("mask"))) - (1 - float(c["value"]))) < 1e-6
