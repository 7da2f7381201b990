"""CPU: oracle/shade_ref.py (restatement of rgca.py:505-618) vs golden vectors produced by the
reference's OWN PyTorch code (tests/golden/make_shade_golden.py -> shade_golden.npz).  PINNED."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cref, shade_ref
from scenes import rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shade_golden.npz")


def load_golden():
    z = np.load(GOLD)
    return {k: torch.from_numpy(z[k]) for k in z.files}


class OracleSG(torch.autograd.Function):
    """sg.cu forward/backward (C oracle) as an autograd node, like sgutils.py:17-63."""

    @staticmethod
    def forward(ctx, dirs, sig, lv, lp, pp, nl, w_type):
        ctx.save_for_backward(dirs, sig, lv, lp, pp, nl)
        ctx.w = w_type
        return cref.evaluate_gaussian_fwd(dirs, sig, lv, lp, pp, nl, w_type)

    @staticmethod
    def backward(ctx, g):
        dirs, sig, lv, lp, pp, nl = ctx.saved_tensors
        gd, gs, _ = cref.evaluate_gaussian_bwd(dirs, sig, lv, lp, pp, nl, g.contiguous(), ctx.w)
        return gd, gs, None, None, None, None, None


def oracle_inputs(G):
    leaf = {n: G[f"in/{n}"].clone().requires_grad_(True) for n in ("f_vnocond", "f_vcond", "postex", "tn_raw", "albedo")}
    return leaf


def run_oracle(G, tag, leaf, sg_eval=OracleSG.apply):
    env = tag.startswith("env")
    train = tag.endswith("train")
    kw = {}
    if env:
        kw.update(envmips=[G[f"in/mip{i}"] for i in range(4)], lightrot=G["in/lightrot"])
    else:
        kw.update(light_intensity=G["in/light_intensity"], light_pos=G["in/light_pos"],
                  n_lights=G["in/n_lights"].int())
    if train:
        kw["light_sh_rand"] = G[f"{tag}/in/light_sh_rand"][:, None, :].expand(-1, 3, -1)
    return shade_ref.shade(leaf["f_vnocond"], leaf["f_vcond"], leaf["postex"], F.normalize(leaf["tn_raw"], dim=1),
                           leaf["albedo"], G["in/light_sh"], G["in/campos"], sg_eval=sg_eval, **kw)


@pytest.mark.parametrize("tag", ["sg_eval", "sg_train", "env_eval"])
def test_shade_oracle_reproduces_reference(tag):
    G = load_golden()
    leaf = oracle_inputs(G)
    out = run_oracle(G, tag, leaf)
    loss = 0.0
    for k, v in out.items():
        ref = G[f"{tag}/out/{k}"]
        assert rel_l2(v.reshape(ref.shape), ref) < 2e-6, (k, rel_l2(v.reshape(ref.shape), ref))
        if v.requires_grad and f"w/{k}" in G:
            loss = loss + (v.reshape(ref.shape) * G[f"w/{k}"]).sum()
    if tag.endswith("train"):
        # cos_weight is plain torch in the reference (rgca.py:603-605)
        cw = (G[f"{tag}/in/light_dir_rand"] * out["spec_nml"]).sum(-1, keepdim=True)
        assert rel_l2(cw, G[f"{tag}/out/cos_weight"]) < 2e-6
    loss.backward()
    for n, t in leaf.items():
        assert rel_l2(t.grad, G[f"{tag}/grad/{n}"]) < 2e-5, (n, rel_l2(t.grad, G[f"{tag}/grad/{n}"]))
