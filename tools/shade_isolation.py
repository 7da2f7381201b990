#!/usr/bin/env python
"""Diagnostic (GPU): shade backward in isolation at config-2 size -- HIP vs the fp32 oracle vs the SAME oracle in fp64, on
identical inputs and an identical upstream gradient.  For the Gaussians where HIP and the fp32 oracle disagree most: how far
is the fp32 oracle from its own fp64 evaluation (i.e. is the disagreement the conditioning of the reference's formulation)?"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from goliath_amd import shade  # noqa: E402
from oracle import shade_ref  # noqa: E402

cfg = dict(bench.CFG, views_per_gpu=1)
N = cfg["gaussians"]
cpu = bench.make_inputs(cfg, "cpu")
gen = torch.Generator().manual_seed(0)
up = {"color": torch.randn(1, N, 3, generator=gen), "primpos": torch.randn(1, N, 3, generator=gen) * 0.01}
leaves = ("f_vn", "f_vc", "postex", "tn")


def run_oracle(dtype):
    t = {k: (v.detach().to(dtype).requires_grad_(v.requires_grad) if torch.is_tensor(v) else [m.to(dtype) for m in v])
         for k, v in cpu.items()}
    pr = shade_ref.shade(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                         envmips=t["mips"], lightrot=t["lightrot"])
    ((pr["color"] * up["color"].to(dtype)).sum() + (pr["primpos"] * up["primpos"].to(dtype)).sum()).backward()
    return {k: t[k].grad.double().reshape(-1, N) for k in leaves}, {k: v.detach() for k, v in pr.items()}, t


g32, pr32, t32 = run_oracle(torch.float32)
g64, _, _ = run_oracle(torch.float64)
g = {k: (v.detach().cuda().requires_grad_(v.requires_grad) if torch.is_tensor(v) else [m.cuda() for m in v]) for k, v in cpu.items()}
ph = shade.shading_tail(g["f_vn"], g["f_vc"], g["postex"], g["tn"], g["albedo"], g["light_sh"], g["campos"],
                        preconv_envmap=g["mips"], lightrot=g["lightrot"])
((ph["color"] * up["color"].cuda()).sum() + (ph["primpos"] * up["primpos"].cuda()).sum()).backward()
gh = {k: g[k].grad.double().cpu().reshape(-1, N) for k in leaves}
with torch.no_grad():
    view = F.normalize(pr32["primpos"] - cpu["campos"][:, None], dim=-1)
    n = pr32["spec_nml"]
    refl = view - 2 * (view * n).sum(-1, keepdim=True) * n
    r = torch.einsum("bxy,bny->bnx", cpu["lightrot"], refl)[0]
rep = {}
for k in leaves:
    e_h = (gh[k] - g32[k]).pow(2).sum(0)          # HIP vs fp32 oracle
    e_o = (g32[k] - g64[k]).pow(2).sum(0)         # fp32 oracle vs fp64 oracle
    e_h64 = (gh[k] - g64[k]).pow(2).sum(0)        # HIP vs fp64 oracle
    tot = g64[k].pow(2).sum()
    top = e_h.topk(200).indices
    rep[k] = {"hip_vs_o32": float((e_h.sum() / tot).sqrt()), "o32_vs_o64": float((e_o.sum() / tot).sqrt()),
              "hip_vs_o64": float((e_h64.sum() / tot).sqrt()),
              "top200_share_of_hip_vs_o32": float(e_h[top].sum() / e_h.sum()),
              "top200_where_o32_is_as_far_from_o64": int((e_o[top] >= 0.1 * e_h[top]).sum()),
              "top200_abs_ry_min_median": [float(r[top, 1].abs().min()), float(r[top, 1].abs().median())],
              "examples": [{"g": int(i), "hip_o32": float(e_h[i].sqrt()), "o32_o64": float(e_o[i].sqrt()),
                            "hip_o64": float(e_h64[i].sqrt()), "ref": float(g64[k][:, i].norm()), "r_y": float(r[i, 1])}
                           for i in top[:12]]}
print("SHADE_ISOLATION " + json.dumps(rep, indent=1))
