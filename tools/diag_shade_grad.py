#!/usr/bin/env python
"""Diagnostic (GPU): which Gaussians carry the shade-backward difference between the HIP kernel and the torch oracle at
BASELINE config 1, and why (distance of the env lookup to a pole / texel border / mip switch)."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from goliath_amd import shade  # noqa: E402
from oracle import shade_ref  # noqa: E402

cfg = dict(bench.CFG, views_per_gpu=1, slab=100, gaussians=10_000, height=512, width=512, focal=1150.0)
cpu = bench.make_inputs(cfg, "cpu")
gen = torch.Generator().manual_seed(0)
up = torch.randn(1, 10_000, 3, generator=gen)
leaves = ("f_vn", "f_vc", "postex", "tn")
# oracle
pr = shade_ref.shade(cpu["f_vn"], cpu["f_vc"], cpu["postex"], cpu["tn"], cpu["albedo"], cpu["light_sh"], cpu["campos"],
                     envmips=cpu["mips"], lightrot=cpu["lightrot"])
(pr["color"] * up).sum().backward()
ref = {k: cpu[k].grad.clone() for k in leaves}
g = {k: (v.detach().cuda().requires_grad_(v.requires_grad) if torch.is_tensor(v) else [m.cuda() for m in v]) for k, v in cpu.items()}
ph = shade.shading_tail(g["f_vn"], g["f_vc"], g["postex"], g["tn"], g["albedo"], g["light_sh"], g["campos"],
                        preconv_envmap=g["mips"], lightrot=g["lightrot"])
(ph["color"] * up.cuda()).sum().backward()
print("color rel", float((ph["color"].cpu() - pr["color"]).norm() / pr["color"].norm()))
with torch.no_grad():
    view = F.normalize(pr["primpos"] - cpu["campos"][:, None], dim=-1)
    n = pr["spec_nml"]
    refl = view - 2 * (view * n).sum(-1, keepdim=True) * n
    r = torch.einsum("bxy,bny->bnx", cpu["lightrot"], refl)[0]
    uv = shade_ref.dir2uv(r)
    level = (pr["sigma"][0] * 5).clamp(0, 3 - 1e-6)
for k in ("tn", "f_vc", "postex", "f_vn"):
    a, b = g[k].grad.cpu().reshape(1, -1, 10_000), ref[k].reshape(1, -1, 10_000)
    err = (a - b).pow(2).sum(1)[0]
    tot = b.pow(2).sum()
    print(k, "rel-L2", float((err.sum() / tot).sqrt()), " share of the error in the 10 worst Gaussians:",
          float(err.topk(10).values.sum() / err.sum()))
    if k == "tn":
        for i in err.topk(8).indices.tolist():
            lvl = float(level[i]); l0 = int(lvl)
            w, h = 1024 >> l0, 512 >> l0
            ix, iy = ((uv[i, 0] + 1) * w - 1) / 2, ((uv[i, 1] + 1) * h - 1) / 2
            print(f"  gaussian {i}: |grad| ref {float(b[0, :, i].norm()):.3e} hip {float(a[0, :, i].norm()):.3e}  r.y {float(r[i, 1]):+.6f}"
                  f"  uv ({float(uv[i, 0]):+.4f}, {float(uv[i, 1]):+.4f}) level {lvl:.3f} texel frac ({float(ix - math.floor(ix)):.4f}, {float(iy - math.floor(iy)):.4f})"
                  f"  median |grad| {float(b[0].norm(dim=0).median()):.3e}")
