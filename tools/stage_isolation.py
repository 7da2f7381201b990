#!/usr/bin/env python
"""Diagnostic (GPU): where do the HIP-vs-oracle gradient residuals of the config-2 chain come from?  Every stage is run
on IDENTICAL inputs (the oracle's), so a stage's number is that stage's own rounding, not inherited differences.
  P   project forward     same means / scales / quats           -> xys, conics, compensation, depths, radii
  PB  project backward    same forward state + same upstream     -> v_mean, v_scale, v_quat
(the raster stage is tests/_raster_boundary_worker.py).  Prints how concentrated each error is."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from goliath_amd import splat  # noqa: E402
from oracle import cref, shade_ref  # noqa: E402


def conc(a, b):
    a, b = a.double().cpu().reshape(a.shape[0], -1), b.double().cpu().reshape(b.shape[0], -1)
    e = (a - b).pow(2).sum(1)
    tot = b.pow(2).sum()
    s = e.sort(descending=True).values
    return {"rel_l2": float((e.sum() / tot).sqrt()), "share_top10": float(s[:10].sum() / e.sum()),
            "share_top1000": float(s[:1000].sum() / e.sum()),
            "rel_l2_without_top1000": float((s[1000:].sum() / tot).sqrt())}


cfg = dict(bench.CFG, views_per_gpu=1)
H, W, N = cfg["height"], cfg["width"], cfg["gaussians"]
cref.set_threads(32)
with torch.no_grad():
    t = bench.make_inputs(cfg, "cpu", rank=0)
    pr = shade_ref.shade(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                         envmips=t["mips"], lightrot=t["lightrot"])
K, vm = t["K"][0], t["Rt"][0]
fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
means, scales, quats = pr["primpos"][0].contiguous(), pr["primscale"][0].contiguous(), pr["primqvec"][0].contiguous()
xys, depths, radii, conics, comp, nth, cov3d = cref.project_gaussians(means, scales, 1.0, quats, vm, fx, fy, cx, cy, H, W, 16, 0.1)
dm, ds, dq = (x.cuda().requires_grad_(True) for x in (means, scales, quats))
hx, hd, hr, hc, hcomp, hn, hcov = splat.project_gaussians(dm, ds, 1.0, dq, vm.cuda(), fx, fy, cx, cy, H, W, 16, 0.1)
vis = radii > 0
rep = {"P": {"xys": conc(hx[vis.cuda()], xys[vis]), "conics": conc(hc[vis.cuda()], conics[vis]),
             "comp": conc(hcomp[vis.cuda()][:, None], comp[vis][:, None]), "depths": conc(hd[vis.cuda()][:, None], depths[vis][:, None]),
             "radii_mismatch": int((hr.cpu() != radii).sum()), "num_tiles_hit_mismatch": int((hn.cpu() != nth).sum())}}
# conic relative error per Gaussian
ce = ((hc.cpu() - conics).norm(dim=1) / conics.norm(dim=1).clamp(min=1e-30))[vis]
rep["P"]["conic_rel_err_quantiles"] = {q: float(ce.quantile(q)) for q in (0.5, 0.99, 0.9999)}
rep["P"]["conic_rel_err_max"] = float(ce.max())
g = torch.Generator().manual_seed(3)
v_xy, v_conic, v_depth, v_comp = (torch.randn(N, 2, generator=g), torch.randn(N, 3, generator=g) * 10.0,
                                  torch.randn(N, generator=g) * 0.01, torch.randn(N, generator=g))
_, _, o_mean, o_scale, o_quat = cref.project_gaussians_backward(means, scales, 1.0, quats, vm, fx, fy, cov3d, radii, conics,
                                                                comp, v_xy, v_depth, v_conic, v_comp)
torch.autograd.backward([hx, hd, hc, hcomp], [v_xy.cuda(), v_depth.cuda(), v_conic.cuda(), v_comp.cuda()])
rep["PB"] = {"v_mean": conc(dm.grad, o_mean), "v_scale": conc(ds.grad, o_scale), "v_quat": conc(dq.grad, o_quat)}
print("STAGE_ISOLATION " + json.dumps(rep, indent=1))
