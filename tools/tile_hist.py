#!/usr/bin/env python
"""Histogram of per-tile list lengths (what the per-tile sort and the rasterizer iterate over) for the bench scene."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from goliath_amd import render_gs, shade, splat

seen = []
orig = splat._Workspace.__init__


def rec(self, *a, **k):
    orig(self, *a, **k)
    seen.append(self)


splat._Workspace.__init__ = rec
cfg = dict(bench.CFG, views_per_gpu=2)
t = bench.make_inputs(cfg, torch.device("cuda"))
with torch.no_grad():
    preds = shade.shading_tail(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                               preconv_envmap=t["mips"], lightrot=t["lightrot"])
    render_gs.render_batch(t["K"], t["Rt"], preds, cfg["height"], cfg["width"])
torch.cuda.synchronize()
bins = seen[-1].tile_bins
n = (bins[..., 1] - bins[..., 0]).flatten().float()
edges = [0, 1, 33, 65, 129, 257, 513, 1025, 2049, 4097, 10 ** 9]
tot = float(n.sum())
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (n >= lo) & (n < hi)
    print(f"len [{lo:5d},{hi:10d}): {int(m.sum()):7d} tiles  {100 * float(n[m].sum()) / tot:5.1f} % of entries  "
          f"n log2^2 share {100 * float((n[m] * torch.log2(n[m].clamp(min=2)) ** 2).sum()) / float((n * torch.log2(n.clamp(min=2)) ** 2).sum()):5.1f} %")
print("tiles", n.numel(), "entries", int(tot), "max", int(n.max()))
