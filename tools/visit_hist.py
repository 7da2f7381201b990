"""Diagnostic (GPU): what does a raster-backward VISIT look like at config 2?  For a sample of non-empty tiles: per (entry,
16x8 half) visit the number of the wave's 64 lanes (2 vertically adjacent pixels each) with at least one TAKEN pixel
(alpha >= 1/255, entry within the pixel's final_idx), and the taken pixels of the 128.  Prints histograms over visits.
Usage: python tools/visit_hist.py [n_tiles]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from goliath_amd import shade, splat

n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cfg = dict(bench.CFG, views_per_gpu=1)
t = bench.make_inputs(cfg, "cuda")
H, W = cfg["height"], cfg["width"]
with torch.no_grad():
    p = shade.shading_tail(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                           preconv_envmap=t["mips"], lightrot=t["lightrot"])
    intr = torch.stack([t["K"][:, 0, 0], t["K"][:, 1, 1], t["K"][:, 0, 2], t["K"][:, 1, 2]], -1)
    out = splat.render_views(p["primpos"], p["primscale"], p["primqvec"], p["opacity"], p["color"], t["Rt"], intr, H, W)
    # screen-space attributes through the gsplat-compatible projection
    xys, depths, radii, conics, comp, nth, cov = splat.project_gaussians(p["primpos"][0], p["primscale"][0], 1.0, p["primqvec"][0],
                                                                          t["Rt"][0], float(intr[0, 0]), float(intr[0, 1]),
                                                                          float(intr[0, 2]), float(intr[0, 3]), H, W, 16, 0.1)
    op = p["opacity"][0, :, 0] * comp
    bins, ids, fidx = out["tile_bins"][0], out["sorted_ids"][0], out["final_idx"][0]
    tiles_x = (W + 15) // 16
    lens = (bins[:, 1] - bins[:, 0])
    nz = torch.nonzero(lens > 0).flatten()
    g = torch.Generator(device="cpu").manual_seed(0)
    pick = nz[torch.randperm(nz.numel(), generator=g)[:n_tiles].to(nz.device)]
    lane_hist = torch.zeros(65, dtype=torch.int64, device="cuda")
    pix_hist = torch.zeros(129, dtype=torch.int64, device="cuda")
    visits = taker_less = both = one = 0
    for tile in pick.tolist():
        ty, tx = divmod(tile, tiles_x)
        lo, hi = int(bins[tile, 0]), int(bins[tile, 1])
        e = ids[lo:hi].long()
        li = torch.arange(lo, hi, device="cuda")
        py = ty * 16 + torch.arange(16, device="cuda")
        px = tx * 16 + torch.arange(16, device="cuda")
        yy, xx = torch.meshgrid(py, px, indexing="ij")                       # [16,16]
        inside = (yy < H) & (xx < W)
        f = fidx[yy.clamp(max=H - 1), xx.clamp(max=W - 1)]                   # last index each pixel walks
        dx = xys[e, 0][:, None, None] - (xx[None] + 0.5)
        dy = xys[e, 1][:, None, None] - (yy[None] + 0.5)
        sg = 0.5 * (conics[e, 0][:, None, None] * dx * dx + conics[e, 2][:, None, None] * dy * dy) + conics[e, 1][:, None, None] * dx * dy
        alpha = torch.clamp(op[e][:, None, None] * torch.exp(-sg), max=0.99)
        taken = (sg >= 0) & (alpha >= 1.0 / 255.0) & (li[:, None, None] <= f[None]) & inside[None]     # [E,16,16]
        any_half = [(taken[:, 8 * h:8 * h + 8].flatten(1).any(1)) for h in range(2)]
        both += int((any_half[0] & any_half[1]).sum())
        one += int((any_half[0] ^ any_half[1]).sum())
        for half in range(2):
            th = taken[:, 8 * half:8 * half + 8]                             # [E,8,16]
            lanes = (th[:, 0::2] | th[:, 1::2]).flatten(1).sum(1)            # [E] lanes with a taker (4 row pairs x 16)
            pix = th.flatten(1).sum(1)
            # a visit happens when the entry can reach the half; approximate by "some pixel of the half within the walk"
            reach = (li <= f[8 * half:8 * half + 8].max()).bool()
            lanes, pix = lanes[reach], pix[reach]
            lane_hist += torch.bincount(lanes, minlength=65)
            pix_hist += torch.bincount(pix, minlength=129)
            visits += int(reach.sum())
            taker_less += int((lanes == 0).sum())
lh = lane_hist.cpu().double()
print(f"tiles sampled {len(pick)}, entry-half pairs within the walk {visits} (no exact-reach culling applied: the kernel skips most taker-less ones before the visit)")
tot = lh[1:].sum()
print("visits with >= 1 taker lane:", int(tot), " taker-less pairs:", int(lh[0]))
cum = 0.0
for a, b in ((1, 4), (5, 8), (9, 16), (17, 24), (25, 32), (33, 48), (49, 63), (64, 64)):
    fr = lh[a:b + 1].sum() / tot
    cum += fr
    print(f"  taker lanes {a:2d}-{b:2d}: {fr:6.3f}   cumulative {cum:6.3f}")
ph = pix_hist.cpu().double()
print("mean taker lanes per visit (given >= 1):", float((lh[1:] * torch.arange(1, 65)).sum() / tot),
      " mean taken pixels:", float((ph[1:] * torch.arange(1, 129)).sum() / ph[1:].sum()))
print(f"tile entries with takers in BOTH 16x8 halves: {both}, in exactly one: {one}  ->  both / (both + one) = {both / max(both + one, 1):.3f}")
print("instruction model (vector instructions per tile entry; visit = 75 at 2 px / lane, ~110 at 4 px / lane over the whole tile):",
      f"2 px/lane {(150 * both + 75 * one) / max(both + one, 1):.1f}   4 px/lane 110.0")
