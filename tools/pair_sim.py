"""Diagnostic (GPU): how many raster visits would PAIRING save?  (round 6 prototype estimate.)
Today a wave (16x8 half of a tile, 2 px per lane) visits every list entry whose alpha >= 1/255 region reaches its half.  Split
the wave's footprint into two 32-lane halves; an entry that reaches only ONE of them could share its visit with an entry that
reaches only the OTHER (they commute: neither touches the other's pixels), the 32-lane reductions cost what the 64-lane one
does.  Between two consecutive entries that reach both halves, a runs of `a` first-half-only and `b` second-half-only entries
cost max(a, b) visits instead of a + b.  Splits tried: strips (two 16x4) and squares (two 8x8).
Usage: python tools/pair_sim.py [n_tiles] [slab] [scale_shift]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from goliath_amd import shade, splat

n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 400
slab = int(sys.argv[2]) if len(sys.argv) > 2 else 500
shift = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cfg = dict(bench.CFG, views_per_gpu=1, slab=slab, gaussians=slab * slab)
t = bench.make_inputs(cfg, "cuda")
H, W = cfg["height"], cfg["width"]
with torch.no_grad():
    t["f_vn"][:, 113 + 7:113 + 10] += shift
    p = shade.shading_tail(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                           preconv_envmap=t["mips"], lightrot=t["lightrot"])
    intr = torch.stack([t["K"][:, 0, 0], t["K"][:, 1, 1], t["K"][:, 0, 2], t["K"][:, 1, 2]], -1)
    out = splat.render_views(p["primpos"], p["primscale"], p["primqvec"], p["opacity"], p["color"], t["Rt"], intr, H, W)
    xys, depths, radii, conics, comp, nth, cov = splat.project_gaussians(p["primpos"][0], p["primscale"][0], 1.0, p["primqvec"][0],
                                                                          t["Rt"][0], float(intr[0, 0]), float(intr[0, 1]),
                                                                          float(intr[0, 2]), float(intr[0, 3]), H, W, 16, 0.1)
    op = p["opacity"][0, :, 0] * comp
    bins, ids, fidx = out["tile_bins"][0], out["sorted_ids"][0], out["final_idx"][0]
    tiles_x = (W + 15) // 16
    nz = torch.nonzero((bins[:, 1] - bins[:, 0]) > 0).flatten()
    g = torch.Generator(device="cpu").manual_seed(0)
    pick = nz[torch.randperm(nz.numel(), generator=g)[:n_tiles].to(nz.device)]
    cur = 0
    # independent sub-wave walks: every sub-footprint walks ITS OWN sequence of reaching entries; a visit serves one entry per
    # sub-footprint, so a batch of 64 list entries costs max over the sub-footprints of their reaching counts
    splits = {"2 strips 16x4": lambda r: [r[:, :4], r[:, 4:]], "2 squares 8x8": lambda r: [r[:, :, :8], r[:, :, 8:]],
              "4 rects 8x4": lambda r: [r[:, :4, :8], r[:, :4, 8:], r[:, 4:, :8], r[:, 4:, 8:]],
              "4 strips 16x2": lambda r: [r[:, 0:2], r[:, 2:4], r[:, 4:6], r[:, 6:8]],
              "4 columns 4x8": lambda r: [r[:, :, 0:4], r[:, :, 4:8], r[:, :, 8:12], r[:, :, 12:16]]}
    new = {k: 0 for k in splits}
    ideal = {k: 0 for k in splits}
    for tile in pick.tolist():
        ty, tx = divmod(tile, tiles_x)
        lo, hi = int(bins[tile, 0]), int(bins[tile, 1])
        e = ids[lo:hi].long()
        li = torch.arange(lo, hi, device="cuda")
        yy, xx = torch.meshgrid(ty * 16 + torch.arange(16, device="cuda"), tx * 16 + torch.arange(16, device="cuda"), indexing="ij")
        inside = (yy < H) & (xx < W)
        f = fidx[yy.clamp(max=H - 1), xx.clamp(max=W - 1)]
        dx = xys[e, 0][:, None, None] - (xx[None] + 0.5)
        dy = xys[e, 1][:, None, None] - (yy[None] + 0.5)
        sg = 0.5 * (conics[e, 0][:, None, None] * dx * dx + conics[e, 2][:, None, None] * dy * dy) + conics[e, 1][:, None, None] * dx * dy
        alpha = op[e][:, None, None] * torch.exp(-sg)
        bmax = int(f.max())
        reach = (sg >= 0) & (alpha >= 1.0 / 255.0) & inside[None] & (li[:, None, None] <= bmax)
        # the kernel's batches: 64 consecutive list entries counted back from the tile's last walked entry
        batch = ((bmax - li).clamp(min=0) // 64)
        nbatch = int(batch.max()) + 1 if batch.numel() else 1
        for wave in range(2):
            r = reach[:, 8 * wave:8 * wave + 8]                                      # [E, 8, 16]
            cur += int(r.flatten(1).any(1).sum())
            for name, fn in splits.items():
                cnt = torch.stack([torch.bincount(batch[h.flatten(1).any(1)], minlength=nbatch) for h in fn(r)])   # [parts, nbatch]
                new[name] += int(cnt.max(0).values.sum())
                ideal[name] += float(cnt.float().mean(0).sum())
print(f"N={slab * slab} tiles sampled {len(pick)}: visits today {cur}")
for name in new:
    print(f"  {name:16s}: visits {new[name]} = {new[name] / cur:.3f} of today's   (perfectly balanced: {ideal[name] / cur:.3f})")
