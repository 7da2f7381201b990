#!/usr/bin/env python
"""Stand-alone timing of gol_bin_sort on the projected Gaussians of the bench scene (run ON THE GPU BOX):
    python tools/bin_probe.py [--slab 500|1024] [--views 8] [--scale-shift 0.0] [--iters 30]
--slab 1024 = 1,048,576 Gaussians (the reference-native size, rgca.py:385-386); --scale-shift s adds s to the softplus^-1
scale channels (the e2e fit's Gaussians are smaller than SURVEY 8d's: -0.9 gives ~4.5 M stored entries per view at 1 M).
Prints entries per view, the list-length distribution and the call's time; with rocprofv3 --kernel-trace --stats around it
(tools/bin_probe.sh) the per-kernel split.  GOLIATH_HIP_LIB selects the build."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from goliath_amd import render_gs, shade, splat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slab", type=int, default=500)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--scale-shift", type=float, default=0.0)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--clustering", action="store_true")
    a = ap.parse_args()
    cfg = dict(bench.CFG, slab=a.slab, gaussians=a.slab * a.slab, views_per_gpu=a.views)
    dev = torch.device("cuda")
    t = bench.make_inputs(cfg, dev)
    with torch.no_grad():
        t["f_vn"][:, 113 + 7:113 + 10] += a.scale_shift
        vs = render_gs.view_set(t["K"], t["Rt"], cfg["height"], cfg["width"])
        preds = shade.shading_tail(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                                   preconv_envmap=t["mips"], lightrot=t["lightrot"], views=vs)
        pr = preds["projected"]
        B, N = pr.records.shape[:2]
        H, W = cfg["height"], cfg["width"]
        T = ((H + 15) // 16) * ((W + 15) // 16)
        xys, depths, radii, conics, opac = (pr.field(k) for k in ("xys", "depths", "radii", "conics", "opac_eff"))
        cap = 12 * N if a.slab <= 500 else 10 * N
        ws = splat._Workspace(B, N, T, cap, dev)

        def call():
            ws.tile_count.zero_()
            splat._bin_sort(B, N, xys, depths, radii, H, W, ws, conics, opac)

        for _ in range(3):
            call()
        torch.cuda.synchronize()
        n = (ws.tile_bins[..., 1] - ws.tile_bins[..., 0]).float()
        assert int(ws.n_isect.max()) <= cap, "capacity too small for this scene"
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
        ev[0].record()
        for i in range(a.iters):
            call()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
        if a.clustering:
            # how evenly does the sort's linear depth -> bucket map spread a tile's keys?  sum b_i^2 / n = compares per key of the
            # in-bucket ranking (2 for a Poisson spread at one key per bucket)
            bins = ws.tile_bins[0].cpu()
            keys = ws.keys[0].cpu()
            nz = torch.nonzero((bins[:, 1] - bins[:, 0]) > 64).flatten()
            g = torch.Generator().manual_seed(0)
            rows = []
            for ti in nz[torch.randperm(nz.numel(), generator=g)[:200]].tolist():
                k = keys[bins[ti, 0]:bins[ti, 1]]
                d = (k >> 32).to(torch.int32).view(torch.float32)
                n_k = d.numel()
                for nb_cap in (1024, 4096):
                    nb = min(max(n_k, 32), nb_cap)
                    lo, hi = d.min(), d.max()
                    bk = ((d - lo) * (nb / (hi - lo))).long().clamp(0, nb - 1) if hi > lo else torch.zeros(n_k, dtype=torch.long)
                    h = torch.bincount(bk, minlength=nb).float()
                    rows.append((n_k, nb_cap, float((h * h).sum() / n_k), int(h.max())))
            for cap_ in (1024, 4096):
                r = [x for x in rows if x[1] == cap_]
                print(f"   clustering (bucket cap {cap_}): {len(r)} lists, n mean {sum(x[0] for x in r) / len(r):.0f}; compares per key "
                      f"mean {sum(x[2] for x in r) / len(r):.2f} max {max(x[2] for x in r):.2f}; fullest bucket mean "
                      f"{sum(x[3] for x in r) / len(r):.1f} max {max(x[3] for x in r)}")
        # checksum of the sorted lists (order-sensitive): two builds must print the same number
        ids = ws.sorted_ids.long()
        pos = torch.arange(ids.shape[1], device=dev)[None]
        valid = pos < ws.tile_bins[..., 1].max(dim=1, keepdim=True).values
        chk = int(((ids * (pos % 1000003 + 1)) * valid).sum() % 2147483647)
        print(f"bin_probe N={N} views={B} entries/view={float(n.sum(1).mean()):.0f} reserved/view={float(ws.n_isect.float().mean()):.0f} "
              f"lists>2048/view={float((n > 2048).float().sum(1).mean()):.0f} >4096={float((n > 4096).float().sum(1).mean()):.0f} "
              f">8192={float((n > 8192).float().sum(1).mean()):.0f} max={int(n.max())} | gol_bin_sort median {ms[len(ms) // 2]:.4f} ms "
              f"min {ms[0]:.4f} | checksum {chk}")


if __name__ == "__main__":
    main()
