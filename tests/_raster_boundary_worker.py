"""Worker of tests/test_gpu_exact_math.py: HIP bin/sort + raster forward + raster backward against the CPU oracle AT THE
RASTER BOUNDARY, i.e. both sides are fed the SAME projected attributes (computed once by the oracle on the CPU), so every
difference is the rasterizer's.  Runs against whichever build of the library GOLIATH_HIP_LIB selects (default: the
product build) and prints one JSON line.  Production configuration of the kernels: planar images, colour + depth in one
pass, L1 fused into the epilogue, 64-byte gradient records.

usage: python _raster_boundary_worker.py VIEWS"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


PPL = int(os.environ.get("RASTER_BOUNDARY_PPL", "2"))   # wave footprint of the raster kernels (include/goliath_hip.h)


def main(n_views):
    import bench
    from goliath_amd import _lib, splat
    from goliath_amd._lib import c_float, c_i64, c_int, fptr, iptr, ptr, stream_ptr
    from oracle import cref, shade_ref

    cfg = dict(bench.CFG, views_per_gpu=1)
    H, W, N = cfg["height"], cfg["width"], cfg["gaussians"]
    T = splat._tiles(H, W)
    cref.set_threads(min(32, os.cpu_count() or 1))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    dev = torch.device("cuda")
    rep = {"lib": os.path.basename(_lib.LIB_PATH), "pixels_per_lane": PPL, "views": n_views, "gaussians": N, "image": [H, W]}
    acc = {k: [0.0, 0.0] for k in ("v_xy", "v_conic", "v_colors", "v_opacity", "v_depth")}   # [err^2, ref^2]
    out_err = {"rgb": 0.0, "alpha": 0.0, "depth_norm": 0.0}
    flips = n_isect_hip = n_isect_orc = 0
    for v in range(n_views):
        with torch.no_grad():
            t = bench.make_inputs(cfg, "cpu", rank=v)
            pr = shade_ref.shade(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                                 envmips=t["mips"], lightrot=t["lightrot"])
            K, vm = t["K"][0], t["Rt"][0]
            xys, depths, radii, conics, comp, nth, _ = cref.project_gaussians(
                pr["primpos"][0], pr["primscale"][0], 1.0, pr["primqvec"][0], vm, float(K[0, 0]), float(K[1, 1]),
                float(K[0, 2]), float(K[1, 2]), H, W, 16, 0.1)
            opac = (pr["opacity"][0, :, 0] * comp).contiguous()
            colors = pr["color"][0].contiguous()
            target = t["target"][0]
            scale = 1.0 / (3 * H * W)
            # ---- oracle
            _, ids, bins = cref.bin_and_sort(xys, depths, radii, nth, H, W, 16)
            col4 = torch.cat([colors, depths[:, None]], 1).contiguous()
            bg4 = torch.zeros(4)
            img, Ts, idx = cref.rasterize_forward(ids, bins, xys, conics, col4, opac, H, W, 16, bg4)
            diff = img[..., :3] - target.permute(1, 2, 0)
            v_out = torch.zeros(H, W, 4)
            v_out[..., :3] = torch.sign(diff) * scale
            o_xy, o_conic, o_col, o_op = cref.rasterize_backward(ids, bins, xys, conics, col4, opac, H, W, 16, bg4, Ts,
                                                                 idx, v_out)
            last_o = torch.where(Ts < 1.0, ids[idx.long()], torch.full_like(idx, -1))
            n_isect_orc += int(ids.numel())
            # ---- HIP, same inputs
            g = lambda x: x.contiguous().to(dev)
            d_xy, d_dep, d_rad, d_con, d_op, d_col, d_tgt = map(g, (xys[None], depths[None], radii[None], conics[None],
                                                                    opac[None], colors[None], target[None]))
            cap = int(nth.sum()) + 1024
            ws = splat._Workspace(1, N, T, cap, dev)
            splat._bin_sort(1, N, d_xy, d_dep, d_rad, H, W, ws, d_con, d_op)
            bgd = torch.zeros(3, device=dev)
            out_img = torch.empty(1, 3, H, W, device=dev)
            fT = torch.empty(1, H, W, device=dev)
            fidx = torch.empty(1, H, W, dtype=torch.int32, device=dev)
            alpha = torch.empty(1, H, W, device=dev)
            dnorm = torch.empty(1, H, W, device=dev)
            sign = torch.empty(1, H, W, dtype=torch.uint8, device=dev)
            part = torch.empty(1, T, device=dev)
            records = splat._pack_records(1, N, d_xy, d_con, d_col, d_dep, d_op)
            _lib.call("gol_rasterize_fwd", c_int(1), c_int(N), c_int(H), c_int(W), c_int(16), c_int(1),
                      iptr(ws.tile_bins), iptr(ws.sorted_ids), c_i64(ws.capacity), fptr(records), c_int(1),
                      fptr(bgd), fptr(out_img), fptr(None), fptr(fT), iptr(fidx), fptr(alpha),
                      fptr(dnorm), c_float(0.05), fptr(d_tgt), fptr(None), c_int(0), ptr(sign, torch.uint8), fptr(part),
                      fptr(None), c_float(1.0), c_int(PPL), stream_ptr())
            rec = torch.zeros(1, N, 16, device=dev)
            field = lambda k: ctypes.c_void_p(rec.data_ptr() + 4 * k)
            vsc = torch.full((1,), scale, device=dev)
            _lib.call("gol_rasterize_bwd", c_int(1), c_int(N), c_int(H), c_int(W), c_int(16), c_int(1),
                      iptr(ws.tile_bins), iptr(ws.sorted_ids), c_i64(ws.capacity), fptr(records), c_int(0),
                      fptr(bgd), fptr(fT), iptr(fidx), fptr(None), fptr(None), fptr(None),
                      field(4), field(6), field(0), fptr(None), field(3), c_int(16), ptr(sign, torch.uint8), fptr(None),
                      c_int(0), fptr(vsc), c_float(1.0), c_int(PPL), stream_ptr())
            torch.cuda.synchronize()
            n_isect_hip += int(ws.n_isect[0])
            Th = fT[0].cpu()
            # a different composited list shows in the transmittance: an entry at the alpha = 1/255 cut moves T by >= 0.39 %
            flip = (Th - Ts).abs() > 1e-3 * Ts.clamp(min=1e-4)
            flips += int(flip.sum())
            out_err["rgb"] = max(out_err["rgb"], rel(out_img[0], img[..., :3].permute(2, 0, 1)))
            out_err["alpha"] = max(out_err["alpha"], rel(alpha[0], 1.0 - Ts))
            out_err["depth_norm"] = max(out_err["depth_norm"], rel(dnorm[0], img[..., 3] / (1.0 - Ts).clamp(0.05, 1.0)))
            r = rec[0].cpu()
            for k, a, b in (("v_colors", r[:, 0:3], o_col[:, :3]), ("v_opacity", r[:, 3:4], o_op),
                            ("v_xy", r[:, 4:6], o_xy), ("v_conic", r[:, 6:9], o_conic)):
                acc[k][0] += float((a.double() - b.double()).pow(2).sum())
                acc[k][1] += float(b.double().pow(2).sum())
    del acc["v_depth"]
    rep["intersections_per_view"] = {"hip_pruned": n_isect_hip / n_views, "oracle_3sigma": n_isect_orc / n_views}
    rep["flip_pixel_fraction"] = flips / (n_views * H * W)
    rep["flip_pixels"] = flips
    rep["outputs_rel_l2"] = out_err
    rep["raster_boundary_grads_rel_l2_all_gaussians"] = {k: (e / max(r, 1e-300)) ** 0.5 for k, (e, r) in acc.items()}
    print("RASTER_BOUNDARY " + json.dumps(rep), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
