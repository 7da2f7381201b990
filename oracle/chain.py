"""oracle/chain.py -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).

The whole per-view hot path of BASELINE config 2 through the CPU oracles, forward + backward, i.e. the CPU
counterpart of one view of bench.py's step (shade -> project -> bin/sort -> colour+depth raster -> L1 ->
raster / project / shade backward):
    shading tail      oracle/shade_ref.py     (rgca.py:505-588, envmap.py:284-292, mipmap_sampler.py:13-69; PINNED)
    project/bin/raster oracle/gsplat_oracle.c (gsplat 0.1.11 as restated in SURVEY.md Appendix A; PARITY UNPINNED)
    render epilogue   rgca.py:137,144-145     (alpha = 1 - T.detach(), depth / alpha.clamp(0.05, 1))
    loss              ca_code/loss/__init__.py:411  ((rgb - target).abs().mean())
Used by tests/test_gpu_fullsize.py (the checker of the benchmarked chain) and by bench.py's cpu_baseline leg.
"""
import torch

from . import cref, shade_ref


def cpu_view(t, height, width, loss_scale=None, backward=True):
    """t: a ONE-view input dict of bench.make_inputs (CPU tensors; leaves f_vn, f_vc, postex, tn, albedo with
    requires_grad).  loss_scale = 1 / (number of elements the L1 mean runs over); default: this view alone.
    Returns dict(rgb[3,H,W], alpha[H,W], depth_norm[H,W], final_T[H,W], last_id[H,W] (Gaussian id of the last
    contributor, -1 where none), loss, n_isect) and, when `backward`, leaves gradients in t[k].grad."""
    H, W = height, width
    preds = shade_ref.shade(t["f_vn"], t["f_vc"], t["postex"], t["tn"], t["albedo"], t["light_sh"], t["campos"],
                            envmips=t["mips"], lightrot=t["lightrot"])
    means, scales, quats = preds["primpos"][0].detach(), preds["primscale"][0].detach(), preds["primqvec"][0].detach()
    K, vm = t["K"][0], t["Rt"][0]
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    xys, depths, radii, conics, comp, nth, cov3d = cref.project_gaussians(means, scales, 1.0, quats, vm, fx, fy, cx,
                                                                           cy, H, W, 16, 0.1)
    _, ids, bins = cref.bin_and_sort(xys, depths, radii, nth, H, W, 16)
    opac = (preds["opacity"][0, :, 0].detach() * comp).contiguous()
    col4 = torch.cat([preds["color"][0].detach(), depths[:, None]], 1).contiguous()  # colour + depth in one pass
    bg = torch.zeros(4)
    img, Ts, idx = cref.rasterize_forward(ids, bins, xys, conics, col4, opac, H, W, 16, bg)
    alpha = 1.0 - Ts
    rgb = img[..., :3].permute(2, 0, 1)
    if loss_scale is None:
        loss_scale = 1.0 / (3 * H * W)
    diff = img[..., :3] - t["target"][0].permute(1, 2, 0)
    out = dict(rgb=rgb, alpha=alpha, depth_norm=img[..., 3] / alpha.clamp(0.05, 1.0), final_T=Ts,
               last_id=torch.where(Ts < 1.0, ids[idx.long()] if ids.numel() else idx, torch.full_like(idx, -1)),
               loss=float(diff.abs().sum()) * loss_scale, n_isect=int(ids.numel()),
               # diagnostics for the parity tests' "explained by" predicates: the screen-space footprints and the
               # shading state of this view
               xys=xys, radii=radii, depths=depths, conics=conics, opac_eff=opac, preds={k: v.detach() for k, v in preds.items()})
    if not backward:
        return out
    v_out = torch.zeros(H, W, 4)
    v_out[..., :3] = torch.sign(diff) * loss_scale
    v_xy, v_conic, v_col, v_op = cref.rasterize_backward(ids, bins, xys, conics, col4, opac, H, W, 16, bg, Ts, idx, v_out)
    v_comp = v_op[:, 0] * preds["opacity"][0, :, 0].detach()
    _, _, v_mean, v_scale, v_quat = cref.project_gaussians_backward(means, scales, 1.0, quats, vm, fx, fy, cov3d,
                                                                    radii, conics, comp, v_xy, v_col[:, 3].contiguous(),
                                                                    v_conic, v_comp)
    # chain into the shading tail through torch autograd (vector-Jacobian product of the five raster inputs)
    (preds["primpos"][0] * v_mean).sum().add((preds["primscale"][0] * v_scale).sum()).add(
        (preds["primqvec"][0] * v_quat).sum()).add((preds["color"][0] * v_col[:, :3]).sum()).add(
        (preds["opacity"][0, :, 0] * v_op[:, 0] * comp).sum()).backward()
    # the gradients at the raster / projection boundary (what the shading tail receives), for stage-wise reports
    out["stage_grads"] = dict(color=v_col[:, :3], opacity=(v_op[:, 0] * comp)[:, None], primpos=v_mean,
                              primscale=v_scale, primqvec=v_quat)
    return out


def shade_leaf_grads(t, stage_grads, dtype=torch.float64):
    """Leaf gradients of ONE view's shading tail for given gradients at the raster / projection boundary (the `stage_grads`
    of cpu_view), with the whole tail evaluated in `dtype`.  In fp64 this is the yardstick for the fp32 oracle's own
    rounding: where the fp32 evaluation of the reference's formulation is itself far from the fp64 one (a lookup at a
    derivative jump, the acos pole), no fp32 implementation can be expected to reproduce it to 1e-4."""
    leaves = ("f_vn", "f_vc", "postex", "tn", "albedo")
    tt = {k: (v.detach().to(dtype).requires_grad_(k in leaves) if torch.is_tensor(v) and v.is_floating_point() else v)
          for k, v in t.items() if k != "mips"}
    mips = [m.to(dtype) for m in t["mips"]]
    pr = shade_ref.shade(tt["f_vn"], tt["f_vc"], tt["postex"], tt["tn"], tt["albedo"], tt["light_sh"], tt["campos"],
                         envmips=mips, lightrot=tt["lightrot"])
    g = {k: v.to(dtype) for k, v in stage_grads.items()}
    (pr["primpos"][0] * g["primpos"]).sum().add((pr["primscale"][0] * g["primscale"]).sum()).add(
        (pr["primqvec"][0] * g["primqvec"]).sum()).add((pr["color"][0] * g["color"]).sum()).add(
        (pr["opacity"][0] * g["opacity"]).sum()).backward()
    return {k: tt[k].grad for k in leaves}
