"""GPU: parity of the BENCHMARKED chain at BASELINE config-2 size (VERDICT r1 "next" #1a, SURVEY.md 8c protocol).

The exact step bench.py times -- env-map relight, 250,000 Gaussians, 2048x1334, B = 4 views per launch:
shade(env) -> project -> pruned bin/sort -> planar 2-px/lane colour+depth raster -> fused L1 -> packed-record raster
backward -> project backward -> shade backward -- is compared as a whole with the CPU oracle chain
(oracle/chain.py): rgb / alpha / depth and EVERY input gradient (f_vnocond, f_vcond, postex, tn, albedo) on whole
images (the OpenMP oracle needs ~2 s per view on the GPU box's cores, so no tile subsampling is necessary), plus the
fraction of pixels whose contributing list differs (threshold flips at alpha = 1/255, T = 1e-4, sigma < 0)."""
import json
import os
import sys

import pytest
import torch

from scenes import rel_l2

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOL_OUT = 1e-4       # north_star: outputs within 1e-4 L2 (rgb, alpha)
TOL_DEPTH = 2e-4     # depth / clamp(alpha, 0.05, 1): the division by a clamped alpha amplifies threshold-flip pixels 20x
TOL_ROBUST = 2e-4    # every gradient, with the 1e-4 fraction of Gaussians with the largest error left out (see below)
TOL_ALL = 3e-3       # every gradient, every Gaussian: the ~4e-5 fraction of threshold-flip pixels (a Gaussian at
                     # alpha ~ 1/255 or a pixel at T ~ 1e-4 is composited by one implementation and skipped by the
                     # other -- fast exp vs expf) concentrates O(1) relative errors on the few Gaussians of those pixels
LEAVES = ("f_vn", "f_vc", "postex", "tn", "albedo")
STAGE = ("color", "opacity", "primpos", "primscale", "primqvec")


def _robust_rel_l2(a, b, drop=1e-4):  # noqa: C901
    """rel-L2 over [B, C, N]-shaped per-Gaussian gradients with the `drop` fraction of Gaussians with the largest error
    left out.  Two legitimate sources put O(1) errors on isolated Gaussians: threshold-flip pixels (above), and the
    env-map specular term, which is piecewise linear in the reflection direction (bilinear texel lookups,
    mipmap_sampler.py:13-69) -- its DERIVATIVE jumps at texel borders, so a lookup within rounding of a border gets a
    different gradient on the two implementations.  Everything systematic shows up in this number."""
    a, b = a.double().cpu(), b.double().cpu()
    err = (a - b).pow(2).sum(1)                     # [B, N]
    k = max(1, int(drop * err.numel()))
    thr = err.flatten().kthvalue(err.numel() - k).values
    keep = (err <= thr)[:, None].expand_as(a)
    return float(((a - b)[keep]).norm() / b[keep].norm())


def _gpu_step(mb, cfg):
    from goliath_amd import losses, render_gs, shade, splat

    preds = shade.shading_tail(mb["f_vn"], mb["f_vc"], mb["postex"], mb["tn"], mb["albedo"], mb["light_sh"],
                               mb["campos"], preconv_envmap=mb["mips"], lightrot=mb["lightrot"])
    rgb, alpha, depth, loss = render_gs.render_batch(mb["K"], mb["Rt"], preds, cfg["height"], cfg["width"],
                                                     l1_target=mb["target"])  # the fused L1 of bench.step
    for k in STAGE:
        preds[k].retain_grad()
    loss.backward()
    mb["_stage"] = {k: preds[k].grad for k in STAGE}
    # the lists of the same views (diagnostics only: last contributor per pixel)
    with torch.no_grad():
        intr = torch.stack([mb["K"][:, 0, 0], mb["K"][:, 1, 1], mb["K"][:, 0, 2], mb["K"][:, 1, 2]], -1)
        d = splat.render_views(preds["primpos"], preds["primscale"], preds["primqvec"], preds["opacity"], preds["color"],
                               mb["Rt"], intr, cfg["height"], cfg["width"])
        B = rgb.shape[0]
        last = torch.gather(d["sorted_ids"], 1, d["final_idx"].reshape(B, -1).long()).reshape(d["final_idx"].shape)
        last = torch.where(d["final_T"][:, 0] < 1.0, last, torch.full_like(last, -1))
    return rgb.detach(), alpha, depth.detach(), float(loss), last, d


CONFIGS = {
    # BASELINE config 2: the benchmarked size, B = 4 views per launch (one micro-batch of bench.py)
    "config2": dict(views_per_gpu=4),
    # BASELINE config 1 (plumbing size): 10k Gaussians, one camera, 512x512
    "config1": dict(views_per_gpu=1, slab=100, gaussians=10_000, height=512, width=512, focal=1150.0),
}


@pytest.mark.parametrize("name", ["config2", "config1"])
def test_bench_step_matches_oracle_chain(name):
    import bench
    from oracle import chain, cref

    cfg = dict(bench.CFG, **CONFIGS[name])
    B = cfg["views_per_gpu"]
    H, W = cfg["height"], cfg["width"]
    cpu = bench.make_inputs(cfg, "cpu", rank=0)
    mb = {k: (v.detach().cuda().requires_grad_(v.requires_grad) if torch.is_tensor(v) else [m.cuda() for m in v])
          for k, v in cpu.items()}
    rgb, alpha, depth, loss, last, diag = _gpu_step(mb, cfg)

    cref.set_threads(min(32, os.cpu_count() or 1))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref_grads = {k: [] for k in LEAVES}
    ref_stage = {k: [] for k in STAGE}
    report = {"views": B, "gaussians": cfg["gaussians"], "image": [H, W], "outputs": {}, "grads": {}}
    flips, big, ref_loss = 0, 0, 0.0
    worst = {"rgb": 0.0, "alpha": 0.0, "depth": 0.0}
    for b in range(B):
        one = {k: (v[b:b + 1].detach().clone().requires_grad_(v.requires_grad) if torch.is_tensor(v) and k != "albedo"
                   else v) for k, v in cpu.items()}
        one["albedo"] = cpu["albedo"].detach().clone().requires_grad_(True)
        one["mips"] = [m[b:b + 1] for m in cpu["mips"]]
        o = chain.cpu_view(one, H, W, loss_scale=1.0 / (B * 3 * H * W))
        ref_loss += o["loss"]
        if name == "config2":
            assert o["n_isect"] > 1_000_000  # config-2 density (the pruned HIP lists are ~2 M, the 3-sigma lists ~3.6 M)
        worst["rgb"] = max(worst["rgb"], rel_l2(rgb[b], o["rgb"]))
        worst["alpha"] = max(worst["alpha"], rel_l2(alpha[b, 0], o["alpha"]))
        worst["depth"] = max(worst["depth"], rel_l2(depth[b, 0], o["depth_norm"]))
        # threshold flips: the last contributor differs, or the transmittance differs by more than rounding
        T_h, T_o = diag["final_T"][b, 0].cpu(), o["final_T"]
        flip = (last[b].cpu() != o["last_id"]) | ((T_h - T_o).abs() > 1e-3 * T_o.clamp(min=1e-4))
        flips += int(flip.sum())
        big += int(((rgb[b].cpu() - o["rgb"]).abs().amax(0) > 1e-3).sum())
        for k in LEAVES:
            ref_grads[k].append(one[k].grad)
        for k in STAGE:
            ref_stage[k].append(o["stage_grads"][k])
    report["outputs"] = worst
    report["flip_pixel_fraction"] = flips / (B * H * W)
    report["pixels_off_by_more_than_1e-3"] = big / (B * H * W)
    report["loss"] = {"hip": loss, "oracle": ref_loss}
    drop = 1e-4 if name == "config2" else 1e-3   # 100 of 1 M Gaussians / 10 of 10 k
    report["dropped_fraction"] = drop
    report["stage_grads"], report["stage_grads_without_worst_1e-4_gaussians"] = {}, {}
    for k in STAGE:
        ref = torch.stack(ref_stage[k])                                   # [B, N, C]
        report["stage_grads"][k] = rel_l2(mb["_stage"][k], ref)
        report["stage_grads_without_worst_1e-4_gaussians"][k] = _robust_rel_l2(
            mb["_stage"][k].reshape(B, ref.shape[1], -1).transpose(1, 2), ref.transpose(1, 2), drop)
    report["grads_without_worst_1e-4_gaussians"] = {}
    for k in LEAVES:
        ref = torch.stack(ref_grads[k]).sum(0) if k == "albedo" else torch.cat(ref_grads[k], 0)
        report["grads"][k] = rel_l2(mb[k].grad, ref)
        if k != "albedo":
            S2 = ref.shape[-1] * ref.shape[-2]
            report["grads_without_worst_1e-4_gaussians"][k] = _robust_rel_l2(
                mb[k].grad.reshape(B, -1, S2), ref.reshape(B, -1, S2), drop)
    print(f"\nCHAIN_PARITY {name} " + json.dumps(report))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "fullsize_parity.json" if name == "config2" else f"chain_parity_{name}.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert abs(loss - ref_loss) < 1e-5 * abs(ref_loss)
    for k, v in worst.items():
        assert v < (TOL_DEPTH if k == "depth" else TOL_OUT), (k, v)
    assert report["flip_pixel_fraction"] < 1e-3, report["flip_pixel_fraction"]  # expected << 0.1 %
    if name == "config1":  # 10 k Gaussians: the handful of pole / texel-border lookups weighs 100x more than at 1 M
        for kk, v in report["stage_grads"].items():
            assert v < 1e-4, (kk, v)
        for kk, v in report["grads_without_worst_1e-4_gaussians"].items():
            assert v < 3e-4, (kk, v)
        for kk, v in report["grads"].items():
            assert v < 5e-2, (kk, v)
        return
    for name in ("stage_grads_without_worst_1e-4_gaussians", "grads_without_worst_1e-4_gaussians"):
        for k, v in report[name].items():
            assert v < TOL_ROBUST, (name, k, v)
    for name in ("stage_grads", "grads"):
        for k, v in report[name].items():
            assert v < TOL_ALL, (name, k, v)
