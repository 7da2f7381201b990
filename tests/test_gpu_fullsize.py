"""GPU: parity of the BENCHMARKED chain at BASELINE config-2 size (VERDICT r1 "next" #1a, SURVEY.md 8c protocol).

The exact step bench.py times -- env-map relight, 250,000 Gaussians, 2048x1334, B = 4 views per launch:
shade(env) -> project -> pruned bin/sort -> planar 2-px/lane colour+depth raster -> fused L1 -> packed-record raster
backward -> project backward -> shade backward -- is compared as a whole with the CPU oracle chain
(oracle/chain.py): rgb / alpha / depth and EVERY input gradient (f_vnocond, f_vcond, postex, tn, albedo) on whole
images (the OpenMP oracle needs ~2 s per view on the GPU box's cores, so no tile subsampling is necessary), plus the
fraction of pixels whose contributing list differs (threshold flips at alpha = 1/255, T = 1e-4, sigma < 0; seen in T)."""
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

TOL = 1e-4           # north_star: within 1e-4 L2 -- outputs AND gradients
# all-Gaussian cap on every gradient tensor incl. its explained outliers, ~3x what is measured (round 5/6: config 2 <= 7.8e-4;
# config 1, 10 k Gaussians, <= 4.7e-4 -- two near-pole Gaussians.  A flip pixel under config 1's few Gaussians can put
# percents on a tensor -- round 3 saw 2.75e-2 from 8 Gaussians: if the cap trips with every member of W flip-explained,
# say so in the failure instead of loosening the cap)
ALL_CAP = {"config2": 2.5e-3, "config1": 1.5e-3}
UNFLAGGED_BAR = 1e-4  # north_star's bar, over the Gaussians NO a-priori predicate flags (79-87 % of them; measured <= 7.6e-5: profiles/r06_fullsize_parity.json)
MAX_EXPLAINED = 2e-3  # at most this fraction of the Gaussians may need an explanation (measured: ~1e-4)
LEAVES = ("f_vn", "f_vc", "postex", "tn", "albedo")
STAGE = ("color", "opacity", "primpos", "primscale", "primqvec")

# How the gradient bar is applied.  Every stage of the chain, run on identical inputs, agrees with the oracle to ~1e-6
# (tests/test_gpu_exact_math.py, tools/stage_isolation.py -> profiles/r03_stage_isolation.json).  In the CHAIN the stages
# inherit each other's rounding (conics differ by 5e-7 rms after shade + projection), which moves alpha across the
# 1/255 cut -- or T across 1e-4 -- for a few 1e-5 of the pixels: one implementation composites a Gaussian the other skips.
# Such a "flip pixel" puts an O(1) relative error on the gradient of the Gaussians under it, and the env-map term (piecewise
# linear in the reflection direction: bilinear texel lookups, acos at the pole, clamps) does the same for a lookup within
# rounding of a derivative jump.  The test does NOT drop "the worst k" blindly: it removes the smallest set W of Gaussians
# whose removal brings the rel-L2 under the bar and asserts that EVERY member of W is explained by one of these
# predicates, evaluated on the oracle's data:
#   flip     the Gaussian reaches (alpha >= 0.5/255) a flagged pixel: a pixel whose contributor list differs, or one where
#            the L1 loss is evaluated at its kink -- rgb equals the target to rounding, so sign(rgb - target), the upstream
#            gradient, differs between the two implementations (the bench's target is random: +-1 signs cancel, a
#            Gaussian's gradient is a sum of ~sqrt(pixels) net units and one flipped sign is a few % of it)
#   border   its env lookup lies within 2e-3 texels of a texel border on one of the two mip levels it blends
#   pole     |r_y| > 0.98 (2 % of the directions): see fp64 -- there BOTH fp32 evaluations are ill-conditioned, and which of
#            the two is further from fp64 on a given Gaussian is chance
#   fp64     the fp32 ORACLE itself is at least 1/4 as far from its own fp64 evaluation (same upstream gradient,
#            oracle/chain.py:shade_leaf_grads) as HIP is from the fp32 oracle, on this Gaussian: the reference's formulation is
#            ill-conditioned there in fp32 -- v = acos(r_y), u = atan2(r_x, r_z) have derivatives 1 / sqrt(1 - r_y^2) and
#            1 / (r_x^2 + r_z^2), which amplify the rounding of the reflection direction by up to (1 - r_y^2)^-3/2 near the
#            poles, and grid_sample's fp32 texel coordinate can itself fall on the other side of a border (tools/shade_isolation.py:
#            in 2/3 of the largest HIP-vs-fp32-oracle disagreements of the shade backward HIP agrees with fp64 to 1e-6)
#   kink     a clamp of the shading tail is active within rounding (diffuse / colour at 0, specular at 1)
# An unexplained member of W fails the test.


from scenes import worst_set as _worst_set  # noqa: E402  (shared with __graft_entry__.smoke)


def _flip_touched(o, flag_yx):
    """[N] bool: does the Gaussian reach one of the flagged pixels (flag_yx [F, 2] = (row, col)) with alpha >= 1/2 of the
    1/255 cut (oracle's projected attributes: exactly the Gaussians a flagged pixel composites, plus the one at the cut)?"""
    xys, conics, op, radii = o["xys"], o["conics"], o["opac_eff"], o["radii"]
    out = torch.zeros(xys.shape[0], dtype=torch.bool)
    px, py = flag_yx[:, 1].float() + 0.5, flag_yx[:, 0].float() + 0.5
    for s in range(0, flag_yx.shape[0], 32):      # [N, 32] blocks
        dx, dy = xys[:, :1] - px[None, s:s + 32], xys[:, 1:] - py[None, s:s + 32]
        sigma = 0.5 * (conics[:, :1] * dx * dx + conics[:, 2:] * dy * dy) + conics[:, 1:2] * dx * dy
        out |= ((op[:, None] * torch.exp(-sigma) >= 0.5 / 255.0) & (sigma >= 0)).any(1)
    return out & (radii > 0)


def _shade_predicates(pr, t):
    """Per-Gaussian predicates of the env-map shading tail on the oracle's state (oracle/shade_ref.py): border, pole, kink."""
    import torch.nn.functional as F

    from oracle import shade_ref

    view = F.normalize(pr["primpos"] - t["campos"][:, None], dim=-1)
    n = pr["spec_nml"]
    refl = view - 2 * (view * n).sum(-1, keepdim=True) * n
    r = torch.einsum("bxy,bny->bnx", t["lightrot"], refl)[0]
    uv = shade_ref.dir2uv(r)
    q = len(t["mips"])
    level = (pr["sigma"][0] * 5).clamp(0, q - 1 - 1e-6)
    l0 = level.floor().long()
    border = torch.zeros(r.shape[0], dtype=torch.bool)
    for dl in (0, 1):
        l = (l0 + dl).clamp(max=q - 1)
        w, h = (1024 >> l).float(), (512 >> l).float()
        ix, iy = ((uv[:, 0] + 1) * w - 1) / 2, ((uv[:, 1] + 1) * h - 1) / 2
        fx, fy = ix - ix.floor(), iy - iy.floor()
        border |= (torch.minimum(fx, 1 - fx) < 2e-3) | (torch.minimum(fy, 1 - fy) < 2e-3)
    bdist = torch.full_like(level, 1.0)
    for dl in (0, 1):
        l = (l0 + dl).clamp(max=q - 1)
        w, h = (1024 >> l).float(), (512 >> l).float()
        ix, iy = ((uv[:, 0] + 1) * w - 1) / 2, ((uv[:, 1] + 1) * h - 1) / 2
        fx, fy = ix - ix.floor(), iy - iy.floor()
        bdist = torch.minimum(bdist, torch.minimum(torch.minimum(fx, 1 - fx), torch.minimum(fy, 1 - fy)))
    border |= (level - level.round()).abs() < 1e-5          # the blend switches mip levels
    _shade_predicates.diag = {"texel_border_dist": bdist, "r_y": r[:, 1], "u": uv[:, 0], "v": uv[:, 1], "mip_level": level,
                              "spec_max": pr["spec_color"][0].amax(-1), "spec_vis": pr["spec_vis"][0, :, 0],
                              "diff_min": pr["diff_color"][0].amin(-1)}
    pole = r[:, 1].abs() > 0.98
    kink = (pr["diff_color"][0].abs() < 1e-5).any(-1) | ((pr["spec_color"][0] / pr["spec_vis"][0].clamp(min=1e-6) - 1).abs() < 1e-5).any(-1)
    kink |= (pr["color"][0].abs() < 1e-6).any(-1) | ((pr["sigma"][0] - 0.01).abs() < 1e-7)
    return {"border": border, "pole": pole, "kink": kink}


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
    # the transmittance images of the same views (diagnostics: which pixels composite a different list)
    with torch.no_grad():
        intr = torch.stack([mb["K"][:, 0, 0], mb["K"][:, 1, 1], mb["K"][:, 0, 2], mb["K"][:, 1, 2]], -1)
        d = splat.render_views(preds["primpos"], preds["primscale"], preds["primqvec"], preds["opacity"], preds["color"],
                               mb["Rt"], intr, cfg["height"], cfg["width"])
        last = None
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
    ref_grads64 = {k: [] for k in LEAVES}
    ref_stage = {k: [] for k in STAGE}
    report = {"views": B, "gaussians": cfg["gaussians"], "image": [H, W], "outputs": {}, "grads": {}}
    flips, big, ref_loss, sign_flips = 0, 0, 0.0, 0
    worst = {"rgb": 0.0, "alpha": 0.0, "depth": 0.0, "depth_without_flip_pixels": 0.0}
    explained = {k: [] for k in ("flip", "border", "pole", "kink")}   # per view: [N] bool
    view_diag = []
    for b in range(B):
        one = {k: (v[b:b + 1].detach().clone().requires_grad_(v.requires_grad) if torch.is_tensor(v) and k != "albedo"
                   else v) for k, v in cpu.items()}
        one["albedo"] = cpu["albedo"].detach().clone().requires_grad_(True)
        one["mips"] = [m[b:b + 1] if m.shape[0] > 1 else m for m in cpu["mips"]]   # (config 2: ONE pyramid for all views)
        o = chain.cpu_view(one, H, W, loss_scale=1.0 / (B * 3 * H * W))
        ref_loss += o["loss"]
        if name == "config2":
            assert o["n_isect"] > 1_000_000  # config-2 density (the pruned HIP lists are ~2 M, the 3-sigma lists ~3.6 M)
        worst["rgb"] = max(worst["rgb"], rel_l2(rgb[b], o["rgb"]))
        worst["alpha"] = max(worst["alpha"], rel_l2(alpha[b, 0], o["alpha"]))
        worst["depth"] = max(worst["depth"], rel_l2(depth[b, 0], o["depth_norm"]))
        # flip pixels: the transmittance differs by more than rounding.  Any difference between the two composited lists
        # shows there: an entry at the alpha = 1/255 cut taken by one side only moves T by >= 0.39 %, a pixel that stops
        # (T <= 1e-4) at different entries ends with different products
        T_h, T_o = diag["final_T"][b, 0].cpu(), o["final_T"]
        flip = (T_h - T_o).abs() > 1e-3 * T_o.clamp(min=1e-4)
        flips += int(flip.sum())
        keep = ~flip
        worst["depth_without_flip_pixels"] = max(worst["depth_without_flip_pixels"],
                                                 rel_l2(depth[b, 0].cpu()[keep], o["depth_norm"][keep]))
        big += int(((rgb[b].cpu() - o["rgb"]).abs().amax(0) > 1e-3).sum())
        # L1 kink: the sign of (rgb - target) differs in some channel
        tgt = cpu["target"][b]
        sign_flip = (torch.sign(rgb[b].cpu() - tgt) != torch.sign(o["rgb"] - tgt)).any(0)
        sign_flips += int(sign_flip.sum())
        explained["flip"].append(_flip_touched(o, torch.nonzero(flip | sign_flip)))
        # depth gap to the nearest neighbour in depth order, in ulps
        dz = o["depths"].double()
        order = dz.argsort()
        gap = torch.full_like(dz, 1e30)
        dd = (dz[order][1:] - dz[order][:-1])
        gap[order[1:]] = torch.minimum(gap[order[1:]], dd)
        gap[order[:-1]] = torch.minimum(gap[order[:-1]], dd)
        ulp = torch.tensor(2.0).pow(torch.floor(torch.log2(dz.clamp(min=1e-30))) - 23)
        view_diag.append({"radius": o["radii"].float(), "x": o["xys"][:, 0], "y": o["xys"][:, 1], "depth": o["depths"],
                          "depth_gap_ulps": (gap / ulp).float(), "opacity": o["preds"]["opacity"][0, :, 0]})
        sp = _shade_predicates(o["preds"], one)
        for kk in ("border", "pole", "kink"):
            explained[kk].append(sp[kk])
        view_diag[-1].update(_shade_predicates.diag)
        for k in LEAVES:
            ref_grads[k].append(one[k].grad)
        g64 = chain.shade_leaf_grads(one, o["stage_grads"])
        for k in LEAVES:
            ref_grads64[k].append(g64[k])
        for k in STAGE:
            ref_stage[k].append(o["stage_grads"][k])
    N = cfg["gaussians"]
    ex = {k: torch.stack(v).flatten() for k, v in explained.items()}             # [B*N]
    ex_raster = ex["flip"]
    ex_any = ex["flip"] | ex["border"] | ex["kink"]      # (pole: only together with the fp64 yardstick, see judge)
    report["outputs"] = worst
    report["flip_pixel_fraction"] = flips / (B * H * W)
    report["flip_pixels"] = flips
    report["l1_sign_flip_pixels"] = sign_flips
    report["pixels_off_by_more_than_1e-3"] = big / (B * H * W)
    report["loss"] = {"hip": loss, "oracle": ref_loss}
    report["predicate_population"] = {k: float(v.float().mean()) for k, v in ex.items()}
    report["tolerance"] = TOL
    failures = []

    def judge(kind, k, a, b, allowed, b64=None, pole=None, exk=None):
        """a, b: [B, C, N].  Remove the smallest worst set W that brings the rest under TOL; every member must be explained.
        ADVICE r3: position alone (`pole`, 2 % of all directions) excuses nothing -- near-pole members of W are explained only
        while HIP is as close to the fp64 evaluation over the WHOLE near-pole class as the fp32 oracle is; and an
        all-Gaussian cap bounds what the explained outliers may add up to."""
        W_idx, all_rel, rest_rel = _worst_set(a, b, TOL)
        exk = ex if exk is None else exk   # predicate vectors in THIS tensor's Gaussian indexing (albedo: any view)
        by = {kk: int(exk[kk][W_idx].sum()) for kk in exk}
        allowed_in = allowed
        # the honest pair of numbers: rel-L2 over ALL Gaussians, and over the Gaussians that NO a-priori
        # predicate flags -- flip / border / kink, + the near-pole class for the leaves (all evaluated on the oracle's data
        # before HIP's gradients are looked at; NOT the worst-k, whose remainder is < TOL by construction)
        flagged = allowed if pole is None else (allowed | pole)
        e2 = (a.double().cpu() - b.double().cpu()).pow(2).sum(1).flatten()
        r2 = b.double().cpu().pow(2).sum(1).flatten()
        rel_unflagged = float((e2[~flagged].sum() / r2[~flagged].sum().clamp(min=1e-300)).sqrt())
        if b64 is not None:   # fp64 predicate: the fp32 oracle's own distance from its fp64 evaluation, per Gaussian
            e_h = (a.double().cpu() - b.double().cpu()).pow(2).sum(1).flatten()
            e_o = (b.double().cpu() - b64.double().cpu()).pow(2).sum(1).flatten()
            fp64 = e_o >= 0.0625 * e_h
            allowed = allowed | fp64
            if pole is not None:
                # near the poles the reference's formulation is ill-conditioned in fp32 (v = acos(r_y): the derivative
                # -1 / sqrt(1 - r_y^2) is formed from a difference that cancels to a few ulps of 1), and which of two fp32
                # evaluations lands further from fp64 on ONE Gaussian is chance -- so the yardstick is taken over the whole
                # class: HIP's distance from the fp64 evaluation over ALL near-pole Gaussians, relative to the fp32 oracle's.
                # Rounds 3-4: HIP was 3.6x (config 2) / 2.1x (config 1) FARTHER than the oracle (it formed 1 - r_y^2 the same
                # way with 2-3 more roundings in the reflection direction).  Round 5: shade.hip forms the polar angle as
                # atan2(sqrt(r_x^2 + r_z^2), r_y) and 1 - r_y^2 as r_x^2 + r_z^2 -- the same quantities without the
                # cancellation -- and is now 0.05x / 0.03x: 20-30x CLOSER to fp64 than the fp32 oracle itself
                # (profiles/r05_fullsize_parity.json: f_vc 3.7e-7 vs 7.5e-6 over 12.8 k Gaussians).  The tripwire is 2x: a
                # regression of the shade backward confined to near-pole directions un-explains every pole member of W.
                # (class = near-pole Gaussians that no other predicate touches: a flip pixel under a near-pole Gaussian changes
                # HIP's UPSTREAM gradient, which the fp64 evaluation -- fed the oracle's upstream -- does not see.  The floor
                # of 1e-7 of the tensor's norm keeps leaves that do not depend on the direction at all -- albedo: both
                # distances ~1e-11 -- from tripping it.)
                e_h64 = (a.double().cpu() - b64.double().cpu()).pow(2).sum(1).flatten()
                pure = pole & ~allowed_in
                hip_vs_64, orc_vs_64 = float(e_h64[pure].sum().sqrt()), float(e_o[pure].sum().sqrt())
                class_ok = hip_vs_64 <= 2.0 * orc_vs_64 + 1e-7 * float(b64.double().pow(2).sum().sqrt())
                report.setdefault("pole_class_vs_fp64", {})[k] = {"hip": hip_vs_64, "fp32_oracle": orc_vs_64,
                                                                   "gaussians": int(pure.sum()), "ok": bool(class_ok)}
                if class_ok:
                    allowed = allowed | pole
                by["pole_class_yardstick_ok"] = int(class_ok)
            by["fp64"] = int(fp64[W_idx].sum())
            report.setdefault("fp32_oracle_vs_fp64_oracle_rel_l2", {})[k] = float(
                (e_o.sum() / b64.double().pow(2).sum()).sqrt())
        unexplained = int((~allowed[W_idx]).sum())
        report.setdefault(kind, {})[k] = {"rel_l2_all_gaussians": all_rel, "rel_l2_without_flagged": rel_unflagged,
                                          "flagged_fraction": float(flagged.float().mean()),
                                          "rel_l2_without_W": rest_rel,
                                          "W_size": int(W_idx.numel()), "W_fraction": W_idx.numel() / (B * N),
                                          "W_explained_by": by, "W_unexplained": unexplained}
        if unexplained:   # diagnostics of the unexplained members (oracle data)
            bad = W_idx[~allowed[W_idx]][:12]
            a2, b2 = a.double().cpu(), b.double().cpu()
            det = []
            for gi in bad.tolist():
                vb, g = divmod(gi, a2.shape[2]) if a2.shape[0] > 1 else (0, gi)
                info = {"view": vb, "g": g, "err_over_ref": float((a2[vb, :, g] - b2[vb, :, g]).norm() / b2[vb, :, g].norm().clamp(min=1e-300)),
                        "ref_over_median": float(b2[vb, :, g].norm() / b2[vb].norm(dim=0).median().clamp(min=1e-300))}
                if vb < len(view_diag):
                    info.update({kk: float(vv[g]) for kk, vv in view_diag[vb].items()})
                det.append(info)
            report[kind][k]["unexplained_examples"] = det
        # the un-flagged 80 % of the Gaussians carry no discontinuity by construction of the predicates: they must agree
        # to the bar outright (what remains there is the fp64-conditioning class, measured <= 1e-4)
        if (unexplained or rest_rel > TOL or W_idx.numel() > MAX_EXPLAINED * B * N or all_rel > ALL_CAP[name]
                or rel_unflagged > UNFLAGGED_BAR):
            failures.append((kind, k, report[kind][k]))

    for b in range(B):   # per-Gaussian relative error of the stage gradients (diagnostics of the leaf outliers)
        for k in ("color", "primpos"):
            a_, b_ = mb["_stage"][k][b].reshape(N, -1).double().cpu(), ref_stage[k][b].reshape(N, -1).double()
            view_diag[b][f"stage_{k}_relerr"] = ((a_ - b_).norm(dim=1) / b_.norm(dim=1).clamp(min=1e-300)).float()
            view_diag[b][f"stage_{k}_ref_norm"] = b_.norm(dim=1).float()
    for k in STAGE:   # gradients at the raster / projection boundary: only a flip pixel explains an outlier there
        ref = torch.stack(ref_stage[k])                                   # [B, N, C]
        judge("stage_grads", k, mb["_stage"][k].reshape(B, N, -1).transpose(1, 2), ref.transpose(1, 2), ex_raster)
    for k in LEAVES:  # leaf gradients: flip pixels + the derivative jumps of the env-map shading tail
        if k == "albedo":
            # shared by the B views: a texel is explained if it is in ANY view
            ref = torch.stack(ref_grads[k]).sum(0)                                              # [1, N, 3]
            ref64 = torch.stack(ref_grads64[k]).sum(0)
            # (no pole yardstick: the albedo gradient is D x upstream and does not depend on the lookup direction; both
            # distances from fp64 are rounding noise, ~1e-11 -- the row said `ok: false` on a zero-vs-zero comparison)
            judge("grads", k, mb[k].grad.reshape(1, N, 3).transpose(1, 2), ref.reshape(1, N, 3).transpose(1, 2),
                  ex_any.reshape(B, N).any(0), ref64.reshape(1, N, 3).transpose(1, 2), None,
                  exk={kk: v.reshape(B, N).any(0) for kk, v in ex.items()})
            report["grads"][k]["W_fraction"] = report["grads"][k]["W_size"] / N
            continue
        ref = torch.cat(ref_grads[k], 0)
        judge("grads", k, mb[k].grad.reshape(B, -1, N), ref.reshape(B, -1, N), ex_any,
              torch.cat(ref_grads64[k], 0).reshape(B, -1, N), ex["pole"])
    print(f"\nCHAIN_PARITY {name} " + json.dumps(report))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        from goliath_amd import build

        report["_stamp"] = {"csrc_sha16": build.source_digest()}   # bench.py quotes the copy under profiles/ while it matches
        with open(os.path.join(out_dir, "fullsize_parity.json" if name == "config2" else f"chain_parity_{name}.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert abs(loss - ref_loss) < 1e-5 * abs(ref_loss)
    for k in ("rgb", "alpha", "depth_without_flip_pixels"):
        assert worst[k] < TOL, (k, worst[k])
    # depth / clamp(alpha, .05, 1): under the flip predicate -- without the flip pixels (a few hundred of 10.9 M, asserted
    # < 1e-3 of the image below) it is 1.2e-6; with them 1.03e-4 (the division amplifies a flipped list 20x)
    assert worst["depth_without_flip_pixels"] < 2e-5, worst["depth_without_flip_pixels"]
    assert worst["depth"] < 1.5e-4, worst["depth"]
    assert report["flip_pixel_fraction"] < 1e-3, report["flip_pixel_fraction"]  # expected << 0.1 %
    assert not failures, failures
