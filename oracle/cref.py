"""oracle/cref.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of the plain-C CPU oracle (oracle/*.c -> oracle/_build/liboracle.so).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package `goliath_amd` never does.

Every function takes/returns CPU float32/int32 torch tensors so the parity tests
read like calls of the reference's own operators:
  gsplat.project_gaussians / rasterize_gaussians   (ca_code/utils/render_gsplat.py:49-104)
  sgutilslib.evaluate_gaussian_{fwd,bwd}           (extensions/sgutils/sg.cu:177-283)
PARITY UNPINNED for both (see the .c headers).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

ALPHA_CAP_FWD = 0.999  # gsplat 0.1.11 forward.cu
ALPHA_CAP_BWD = 0.99  # gsplat 0.1.11 backward.cu


def build(force=False):
    """Build oracle/_build/liboracle.so (our restatements) and, where /root/reference exists, oracle/_ref/libref.so (the
    reference's own sgutils / raydirs kernels compiled for the host).  `make` decides what is stale."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_bin_sort.restype = ctypes.c_int64
    return _lib


def set_threads(n):
    """Number of OpenMP threads the C oracle uses (it is linked against libgomp)."""
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    lib()


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.device.type == "cpu" and t.is_contiguous(), "oracle takes contiguous CPU tensors"
    return ctypes.c_void_p(t.data_ptr())


def _f(t):
    return t.detach().to(torch.float32).contiguous().cpu()


c_int, c_float, c_i64 = ctypes.c_int, ctypes.c_float, ctypes.c_int64


# --------------------------------------------------------------------------- gsplat
def project_gaussians(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy,
                      img_height, img_width, block_width, clip_thresh=0.01):
    """gsplat.project_gaussians forward; same return tuple as gsplat 0.1.11."""
    means3d, scales, quats, viewmat = _f(means3d), _f(scales), _f(quats), _f(viewmat)
    N = means3d.shape[0]
    cov3d = torch.zeros(N, 6)
    xys = torch.zeros(N, 2)
    depths = torch.zeros(N)
    radii = torch.zeros(N, dtype=torch.int32)
    conics = torch.zeros(N, 3)
    comp = torch.zeros(N)
    nth = torch.zeros(N, dtype=torch.int32)
    lib().orc_project_fwd(c_int(N), _p(means3d), _p(scales), c_float(glob_scale), _p(quats),
                          _p(viewmat), c_float(fx), c_float(fy), c_float(cx), c_float(cy),
                          c_int(img_height), c_int(img_width), c_int(block_width),
                          c_float(clip_thresh), _p(cov3d), _p(xys), _p(depths), _p(radii),
                          _p(conics), _p(comp), _p(nth))
    return xys, depths, radii, conics, comp, nth, cov3d


def project_gaussians_backward(means3d, scales, glob_scale, quats, viewmat, fx, fy, cov3d, radii,
                               conics, compensation, v_xy, v_depth, v_conic, v_compensation):
    means3d, scales, quats, viewmat = _f(means3d), _f(scales), _f(quats), _f(viewmat)
    N = means3d.shape[0]
    v_cov2d = torch.zeros(N, 3)
    v_cov3d = torch.zeros(N, 6)
    v_mean = torch.zeros(N, 3)
    v_scale = torch.zeros(N, 3)
    v_quat = torch.zeros(N, 4)
    lib().orc_project_bwd(c_int(N), _p(means3d), _p(scales), c_float(glob_scale), _p(quats),
                          _p(viewmat), c_float(fx), c_float(fy), _p(_f(cov3d)),
                          _p(radii.to(torch.int32).contiguous()), _p(_f(conics)),
                          _p(_f(compensation)), _p(_f(v_xy)), _p(_f(v_depth)), _p(_f(v_conic)),
                          _p(_f(v_compensation)), _p(v_cov2d), _p(v_cov3d), _p(v_mean),
                          _p(v_scale), _p(v_quat))
    return v_cov2d, v_cov3d, v_mean, v_scale, v_quat


def bin_and_sort(xys, depths, radii, num_tiles_hit, img_height, img_width, block_width):
    """SURVEY A.2.  Returns isect_ids_sorted[I] i64, gaussian_ids_sorted[I] i32, tile_bins[T,2] i32."""
    xys, depths = _f(xys), _f(depths)
    radii = radii.to(torch.int32).contiguous()
    I = int(num_tiles_hit.to(torch.int64).sum())
    tiles = ((img_width + block_width - 1) // block_width) * ((img_height + block_width - 1) // block_width)
    keys = torch.zeros(max(I, 1), dtype=torch.int64)
    ids = torch.zeros(max(I, 1), dtype=torch.int32)
    bins = torch.zeros(tiles, 2, dtype=torch.int32)
    n = lib().orc_bin_sort(c_int(xys.shape[0]), _p(xys), _p(depths), _p(radii), c_int(img_height),
                           c_int(img_width), c_int(block_width), c_i64(I), _p(keys), _p(ids), _p(bins))
    assert n == I, (n, I)
    return keys[:I], ids[:I], bins


def rasterize_forward(ids_sorted, tile_bins, xys, conics, colors, opacity, img_height, img_width,
                      block_width, background, alpha_cap=ALPHA_CAP_FWD):
    """SURVEY A.3.  Returns out_img[H,W,C], final_Ts[H,W], final_idx[H,W]."""
    xys, conics, colors, opacity, background = map(_f, (xys, conics, colors, opacity.reshape(-1), background))
    C = colors.shape[-1]
    assert C <= 16
    out = torch.zeros(img_height, img_width, C)
    Ts = torch.zeros(img_height, img_width)
    idx = torch.zeros(img_height, img_width, dtype=torch.int32)
    lib().orc_rasterize_fwd(c_int(img_height), c_int(img_width), c_int(block_width), c_int(C),
                            _p(ids_sorted.contiguous()), _p(tile_bins.contiguous()), _p(xys),
                            _p(conics), _p(colors), _p(opacity), _p(background),
                            c_float(alpha_cap), _p(out), _p(Ts), _p(idx))
    return out, Ts, idx


def rasterize_backward(ids_sorted, tile_bins, xys, conics, colors, opacity, img_height, img_width,
                       block_width, background, final_Ts, final_idx, v_out, v_out_alpha=None,
                       alpha_cap_bwd=ALPHA_CAP_BWD):
    """SURVEY A.4.  Returns v_xy[N,2], v_conic[N,3], v_colors[N,C], v_opacity[N,1]."""
    xys, conics, colors, opacity, background = map(_f, (xys, conics, colors, opacity.reshape(-1), background))
    N, C = colors.shape
    v_xy = torch.zeros(N, 2)
    v_conic = torch.zeros(N, 3)
    v_col = torch.zeros(N, C)
    v_op = torch.zeros(N)
    lib().orc_rasterize_bwd(c_int(img_height), c_int(img_width), c_int(block_width), c_int(C),
                            c_int(N), _p(ids_sorted.contiguous()), _p(tile_bins.contiguous()),
                            _p(xys), _p(conics), _p(colors), _p(opacity), _p(background),
                            _p(_f(final_Ts)), _p(final_idx.to(torch.int32).contiguous()),
                            _p(_f(v_out)), _p(None if v_out_alpha is None else _f(v_out_alpha)),
                            c_float(alpha_cap_bwd), _p(v_xy), _p(v_conic), _p(v_col), _p(v_op))
    return v_xy, v_conic, v_col, v_op[:, None]


def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                        img_width, block_width, background=None, return_alpha=False):
    """gsplat.rasterize_gaussians forward semantics incl. the I<1 quirk."""
    C = colors.shape[-1]
    if background is None:
        background = torch.ones(C)
    I = int(num_tiles_hit.to(torch.int64).sum())
    if I < 1:
        out = torch.ones(img_height, img_width, C) * _f(background)
        Ts = torch.zeros(img_height, img_width)
        ctx = None
    else:
        _, ids, bins = bin_and_sort(xys, depths, radii, num_tiles_hit, img_height, img_width, block_width)
        out, Ts, idx = rasterize_forward(ids, bins, xys, conics, colors, opacity, img_height,
                                         img_width, block_width, background)
        ctx = (ids, bins, Ts, idx)
    if return_alpha:
        return out, 1.0 - Ts, ctx
    return out, ctx


# --------------------------------------------------------------------------- sgutils
def evaluate_gaussian_fwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, w_type=0):
    """sgutilslib.evaluate_gaussian_fwd (sg.cu:177-226); returns integral[N,D,3]."""
    lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts = map(
        _f, (lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts))
    N, D = lobe_dirs.shape[:2]
    L = light_values.shape[1]
    out = torch.empty(N, D, 3)
    lib().orc_sg_fwd(c_int(N), c_int(D), c_int(L), _p(lobe_dirs), _p(lobe_sigmas), _p(light_values),
                     _p(light_pts), _p(prim_pts), _p(n_lights.to(torch.int32).contiguous().cpu()),
                     c_int(w_type), _p(out))
    return out


def evaluate_gaussian_bwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights,
                          grad_integral, w_type=0, want_light_grad=False):
    """sgutilslib.evaluate_gaussian_bwd (sg.cu:228-278)."""
    lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, grad_integral = map(
        _f, (lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, grad_integral))
    N, D = lobe_dirs.shape[:2]
    L = light_values.shape[1]
    gd = torch.zeros(N, D, 3)
    gs = torch.zeros(lobe_sigmas.shape)
    gl = torch.zeros(N, L, 3) if want_light_grad else None
    lib().orc_sg_bwd(c_int(N), c_int(D), c_int(L), _p(lobe_dirs), _p(lobe_sigmas), _p(light_values),
                     _p(light_pts), _p(prim_pts), _p(n_lights.to(torch.int32).contiguous().cpu()),
                     _p(grad_integral), c_int(w_type), _p(gd), _p(gs), _p(gl))
    return gd, gs, gl


# --------------------------------------------------------------------------- mvpraymarch / raydirs
def mvp_set_footprint(w=0, h=0, cap=0):
    """Hit-list semantics of mvp_forward / mvp_backward: (0, 0, 0) = per ray, uncapped (default);
    (8, 4, 512) = the reference's warp footprint and cap; (8, 8, 512) = csrc/mvp.hip's wave footprint."""
    lib().orc_mvp_set_footprint(c_int(w), c_int(h), c_int(cap))


def mvp_forward(raypos, raydir, stepsize, tminmax, primpos, primrot, primscale, template, fadescale=8.0,
                fadeexp=8.0, want_raysat=True, with_shadow=False, warp=None):
    """mvpraymarchlib.raymarch_forward semantics (algo 0, or algo 1 with warp[N,K,WD,WH,WW,3]; chlast).
    template[N,K,TD,TH,TW,4].  Returns rayrgba[N,H,W,4], raysat[N,H,W,3] (or None), shadow[N,K,TD,TH,TW,2] (or None)."""
    raypos, raydir, tminmax, primpos, primrot, primscale, template = map(
        _f, (raypos, raydir, tminmax, primpos, primrot, primscale, template))
    N, H, W = raypos.shape[:3]
    K, TD, TH, TW = template.shape[1:5]
    out = torch.empty(N, H, W, 4)
    sat = torch.empty(N, H, W, 3) if want_raysat else None
    shadow = torch.zeros(N, K, TD, TH, TW, 2) if with_shadow else None
    warp = None if warp is None else _f(warp)
    WD, WH, WW = (0, 0, 0) if warp is None else warp.shape[2:5]
    lib().orc_mvp_fwd_warp(c_int(N), c_int(H), c_int(W), c_int(K), _p(raypos), _p(raydir), c_float(stepsize), _p(tminmax),
                           _p(primpos), _p(primrot), _p(primscale), _p(template), c_int(TD), c_int(TH), c_int(TW),
                           _p(warp), c_int(WD), c_int(WH), c_int(WW),
                           c_float(fadescale), c_float(fadeexp), _p(out), _p(sat), _p(shadow))
    return out, sat, shadow


def mvp_backward(raypos, raydir, stepsize, tminmax, primpos, primrot, primscale, template, raysat, grad_rayrgba,
                 fadescale=8.0, fadeexp=8.0, warp=None):
    """mvpraymarchlib.raymarch_backward semantics; returns grads (primpos, primrot, primscale, template) -- and the warp
    field's as a fifth value when `warp` is given."""
    raypos, raydir, tminmax, primpos, primrot, primscale, template, raysat, grad_rayrgba = map(
        _f, (raypos, raydir, tminmax, primpos, primrot, primscale, template, raysat, grad_rayrgba))
    N, H, W = raypos.shape[:3]
    K, TD, TH, TW = template.shape[1:5]
    gp, gr, gs, gt = torch.zeros_like(primpos), torch.zeros_like(primrot), torch.zeros_like(primscale), torch.zeros_like(template)
    warp = None if warp is None else _f(warp)
    gw = None if warp is None else torch.zeros_like(warp)
    WD, WH, WW = (0, 0, 0) if warp is None else warp.shape[2:5]
    lib().orc_mvp_bwd_warp(c_int(N), c_int(H), c_int(W), c_int(K), _p(raypos), _p(raydir), c_float(stepsize), _p(tminmax),
                           _p(primpos), _p(primrot), _p(primscale), _p(template), c_int(TD), c_int(TH), c_int(TW),
                           _p(warp), c_int(WD), c_int(WH), c_int(WW),
                           c_float(fadescale), c_float(fadeexp), _p(raysat), _p(grad_rayrgba), _p(gp), _p(gr), _p(gs), _p(gt),
                           _p(gw))
    return (gp, gr, gs, gt) if warp is None else (gp, gr, gs, gt, gw)


def mvp_aabb(primpos, primrot, primscale):
    primpos, primrot, primscale = map(_f, (primpos, primrot, primscale))
    N, K = primpos.shape[:2]
    out = torch.empty(N, 2 * K - 1, 2, 3)
    lib().orc_mvp_aabb(c_int(N), c_int(K), _p(primpos), _p(primrot), _p(primscale), _p(out))
    return out


def compute_raydirs(viewpos, viewrot, focal, princpt, pixelcoords, volradius):
    """utilslib.compute_raydirs_forward; pixelcoords a [N,H,W,2] tensor or a (W, H) tuple."""
    viewpos, viewrot, focal, princpt = map(_f, (viewpos, viewrot, focal, princpt))
    N = viewpos.shape[0]
    if isinstance(pixelcoords, tuple):
        W, H = pixelcoords
        pc = None
    else:
        pc = _f(pixelcoords)
        H, W = pc.shape[1:3]
    rp, rd, tm = torch.empty(N, H, W, 3), torch.empty(N, H, W, 3), torch.empty(N, H, W, 2)
    lib().orc_raydirs(c_int(N), c_int(H), c_int(W), _p(viewpos), _p(viewrot), _p(focal), _p(princpt), _p(pc),
                      c_float(volradius), _p(rp), _p(rd), _p(tm))
    return rp, rd, tm
