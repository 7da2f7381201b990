"""oracle/cref64.py -- TEST INFRASTRUCTURE ONLY.

fp64 twins (orc64_*) of the gsplat restatement's projection / rasterization passes (oracle/gsplat_oracle.c compiled
with -DORC_FP64).  Used by tests/test_oracle_gsplat_fd.py only: finite differences of the fp64 FORWARD against the
hand-written BACKWARD, with the tile lists frozen (the lists are piecewise constant in the inputs)."""
import ctypes

import torch

from . import cref

c_int, c_double = ctypes.c_int, ctypes.c_double


def _d(t):
    return t.detach().to(torch.float64).contiguous().cpu()


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def project_fwd(means, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, H, W, block, clip):
    means, scales, quats, viewmat = map(_d, (means, scales, quats, viewmat))
    N = means.shape[0]
    cov3d, xys, depths = torch.zeros(N, 6, dtype=torch.float64), torch.zeros(N, 2, dtype=torch.float64), torch.zeros(N, dtype=torch.float64)
    radii, nth = torch.zeros(N, dtype=torch.int32), torch.zeros(N, dtype=torch.int32)
    conics, comp = torch.zeros(N, 3, dtype=torch.float64), torch.zeros(N, dtype=torch.float64)
    cref.lib().orc64_project_fwd(c_int(N), _p(means), _p(scales), c_double(glob_scale), _p(quats), _p(viewmat),
                                 c_double(fx), c_double(fy), c_double(cx), c_double(cy), c_int(H), c_int(W), c_int(block),
                                 c_double(clip), _p(cov3d), _p(xys), _p(depths), _p(radii), _p(conics), _p(comp), _p(nth))
    return xys, depths, radii, conics, comp, nth, cov3d


def project_bwd(means, scales, glob_scale, quats, viewmat, fx, fy, cov3d, radii, conics, comp, v_xy, v_depth, v_conic,
                v_comp):
    means, scales, quats, viewmat = map(_d, (means, scales, quats, viewmat))
    N = means.shape[0]
    z = lambda *s: torch.zeros(*s, dtype=torch.float64)
    v_cov2d, v_cov3d, v_mean, v_scale, v_quat = z(N, 3), z(N, 6), z(N, 3), z(N, 3), z(N, 4)
    cref.lib().orc64_project_bwd(c_int(N), _p(means), _p(scales), c_double(glob_scale), _p(quats), _p(viewmat),
                                 c_double(fx), c_double(fy), _p(_d(cov3d)), _p(radii.to(torch.int32).contiguous()),
                                 _p(_d(conics)), _p(_d(comp)), _p(_d(v_xy)), _p(_d(v_depth)), _p(_d(v_conic)),
                                 _p(_d(v_comp)), _p(v_cov2d), _p(v_cov3d), _p(v_mean), _p(v_scale), _p(v_quat))
    return v_mean, v_scale, v_quat


def rasterize_fwd(ids, bins, xys, conics, colors, opacity, H, W, block, background, alpha_cap=cref.ALPHA_CAP_FWD):
    xys, conics, colors, opacity, background = map(_d, (xys, conics, colors, opacity.reshape(-1), background))
    C = colors.shape[-1]
    out = torch.zeros(H, W, C, dtype=torch.float64)
    Ts = torch.zeros(H, W, dtype=torch.float64)
    idx = torch.zeros(H, W, dtype=torch.int32)
    cref.lib().orc64_rasterize_fwd(c_int(H), c_int(W), c_int(block), c_int(C), _p(ids.contiguous()), _p(bins.contiguous()),
                                   _p(xys), _p(conics), _p(colors), _p(opacity), _p(background), c_double(alpha_cap),
                                   _p(out), _p(Ts), _p(idx))
    return out, Ts, idx


def rasterize_bwd(ids, bins, xys, conics, colors, opacity, H, W, block, background, Ts, idx, v_out, v_out_alpha,
                  alpha_cap_bwd=cref.ALPHA_CAP_BWD):
    xys, conics, colors, opacity, background = map(_d, (xys, conics, colors, opacity.reshape(-1), background))
    N, C = colors.shape
    z = lambda *s: torch.zeros(*s, dtype=torch.float64)
    v_xy, v_conic, v_col, v_op = z(N, 2), z(N, 3), z(N, C), z(N)
    cref.lib().orc64_rasterize_bwd(c_int(H), c_int(W), c_int(block), c_int(C), c_int(N), _p(ids.contiguous()),
                                   _p(bins.contiguous()), _p(xys), _p(conics), _p(colors), _p(opacity), _p(background),
                                   _p(_d(Ts)), _p(idx.contiguous()), _p(_d(v_out)),
                                   _p(None if v_out_alpha is None else _d(v_out_alpha)), c_double(alpha_cap_bwd),
                                   _p(v_xy), _p(v_conic), _p(v_col), _p(v_op))
    return v_xy, v_conic, v_col, v_op
