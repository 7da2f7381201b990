"""oracle/gsplat_module.py -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).

A `gsplat`-shaped module over the plain-C CPU oracle (oracle/gsplat_oracle.c through oracle/cref.py): the two operators
the reference imports (ca_code/utils/render_gsplat.py:10-11) as differentiable torch functions with gsplat 0.1.11's
call signatures, return tuples and backward wiring, so that the REFERENCE's own Python -- `render_gsplat.render`,
`AutoEncoder.render`, `AutoEncoder.forward` -- runs unchanged on the CPU in the build container:

    import oracle.gsplat_module as g;  sys.modules["gsplat"] = g.as_module()

Used by tests/golden/make_rgca_model_golden.py (the generator of the model-level fixture).  PARITY UNPINNED for the
arithmetic inside (gsplat's sources are not in /root/reference; see gsplat_oracle.c) -- what this module adds is only the
autograd plumbing, which the reference's call sites fix: which outputs are differentiable and what their gradients feed.
"""
import types

import torch

from . import cref


class _Project(torch.autograd.Function):
    """project_gaussians: gradients reach means3d, scales, quats from (xys, depths, conics, compensation); radii,
    num_tiles_hit and cov3d carry none (gsplat 0.1.11 project_gaussians.py backward)."""

    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height, img_width, block_width,
                clip_thresh):
        xys, depths, radii, conics, comp, nth, cov3d = cref.project_gaussians(
            means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height, img_width, block_width, clip_thresh)
        ctx.save_for_backward(means3d.detach(), scales.detach(), quats.detach(), viewmat.detach(), cov3d, radii, conics, comp)
        ctx.cfg = (float(glob_scale), float(fx), float(fy))
        ctx.mark_non_differentiable(radii, nth, cov3d)
        return xys, depths, radii, conics, comp, nth, cov3d

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_comp, v_nth, v_cov3d):
        means3d, scales, quats, viewmat, cov3d, radii, conics, comp = ctx.saved_tensors
        glob_scale, fx, fy = ctx.cfg
        z = lambda g, like: torch.zeros_like(like) if g is None else g
        _, _, v_mean, v_scale, v_quat = cref.project_gaussians_backward(
            means3d, scales, glob_scale, quats, viewmat, fx, fy, cov3d, radii, conics, comp,
            z(v_xys, conics[:, :2]), z(v_depths, comp), z(v_conics, conics), z(v_comp, comp))
        return (v_mean, v_scale, None, v_quat) + (None,) * 9


class _Rasterize(torch.autograd.Function):
    """rasterize_gaussians: gradients reach xys, conics, colors, opacity (not depths / radii: they only order and
    bin).  Fewer than one intersection: the background image and zero alpha, no graph (rasterize.py quirk, SURVEY B#8)."""

    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width, block_width,
                background):
        C = colors.shape[-1]
        I = int(num_tiles_hit.to(torch.int64).sum())
        ctx.empty = I < 1
        if ctx.empty:
            return torch.ones(img_height, img_width, C) * background, torch.zeros(img_height, img_width)
        _, ids, bins = cref.bin_and_sort(xys, depths, radii, num_tiles_hit, img_height, img_width, block_width)
        out, Ts, idx = cref.rasterize_forward(ids, bins, xys, conics, colors, opacity, img_height, img_width, block_width,
                                              background)
        ctx.save_for_backward(ids, bins, xys.detach(), conics.detach(), colors.detach(), opacity.detach(), background, Ts, idx)
        ctx.cfg = (img_height, img_width, block_width)
        return out, 1.0 - Ts

    @staticmethod
    def backward(ctx, v_out, v_alpha):
        if ctx.empty:
            return (None,) * 11
        ids, bins, xys, conics, colors, opacity, background, Ts, idx = ctx.saved_tensors
        H, W, block = ctx.cfg
        v_out = torch.zeros(H, W, colors.shape[-1]) if v_out is None else v_out
        v_xy, v_conic, v_col, v_op = cref.rasterize_backward(ids, bins, xys, conics, colors, opacity, H, W, block, background,
                                                             Ts, idx, v_out, v_alpha)
        return v_xy, None, None, v_conic, None, v_col, v_op.reshape(opacity.shape), None, None, None, None


def project_gaussians(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height, img_width, block_width,
                      clip_thresh=0.01):
    return _Project.apply(means3d.contiguous(), scales.contiguous(), glob_scale, quats.contiguous(), viewmat.contiguous(),
                          fx, fy, cx, cy, img_height, img_width, block_width, clip_thresh)


def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width, block_width,
                        background=None, return_alpha=False):
    if colors.dim() != 2 or opacity.dim() != 2 or opacity.shape[1] != 1:
        raise ValueError("colors [N,C] and opacity [N,1] expected")
    if background is None:
        background = torch.ones(colors.shape[-1])
    out, alpha = _Rasterize.apply(xys.contiguous(), depths.contiguous(), radii.contiguous(), conics.contiguous(),
                                  num_tiles_hit.contiguous(), colors.contiguous(), opacity.contiguous(), img_height,
                                  img_width, block_width, background.to(torch.float32).contiguous())
    return (out, alpha) if return_alpha else out


def as_module():
    m = types.ModuleType("gsplat")
    m.project_gaussians, m.rasterize_gaussians = project_gaussians, rasterize_gaussians
    m.__version__ = "0.1.11+cpu-oracle"
    return m
