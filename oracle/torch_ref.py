"""oracle/torch_ref.py -- TEST INFRASTRUCTURE ONLY.

Independent, differentiable PyTorch restatements of the hot-path operators.  They
exist so that torch autograd can cross-check the explicit backward passes of the
C oracle (oracle/*.c) and, through it, of the HIP kernels.  Written in a
different style (vectorised, autograd) from the scalar C code on purpose.

  project_gaussians / rasterize : gsplat 0.1.11 semantics, SURVEY.md Appendix A
                                  (call sites ca_code/utils/render_gsplat.py:49-104)
  evaluate_gaussian             : extensions/sgutils/sg.cu:27-76
PARITY UNPINNED (no reference test or golden vector exists for either).
"""
import math

import torch
import torch.nn.functional as F

BLUR, FOV_CLAMP, EIG_FLOOR, Z_EPS = 0.3, 1.3, 0.1, 1e-6
ALPHA_FLOOR, T_STOP = 1.0 / 255.0, 1e-4


def quat_to_rotmat(q):
    q = F.normalize(q, dim=-1)
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(*q.shape[:-1], 3, 3)


def project_gaussians(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy,
                      img_height, img_width, block_width, clip_thresh=0.01):
    """Differentiable SURVEY A.1.  Returns (xys, depths, radii, conics, compensation,
    num_tiles_hit, cov3d); culled rows are zero."""
    R = viewmat[:3, :3]
    t = means3d @ R.T + viewmat[:3, 3]
    z = t[:, 2]
    valid = z > clip_thresh
    M = quat_to_rotmat(quats) * (glob_scale * scales)[:, None, :]
    cov3 = M @ M.transpose(1, 2)
    tan_x, tan_y = 0.5 * img_width / fx, 0.5 * img_height / fy
    zs = torch.where(valid, z, torch.ones_like(z))
    ex = zs * torch.clamp(t[:, 0] / zs, -FOV_CLAMP * tan_x, FOV_CLAMP * tan_x)
    ey = zs * torch.clamp(t[:, 1] / zs, -FOV_CLAMP * tan_y, FOV_CLAMP * tan_y)
    rz = 1.0 / zs
    zero = torch.zeros_like(rz)
    J = torch.stack([fx * rz, zero, -fx * ex * rz * rz, zero, fy * rz, -fy * ey * rz * rz], -1).reshape(-1, 2, 3)
    T = J @ R
    cov2 = T @ cov3 @ T.transpose(1, 2)
    c00, c01, c11 = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    det_orig = c00 * c11 - c01 * c01
    a, b, c = c00 + BLUR, c01, c11 + BLUR
    det = a * c - b * b
    comp = torch.sqrt(torch.clamp(det_orig / det, min=0.0))
    valid = valid & (det != 0)
    conics = torch.stack([c / det, -b / det, a / det], -1)
    bb = 0.5 * (a + c)
    sq = torch.sqrt(torch.clamp(bb * bb - det, min=EIG_FLOOR))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(bb + sq, bb - sq)))
    rw = 1.0 / (zs + Z_EPS)
    xys = torch.stack([fx * t[:, 0] * rw + cx, fy * t[:, 1] * rw + cy], -1)
    tiles_x = (img_width + block_width - 1) // block_width
    tiles_y = (img_height + block_width - 1) // block_width
    tc = (xys / block_width).detach()
    tr = (radius / block_width).detach()[:, None]
    tmin = torch.stack([(tc[:, 0:1] - tr).int().clamp(0, tiles_x), (tc[:, 1:2] - tr).int().clamp(0, tiles_y)], -1)
    tmax = torch.stack([(tc[:, 0:1] + tr + 1).int().clamp(0, tiles_x), (tc[:, 1:2] + tr + 1).int().clamp(0, tiles_y)], -1)
    area = ((tmax - tmin)[..., 0] * (tmax - tmin)[..., 1]).reshape(-1)
    valid = valid & (area > 0)
    vf = valid.to(means3d.dtype)
    cov3d = torch.stack([cov3[:, 0, 0], cov3[:, 0, 1], cov3[:, 0, 2], cov3[:, 1, 1], cov3[:, 1, 2], cov3[:, 2, 2]], -1)
    return (xys * vf[:, None], z * vf, (radius * vf).int(), conics * vf[:, None], comp * vf,
            (area * valid).int(), cov3d)


def rasterize(ids_sorted, tile_bins, xys, conics, colors, opacity, img_height, img_width,
              block_width, background, alpha_cap=0.999):
    """Differentiable SURVEY A.3 given a sorted intersection list (tile by tile, vectorised
    over the tile's pixels).  Returns out_img[H,W,C], final_Ts[H,W]."""
    C = colors.shape[-1]
    opacity = opacity.reshape(-1)
    tiles_x = (img_width + block_width - 1) // block_width
    tiles_y = (img_height + block_width - 1) // block_width
    out = torch.zeros(img_height, img_width, C, dtype=colors.dtype)
    Ts = torch.ones(img_height, img_width, dtype=colors.dtype)
    out_rows, T_rows = [], []
    for ty in range(tiles_y):
        y0, y1 = ty * block_width, min((ty + 1) * block_width, img_height)
        row_out, row_T = [], []
        for tx in range(tiles_x):
            x0, x1 = tx * block_width, min((tx + 1) * block_width, img_width)
            lo, hi = int(tile_bins[ty * tiles_x + tx, 0]), int(tile_bins[ty * tiles_x + tx, 1])
            h, w = y1 - y0, x1 - x0
            if hi <= lo:
                row_out.append(background.expand(h, w, C) * torch.ones(h, w, 1, dtype=colors.dtype))
                row_T.append(torch.ones(h, w, dtype=colors.dtype))
                continue
            g = ids_sorted[lo:hi].long()
            py, px = torch.meshgrid(torch.arange(y0, y1, dtype=colors.dtype) + 0.5,
                                    torch.arange(x0, x1, dtype=colors.dtype) + 0.5, indexing="ij")
            dx = xys[g, 0][:, None, None] - px[None]
            dy = xys[g, 1][:, None, None] - py[None]
            ca, cb, cc = (conics[g, k][:, None, None] for k in range(3))
            sigma = 0.5 * (ca * dx * dx + cc * dy * dy) + cb * dx * dy
            alpha = torch.clamp(opacity[g][:, None, None] * torch.exp(-sigma), max=alpha_cap)
            keep = (sigma >= 0) & (alpha >= ALPHA_FLOOR)
            a = torch.where(keep, alpha, torch.zeros_like(alpha))
            T_incl = torch.cumprod(1 - a, 0)
            T_excl = torch.cat([torch.ones_like(T_incl[:1]), T_incl[:-1]], 0)
            stop = keep & (T_incl <= T_STOP)
            stopped = torch.cumsum(stop.int(), 0) > 0
            contrib = keep & ~stopped
            wgt = torch.where(contrib, a * T_excl, torch.zeros_like(a))
            img = (wgt[..., None] * colors[g][:, None, None, :]).sum(0)
            # final T = T after the last contributing Gaussian
            Tf = torch.where(contrib, 1 - a, torch.ones_like(a)).prod(0)
            row_out.append(img + Tf[..., None] * background)
            row_T.append(Tf)
        out_rows.append(torch.cat(row_out, 1))
        T_rows.append(torch.cat(row_T, 1))
    return torch.cat(out_rows, 0), torch.cat(T_rows, 0)


SQRT2PI23 = 3.03352966508


def evaluate_gaussian(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, w_type=0):
    """Differentiable restatement of sg.cu:27-76 (autograd supplies the backward; the -20 edge
    substitution of sg.cu:129 is NOT reproduced -- keep |cos| < 1 in tests that compare grads)."""
    N, D = lobe_dirs.shape[:2]
    L = light_values.shape[1]
    ld = light_pts[:, None, :, :] - prim_pts[:, :, None, :]
    ld = ld / ld.norm(dim=-1, keepdim=True)
    c = (ld * lobe_dirs[:, :, None, :]).sum(-1).clamp(-1, 1)
    sig = lobe_sigmas.reshape(N, D)[:, :, None]
    ang = torch.acos(c)
    if w_type == 0:
        w = torch.exp(-0.5 * (ang / sig) ** 2) / (sig * SQRT2PI23)
    elif w_type == 1:
        w = torch.exp(-0.5 * (ang / sig) ** 2)
    elif w_type == 2:
        w = torch.exp((c - 1) / sig) / (sig * 2 * math.pi)
    else:
        w = torch.exp((c - 1) / sig)
    mask = (torch.arange(L)[None, :] < n_lights[:, None]).to(w.dtype)
    w = w * mask[:, None, :]
    return (w[..., None] * light_values[:, None, :, :]).sum(2)
