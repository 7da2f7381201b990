"""Seeded synthetic inputs shared by the parity tests (SURVEY.md section 8d recipes, scaled)."""
import math

import torch


def look_at_viewmat(eye, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
    """World->camera [3,4] with +z forward (the convention gsplat's viewmat expects)."""
    eye = torch.tensor(eye, dtype=torch.float32)
    fwd = torch.tensor(target, dtype=torch.float32) - eye
    fwd = fwd / fwd.norm()
    upv = torch.tensor(up, dtype=torch.float32)
    right = torch.linalg.cross(upv, fwd)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    R = torch.stack([right, down, fwd], 0)
    return torch.cat([R, (-R @ eye)[:, None]], 1).contiguous()


def head_scene(N, H, W, seed=1234, cam_angle=0.3, focal=None, radii=(90.0, 120.0, 100.0),
               scale_range=(0.3, 3.0), cam_dist=700.0, max_opacity=1.0):
    """Gaussians uniform in an ellipsoid (mm), ring camera looking at the origin."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(N, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    r = torch.rand(N, 1, generator=g) ** (1.0 / 3.0)
    means = d * r * torch.tensor(radii)
    quats = torch.randn(N, 4, generator=g)
    quats = quats / quats.norm(dim=-1, keepdim=True)
    lo, hi = math.log(scale_range[0]), math.log(scale_range[1])
    scales = torch.exp(lo + (hi - lo) * torch.rand(N, 3, generator=g))
    opacity = torch.sigmoid(1.5 * torch.randn(N, 1, generator=g)) * max_opacity
    colors = torch.rand(N, 3, generator=g)
    if focal is None:
        focal = 3000.0 * W / 1334.0
    eye = (cam_dist * math.sin(cam_angle), 0.0, -cam_dist * math.cos(cam_angle))
    viewmat = look_at_viewmat(eye)
    return dict(means=means, scales=scales, quats=quats, opacity=opacity, colors=colors,
                viewmat=viewmat, fx=focal, fy=focal, cx=W / 2.0, cy=H / 2.0, H=H, W=W)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def icosphere(subdiv=2, radius=1.0):
    """Closed genus-0 triangle mesh (20 * 4^subdiv faces): vertices [V,3] float32, faces [F,3] int64."""
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1),
         (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]
    v = [torch.tensor(p, dtype=torch.float64) / torch.tensor(p, dtype=torch.float64).norm() for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / m.norm())
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (torch.stack(v) * radius).float(), torch.tensor(f, dtype=torch.int64)


def worst_set(a, b, tol):
    """Smallest set of Gaussians (indices into the flattened [B*N] axis) whose removal brings rel-L2(a, b) under `tol`.
    a, b: [B, C, N].  Returns (indices, rel-L2 over everything, rel-L2 over the rest)."""
    a, b = a.double().cpu(), b.double().cpu()
    err = (a - b).pow(2).sum(1).flatten()        # [B*N]
    ref = b.pow(2).sum(1).flatten()
    e_tot, r_tot = float(err.sum()), float(ref.sum())
    order = err.argsort(descending=True)
    ce = torch.cumsum(err[order], 0)
    cr = torch.cumsum(ref[order], 0)
    ok = (e_tot - ce) <= tol * tol * (r_tot - cr).clamp(min=1e-300)
    k = 0 if e_tot <= tol * tol * r_tot else int(torch.nonzero(ok)[0]) + 1
    rest = (max(e_tot - float(ce[k - 1]), 0.0) / max(r_tot - float(cr[k - 1]), 1e-300)) ** 0.5 if k else (e_tot / r_tot) ** 0.5
    return order[:k], (e_tot / r_tot) ** 0.5, rest
