"""oracle/shade_ref.py -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).

PyTorch restatement of the RGCA shading tail, i.e. everything `PrimDecoder.forward` does after
the two transposed-conv decoders have produced f_vnocond[B,125,S,S] and f_vcond[B,4,S,S]:
    /root/reference/ca_code/models/rgca.py:505-588   (diffuse SH, Gaussian parameters, roughness,
                                                      specular normal, reflection, specular colour)
    /root/reference/ca_code/models/rgca.py:590-618   (training-only random-light diffuse term)
    /root/reference/ca_code/utils/envmap.py:284-292  (dir2uv)
    /root/reference/ca_code/utils/mipmap_sampler.py:13-69 (mipmap_grid_sample)
    /root/reference/extensions/sgutils/sgutils.py:65-98   (evaluate_gaussian; autograd restated in
                                                      oracle/torch_ref.py, explicit quirks in sg_oracle.c)
torch autograd supplies the backward.  Pinned against the reference module itself by
tests/golden/make_shade_golden.py (run in the build container where /root/reference exists);
the committed fixture is checked in tests/test_oracle_shade.py.
"""
import math

import torch
import torch.nn.functional as F

from . import torch_ref

PRIMSCALE_RANGE = (0.1, 20.0)  # rgca.py:47


def dir2uv(d):
    """envmap.py:284-292 for [...,3] directions -> [...,2] in [-1,1]."""
    u = torch.atan2(d[..., 0], d[..., 2]) / math.pi
    v = 2.0 * torch.acos(d[..., 1]) / math.pi - 1.0
    return torch.stack([u, v], -1)


def mip_sample(mips, uv, level):
    """mipmap_sampler.py:13-69: bilinear/border/align_corners=False sample of every level, then a
    lerp between the two levels around `level` (selection under no_grad).  mips: list of
    [B,3,h,w]; uv [B,N,2]; level [B,N] -> [B,N,3]."""
    q = len(mips)
    # one pyramid for the batch: expanded over it as the reference's driver does (light_decorator.py:96-100)
    mips = [m.expand(uv.shape[0], -1, -1, -1) if m.shape[0] == 1 else m for m in mips]
    with torch.no_grad():
        lam = level.clamp(min=0, max=q - 1 - 1e-6)
        d1 = lam.floor().long()
        a = lam - d1.float()
    samples = [F.grid_sample(m, uv[:, :, None, :], mode="bilinear", padding_mode="border",
                             align_corners=False)[..., 0].permute(0, 2, 1) for m in mips]  # [B,N,3] each
    st = torch.stack(samples, 0)  # [q,B,N,3]
    idx0 = d1[None, :, :, None].expand(1, -1, -1, 3)
    s0 = torch.gather(st, 0, idx0)[0]
    s1 = torch.gather(st, 0, (idx0 + 1).clamp(max=q - 1))[0]
    return torch.lerp(s0, s1, a[..., None])


def shade(f_vnocond, f_vcond, postex, tn, albedo, light_sh, campos, light_intensity=None,
          light_pos=None, n_lights=None, envmips=None, lightrot=None, light_sh_rand=None,
          n_color_sh=3, n_diff_sh=8, sg_eval=None):
    """Returns the dict of rgca.py:574-588 (+ color_rand when light_sh_rand is given).

    f_vnocond[B,3*(n_color_sh+1)^2 + ((n_diff_sh+1)^2-(n_color_sh+1)^2) + 12, S, S], f_vcond[B,4,S,S],
    postex[B,3,S,S] (uv position map), tn[B,3,S,S] (normalised uv normal map), albedo[1,N,3],
    light_sh[B,3,(n_diff_sh+1)^2], campos[B,3]; point lights (light_intensity[B,L,3],
    light_pos[B,L,3], n_lights[B]) or an environment (envmips list, lightrot[B,3,3])."""
    B = f_vnocond.shape[0]
    ncol = (n_color_sh + 1) ** 2
    nmono = (n_diff_sh + 1) ** 2 - ncol
    nd = 3 * ncol + nmono
    flat = lambda t: t.reshape(B, t.shape[1], -1).permute(0, 2, 1)  # [B,C,S,S] -> [B,N,C]
    fv, fc = flat(f_vnocond), flat(f_vcond)
    posbase, nmlbase = flat(postex), flat(tn)

    sh_col = fv[..., : 3 * ncol].reshape(B, -1, 3, ncol)
    sh_mono = fv[..., 3 * ncol: nd].reshape(B, -1, 1, nmono)
    sh_all = torch.cat([sh_col, sh_mono.expand(-1, -1, 3, -1)], -1)  # rgca.py:506-514
    geo = fv[..., nd: nd + 11]
    primpos = geo[..., 0:3] + posbase
    primqvec = F.normalize(geo[..., 3:7], dim=-1)
    primscale = F.softplus(geo[..., 7:10])
    opacity = torch.sigmoid(geo[..., 10:11])
    sigma = (torch.exp(fv[..., nd + 11]) * 0.1).clamp(min=0.01)  # rgca.py:525-527
    spec_vis = torch.sigmoid(fc[..., :1])
    spec_dnml = fc[..., 1:]
    spec_nml = F.normalize(spec_dnml + nmlbase, dim=-1)
    diff_sum = (sh_all * light_sh[:, None]).sum(-1)
    diff_color = albedo.expand(B, -1, -1) * diff_sum
    view = F.normalize(primpos - campos[:, None], dim=-1)
    ref = view - 2.0 * (view * spec_nml).sum(-1, keepdim=True) * spec_nml
    if envmips is not None:
        r = torch.einsum("bxy,bny->bnx", lightrot, ref)
        spec = mip_sample(envmips, dir2uv(r), sigma * 5).clamp(max=1.0) * spec_vis  # rgca.py:548-556
    else:
        ev = sg_eval or torch_ref.evaluate_gaussian
        spec = ev(F.normalize(ref, dim=-1), sigma, light_intensity, light_pos, primpos, n_lights.int(), 0) * spec_vis
    color = diff_color.clamp(min=0.0) + spec
    out = dict(color=color.clamp(min=0.0), opacity=opacity, primpos=primpos, primqvec=primqvec,
               primscale=primscale.clamp(*PRIMSCALE_RANGE), primscale_preclip=primscale, sigma=sigma,
               spec_vis=spec_vis, spec_nml=spec_nml, spec_dnml=spec_dnml, diff_color=diff_color,
               spec_color=spec, primnmlbase=nmlbase)
    if light_sh_rand is not None:
        out["color_rand"] = (sh_all * light_sh_rand[:, None]).sum(-1).clamp(min=0.0)  # rgca.py:613-616
    return out
