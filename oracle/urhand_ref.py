"""oracle/urhand_ref.py -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).

PyTorch restatement of URHand's per-texel-per-light UV feature loops:
    /root/reference/ca_code/models/urhand.py:419-445   Lambert + Phong^{1,16,32}, shadow-weighted sums
    /root/reference/ca_code/models/urhand.py:508-567   GGX/Schlick specular, features, physically based texture
torch autograd supplies the backward.  PINNED: tests/test_oracle_urhand.py checks outputs and every
gradient against tests/golden/urhand_golden.npz, produced by exec()-ing those reference lines
themselves (tests/golden/make_urhand_golden.py).
"""
import math

import torch
import torch.nn.functional as F

SPEC_POWERS = (1, 16, 32)  # urhand.py:277


def phong_features(p_uv, nml, cam_pos, light_pos, light_intensity, shadow_map=None, spec_powers=SPEC_POWERS):
    """p_uv, nml [B,3,S,S]; cam_pos [B,3]; light_pos [B,L,3]; light_intensity [B,L,1]; shadow_map
    [B,L,1,S,S] or None -> diff_feature_raw [B,1,S,S], spec_feature_raw [B,P,1,S,S]."""
    I = light_intensity[..., None, None]
    v_uv = F.normalize(cam_pos[..., None, None] - p_uv, dim=1)
    l_uv = F.normalize(light_pos[..., None, None] - p_uv[:, None], dim=2)
    view = -v_uv
    ref = view - 2.0 * (view * nml).sum(1, keepdim=True) * nml
    diff = (nml[:, None] * l_uv).sum(2, keepdim=True).clamp(0.0, 1.0)
    spec = (ref[:, None] * l_uv).sum(2, keepdim=True).clamp(min=0.0)
    spec = torch.stack([spec.pow(v).clamp(max=1.0) for v in spec_powers], 2)
    sh = shadow_map if shadow_map is not None else torch.ones_like(diff)
    diff_p = (diff * I * sh).sum(1)
    spec_p = (spec * I[:, :, None] * sh[:, :, None]).sum(1)
    inv = 1.0 / (I.sum(1) + 1e-6)
    return inv * diff_p, inv[:, None] * spec_p


def ggx_features(p_uv, nml, cam_pos, light_pos, light_intensity, roughness, tex_mean, shadow_map=None,
                 fresnel=0.04, spec_powers=SPEC_POWERS):
    """-> feat_p [B,1+P,S,S] (diffuse + 10x GGX^p, intensity-normalised), rgb [B,3,S,S] (before the
    global scale of urhand.py:567)."""
    I = light_intensity[..., None, None]
    V = F.normalize(cam_pos[..., None, None] - p_uv, dim=1)
    Lv = F.normalize(light_pos[..., None, None] - p_uv[:, None], dim=2)
    Hh = F.normalize((Lv + V[:, None]) / 2.0, dim=2)
    nov0 = (V * nml).sum(1, keepdim=True)
    N = nml * nov0.sign()
    nol = (N[:, None] * Lv).sum(2, keepdim=True).clamp(1e-6, 1)
    nov = (N * V).sum(1, keepdim=True)
    noh = (N[:, None] * Hh).sum(2, keepdim=True).clamp(1e-6, 1)
    voh = (V[:, None] * Hh).sum(2, keepdim=True).clamp(1e-6, 1)
    alpha = roughness * roughness
    alpha2 = alpha * alpha
    k = (alpha + 2 * roughness + 1) / 8.0
    fmi = ((-5.55473) * voh - 6.98316) * voh
    frac = (fresnel + (1 - fresnel) * torch.pow(2.0, fmi)) * alpha2[:, None]
    nom0 = noh * noh * (alpha2[:, None] - 1) + 1
    nom1 = nov * (1 - k) + k
    nom2 = nol * (1 - k[:, None]) + k[:, None]
    nom = (4 * math.pi * nom0 * nom0 * nom1[:, None] * nom2).clamp(1e-6, 4 * math.pi)
    specular = frac / nom
    diff_cos = (nml[:, None] * Lv).sum(2, keepdim=True).clamp(0.0, 1.0)
    spec = torch.stack([specular.pow(v).clamp(max=1.0) for v in spec_powers], 2)
    sh = shadow_map if shadow_map is not None else torch.ones_like(diff_cos)
    lit = (diff_cos[:, :, None] > 0) * 1.0
    diff_p = (diff_cos * I * sh).sum(1)
    spec_p = (spec * I[:, :, None] * sh[:, :, None] * lit).sum(1) * 10
    inv = 1.0 / (I.sum(1) + 1e-6)
    feat_p = inv[:, None] * torch.cat([diff_p[:, None], spec_p], 1)
    brdf = (tex_mean[:, None] / 255.0) / math.pi + specular
    cosine = (Lv * nml[:, None]).sum(2).clamp(min=0.0)
    rgb = (4 * math.pi * brdf * I * cosine[:, :, None]).mean(1)
    return feat_p[:, :, 0], rgb
