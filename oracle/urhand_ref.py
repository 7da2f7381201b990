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


def shadow_pcf(depth, Rt, postex, nml=None, focal=1000.0):
    """Restatement of the per-texel part of get_shadow_map (/root/reference/ca_code/utils/shadowmap.py:30-96) for a
    given depth image [B,h,w] (the reference obtains it from its drtk render layer): Rt [B,3,4], postex [B,3,H,W],
    nml [B,3,H,W] or None -> in_shadow [B,1,H,W].  Pinned by tests/golden/shadow_golden.npz (reference-generated)."""
    import math

    import torch.nn.functional as F

    B, _, H, W = postex.shape
    dh, dw = depth.shape[-2:]
    K = torch.eye(3)[None].repeat(B, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = focal
    K[:, 0, 2], K[:, 1, 2] = dw / 2, dh / 2
    p = postex.permute(0, 2, 3, 1).reshape(B, -1, 3)
    p_cam = p @ Rt[:, :3, :3].mT + Rt[:, :3, 3][:, None]                      # geom.py:619
    p_pix = p_cam @ K.mT
    z = p_pix[:, :, 2:]
    uv = (p_pix[..., :2] / z).view(B, H, W, 2).clone()
    d1 = z.view(B, H, W, 1).permute(0, 3, 1, 2)
    uv[..., 0] = (uv[..., 0] - dw / 2.0 - 0.5) / (dw / 2.0)                   # shadowmap.py:55-56
    uv[..., 1] = (uv[..., 1] - dh / 2.0 - 0.5) / (dh / 2.0)
    dimg = depth[:, None]
    sigma = 0.3 * ((3 - 1) * 0.5 - 1) + 0.8
    vsum, ssum = 0.0, 0.0
    for x in range(3):
        for y in range(3):
            wgt = math.exp(-((x - 1) ** 2 + (y - 1) ** 2) / (2.0 * sigma ** 2))
            g = uv.clone()
            g[..., 0] += 2.0 / dw * (x - 1)
            g[..., 1] += 2.0 / dh * (y - 1)
            d = F.grid_sample(dimg, g, mode="nearest", align_corners=False)
            w = F.grid_sample((dimg > 0.0).float(), g, mode="nearest", align_corners=False)
            valid = wgt * (w > 1e-4).float()
            vsum = vsum + valid
            ssum = ssum + valid * (d1 - d / (w + 1e-8)).clamp(min=0)
    out = ssum / (vsum + 1e-6)
    if nml is not None:
        vdir = F.normalize(Rt[:, :, -1][..., None, None] - postex, dim=1)
        bc = torch.sigmoid(10 * (nml * vdir).sum(1, keepdim=True))
        out = bc * out + (1.0 - bc) * 1e3
    return out
