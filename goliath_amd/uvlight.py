"""URHand UV light-feature loops (host side) on top of the C ABI.

Drop-in for the two per-texel-per-light blocks of `ConvTeacherDecoder.forward`
(/root/reference/ca_code/models/urhand.py:419-445 and :508-567): same tensor shapes in and out,
gradients to the position map, the normal map, roughness and the mean texture.
"""
import ctypes

import torch

from . import _lib
from ._lib import stream_ptr

SPEC_POWERS = (1, 16, 32)  # urhand.py:277
MAX_POW = 4
_fp = ctypes.c_void_p


class UvLightIn(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("L", ctypes.c_int32), ("HW", ctypes.c_int32), ("n_pow", ctypes.c_int32),
                ("pow", ctypes.c_float * MAX_POW), ("fresnel", ctypes.c_float), ("p_uv", _fp), ("nml", _fp),
                ("cam_pos", _fp), ("light_pos", _fp), ("light_intensity", _fp), ("shadow_map", _fp),
                ("roughness", _fp), ("tex_mean", _fp)]


def _p(t):
    return _lib.fptr(t).value if t is not None else None


def _c(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _mk(p_uv, nml, cam_pos, light_pos, light_intensity, shadow_map, powers, fresnel=0.0, roughness=None, tex_mean=None):
    if not p_uv.is_cuda:
        raise _lib.GoliathHipError("uvlight needs CUDA(HIP) tensors; there is no CPU path")
    if len(powers) > MAX_POW:
        raise ValueError("at most 4 specular powers")
    B, _, H, W = p_uv.shape
    s = UvLightIn()
    s.B, s.L, s.HW, s.n_pow = B, light_pos.shape[1], H * W, len(powers)
    for i, v in enumerate(powers):
        s.pow[i] = float(v)
    s.fresnel = float(fresnel)
    s.p_uv, s.nml, s.cam_pos, s.light_pos = _p(p_uv), _p(nml), _p(cam_pos), _p(light_pos)
    s.light_intensity, s.shadow_map, s.roughness, s.tex_mean = _p(light_intensity), _p(shadow_map), _p(roughness), _p(tex_mean)
    return s


class _Phong(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p_uv, nml, cam_pos, light_pos, light_intensity, shadow_map, powers):
        B, _, H, W = p_uv.shape
        args = [_c(t) for t in (p_uv, nml, cam_pos, light_pos, light_intensity.reshape(B, -1), shadow_map)]
        s = _mk(*args, powers)
        diff = torch.empty(B, 1, H, W, device=p_uv.device)
        spec = torch.empty(B, len(powers), 1, H, W, device=p_uv.device)
        with _lib.device_guard(p_uv.device):
            _lib.call("gol_uvlight_phong_fwd", ctypes.byref(s), _lib.fptr(diff), _lib.fptr(spec), stream_ptr())
        ctx.args, ctx.powers = args, powers
        return diff, spec

    @staticmethod
    def backward(ctx, u_diff, u_spec):
        p_uv = ctx.args[0]
        s = _mk(*ctx.args, ctx.powers)
        g_p, g_n = torch.empty_like(p_uv), torch.empty_like(p_uv)
        u_diff, u_spec = _c(u_diff), _c(u_spec)  # bound to locals: must outlive the launch
        with _lib.device_guard(p_uv.device):
            _lib.call("gol_uvlight_phong_bwd", ctypes.byref(s), _lib.fptr(u_diff), _lib.fptr(u_spec),
                      _lib.fptr(g_p), _lib.fptr(g_n), stream_ptr())
        return g_p, g_n, None, None, None, None, None


class _Ggx(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p_uv, nml, roughness, tex_mean, cam_pos, light_pos, light_intensity, shadow_map, powers, fresnel):
        B, _, H, W = p_uv.shape
        args = [_c(t) for t in (p_uv, nml, cam_pos, light_pos, light_intensity.reshape(B, -1), shadow_map)]
        extra = dict(fresnel=fresnel, roughness=_c(roughness), tex_mean=_c(tex_mean))
        s = _mk(*args, powers, **extra)
        feat = torch.empty(B, 1 + len(powers), H, W, device=p_uv.device)
        rgb = torch.empty(B, 3, H, W, device=p_uv.device)
        with _lib.device_guard(p_uv.device):
            _lib.call("gol_uvlight_ggx_fwd", ctypes.byref(s), _lib.fptr(feat), _lib.fptr(rgb), stream_ptr())
        ctx.args, ctx.powers, ctx.extra = args, powers, extra
        return feat, rgb

    @staticmethod
    def backward(ctx, u_feat, u_rgb):
        p_uv = ctx.args[0]
        s = _mk(*ctx.args, ctx.powers, **ctx.extra)
        g_p, g_n = torch.empty_like(p_uv), torch.empty_like(p_uv)
        g_r, g_t = torch.empty_like(ctx.extra["roughness"]), torch.empty_like(ctx.extra["tex_mean"])
        u_feat, u_rgb = _c(u_feat), _c(u_rgb)  # bound to locals: must outlive the launch
        with _lib.device_guard(p_uv.device):
            _lib.call("gol_uvlight_ggx_bwd", ctypes.byref(s), _lib.fptr(u_feat), _lib.fptr(u_rgb),
                      _lib.fptr(g_p), _lib.fptr(g_n), _lib.fptr(g_r), _lib.fptr(g_t), stream_ptr())
        return g_p, g_n, g_r, g_t, None, None, None, None, None, None


def phong_features(p_uv, nml, cam_pos, light_pos, light_intensity, shadow_map=None, spec_powers=SPEC_POWERS):
    """urhand.py:419-445.  p_uv, nml [B,3,S,S]; cam_pos [B,3]; light_pos [B,L,3]; light_intensity [B,L,1];
    shadow_map [B,L,1,S,S] or None -> (diff_feature_raw [B,1,S,S], spec_feature_raw [B,P,1,S,S])."""
    return _Phong.apply(p_uv, nml, cam_pos, light_pos, light_intensity, shadow_map, tuple(spec_powers))


def ggx_features(p_uv, nml, cam_pos, light_pos, light_intensity, roughness, tex_mean, shadow_map=None,
                 fresnel=0.04, spec_powers=SPEC_POWERS):
    """urhand.py:508-567 -> (feat_p [B,1+P,S,S], rgb [B,3,S,S] before the global scale of :567)."""
    return _Ggx.apply(p_uv, nml, roughness, tex_mean, cam_pos, light_pos, light_intensity, shadow_map,
                      tuple(spec_powers), float(fresnel))
