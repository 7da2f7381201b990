"""Fused RGCA shading tail (host side) on top of the C ABI (gol_shade_fwd / gol_shade_bwd).

Drop-in for the part of `PrimDecoder.forward` that follows the two transposed-conv decoders
(/root/reference/ca_code/models/rgca.py:505-588, training extra :590-618): same inputs, same keys and
shapes in the returned dict, gradients to f_vnocond, f_vcond, the uv position / normal maps and the
albedo parameter.  The decoder outputs are consumed in their native NCHW layout (no permute copies).
"""
import ctypes
import os
import weakref

import torch

from . import _lib, views as _views
from ._lib import stream_ptr

MAX_MIPS = 8
PRIMSCALE_RANGE = (0.1, 20.0)  # rgca.py:47

_fp = ctypes.c_void_p


class ShadeIn(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("n_color_coef", ctypes.c_int32),
                ("n_mono_coef", ctypes.c_int32), ("f_vnocond", _fp), ("f_vcond", _fp), ("postex", _fp),
                ("tn", _fp), ("albedo", _fp), ("light_sh", _fp), ("light_sh_rand", _fp), ("campos", _fp),
                ("L", ctypes.c_int32), ("light_intensity", _fp), ("light_pos", _fp), ("n_lights", _fp),
                ("n_mips", ctypes.c_int32), ("mips", _fp * MAX_MIPS), ("mip_h", ctypes.c_int32 * MAX_MIPS),
                ("mip_w", ctypes.c_int32 * MAX_MIPS), ("lightrot", _fp), ("mips_packed", _fp * MAX_MIPS),
                ("primscale_min", ctypes.c_float),
                ("primscale_max", ctypes.c_float), ("mips_shared", ctypes.c_int32), ("mips_scale", ctypes.c_float)]


OUT_FIELDS = [("color", 3), ("opacity", 1), ("primpos", 3), ("primqvec", 4), ("primscale", 3),
              ("primscale_preclip", 3), ("sigma", 1), ("spec_vis", 1), ("spec_nml", 3), ("spec_dnml", 3),
              ("diff_color", 3), ("spec_color", 3), ("primnmlbase", 3), ("color_rand", 3), ("diff_sum", 3),
              ("env_saved", 9)]
GRAD_FIELDS = [n for n, _ in OUT_FIELDS if n not in ("diff_sum", "env_saved")]


class ShadeOut(ctypes.Structure):
    _fields_ = [(n, _fp) for n, _ in OUT_FIELDS]


class ShadeOutGrad(ctypes.Structure):
    _fields_ = [(n, _fp) for n in GRAD_FIELDS]


class ShadeInGrad(ctypes.Structure):
    _fields_ = [("f_vnocond", _fp), ("f_vcond", _fp), ("postex", _fp), ("tn", _fp), ("albedo_per_view", _fp),
                ("albedo", _fp)]


def _p(t, dtype=torch.float32):
    return _lib.ptr(t, dtype).value if t is not None else None


_PACKED = {}   # (data_ptr, shape, _version) -> (the level tensor, packed copy): the pyramid is static across frames


def invalidate_envmap_cache():
    """Forget every packed env-map level (see pack_envmap: writes the version counter does not see)."""
    _PACKED.clear()


def pack_envmap(mips, refresh=False):
    """[B,3,h,w] mip levels -> [B,h,w,16] footprint records (the four taps of the bilinear footprint whose top-left texel
    is (y, x), 64 bytes: a lookup fetches one HBM sector instead of ~2.5).  The reference builds the
    SG-prefiltered pyramid once per environment (light_decorator.py:18-164) and only rotates it per frame (`lightrot`),
    so the packed copy of a level is cached on the tensor's memory and in-place version counter: a level that was
    neither replaced nor written since the last call is not packed again (4 launches per step in rounds 1-2).  The cache
    keeps the source tensor alive (at most 16 levels), so its address cannot be handed to another tensor meanwhile.
    CAVEAT: writes that do not bump the version counter are NOT seen -- `mip.data.copy_(...)` / in-place ops through `.data`,
    and kernels that write the pyramid through a raw pointer.  After such a write pass refresh=True once (re-packs these
    levels and replaces their cache entries), call invalidate_envmap_cache(), or set GOLIATH_ENVMAP_CACHE=0 (packs on every
    call: 4 small launches per step)."""
    refresh = refresh or os.environ.get("GOLIATH_ENVMAP_CACHE", "1") == "0"
    out = []
    for m in mips:
        key = (m.data_ptr(), tuple(m.shape), m._version, m.device.index)
        hit = None if refresh else _PACKED.get(key)
        if hit is not None:
            if hit[2] is not None:          # packed on another stream and possibly still in flight: order after it
                if hit[2].query():
                    _PACKED[key] = (hit[0], hit[1], None)
                else:
                    torch.cuda.current_stream(m.device).wait_event(hit[2])
            _PACKED[key] = _PACKED.pop(key)    # most recently used last: eviction below drops the least recently used
            out.append(hit[1])
            continue
        B, _, h, w = m.shape
        p = torch.empty(B, h, w, 16, device=m.device)
        _lib.call("gol_envmap_pack", _lib.c_int(B), _lib.c_int(h), _lib.c_int(w), _lib.fptr(m), _lib.fptr(p),
                  stream_ptr())
        while len(_PACKED) >= 16:
            _PACKED.pop(next(iter(_PACKED)))   # least recently used entry
        ev = torch.cuda.Event()
        ev.record()
        _PACKED[key] = (m, p, ev)
        out.append(p)
    return out


def _make_in(f_vnocond, f_vcond, postex, tn, albedo, light_sh, light_sh_rand, campos, light_intensity,
             light_pos, n_lights, mips, lightrot, ncol, nmono, packed=None, mips_scale=1.0):
    B, C = f_vnocond.shape[:2]
    shared = bool(mips) and B > 1 and all(m.shape[0] == 1 for m in mips)
    N = f_vnocond[0, 0].numel()
    s = ShadeIn()
    s.B, s.N, s.n_color_coef, s.n_mono_coef = B, N, ncol, nmono
    s.f_vnocond, s.f_vcond, s.postex, s.tn = _p(f_vnocond), _p(f_vcond), _p(postex), _p(tn)
    s.albedo, s.light_sh, s.light_sh_rand, s.campos = _p(albedo), _p(light_sh), _p(light_sh_rand), _p(campos)
    if mips:
        s.n_mips = len(mips)
        for i, m in enumerate(mips):
            s.mips[i] = _p(m)
            s.mip_h[i], s.mip_w[i] = m.shape[-2], m.shape[-1]
            if packed:
                s.mips_packed[i] = _p(packed[i])
        s.lightrot = _p(lightrot)
        s.mips_shared = int(shared)
        s.mips_scale = float(mips_scale)
        if not shared and any(m.shape[0] != B for m in mips):
            raise ValueError(f"env-map levels must be [B={B},3,h,w] (one pyramid per view) or [1,3,h,w] (one for all views)")
    else:
        s.n_mips = 0
        s.L = light_intensity.shape[1]
        s.light_intensity, s.light_pos = _p(light_intensity), _p(light_pos)
        s.n_lights = _p(n_lights, torch.int32)
    s.primscale_min, s.primscale_max = PRIMSCALE_RANGE
    return s


class _Shade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f_vnocond, f_vcond, postex, tn, albedo, light_sh, light_sh_rand, campos, light_intensity,
                light_pos, n_lights, lightrot, ncol, nmono, view_set, mips_scale, *mips):
        B = f_vnocond.shape[0]
        N = f_vnocond[0, 0].numel()
        dev = f_vnocond.device
        rand = light_sh_rand is not None
        need_grad = any(ctx.needs_input_grad[:5])
        outs = {n: torch.empty(B, N, k, device=dev) for n, k in OUT_FIELDS
                if (rand or n != "color_rand") and (n != "env_saved" or (mips and need_grad))}
        with _lib.device_guard(dev):
            # GOLIATH_ENVMAP_RECORDS=0: gather from the planar [.,3,h,w] levels as given (no footprint records)
            packed = pack_envmap(mips) if mips and os.environ.get("GOLIATH_ENVMAP_RECORDS", "1") != "0" else None
        sin = _make_in(f_vnocond, f_vcond, postex, tn, albedo, light_sh, light_sh_rand, campos,
                       light_intensity, light_pos, n_lights, list(mips), lightrot, ncol, nmono, packed, mips_scale)
        sout = ShadeOut()
        for n, t in outs.items():
            setattr(sout, n, _p(t))
        records = pack = None
        if view_set is not None:
            # the projection of these Gaussians onto the views' cameras runs as the kernel's epilogue (views.py)
            records = torch.empty(B, N, _views.SPLAT_RECORD, device=dev)
            pack = torch.empty(_views.PACK_FLOATS, B * N, device=dev)
            pj = _views.proj_struct(view_set, records, pack)
            with _lib.device_guard(dev):
                _lib.call("gol_shade_project_fwd", ctypes.byref(sin), ctypes.byref(sout), ctypes.byref(pj), stream_ptr())
        else:
            with _lib.device_guard(dev):
                _lib.call("gol_shade_fwd", ctypes.byref(sin), ctypes.byref(sout), stream_ptr())
        ctx.view_set = view_set
        ctx.cfg = (ncol, nmono, len(mips), rand)
        ctx.mips_scale = mips_scale
        ctx.packed = packed  # not an input/output of the node: plain attribute
        ctx.save_for_backward(f_vnocond, f_vcond, postex, tn, albedo, light_sh, light_sh_rand, campos,
                              light_intensity, light_pos, n_lights, lightrot, outs["diff_sum"],
                              outs.get("color_rand"), outs.get("env_saved"), pack, *mips)
        ctx.set_materialize_grads(False)
        names = [n for n in GRAD_FIELDS if n in outs]
        ctx.names = names
        if view_set is not None:
            ctx.mark_non_differentiable(pack)
            return tuple(outs[n] for n in names) + (records, pack)
        return tuple(outs[n] for n in names)

    @staticmethod
    def backward(ctx, *grads):
        ncol, nmono, n_mips, rand = ctx.cfg
        sv = ctx.saved_tensors
        (f_vnocond, f_vcond, postex, tn, albedo, light_sh, light_sh_rand, campos, light_intensity, light_pos,
         n_lights, lightrot, diff_sum, color_rand, env_saved, pack) = sv[:16]
        mips = list(sv[16:])
        B = f_vnocond.shape[0]
        N = f_vnocond[0, 0].numel()
        dev = f_vnocond.device
        sin = _make_in(f_vnocond, f_vcond, postex, tn, albedo, light_sh, light_sh_rand, campos,
                       light_intensity, light_pos, n_lights, mips, lightrot, ncol, nmono, ctx.packed, ctx.mips_scale)
        saved = ShadeOut()
        saved.diff_sum = _p(diff_sum)
        saved.color_rand = _p(color_rand)
        saved.env_saved = _p(env_saved)
        up = ShadeOutGrad()
        keep = []
        for n, g in zip(ctx.names, grads):
            if g is not None:
                g = g.to(torch.float32).contiguous()
                keep.append(g)
                setattr(up, n, _p(g))
        g_vn, g_vc = torch.empty_like(f_vnocond), torch.empty_like(f_vcond)
        g_pt, g_tn = torch.empty_like(postex), torch.empty_like(tn)
        g_alb = torch.empty(B, N, 3, device=dev)
        gin = ShadeInGrad()
        gin.f_vnocond, gin.f_vcond, gin.postex, gin.tn = _p(g_vn), _p(g_vc), _p(g_pt), _p(g_tn)
        gin.albedo_per_view = _p(g_alb)
        # the shared albedo's gradient = the sum of the per-view ones, by a second small kernel of the same call
        g_albedo = torch.empty(N, 3, device=dev) if ctx.needs_input_grad[4] else None
        gin.albedo = _p(g_albedo)
        g_rec = grads[len(ctx.names)] if ctx.view_set is not None else None
        with _lib.device_guard(dev):
            if g_rec is not None:
                # the raster backward's gradient records go through the projection's vjp in the kernel's prologue
                g_rec = g_rec.to(torch.float32).contiguous()
                pj = _views.proj_struct(ctx.view_set, g_rec, pack)   # (the records pointer is not read by the backward)
                _lib.call("gol_shade_project_bwd", ctypes.byref(sin), ctypes.byref(saved), ctypes.byref(up),
                          ctypes.byref(pj), _lib.fptr(g_rec), _lib.c_int(1), ctypes.byref(gin), stream_ptr())
            else:
                _lib.call("gol_shade_bwd", ctypes.byref(sin), ctypes.byref(saved), ctypes.byref(up), ctypes.byref(gin),
                          stream_ptr())
        if g_albedo is not None:
            g_albedo = g_albedo.reshape(albedo.shape)
        return (g_vn, g_vc, g_pt, g_tn, g_albedo) + (None,) * (11 + n_mips)


_ROT_CHECKED = {}


def _check_rotation(lightrot):
    """gol_shade_in.lightrot must hold rotations (the kernel forms the polar angle of lightrot x reflection as
    atan2(|r_xz|, r_y), which equals the reference's acos(r_y), envmap.py:289, only for a unit vector).  Checked once per
    tensor version (one small device reduction + sync; the relight driver builds a new lightrot per frame on the HOST and
    copies it over, light_decorator.py:108-120, so GOLIATH_CHECK_LIGHTROT=0 skips it in a sync-free loop)."""
    if os.environ.get("GOLIATH_CHECK_LIGHTROT", "1") == "0" or torch.cuda.is_current_stream_capturing():
        return
    key = (lightrot.data_ptr(), lightrot._version, tuple(lightrot.shape))
    seen = _ROT_CHECKED.get(key)
    if seen is not None and seen() is lightrot:     # the SAME tensor object (a freed tensor's address can come back)
        return
    err = float((lightrot @ lightrot.transpose(-1, -2) - torch.eye(3, device=lightrot.device)).abs().max())
    if not err < 1e-3:
        raise ValueError(f"lightrot is not a rotation (max |R R^T - I| = {err:.3g}): the env-map lookup needs a unit direction")
    while len(_ROT_CHECKED) >= 64:
        _ROT_CHECKED.pop(next(iter(_ROT_CHECKED)))
    _ROT_CHECKED[key] = weakref.ref(lightrot)


def shading_tail(f_vnocond, f_vcond, postex, tn, albedo, headrel_light_sh, headrel_campos,
                 light_intensity=None, headrel_light_pos=None, n_lights=None, preconv_envmap=None,
                 lightrot=None, light_sh_rand=None, n_color_sh=3, n_diff_sh=8, views=None):
    """f_vnocond[B,125,S,S], f_vcond[B,4,S,S] (decoder outputs, NCHW), postex[B,3,S,S]
    (geo_fn.to_uv(geom)), tn[B,3,S,S] (normalised uv normal map), albedo[1,N,3] -> dict with the keys
    and [B,N,k] shapes of rgca.py:574-588 (+ "color_rand" when light_sh_rand[B,3,81] is given).
    views (a goliath_amd.views.ViewSet): the cameras the result will be rendered with -- the kernel then also projects the
    Gaussians (preds["projected"]), and render_gs.render_batch called with the same K / Rt starts at the tile count."""
    ncol = (n_color_sh + 1) ** 2
    return shading_tail_coefs(f_vnocond, f_vcond, postex, tn, albedo, headrel_light_sh, headrel_campos, ncol,
                              (n_diff_sh + 1) ** 2 - ncol, light_intensity, headrel_light_pos, n_lights,
                              preconv_envmap, lightrot, light_sh_rand, views)


def shading_tail_coefs(f_vnocond, f_vcond, postex, tn, albedo, headrel_light_sh, headrel_campos, ncol, nmono,
                       light_intensity=None, headrel_light_pos=None, n_lights=None, preconv_envmap=None,
                       lightrot=None, light_sh_rand=None, views=None):
    """shading_tail with the SH layout given as coefficient counts: f_vnocond has 3*ncol colour-SH
    channels, nmono monochrome ones and the 12 Gaussian-parameter channels; light_sh is [B,3,ncol+nmono]."""
    if not f_vnocond.is_cuda:
        raise _lib.GoliathHipError("shading_tail needs CUDA(HIP) tensors; there is no CPU path")
    B, C = f_vnocond.shape[:2]
    if C != 3 * ncol + nmono + 12:
        raise ValueError(f"f_vnocond has {C} channels, expected {3 * ncol + nmono + 12}")
    c = lambda t: None if t is None else t.to(torch.float32).contiguous()
    N = f_vnocond[0, 0].numel()
    mips, mips_scale = [], 1.0
    if preconv_envmap is not None:
        mips = list(preconv_envmap) if isinstance(preconv_envmap, (list, tuple)) else [preconv_envmap]
        if len(mips) > MAX_MIPS:
            raise ValueError("too many mip levels")
        # ONE pyramid for the whole batch -- [1,3,h,w] levels, or the stride-0 batch views `expand(B, ...)` gives (what
        # dropin.patch_light_decorator makes of EnvSpinDecorator.mipmap, light_decorator.py:96-100): the kernel reads the
        # single map for every view (gol_shade_in.mips_shared) instead of B materialised copies
        if all(m.dim() == 4 and (m.shape[0] == 1 or m.stride(0) == 0) for m in mips):
            if any(m.shape[0] not in (1, B) for m in mips):   # an expand(B', ...) view of another batch size
                raise ValueError(f"env-map levels are expanded over {[m.shape[0] for m in mips]} views, the batch has {B}")
            # levels that come from dropin._shared_mipmap carry the UNSCALED registered buffer and the frame's scale: the
            # packed records are cached on the buffer (stable address and version across frames), the scale goes to the kernel
            base = [getattr(m, "_gol_base", None) for m in mips]
            if all(b is not None for b in base) and len({float(m._gol_scale) for m in mips}) == 1:
                mips_scale, mips = float(mips[0]._gol_scale), base
            else:
                mips = [m[:1] for m in mips]
        mips = [c(m) for m in mips]
        lightrot = c(lightrot)
        if lightrot is None or lightrot.shape[-2:] != (3, 3):
            raise ValueError("preconv_envmap needs lightrot [B,3,3]")
        _check_rotation(lightrot)
        li = lp = nl = None
    else:
        li, lp = c(light_intensity.expand(-1, -1, 3)), c(headrel_light_pos)
        nl = n_lights.to(torch.int32).contiguous()
        lightrot = None
    if views is not None:   # before any launch: the kernel indexes the cameras by view
        if views.K.shape[0] != B or views.Rt.shape[0] != B:
            raise ValueError(f"the ViewSet holds {views.K.shape[0]} cameras, the batch {B} views")
        if views.height <= 0 or views.width <= 0:
            raise ValueError(f"bad image size {views.height} x {views.width}")
    outs = _Shade.apply(c(f_vnocond), c(f_vcond), c(postex), c(tn), c(albedo).reshape(N, 3),
                        c(headrel_light_sh), c(light_sh_rand), c(headrel_campos), li, lp, nl, lightrot,
                        ncol, nmono, views, mips_scale, *mips)
    names = [n for n in GRAD_FIELDS if light_sh_rand is not None or n != "color_rand"]
    preds = dict(zip(names, outs))
    if views is not None:
        preds["projected"] = _views.Projected(views, outs[len(names)], outs[len(names) + 1], preds)
    preds["sigma"] = preds["sigma"][..., 0]  # [B,N] like rgca.py:526
    return preds
