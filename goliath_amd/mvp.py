"""Mixture-of-Volumetric-Primitives ray marching (host side) on top of the C ABI.

Drop-in for the reference operators (interfaces only; the work happens in libgoliath_hip.so):
  mvpraymarch(...)                      <- extensions/mvpraymarch/mvpraymarch.py:313-418
  mvpraymarchlib.compute_aabb / raymarch_forward / raymarch_backward
                                        <- extensions/mvpraymarch/mvpraymarch.cpp:145-409 (pybind module)
  compute_raydirs(...), utilslib.compute_raydirs_forward
                                        <- extensions/utils/utils.py:26-54, utils.cpp:46-82
  Raymarcher                            <- ca_code/utils/render_raymarcher.py:18-71
Supported configuration = the one every model in the reference uses: algo 0, usebvh="fixedorder",
channels-last template, additive accumulation -- plus algo 1 (a warp field per box, the second case of the reference's
in-tree gradcheck, mvpraymarch.py:790-803).  The options the
reference accepts but its kernels ignore (sortprims, maxhitboxes, synchitboxes, accum, termthresh,
griddim, blocksize) are accepted and ignored here too.  Everything runs on the CURRENT stream with a
device guard (the reference uses stream 0).
"""
import torch

from . import _lib
from ._lib import c_float, c_int, fptr, stream_ptr


def _need_gpu(t, name):
    if not t.is_cuda:
        raise _lib.GoliathHipError(f"{name} must be a CUDA tensor (there is no CPU path)")  # CHECK_INPUT


def _gpu_contig(t, name):
    _need_gpu(t, name)
    return t.detach().to(torch.float32).contiguous()


def _dims(template):
    if template.dim() != 6 or template.size(-1) != 4:
        raise RuntimeError("template must be channels-last [N, K, TD, TH, TW, 4]")
    return template.shape[1], template.shape[2], template.shape[3], template.shape[4]


def _check_algo(algo, warp):
    """algo 0 = no warp field, algo 1 = one warp field per box (mvpraymarch_kernel.cu:92-104); algo 2 (no softplus'd
    template, experimental in the reference) is not built."""
    if algo not in (0, 1) or (algo == 1) != (warp is not None):
        raise NotImplementedError(f"algo={algo} with warp {'given' if warp is not None else 'absent'}: algo 0 takes no warp "
                                  "field, algo 1 needs one; other algorithms are not implemented")


def _warp_dims(warp, template):
    if warp.dim() != 6 or warp.size(-1) != 3 or warp.shape[:2] != template.shape[:2]:
        raise RuntimeError("warp must be channels-last [N, K, WD, WH, WW, 3] with the template's N, K")
    return warp.shape[2], warp.shape[3], warp.shape[4]


class _MvpLib:
    """Same entry points as the reference's compiled `mvpraymarchlib`."""

    @staticmethod
    def compute_morton(*_a, **_k):
        raise NotImplementedError("Morton/LBVH build is a dead path in the reference (SURVEY 2.3): use usebvh='fixedorder'")

    build_tree = compute_morton

    @staticmethod
    def compute_aabb(primpos, primrot, primscale, sortedobjid, nodechildren, nodeparent, nodeaabb, algo=0):
        N, K = primpos.shape[:2]
        _need_gpu(primpos, "primpos")
        with _lib.device_guard(primpos.device):
            _lib.call("gol_mvp_aabb", c_int(N), c_int(K), fptr(primpos, "primpos"), fptr(primrot, "primrot"),
                      fptr(primscale, "primscale"), fptr(nodeaabb, "nodeaabb"), stream_ptr())
        return []

    @staticmethod
    def raymarch_forward(raypos, raydir, stepsize, tminmax, sortedobjid, nodechildren, nodeaabb, primpos, primrot,
                         primscale, template, warp, rayrgba, raysat, rayterm, shadow, algo=0, sortboxes=False,
                         maxhitboxes=512, synchitboxes=True, chlast=True, fadescale=8.0, fadeexp=8.0, accum=0,
                         termthresh=0.0, griddim=3, blocksizex=8, blocksizey=16):
        _check_algo(algo, warp)
        if not chlast:
            raise NotImplementedError("only channels-last templates (chlast=True, the reference default)")
        if nodeaabb is None:
            raise NotImplementedError("usebvh=False is not implemented; use 'fixedorder'")
        N, H, W = raypos.shape[:3]
        K, TD, TH, TW = _dims(template)
        _need_gpu(raypos, "raypos")
        if warp is not None:
            WD, WH, WW = _warp_dims(warp, template)
            with _lib.device_guard(raypos.device):
                _lib.call("gol_mvp_march_warp_fwd", c_int(N), c_int(H), c_int(W), c_int(K), fptr(raypos, "raypos"),
                          fptr(raydir, "raydir"), c_float(stepsize), fptr(tminmax, "tminmax"), fptr(nodeaabb, "nodeaabb"),
                          fptr(primpos, "primpos"), fptr(primrot, "primrot"), fptr(primscale, "primscale"),
                          fptr(template, "template"), c_int(TD), c_int(TH), c_int(TW), fptr(warp, "warp"), c_int(WD),
                          c_int(WH), c_int(WW), c_float(fadescale), c_float(fadeexp), fptr(rayrgba, "rayrgba"),
                          fptr(raysat, "raysat"), fptr(shadow, "shadow"), stream_ptr())
            return []
        with _lib.device_guard(raypos.device):
            _lib.call("gol_mvp_march_fwd", c_int(N), c_int(H), c_int(W), c_int(K), fptr(raypos, "raypos"),
                      fptr(raydir, "raydir"), c_float(stepsize), fptr(tminmax, "tminmax"), fptr(nodeaabb, "nodeaabb"),
                      fptr(primpos, "primpos"), fptr(primrot, "primrot"), fptr(primscale, "primscale"),
                      fptr(template, "template"), c_int(TD), c_int(TH), c_int(TW), c_float(fadescale),
                      c_float(fadeexp), fptr(rayrgba, "rayrgba"), fptr(raysat, "raysat"), fptr(shadow, "shadow"),
                      stream_ptr())
        return []

    @staticmethod
    def raymarch_backward(raypos, raydir, stepsize, tminmax, sortedobjid, nodechildren, nodeaabb, primpos,
                          grad_primpos, primrot, grad_primrot, primscale, grad_primscale, template, grad_template,
                          warp, grad_warp, rayrgba, grad_rayrgba, raysat, rayterm, algo=0, sortboxes=False,
                          maxhitboxes=512, synchitboxes=True, chlast=True, fadescale=8.0, fadeexp=8.0, accum=0,
                          termthresh=0.0, griddim=3, blocksizex=8, blocksizey=16):
        _check_algo(algo, warp)
        if not chlast:
            raise NotImplementedError("only channels-last templates (chlast=True, the reference default)")
        N, H, W = raypos.shape[:3]
        K, TD, TH, TW = _dims(template)
        _need_gpu(raypos, "raypos")
        if warp is not None:
            WD, WH, WW = _warp_dims(warp, template)
            if grad_warp is None or grad_warp.shape != warp.shape:
                raise RuntimeError("grad_warp must have the shape of warp")
            with _lib.device_guard(raypos.device):
                _lib.call("gol_mvp_march_warp_bwd", c_int(N), c_int(H), c_int(W), c_int(K), fptr(raypos), fptr(raydir),
                          c_float(stepsize), fptr(tminmax), fptr(nodeaabb), fptr(primpos), fptr(primrot),
                          fptr(primscale), fptr(template), c_int(TD), c_int(TH), c_int(TW), fptr(warp, "warp"),
                          c_int(WD), c_int(WH), c_int(WW), c_float(fadescale), c_float(fadeexp), fptr(raysat, "raysat"),
                          fptr(grad_rayrgba, "grad_rayrgba"), fptr(grad_primpos), fptr(grad_primrot),
                          fptr(grad_primscale), fptr(grad_template), fptr(grad_warp, "grad_warp"), stream_ptr())
            return []
        with _lib.device_guard(raypos.device):
            _lib.call("gol_mvp_march_bwd", c_int(N), c_int(H), c_int(W), c_int(K), fptr(raypos), fptr(raydir),
                      c_float(stepsize), fptr(tminmax), fptr(nodeaabb), fptr(primpos), fptr(primrot), fptr(primscale),
                      fptr(template), c_int(TD), c_int(TH), c_int(TW), c_float(fadescale), c_float(fadeexp),
                      fptr(raysat, "raysat"), fptr(grad_rayrgba, "grad_rayrgba"), fptr(grad_primpos),
                      fptr(grad_primrot), fptr(grad_primscale), fptr(grad_template), stream_ptr())
        return []


mvpraymarchlib = _MvpLib()


def build_accel(primtransfin, algo=0, fixedorder=True):
    """Fixed-order tree (mvpraymarch.py:21-84 with fixedorder=True): the implicit heap over the
    primitives in their given order; only the node boxes are computed.  Returns (sortedobjid,
    nodechildren, nodeaabb) like the reference (the first two are index tables the kernels never read)."""
    if not fixedorder:
        raise NotImplementedError("only usebvh='fixedorder' (the default of every caller in the reference)")
    primpos, primrot, primscale = primtransfin
    N, K = primpos.shape[:2]
    dev = primpos.device
    sortedobjid = torch.arange(K, dtype=torch.int32, device=dev).repeat(N, 1)
    nodes = torch.arange(2 * K - 1, dtype=torch.int32, device=dev)
    children = torch.stack([2 * nodes + 1, 2 * nodes + 2], -1)
    children[K - 1:] = -1 - torch.arange(K, dtype=torch.int32, device=dev)[:, None]  # leaves: -(k+1) markers
    nodechildren = children[None].repeat(N, 1, 1)
    nodeaabb = torch.empty(N, 2 * K - 1, 2, 3, device=dev)
    mvpraymarchlib.compute_aabb(primpos, primrot, primscale, sortedobjid, nodechildren, None, nodeaabb, algo)
    return sortedobjid, nodechildren, nodeaabb


class MVPRaymarch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raypos, raydir, stepsize, tminmax, primpos, primrot, primscale, template, gradmode, opts, warp=None):
        for name, t, last in (("raypos", raypos, 3), ("raydir", raydir, 3), ("tminmax", tminmax, 2)):
            assert t.is_contiguous() and t.size(3) == last, name
        for name, t in (("primpos", primpos), ("primrot", primrot), ("primscale", primscale)):
            assert t.is_contiguous() and t.size(2) == 3, name
        assert template.is_contiguous() and template.dim() == 6 and template.size(-1) == 4
        assert warp is None or (warp.is_contiguous() and warp.size(-1) == 3)  # mvpraymarch.py:125
        _, _, nodeaabb = build_accel((primpos, primrot, primscale), opts["algo"], fixedorder=True)
        N, H, W = raypos.shape[:3]
        rayrgba = torch.empty(N, H, W, 4, device=raypos.device)
        raysat = torch.full((N, H, W, 3), -1.0, device=raypos.device) if gradmode else None
        shadow = torch.zeros(*template.shape[:5], 2, device=template.device) if opts["with_shadow"] else None
        mvpraymarchlib.raymarch_forward(raypos, raydir, stepsize, tminmax, None, None, nodeaabb, primpos, primrot,
                                        primscale, template, warp, rayrgba, raysat, None, shadow, algo=opts["algo"],
                                        fadescale=opts["fadescale"], fadeexp=opts["fadeexp"])
        ctx.save_for_backward(raypos, raydir, tminmax, nodeaabb, primpos, primrot, primscale, template, raysat, warp)
        ctx.opts, ctx.stepsize = opts, stepsize
        if shadow is not None:
            ctx.mark_non_differentiable(shadow)
        return rayrgba, shadow

    @staticmethod
    def backward(ctx, grad_rayrgba, _grad_shadow):
        raypos, raydir, tminmax, nodeaabb, primpos, primrot, primscale, template, raysat, warp = ctx.saved_tensors
        if raysat is None:
            raise RuntimeError("mvpraymarch was run with gradients disabled")
        g_pos, g_rot, g_scale = torch.zeros_like(primpos), torch.zeros_like(primrot), torch.zeros_like(primscale)
        g_tpl = torch.zeros_like(template)
        g_warp = None if warp is None else torch.zeros_like(warp)
        mvpraymarchlib.raymarch_backward(raypos, raydir, ctx.stepsize, tminmax, None, None, nodeaabb, primpos, g_pos,
                                         primrot, g_rot, primscale, g_scale, template, g_tpl, warp, g_warp, None,
                                         grad_rayrgba.contiguous(), raysat, None, algo=ctx.opts["algo"],
                                         fadescale=ctx.opts["fadescale"], fadeexp=ctx.opts["fadeexp"])
        return None, None, None, None, g_pos, g_rot, g_scale, g_tpl, None, None, g_warp


def mvpraymarch(raypos, raydir, stepsize, tminmax, primtransf, template, warp, rayterm=None, algo=0,
                usebvh="fixedorder", sortprims=False, randomorder=False, maxhitboxes=512, synchitboxes=True,
                chlast=True, fadescale=8.0, fadeexp=8.0, accum=2, termthresh=0.99, griddim=3, blocksize=(8, 16),
                bwdblocksize=(8, 16), with_shadow=False):
    """raypos/raydir[N,H,W,3], tminmax[N,H,W,2], primtransf = (primpos[N,K,3], primrot[N,K,3,3],
    primscale[N,K,3]) or the packed [N,K,5,3] tensor, template[N,K,TD,TH,TW,4], warp = None (algo 0) or the warp fields
    [N,K,WD,WH,WW,3] (algo 1) -> rayrgba[N,H,W,4] (and the normalised shadow grid when with_shadow)."""
    _check_algo(algo, warp)
    if usebvh != "fixedorder" or randomorder or not chlast:
        raise NotImplementedError("only usebvh='fixedorder', randomorder=False, chlast=True")
    if isinstance(primtransf, tuple):
        primpos, primrot, primscale = primtransf
    else:
        primpos, primrot, primscale = (primtransf[:, :, 0, :].contiguous(), primtransf[:, :, 1:4, :].contiguous(),
                                       primtransf[:, :, 4, :].contiguous())
    opts = dict(algo=algo, fadescale=float(fadescale), fadeexp=float(fadeexp), with_shadow=bool(with_shadow))
    out, shadow = MVPRaymarch.apply(raypos, raydir, float(stepsize), tminmax, primpos, primrot, primscale, template,
                                    torch.is_grad_enabled(), opts, warp)
    if with_shadow:
        return out, shadow[..., 0:1] / (shadow[..., 1:] + 1e-5)
    return out


def shadow_march(raypos, raydir, stepsize, tminmax, primtransf, alpha_template, lights_per_frame, fadescale=8.0,
                 fadeexp=8.0, return_image=False):
    """The teacher model's deep-shadow march, hand_teacher_mvp.py:271-358 (`with th.no_grad()`), for all L lights of a
    frame in ONE launch WITHOUT the reference's L-fold copies of the primitive set:
        raypos / raydir [B*L,H,W,3], tminmax [B*L,H,W,2]   light-camera rays, lights of a frame consecutive
        primtransf = (primpos[B,K,3], primrot[B,K,3,3], primscale[B,K,3])   once per frame
        alpha_template [B,K,TD,TH,TW] or [B,K,TD,TH,TW,1]                   opacity only (the reference pads it with a
                                                                            constant colour it never reads back)
    -> shadow[B*L,K,TD,TH,TW,1], normalised like mvpraymarch(with_shadow=True) (accumulated visibility / weight + 1e-5);
    with return_image also the marched rayrgba[B*L,H,W,4] (colour channels 0)."""
    primpos, primrot, primscale = (_gpu_contig(t, n) for t, n in zip(primtransf, ("primpos", "primrot", "primscale")))
    raypos, raydir, tminmax = _gpu_contig(raypos, "raypos"), _gpu_contig(raydir, "raydir"), _gpu_contig(tminmax, "tminmax")
    tpl = _gpu_contig(alpha_template.reshape(alpha_template.shape[:5]), "alpha_template")
    B, K = primpos.shape[:2]
    N, H, W = raypos.shape[:3]
    L = int(lights_per_frame)
    if N != B * L:
        raise ValueError(f"{N} ray images for {B} frames x {L} lights")
    with torch.no_grad():
        _, _, nodeaabb = build_accel((primpos, primrot, primscale))
        TD, TH, TW = tpl.shape[2:5]
        shadow = torch.zeros(N, K, TD, TH, TW, 2, device=tpl.device)
        img = torch.empty(N, H, W, 4, device=tpl.device) if return_image else None
        with _lib.device_guard(tpl.device):
            _lib.call("gol_mvp_shadow_march", c_int(N), c_int(L), c_int(H), c_int(W), c_int(K), fptr(raypos), fptr(raydir),
                      c_float(float(stepsize)), fptr(tminmax), fptr(nodeaabb), fptr(primpos), fptr(primrot),
                      fptr(primscale), fptr(tpl), c_int(1), c_int(TD), c_int(TH), c_int(TW), c_float(float(fadescale)),
                      c_float(float(fadeexp)), fptr(img), fptr(shadow), stream_ptr())
        out = shadow[..., 0:1] / (shadow[..., 1:] + 1e-5)
    return (out, img) if return_image else out


# ------------------------------------------------------------------------------------------- raydirs
class _UtilsLib:
    @staticmethod
    def compute_raydirs_forward(viewpos, viewrot, focal, princpt, pixelcoords, W, H, volradius, raypos, raydir,
                                tminmax):
        N = viewpos.shape[0]
        _need_gpu(viewpos, "viewpos")
        with _lib.device_guard(viewpos.device):
            _lib.call("gol_raydirs_fwd", c_int(N), c_int(H), c_int(W), fptr(viewpos, "viewpos"),
                      fptr(viewrot, "viewrot"), fptr(focal, "focal"), fptr(princpt, "princpt"),
                      fptr(pixelcoords, "pixelcoords"), c_float(volradius), fptr(raypos), fptr(raydir),
                      fptr(tminmax), stream_ptr())
        return []

    @staticmethod
    def compute_raydirs_backward(*_a, **_k):  # the reference's backward kernel writes nothing (utils.py:49-50)
        return []


utilslib = _UtilsLib()


def compute_raydirs(viewpos, viewrot, focal, princpt, pixelcoords, volradius):
    """pixelcoords: [N,H,W,2] tensor or a (W, H) tuple for the implicit pixel grid.  No gradients
    (the reference returns None for every input, utils.py:49-50)."""
    N = viewpos.size(0)
    if isinstance(pixelcoords, tuple):
        W, H = pixelcoords
        pc = None
    else:
        pc = pixelcoords.detach()
        H, W = pc.size(1), pc.size(2)
    for t in (viewpos, viewrot, focal, princpt) + ((pc,) if pc is not None else ()):
        assert t.is_contiguous()
    dev = viewpos.device
    raypos, raydir = torch.empty(N, H, W, 3, device=dev), torch.empty(N, H, W, 3, device=dev)
    tminmax = torch.empty(N, H, W, 2, device=dev)
    utilslib.compute_raydirs_forward(viewpos.detach(), viewrot.detach(), focal.detach(), princpt.detach(), pc, W, H,
                                     float(volradius), raypos, raydir, tminmax)
    return raypos, raydir, tminmax


class Raymarcher(torch.nn.Module):
    """ca_code/utils/render_raymarcher.py:17-71: scale positions by the volume radius, keep only
    `valid_prims`, march, return (rgb[N,3,H,W], alpha[N,1,H,W], rgba[N,4,H,W], shadow)."""

    def __init__(self, volradius, dt: float = 1.0):
        super().__init__()
        self.volume_radius = volradius
        self.dt = dt / self.volume_radius  # step size in normalised volume units

    def forward(self, raypos, raydir, tminmax, decout, renderoptions={}, rayterm=None, with_shadow=False):
        primpos = decout["primpos"] / self.volume_radius
        primrot, primscale, template = decout["primrot"], decout["primscale"], decout["primrgba"]
        keep = decout.get("valid_prims", None)
        if keep is not None:
            assert keep.shape[0] == template.shape[1]
            template, primpos = template[:, keep].contiguous(), primpos[:, keep].contiguous()
            primrot, primscale = primrot[:, keep].contiguous(), primscale[:, keep].contiguous()
        known = mvpraymarch.__code__.co_varnames
        out = mvpraymarch(raypos, raydir, self.dt, tminmax, (primpos, primrot, primscale), template=template,
                          warp=decout["warp"] if "warp" in decout else None, rayterm=rayterm,
                          with_shadow=with_shadow, **{k: v for k, v in renderoptions.items() if k in known})
        rayrgba, shadow = out if with_shadow else (out, None)
        rayrgba = rayrgba.permute(0, 3, 1, 2)
        return rayrgba[:, :3].contiguous(), rayrgba[:, 3:4].contiguous(), rayrgba, shadow
