"""Decoder-tail fusion by light contraction.

The last layer of `vnocond_mod` is a transposed conv 16 -> 125 channels with an untied bias
(/root/reference/ca_code/models/rgca.py:427), and 113 of those channels are diffuse SH coefficients that
the shading tail immediately contracts with the light's SH coefficients (rgca.py:506-514,528-530):
    diff_sum[b,c,n] = sum_k f_vnocond[b, k, n] * Lc[b,c,k],   f_vnocond = convT(x, W) + bias.
Both steps are linear, so the contraction moves in front of the convolution:
    diff_sum[b,c] = convT(x[b], sum_k Lc[b,c,k] W[:,k])  +  sum_k Lc[b,c,k] bias[k].
The 125-channel activation (516 B per Gaussian per view, written by the conv, re-read and re-written by the
bias add, re-read by the shading kernel, and the same again for its gradient) is never materialised:
the conv produces 3 (+3 for the training-only random light) + 12 channels per view, the bias is read once per
batch as a [3B x 113] x [113 x N] contraction.  The HIP shading kernels then run unchanged on the compact
tensor (one "SH coefficient" per colour whose light coefficient is 1).

The per-view transposed conv and the bias contraction run as HIP kernels behind the C ABI
(gol_tail_conv_fwd / gol_tail_conv_bwd, csrc/tail.hip: SGPR-weight conv, gather-form input gradient, fp32-MFMA
weight gradient); `contracted_vnocond_torch` is the same computation in plain PyTorch (the parity reference of
tests/test_gpu_tail.py, and what tests/test_decoder.py checks against the reference's order of operations).

Autograd: the light contraction of the weights is torch linear algebra on small tensors, so gradients reach
weight_v / weight_g through the ordinary graph; x, the contracted weights and the untied bias get theirs from
gol_tail_conv_bwd.
"""

import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import c_int, fptr, stream_ptr
from .shade import shading_tail_coefs


class _TailConv(torch.autograd.Function):
    """out[B,CH,2h,2w] = convT(x[b], weff[b]) + [lc[b] . bias[:nd] ; bias[nd:]]  (include/goliath_hip.h)."""

    @staticmethod
    def forward(ctx, x, weff, lc, bias):
        B, Ci, h, w = x.shape
        wB, _, CH = weff.shape[:3]
        nd, E = (lc.shape[0], lc.shape[2]) if lc is not None else (0, 0)   # lc is plane-major [nd,B,E]
        out = torch.empty(B, CH, 2 * h, 2 * w, device=x.device)
        with _lib.device_guard(x.device):
            _lib.call("gol_tail_conv_fwd", c_int(B), c_int(Ci), c_int(h), c_int(w), c_int(CH), c_int(E), c_int(nd),
                      c_int(wB), fptr(x, "x"), fptr(weff, "weff"), fptr(lc, "lc"), fptr(bias, "bias"), fptr(out),
                      stream_ptr())
        ctx.save_for_backward(x, weff, lc)
        ctx.dims = (B, Ci, h, w, CH, E, nd, wB, bias.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weff, lc = ctx.saved_tensors
        B, Ci, h, w, CH, E, nd, wB, bshape = ctx.dims
        g = g.to(torch.float32).contiguous()
        need_x, need_w, _, need_b = ctx.needs_input_grad
        g_x = torch.empty_like(x) if need_x else None
        g_w = torch.zeros_like(weff) if need_w else None
        g_b = torch.empty(bshape, device=x.device) if need_b else None
        wt = weff.permute(0, 2, 3, 4, 1).contiguous() if need_x else None   # [wB,CH,4,4,16]
        scratch = None
        if need_w:  # per-workgroup partial sums of the weight gradient (reduced by a second kernel: no float atomics)
            fn = _lib.load().gol_tail_conv_bwd_scratch_floats
            fn.restype = ctypes.c_longlong
            scratch = torch.empty(int(fn(c_int(B), c_int(h), c_int(w), c_int(CH))), device=x.device)
        with _lib.device_guard(x.device):
            _lib.call("gol_tail_conv_bwd", c_int(B), c_int(Ci), c_int(h), c_int(w), c_int(CH), c_int(E), c_int(nd),
                      c_int(wB), fptr(x, "x"), fptr(wt, "weff_t"), fptr(lc, "lc"), fptr(g, "g_out"), fptr(g_x),
                      fptr(g_w), fptr(g_b), fptr(scratch), stream_ptr())
        return g_x, g_w, None, g_b


def tail_conv(x, weff, lc, bias):
    """x[B,16,h,w], weff[B|1,16,CH,4,4], lc[B,E,nd] or None, bias[nd+CH-E,2h,2w] -> [B,CH,2h,2w]."""
    if lc is not None:
        lc = lc.detach().permute(2, 0, 1)  # the ABI wants it plane-major
    if not x.is_cuda:
        raise _lib.GoliathHipError("tail_conv needs CUDA(HIP) tensors; there is no CPU path")
    c = lambda t: None if t is None else t.to(torch.float32).contiguous()
    return _TailConv.apply(c(x), c(weff), c(lc), c(bias))


def wn_weight(layer):
    """Effective weight of a weight-normalised layer (ca_code/nn/layers.py:157-244: one magnitude per
    output channel, direction normalised over the whole tensor)."""
    return layer.weight_v * (layer.weight_g / layer.weight_v.norm())


def light_matrix(light_sh, ncol, nmono):
    """light_sh[B,3,ncol+nmono] -> Lc[B,3,3*ncol+nmono] acting on the SH channels of f_vnocond
    (channel c*ncol+k is colour c's coefficient k; channel 3*ncol+k is shared, rgca.py:506-514)."""
    B = light_sh.shape[0]
    Lc = light_sh.new_zeros(B, 3, 3 * ncol + nmono)
    for c in range(3):
        Lc[:, c, c * ncol:(c + 1) * ncol] = light_sh[:, c, :ncol]
    Lc[:, :, 3 * ncol:] = light_sh[:, :, ncol:]
    return Lc


def _contract(x_vn, weight, light_sh, light_sh_rand, ncol, nmono):
    B = x_vn.shape[0]
    nd = 3 * ncol + nmono
    Lc = light_matrix(light_sh, ncol, nmono)
    if light_sh_rand is not None:  # rows interleaved (c0, c0_rand, c1, c1_rand, ...): the kernel's channel order
        Lc = torch.stack([Lc, light_matrix(light_sh_rand, ncol, nmono)], 2).reshape(B, 6, nd)
    w_sh = torch.einsum("bek,ikyx->bieyx", Lc, weight[:, :nd])                       # [B,16,E,4,4]
    w_eff = torch.cat([w_sh, weight[None, :, nd:].expand(B, -1, -1, -1, -1)], 2)      # [B,16,E+12,4,4]
    return Lc, w_eff, nd


def contracted_vnocond(x_vn, weight, bias, light_sh, light_sh_rand, ncol, nmono):
    """x_vn[B,16,h,w], weight[16,125,4,4], bias[125,2h,2w] -> f_c[B,E+12,2h,2w] with E = 3 (or 6 with a random
    light): channels [diff_sum per colour (interleaved with diff_sum_rand when given), 12 Gaussian-parameter channels].
    HIP kernels."""
    Lc, w_eff, _ = _contract(x_vn, weight, light_sh, light_sh_rand, ncol, nmono)
    return tail_conv(x_vn, w_eff, Lc.detach(), bias), Lc.shape[1]


def contracted_vnocond_torch(x_vn, weight, bias, light_sh, light_sh_rand, ncol, nmono):
    """The same in plain PyTorch (grouped transposed conv + einsum): parity reference, runs on CPU."""
    B, Ci, h, w = x_vn.shape
    Lc, w_eff, nd = _contract(x_vn, weight, light_sh, light_sh_rand, ncol, nmono)
    E = Lc.shape[1]
    f_c = F.conv_transpose2d(x_vn.reshape(1, B * Ci, h, w), w_eff.reshape(B * Ci, E + 12, 4, 4), None, 2, 1,
                             groups=B).view(B, E + 12, 2 * h, 2 * w)
    b_sh = torch.einsum("bek,kn->ben", Lc, bias[:nd].reshape(nd, -1)).view(B, E, 2 * h, 2 * w)
    return f_c + torch.cat([b_sh, bias[None, nd:].expand(B, -1, -1, -1)], 1), E


def fused_tail(last_vn, last_vc, x_vn, x_vc, postex, tn, albedo, headrel_light_sh, headrel_campos,
               light_intensity=None, headrel_light_pos=None, n_lights=None, preconv_envmap=None, lightrot=None,
               light_sh_rand=None, n_color_sh=3, n_diff_sh=8, views=None):
    """Same outputs as `shading_tail(last_vn(x_vn), last_vc(x_vc), ...)` without the 125-channel tensor.
    last_vn / last_vc: the final ConvTranspose2dWNUB modules (parameters weight_v, weight_g, bias)."""
    ncol = (n_color_sh + 1) ** 2
    nmono = (n_diff_sh + 1) ** 2 - ncol
    f_c, E = contracted_vnocond(x_vn, wn_weight(last_vn), last_vn.bias, headrel_light_sh, light_sh_rand, ncol, nmono)
    f_vc = tail_conv(x_vc, wn_weight(last_vc)[None], None, last_vc.bias)
    B = x_vn.shape[0]
    if E == 3:   # one coefficient per colour, light coefficient 1
        f_in, sel, sel_r, kc = f_c, f_c.new_ones(B, 3, 1), None, 1
    else:        # two "coefficients" per colour: (diff_sum, diff_sum_rand), selected by (1,0) / (0,1)
        f_in = f_c
        sel = f_c.new_tensor([1.0, 0.0]).expand(B, 3, 2).contiguous()
        sel_r = f_c.new_tensor([0.0, 1.0]).expand(B, 3, 2).contiguous()
        kc = 2
    return shading_tail_coefs(f_in, f_vc, postex, tn, albedo, sel, headrel_campos, kc, 0, light_intensity,
                              headrel_light_pos, n_lights, preconv_envmap, lightrot, sel_r, views)
