"""Spherical-Gaussian specular lobe operator (host side) on top of the C ABI.

Drop-in for the reference operator (interface only; the implementation is libgoliath_hip.so):
  evaluate_gaussian(...)                      <- extensions/sgutils/sgutils.py:65-98
  sgutilslib.evaluate_gaussian_fwd / _bwd     <- extensions/sgutils/sg.cu:177-283 (pybind module)
Contract kept from the reference: outputs are caller-allocated and written in place on the
current stream; shape/device/contiguity violations raise RuntimeError; prim_pts, light_pts and
n_lights never receive gradients (sgutils.py:30); light_values does only if it requires grad.
"""
import torch
import torch.nn.functional as F

from . import _lib
from ._lib import c_int, fptr, iptr, stream_ptr

_W_TYPES = (0, 1, 2, 3)


def _dims(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, extra=()):
    if not lobe_dirs.is_cuda:
        raise RuntimeError("lobe_dirs must be a CUDA tensor")  # CHECK_INPUT, sgutils/utils.h:1-5
    n_views = lobe_dirs.size(0)
    named = [("lobe_sigmas", lobe_sigmas), ("light_values", light_values), ("light_pts", light_pts),
             ("prim_pts", prim_pts), *extra]
    for name, t in named:
        if t.size(0) != n_views:
            raise RuntimeError(f"Batch dim mismatch for {name}.")
    if n_lights.dtype != torch.int32:
        raise RuntimeError("n_lights must be an int32 tensor")
    return n_views, lobe_dirs.size(1), light_values.size(1)


class _Lib:
    """Same two entry points as the reference's compiled `sgutilslib` module."""

    @staticmethod
    def evaluate_gaussian_fwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights,
                              integral, w_type):
        N, D, L = _dims(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights,
                        extra=[("integral", integral)])
        with _lib.device_guard(lobe_dirs.device):
            _lib.call("gol_sg_eval_fwd", c_int(N), c_int(D), c_int(L), fptr(lobe_dirs, "lobe_dirs"),
                      fptr(lobe_sigmas, "lobe_sigmas"), fptr(light_values, "light_values"),
                      fptr(light_pts, "light_pts"), fptr(prim_pts, "prim_pts"),
                      iptr(n_lights, "n_lights"), fptr(integral, "integral"), c_int(int(w_type)),
                      stream_ptr())
        return []

    @staticmethod
    def evaluate_gaussian_bwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights,
                              grad_integral, grad_dirs, grad_lobe_sigmas, grad_light_values, w_type):
        N, D, L = _dims(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights)
        with _lib.device_guard(lobe_dirs.device):
            _lib.call("gol_sg_eval_bwd", c_int(N), c_int(D), c_int(L), fptr(lobe_dirs, "lobe_dirs"),
                      fptr(lobe_sigmas, "lobe_sigmas"), fptr(light_values, "light_values"),
                      fptr(light_pts, "light_pts"), fptr(prim_pts, "prim_pts"),
                      iptr(n_lights, "n_lights"), fptr(grad_integral, "grad_integral"),
                      fptr(grad_dirs, "grad_dirs"), fptr(grad_lobe_sigmas, "grad_lobe_sigmas"),
                      fptr(grad_light_values, "grad_light_values"), c_int(int(w_type)), stream_ptr())
        return []


sgutilslib = _Lib()


class EvaluateGaussian(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, w_type):
        if w_type not in _W_TYPES or light_values.shape[-1] != 3:
            raise AssertionError("w_type must be 0..3 and light_values RGB")
        out = torch.empty(*lobe_dirs.shape[:2], 3, device=lobe_dirs.device)
        sgutilslib.evaluate_gaussian_fwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts,
                                         n_lights, out, w_type)
        ctx.mark_non_differentiable(light_pts, prim_pts, n_lights)
        ctx.save_for_backward(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights)
        ctx.w_type = w_type
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dirs, sigmas, values, pts, prims, n_lights = ctx.saved_tensors
        g_dirs, g_sigmas = torch.zeros_like(dirs), torch.zeros_like(sigmas)
        g_values = torch.zeros_like(values) if ctx.needs_input_grad[2] else None
        sgutilslib.evaluate_gaussian_bwd(dirs, sigmas, values, pts, prims, n_lights, grad_out.contiguous(),
                                         g_dirs, g_sigmas, g_values, ctx.w_type)
        return g_dirs, g_sigmas, g_values, None, None, None, None


def evaluate_gaussian(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights,
                      w_type: int = 0, normalize_lobe_dirs: bool = True):
    """integral[n,d,:] = sum_{l < n_lights[n]} light_values[n,l,:] * w(angle(lobe_dirs[n,d], light l), sigma)."""
    if normalize_lobe_dirs:
        lobe_dirs = F.normalize(lobe_dirs, dim=-1)
    assert lobe_dirs.shape[-1] == 3 and prim_pts.shape[-1] == 3 and light_pts.shape[-1] == 3
    assert light_pts.dim() == 3 and prim_pts.dim() == 3
    assert lobe_sigmas.dim() == 2 or lobe_sigmas.shape[2] == 1
    return EvaluateGaussian.apply(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, w_type)
