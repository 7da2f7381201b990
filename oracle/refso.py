"""oracle/refso.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/_ref/libref.so: the REFERENCE's own kernels
    /root/reference/extensions/sgutils/sg.cu:27-175          evaluate_gaussian_{fwd,bwd}_kernel
    /root/reference/extensions/utils/utils_kernel.cu:11-51   compute_raydirs_forward_kernel
compiled for the host by oracle/Makefile (target `ref`) through the CUDA stand-in header
oracle/ref_shim/cuda_runtime.h.  It pins the sgutils restatement (oracle/sg_oracle.c) and is itself a checker of the
HIP kernels.  The library is built where /root/reference exists and travels with the tree (git-ignored); `available()`
is False when it was never built.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libref.so")
_lib = None


def available():
    return os.path.exists(SO)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(SO)
    return _lib


def _f(t):
    return t.detach().to(torch.float32).contiguous().cpu()


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


c_int, c_float = ctypes.c_int, ctypes.c_float


def evaluate_gaussian_fwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, w_type=0):
    a = [_f(t) for t in (lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts)]
    nl = n_lights.to(torch.int32).contiguous().cpu()
    N, D = a[0].shape[:2]
    L = a[2].shape[1]
    out = torch.empty(N, D, 3)  # sgutils.py:26: th.empty -- the kernel writes every element
    lib().ref_sg_fwd(c_int(N), c_int(D), c_int(L), *[_p(t) for t in a], _p(nl), c_int(w_type), _p(out))
    return out


def evaluate_gaussian_bwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, grad_integral, w_type=0,
                          want_light_grad=False):
    a = [_f(t) for t in (lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts)]
    nl = n_lights.to(torch.int32).contiguous().cpu()
    g = _f(grad_integral)
    N, D = a[0].shape[:2]
    L = a[2].shape[1]
    gd, gs = torch.zeros(N, D, 3), torch.zeros(a[1].shape)  # sgutils.py:43-47: zeros
    gl = torch.zeros(N, L, 3) if want_light_grad else None
    lib().ref_sg_bwd(c_int(N), c_int(D), c_int(L), *[_p(t) for t in a], _p(nl), _p(g), c_int(w_type), _p(gd), _p(gs), _p(gl))
    return gd, gs, gl


def compute_raydirs(viewpos, viewrot, focal, princpt, pixelcoords, volradius):
    viewpos, viewrot, focal, princpt = map(_f, (viewpos, viewrot, focal, princpt))
    N = viewpos.shape[0]
    if isinstance(pixelcoords, tuple):
        W, H = pixelcoords
        pc = None
    else:
        pc = _f(pixelcoords)
        H, W = pc.shape[1:3]
    rp, rd, tm = torch.empty(N, H, W, 3), torch.empty(N, H, W, 3), torch.empty(N, H, W, 2)
    lib().ref_raydirs(c_int(N), c_int(H), c_int(W), _p(viewpos), _p(viewrot), _p(focal), _p(princpt), _p(pc),
                      c_float(volradius), _p(rp), _p(rd), _p(tm))
    return rp, rd, tm
