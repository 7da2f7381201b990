"""ctypes binding of libgoliath_hip.so (C ABI: include/goliath_hip.h).

PyTorch is only plumbing here: tensors provide device memory (`data_ptr()`) and the current HIP
stream.  There is NO CPU fallback: if the HIP library is missing or a tensor is not on the GPU the
call raises, loudly.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GOLIATH_HIP_LIB: load another build of the same ABI instead (the test-only exact-math twin, goliath_amd/build.py)
LIB_PATH = os.environ.get("GOLIATH_HIP_LIB") or os.path.join(_HERE, "lib", "libgoliath_hip.so")

_lib = None

_SYMBOLS = [
    "gol_version", "gol_last_error", "gol_sg_eval_fwd", "gol_sg_eval_bwd", "gol_project_fwd",
    "gol_project_bwd", "gol_project_bwd_records", "gol_bin_sort", "gol_splat_pack", "gol_render_layout", "gol_render_fwd", "gol_render_bwd", "gol_rasterize_fwd", "gol_rasterize_bwd", "gol_raster_plan", "gol_raster_count_pairs", "gol_shade_fwd",
    "gol_shade_bwd", "gol_shade_project_fwd", "gol_shade_project_bwd", "gol_render_layout_projected", "gol_render_fwd_projected",
    "gol_render_bwd_projected", "gol_selftest_wave_sum4", "gol_raydirs_fwd", "gol_mvp_aabb", "gol_mvp_march_fwd",
    "gol_mvp_march_bwd", "gol_mvp_march_warp_fwd", "gol_mvp_march_warp_bwd", "gol_envmap_pack", "gol_uvlight_phong_fwd", "gol_uvlight_phong_bwd",
    "gol_uvlight_ggx_fwd", "gol_uvlight_ggx_bwd", "gol_l1_blocks", "gol_l1_fwd", "gol_l1_bwd",
    "gol_tail_conv_fwd", "gol_tail_conv_bwd", "gol_tail_conv_bwd_scratch_floats", "gol_ssim_blocks", "gol_ssim_fwd", "gol_ssim_bwd",
    "gol_shadow_pcf", "gol_imgtail_partial_floats", "gol_imgtail_fwd", "gol_imgtail_bwd", "gol_mvp_shadow_march", "gol_mesh_raster_workspace_bytes", "gol_mesh_raster",
]


class GoliathHipError(RuntimeError):
    pass


def load():
    """Load the shared library (build it with `python -m goliath_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GoliathHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -m goliath_amd.build` (or __graft_entry__.build()). There is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        lib.gol_version.restype = ctypes.c_char_p
        lib.gol_last_error.restype = ctypes.c_char_p
        for name in _SYMBOLS:
            getattr(lib, name)  # AttributeError if the ABI and the build disagree
        _lib = lib
    return _lib


def exported_symbols():
    return list(_SYMBOLS)


def version():
    return load().gol_version().decode()


def ptr(t, dtype=None, name="tensor"):
    """Device pointer of a contiguous CUDA(HIP) tensor; None -> NULL.  Mirrors the reference's
    CHECK_INPUT (extensions/sgutils/utils.h:1-5): RuntimeError for non-GPU / non-contiguous."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise GoliathHipError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise GoliathHipError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise GoliathHipError(f"{name} must have dtype {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def fptr(t, name="tensor"):
    return ptr(t, torch.float32, name)


def iptr(t, name="tensor"):
    return ptr(t, torch.int32, name)


def stream_ptr():
    """The current PyTorch HIP stream of the current device as a raw handle.  (torch.cuda.current_stream() builds a Python
    Stream object, ~10 us per call and 20 calls per step; the raw query is what the launch path needs.)"""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


class device_guard:
    """`with device_guard(t.device):` -- make the tensor's GPU current for the enclosed launches (the reference extensions
    use a CUDAGuard, sg.cu:204; mvpraymarch has none).  A cheap stand-in for torch.cuda.device(): no
    device switch at all in the common case that the device already is current."""
    __slots__ = ("idx", "prev")

    def __init__(self, dev):
        self.idx = dev.index if isinstance(dev, torch.device) else dev
        self.prev = None

    def __enter__(self):
        if self.idx is not None:
            self.prev = torch._C._cuda_getDevice()
            if self.prev != self.idx:
                torch._C._cuda_setDevice(self.idx)
        return self

    def __exit__(self, *exc):
        if self.idx is not None and self.prev != self.idx:
            torch._C._cuda_setDevice(self.prev)
        return False


TIMING = None  # set to a list to record (name, start_event, end_event) around every ABI call


def call(fn_name, *args):
    lib = load()
    if TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # current stream == the stream the kernels are launched on (stream_ptr())
        rc = getattr(lib, fn_name)(*args)
        e1.record()
        TIMING.append((fn_name, e0, e1))
    else:
        rc = getattr(lib, fn_name)(*args)
    if rc != 0:
        raise GoliathHipError(f"{fn_name} failed ({rc}): {lib.gol_last_error().decode()}")


c_int = ctypes.c_int
c_float = ctypes.c_float
c_i64 = ctypes.c_int64
