"""TEST INFRASTRUCTURE: stand-ins for the third-party packages the reference's pure-Python modules import but this image
lacks (cv2, torchvision, pytorch3d, drtk, addict, omegaconf, gsplat), so that `ca_code.*` imports UNCHANGED from
/root/reference in the build container -- for generating goldens with the reference's own code and for checking the
drop-in against the real classes.  None of the stand-ins computes anything on the paths under test, except
`gaussian_blur`, which restates torchvision's (reflect padding, sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8, depthwise
separable kernel) with plain torch ops because LearnableBlur's golden needs it."""
import sys
import types

import torch
import torch.nn.functional as F

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Any(types.ModuleType):
    """Module whose every attribute exists (cv2.INTER_LINEAR as a default argument, torchvision.models.vgg19, ...)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return 0


def gaussian_blur(img, kernel_size, sigma=None):
    """torchvision.transforms.functional.gaussian_blur for [..., C, H, W] float tensors."""
    ks = [kernel_size, kernel_size] if isinstance(kernel_size, int) else list(kernel_size)
    if sigma is None:
        sigma = [k * 0.15 + 0.35 for k in ks]  # == 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    elif isinstance(sigma, (int, float)):
        sigma = [float(sigma)] * 2

    def k1d(k, s):
        x = torch.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k, dtype=img.dtype, device=img.device)
        pdf = torch.exp(-0.5 * (x / s) ** 2)
        return pdf / pdf.sum()

    kx, ky = k1d(ks[0], sigma[0]), k1d(ks[1], sigma[1])
    k2d = ky[:, None] * kx[None, :]
    shape = img.shape
    x = img.reshape(-1, 1, shape[-2], shape[-1])
    x = F.pad(x, [ks[0] // 2, ks[0] // 2, ks[1] // 2, ks[1] // 2], mode="reflect")
    return F.conv2d(x, k2d[None, None]).reshape(shape)


class AttrDict(dict):
    """addict.Dict: attribute access + recursive conversion (the two features the reference uses)."""

    def __init__(self, *a, **k):
        super().__init__()
        for key, v in dict(*a, **k).items():
            self[key] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def install(sgutilslib=None):
    """Register the stand-ins (idempotent) and put /root/reference on sys.path.  sgutilslib: optional object serving
    evaluate_gaussian_fwd / _bwd (the goldens use the reference's own kernels compiled for the host, oracle/refso)."""
    sys.modules.setdefault("cv2", _Any("cv2"))
    if "torchvision" not in sys.modules or isinstance(sys.modules["torchvision"], types.ModuleType) and not hasattr(
            sys.modules["torchvision"], "__version__"):
        tv = _stub("torchvision")
        tv.utils = _stub("torchvision.utils", make_grid=lambda *a, **k: None)
        tv.transforms = _stub("torchvision.transforms")
        tv.transforms.functional = _stub("torchvision.transforms.functional", gaussian_blur=gaussian_blur)
        tv.models = _Any("torchvision.models")
        tv.models.__path__ = []  # a package: the perceptual losses import torchvision.models.<arch>
        sys.modules["torchvision.models"] = tv.models
        for sub in ("efficientnet", "vgg"):
            sys.modules[f"torchvision.models.{sub}"] = _Any(f"torchvision.models.{sub}")
    _stub("turtle", forward=None)           # ca_code/nn/blocks.py:8 has a stray `from turtle import forward` (needs tkinter)
    _stub("pytorch3d")
    _stub("pytorch3d.renderer", RasterizationSettings=None, MeshRasterizer=None)
    _stub("pytorch3d.renderer.mesh.textures", TexturesUV=None)
    _stub("pytorch3d.utils", cameras_from_opencv_projection=None)
    _stub("pytorch3d.transforms", axis_angle_to_matrix=None, euler_angles_to_matrix=None, matrix_to_axis_angle=None)
    _stub("pytorch3d.io", load_ply=None)
    _stub("pytorch3d.renderer.mesh")
    _stub("pytorch3d.renderer.mesh.rasterize_meshes", rasterize_meshes=None)
    _stub("pytorch3d.structures", Meshes=None)
    sys.modules.setdefault("gsplat", _stub("gsplat", project_gaussians=None, rasterize_gaussians=None))
    _stub("drtk", rasterize=None, render=None, transform=None, interpolate=None, edge_grad_estimator=None)
    _stub("addict", Dict=AttrDict)
    _stub("omegaconf", DictConfig=dict, OmegaConf=_Any("OmegaConf"))
    if sgutilslib is not None:
        _stub("sgutilslib", evaluate_gaussian_fwd=sgutilslib.evaluate_gaussian_fwd,
              evaluate_gaussian_bwd=sgutilslib.evaluate_gaussian_bwd)
    if REF not in sys.path:
        sys.path.insert(0, REF)
