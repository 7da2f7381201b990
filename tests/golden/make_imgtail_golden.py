"""Generate tests/golden/imgtail_golden.npz by running the REFERENCE's own modules on CPU:
    CalV5.forward                /root/reference/ca_code/nn/color_cal.py:211-241 (training mode: gradient hook included)
    the background composite     /root/reference/ca_code/models/rgca.py:226-230 (copied statement by statement below)
    LearnableBlur.forward        /root/reference/ca_code/nn/dof_cal.py:44-56
`torchvision.transforms.functional.gaussian_blur` is not installed here; tests/golden/ref_stubs.py restates it with
plain torch ops (reflect padding, sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8, depthwise conv).  Run in the build container
only: python tests/golden/make_imgtail_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
from ca_code.nn.color_cal import CalV5  # noqa: E402
from ca_code.nn.dof_cal import LearnableBlur  # noqa: E402

CAMERAS = ["400002", "400004", "410011", "400013", "410020"]   # "41..." = grey-scale cameras (color_cal.py:134-136)
IDENTITY = "400004"


def case(tag, B, H, W, cams, use_cal, use_bg, use_blur, seed, out):
    g = torch.Generator().manual_seed(seed)
    rgb = torch.rand(B, 3, H, W, generator=g).requires_grad_(True)
    alpha = torch.rand(B, 1, H, W, generator=g)
    background = torch.rand(B, 3, H, W, generator=g)
    lit = torch.tensor([True, False, True, True, False][:B])
    cal = CalV5(cameras=CAMERAS, identity_camera=IDENTITY)
    blur = LearnableBlur(CAMERAS)
    with torch.no_grad():
        cal.holder.params[:, :3] += 0.2 * torch.randn(len(CAMERAS), 3, generator=g)
        cal.holder.params[:, 3:] += 0.1 * torch.randn(len(CAMERAS), 3, generator=g)
        blur.weights_raw += torch.randn(len(CAMERAS), 3, generator=g)
    cal.train()
    x = rgb
    if use_cal:
        x = cal(x, cal.name_to_idx(cams))                       # rgca.py:223-224
    if use_bg:                                                  # rgca.py:226-230
        bg = background[:, :3].clone()
        bg[torch.logical_not(lit)] *= 0.0
        x = x + (1.0 - alpha) * bg
    if use_blur:
        x = blur(x, cams)                                       # rgca.py:249-250
    w = torch.randn(x.shape, generator=g)
    (x * w).sum().backward()
    out[f"{tag}/cams"] = np.array(cams)
    out[f"{tag}/flags"] = np.array([use_cal, use_bg, use_blur])
    for k, v in dict(rgb=rgb, alpha=alpha, background=background, lit=lit, w=w, out=x, cal_params=cal.holder.params,
                     blur_raw=blur.weights_raw, g_rgb=rgb.grad,
                     g_cal=cal.holder.params.grad if use_cal else torch.zeros(len(CAMERAS), 6),
                     g_blur=blur.weights_raw.grad if use_blur else torch.zeros(len(CAMERAS), 3)).items():
        out[f"{tag}/{k}"] = v.detach().numpy()


def main():
    out = {"cameras": np.array(CAMERAS), "identity": np.array(IDENTITY)}
    # views: colour, identity, grey, colour, grey -- ragged image sizes (tile = 32), borders within the 7-tap radius
    cams5 = ["400002", "400004", "410011", "400013", "410020"]
    case("all", 5, 37, 45, cams5, True, True, True, 1, out)
    case("blur_only", 2, 33, 32, ["400013", "410011"], False, False, True, 2, out)
    case("cal_only", 3, 20, 37, ["410020", "400004", "400002"], True, False, False, 3, out)
    case("tiny", 1, 4, 5, ["400002"], True, True, True, 4, out)   # the smallest image reflect padding allows
    path = os.path.join(HERE, "imgtail_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
