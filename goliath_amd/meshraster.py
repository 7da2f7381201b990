"""Mesh render layer on top of `gol_mesh_raster` (csrc/meshraster.hip): what the reference gets from drtk.

    transform(verts, K, Rt)                 drtk.transform as called at ca_code/utils/render_drtk.py:42
    rasterize(v_pix, vi, h, w)              drtk.rasterize + drtk.render (:44-46): index_img, depth_img, bary_img
    RenderLayer(h, w, vi, vt, vti)          ca_code/utils/render_drtk.py:14-82 (same constructor / forward / dict keys)
drtk is a third-party dependency that is NOT in the reference tree (requirements.txt:6); the sampling conventions are
stated in csrc/meshraster.hip (pixel centres at (j + 0.5, i + 0.5), as pytorch3d's cameras_from_opencv_projection places
them for the reference's other back-end, render_pytorch3d.py:49-51; if drtk samples at integer coordinates instead the
shadow depth map is shifted by half a pixel -- not verifiable here, and below the 3x3 PCF footprint of its only consumer).
The hot-path consumer is the shadow-map depth render (ca_code/utils/shadowmap.py:39-50, under no_grad in
ca_code/models/urhand.py:404,492).  Texture gradients flow (autograd through grid_sample); drtk's edge-gradient
estimator (vertex gradients) has no counterpart here, so a call that needs it raises instead of silently dropping them.
"""
import ctypes
from typing import List, Optional

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import c_int, fptr, iptr, stream_ptr


def transform(verts: torch.Tensor, K: torch.Tensor, Rt: torch.Tensor) -> torch.Tensor:
    """verts[B,V,3] (world), K[B,3,3], Rt[B,3,4] -> v_pix[B,V,3] = (pixel x, pixel y, camera-space z)."""
    v_cam = verts @ Rt[:, :3, :3].transpose(1, 2) + Rt[:, None, :3, 3]
    z = v_cam[..., 2:3]
    uv = (v_cam / z.clamp(min=1e-8)) @ K.transpose(1, 2)
    return torch.cat([uv[..., :2], z], dim=-1)


def rasterize(v_pix: torch.Tensor, vi: torch.Tensor, height: int, width: int, with_bary: bool = True):
    """v_pix[B,V,3], vi[F,3] -> index_img[B,H,W] int32 (-1 = empty), depth_img[B,H,W] (0 = empty),
    bary_img[B,3,H,W] (perspective-correct; None unless with_bary)."""
    if not v_pix.is_cuda:
        raise _lib.GoliathHipError("rasterize needs CUDA(HIP) tensors; there is no CPU path")
    v_pix = v_pix.detach().to(torch.float32).contiguous()
    vi = vi.to(device=v_pix.device, dtype=torch.int32).contiguous()
    B, V = v_pix.shape[:2]
    Fc = vi.shape[0]
    dev = v_pix.device
    index_img = torch.empty(B, height, width, dtype=torch.int32, device=dev)
    depth_img = torch.empty(B, height, width, device=dev)
    bary_img = torch.empty(B, 3, height, width, device=dev) if with_bary else None
    fn = _lib.load().gol_mesh_raster_workspace_bytes
    fn.restype = ctypes.c_int64
    ws = torch.empty(max(int(fn(c_int(B), c_int(Fc))), 4) // 4, dtype=torch.int32, device=dev)
    with _lib.device_guard(dev):
        _lib.call("gol_mesh_raster", c_int(B), c_int(V), c_int(Fc), c_int(height), c_int(width), fptr(v_pix), iptr(vi),
                  iptr(index_img), fptr(depth_img), fptr(bary_img), iptr(ws), stream_ptr())
    return index_img, depth_img, bary_img


def interpolate(attr: torch.Tensor, ati: torch.Tensor, index_img: torch.Tensor, bary_img: torch.Tensor) -> torch.Tensor:
    """drtk.interpolate: per-vertex attributes attr[B,Va,C] with their own face table ati[F,3] -> [B,C,H,W]."""
    B, H, W = index_img.shape
    idx = index_img.clamp(min=0).long().reshape(B, -1)                      # [B,P]
    tri = ati.long()[idx]                                                    # [B,P,3]
    C = attr.shape[-1]
    vals = torch.gather(attr, 1, tri.reshape(B, -1, 1).expand(-1, -1, C)).reshape(B, -1, 3, C)   # [B,P,3,C]
    out = (vals * bary_img.reshape(B, 3, -1).permute(0, 2, 1)[..., None]).sum(2)  # [B,P,C]
    out = out * (index_img.reshape(B, -1, 1) >= 0)
    return out.permute(0, 2, 1).reshape(B, -1, H, W)


class RenderLayer(torch.nn.Module):
    """render_drtk.RenderLayer (ca_code/utils/render_drtk.py:14-82) with the same constructor, forward arguments and
    output dict."""

    def __init__(self, h, w, vi, vt, vti, flip_uvs=False):
        super().__init__()
        self.h, self.w = h, w
        self.register_buffer("vi", vi, persistent=False)
        self.register_buffer("vt", vt.clone(), persistent=False)
        self.register_buffer("vti", vti, persistent=False)
        self.flip_uvs = flip_uvs
        if flip_uvs:
            self.vt[:, 1] = 1 - self.vt[:, 1]
        self.register_buffer("image_size", torch.as_tensor([h, w], dtype=torch.int32))

    def forward(self, verts: torch.Tensor, tex: torch.Tensor, K: torch.Tensor, Rt: torch.Tensor,
                background: Optional[torch.Tensor] = None, output_filters: Optional[List[str]] = None,
                edge_grad: bool = True):
        assert output_filters is None
        assert background is None
        # Gradients: the texture gradient is plain autograd through grid_sample (vt_img comes from detached rasterizer
        # outputs); what is missing is drtk's edge_grad_estimator, i.e. the gradient w.r.t. the vertices -- a call that
        # needs it raises instead of silently dropping it.
        if torch.is_grad_enabled() and verts.requires_grad:
            raise NotImplementedError("goliath_amd.meshraster.RenderLayer has no edge-gradient estimator: gradients flow "
                                      "to `tex` only; pass verts.detach() (or call under torch.no_grad(), as the "
                                      "shadow-map path does, urhand.py:404,492)")
        v_pix = transform(verts, K=K, Rt=Rt)
        index_img, depth_img, bary_img = rasterize(v_pix, self.vi, self.h, self.w)
        vt_img = interpolate((self.vt * 2.0 - 1.0)[None].expand(verts.shape[0], -1, -1), self.vti, index_img, bary_img)
        mask = (index_img != -1)[:, None].float()
        img = F.grid_sample(tex, vt_img.permute(0, 2, 3, 1), mode="bilinear", align_corners=False) * mask
        return {"render": img, "depth_img": depth_img, "v_pix": v_pix, "vt_img": vt_img, "index_img": index_img,
                "bary_img": bary_img, "mask": mask}
