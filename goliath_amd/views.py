"""The cameras of a batch of views as the shading tail and the renderer share them (include/goliath_hip.h: gol_shade_proj).

AutoEncoder.forward knows K and the head-relative Rt of every view BEFORE it runs the decoder (ca_code/models/rgca.py:175-195),
so the shading kernel can project the Gaussians it has just produced (gol_shade_project_fwd) and AutoEncoder.render
(rgca.py:112-151 -> ca_code/utils/render_gsplat.py:49-104) starts at the tile count.  A ViewSet carries the cameras from
the one call to the other; `Projected` is what the shading call leaves for the render call.
"""
import ctypes

import torch

from . import _lib

_fp = ctypes.c_void_p


class ShadeProj(ctypes.Structure):
    """include/goliath_hip.h: gol_shade_proj."""
    _fields_ = [("viewmats", _fp), ("intrins", _fp), ("img_h", ctypes.c_int32), ("img_w", ctypes.c_int32),
                ("glob_scale", ctypes.c_float), ("clip_thresh", ctypes.c_float), ("xys", _fp), ("depths", _fp),
                ("radii", _fp), ("conics", _fp), ("comp", _fp), ("opac_eff", _fp), ("records", _fp)]


SPLAT_RECORD = 16   # include/goliath_hip.h: GOL_SPLAT_RECORD
PACK_FLOATS = 9     # xy 2, depth 1, radius 1 (int32), conic 3, compensation 1, effective opacity 1


class ViewSet:
    """K[B,3,3], Rt[B,3,4] (world -> camera; the tensors AutoEncoder.render will be called with), image size, gsplat's
    global scale and near clip (render_gsplat.py:28-30).  No host sync: intrinsics and matrices stay on the device."""

    def __init__(self, K, Rt, height, width, glob_scale=1.0, clip_thresh=0.1):
        self.K, self.Rt = K, Rt
        self.height, self.width = int(height), int(width)
        self.glob_scale, self.clip_thresh = float(glob_scale), float(clip_thresh)
        B = K.shape[0]
        with torch.no_grad():
            self.intrins = torch.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], dim=-1).to(torch.float32).contiguous()
            self.viewmats = Rt.to(torch.float32).reshape(B, -1)[:, :12].contiguous()

    def matches(self, K, Rt, height, width):
        return K is self.K and Rt is self.Rt and (int(height), int(width)) == (self.height, self.width)


class Projected:
    """Output of a shading call with a ViewSet: records[B,N,16] (differentiable: its gradient is the raster backward's
    gradient records) and pack[9,B*N] (screen position, depth, radius, conic, compensation, effective opacity)."""

    SOURCES = ("primpos", "primscale", "primqvec", "opacity", "color")

    def __init__(self, views, records, pack, sources):
        self.views, self.records, self.pack = views, records, pack
        self.sources = {k: sources[k] for k in self.SOURCES}   # the tensors the records were computed from

    def valid_for(self, preds, K, Rt, height, width):
        """Do these records describe what render(K, Rt, preds) would project?  Same cameras, and the five attribute
        tensors are still the ones the shading call returned (a caller that swaps preds["color"], as the diffuse /
        specular breakdown renders of rgca.py:232-245 do, gets the separate projection)."""
        return self.views.matches(K, Rt, height, width) and all(preds.get(k) is v for k, v in self.sources.items())

    def field(self, name):
        B, N = self.records.shape[:2]
        BN = B * N
        off, k, dt = {"xys": (0, 2, torch.float32), "depths": (2, 1, torch.float32), "radii": (3, 1, torch.int32),
                      "conics": (4, 3, torch.float32), "comp": (7, 1, torch.float32),
                      "opac_eff": (8, 1, torch.float32)}[name]
        v = self.pack.reshape(-1)[off * BN:(off + k) * BN]
        v = v.view(dt) if dt != torch.float32 else v
        return v.view(B, N, k) if k > 1 else v.view(B, N)


def proj_struct(views, records, pack):
    """gol_shade_proj over a ViewSet and the two buffers of a `Projected`."""
    B, N = records.shape[:2]
    BN = B * N
    base = pack.data_ptr()
    s = ShadeProj()
    s.viewmats, s.intrins = _lib.fptr(views.viewmats).value, _lib.fptr(views.intrins).value
    s.img_h, s.img_w = views.height, views.width
    s.glob_scale, s.clip_thresh = views.glob_scale, views.clip_thresh
    s.xys, s.depths, s.radii = base, base + 8 * BN, base + 12 * BN
    s.conics, s.comp, s.opac_eff = base + 16 * BN, base + 28 * BN, base + 32 * BN
    s.records = _lib.fptr(records).value
    return s
