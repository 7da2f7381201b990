"""Generate tests/golden/shadow_golden.npz by running the REFERENCE's get_shadow_map
(/root/reference/ca_code/utils/shadowmap.py:17-96) on CPU with a fake render layer that returns a fixed depth
image (the drtk mesh rasteriser is third-party and absent here; cv2 / render_drtk are stubbed).  Build container only."""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


class _Any(types.ModuleType):
    def __getattr__(self, name):
        return 0


sys.modules["cv2"] = _Any("cv2")
for name, attrs in (("pytorch3d", {}), ("pytorch3d.renderer", {}), ("pytorch3d.renderer.mesh", {}),
                    ("pytorch3d.renderer.mesh.rasterize_meshes", {"rasterize_meshes": None}),
                    ("pytorch3d.structures", {"Meshes": None})):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
rd = types.ModuleType("ca_code.utils.render_drtk")
rd.RenderLayer = object
sys.modules["ca_code.utils.render_drtk"] = rd
from ca_code.utils.shadowmap import get_shadow_map  # noqa: E402


class FakeRL:
    def __init__(self, depth):
        self.depth, self.h, self.w = depth, depth.shape[-2], depth.shape[-1]

    def __call__(self, verts, tex, K, Rt):
        return {"depth_img": self.depth}


def look_at(pos):
    z = torch.nn.functional.normalize(-pos, dim=-1)
    up = torch.tensor([0.0, 1.0, 0.0]).expand_as(pos)
    x = torch.nn.functional.normalize(torch.linalg.cross(up, z), dim=-1)
    y = torch.linalg.cross(z, x)
    return torch.stack([x, y, z], 1)


def main():
    # shadowmap.py:21 writes into `th.eye(3)[None].expand(...).to(device)`: on the GPU the .to() is a copy that
    # materialises the expanded tensor; on the CPU it is a no-op and the in-place write is refused.  Emulate the
    # device copy while the reference function runs.
    orig_to = torch.Tensor.to

    def to_materialised(self, *a, **k):
        r = orig_to(self, *a, **k)
        return r.contiguous() if r is self and 0 in self.stride() else r

    torch.Tensor.to = to_materialised
    out = {}
    for tag, (BL, S, hw, use_n) in {"a": (3, 24, 64, True), "b": (2, 17, 48, False)}.items():
        g = torch.Generator().manual_seed(ord(tag))
        lightpos = 600 * torch.nn.functional.normalize(torch.randn(BL, 3, generator=g), dim=-1)
        R = look_at(lightpos)
        Rt = torch.cat([R, lightpos[..., None]], 2)                     # the reference's convention (urhand.py:415)
        # texels in front of the light camera under p_cam = R p + t: p = R^T (q - t) with q.z > 0
        q = torch.stack([40 * torch.randn(BL, S * S, generator=g), 40 * torch.randn(BL, S * S, generator=g),
                         700 + 100 * torch.rand(BL, S * S, generator=g)], -1)
        p = torch.einsum("bji,bnj->bni", R, q - lightpos[:, None])
        postex = p.permute(0, 2, 1).reshape(BL, 3, S, S).contiguous()
        nml = torch.nn.functional.normalize(torch.randn(BL, 3, S, S, generator=g), dim=1) if use_n else None
        depth = 650 + 150 * torch.rand(BL, hw, hw, generator=g)
        depth[torch.rand(BL, hw, hw, generator=g) < 0.25] = 0.0         # holes (no geometry)
        verts = torch.randn(BL, 10, 3, generator=g)
        val = get_shadow_map(FakeRL(depth), Rt, None, verts, postex.clone(), nml)
        out[f"{tag}/Rt"], out[f"{tag}/postex"], out[f"{tag}/depth"] = Rt.numpy(), postex.numpy(), depth.numpy()
        if nml is not None:
            out[f"{tag}/nml"] = nml.numpy()
        out[f"{tag}/out"] = val.numpy()
        print(tag, val.shape, float(val.min()), float(val.max()), float((val > 0).float().mean()))
    torch.Tensor.to = orig_to
    np.savez_compressed(os.path.join(HERE, "shadow_golden.npz"), **out)


if __name__ == "__main__":
    main()
