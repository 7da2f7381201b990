"""Generate tests/golden/urhand_golden.npz by executing the REFERENCE's own lines of the URHand UV
light loops on CPU:
    /root/reference/ca_code/models/urhand.py:419-445   (Lambert + Phong^{1,16,32} features)
    /root/reference/ca_code/models/urhand.py:508-567   (GGX/Schlick features + physically based texture)
The surrounding method needs a mesh, drtk and pytorch3d; the two blocks only need a handful of
tensors, so this script reads the source lines at run time, dedents them and exec()s them in a
namespace holding seeded inputs (nothing is copied into the repository).  Run in the build container.
"""
import os
import textwrap
import types

import numpy as np
import torch as th
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/ca_code/models/urhand.py"


def block(lo, hi):
    lines = open(SRC).read().split("\n")[lo - 1:hi]
    return textwrap.dedent("\n".join(lines))


def main():
    th.manual_seed(4242)
    B, L, S = 2, 5, 12
    p_uv = (40 * th.randn(B, 3, S, S)).requires_grad_(True)
    nml_raw = th.randn(B, 3, S, S)
    nml = F.normalize(nml_raw, dim=1).requires_grad_(True)
    cam_pos = th.tensor([[50.0, -30.0, -800.0], [-300.0, 80.0, -650.0]])
    light_pos = F.normalize(th.randn(B, L, 3), dim=-1) * 1100.0
    light_intensity = (th.rand(B, L, 1) + 0.1)
    shadow_map = th.rand(B, L, 1, S, S)
    roughness = (0.1 + 0.8 * th.rand(B, 1, S, S)).requires_grad_(True)
    tex_mean = (255 * th.rand(B, 3, S, S)).requires_grad_(True)
    out = {}
    gen = th.Generator().manual_seed(3)

    def tbn_from(n):  # the blocks read the normal as tbn_rot_uv[:, :, :, 2:] permuted
        t = th.zeros(B, S, S, 3, 3)
        return th.cat([t[:, :, :, :2], n.permute(0, 2, 3, 1)[:, :, :, None, :]], dim=3)

    for shadow in (True, False):
        tag = "sh" if shadow else "nosh"
        self = types.SimpleNamespace(spec_powers=[1, 16, 32], shadow=shadow, fresnel=0.04, scaled_albedo=False)
        # ---- Phong block
        for t in (p_uv, nml, roughness, tex_mean):
            t.grad = None
        ns = dict(th=th, F=F, np=np, self=self, light_pos=light_pos, p_uv=p_uv,
                  v_uv=F.normalize(cam_pos[..., None, None] - p_uv, dim=1), tbn_rot_uv=tbn_from(nml),
                  light_intensity=light_intensity[..., None, None], shadow_map=shadow_map, lightmap=None)
        exec(block(419, 445), ns)
        d, s = ns["outputs"]["diff_feature_raw"], ns["outputs"]["spec_feature_raw"]
        wd, ws = th.randn(d.shape, generator=gen), th.randn(s.shape, generator=gen)
        ((d * wd).sum() + (s * ws).sum()).backward()
        out.update({f"{tag}/phong/diff": d, f"{tag}/phong/spec": s, f"{tag}/phong/w_diff": wd, f"{tag}/phong/w_spec": ws,
                    f"{tag}/phong/g_p_uv": p_uv.grad.clone(), f"{tag}/phong/g_nml": nml.grad.clone()})
        # ---- GGX block
        for t in (p_uv, nml, roughness, tex_mean):
            t.grad = None
        ns = dict(th=th, F=F, np=np, self=self, light_pos=light_pos, p_uv=p_uv,
                  v_uv=F.normalize(cam_pos[..., None, None] - p_uv, dim=1), tbn_rot_uv=tbn_from(nml),
                  light_intensity=light_intensity[..., None, None], shadow_map=shadow_map, roughness=roughness,
                  tex_mean=tex_mean)
        exec(block(508, 567), ns)
        f, rgb = ns["feat_p"], ns["rgb"]
        wf, wr = th.randn(f.shape, generator=gen), th.randn(rgb.shape, generator=gen)
        ((f * wf).sum() + (rgb * wr).sum()).backward()
        out.update({f"{tag}/ggx/feat": f, f"{tag}/ggx/rgb": rgb, f"{tag}/ggx/w_feat": wf, f"{tag}/ggx/w_rgb": wr,
                    f"{tag}/ggx/g_p_uv": p_uv.grad.clone(), f"{tag}/ggx/g_nml": nml.grad.clone(),
                    f"{tag}/ggx/g_roughness": roughness.grad.clone(), f"{tag}/ggx/g_tex": tex_mean.grad.clone()})
    out.update({"in/p_uv": p_uv, "in/nml": nml, "in/cam_pos": cam_pos, "in/light_pos": light_pos,
                "in/light_intensity": light_intensity, "in/shadow_map": shadow_map, "in/roughness": roughness,
                "in/tex_mean": tex_mean})
    path = os.path.join(HERE, "urhand_golden.npz")
    np.savez_compressed(path, **{k: v.detach().numpy().astype(np.float32) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
