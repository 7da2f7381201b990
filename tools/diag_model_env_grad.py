"""Diagnostic (GPU): the env-path shading backward on the eval_env case of the model fixture -- HIP vs the fp32 oracle vs the
same oracle in fp64, identical decoder outputs and upstream gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.nn.functional as F

import rgca_shaped as S
from goliath_amd import decoder as D, shade
from oracle import shade_ref

D._wn = lambda v, g: v * (g / v.double().pow(2).sum().sqrt().to(v.dtype))
G = np.load(os.path.join(ROOT, "tests", "golden", "rgca_model_golden.npz"))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
tag, B, seed = "eval_env", 2, 100
st = {k.split("/stored/")[1]: G[k] for k in G.files if k.startswith(f"{tag}/stored/")}
embs, geom = S.leaves(B, seed, st)
m = S.ShapedAutoEncoder(embs, geom, 0, nudges=(G["nudges/index"], G["nudges/dz"]))
dec = m.decoder.eval()
batch = S.batch_inputs(B, seed, stored=st)
hp = batch["head_pose"]
rot, trans = hp[:, :3, :3], hp[:, :3, 3]
campos = ((batch["campos"] - trans)[:, None] @ rot)[:, 0]
with torch.no_grad():
    postex = dec.geo_fn.to_uv(geom)
    tn = F.normalize(dec.geo_fn.to_uv(dec.geo_fn.vn(geom)), dim=1)
    z = dec.encmod(embs).view(-1, 256, 8, 8)
    view = dec.viewmod(F.normalize(campos, dim=1))[:, :, None, None].expand(-1, -1, 8, 8)
    f_vn, f_vc = dec.vnocond_mod(z), dec.vcond_mod(torch.cat([z, view], 1))
light_sh = t(G[f"{tag}/out/headrel_light_sh"])
lightrot = t(G[f"{tag}/in/lightrot"]) @ rot
mips = [t(G[f"{tag}/in/preconv_envmap_{i}"]) for i in range(4)]
N = S.S * S.S
gen = torch.Generator().manual_seed(0)
up = torch.randn(B, N, 3, generator=gen)
leaves = ("f_vn", "f_vc", "postex", "tn")
src = dict(f_vn=f_vn, f_vc=f_vc, postex=postex, tn=tn)


def oracle(dtype):
    x = {k: v.detach().to(dtype).requires_grad_(True) for k, v in src.items()}
    pr = shade_ref.shade(x["f_vn"], x["f_vc"], x["postex"], x["tn"], dec.albedo.detach().to(dtype), light_sh.to(dtype), campos.to(dtype),
                         envmips=[mm.to(dtype) for mm in mips], lightrot=lightrot.to(dtype))
    (pr["color"] * up.to(dtype)).sum().backward()
    return {k: x[k].grad.double().reshape(B, -1, N) for k in leaves}, pr


g32, pr32 = oracle(torch.float32)
g64, _ = oracle(torch.float64)
x = {k: v.detach().cuda().requires_grad_(True) for k, v in src.items()}
for shared in (True, False):
    for k in x:
        x[k].grad = None
    mm = [q.cuda() if shared else q.cuda().expand(B, -1, -1, -1).contiguous() for q in mips]
    ph = shade.shading_tail(x["f_vn"], x["f_vc"], x["postex"], x["tn"], dec.albedo.detach().cuda(), light_sh.cuda(), campos.cuda(),
                            preconv_envmap=mm, lightrot=lightrot.cuda())
    (ph["color"] * up.cuda()).sum().backward()
    gh = {k: x[k].grad.double().cpu().reshape(B, -1, N) for k in leaves}
    print("shared pyramid" if shared else "per-view copies")
    with torch.no_grad():
        v = F.normalize(pr32["primpos"] - campos[:, None], dim=-1)
        n = pr32["spec_nml"]
        refl = v - 2 * (v * n).sum(-1, keepdim=True) * n
        r = torch.einsum("bxy,bny->bnx", lightrot, refl)
        uv = shade_ref.dir2uv(r)
    for k in leaves:
        e_h = (gh[k] - g32[k]).pow(2).sum(1).flatten()
        e_o = (g32[k] - g64[k]).pow(2).sum(1).flatten()
        e_h64 = (gh[k] - g64[k]).pow(2).sum(1).flatten()
        tot = g64[k].pow(2).sum()
        top = e_h.topk(8).indices
        print(f"  {k:7s} hip-vs-o32 {float((e_h.sum() / tot).sqrt()):.2e}  o32-vs-o64 {float((e_o.sum() / tot).sqrt()):.2e}  hip-vs-o64 "
              f"{float((e_h64.sum() / tot).sqrt()):.2e}  top8 share {float(e_h[top].sum() / e_h.sum()):.2f}")
        if k == "f_vc":
            for i in top.tolist():
                b, j = divmod(i, N)
                lvl = float(pr32["sigma"][b, j] * 5)
                w0, h0 = 128 >> int(lvl), 64 >> int(lvl)
                ix, iy = ((float(uv[b, j, 0]) + 1) * w0 - 1) / 2, ((float(uv[b, j, 1]) + 1) * h0 - 1) / 2
                print(f"     g {i}: hip-o32 {float(e_h[i].sqrt()):.3e} o32-o64 {float(e_o[i].sqrt()):.3e} hip-o64 {float(e_h64[i].sqrt()):.3e} "
                      f"|g| {float(g64[k].reshape(B, -1, N)[b, :, j].norm()):.3e} r_y {float(r[b, j, 1]):.4f} level {lvl:.3f} ix {ix:.4f} iy {iy:.4f}")
