"""Diagnostic: does the CPU oracle give bit-identical results on two hosts?  Prints checksums of every stage for view 1 of the
train_point case of the model fixture."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import rgca_shaped as S
from oracle import cref

G = np.load(os.path.join(ROOT, "tests", "golden", "rgca_model_golden.npz"))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
h = lambda x: hashlib.sha1(np.ascontiguousarray(x.numpy()).tobytes()).hexdigest()[:12]
batch = S.batch_inputs(2, 0)
hp = batch["head_pose"]
bottom = torch.tensor([[[0.0, 0.0, 0.0, 1.0]]]).expand(2, -1, -1)
hRt = batch["Rt"] @ torch.cat([hp, bottom], 1)
K = batch["K"]
print("inputs", h(hRt), h(K), h(batch["background"]))
b = 1
g = {k: t(G[f"train_point/out/{k}"])[b].contiguous() for k in ("primpos", "primqvec", "primscale", "opacity", "color")}
print("attrs", {k: h(v) for k, v in g.items()})
xys, depths, radii, conics, comp, nth, cov = cref.project_gaussians(g["primpos"], g["primscale"], 1.0, g["primqvec"], hRt[b], float(K[b, 0, 0]),
                                                                     float(K[b, 1, 1]), float(K[b, 0, 2]), float(K[b, 1, 2]), S.H, S.W, 16, 0.1)
print("project", h(xys), h(depths), h(radii), h(conics), h(comp), h(nth))
keys, ids, bins = cref.bin_and_sort(xys, depths, radii, nth, S.H, S.W, 16)
print("sort", h(keys), h(ids), h(bins))
op = (g["opacity"] * comp[:, None]).contiguous()
print("opac", h(op))
img, Ts, idx = cref.rasterize_forward(ids, bins, xys, conics, g["color"], op, S.H, S.W, 16, torch.zeros(3))
print("raster", h(img), h(Ts), h(idx))
tile = 9 * ((S.W + 15) // 16) + 10
lo, hi = int(bins[tile, 0]), int(bins[tile, 1])
print("tile (9,10) ids", ids[lo:hi].tolist()[:40])
print("pixel (148,171)", img[148, 171].tolist(), float(Ts[148, 171]), int(idx[148, 171]))
