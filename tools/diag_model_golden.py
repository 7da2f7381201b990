"""Diagnostic (GPU box): where does the model-level RGCA path leave the fixture?  Layer-by-layer comparison of the stand-in's
decoder ladder on the GPU (PyTorch-ROCm / MIOpen) against the same ladder on the CPU, then the fused tail against the unfused
one.  Usage: python tools/diag_model_golden.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.nn.functional as F

import rgca_shaped as S
from scenes import rel_l2

G = np.load(os.path.join(ROOT, "tests", "golden", "rgca_model_golden.npz"))
embs, geom = S.leaves(2, 0)
m = S.ShapedAutoEncoder(embs, geom, 0)
dec = m.decoder.eval()
batch = S.batch_inputs(2, 0)
hp = batch["head_pose"]
rot, trans = hp[:, :3, :3], hp[:, :3, 3]
campos = ((batch["campos"] - trans)[:, None] @ rot)[:, 0]


def ladder(dec, embs, campos):
    acts = {}
    z = dec.encmod(embs).view(-1, 256, 8, 8)
    acts["z"] = z
    view = dec.viewmod(F.normalize(campos, dim=1))[:, :, None, None].expand(-1, -1, 8, 8)
    acts["view"] = view
    x = z
    for i, layer in enumerate(dec.vnocond_mod):
        x = layer(x)
        acts[f"vnocond.{i}"] = x.clone()
    x = torch.cat([z, view], 1)
    for i, layer in enumerate(dec.vcond_mod):
        x = layer(x)
        acts[f"vcond.{i}"] = x.clone()
    return acts


with torch.no_grad():
    cpu = ladder(dec, embs, campos)
    import copy

    dg = copy.deepcopy(dec).cuda()
    gpu = ladder(dg, embs.cuda(), campos.cuda())
    for k in cpu:
        print(f"{k:12s} gpu-vs-cpu rel_l2 {rel_l2(gpu[k], cpu[k]):.2e}   max-abs {float((gpu[k].cpu() - cpu[k]).abs().max()):.2e}  (|x| max {float(cpu[k].abs().max()):.2f})")
    for flag in ("1", "0"):
        torch.backends.cudnn.allow_tf32 = flag == "1"
        torch.backends.cuda.matmul.allow_tf32 = flag == "1"
        g2 = ladder(dg, embs.cuda(), campos.cuda())
        print("allow_tf32", flag, "vnocond.4", f"{rel_l2(g2['vnocond.4'], cpu['vnocond.4']):.2e}")
    # fp64 yardstick of the CPU ladder
    d64 = copy.deepcopy(dec).double()
    c64 = ladder(d64, embs.double(), campos.double())
    for k in ("z", "vnocond.0", "vnocond.2", "vnocond.4", "vcond.4"):
        print(f"{k:12s} cpu32-vs-fp64 {rel_l2(cpu[k], c64[k]):.2e}   gpu32-vs-fp64 {rel_l2(gpu[k], c64[k]):.2e}")
    # direct conv_transpose of the last layer on identical inputs
    last = dg.vnocond_mod[-1]
    x = cpu["vnocond.3"].cuda()
    y_gpu = F.conv_transpose2d(x, last.weight(), None, 2, 1) + last.bias[None]
    y_cpu = F.conv_transpose2d(cpu["vnocond.3"], dec.vnocond_mod[-1].weight(), None, 2, 1) + dec.vnocond_mod[-1].bias[None]
    print("last layer alone, identical input: gpu-vs-cpu", f"{rel_l2(y_gpu, y_cpu):.2e}")
