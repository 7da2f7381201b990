"""Generates tests/golden/urhand_model_golden.npz by running the REFERENCE's own model code on the CPU:
    ca_code.models.urhand.ConvTeacherDecoder.forward          (urhand.py:349-630)
    ca_code.models.hand_teacher_mvp.OLATRGBDecoder.forward_rgb (hand_teacher_mvp.py:253-494)
as unbound methods on the seeded stand-in modules of tests/urhand_shaped.py (same attribute layout, small sub-modules).
What the reference obtains from third-party / CUDA code is served by the CPU oracles: the shadow depth render
(drtk RenderLayer -> oracle/mesh_ref.py), the reference's OWN get_shadow_map (ca_code/utils/shadowmap.py, unchanged),
compute_raydirs and the with_shadow ray march (oracle/mvp_oracle.c), drtk.transform (restated in urhand_shaped.py).
Run in the build container (needs /root/reference):   python tests/golden/make_urhand_model_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import ref_stubs  # noqa: E402

ref_stubs.install()
for n in ("sgutilslib", "utilslib", "mvpraymarchlib"):
    sys.modules.setdefault(n, types.ModuleType(n))
import ca_code.models.hand_teacher_mvp as T  # noqa: E402
import ca_code.models.urhand as U  # noqa: E402
import urhand_shaped as S  # noqa: E402
from oracle import cref  # noqa: E402

# shadowmap.py:21 writes into `th.eye(3)[None].expand(...).to(device)`: on the GPU the .to() is a copy that materialises the
# expanded tensor; on the CPU it is a no-op and the in-place write is refused.  Emulate the device copy.
_orig_to = torch.Tensor.to
torch.Tensor.to = lambda self, *a, **k: (lambda r: r.contiguous() if r is self and 0 in self.stride() else r)(_orig_to(self, *a, **k))

out = {}
# ---------------------------------------------------------------- URHand (config 4): ConvTeacherDecoder.forward
for tag, training in (("eval", False), ("train", True)):
    dec = S.ShapedConvTeacherDecoder(seed=0)
    dec.rl = S.OracleRenderLayer(512, 512, dec.geo_fn.vi)
    dec.train(training)
    inp = S.urhand_inputs(B=1, L=2)
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("verts_rec", "tex_mean")}
    res = U.ConvTeacherDecoder.forward(dec, **{**inp, **leaves})
    keys = ("tex", "phys_tex", "diff_feature", "spec_feature", "diff_feature_raw", "spec_feature_raw", "shadow",
            "shadow_raw", "feature_normal", "verts_displaced", "displacement", "roughness", "id_pose_conv")
    for k in keys:
        out[f"urhand_{tag}_{k}"] = res[k].detach().numpy()
    for i, d in enumerate(dec.rl.rendered):     # the two shadow depth renders (float16-exact storage is not needed: npz compresses)
        out[f"urhand_{tag}_depth{i}"] = d.numpy()
    g = torch.Generator().manual_seed(5)
    loss = sum((res[k] * torch.randn(res[k].shape, generator=g)).sum() for k in ("tex", "phys_tex", "diff_feature_raw"))
    loss.backward()
    for k, v in leaves.items():
        out[f"urhand_{tag}_grad_{k}"] = v.grad.numpy()
    for n, p in dec.named_parameters():
        if n in ("global_scale", "global_albedo_scale", "geo_refiner.geo.weight", "texmod1.0.weight"):
            out[f"urhand_{tag}_grad_{n}"] = p.grad.numpy()

# ---------------------------------------------------------------- MVP teacher (config 5): OLATRGBDecoder.forward_rgb
T.transform = S.drtk_transform
T.compute_raydirs = lambda vp, vr, f, pp, pc, vol: cref.compute_raydirs(vp, vr, f, pp, pc, vol)
torch.Tensor.cuda = lambda self, *a, **k: self      # forward_rgb calls .cuda() on three linspaces (hand_teacher_mvp.py:378-380)
dec = S.ShapedOLATRGBDecoder(S.OracleRaymarcher(200.0), 200.0, seed=0).eval()
captured = {}
dec.enc_layers[0].register_forward_pre_hook(lambda m, a: captured.__setitem__("x", a[0].detach().clone()))
with torch.no_grad():
    res = T.OLATRGBDecoder.forward_rgb(dec, **S.teacher_inputs(B=1, L=2))
out["teacher_unet_input"] = captured["x"].numpy()       # [B*L, 7 Z, S, S] = light dirs, view dirs, 1 - deep shadow
out["teacher_primrgb"] = res["primrgb"].numpy()
out["teacher_primshadow"] = res["primshadow"].numpy()
np.savez_compressed(os.path.join(HERE, "urhand_model_golden.npz"), **out)
sh = out["urhand_eval_shadow"]
print("written", len(out), "arrays; shadow map range", float(sh.min()), float(sh.max()), "mean", float(sh.mean()),
      "| deep shadow range", float(out["teacher_primshadow"].min()), float(out["teacher_primshadow"].max()))
