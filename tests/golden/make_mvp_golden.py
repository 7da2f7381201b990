"""Generate tests/golden/mvp_golden.npz from the REFERENCE's own in-tree PyTorch ray marcher
(/root/reference/extensions/mvpraymarch/mvpraymarch.py:446-669, the `gradcheck()` self-check) and
tests/golden/raydirs_golden.npz from /root/reference/extensions/utils/utils.py:78-143.

Run in the build container only.  The reference functions hard-code device "cuda" and a large
problem (N=2, 65x65 rays, 64 primitives of 32^3 voxels = 67 MB of template): this script reads the
reference source text at run time, substitutes the device literal and the size constants, cuts the
function where its CUDA half begins, and executes the reference's PyTorch half unchanged on CPU.
Nothing from the reference is copied into the repository.  Recipe kept from the reference:
torch.manual_seed(1112), coherent pinhole rays, coherent centres, Rodrigues rotations,
usebvh="fixedorder", sortprims=False, chlast=True, fadescale=6.5, fadeexp=7.5, accum=0, algo=0; set "w": the same recipe
with dowarp=True, algo=1 (the second gradcheck of the reference's __main__, mvpraymarch.py:790-803).
"""
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/extensions"


def load_patched(path, subs, cut_marker, lib_names):
    src = open(path).read()
    src = src.replace('"cuda"', '"cpu"').replace("torch.cuda.synchronize()", "pass")
    for a, b in subs:
        assert re.search(a, src), a
        src = re.sub(a, b, src)
    head, tail = src.split(cut_marker, 1)
    rest = tail[tail.index("\nif __name__"):]
    src = head + "    return dict(locals())\n" + rest
    for n in lib_names:
        sys.modules[n] = types.ModuleType(n)
    ns = {"__name__": "reference_patched"}
    exec(compile(src, path, "exec"), ns)
    return ns


def mvp():
    ns = load_patched(
        f"{REF}/mvpraymarch/mvpraymarch.py",
        [(r"\n    H = 65\n", "\n    H = 24\n"), (r"\n    W = 65\n", "\n    W = 20\n"), (r"\n    M = 32\n", "\n    M = 8\n")],
        "    ############################## run cuda version", ["mvpraymarchlib"])
    out = {}
    for tag, fs, fe in (("a", 6.5, 7.5), ("b", 8.0, 8.0)):
        L = ns["gradcheck"](usebvh="fixedorder", sortprims=False, maxhitboxes=512, synchitboxes=True, dowarp=False,
                            chlast=True, fadescale=fs, fadeexp=fe, accum=0, algo=0, griddim=3)
        g = dict(zip(L["paramnames"], L["grads0"]))
        out.update({
            f"{tag}/raypos": L["_raypos"], f"{tag}/raydir": L["_raydir"], f"{tag}/tminmax": L["_tminmax"],
            f"{tag}/stepsize": torch.tensor(L["_stepsize"]), f"{tag}/fade": torch.tensor([fs, fe]),
            f"{tag}/leaf_template": L["_template"].detach(), f"{tag}/leaf_primpos": L["_primpos"].detach(),
            f"{tag}/leaf_primrot": L["_primrot"].detach(), f"{tag}/leaf_primscale": L["_primscale"].detach(),
            f"{tag}/rayrgba": L["sample0"].detach(),
            f"{tag}/grad_template": g["template"], f"{tag}/grad_primpos": g["primpos"],
            f"{tag}/grad_primrot": g["primrot"], f"{tag}/grad_primscale": g["primscale"],
        })
        print(tag, "alpha range", float(L["sample0"][..., 3].min()), float(L["sample0"][..., 3].max()))
    # warp fields (algo 1, mvpraymarch.py:777-803 second gradcheck): warp [N,K,3,M/2,M/2,M/2] = identity grid + noise
    L = ns["gradcheck"](usebvh="fixedorder", sortprims=False, maxhitboxes=512, synchitboxes=True, dowarp=True,
                        chlast=True, fadescale=6.5, fadeexp=7.5, accum=0, algo=1, griddim=3)
    g = dict(zip(L["paramnames"], L["grads0"]))
    out.update({
        "w/raypos": L["_raypos"], "w/raydir": L["_raydir"], "w/tminmax": L["_tminmax"],
        "w/stepsize": torch.tensor(L["_stepsize"]), "w/fade": torch.tensor([6.5, 7.5]),
        "w/leaf_template": L["_template"].detach(), "w/leaf_warp": L["_warp"].detach(),
        "w/leaf_primpos": L["_primpos"].detach(), "w/leaf_primrot": L["_primrot"].detach(),
        "w/leaf_primscale": L["_primscale"].detach(), "w/rayrgba": L["sample0"].detach(),
        "w/grad_template": g["template"], "w/grad_warp": g["warp"], "w/grad_primpos": g["primpos"],
        "w/grad_primrot": g["primrot"], "w/grad_primscale": g["primscale"],
    })
    print("w alpha range", float(L["sample0"][..., 3].min()), float(L["sample0"][..., 3].max()))
    path = os.path.join(HERE, "mvp_golden.npz")
    np.savez_compressed(path, **{k: v.detach().cpu().numpy().astype(np.float32) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path))


def raydirs():
    ns = load_patched(f"{REF}/utils/utils.py", [], "    ############################## run cuda version", ["utilslib"])
    L = ns["gradcheck"]()
    out = {"viewpos": L["_viewpos"], "viewrot": L["_viewrot"], "focal": L["_focal"], "princpt": L["_princpt"],
           "pixelcoords": L["_pixelcoords"], "raydir": L["sample0"], "tminmax": L["tminmax"]}
    path = os.path.join(HERE, "raydirs_golden.npz")
    np.savez_compressed(path, **{k: v.detach().cpu().numpy().astype(np.float32) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    mvp()
    raydirs()
