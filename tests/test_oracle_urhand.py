"""CPU: oracle/urhand_ref.py vs golden vectors produced by exec()-ing the reference's own source lines
(urhand.py:419-445, 508-567).  PINNED."""
import os

import numpy as np
import pytest
import torch

from oracle import urhand_ref
from scenes import rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "urhand_golden.npz")


def load_golden():
    z = np.load(GOLD)
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.mark.parametrize("tag", ["sh", "nosh"])
def test_urhand_oracle_reproduces_reference(tag):
    G = load_golden()
    leaf = {k: G[f"in/{k}"].clone().requires_grad_(True) for k in ("p_uv", "nml", "roughness", "tex_mean")}
    sh = G["in/shadow_map"] if tag == "sh" else None
    d, s = urhand_ref.phong_features(leaf["p_uv"], leaf["nml"], G["in/cam_pos"], G["in/light_pos"],
                                     G["in/light_intensity"], sh)
    assert rel_l2(d, G[f"{tag}/phong/diff"]) < 1e-6 and rel_l2(s, G[f"{tag}/phong/spec"]) < 1e-6
    ((d * G[f"{tag}/phong/w_diff"]).sum() + (s * G[f"{tag}/phong/w_spec"]).sum()).backward()
    assert rel_l2(leaf["p_uv"].grad, G[f"{tag}/phong/g_p_uv"]) < 1e-5
    assert rel_l2(leaf["nml"].grad, G[f"{tag}/phong/g_nml"]) < 1e-5
    for t in leaf.values():
        t.grad = None
    f, rgb = urhand_ref.ggx_features(leaf["p_uv"], leaf["nml"], G["in/cam_pos"], G["in/light_pos"],
                                     G["in/light_intensity"], leaf["roughness"], leaf["tex_mean"], sh)
    gf = G[f"{tag}/ggx/feat"].reshape(f.shape)
    assert rel_l2(f, gf) < 1e-6 and rel_l2(rgb, G[f"{tag}/ggx/rgb"]) < 1e-6
    ((f * G[f"{tag}/ggx/w_feat"].reshape(f.shape)).sum() + (rgb * G[f"{tag}/ggx/w_rgb"]).sum()).backward()
    for k, n in (("p_uv", "g_p_uv"), ("nml", "g_nml"), ("roughness", "g_roughness"), ("tex_mean", "g_tex")):
        assert rel_l2(leaf[k].grad, G[f"{tag}/ggx/{n}"]) < 1e-5, k


def test_shadow_pcf_oracle_matches_reference_golden():
    import os

    import numpy as np
    import torch

    from oracle import urhand_ref

    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "shadow_golden.npz"))
    for tag in ("a", "b"):
        c = {k.split("/")[1]: torch.from_numpy(G[k]) for k in G.files if k.startswith(tag + "/")}
        got = urhand_ref.shadow_pcf(c["depth"], c["Rt"], c["postex"], c.get("nml"))
        # nearest-neighbour lookups: allow a handful of texels to pick the other side of a rounding tie
        bad = ((got - c["out"]).abs() > 1e-3 * (1 + c["out"].abs())).float().mean()
        assert float(bad) < 0.005, tag
