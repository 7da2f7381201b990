"""GPU: gol_shadow_pcf (goliath_amd.shadowmap) vs the golden vectors made by the reference's get_shadow_map."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "shadow_golden.npz")


def _cases():
    G = np.load(GOLD)
    for tag in ("a", "b"):
        yield tag, {k.split("/")[1]: torch.from_numpy(G[k]).cuda() for k in G.files if k.startswith(tag + "/")}


def _close(a, b, frac=0.005):
    # nearest-neighbour lookups: a rounding tie may fall on the other side for a handful of texels
    bad = ((a - b).abs() > 1e-3 * (1 + b.abs())).float().mean()
    return float(bad) < frac


def test_shadow_pcf_matches_reference_golden():
    from goliath_amd import shadowmap

    for tag, c in _cases():
        got = shadowmap.shadow_pcf(c["depth"], c["Rt"], c["postex"], c.get("nml"))
        assert got.shape == c["out"].shape
        assert _close(got, c["out"]), tag
        # the same comparison as ONE number (parity ledger): rel-L2 over all texels; measured 6.7e-7 -- no tie texel in
        # these two cases (the _close() allowance above is for scenes that have one)
        from scenes import rel_l2

        assert rel_l2(got, c["out"]) < 1e-5, tag
        fused = shadowmap.shadow_pcf(c["depth"], c["Rt"], c["postex"], c.get("nml"), exp_scale=8.0)
        assert torch.allclose(fused, torch.exp(-got / 8.0), atol=1e-6)


def test_get_shadow_map_dropin_and_light_inner_loop():
    from goliath_amd import shadowmap

    tag, c = next(_cases())

    class RL:
        h, w = c["depth"].shape[-2:]

        def __call__(self, verts, tex, K, Rt):
            assert float(K[0, 0, 0]) == 1000.0 and float(K[0, 0, 2]) == self.w / 2
            return {"depth_img": c["depth"]}

    got = shadowmap.get_shadow_map(RL(), c["Rt"], None, torch.zeros(3, 10, 3).cuda(), c["postex"], c["nml"])
    assert _close(got, c["out"])
    # native form: one set of texels, L light cameras -> same as repeating the texels per light
    p0, n0 = c["postex"][:1], c["nml"][:1]
    L = c["Rt"].shape[0]
    a = shadowmap.shadow_pcf(c["depth"], c["Rt"], p0, n0)
    b = shadowmap.shadow_pcf(c["depth"], c["Rt"], p0.expand(L, -1, -1, -1).contiguous(), n0.expand(L, -1, -1, -1).contiguous())
    assert torch.equal(a, b)
