"""CPU: the host logic that carries the cameras from the shading call to the renderer (goliath_amd/views.py) -- when the
renderer may take the records the shading kernel wrote, and how the packed projection buffer is laid out."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from goliath_amd import views  # noqa: E402


def _cams(B=3):
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = 900.0, 910.0, 100.5, 80.25, 1.0
    Rt = torch.randn(B, 3, 4)
    return K, Rt


def test_view_set_keeps_intrinsics_and_matrices_on_the_tensors_device_without_a_sync():
    K, Rt = _cams()
    vs = views.ViewSet(K, Rt, 160, 208)
    assert vs.intrins.shape == (3, 4) and vs.viewmats.shape == (3, 12)
    assert torch.equal(vs.intrins[0], torch.tensor([900.0, 910.0, 100.5, 80.25]))
    assert torch.equal(vs.viewmats, Rt.reshape(3, 12))
    # a [B,4,4] pose is cut to its upper 3x4 (render_gsplat.py:43-47 takes Rt[:3])
    Rt44 = torch.cat([Rt, torch.tensor([0.0, 0, 0, 1]).expand(3, 1, 4)], 1)
    assert torch.equal(views.ViewSet(K, Rt44, 160, 208).viewmats, Rt.reshape(3, 12))


def test_only_the_same_cameras_and_image_size_match():
    K, Rt = _cams()
    vs = views.ViewSet(K, Rt, 160, 208)
    assert vs.matches(K, Rt, 160, 208)
    assert not vs.matches(K.clone(), Rt, 160, 208)      # identity, not value: no device read to find out
    assert not vs.matches(K, Rt.clone(), 160, 208)
    assert not vs.matches(K, Rt, 208, 160)


def test_records_are_only_valid_for_the_attribute_tensors_they_were_computed_from():
    K, Rt = _cams(2)
    vs = views.ViewSet(K, Rt, 64, 64)
    N = 5
    preds = {"primpos": torch.randn(2, N, 3), "primscale": torch.rand(2, N, 3), "primqvec": torch.randn(2, N, 4),
             "opacity": torch.rand(2, N, 1), "color": torch.rand(2, N, 3), "diff_color": torch.rand(2, N, 3)}
    pr = views.Projected(vs, torch.zeros(2, N, views.SPLAT_RECORD), torch.zeros(views.PACK_FLOATS, 2 * N), preds)
    preds["projected"] = pr
    assert pr.valid_for(preds, K, Rt, 64, 64)
    swapped = dict(preds, color=preds["diff_color"].clamp(min=0.0))   # rgca.py:232-245
    assert not pr.valid_for(swapped, K, Rt, 64, 64)
    assert not pr.valid_for({k: v for k, v in preds.items() if k != "opacity"}, K, Rt, 64, 64)
    assert not pr.valid_for(preds, K, Rt.clone(), 64, 64)


def test_pack_fields_are_contiguous_slices_with_the_abi_offsets():
    K, Rt = _cams(2)
    vs = views.ViewSet(K, Rt, 64, 64)
    B, N = 2, 7
    pack = torch.arange(views.PACK_FLOATS * B * N, dtype=torch.float32).reshape(views.PACK_FLOATS, B * N)
    src = {k: None for k in views.Projected.SOURCES}
    pr = views.Projected(vs, torch.zeros(B, N, views.SPLAT_RECORD), pack, src)
    BN = B * N
    flat = pack.reshape(-1)
    assert pr.field("xys").shape == (B, N, 2) and torch.equal(pr.field("xys").reshape(-1), flat[:2 * BN])
    assert torch.equal(pr.field("depths").reshape(-1), flat[2 * BN:3 * BN])
    assert pr.field("radii").dtype == torch.int32 and pr.field("radii").shape == (B, N)
    assert torch.equal(pr.field("conics").reshape(-1), flat[4 * BN:7 * BN])
    assert torch.equal(pr.field("comp").reshape(-1), flat[7 * BN:8 * BN])
    assert torch.equal(pr.field("opac_eff").reshape(-1), flat[8 * BN:])
    # byte offsets of gol_shade_proj (proj_struct) = 4 x the float offsets above
    assert views.ShadeProj.xys.offset < views.ShadeProj.records.offset
