"""GPU parity: fused masked L1 image loss vs the reference expression (ca_code/loss/__init__.py:411)."""
import pytest
import torch

from scenes import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,mask_c", [((2, 3, 64, 52), 1), ((1, 3, 33, 17), 3), ((2, 3, 40, 40), 0)])
def test_l1_image_matches_torch(shape, mask_c):
    from goliath_amd import losses

    g = torch.Generator().manual_seed(0)
    pred = torch.rand(shape, generator=g).cuda().requires_grad_(True)
    tgt = torch.rand(shape, generator=g).cuda()
    mask = None if mask_c == 0 else (torch.rand(shape[0], mask_c, *shape[2:], generator=g) > 0.3).float().cuda()
    ref_in = pred.detach().clone().requires_grad_(True)
    ref = ((ref_in - tgt) * (mask if mask is not None else 1.0)).abs().mean()
    out = losses.l1_image(pred, tgt, mask)
    assert abs(float(out) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    (2.5 * ref).backward()
    (2.5 * out).backward()
    assert rel_l2(pred.grad, ref_in.grad) < 1e-6
    d = {"rendered_rgb": pred.detach()}
    t = {"image": tgt}
    if mask is not None:
        t["image_mask"] = mask
    assert abs(float(losses.rgb_l1(d, t)) - float(ref)) < 1e-6
