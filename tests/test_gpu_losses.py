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


@pytest.mark.parametrize("mask_c", [0, 1, 3])
def test_l1_fused_into_the_raster_equals_the_standalone_loss(mask_c):
    """render_views(l1_target=...) -- the loss in the raster epilogue, its gradient carried by the raster backward -- is the
    same function as losses.l1_image on the rendered image: value and every input gradient, alone and next to a second
    consumer of the image."""
    from goliath_amd import losses, splat
    from scenes import head_scene

    H, W, N, B = 150, 130, 3000, 2
    views = [head_scene(N, H, W, seed=30 + b, cam_angle=0.3 * b) for b in range(B)]
    g = torch.Generator().manual_seed(mask_c)
    target = torch.rand(B, 3, H, W, generator=g).cuda()
    mask = (torch.rand(B, mask_c, H, W, generator=g) > 0.3).float().cuda() if mask_c else None

    def leaves():
        d = {k: torch.stack([v[k] for v in views]).cuda().requires_grad_(True) for k in
             ("means", "scales", "quats", "opacity", "colors")}
        d["viewmats"] = torch.stack([v["viewmat"] for v in views]).cuda()
        d["intrins"] = torch.tensor([[v["fx"], v["fy"], v["cx"], v["cy"]] for v in views]).cuda()
        return d

    for extra in (False, True):
        a, b_ = leaves(), leaves()
        out = splat.render_views(**a, img_h=H, img_w=W, l1_target=target, l1_mask=mask)
        ref = splat.render_views(**b_, img_h=H, img_w=W)
        l_ref = losses.l1_image(ref["render"], target, mask)
        assert abs(float(out["l1_loss"]) - float(l_ref)) < 1e-6 * max(1.0, abs(float(l_ref)))
        assert torch.equal(out["render"], ref["render"])
        la, lb = 3.0 * out["l1_loss"], 3.0 * l_ref
        if extra:  # the image also feeds another term: both gradients must add up
            la, lb = la + (out["render"] ** 2).mean(), lb + (ref["render"] ** 2).mean()
        la.backward()
        lb.backward()
        for k in ("means", "scales", "quats", "opacity", "colors"):
            assert rel_l2(a[k].grad, b_[k].grad) < 1e-5, (k, extra, rel_l2(a[k].grad, b_[k].grad))
