"""CPU: the drop-in shims resolve under the module names the reference imports."""
import os
import sys

import pytest

REF = "/root/reference"


def test_install_registers_reference_module_names():
    from goliath_amd import dropin, mvp, sg, splat

    names = dropin.install()
    assert names == ["gsplat", "sgutilslib", "mvpraymarchlib", "utilslib"]
    import gsplat
    import mvpraymarchlib
    import sgutilslib
    import utilslib

    assert gsplat.project_gaussians is splat.project_gaussians
    assert gsplat.rasterize_gaussians is splat.rasterize_gaussians
    assert sgutilslib.evaluate_gaussian_fwd is sg.sgutilslib.evaluate_gaussian_fwd
    assert mvpraymarchlib.raymarch_backward is mvp.mvpraymarchlib.raymarch_backward
    assert utilslib.compute_raydirs_forward is mvp.utilslib.compute_raydirs_forward


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_python_wrappers_import_on_top_of_the_shims():
    from goliath_amd import dropin, mvp, sg

    dropin.install()
    sys.path.insert(0, REF)
    try:
        for m in [k for k in sys.modules if k.startswith("extensions")]:
            del sys.modules[m]
        import extensions.mvpraymarch.mvpraymarch as ref_mvp
        import extensions.sgutils.sgutils as ref_sg
        import extensions.utils.utils as ref_utils

        assert ref_sg.sgutilslib.evaluate_gaussian_fwd is sg.sgutilslib.evaluate_gaussian_fwd
        assert ref_mvp.mvpraymarchlib.compute_aabb is mvp.mvpraymarchlib.compute_aabb
        assert ref_utils.utilslib.compute_raydirs_forward is mvp.utilslib.compute_raydirs_forward
        # same public signatures as the reference wrappers
        import inspect

        ours = inspect.signature(mvp.mvpraymarch).parameters
        theirs = inspect.signature(ref_mvp.mvpraymarch).parameters
        assert list(ours) == list(theirs)
        assert list(inspect.signature(sg.evaluate_gaussian).parameters) == list(
            inspect.signature(ref_sg.evaluate_gaussian).parameters)
    finally:
        sys.path.remove(REF)


def test_patch_losses_and_urhand_on_stand_in_modules():
    """patch_losses / patch_urhand rebind by name; exercised on stand-ins (the reference's loss package needs
    `addict` / `omegaconf`, which this image does not have)."""
    import types

    import torch

    from goliath_amd import dropin, losses, shadowmap

    class FnLoss(torch.nn.Module):  # same contract as ca_code/loss/registry.py:40-56
        def __init__(self, fn, function_args):
            super().__init__()
            self.fn, self.extra_args = fn, function_args

        def forward(self, preds, targets):
            return self.fn(preds, targets, **self.extra_args)

    reg = types.SimpleNamespace(loss_registry={"rgb_l1": "reference", "kl": "untouched"}, FnLoss=FnLoss)
    assert dropin.patch_losses(reg) is reg
    mod = reg.loss_registry["rgb_ssim"](None, src_key="rgb", tgt_key="image", mask_key="image_weight")
    assert isinstance(mod, FnLoss) and mod.fn is losses.rgb_ssim and mod.extra_args["src_key"] == "rgb"
    assert reg.loss_registry["rgb_l1"](None).fn is losses.rgb_l1 and reg.loss_registry["kl"] == "untouched"
    from goliath_amd import meshraster, urhand

    class ConvTeacherDecoder:  # stand-in for ca_code.models.urhand.ConvTeacherDecoder
        def forward(self):
            return "reference"

    ur = types.SimpleNamespace(get_shadow_map="reference", RenderLayer="drtk", ConvTeacherDecoder=ConvTeacherDecoder)
    assert dropin.patch_urhand(ur).get_shadow_map is shadowmap.get_shadow_map
    # the module-level RenderLayer also builds the model's final DIFFERENTIABLE render (urhand.py:684): left alone
    assert ur.RenderLayer == "drtk" and ConvTeacherDecoder.forward is urhand.conv_teacher_decoder_forward
    assert dropin.patch_urhand(ur, mesh_render_layer=True).RenderLayer is meshraster.RenderLayer

    class OLATRGBDecoder:  # stand-in for ca_code.models.hand_teacher_mvp.OLATRGBDecoder
        def forward_rgb(self):
            return "reference"

    tm = types.SimpleNamespace(OLATRGBDecoder=OLATRGBDecoder)
    assert dropin.patch_hand_teacher(tm) is tm and OLATRGBDecoder.forward_rgb is urhand.olat_rgb_decoder_forward_rgb
