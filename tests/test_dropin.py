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
