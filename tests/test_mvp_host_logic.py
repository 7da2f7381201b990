"""CPU: argument validation of the MVP host mirror (no kernels): which algorithm / warp-field combinations the operator
accepts (mvpraymarch_kernel.cu:92-104: algo 1 = one warp field per box, algo 0 = none)."""
import pytest
import torch


def test_algo_and_warp_must_agree():
    from goliath_amd import mvp

    mvp._check_algo(0, None)
    mvp._check_algo(1, torch.zeros(1, 2, 2, 2, 2, 3))
    for algo, warp in ((0, torch.zeros(1, 2, 2, 2, 2, 3)), (1, None), (2, None)):
        with pytest.raises(NotImplementedError):
            mvp._check_algo(algo, warp)


def test_warp_field_shape_is_checked_against_the_template():
    from goliath_amd import mvp

    tpl = torch.zeros(2, 5, 4, 4, 4, 4)
    assert mvp._warp_dims(torch.zeros(2, 5, 3, 2, 4, 3), tpl) == (3, 2, 4)
    for bad in (torch.zeros(2, 5, 3, 3, 3, 4), torch.zeros(2, 4, 3, 3, 3, 3), torch.zeros(2, 5, 3, 3, 3)):
        with pytest.raises(RuntimeError):
            mvp._warp_dims(bad, tpl)


def test_cpu_tensors_are_refused_before_any_call_into_the_library():
    from goliath_amd import _lib, mvp

    z = torch.zeros
    with pytest.raises((RuntimeError, _lib.GoliathHipError)):
        mvp.mvpraymarch(z(1, 4, 4, 3), z(1, 4, 4, 3), 0.1, z(1, 4, 4, 2), (z(1, 2, 3), z(1, 2, 3, 3), z(1, 2, 3)),
                        z(1, 2, 2, 2, 2, 4), None)
