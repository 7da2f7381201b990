"""GPU: wave64 cross-lane reduction primitives used by the rasterizer backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_wave_sum4_and_dpp_ladder():
    from goliath_amd import _lib

    x = torch.randn(4, 64, device="cuda")
    out = torch.zeros(128, device="cuda")
    _lib.call("gol_selftest_wave_sum4", _lib.fptr(x), _lib.fptr(out), _lib.stream_ptr())
    sums = x.double().sum(1)
    got = out[[15, 31, 47, 63]].double()
    assert torch.allclose(got, sums, atol=1e-4), (got, sums)
    assert abs(float(out[64 + 63]) - float(sums[0])) < 1e-4
