"""CPU: the decoder plumbing (goliath_amd.decoder) matches the reference architecture, and the light
contraction of the decoder tail (goliath_amd.tail) is the identity it claims to be."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")


def _ref_layers():
    sys.path.insert(0, REF)
    try:
        import ca_code.nn.layers as la
    finally:
        sys.path.remove(REF)
    return la


@needs_ref
def test_layers_match_reference_layers():
    from goliath_amd import decoder

    la = _ref_layers()
    torch.manual_seed(0)
    ref = la.ConvTranspose2dWNUB(16, 7, 12, 10, 4, 2, 1)
    la.glorot(ref, 1.0)
    with torch.no_grad():
        ref.bias.normal_()
        ref.weight_g.mul_(1.0 + 0.3 * torch.rand_like(ref.weight_g))
    mine = decoder.ConvTranspose2dWNUB(16, 7, 12, 10)
    mine.load_state_dict(ref.state_dict())  # same keys, same shapes
    x = torch.randn(2, 16, 6, 5)
    assert torch.allclose(mine(x), ref(x), atol=1e-5)
    lin_ref, lin = la.LinearWN(5, 9), decoder.LinearWN(5, 9)
    lin.load_state_dict(lin_ref.state_dict())
    assert torch.allclose(lin(x[:, :5, 0, 0]), lin_ref(x[:, :5, 0, 0]), atol=1e-6)
    # the initialiser: same spread, and the four stride phases of a transposed conv start identical
    a, b = la.ConvTranspose2dWNUB(64, 32, 4, 4, 4, 2, 1), decoder.ConvTranspose2dWNUB(64, 32, 4, 4)
    la.glorot(a, 0.2)
    assert abs(float(a.weight_v.std()) / float(b.weight_v.std()) - 1) < 0.05
    assert torch.equal(b.weight_v[:, :, 1::2, 1::2], b.weight_v[:, :, 0::2, 0::2])
    assert torch.allclose(b.weight_g, b.weight_v.norm().expand_as(b.weight_g))


@needs_ref
def test_decoder_state_dict_matches_reference_prim_decoder():
    from goliath_amd import decoder

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_shade_golden as g

    g.install_stubs()
    sys.path.insert(0, REF)
    try:
        import ca_code.models.rgca as R

        ref = R.PrimDecoder(256, None, torch.zeros(3, 1024, 1024))
    finally:
        sys.path.remove(REF)
    want = {k: tuple(v.shape) for k, v in ref.state_dict().items() if k != "albedo"}
    del ref
    got = {k: tuple(v.shape) for k, v in decoder.PrimDecoderConvs().state_dict().items()}
    assert got == want


def test_decoder_shapes_small():
    from goliath_amd import decoder

    dec = decoder.PrimDecoderConvs(base=1)
    f_vn, f_vc = dec(torch.randn(2, 256), torch.randn(2, 3))
    assert f_vn.shape == (2, 125, 128, 128) and f_vc.shape == (2, 4, 128, 128)
    x_vn, x_vc = dec.trunk(torch.randn(2, 256), torch.randn(2, 3))
    assert x_vn.shape == (2, 16, 64, 64) and x_vc.shape == (2, 16, 64, 64)


@pytest.mark.parametrize("rand", [False, True])
def test_light_contraction_is_exact(rand):
    from goliath_amd import decoder, tail

    torch.manual_seed(1)
    B, h, ncol, nmono = 3, 5, 16, 65
    nd = 3 * ncol + nmono
    layer = decoder.ConvTranspose2dWNUB(16, nd + 12, 2 * h, 2 * h, alpha=1.0).double()
    with torch.no_grad():
        layer.bias.normal_()
    x = torch.randn(B, 16, h, h, dtype=torch.float64)
    L = torch.randn(B, 3, ncol + nmono, dtype=torch.float64)
    Lr = torch.randn(B, 3, ncol + nmono, dtype=torch.float64) if rand else None
    f = layer(x)                                                                  # the reference order of operations
    sh = torch.cat([f[:, :3 * ncol].view(B, 3, ncol, -1), f[:, None, 3 * ncol:nd].expand(-1, 3, -1, -1, -1)
                    .reshape(B, 3, nmono, -1)], 2)                                # rgca.py:506-514
    want = (sh * L[..., None]).sum(2)                                             # rgca.py:528-530 before albedo
    if rand:  # interleaved per colour: (c, c_rand)
        want = torch.stack([want, (sh * Lr[..., None]).sum(2)], 2).reshape(B, 6, -1)
    want = torch.cat([want, f[:, nd:].reshape(B, 12, -1)], 1)
    got, E = tail.contracted_vnocond_torch(x, tail.wn_weight(layer), layer.bias, L, Lr, ncol, nmono)
    assert E == (6 if rand else 3)
    assert torch.allclose(got.reshape(B, E + 12, -1), want, atol=1e-10)
