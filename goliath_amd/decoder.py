"""The two transposed-conv decoders that feed the shading tail -- PyTorch plumbing for the end-to-end
measurement (SURVEY 8d mode B) and the host of the fused decoder tail.

Same architecture, parameter names and shapes as `PrimDecoder.__init__`
(/root/reference/ca_code/models/rgca.py:392-456): `viewmod`, `encmod`, `vnocond_mod`, `vcond_mod` built from
weight-normalised layers with an untied (per-pixel) bias (`ca_code/nn/layers.py:331-397,470-480`;
weight norm: one magnitude per output channel, direction normalised over the whole tensor,
`layers.py:157-244`), so a reference state_dict loads unchanged (tests/test_decoder.py checks the key
set against the reference class when the reference tree is present).

`base` is the spatial size of the latent grid: 8 in the reference (slab 1024, N = 1,048,576 Gaussians);
smaller values keep the architecture and shrink the slab for tests.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

LEAK = 0.2
CHANNELS = (256, 128, 128, 64, 32, 16)  # outputs of the six hidden layers (rgca.py:408-426)


def _wn(v, g):
    return v * (g / v.norm())


def _glorot_(v, fan_in, fan_out, ksize, alpha):
    gain = math.sqrt(2.0 / (1.0 + alpha * alpha))
    std = gain * math.sqrt(2.0 / ((fan_in + fan_out) * ksize))
    with torch.no_grad():
        v.uniform_(-std * math.sqrt(3.0), std * math.sqrt(3.0))


class LinearWN(nn.Module):
    def __init__(self, n_in, n_out, alpha=LEAK):
        super().__init__()
        self.weight_v = nn.Parameter(torch.empty(n_out, n_in))
        _glorot_(self.weight_v, n_in, n_out, 1, alpha)
        self.weight_g = nn.Parameter(torch.full((n_out, 1), float(self.weight_v.detach().norm())))
        self.bias = nn.Parameter(torch.zeros(n_out))

    def forward(self, x):
        return F.linear(x, _wn(self.weight_v, self.weight_g), self.bias)


class ConvTranspose2dWNUB(nn.Module):
    """4x4 / stride 2 / pad 1 transposed conv, weight-normalised, untied bias [C_out, H, W]."""

    def __init__(self, n_in, n_out, height, width, alpha=LEAK):
        super().__init__()
        self.weight_v = nn.Parameter(torch.empty(n_in, n_out, 4, 4))
        _glorot_(self.weight_v, n_in, n_out, 4, alpha)
        with torch.no_grad():  # the four stride phases start identical (layers.py:642-647)
            w = self.weight_v
            w[:, :, 0::2, 1::2] = w[:, :, 0::2, 0::2]
            w[:, :, 1::2, 0::2] = w[:, :, 0::2, 0::2]
            w[:, :, 1::2, 1::2] = w[:, :, 0::2, 0::2]
        self.weight_g = nn.Parameter(torch.full((1, n_out, 1, 1), float(self.weight_v.detach().norm())))
        self.bias = nn.Parameter(torch.zeros(n_out, height, width))

    def weight(self):
        return _wn(self.weight_v, self.weight_g)

    def forward(self, x):
        return F.conv_transpose2d(x, self.weight(), None, 2, 1) + self.bias[None]


def _stack(n_in, n_out, base):
    layers, c, s = [], n_in, base
    for co in CHANNELS:
        s *= 2
        layers += [ConvTranspose2dWNUB(c, co, s, s), nn.LeakyReLU(LEAK, inplace=True)]
        c = co
    layers.append(ConvTranspose2dWNUB(c, n_out, 2 * s, 2 * s, alpha=1.0))
    return nn.Sequential(*layers)


class PrimDecoderConvs(nn.Module):
    """embs[B,n_embs], headrel_campos[B,3] -> f_vnocond[B,125,S,S], f_vcond[B,4,S,S]  (rgca.py:494-503)."""

    def __init__(self, n_embs=256, n_diff_sh=8, n_color_sh=3, base=8):
        super().__init__()
        self.base, self.slabsize = base, base * 128
        ncol = (n_color_sh + 1) ** 2
        self.n_vnocond = 3 * ncol + ((n_diff_sh + 1) ** 2 - ncol) + 12
        self.viewmod = nn.Sequential(LinearWN(3, 8), nn.LeakyReLU(LEAK, inplace=True))
        self.encmod = nn.Sequential(LinearWN(n_embs, 256 * base * base), nn.LeakyReLU(LEAK, inplace=True))
        self.vnocond_mod = _stack(256, self.n_vnocond, base)
        self.vcond_mod = _stack(256 + 8, 4, base)

    def trunk(self, embs, headrel_campos):
        """Everything up to the inputs of the two last layers: x_vnocond, x_vcond [B,16,S/2,S/2]."""
        b = self.base
        z = self.encmod(embs).view(-1, 256, b, b)
        view = self.viewmod(F.normalize(headrel_campos, dim=1))[:, :, None, None].expand(-1, -1, b, b)
        return self.vnocond_mod[:-1](z), self.vcond_mod[:-1](torch.cat([z, view], dim=1))

    def forward(self, embs, headrel_campos):
        x_vn, x_vc = self.trunk(embs, headrel_campos)
        return self.vnocond_mod[-1](x_vn), self.vcond_mod[-1](x_vc)
