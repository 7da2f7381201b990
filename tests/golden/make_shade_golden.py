"""Generate tests/golden/shade_golden.npz by running the REFERENCE's own PyTorch code
(/root/reference/ca_code/models/rgca.py PrimDecoder.forward, lines 466-620) on CPU.

Run in the build container only (`python tests/golden/make_shade_golden.py`): /root/reference
does not exist on the GPU box, which is why the vectors are committed.  Third-party imports that
are missing here (cv2, torchvision, pytorch3d, gsplat) are stubbed -- none of them is executed by
the shading tail.  The two conv decoders and the geometry module are replaced by fakes that return
seeded tensors, so lines 505-620 run unmodified on known inputs.  The CUDA-only
`sgutilslib.evaluate_gaussian_fwd/bwd` is served by the reference's OWN sg.cu kernels compiled for the
host (oracle/_ref/libref.so, see oracle/Makefile); the environment-map branch (dir2uv +
mipmap_grid_sample) is reference code as well -- every number in the fixture comes from reference code.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

from oracle import cref  # noqa: E402


def install_stubs():
    """Stand-ins for the third-party imports (tests/golden/ref_stubs.py).  The CUDA-only sgutilslib is served by the
    reference's OWN kernels compiled for the host (oracle/_ref/libref.so) when built, else by the C restatement."""
    import ref_stubs
    from oracle import refso

    impl = refso if refso.available() else cref

    class _SgLib:
        @staticmethod
        def evaluate_gaussian_fwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, integral, w_type):
            integral.copy_(impl.evaluate_gaussian_fwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts,
                                                      n_lights, w_type))
            return []

        @staticmethod
        def evaluate_gaussian_bwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, grad_integral,
                                  grad_dirs, grad_sigmas, grad_light_values, w_type):
            gd, gs, _ = impl.evaluate_gaussian_bwd(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights,
                                                   grad_integral, w_type)
            grad_dirs.copy_(gd)
            grad_sigmas.copy_(gs)
            return []

    ref_stubs.install(sgutilslib=_SgLib)


class FakeGeo:
    def __init__(self, postex, tn_raw):
        self.postex, self.tn_raw, self.calls = postex, tn_raw, 0

    def to_uv(self, x):
        self.calls += 1
        return self.postex if self.calls % 2 == 1 else self.tn_raw

    def vn(self, geom):
        return geom


class Const(torch.nn.Module):
    def __init__(self, value):
        super().__init__()
        self.value = value

    def forward(self, *_):
        return self.value


def main():
    install_stubs()
    import ca_code.utils.sh as ref_sh
    from ca_code.models.rgca import PrimDecoder

    torch.manual_seed(20240917)
    B, S, L = 2, 8, 3
    N = S * S
    f_vnocond = (0.3 * torch.randn(B, 125, S, S)).requires_grad_(True)
    f_vcond = (0.3 * torch.randn(B, 4, S, S)).requires_grad_(True)
    postex = (60.0 * torch.randn(B, 3, S, S)).requires_grad_(True)
    tn_raw = torch.randn(B, 3, S, S).requires_grad_(True)
    albedo = torch.nn.Parameter(0.2 + 0.6 * torch.rand(1, N, 3))
    campos = torch.tensor([[30.0, -40.0, -900.0], [-500.0, 100.0, -700.0]])
    light_intensity = torch.rand(B, L, 3) + 0.2
    light_pos = torch.nn.functional.normalize(torch.randn(B, L, 3), dim=-1) * 1100.0
    n_lights = torch.tensor([3, 2])
    light_dir = torch.nn.functional.normalize(light_pos, dim=-1)
    sh_coeffs = ref_sh.dir2sh_torch(8, light_dir)
    light_sh = (sh_coeffs[:, :, None] * light_intensity[..., None]).sum(dim=1)  # rgca.py:187-191
    mips = [torch.rand(B, 3, 16 >> i, 32 >> i) * 1.6 for i in range(4)]
    rot = torch.linalg.qr(torch.randn(B, 3, 3))[0]

    dec = PrimDecoder.__new__(PrimDecoder)
    torch.nn.Module.__init__(dec)
    dec.n_color_sh_coeffs, dec.n_mono_sh_coeffs, dec.n_diff_coeffs = 16, 65, 113
    dec.diff_sh_degree = 8
    dec.encmod = Const(torch.zeros(B, 256 * 8 * 8))
    dec.viewmod = Const(torch.zeros(B, 8))
    dec.vnocond_mod = Const(f_vnocond)
    dec.vcond_mod = Const(f_vcond)
    dec.albedo = albedo

    out = {}
    w = {}  # fixed random cotangents per output
    gen = torch.Generator().manual_seed(7)

    def run(tag, training, env):
        dec.geo_fn = FakeGeo(postex, tn_raw)
        dec.train(training)
        for t in (f_vnocond, f_vcond, postex, tn_raw, albedo):
            t.grad = None
        torch.manual_seed(99)  # makes the training branch's th.rand reproducible
        preds = dec.forward(torch.zeros(B, 256), torch.zeros(B, 1, 3), campos, light_intensity, light_pos,
                            light_sh, n_lights, mips if env else None, rot if env else None)
        loss = 0.0
        for k, v in preds.items():
            if not v.requires_grad:
                continue
            if k not in w:
                w[k] = torch.randn(v.shape, generator=gen)
            loss = loss + (v * w[k]).sum()
        loss.backward()
        for k, v in preds.items():
            out[f"{tag}/out/{k}"] = v.detach().numpy()
        for n, t in (("f_vnocond", f_vnocond), ("f_vcond", f_vcond), ("postex", postex), ("tn_raw", tn_raw),
                     ("albedo", albedo)):
            out[f"{tag}/grad/{n}"] = t.grad.detach().numpy().copy()
        if training:
            torch.manual_seed(99)
            ld = torch.nn.functional.normalize(torch.rand(B, 1, 3) - 0.5, p=2, dim=-1)
            out[f"{tag}/in/light_dir_rand"] = ld.numpy()
            out[f"{tag}/in/light_sh_rand"] = ref_sh.dir2sh_torch(8, ld).sum(dim=1).numpy()  # intensity = 1

    run("sg_eval", False, False)
    run("sg_train", True, False)
    run("env_eval", False, True)
    for k, v in w.items():
        out[f"w/{k}"] = v.numpy()
    for n, t in (("f_vnocond", f_vnocond), ("f_vcond", f_vcond), ("postex", postex), ("tn_raw", tn_raw),
                 ("albedo", albedo), ("campos", campos), ("light_intensity", light_intensity),
                 ("light_pos", light_pos), ("n_lights", n_lights), ("light_sh", light_sh), ("lightrot", rot),
                 *[(f"mip{i}", m) for i, m in enumerate(mips)]):
        out[f"in/{n}"] = t.detach().numpy()
    path = os.path.join(HERE, "shade_golden.npz")
    np.savez_compressed(path, **{k: np.asarray(v, dtype=np.float32 if v.dtype.kind == "f" else v.dtype)
                                for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
