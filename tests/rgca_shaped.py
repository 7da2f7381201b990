"""TEST INFRASTRUCTURE: a small, seeded stand-in with the attribute layout of the reference's RGCA `AutoEncoder`
(/root/reference/ca_code/models/rgca.py:50-110) and `PrimDecoder` (:380-464), so that

  * tests/golden/make_rgca_model_golden.py can run the REFERENCE's own `AutoEncoder.forward` / `AutoEncoder.render` /
    `PrimDecoder.forward` (rgca.py:112-253, 466-620) and `EnvSpinDecorator.forward` (ca_code/utils/light_decorator.py:102-164)
    as unbound methods on it on the CPU (the build container has /root/reference), and
  * tests/test_gpu_rgca_model_golden.py can run goliath_amd.rgca's drop-ins on an identical object where /root/reference
    does not exist, and compare every returned key and the parameter gradients with the committed fixture.

Nothing here is under test: the encoder / geometry decoder are closures over tensors the fixture stores, the geometry
module is a plain-torch grid surface, the decoder stacks are goliath_amd.decoder's layer classes (state-dict compatible with
the reference's, tests/test_dropin_real_classes.py) in a shortened 8 -> S ladder, filled by a seeded, name-ordered recipe
both sides regenerate bit-identically on the CPU.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

CAMERAS = ["400002", "400004", "400013", "410011"]     # sorted (ParamHolder sorts its keys); "41..." = grey-scale cameras
IDENTITY = "400004"
S = 64                    # slab: 4096 Gaussians
H, W = 256, 208           # >= 200 in both directions: compose_envmap's mirror ball (envmap.py:326-334)
GRID = 8                  # the coarse mesh: (GRID + 1)^2 vertices
HIDDEN = (32, 16)         # 8 -> 16 -> 32 -> 64: two hidden transposed convs + the output layer


class GridGeo:
    """GeometryModule look-alike (ca_code/utils/geom.py:50-130: to_uv, vn) for a (GRID+1)^2-vertex grid whose uv island is
    the whole S x S map: to_uv = bilinear interpolation of the vertex grid, vn = normalised cross product of the grid's
    central differences.  Plain differentiable torch on both sides of the comparison."""

    def __init__(self, S=S, n=GRID):
        self.S, self.n = S, n

    def to_uv(self, values):
        B = values.shape[0]
        g = values.reshape(B, self.n + 1, self.n + 1, 3).permute(0, 3, 1, 2)
        return F.interpolate(g, size=(self.S, self.S), mode="bilinear", align_corners=True)

    def vn(self, verts):
        B = verts.shape[0]
        g = verts.reshape(B, self.n + 1, self.n + 1, 3)
        p = torch.cat([2 * g[:, :1] - g[:, 1:2], g, 2 * g[:, -1:] - g[:, -2:-1]], 1)
        p = torch.cat([2 * p[:, :, :1] - p[:, :, 1:2], p, 2 * p[:, :, -1:] - p[:, :, -2:-1]], 2)
        du, dv = p[:, 2:, 1:-1] - p[:, :-2, 1:-1], p[:, 1:-1, 2:] - p[:, 1:-1, :-2]
        return F.normalize(torch.cross(du, dv, dim=-1), dim=-1).reshape(B, -1, 3)


def base_mesh():
    """[(GRID+1)^2, 3] a dome of ~160 x 200 mm bulging towards the cameras (-z), roughly a face-sized surface."""
    t = torch.linspace(-1.0, 1.0, GRID + 1)
    v, u = torch.meshgrid(t, t, indexing="ij")
    return torch.stack([80.0 * u, 100.0 * v, -60.0 * (1.0 - 0.5 * (u * u + v * v))], -1).reshape(-1, 3)


def fill_(module, seed):
    """Seeded, name-ordered parameter recipe (CPU generator: identical on every machine with this torch build).  Weights
    ~ their initial scale; biases small -- with a push on the Gaussian-parameter channels so that the Gaussians have
    sensible sizes and opacities."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            if name.endswith("weight_g") or name == "albedo":
                continue
            if name.endswith("weight_v"):
                a, b = p.shape[0], p.shape[1]            # a fixed, shape-derived scale (NOT the module's own random init)
                p.copy_(torch.randn(p.shape, generator=g) * math.sqrt(3.8 / ((a + b) * (p.numel() // (a * b)))))
            elif name.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
        for name, p in sorted(module.named_parameters()):
            if name.endswith("weight_g"):
                v = dict(module.named_parameters())[name[:-1] + "v"]
                p.fill_(float(v.norm()) * 2.5)      # gain 2.5 per layer: decoder outputs of ~0.5 standard deviation over the slab


class ShapedPrimDecoder(nn.Module):
    """Attribute layout of PrimDecoder (rgca.py:392-464) with an 8 -> S ladder of 3 transposed convs per stack."""

    def __init__(self, seed=0, nudges=None, raw=False):
        """raw: the UN-manicured scene of the fourth fixture case (tests/golden/rgca_model_raw_golden.npz): no depth nudges, no
        roughness conditioning -- whatever discontinuities the seeded scene happens to contain stay in."""
        super().__init__()
        from goliath_amd import decoder as D

        self.slabsize, self.n_splats, self.n_embs = S, S * S, 256
        self.geo_fn = GridGeo()
        self.diff_sh_degree, self.color_sh_degree = 8, 3
        self.n_color_sh_coeffs, self.n_mono_sh_coeffs = 16, 65
        self.n_diff_coeffs = 3 * 16 + 65
        self.viewmod = nn.Sequential(D.LinearWN(3, 8), nn.LeakyReLU(0.2, inplace=True))
        self.encmod = nn.Sequential(D.LinearWN(256, 256 * 8 * 8), nn.LeakyReLU(0.2, inplace=True))

        def stack(n_in, n_out):
            layers, c, s = [], n_in, 8
            for co in HIDDEN:
                s *= 2
                layers += [D.ConvTranspose2dWNUB(c, co, s, s), nn.LeakyReLU(0.2, inplace=True)]
                c = co
            layers.append(D.ConvTranspose2dWNUB(c, n_out, 2 * s, 2 * s, alpha=1.0))
            return nn.Sequential(*layers)

        self.vnocond_mod = stack(256, self.n_diff_coeffs + 12)
        self.vcond_mod = stack(256 + 8, 4)
        g = torch.Generator().manual_seed(seed + 17)
        self.albedo = nn.Parameter(0.2 + 0.6 * torch.rand(1, S * S, 3, generator=g))
        fill_(self, seed)
        with torch.no_grad():
            b = self.vnocond_mod[-1].bias
            nd = self.n_diff_coeffs
            b[0], b[16], b[32] = 1.2, 1.0, 0.8               # DC colour SH: a lit surface
            b[nd + 7:nd + 10] += 2.3                          # softplus^-1 of a ~2.4 mm scale
            b[nd + 10] += 1.0                                 # opacity logit
            # roughness: sigma = 0.1 exp(x) in about 0.08 .. 0.25 -- SG lobes that fp32 resolves.  (At sigma = 0.01, the floor
            # of rgca.py:527, two fp32 evaluations of exp(-angle^2 / 2 sigma^2) differ by 1e-3: d/d angle = angle / sigma^2 times
            # the 1e-6 rad rounding of acos near 1; the reference's own sg.cu is that far from its fp64 evaluation there.)
            if not raw:
                b[nd + 11] += 0.3
                self.vnocond_mod[-1].weight_v[:, nd + 11] *= 0.3
            # depth separation (computed once by the fixture's generator, stored in the fixture): per-Gaussian z offsets through
            # the untied bias, so that no two Gaussians sharing a tile are within 32 ulps in depth in any view of the fixture.
            # Otherwise their compositing ORDER -- and with it rgb and every gradient -- hinges on the last bit of
            # Rt @ head_pose, which differs between BLAS builds and between the CPU and the GPU.
            if nudges is not None:
                idx, dz = nudges
                b[nd + 2].view(-1)[torch.as_tensor(idx, dtype=torch.long)] += torch.as_tensor(dz, dtype=b.dtype)


class ShapedCal(nn.Module):
    """CalV5's attribute layout (color_cal.py:112-150): `holder(idxs)`, name_to_idx, identity / grey indices, lr scales."""

    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed + 5)
        p = torch.tensor([1.0, 1.0, 1.0, 0.0, 0.0, 0.0]).repeat(len(CAMERAS), 1)
        p[:, :3] += 0.2 * torch.randn(len(CAMERAS), 3, generator=g)
        p[:, 3:] += 0.1 * torch.randn(len(CAMERAS), 3, generator=g)
        self.params = nn.Parameter(p)
        self.identity_idx = CAMERAS.index(IDENTITY)
        self.grey_idxs = [i for i, c in enumerate(CAMERAS) if c.startswith("41")]
        self.gs_lrscale, self.col_lrscale = 1.0, 0.1

    def holder(self, idxs):
        return self.params[idxs]

    def name_to_idx(self, names):
        return torch.tensor([CAMERAS.index(n) for n in names], device=self.params.device, dtype=torch.long)


class ShapedBlur(nn.Module):
    """LearnableBlur's attribute layout (dof_cal.py:21-42)."""

    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed + 9)
        self.weights_raw = nn.Parameter(torch.ones(len(CAMERAS), 3) + torch.randn(len(CAMERAS), 3, generator=g))

    def reg(self, cameras):
        return self.weights_raw[torch.tensor([CAMERAS.index(c) for c in cameras], device=self.weights_raw.device)]


class ShapedAutoEncoder(nn.Module):
    """Attribute layout of AutoEncoder (rgca.py:50-110).  `encoder` / `geomdecoder` return the leaves `embs` / `geom`
    (they are inputs of the path under test; gradients w.r.t. them are compared)."""

    def __init__(self, embs, geom, seed=0, cal=True, blur=True, nudges=None, raw=False):
        super().__init__()
        self.height, self.width, self.n_diff_sh, self.bg_weight = H, W, 8, 1.0
        self.decoder = ShapedPrimDecoder(seed, nudges, raw)
        self.geo_fn = self.decoder.geo_fn
        self._embs, self._geom = embs, geom
        self.cal_enabled, self.learn_blur_enabled = bool(cal), bool(blur)
        if cal:
            self.cal = ShapedCal(seed)
        if blur:
            self.learn_blur = ShapedBlur(seed)

    def encoder(self, registration_vertices, color):
        return {"embs": self._embs, "embs_mu": self._embs * 1.0, "embs_logvar": self._embs * 0.01}

    def geomdecoder(self, embs):
        return {"face_geom": self._geom}


def cameras(B, radius=650.0):
    """K[B,3,3], Rt[B,3,4], campos[B,3]: cameras on a ring in front of the dome (which bulges towards -z)."""
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = 620.0
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = W / 2.0 + 3.0, H / 2.0 - 5.0, 1.0
    Rt, campos = [], []
    for b in range(B):
        ang = 0.5 * b - 0.25
        eye = torch.tensor([radius * math.sin(ang), 40.0 * b - 20.0, -radius * math.cos(ang)])
        fwd = -eye / eye.norm()
        right = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), fwd)
        right = right / right.norm()
        R = torch.stack([right, torch.linalg.cross(fwd, right), fwd])
        Rt.append(torch.cat([R, (-R @ eye)[:, None]], 1))
        campos.append(eye)
    return K, torch.stack(Rt), torch.stack(campos)


def head_pose(B):
    """[B,3,4]: a small rotation about y and z + a translation (the head-relative glue of rgca.py:175-195 must matter)."""
    out = []
    for b in range(B):
        a, c = 0.15 + 0.1 * b, -0.08 * (b + 1)
        Ry = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        Rz = torch.tensor([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
        out.append(torch.cat([Ry @ Rz, torch.tensor([[6.0 - 3 * b], [-4.0], [12.0 + 2 * b]])], 1))
    return torch.stack(out)


def world_from_head(hp, x):
    """x[B,...,3] head-relative -> world: the inverse of rgca.py:178-183 ((p - t) @ R)."""
    return x @ hp[:, :3, :3].transpose(1, 2) + hp[:, None, :3, 3]


def batch_inputs(B, seed, n_lights_max=3, stored=None):
    """The per-frame batch entries AutoEncoder.forward takes (rgca.py:153-171), seeded.  Cameras and lights are placed in
    the head frame and mapped to the world through head_pose, so the model's world -> head transform is exercised.
    stored: {"head_pose", "Rt", "campos", "light_pos"} from the fixture -- these come out of matrix products / an inverse
    whose last bit depends on the host's BLAS / LAPACK build, so the fixture carries the generator's values."""
    g = torch.Generator().manual_seed(seed)
    hp = head_pose(B)
    K, Rt_head, campos_head = cameras(B)
    bottom = torch.tensor([[[0.0, 0.0, 0.0, 1.0]]]).expand(B, -1, -1)
    hp4 = torch.cat([hp, bottom], 1)
    Rt = (torch.cat([Rt_head, bottom], 1) @ torch.linalg.inv(hp4))[:, :3]       # so that Rt @ head_pose = the ring camera
    campos = world_from_head(hp, campos_head[:, None])[:, 0]
    lp_head = 1100.0 * F.normalize(torch.randn(B, n_lights_max, 3, generator=g) + torch.tensor([0.0, 0.0, -1.5]), dim=-1)
    light_pos = world_from_head(hp, lp_head)
    if stored is not None:
        hp, Rt, campos, light_pos = (torch.as_tensor(stored[k]) for k in ("head_pose", "Rt", "campos", "light_pos"))
    return dict(head_pose=hp, campos=campos, registration_vertices=torch.zeros(B, (GRID + 1) ** 2, 3),
                color=torch.zeros(B, 3, 8, 8), light_intensity=0.4 + torch.rand(B, n_lights_max, 1, generator=g),
                light_pos=light_pos, n_lights=torch.tensor([n_lights_max, max(1, n_lights_max - 1)][:B]),
                K=K, Rt=Rt, background=torch.rand(B, 3, H, W, generator=g),
                is_fully_lit_frame=torch.tensor([True, False][:B]), camera_id=["400013", "410011"][:B],
                frame_id=torch.arange(B), iteration=0)


def leaves(B, seed, stored=None):
    """embs[B,256], geom[B,(GRID+1)^2,3] (requires_grad): what the stand-in encoder / geometry decoder return."""
    if stored is not None:
        return (torch.as_tensor(stored["embs"]).clone().requires_grad_(True),
                torch.as_tensor(stored["geom"]).clone().requires_grad_(True))
    g = torch.Generator().manual_seed(seed + 3)
    embs = torch.randn(B, 256, generator=g)
    geom = base_mesh()[None] + 2.0 * torch.randn(B, (GRID + 1) ** 2, 3, generator=g)
    return embs.requires_grad_(True), geom.requires_grad_(True)


GRAD_PARAMS = ("decoder.albedo", "decoder.vnocond_mod.4.weight_g", "decoder.vnocond_mod.4.weight_v", "decoder.vcond_mod.4.bias",
               "decoder.vnocond_mod.2.bias", "decoder.vnocond_mod.2.weight_v", "decoder.vcond_mod.2.weight_v",
               "decoder.encmod.0.weight_g", "decoder.viewmod.0.weight_v", "cal.params", "learn_blur.weights_raw")
OUTPUT_KEYS = ("geom", "headrel_light_sh", "embs", "embs_mu", "embs_logvar", "color", "opacity", "primpos", "primqvec",
               "primscale", "primscale_preclip", "sigma", "spec_vis", "spec_nml", "spec_dnml", "diff_color", "spec_color",
               "primnmlbase", "rgb", "alpha", "depth")


def loss_weights(seed):
    """Fixed random cotangents for the scalar the fixture back-propagates: sum_k <w_k, preds[k]> over rgb, depth and a few
    per-Gaussian keys the training losses read (rgca.py losses: rgb, primscale_preclip, color_rand / cos_weight ...)."""
    g = torch.Generator().manual_seed(seed + 11)
    return {"rgb": lambda t: torch.randn(t.shape, generator=g), "depth": lambda t: 1e-3 * torch.randn(t.shape, generator=g),
            "primscale_preclip": lambda t: 1e-2 * torch.randn(t.shape, generator=g),
            "spec_nml": lambda t: 1e-2 * torch.randn(t.shape, generator=g),
            "color_rand": lambda t: 1e-2 * torch.randn(t.shape, generator=g)}
