"""TEST INFRASTRUCTURE: small, seeded stand-ins with the attribute layout of the reference's `ConvTeacherDecoder`
(/root/reference/ca_code/models/urhand.py:244-346) and `OLATRGBDecoder` (ca_code/models/hand_teacher_mvp.py:159-251), so that

  * tests/golden/make_urhand_model_golden.py can run the REFERENCE's own `forward` / `forward_rgb` code on them on the CPU
    (unbound method call on these objects; the build container has /root/reference), and
  * the GPU tests can run goliath_amd.urhand's drop-ins on identical objects where /root/reference does not exist.

The module also restates the handful of geometry helpers the reference forwards call (`vert_normals`, `index`,
`compute_tbn_uv_given_normal`, `xyz2normals`, `tile2d`, `build_cam_rot_mat`; ca_code/utils/geom.py:337-346, 432-470,
665-686, ca_code/utils/torchutils.py:234-248, ca_code/nn/blocks.py:731-743, urhand.py:62-80): the drop-ins look them up in
the module that defines the decoder class, which for these stand-ins is this one.  tests/test_dropin_real_classes.py
checks every restatement against the reference's function where the reference exists.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------- helper restatements
def index(x, idxs, dim):
    shape = [*x.shape]
    del shape[dim]
    shape[dim:dim] = [*idxs.shape]
    return x.index_select(dim, idxs.reshape(-1)).reshape(shape)


def vert_normals(v, vi, eps=1.0e-5):
    p0, p1, p2 = (v[:, vi[:, k]] for k in range(3))
    fn = torch.cross(p1 - p0, p2 - p0, dim=-1)
    fn = fn / torch.norm(fn, dim=-1, keepdim=True).clamp(min=eps)     # geom.py face_normals: unit face normals
    fn = fn[:, :, None].expand(-1, -1, 3, -1).reshape(fn.shape[0], -1, 3)
    flat = vi.reshape(1, -1).expand(v.shape[0], -1)
    vn = torch.zeros_like(v)
    for j in range(3):
        vn[..., j].scatter_add_(1, flat, fn[..., j])
    return vn / torch.norm(vn, dim=-1, keepdim=True).clamp(min=eps)


def compute_tbn_uv_given_normal(tri_xyz, tri_uv, normals, eps=1e-5):
    tri_uv = tri_uv[None]
    v01, v02 = tri_xyz[:, :, 1] - tri_xyz[:, :, 0], tri_xyz[:, :, 2] - tri_xyz[:, :, 0]
    t01, t02 = tri_uv[:, :, 1] - tri_uv[:, :, 0], tri_uv[:, :, 2] - tri_uv[:, :, 0]
    fin = t01[..., 0] * t02[..., 1] - t01[..., 1] * t02[..., 0]
    fin[torch.abs(fin) < 1.0e-8] = 1.0e-8
    tan = (1.0 / fin)[..., None] * (v01 * t02[..., 1][..., None] - v02 * t01[..., 1][..., None])
    tan = tan / torch.norm(tan, dim=-1, keepdim=True).clamp(min=eps)
    bit = torch.cross(normals, tan, dim=-1)
    bit = bit / torch.norm(bit, dim=-1, keepdim=True).clamp(min=eps)
    tan = torch.cross(bit, normals, dim=-1)
    tan = tan / torch.norm(tan, dim=-1, keepdim=True).clamp(min=eps)
    return tan, bit, normals


def xyz2normals(xyz, eps=1e-8):
    n = torch.zeros_like(xyz)
    p = F.pad(xyz, (1, 1, 1, 1))
    U = (p[:, :, 2:, 1:-1] - p[:, :, :-2, 1:-1]) / -2
    V = (p[:, :, 1:-1, 2:] - p[:, :, 1:-1, :-2]) / -2
    n[:, 0] = U[:, 1] * V[:, 2] - U[:, 2] * V[:, 1]
    n[:, 1] = U[:, 2] * V[:, 0] - U[:, 0] * V[:, 2]
    n[:, 2] = U[:, 0] * V[:, 1] - U[:, 1] * V[:, 0]
    return n / torch.norm(n, dim=1, keepdim=True).clamp(min=eps)


def tile2d(x, size):
    return x[:, :, None, None].expand(-1, -1, size, size)


def build_cam_rot_mat(campos, objcenter=None):
    campos[(campos[:, 0].abs() + campos[:, 2].abs()) < 1e-8, 2] += 1e-2
    z = F.normalize(-campos if objcenter is None else objcenter - campos, dim=1)
    up = torch.zeros_like(campos)
    up[:, 1] = 1
    x = F.normalize(torch.cross(z, up, dim=1), dim=1)
    y = F.normalize(torch.cross(z, x, dim=1), dim=1)
    return torch.stack([x, y, z], dim=1)


def drtk_transform(v, campos, camrot, focal, princpt):
    """drtk.transform(v, campos=, camrot=, focal=[N,2,2], princpt=[N,2]) as hand_teacher_mvp.py:318 uses it."""
    cam = (v - campos[:, None]) @ camrot.transpose(1, 2)
    pix = (cam[..., :2] / cam[..., 2:3]) @ focal.transpose(1, 2) + princpt[:, None]
    return torch.cat([pix, cam[..., 2:3]], -1)


# ---------------------------------------------------------------------------------------------- geometry stand-in
class FakeGeo(nn.Module):
    """A GeometryModule look-alike for a (n+1)^2-vertex grid mesh whose uv island covers [0.1, 0.9]^2 of a S x S map."""

    def __init__(self, S=64, n=8):
        super().__init__()
        self.uv_size = S
        ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
        vt = np.stack([0.1 + 0.8 * jj / n, 0.1 + 0.8 * ii / n], -1).reshape(-1, 2).astype(np.float32)
        vid = lambda i, j: i * (n + 1) + j
        faces = []
        for i in range(n):
            for j in range(n):
                faces += [[vid(i, j), vid(i, j + 1), vid(i + 1, j)], [vid(i + 1, j + 1), vid(i + 1, j), vid(i, j + 1)]]
        vi = np.asarray(faces, np.int64)
        index_image = -np.ones((S, S, 3), np.int64)
        face_image = np.zeros((S, S), np.int64)
        bary = np.zeros((S, S, 3), np.float32)
        for r in range(S):
            for c in range(S):
                u, v = ((c + 0.5) / S - 0.1) / 0.8 * n, ((r + 0.5) / S - 0.1) / 0.8 * n
                if not (0 <= u < n and 0 <= v < n):
                    continue
                j, i = int(u), int(v)
                fu, fv = u - j, v - i
                if fu + fv <= 1.0:
                    f, w = 2 * (i * n + j), (1 - fu - fv, fu, fv)
                else:
                    f, w = 2 * (i * n + j) + 1, (fu + fv - 1, 1 - fu, 1 - fv)
                index_image[r, c], face_image[r, c], bary[r, c] = vi[f], f, w
        self.register_buffer("vt", torch.from_numpy(vt))
        self.register_buffer("vi", torch.from_numpy(vi))
        self.register_buffer("vti", torch.from_numpy(vi))
        self.register_buffer("v2uv", torch.arange(vt.shape[0])[:, None])
        self.register_buffer("index_image", torch.from_numpy(index_image))
        self.register_buffer("face_index_image", torch.from_numpy(face_image))
        self.register_buffer("bary_image", torch.from_numpy(bary))
        self.n = n

    def to_uv(self, values):
        mask = (self.index_image != -1).any(dim=-1)
        idx = self.index_image.clamp(min=0)
        out = (values[:, idx] * self.bary_image[None, ..., None]).sum(3)          # [B,S,S,C]
        return (out * mask[None, ..., None]).permute(0, 3, 1, 2).contiguous()

    def from_uv(self, uv):
        g = (self.vt * 2.0 - 1.0)[None, :, None].expand(uv.shape[0], -1, -1, -1)
        return F.grid_sample(uv, g, mode="bilinear", align_corners=False)[..., 0].permute(0, 2, 1)


class OracleRenderLayer:
    """CPU stand-in for render_drtk.RenderLayer as the shadow-map path uses it: depth image of the mesh through
    oracle/mesh_ref.py (the conventions of goliath_amd/csrc/meshraster.hip).  Golden generation only."""

    def __init__(self, h, w, vi):
        self.h, self.w, self.vi = h, w, vi
        self.rendered = []       # the depth images, in call order (stored in the golden)

    def __call__(self, verts, tex, K, Rt):
        from goliath_amd import meshraster
        from oracle import mesh_ref

        v_pix = meshraster.transform(verts, K, Rt)
        _, depth, _ = mesh_ref.rasterize(v_pix.numpy(), self.vi.numpy(), self.h, self.w)
        self.rendered.append(torch.from_numpy(depth).float())
        return {"depth_img": self.rendered[-1]}


class ReplayRenderLayer:
    """Hands back stored depth images in call order (the golden's): isolates everything after the depth render."""

    replays_depth = True   # goliath_amd.urhand.shadow_maps calls it instead of rendering

    def __init__(self, h, w, depths):
        self.h, self.w, self.depths, self.calls = h, w, list(depths), 0

    def __call__(self, verts, tex, K, Rt):
        d = self.depths[self.calls]
        self.calls += 1
        return {"depth_img": d.to(verts.device)}


# ---------------------------------------------------------------------------------------------- URHand decoder stand-in
class _Refiner(nn.Module):
    def __init__(self, pose_dims):
        super().__init__()
        self.geo = nn.Conv2d(6, 2, 3, padding=1)
        self.pose = nn.Conv2d(pose_dims, 5, 3, padding=1)

    def forward(self, feat_uv, pose_cond):
        g = self.geo(feat_uv)
        return 1.5 * torch.tanh(g[:, :1]), 0.25 + 0.5 * torch.sigmoid(g[:, 1:]), self.pose(pose_cond)


class _FeatEnc(nn.Module):
    def __init__(self):
        super().__init__()
        self.z = nn.Conv2d(4, 6, 3, padding=1)
        self.gb = nn.Conv2d(4, 5, 3, padding=1)

    def forward(self, x):
        return F.avg_pool2d(self.z(x), 2), [self.gb(x)]


class ShapedConvTeacherDecoder(nn.Module):
    """Attribute layout of urhand.ConvTeacherDecoder (urhand.py:264-346) with small seeded sub-modules; S = 64."""

    def __init__(self, render_layer=None, pose_dims=4, seed=0):
        super().__init__()
        torch.manual_seed(seed)
        self.geo_fn = FakeGeo(64, 8)
        self.shadow, self.view_cond, self.refine_geo, self.feat_uv = True, True, True, "texmean"
        self.fresnel, self.scaled_albedo, self.masked_refiner_input, self.impaint_uv = 0.04, True, True, True
        self.spec_powers = [1, 16, 32]
        self.init_uv_size = 8
        self.register_buffer("raw_index_mask", (self.geo_fn.index_image != -1).any(dim=-1))
        self.featenc = _FeatEnc()
        self.texmod0 = nn.ModuleList([nn.Conv2d(7, 5, 3, padding=1), nn.Conv2d(5, 3, 3, padding=1)])
        self.texmod1 = nn.ModuleList([nn.Conv2d(6, 5, 3, padding=1, bias=False), nn.Conv2d(5, 3, 3, padding=1, bias=False)])
        self.n_layers_tex = 2
        self.joint_conv_block_tex = nn.Conv2d(5 + 3, 7, 3, padding=1)
        self.geo_refiner = _Refiner(pose_dims)
        self.rl = render_layer
        self.global_scale = nn.Parameter(torch.ones(1) * 0.3)
        self.global_albedo_scale = nn.Parameter(torch.ones(1) * -0.2)


def urhand_inputs(B=1, L=2, pose_dims=4, seed=1):
    """A curved 100 mm patch seen from 700 mm, lights ~1100 mm away with a dominant +z component (the reference hands
    [R | light_pos] to the light camera, urhand.py:415: the patch is in view when the light's world z dominates)."""
    g = torch.Generator().manual_seed(seed)
    n = 8
    ii, jj = torch.meshgrid(torch.arange(n + 1.0), torch.arange(n + 1.0), indexing="ij")
    x, y = (jj / n - 0.5) * 100.0, (ii / n - 0.5) * 100.0
    z = 0.012 * (x * x + 0.5 * y * y) + 6.0 * torch.sin(x / 14.0)          # a ridge that shadows part of the patch
    base = torch.stack([x, y, z], -1).reshape(1, -1, 3)
    verts = base + 0.5 * torch.randn(B, (n + 1) ** 2, 3, generator=g)
    light_pos = torch.tensor([[[250.0, 120.0, 1050.0], [-300.0, -60.0, 1000.0], [80.0, 320.0, 1020.0]][:L]]).repeat(B, 1, 1)
    light_pos = light_pos + 5.0 * torch.randn(B, L, 3, generator=g)
    return dict(lbs_motion=torch.randn(B, pose_dims, generator=g), id_mesh=base.repeat(B, 1, 1) * 0.9,
                tex_mean=255.0 * torch.rand(B, 3, 64, 64, generator=g), verts_rec=verts,
                cam_pos=torch.tensor([[30.0, -40.0, 700.0]]).repeat(B, 1),
                light_pos=light_pos, light_intensity=0.5 + torch.rand(B, L, 1, generator=g))


# ---------------------------------------------------------------------------------------------- teacher decoder stand-in
class OracleRaymarcher:
    """CPU stand-in for render_raymarcher.Raymarcher(with_shadow=True) through oracle/mvp_oracle.c.  Golden generation only."""

    def __init__(self, volradius, dt=1.0):
        self.volume_radius, self.dt = volradius, dt / volradius

    def __call__(self, raypos, raydir, tminmax, decout, with_shadow=False):
        from oracle import cref

        _, _, sh = cref.mvp_forward(raypos, raydir, self.dt, tminmax, decout["primpos"] / self.volume_radius,
                                    decout["primrot"], decout["primscale"], decout["primrgba"], want_raysat=False,
                                    with_shadow=True)
        return None, None, None, sh[..., 0:1] / (sh[..., 1:] + 1e-5)


class ShapedOLATRGBDecoder(nn.Module):
    """Attribute layout of hand_teacher_mvp.OLATRGBDecoder (hand_teacher_mvp.py:160-251): 4 x 4 primitives of 4 x 4 x 2
    voxels, a 64^2 light camera, a two-level UNet."""

    def __init__(self, raymarcher, volradius=200.0, seed=0):
        super().__init__()
        torch.manual_seed(seed)
        self.primsize = (4, 4, 2)
        self.n_prim_x = self.n_prim_y = 4
        self.uv_size = 16
        self.volradius = volradius
        self.raymarcher = raymarcher
        self.sizes = [16, 8]
        act = lambda: nn.LeakyReLU(0.2)
        self.enc_layers = nn.ModuleList([nn.Sequential(nn.Conv2d(14, 8, 3, padding=1), act()),
                                         nn.Sequential(nn.Conv2d(8, 8, 3, padding=1), act())])
        self.dec_layers = nn.ModuleList([nn.Sequential(nn.Conv2d(8 + 6, 8, 3, padding=1), act()),
                                         nn.Sequential(nn.Conv2d(16, 8, 3, padding=1), act())])
        y, x = torch.meshgrid(torch.arange(64), torch.arange(64), indexing="ij")
        self.register_buffer("pixel_coords", torch.stack([y, x], dim=-1).float(), persistent=False)


def teacher_inputs(B=1, L=2, seed=2, volradius=200.0):
    """16 boxes of half-extent ~30 mm on a 4 x 4 lattice inside a 200 mm volume, lights ~600 mm away."""
    g = torch.Generator().manual_seed(seed)
    K = 16
    ii, jj = torch.meshgrid(torch.arange(4.0), torch.arange(4.0), indexing="ij")
    centres = torch.stack([(jj - 1.5) * 45.0, (ii - 1.5) * 45.0, 20.0 * torch.sin(ii + jj)], -1).reshape(1, K, 3)
    primpos = centres + 3.0 * torch.randn(B, K, 3, generator=g)
    q = F.normalize(torch.randn(B, K, 4, generator=g) * 0.15 + torch.tensor([1.0, 0, 0, 0]), dim=-1)
    w, x, y, z = q.unbind(-1)
    primrot = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                           1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                           1 - 2 * (x * x + y * y)], -1).reshape(B, K, 3, 3)
    primscale = (volradius / 30.0) * (1.0 + 0.1 * torch.rand(B, K, 3, generator=g))
    valid = torch.ones(K)
    valid[5] = 0.0
    lp = torch.tensor([[[150.0, 80.0, 560.0], [-220.0, -40.0, 520.0], [60.0, 260.0, 540.0]][:L]]).repeat(B, 1, 1)
    return dict(campos=torch.tensor([[20.0, -30.0, 650.0]]).repeat(B, 1), K=None, Rt=None, primpos=primpos,
                primrot=primrot, primscale=primscale,
                primalpha=F.softplus(1.5 * torch.randn(B, 2, 16, 16, generator=g) - 1.0),   # [B, Z*1, ny*Y, nx*X]
                valid_prims=valid, joint_feat=torch.randn(B, 6, 8, 8, generator=g),
                light_pos=lp + 4.0 * torch.randn(B, L, 3, generator=g), light_intensity=0.5 + torch.rand(B, L, 3, generator=g))
