"""GPU: the mesh z-buffer rasterizer (drtk stand-in, csrc/meshraster.hip) against the numpy oracle of the same stated
conventions (oracle/mesh_ref.py; drtk itself is absent: parity unpinned), the RenderLayer mirror, and the shadow map
end to end: get_shadow_map(our RenderLayer, ...) reproduces the golden of the reference's get_shadow_map when the depth
render the golden was made with is replaced by our render of the same plane."""
import math

import numpy as np
import pytest
import torch

from scenes import icosphere, look_at_viewmat, rel_l2

pytestmark = pytest.mark.gpu


def _camera(B, H, W, dist=3.0):
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = 0.9 * W
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = W / 2.0, H / 2.0, 1.0
    Rt = torch.stack([look_at_viewmat((dist * math.sin(0.7 * b), 0.3 * b, -dist * math.cos(0.7 * b))) for b in range(B)])
    return K, Rt


@pytest.mark.parametrize("H,W,subdiv", [(96, 80, 2), (130, 67, 3), (16, 16, 0)])
def test_mesh_raster_matches_numpy_oracle(H, W, subdiv):
    from goliath_amd import meshraster
    from oracle import mesh_ref

    B = 2
    verts, faces = icosphere(subdiv)
    g = torch.Generator().manual_seed(subdiv)
    verts = verts[None].repeat(B, 1, 1) * (1.0 + 0.15 * torch.rand(B, verts.shape[0], 1, generator=g))  # bumpy, two poses
    K, Rt = _camera(B, H, W)
    v_pix = meshraster.transform(verts.cuda(), K.cuda(), Rt.cuda())
    index, depth, bary = meshraster.rasterize(v_pix, faces.cuda(), H, W)
    ri, rd, rb = mesh_ref.rasterize(v_pix.cpu().numpy(), faces.numpy(), H, W)
    index, depth, bary = index.cpu().numpy(), depth.cpu().numpy(), bary.cpu().numpy()
    covered = (ri >= 0).mean()
    assert covered > 0.15
    # samples within fp32 rounding of an edge may flip between neighbouring faces / background
    same = index == ri
    assert same.mean() > 0.998, same.mean()
    # fp32 edge equations on pixel coordinates ~1e2 (kernel) vs float64 (oracle): grazing faces amplify the rounding
    assert np.abs(depth - rd)[same].max() < 1e-3 * rd.max()
    assert np.median(np.abs(depth - rd)[same & (ri >= 0)]) < 2e-5 * rd.max()
    assert np.abs(bary - rb)[np.broadcast_to(same[:, None], bary.shape)].max() < 5e-3
    inside = same & (ri >= 0)
    assert np.abs(bary.sum(1) - 1.0)[inside].max() < 1e-5
    # a flipped sample still picks a face adjacent in depth: depth error stays small
    assert np.abs(depth - rd)[(index >= 0) & (ri >= 0)].max() < 0.05 * rd.max()


def test_render_layer_mirror_and_its_gradients():
    from goliath_amd import meshraster

    H, W = 64, 48
    verts, faces = icosphere(1)
    vt = torch.rand(verts.shape[0], 2)
    rl = meshraster.RenderLayer(H, W, faces, vt, faces).cuda()
    K, Rt = _camera(1, H, W)
    tex = torch.rand(1, 3, 32, 32).cuda()
    with torch.no_grad():
        out = rl(verts[None].cuda(), tex, K.cuda(), Rt.cuda())
    assert set(out) == {"render", "depth_img", "v_pix", "vt_img", "index_img", "bary_img", "mask"}
    assert out["render"].shape == (1, 3, H, W) and out["depth_img"].shape == (1, H, W)
    assert float(out["mask"].mean()) > 0.1 and float((out["render"] * (1 - out["mask"])).abs().max()) == 0.0
    # uv interpolation: inside the silhouette vt_img is a convex combination of the face's uv corners
    m = out["mask"][0, 0] > 0
    assert float(out["vt_img"][0][:, m].abs().max()) <= 1.0 + 1e-5
    # round 4: differentiable like drtk's layer (render_drtk.py:42-70).  (a) the differentiable re-evaluation of depth /
    # barycentrics is the rasterizer's own arithmetic; (b) gradients reach the texture AND the vertices; (c) the whole
    # backward equals the same PyTorch code run on the CPU on the numpy oracle's images (the edge estimator itself is
    # checked against a supersampled render in tests/test_mesh_edge_grad.py)
    vg = verts[None].cuda().requires_grad_(True)
    tg = tex.clone().requires_grad_(True)
    res = rl(vg, tg, K.cuda(), Rt.cuda(), edge_grad=True)
    assert rel_l2(res["depth_img"], out["depth_img"]) < 1e-5 and rel_l2(res["bary_img"], out["bary_img"]) < 1e-4
    assert torch.equal(res["index_img"], out["index_img"])
    g = torch.Generator().manual_seed(2)
    up = torch.randn(1, 3, H, W, generator=g)
    (res["render"] * up.cuda()).sum().backward()
    assert float(vg.grad.abs().sum()) > 0 and float(tg.grad.abs().sum()) > 0 and bool(torch.isfinite(vg.grad).all())
    from oracle import mesh_ref

    vc = verts[None].clone().requires_grad_(True)
    tc = tex.cpu().clone().requires_grad_(True)
    v_pix = meshraster.transform(vc, K, Rt)
    idx, _, _ = mesh_ref.rasterize(v_pix.detach().numpy(), faces.numpy(), H, W)
    idx = torch.from_numpy(idx)
    assert float((idx.cuda() != res["index_img"]).float().mean()) < 2e-3     # (samples on an edge may flip)
    depth_c, bary_c = meshraster.render(v_pix, faces, idx)
    vt_img = meshraster.interpolate((vt * 2.0 - 1.0)[None], faces, idx, bary_c)
    img = torch.nn.functional.grid_sample(tc, vt_img.permute(0, 2, 3, 1), mode="bilinear", align_corners=False) * (idx != -1)[:, None].float()
    img = meshraster.edge_grad_estimator(v_pix, faces, bary_c, img, idx, depth_c)
    (img * up).sum().backward()
    assert rel_l2(tg.grad, tc.grad) < 2e-2 and rel_l2(vg.grad, vc.grad) < 5e-2   # identical but for the flipped edge samples


def test_shadow_map_with_our_depth_render_of_a_plane():
    """get_shadow_map (goliath_amd.shadowmap, reference signature) driven by our RenderLayer: a big occluder quad between
    the light and a receiver plane shadows exactly the texels behind it."""
    from goliath_amd import meshraster, shadowmap

    S = 64
    # receiver: the z = 0 plane, texels on [-1, 1]^2; occluder: the quad [-0.5, 0.5]^2 at z = -1; light at z = -4
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing="ij")
    postex = torch.stack([xs, ys, torch.zeros_like(xs)])[None].cuda()
    verts = torch.tensor([[-0.5, -0.5, -1.0], [0.5, -0.5, -1.0], [0.5, 0.5, -1.0], [-0.5, 0.5, -1.0]])[None].cuda()
    faces = torch.tensor([[0, 1, 2], [0, 2, 3]])
    rl = meshraster.RenderLayer(1024, 1024, faces, torch.zeros(4, 2), faces).cuda()
    Rt = torch.cat([torch.eye(3), torch.tensor([[0.0], [0.0], [4.0]])], 1)[None].cuda()  # camera at z = -4 looking +z
    with torch.no_grad():
        sm = shadowmap.get_shadow_map(rl, Rt, None, verts, postex, None)  # (no back-face blend: shadowmap.py:60-63)
    sm = sm.reshape(S, S).cpu()
    # the map holds the PCF-averaged depth excess over the nearest occluder (shadowmap.py:84-90): 4 - 3 = 1 behind the
    # quad, 0 where the light sees the receiver.  Similar triangles: the quad's shadow on z = 0 is [-2/3, 2/3]^2
    inside = (xs.abs() < 0.6) & (ys.abs() < 0.6)
    outside = (xs.abs() > 0.75) | (ys.abs() > 0.75)
    assert abs(float(sm[inside].mean()) - 1.0) < 1e-3 and float(sm[inside].min()) > 0.99
    assert float(sm[outside].abs().max()) == 0.0
