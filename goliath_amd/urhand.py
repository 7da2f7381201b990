"""Model-level bindings for BASELINE configs 4 and 5: URHand's `ConvTeacherDecoder.forward` and the MVP teacher's
`OLATRGBDecoder.forward_rgb` with their per-texel-per-light hot loops on the HIP kernels.

    conv_teacher_decoder_forward   <- /root/reference/ca_code/models/urhand.py:349-630
        * the two shadow-map evaluations (:403-417, :491-505: B*L depth renders + get_shadow_map's 18 grid_samples per
          light + exp(-x/8))                              -> depth render + gol_shadow_pcf (exp fused, texels read once)
        * Lambert / Phong^{1,16,32} light loop (:419-445) -> gol_uvlight_phong_fwd/bwd
        * GGX / Schlick light loop + physical texture (:508-567) -> gol_uvlight_ggx_fwd/bwd
      nothing of size [B,L,3,S,S] is materialised; everything else (TBN frames, the displacement / texture UNets, the
      view conditioning) is the decoder's own PyTorch sub-modules, called exactly as the reference calls them.
    olat_rgb_decoder_forward_rgb   <- /root/reference/ca_code/models/hand_teacher_mvp.py:253-494
        * the deep-shadow march (:271-358: L-fold copies of the primitive set and of a 4-channel template, an ordinary
          ray march with with_shadow=True) -> gol_mvp_shadow_march (all lights of a frame share transforms, tree and an
          alpha-only template)

Both are installed by `goliath_amd.dropin.patch_urhand()` / `patch_hand_teacher()` as methods of the reference classes
(same parameter lists: tests/test_dropin_real_classes.py).  The geometry helpers they call (`vert_normals`,
`compute_tbn_uv_given_normal`, `xyz2normals`, `tile2d`, `index`, `build_cam_rot_mat`) are looked up in the module that
defines the decoder class -- for the reference that is ca_code.models.urhand / hand_teacher_mvp themselves.
"""
import sys
from typing import Optional

import torch
import torch.nn.functional as F

from . import meshraster, mvp, shadowmap, uvlight


def _home(obj):
    """The module that defines obj's class: where the reference keeps the helpers its forward calls."""
    return sys.modules[type(obj).__module__]


def _uv_frames(self, mod, verts, normals_from=None, flip_normal=False):
    """TBN frames of the texels as a [B,S,S,3,3] image (urhand.py:366-392 and :477-486).  normals_from = None: normals
    interpolated from the vertex normals of `verts`; else a [B,3,S,S] position map whose finite-difference normals are
    used (the displaced surface)."""
    gf = self.geo_fn
    B = verts.shape[0]
    mask = (gf.index_image != -1).any(dim=-1)
    idxs = gf.index_image[mask]
    tri_uv = gf.vt[gf.v2uv[idxs, 0].to(torch.long)]
    tri_xyz = verts[:, idxs]
    if normals_from is None:
        vert_nml = mod.vert_normals(verts, gf.vi)
        vi_img = gf.vi[gf.face_index_image[mask]]
        bary = gf.bary_image[mask]
        corners = [torch.stack([mod.index(vert_nml[i], vi_img[..., c], 0) for i in range(B)]) for c in range(3)]
        n = (torch.stack(corners, dim=3) * bary[None, :, None, :]).sum(-1)
        n = n / torch.norm(n, dim=-1, keepdim=True).clamp(min=1e-5)
    else:
        n = mod.xyz2normals(normals_from)[:, :, mask].permute(0, 2, 1)
    t, b, n = mod.compute_tbn_uv_given_normal(tri_xyz, tri_uv, n)
    frames = torch.zeros((B, gf.uv_size, gf.uv_size, 3, 3), dtype=torch.float32, device=verts.device)
    frames[:, mask] = torch.stack((t, -b, -n if flip_normal else n), dim=-2)
    return frames, mask


def _normal_map(frames):
    return frames[:, :, :, 2:].permute(0, 3, 4, 1, 2)[:, 0, ...]   # [B,3,S,S]


def shadow_maps(rl, mod, verts, p_uv, nml, light_pos):
    """exp(-get_shadow_map / 8) for all B*L lights, [B,L,1,S,S] (urhand.py:403-417 / :491-505, under no_grad there too).
    One depth render per (frame, light); the texel positions / normals are read once for all L lights."""
    B, L = light_pos.shape[:2]
    with torch.no_grad():
        centre = (verts.max(1)[0] + verts.min(1)[0]) / 2
        centre = centre[:, None].expand(-1, L, -1).reshape(-1, 3)
        lpos = light_pos.reshape(-1, 3).clone()          # build_cam_rot_mat nudges degenerate positions in place
        lrot = mod.build_cam_rot_mat(lpos, centre)
        Rt = torch.cat([lrot, lpos[..., None]], dim=2)   # the reference's [R | light_pos] (urhand.py:415)
        vrep = verts[:, None].expand(-1, L, -1, -1).reshape(B * L, verts.shape[1], 3)
        Kl = torch.eye(3, device=verts.device)[None].repeat(B * L, 1, 1)   # shadowmap.py:21-26
        Kl[:, 0, 0] = Kl[:, 1, 1] = 1000.0
        Kl[:, 0, 2], Kl[:, 1, 2] = rl.w / 2, rl.h / 2
        if getattr(rl, "replays_depth", False):          # test stand-in that hands back recorded depth images
            depth = rl(vrep, torch.empty(B * L, 1, 1024, 1024, device=verts.device), Kl, Rt)["depth_img"]
        else:
            # the light cameras' depth images on gol_mesh_raster, from the topology of WHATEVER render layer the model built
            # (the reference's drtk RenderLayer, urhand.py:336-343, or meshraster.RenderLayer: both carry h, w, vi) -- depth
            # only: no uv interpolation, no texture lookup.  The layer object itself is not called, so the model's other
            # layer (the final textured render with its edge gradients, urhand.py:684) stays whatever the model made it
            depth = meshraster.rasterize(meshraster.transform(vrep, Kl, Rt), rl.vi.int(), rl.h, rl.w, with_bary=False)[1]
        sm = shadowmap.shadow_pcf(depth, Rt, p_uv, nml, exp_scale=8.0)
        return sm.reshape(B, L, 1, sm.shape[-2], sm.shape[-1])


def conv_teacher_decoder_forward(
    self,
    lbs_motion,
    id_mesh,
    tex_mean,
    verts_rec,
    cam_pos,
    light_pos,
    light_intensity,
    seam_sampler = None,
    ccm = None,
    falloff_dist = None,
    nearfield = False,
    iteration: Optional[int] = None,
):
    """Drop-in for ConvTeacherDecoder.forward (urhand.py:349-630): same arguments, same output dict."""
    mod = _home(self)
    gf = self.geo_fn
    B = verts_rec.shape[0]
    # ---- pass 1 on the LBS surface: Lambert + Phong features (urhand.py:366-445) ----
    frames, mask = _uv_frames(self, mod, verts_rec)
    p_uv = gf.to_uv(verts_rec)
    if not self.impaint_uv:
        vert_mask = (verts_rec.detach() - gf.from_uv(p_uv).detach()).abs() > 1
    nml = _normal_map(frames)
    shadow_map = shadow_maps(self.rl, mod, verts_rec, p_uv, nml, light_pos) if self.shadow else None
    diff_raw, spec_raw = uvlight.phong_features(p_uv, nml, cam_pos, light_pos, light_intensity, shadow_map,
                                                self.spec_powers)
    outputs = {"diff_feature_raw": diff_raw, "spec_feature_raw": spec_raw, "shadow_raw": shadow_map,
               "feature_normal_raw": nml}
    lint = light_intensity[..., None, None]              # [B,L,1,1,1]
    lint_scale = lint.sum(1)
    # ---- displacement / roughness from the identity texture and the pose (urhand.py:447-467) ----
    uv_id_mesh = gf.to_uv(id_mesh)
    pose_cond = mod.tile2d(lbs_motion, self.init_uv_size)
    normalized_tex = (tex_mean / 255.0) * 2.0 - 1.0
    if self.masked_refiner_input:
        uv_id_mesh[:, :, ~self.raw_index_mask] *= 0
        normalized_tex[:, :, ~self.raw_index_mask] *= 0
    if self.feat_uv == "texmean":
        refiner_in = torch.cat([normalized_tex, normalized_tex], dim=1)
    elif self.feat_uv == "texmean_geo":
        refiner_in = torch.cat([normalized_tex, uv_id_mesh], dim=1)
    elif self.feat_uv == "geo":
        refiner_in = torch.cat([uv_id_mesh, nml], dim=1)
    else:
        raise NotImplementedError("{} not supported".format(self.feat_uv))
    displacement, roughness, id_pose_feat = self.geo_refiner(refiner_in, pose_cond)
    if not self.refine_geo:
        displacement = displacement * 0
    p_uv = p_uv + nml.detach() * displacement
    verts_displaced = gf.from_uv(p_uv)
    if not self.impaint_uv:
        verts_displaced[vert_mask] = verts_rec[vert_mask]
    # ---- pass 2 on the displaced surface: GGX features + physically based texture (urhand.py:469-571) ----
    frames, _ = _uv_frames(self, mod, verts_displaced, normals_from=p_uv, flip_normal=True)
    nml = _normal_map(frames)
    v_uv = F.normalize(cam_pos[..., None, None] - p_uv, dim=1)
    if self.shadow:
        shadow_map = shadow_maps(self.rl, mod, verts_displaced, p_uv, nml, light_pos)
    if self.scaled_albedo:
        tex_mean = tex_mean.clone() * (torch.sigmoid(self.global_albedo_scale) / 2.0 + 0.7)
    feat_p, rgb_phys = uvlight.ggx_features(p_uv, nml, cam_pos, light_pos, light_intensity, roughness, tex_mean, shadow_map,
                                            fresnel=self.fresnel, spec_powers=self.spec_powers)
    outputs.update(phys_tex=rgb_phys * (torch.sigmoid(self.global_scale) / 2.0 + 0.3), roughness=roughness)
    # ---- view conditioning + texture decoder (urhand.py:573-620), the decoder's own sub-modules ----
    if self.view_cond:
        viewout = v_uv.permute(0, 2, 3, 1)[:, :, :, None, :] @ frames.transpose(-2, -1)
        viewout = viewout[:, :, :, 0, :].permute(0, 3, 1, 2)
        viewout = F.interpolate(viewout, (id_pose_feat.shape[2:]), mode="bilinear")
        id_pose_feat = torch.cat([id_pose_feat, viewout], dim=1)
    outputs.update(id_pose_conv=id_pose_feat)
    joint_feat = self.joint_conv_block_tex(id_pose_feat)
    z, gainbias = self.featenc(feat_p.detach().reshape(B, -1, feat_p.shape[-2], feat_p.shape[-1]))
    acts, x, hh = [], joint_feat, 64
    for i in range(self.n_layers_tex):                   # non-linear branch
        x = F.interpolate(x, (hh, hh), mode="bilinear", align_corners=True)
        x = F.leaky_relu(self.texmod0[i](x), 0.2)
        acts.append(x)
        hh *= 2
    x, hh = z, 64
    for i in range(self.n_layers_tex):                   # energy-in / energy-out linear branch, gated by the activations
        x = F.interpolate(x, (hh, hh), mode="bilinear", align_corners=True)
        x = self.texmod1[i](x) * acts[i]
        hh *= 2
        if i < len(gainbias):
            x = (x + gainbias[i]) * 0.707107
    rgb = F.interpolate(x, (gf.uv_size, gf.uv_size), mode="bilinear", align_corners=True)
    if self.shadow and not self.training:                # "for better shadow generalization" (urhand.py:611-613)
        rgb = rgb * ((lint / lint_scale[:, None]) * shadow_map).sum(1)
    rgb = lint_scale * rgb
    outputs.update(
        tex=rgb.clamp(min=0),
        shadow=shadow_map,
        verts_displaced=verts_displaced,
        diff_feature=feat_p[:, 0:1],
        spec_feature=feat_p[:, 1:, None],
        displacement=displacement,
        feature_normal=nml,
        interm_features2reg=gainbias,
    )
    return outputs


# ---------------------------------------------------------------------------------------------------------------------
def deep_shadow(self, mod, primpos, primrot, primscale, primalpha, valid_prims, light_pos):
    """The teacher's deep shadow volumes, hand_teacher_mvp.py:271-358: [B, L, n_prim_y, n_prim_x, 1, Z, Y, X]."""
    B, L = light_pos.shape[:2]
    X, Y, Z = self.primsize
    K = self.n_prim_x * self.n_prim_y
    with torch.no_grad():
        pts = primpos[:, valid_prims.bool()]                                   # [B, Kv, 3]
        alpha = primalpha.reshape(B, Z, 1, self.n_prim_y, Y, self.n_prim_x, X)
        alpha = alpha.permute(0, 3, 5, 1, 4, 6, 2).reshape(B, K, Z, Y, X)       # one channel; valid primitives only
        alpha = valid_prims[None, :, None, None, None] * alpha
        centre = (pts.max(1)[0] + pts.min(1)[0]) / 2
        centre = centre[:, None].expand(-1, L, -1).reshape(-1, 3)
        lpos = light_pos.reshape(-1, 3).clone()
        lrot = mod.build_cam_rot_mat(lpos, centre)
        S0, S1 = self.pixel_coords.shape[:2]
        princpt = torch.ones(B * L, 2, device=lpos.device)
        princpt[:, 0] *= S1 / 2
        princpt[:, 1] *= S0 / 2
        # drtk.transform(postex, campos, camrot, focal = 1000 I, princpt): pixel coordinates of the primitive centres
        cam = (pts[:, None].expand(-1, L, -1, -1).reshape(B * L, -1, 3) - lpos[:, None]) @ lrot.transpose(1, 2)
        pix = cam[..., :2] / cam[..., 2:3] * 1000.0 + princpt[:, None]
        extent = torch.tensor([S1, S0], device=lpos.device)
        ratio = (pix - princpt[:, None]) / (0.45 * extent[None, None])
        focal = 1000.0 / ratio.abs().max(1)[0]                                  # zoom so the hand fills 90 % of the image
        pixelcoords = self.pixel_coords[None].expand(B * L, -1, -1, -1).contiguous()
        raypos, raydir, tminmax = mvp.compute_raydirs(lpos, lrot, focal, princpt, pixelcoords, self.volradius)
        rm = self.raymarcher
        shadow = mvp.shadow_march(raypos, raydir, rm.dt, tminmax,
                                  (primpos / rm.volume_radius, primrot, primscale), alpha, L)   # [B*L, K, Z, Y, X, 1]
        return shadow.reshape(B, L, self.n_prim_y, self.n_prim_x, 1, Z, Y, X)


def olat_rgb_decoder_forward_rgb(
    self,
    campos: torch.Tensor,
    K: torch.Tensor,
    Rt: torch.Tensor,
    primpos: torch.Tensor,
    primrot: torch.Tensor,
    primscale: torch.Tensor,
    primalpha: torch.Tensor,
    valid_prims: torch.Tensor,
    joint_feat: torch.Tensor,
    light_pos: torch.Tensor,
    light_intensity: torch.Tensor,
    iteration: Optional[int] = None,
):
    """Drop-in for OLATRGBDecoder.forward_rgb (hand_teacher_mvp.py:253-494): same arguments, same output dict."""
    mod = _home(self)
    B, L = light_pos.shape[:2]
    X, Y, Z = self.primsize
    S = self.uv_size
    with torch.no_grad():
        shadow = deep_shadow(self, mod, primpos, primrot, primscale, primalpha, valid_prims, light_pos)
        shadow_feat = shadow.permute(0, 1, 5, 4, 2, 6, 3, 7).reshape(B * L, -1, S, S)     # B*L x Z*C x H*Y x W*X
        # per-voxel light / view directions in the primitives' frames (hand_teacher_mvp.py:376-432)
        dev = primpos.device
        grid = torch.stack(torch.meshgrid([torch.linspace(-1.0, 1.0, n, device=dev) for n in (Z, Y, X)], indexing="ij"))
        vox = grid.transpose(1, 3).reshape(3, -1)
        vox = primrot @ (vox[None, None] / primscale[..., None])
        vox = self.volradius * (primpos[..., None] + vox)
        vox = vox.view(B, self.n_prim_y, self.n_prim_x, 3, Z, Y, X).permute(0, 4, 3, 1, 5, 2, 6)   # B Z C H Y W X
        lvec = F.normalize(light_pos[:, :, None, :, None, None, None, None] - vox[:, None], dim=3)
        vvec = F.normalize(campos[:, None, :, None, None, None, None] - vox, dim=2)
        rot = primrot.reshape(B, self.n_prim_y, self.n_prim_x, 3, 3)
        lvec = torch.einsum("bhwef,blzehywx->blzfhywx", rot, lvec)
        vvec = torch.einsum("bhwef,bzehywx->bzfhywx", rot, vvec)
        vp = valid_prims.reshape(self.n_prim_y, self.n_prim_x)
        lvec = (vp[None, None, None, None, :, None, :, None] * lvec).reshape(B * L, -1, S, S)
        vvec = (vp[None, None, None, :, None, :, None] * vvec).reshape(B, -1, S, S)
        vvec = vvec[:, None].expand(-1, L, -1, -1, -1).reshape(B * L, -1, S, S)
        light_intensity = light_intensity[:, :, None, :, None, None]
    if self.training:
        lvec.requires_grad = True
        vvec.requires_grad = True
        shadow_feat.requires_grad = True
    x = torch.cat([lvec, vvec, 1.0 - shadow_feat], dim=1)
    joint_feat = joint_feat[:, None].expand(-1, L, -1, -1, -1).reshape(B * L, *joint_feat.shape[-3:])
    enc_acts = []
    for i, layer in enumerate(self.enc_layers):          # UNet encoder
        x = layer(x)
        enc_acts.append(x)
        if i < len(self.sizes) - 1:
            x = F.interpolate(x, scale_factor=0.5, mode="bilinear", recompute_scale_factor=True, align_corners=True)
    for i, layer in enumerate(self.dec_layers):          # UNet decoder with skip connections
        if i == 0:
            x = torch.cat([x, joint_feat], dim=1)
        else:
            skip = enc_acts[-i - 1]
            x = torch.cat([F.interpolate(x, size=skip.shape[2:4], mode="bilinear", align_corners=True), skip], dim=1)
        x = layer(x)
    tex = x.view(B, L, Z, 4, *x.shape[2:])
    if self.training and iteration is not None and iteration < 1000:
        shadowolat = shadow_feat.reshape(B, L, Z, 1, S, S)
    else:
        shadowolat = torch.sigmoid(tex[:, :, :, :1])
    texolat = 25.0 * tex[:, :, :, 1:] + 100.0
    rgb = (shadowolat * F.relu(texolat) * light_intensity).sum(1).view(B, Z, 3, S, S)
    primshadow = shadow_feat[:, :, None].expand(-1, -1, 3, -1, -1).reshape(B, L, Z, 3, S, S).sum(1) / L
    output = {"primrgb": rgb, "primshadow": primshadow}
    if self.training:
        output["texolat"] = texolat
    return output
