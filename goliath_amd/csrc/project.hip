// project.hip -- EWA projection of 3-D Gaussians to screen space, forward + backward, gfx950.
//
// Replaces gsplat 0.1.11 project_gaussians_forward/backward kernels (not in the reference tree;
// call site /root/reference/ca_code/utils/render_gsplat.py:49-63, semantics SURVEY.md A.1/A.5).
// One lane per (view, Gaussian); the view's 3x4 matrix and intrinsics are wave-uniform and read
// from device memory through scalar loads, so a batch of B views is ONE launch with no host
// sync (the reference reads K with 4 .item() calls per view, rgca.py:123-126).
// Streaming kernel: 40 B in, ~88 B out per Gaussian -> HBM-bound.
#include "gol_project.h"

namespace {

using namespace gol_proj;

__global__ __launch_bounds__(256) void project_fwd_kernel(
    int N, const float* __restrict__ means3d, const float* __restrict__ scales, float glob_scale,
    const float* __restrict__ quats, const float* __restrict__ viewmats,
    const float* __restrict__ intrins, int img_h, int img_w, int block, float clip_thresh,
    float* __restrict__ cov3d, float* __restrict__ xys, float* __restrict__ depths,
    int32_t* __restrict__ radii, float* __restrict__ conics, float* __restrict__ compensation,
    int32_t* __restrict__ num_tiles_hit, const float* __restrict__ opacities,
    float* __restrict__ opac_eff, const float* __restrict__ colors, float* __restrict__ records) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const size_t e = (size_t)b * N + i;
  const View cam = view_of(viewmats, intrins, b, img_h, img_w, block, clip_thresh);
  const float p[3] = {means3d[3 * e], means3d[3 * e + 1], means3d[3 * e + 2]};
  const float4 qv = *reinterpret_cast<const float4*>(quats + 4 * e);
  const float q[4] = {qv.x, qv.y, qv.z, qv.w};
  const float s[3] = {glob_scale * scales[3 * e], glob_scale * scales[3 * e + 1], glob_scale * scales[3 * e + 2]};
  const Projected o = project_point(p, q, s, cam);

  if (cov3d) {   // (NULL in the fused path: the backward recomputes it from the scales and the quaternion it reads anyway)
#pragma unroll
    for (int k = 0; k < 6; ++k) cov3d[6 * e + k] = o.cov[k];
  }
  *reinterpret_cast<float2*>(xys + 2 * e) = make_float2(o.xy[0], o.xy[1]);
  depths[e] = o.depth;
  radii[e] = o.radius;
  conics[3 * e] = o.conic[0]; conics[3 * e + 1] = o.conic[1]; conics[3 * e + 2] = o.conic[2];
  compensation[e] = o.comp;
  if (num_tiles_hit) num_tiles_hit[e] = o.tiles;
  const float op_eff = opacities ? opacities[e] * o.comp : 0.f;
  if (opac_eff) opac_eff[e] = op_eff;
  // the rasterizer's 64-byte record of this Gaussian (gol_common.h): screen position, scaled conic, effective opacity,
  // colour, depth as the 4th channel, and the per-Gaussian part of the alpha >= 1/255 reach test
  if (records)
    gol_record_write(records + e * GOL_SPLAT_RECORD, o.xy[0], o.xy[1], o.conic[0], o.conic[1], o.conic[2], op_eff,
                     colors[3 * e], colors[3 * e + 1], colors[3 * e + 2], o.depth);
}

__global__ __launch_bounds__(256) void project_bwd_kernel(
    int N, const float* __restrict__ means3d, const float* __restrict__ scales, float glob_scale,
    const float* __restrict__ quats, const float* __restrict__ viewmats,
    const float* __restrict__ intrins, const float* __restrict__ cov3d,
    const int32_t* __restrict__ radii, const float* __restrict__ conics,
    const float* __restrict__ compensation, const float* __restrict__ v_xy,
    const float* __restrict__ v_depth, const float* __restrict__ v_conic,
    const float* __restrict__ v_compensation, const float* __restrict__ opacities,
    const float* __restrict__ v_opac_eff, int gs, float* __restrict__ v_mean3d, float* __restrict__ v_scale,
    float* __restrict__ v_quat, float* __restrict__ v_opacity, const float* __restrict__ v_rgb_rec,
    float* __restrict__ v_colors_out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const size_t e = (size_t)b * N + i;
  if (v_colors_out) {  // the colour gradient out of the record (same cache line as the fields read below) as a dense [B,N,3]
    const float* r = v_rgb_rec + e * gs;
    v_colors_out[3 * e] = r[0]; v_colors_out[3 * e + 1] = r[1]; v_colors_out[3 * e + 2] = r[2];
  }
  // upstream gradients: dense arrays (gs == 0) or fields of per-Gaussian records of gs floats (rasterize_bwd)
  const size_t e1 = gs ? e * gs : e, e2 = gs ? e * gs : 2 * e, e3 = gs ? e * gs : 3 * e;
  ProjGrad g;
  g.mean[0] = g.mean[1] = g.mean[2] = 0.f; g.scale[0] = g.scale[1] = g.scale[2] = 0.f;
  g.quat[0] = g.quat[1] = g.quat[2] = g.quat[3] = 0.f; g.opacity = 0.f;
  if (radii[e] > 0) {
    const View cam = view_of(viewmats, intrins, b, 0, 0, 16, 0.f);
    const float p[3] = {means3d[3 * e], means3d[3 * e + 1], means3d[3 * e + 2]};
    const float4 qv = *reinterpret_cast<const float4*>(quats + 4 * e);
    const float q[4] = {qv.x, qv.y, qv.z, qv.w};
    const float s[3] = {glob_scale * scales[3 * e], glob_scale * scales[3 * e + 1], glob_scale * scales[3 * e + 2]};
    const float X[3] = {conics[3 * e], conics[3 * e + 1], conics[3 * e + 2]};
    ProjUp up;
    up.xy[0] = v_xy ? v_xy[e2] : 0.f; up.xy[1] = v_xy ? v_xy[e2 + 1] : 0.f;
    up.depth = v_depth ? v_depth[e1] : 0.f;
    up.conic[0] = v_conic ? v_conic[e3] : 0.f; up.conic[1] = v_conic ? v_conic[e3 + 1] : 0.f;
    up.conic[2] = v_conic ? v_conic[e3 + 2] : 0.f;
    up.comp = v_compensation ? v_compensation[e] : 0.f;
    up.opac_eff = (opacities && v_opac_eff) ? v_opac_eff[e1] : 0.f;
    g = project_vjp(p, q, s, glob_scale, cam, X, compensation[e], up, opacities != nullptr,
                    opacities ? opacities[e] : 0.f, cov3d ? cov3d + 6 * e : nullptr);
  }
  v_mean3d[3 * e] = g.mean[0]; v_mean3d[3 * e + 1] = g.mean[1]; v_mean3d[3 * e + 2] = g.mean[2];
  v_scale[3 * e] = g.scale[0]; v_scale[3 * e + 1] = g.scale[1]; v_scale[3 * e + 2] = g.scale[2];
  *reinterpret_cast<float4*>(v_quat + 4 * e) = make_float4(g.quat[0], g.quat[1], g.quat[2], g.quat[3]);
  if (v_opacity) v_opacity[e] = g.opacity;
}

}  // namespace

extern "C" int gol_project_fwd(int B, int N, const float* means3d, const float* scales, float glob_scale,
                               const float* quats, const float* viewmats, const float* intrins, int img_h,
                               int img_w, int block, float clip_thresh, float* cov3d, float* xys,
                               float* depths, int32_t* radii, float* conics, float* compensation,
                               int32_t* num_tiles_hit, const float* opacities, float* opac_eff, const float* colors,
                               float* records, void* stream) {
  GOL_REQUIRE(B >= 0 && N >= 0, "negative size");
  GOL_REQUIRE(block > 1 && block <= 16, "block_width must be between 2 and 16");
  GOL_REQUIRE(img_h > 0 && img_w > 0, "empty image");
  if (B == 0 || N == 0) return GOL_OK;
  GOL_REQUIRE(means3d && scales && quats && viewmats && intrins, "null input");
  GOL_REQUIRE(xys && depths && radii && conics && compensation, "null output");
  GOL_REQUIRE((opac_eff == nullptr) || (opacities != nullptr), "opac_eff needs opacities");
  GOL_REQUIRE((records == nullptr) || (opacities != nullptr && colors != nullptr), "records need opacities and colors");
  GOL_REQUIRE(B <= 65535, "B > 65535");
  dim3 grid(gol_cdiv(N, 256), B);
  project_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(
      N, means3d, scales, glob_scale, quats, viewmats, intrins, img_h, img_w, block, clip_thresh, cov3d, xys,
      depths, radii, conics, compensation, num_tiles_hit, opacities, opac_eff, colors, records);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

static int project_bwd_launch(int B, int N, const float* means3d, const float* scales, float glob_scale,
                              const float* quats, const float* viewmats, const float* intrins,
                              const float* cov3d, const int32_t* radii, const float* conics,
                              const float* compensation, const float* v_xy, const float* v_depth,
                              const float* v_conic, const float* v_compensation, const float* opacities,
                              const float* v_opac_eff, int grad_stride, float* v_mean3d, float* v_scale,
                              float* v_quat, float* v_opacity, const float* v_rgb_rec, float* v_colors_out, void* stream) {
  GOL_REQUIRE(B >= 0 && N >= 0, "negative size");
  GOL_REQUIRE(grad_stride >= 0, "negative grad_stride");
  if (B == 0 || N == 0) return GOL_OK;
  GOL_REQUIRE(means3d && scales && quats && viewmats && intrins && radii && conics && compensation, "null input");
  GOL_REQUIRE(v_mean3d && v_scale && v_quat, "null output");
  GOL_REQUIRE(!opacities || v_opacity, "v_opacity must be given with opacities");
  GOL_REQUIRE(B <= 65535, "B > 65535");
  dim3 grid(gol_cdiv(N, 256), B);
  project_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(
      N, means3d, scales, glob_scale, quats, viewmats, intrins, cov3d, radii, conics, compensation, v_xy,
      v_depth, v_conic, v_compensation, opacities, v_opac_eff, grad_stride, v_mean3d, v_scale, v_quat, v_opacity,
      v_rgb_rec, v_colors_out);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_project_bwd(int B, int N, const float* means3d, const float* scales, float glob_scale,
                               const float* quats, const float* viewmats, const float* intrins,
                               const float* cov3d, const int32_t* radii, const float* conics,
                               const float* compensation, const float* v_xy, const float* v_depth,
                               const float* v_conic, const float* v_compensation, const float* opacities,
                               const float* v_opac_eff, int grad_stride, float* v_mean3d, float* v_scale,
                               float* v_quat, float* v_opacity, void* stream) {
  return project_bwd_launch(B, N, means3d, scales, glob_scale, quats, viewmats, intrins, cov3d, radii, conics, compensation,
                            v_xy, v_depth, v_conic, v_compensation, opacities, v_opac_eff, grad_stride, v_mean3d, v_scale,
                            v_quat, v_opacity, nullptr, nullptr, stream);
}

extern "C" int gol_project_bwd_records(int B, int N, const float* means3d, const float* scales, float glob_scale,
                                       const float* quats, const float* viewmats, const float* intrins,
                                       const int32_t* radii, const float* conics, const float* compensation,
                                       const float* opacities, const float* grad_records, int with_depth, float* v_mean3d,
                                       float* v_scale, float* v_quat, float* v_opacity, float* v_colors, void* stream) {
  GOL_REQUIRE(grad_records != nullptr, "null gradient records");
  const float* g = grad_records;
  return project_bwd_launch(B, N, means3d, scales, glob_scale, quats, viewmats, intrins, nullptr, radii, conics, compensation,
                            g + 4, with_depth ? g + 9 : nullptr, g + 6, nullptr, opacities, g + 3, GOL_GRAD_RECORD, v_mean3d,
                            v_scale, v_quat, v_opacity, v_colors ? g : nullptr, v_colors, stream);
}
