// project.hip -- EWA projection of 3-D Gaussians to screen space, forward + backward, gfx950.
//
// Replaces gsplat 0.1.11 project_gaussians_forward/backward kernels (not in the reference tree;
// call site /root/reference/ca_code/utils/render_gsplat.py:49-63, semantics SURVEY.md A.1/A.5).
// One lane per (view, Gaussian); the view's 3x4 matrix and intrinsics are wave-uniform and read
// from device memory through scalar loads, so a batch of B views is ONE launch with no host
// sync (the reference reads K with 4 .item() calls per view, rgca.py:123-126).
// Streaming kernel: 40 B in, ~88 B out per Gaussian -> HBM-bound.
#include "gol_common.h"

namespace {

struct M3 { float m[9]; };  // row-major

__device__ __forceinline__ M3 mul(const M3& a, const M3& b) {
  M3 o;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      o.m[r * 3 + c] = a.m[r * 3] * b.m[c] + a.m[r * 3 + 1] * b.m[3 + c] + a.m[r * 3 + 2] * b.m[6 + c];
  return o;
}
__device__ __forceinline__ M3 transpose(const M3& a) {
  return M3{{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}};
}

__device__ __forceinline__ M3 quat_to_rotmat(float qw, float qx, float qy, float qz) {
  const float s = rsqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
  const float w = qw * s, x = qx * s, y = qy * s, z = qz * s;
  return M3{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z), 2.f * (x * z + w * y),
             2.f * (x * y + w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x),
             2.f * (x * z - w * y), 2.f * (y * z + w * x), 1.f - 2.f * (x * x + y * y)}};
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// upper triangle of Sigma = M M^T, M = R(q) diag(glob_scale * scale)   
__device__ __forceinline__ void cov3d_of(const float* __restrict__ quats, const float* __restrict__ scales,
                                         float glob_scale, size_t e, float (&o_cov)[6]) {
  const float4 q = *reinterpret_cast<const float4*>(quats + 4 * e);
  const M3 R = quat_to_rotmat(q.x, q.y, q.z, q.w);
  const float s0 = glob_scale * scales[3 * e], s1 = glob_scale * scales[3 * e + 1], s2 = glob_scale * scales[3 * e + 2];
  M3 M;
#pragma unroll
  for (int r = 0; r < 3; ++r) { M.m[r * 3] = R.m[r * 3] * s0; M.m[r * 3 + 1] = R.m[r * 3 + 1] * s1; M.m[r * 3 + 2] = R.m[r * 3 + 2] * s2; }
  const M3 S3 = mul(M, transpose(M));
  o_cov[0] = S3.m[0]; o_cov[1] = S3.m[1]; o_cov[2] = S3.m[2]; o_cov[3] = S3.m[4]; o_cov[4] = S3.m[5]; o_cov[5] = S3.m[8];
}

// SURVEY A.1 tile bbox with C (int) truncation; [x0,x1) x [y0,y1) in tile units.
__device__ __forceinline__ void tile_bbox(float cx, float cy, float radius, int tiles_x, int tiles_y,
                                          float inv_block, int& x0, int& x1, int& y0, int& y1) {
  const float tcx = cx * inv_block, tcy = cy * inv_block, tr = radius * inv_block;
  x0 = clampi((int)(tcx - tr), 0, tiles_x);
  x1 = clampi((int)(tcx + tr + 1.f), 0, tiles_x);
  y0 = clampi((int)(tcy - tr), 0, tiles_y);
  y1 = clampi((int)(tcy + tr + 1.f), 0, tiles_y);
}

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
  const float* V = viewmats + 12 * b;
  const float fx = intrins[4 * b], fy = intrins[4 * b + 1], cx = intrins[4 * b + 2], cy = intrins[4 * b + 3];
  const int tiles_x = (img_w + block - 1) / block, tiles_y = (img_h + block - 1) / block;

  // defaults for culled Gaussians (gsplat allocates zeros)
  float o_cov[6] = {0, 0, 0, 0, 0, 0}, o_xy[2] = {0, 0}, o_depth = 0.f, o_con[3] = {0, 0, 0}, o_comp = 0.f;
  int o_rad = 0, o_tiles = 0;
  int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;

  const float p0 = means3d[3 * e], p1 = means3d[3 * e + 1], p2 = means3d[3 * e + 2];
  const float tx = V[0] * p0 + V[1] * p1 + V[2] * p2 + V[3];
  const float ty = V[4] * p0 + V[5] * p1 + V[6] * p2 + V[7];
  const float tz = V[8] * p0 + V[9] * p1 + V[10] * p2 + V[11];
  if (tz > clip_thresh) {
    cov3d_of(quats, scales, glob_scale, e, o_cov);

    const float lim_x = GOL_FOV_CLAMP * (0.5f * (float)img_w / fx), lim_y = GOL_FOV_CLAMP * (0.5f * (float)img_h / fy);
    const float ex = tz * fminf(lim_x, fmaxf(-lim_x, tx / tz));
    const float ey = tz * fminf(lim_y, fmaxf(-lim_y, ty / tz));
    const float rz = 1.f / tz, rz2 = rz * rz;
    const M3 J{{fx * rz, 0.f, -fx * ex * rz2, 0.f, fy * rz, -fy * ey * rz2, 0.f, 0.f, 0.f}};
    const M3 W{{V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]}};
    const M3 T = mul(J, W);
    const M3 Vc{{o_cov[0], o_cov[1], o_cov[2], o_cov[1], o_cov[3], o_cov[4], o_cov[2], o_cov[4], o_cov[5]}};
    const M3 cov = mul(mul(T, Vc), transpose(T));
    const float c00 = cov.m[0], c01 = cov.m[1], c11 = cov.m[4];
    const float det_orig = c00 * c11 - c01 * c01;
    const float a = c00 + GOL_BLUR, bq = c01, c = c11 + GOL_BLUR;
    const float det = a * c - bq * bq;
    if (det != 0.f) {
      const float inv_det = 1.f / det;
      const float bb = 0.5f * (a + c);
      const float sq = sqrtf(fmaxf(GOL_EIG_FLOOR, bb * bb - det));
      const float radius = ceilf(GOL_RADIUS_SIGMAS * sqrtf(fmaxf(bb + sq, bb - sq)));
      const float rw = 1.f / (tz + GOL_Z_EPS);
      const float px = fx * (tx * rw) + cx, py = fy * (ty * rw) + cy;
      tile_bbox(px, py, radius, tiles_x, tiles_y, 1.f / (float)block, bx0, bx1, by0, by1);
      const int area = (bx1 - bx0) * (by1 - by0);
      // gsplat writes conics before the tile-area test (forward.cu order): keep that
      o_con[0] = c * inv_det; o_con[1] = -bq * inv_det; o_con[2] = a * inv_det;
      if (area > 0) {
        o_tiles = area; o_depth = tz; o_rad = (int)radius; o_xy[0] = px; o_xy[1] = py;
        o_comp = sqrtf(fmaxf(0.f, det_orig / det));
      }
    }
  } else {
    o_cov[0] = o_cov[1] = o_cov[2] = o_cov[3] = o_cov[4] = o_cov[5] = 0.f;
  }

  if (cov3d) {   // (NULL in the fused path: the backward recomputes it from the scales and the quaternion it reads anyway)
#pragma unroll
    for (int k = 0; k < 6; ++k) cov3d[6 * e + k] = o_cov[k];
  }
  *reinterpret_cast<float2*>(xys + 2 * e) = make_float2(o_xy[0], o_xy[1]);
  depths[e] = o_depth;
  radii[e] = o_rad;
  conics[3 * e] = o_con[0]; conics[3 * e + 1] = o_con[1]; conics[3 * e + 2] = o_con[2];
  compensation[e] = o_comp;
  if (num_tiles_hit) num_tiles_hit[e] = o_tiles;
  const float op_eff = opacities ? opacities[e] * o_comp : 0.f;
  if (opac_eff) opac_eff[e] = op_eff;
  // the rasterizer's 64-byte record of this Gaussian (gol_common.h): screen position, scaled conic, effective opacity,
  // colour, depth as the 4th channel, and the per-Gaussian part of the alpha >= 1/255 reach test
  if (records)
    gol_record_write(records + e * GOL_SPLAT_RECORD, o_xy[0], o_xy[1], o_con[0], o_con[1], o_con[2], op_eff,
                     colors[3 * e], colors[3 * e + 1], colors[3 * e + 2], o_depth);
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
  float vm[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, vq[4] = {0, 0, 0, 0}, vo = 0.f;
  if (radii[e] > 0) {
    const float* V = viewmats + 12 * b;
    const float fx = intrins[4 * b], fy = intrins[4 * b + 1];
    const float p0 = means3d[3 * e], p1 = means3d[3 * e + 1], p2 = means3d[3 * e + 2];
    const float tx = V[0] * p0 + V[1] * p1 + V[2] * p2 + V[3];
    const float ty = V[4] * p0 + V[5] * p1 + V[6] * p2 + V[7];
    const float tz = V[8] * p0 + V[9] * p1 + V[10] * p2 + V[11];
    const float comp = compensation[e];
    float v_comp = v_compensation ? v_compensation[e] : 0.f;
    if (opacities) {  // opac_eff = opacity * comp  (render_gsplat.py:72)
      const float g = v_opac_eff ? v_opac_eff[e1] : 0.f;
      v_comp += g * opacities[e];
      vo = g * comp;
    }
    // project_pix vjp
    const float rw = 1.f / (tz + GOL_Z_EPS);
    const float vpx = v_xy ? fx * v_xy[e2] : 0.f, vpy = v_xy ? fy * v_xy[e2 + 1] : 0.f;
    const float vv0 = vpx * rw, vv1 = vpy * rw, vv2 = -(vpx * tx + vpy * ty) * rw * rw;
#pragma unroll
    for (int c = 0; c < 3; ++c) vm[c] = V[c] * vv0 + V[4 + c] * vv1 + V[8 + c] * vv2;
    const float vz = v_depth ? v_depth[e1] : 0.f;
    vm[0] += V[8] * vz; vm[1] += V[9] * vz; vm[2] += V[10] * vz;

    // conic (inverse cov2d) vjp: v_Sigma = -X G X
    const float X0 = conics[3 * e], X1 = conics[3 * e + 1], X2 = conics[3 * e + 2];
    const float G0 = v_conic ? v_conic[e3] : 0.f, G1 = v_conic ? 0.5f * v_conic[e3 + 1] : 0.f,
                G2 = v_conic ? v_conic[e3 + 2] : 0.f;
    const float a00 = X0 * G0 + X1 * G1, a01 = X0 * G1 + X1 * G2;
    const float a10 = X1 * G0 + X2 * G1, a11 = X1 * G1 + X2 * G2;
    float vc0 = -(a00 * X0 + a01 * X1);
    float vc1 = -(a00 * X1 + a01 * X2) - (a10 * X0 + a11 * X1);
    float vc2 = -(a10 * X1 + a11 * X2);
    {  // compensation vjp (upstream ignores the max(0,.) clamp and uses comp + 1e-6)
      const float inv_det = X0 * X2 - X1 * X1;
      const float om = 1.f - comp * comp;
      const float vsq = v_comp * 0.5f / (comp + GOL_COMP_EPS);
      vc0 += vsq * (om * X0 - GOL_BLUR * inv_det);
      vc1 += 2.f * vsq * (om * X1);
      vc2 += vsq * (om * X2 - GOL_BLUR * inv_det);
    }
    // EWA vjp with the UNCLAMPED camera-space point, as upstream
    const float rz = 1.f / tz, rz2 = rz * rz, rz3 = rz2 * rz;
    const M3 J{{fx * rz, 0.f, -fx * tx * rz2, 0.f, fy * rz, -fy * ty * rz2, 0.f, 0.f, 0.f}};
    const M3 W{{V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]}};
    const M3 T = mul(J, W);
    float c3[6];
    if (cov3d) {
#pragma unroll
      for (int k = 0; k < 6; ++k) c3[k] = cov3d[6 * e + k];
    } else {
      cov3d_of(quats, scales, glob_scale, e, c3);   // the forward's own arithmetic: identical values
    }
    const M3 Vc{{c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]}};
    const M3 Gc{{vc0, 0.5f * vc1, 0.f, 0.5f * vc1, vc2, 0.f, 0.f, 0.f, 0.f}};
    const M3 vV = mul(mul(transpose(T), Gc), T);
    M3 vT = mul(mul(Gc, T), Vc);
#pragma unroll
    for (int k = 0; k < 9; ++k) vT.m[k] *= 2.f;
    const float vc3[6] = {vV.m[0], vV.m[1] + vV.m[3], vV.m[2] + vV.m[6], vV.m[4], vV.m[5] + vV.m[7], vV.m[8]};
    const M3 vJ = mul(vT, transpose(W));
    const float vt0 = -fx * rz2 * vJ.m[2], vt1 = -fy * rz2 * vJ.m[5];
    const float vt2 = -fx * rz2 * vJ.m[0] + 2.f * fx * tx * rz3 * vJ.m[2] - fy * rz2 * vJ.m[4] +
                      2.f * fy * ty * rz3 * vJ.m[5];
#pragma unroll
    for (int c = 0; c < 3; ++c) vm[c] += vt0 * V[c] + vt1 * V[4 + c] + vt2 * V[8 + c];

    // cov3d = M M^T, M = R(q) diag(s): vjp to scale and (normalised) quaternion
    const M3 vVs{{vc3[0], 0.5f * vc3[1], 0.5f * vc3[2], 0.5f * vc3[1], vc3[3], 0.5f * vc3[4],
                  0.5f * vc3[2], 0.5f * vc3[4], vc3[5]}};
    const float4 q = *reinterpret_cast<const float4*>(quats + 4 * e);
    const M3 R = quat_to_rotmat(q.x, q.y, q.z, q.w);
    const float sc[3] = {glob_scale * scales[3 * e], glob_scale * scales[3 * e + 1], glob_scale * scales[3 * e + 2]};
    M3 M;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) M.m[r * 3 + c] = R.m[r * 3 + c] * sc[c];
    M3 vM = mul(vVs, M);
#pragma unroll
    for (int k = 0; k < 9; ++k) vM.m[k] *= 2.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      vs[c] = (R.m[c] * vM.m[c] + R.m[3 + c] * vM.m[3 + c] + R.m[6 + c] * vM.m[6 + c]) * glob_scale;
    float vR[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) vR[r * 3 + c] = vM.m[r * 3 + c] * sc[c];
    const float s = rsqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float w = q.x * s, x = q.y * s, y = q.z * s, z = q.w * s;
#define VR(r, c) vR[(r) * 3 + (c)]
    vq[0] = 2.f * (x * (VR(2, 1) - VR(1, 2)) + y * (VR(0, 2) - VR(2, 0)) + z * (VR(1, 0) - VR(0, 1)));
    vq[1] = 2.f * (-2.f * x * (VR(1, 1) + VR(2, 2)) + y * (VR(1, 0) + VR(0, 1)) + z * (VR(2, 0) + VR(0, 2)) +
                   w * (VR(2, 1) - VR(1, 2)));
    vq[2] = 2.f * (x * (VR(1, 0) + VR(0, 1)) - 2.f * y * (VR(0, 0) + VR(2, 2)) + z * (VR(2, 1) + VR(1, 2)) +
                   w * (VR(0, 2) - VR(2, 0)));
    vq[3] = 2.f * (x * (VR(2, 0) + VR(0, 2)) + y * (VR(2, 1) + VR(1, 2)) - 2.f * z * (VR(0, 0) + VR(1, 1)) +
                   w * (VR(1, 0) - VR(0, 1)));
#undef VR
  }
  v_mean3d[3 * e] = vm[0]; v_mean3d[3 * e + 1] = vm[1]; v_mean3d[3 * e + 2] = vm[2];
  v_scale[3 * e] = vs[0]; v_scale[3 * e + 1] = vs[1]; v_scale[3 * e + 2] = vs[2];
  *reinterpret_cast<float4*>(v_quat + 4 * e) = make_float4(vq[0], vq[1], vq[2], vq[3]);
  if (v_opacity) v_opacity[e] = vo;
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
