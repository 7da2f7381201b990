// gol_project.h -- the EWA projection of ONE Gaussian and its vector-Jacobian product as device functions on values in
// registers, shared by the stand-alone projection kernels (project.hip: gsplat 0.1.11 project_gaussians_forward /
// backward, call site /root/reference/ca_code/utils/render_gsplat.py:49-63, semantics SURVEY.md A.1 / A.5) and by the
// shading kernels that run the projection as their epilogue / prologue (shade.hip: rgca.py:505-588 -> render_gsplat.py:49-63
// without the round trip of the Gaussian attributes through HBM).  One arithmetic, two callers: the fused path's tile
// lists and images are the separate path's.
#pragma once
#include "gol_common.h"

// Both callers must produce the SAME bits (the fused path's tile lists are the separate path's): with hipcc's default
// -ffp-contract=fast the backend fuses a multiply into an add wherever the surrounding code lets it, i.e. differently in the
// two kernels this code is inlined into.  contract(on) fuses only within an expression, in the front end: one result.
#define GOL_FP_DETERMINISTIC _Pragma("clang fp contract(on)")

namespace gol_proj {

struct M3 { float m[9]; };  // row-major

__device__ __forceinline__ M3 mul(const M3& a, const M3& b) {
  GOL_FP_DETERMINISTIC;
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
  GOL_FP_DETERMINISTIC;
  const float s = rsqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
  const float w = qw * s, x = qx * s, y = qy * s, z = qz * s;
  return M3{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z), 2.f * (x * z + w * y),
             2.f * (x * y + w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x),
             2.f * (x * z - w * y), 2.f * (y * z + w * x), 1.f - 2.f * (x * x + y * y)}};
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// upper triangle of Sigma = M M^T, M = R(q) diag(s)   (SURVEY A.1); s already carries glob_scale
__device__ __forceinline__ void cov3d_of(const float (&q)[4], const float (&s)[3], float (&o_cov)[6]) {
  GOL_FP_DETERMINISTIC;
  const M3 R = quat_to_rotmat(q[0], q[1], q[2], q[3]);
  M3 M;
#pragma unroll
  for (int r = 0; r < 3; ++r) { M.m[r * 3] = R.m[r * 3] * s[0]; M.m[r * 3 + 1] = R.m[r * 3 + 1] * s[1]; M.m[r * 3 + 2] = R.m[r * 3 + 2] * s[2]; }
  const M3 S3 = mul(M, transpose(M));
  o_cov[0] = S3.m[0]; o_cov[1] = S3.m[1]; o_cov[2] = S3.m[2]; o_cov[3] = S3.m[4]; o_cov[4] = S3.m[5]; o_cov[5] = S3.m[8];
}

// SURVEY A.1 tile bbox with C (int) truncation; [x0,x1) x [y0,y1) in tile units.
__device__ __forceinline__ void tile_bbox(float cx, float cy, float radius, int tiles_x, int tiles_y,
                                          float inv_block, int& x0, int& x1, int& y0, int& y1) {
  GOL_FP_DETERMINISTIC;
  const float tcx = cx * inv_block, tcy = cy * inv_block, tr = radius * inv_block;
  x0 = clampi((int)(tcx - tr), 0, tiles_x);
  x1 = clampi((int)(tcx + tr + 1.f), 0, tiles_x);
  y0 = clampi((int)(tcy - tr), 0, tiles_y);
  y1 = clampi((int)(tcy + tr + 1.f), 0, tiles_y);
}

// the camera of a view: wave-uniform (scalar loads)
struct View {
  const float* V;  // 3x4 world -> camera, row-major
  float fx, fy, cx, cy;
  int img_h, img_w, block;
  float clip;
};

__device__ __forceinline__ View view_of(const float* __restrict__ viewmats, const float* __restrict__ intrins, int b,
                                        int img_h, int img_w, int block, float clip) {
  return View{viewmats + 12 * b, intrins[4 * b], intrins[4 * b + 1], intrins[4 * b + 2], intrins[4 * b + 3],
              img_h, img_w, block, clip};
}

// what gsplat's forward writes for one Gaussian; zeros for a culled one (gsplat allocates zeros)
struct Projected {
  float cov[6], xy[2], depth, conic[3], comp;
  int radius, tiles;
};

// p: world position, q: quaternion (w, x, y, z; normalised inside), s: scales x glob_scale
__device__ __forceinline__ Projected project_point(const float (&p)[3], const float (&q)[4], const float (&s)[3],
                                                   const View& cam) {
  GOL_FP_DETERMINISTIC;
  Projected o;
#pragma unroll
  for (int k = 0; k < 6; ++k) o.cov[k] = 0.f;
  o.xy[0] = o.xy[1] = 0.f; o.depth = 0.f; o.conic[0] = o.conic[1] = o.conic[2] = 0.f; o.comp = 0.f;
  o.radius = 0; o.tiles = 0;
  const float* V = cam.V;
  const float fx = cam.fx, fy = cam.fy;
  const int tiles_x = (cam.img_w + cam.block - 1) / cam.block, tiles_y = (cam.img_h + cam.block - 1) / cam.block;
  const float p0 = p[0], p1 = p[1], p2 = p[2];
  const float tx = V[0] * p0 + V[1] * p1 + V[2] * p2 + V[3];
  const float ty = V[4] * p0 + V[5] * p1 + V[6] * p2 + V[7];
  const float tz = V[8] * p0 + V[9] * p1 + V[10] * p2 + V[11];
  if (tz > cam.clip) {
    cov3d_of(q, s, o.cov);
    const float lim_x = GOL_FOV_CLAMP * (0.5f * (float)cam.img_w / fx), lim_y = GOL_FOV_CLAMP * (0.5f * (float)cam.img_h / fy);
    const float ex = tz * fminf(lim_x, fmaxf(-lim_x, tx / tz));
    const float ey = tz * fminf(lim_y, fmaxf(-lim_y, ty / tz));
    const float rz = 1.f / tz, rz2 = rz * rz;
    const M3 J{{fx * rz, 0.f, -fx * ex * rz2, 0.f, fy * rz, -fy * ey * rz2, 0.f, 0.f, 0.f}};
    const M3 W{{V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]}};
    const M3 T = mul(J, W);
    const M3 Vc{{o.cov[0], o.cov[1], o.cov[2], o.cov[1], o.cov[3], o.cov[4], o.cov[2], o.cov[4], o.cov[5]}};
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
      const float px = fx * (tx * rw) + cam.cx, py = fy * (ty * rw) + cam.cy;
      int bx0, bx1, by0, by1;
      tile_bbox(px, py, radius, tiles_x, tiles_y, 1.f / (float)cam.block, bx0, bx1, by0, by1);
      const int area = (bx1 - bx0) * (by1 - by0);
      // gsplat writes conics before the tile-area test (forward.cu order): keep that
      o.conic[0] = c * inv_det; o.conic[1] = -bq * inv_det; o.conic[2] = a * inv_det;
      if (area > 0) {
        o.tiles = area; o.depth = tz; o.radius = (int)radius; o.xy[0] = px; o.xy[1] = py;
        o.comp = sqrtf(fmaxf(0.f, det_orig / det));
      }
    }
  }
  return o;
}

// upstream gradients of one Gaussian's projection outputs (zeros where nothing flows)
struct ProjUp {
  float xy[2], depth, conic[3], comp, opac_eff;
};
struct ProjGrad {
  float mean[3], scale[3], quat[4], opacity;
};

// vjp of project_point (+ opac_eff = opacity * comp, render_gsplat.py:72) for a Gaussian with radius > 0.
// s = scales x glob_scale; cov3d given (the stand-alone operator saved it) or recomputed with the forward's arithmetic.
__device__ __forceinline__ ProjGrad project_vjp(const float (&p)[3], const float (&q)[4], const float (&s)[3],
                                                float glob_scale, const View& cam, const float (&X)[3], float comp,
                                                const ProjUp& up, bool has_opacity, float opacity,
                                                const float* __restrict__ cov3d_saved) {
  GOL_FP_DETERMINISTIC;
  ProjGrad g;
  const float* V = cam.V;
  const float fx = cam.fx, fy = cam.fy;
  const float p0 = p[0], p1 = p[1], p2 = p[2];
  const float tx = V[0] * p0 + V[1] * p1 + V[2] * p2 + V[3];
  const float ty = V[4] * p0 + V[5] * p1 + V[6] * p2 + V[7];
  const float tz = V[8] * p0 + V[9] * p1 + V[10] * p2 + V[11];
  float v_comp = up.comp;
  g.opacity = 0.f;
  if (has_opacity) {  // opac_eff = opacity * comp  (render_gsplat.py:72)
    v_comp += up.opac_eff * opacity;
    g.opacity = up.opac_eff * comp;
  }
  // project_pix vjp
  const float rw = 1.f / (tz + GOL_Z_EPS);
  const float vpx = fx * up.xy[0], vpy = fy * up.xy[1];
  const float vv0 = vpx * rw, vv1 = vpy * rw, vv2 = -(vpx * tx + vpy * ty) * rw * rw;
  float vm[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) vm[c] = V[c] * vv0 + V[4 + c] * vv1 + V[8 + c] * vv2;
  const float vz = up.depth;
  vm[0] += V[8] * vz; vm[1] += V[9] * vz; vm[2] += V[10] * vz;

  // conic (inverse cov2d) vjp: v_Sigma = -X G X
  const float X0 = X[0], X1 = X[1], X2 = X[2];
  const float G0 = up.conic[0], G1 = 0.5f * up.conic[1], G2 = up.conic[2];
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
  if (cov3d_saved) {
#pragma unroll
    for (int k = 0; k < 6; ++k) c3[k] = cov3d_saved[k];
  } else {
    cov3d_of(q, s, c3);   // the forward's own arithmetic: identical values
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
  for (int c = 0; c < 3; ++c) g.mean[c] = vm[c] + (vt0 * V[c] + vt1 * V[4 + c] + vt2 * V[8 + c]);

  // cov3d = M M^T, M = R(q) diag(s): vjp to scale and (normalised) quaternion
  const M3 vVs{{vc3[0], 0.5f * vc3[1], 0.5f * vc3[2], 0.5f * vc3[1], vc3[3], 0.5f * vc3[4],
                0.5f * vc3[2], 0.5f * vc3[4], vc3[5]}};
  const M3 R = quat_to_rotmat(q[0], q[1], q[2], q[3]);
  M3 M;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) M.m[r * 3 + c] = R.m[r * 3 + c] * s[c];
  M3 vM = mul(vVs, M);
#pragma unroll
  for (int k = 0; k < 9; ++k) vM.m[k] *= 2.f;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    g.scale[c] = (R.m[c] * vM.m[c] + R.m[3 + c] * vM.m[3 + c] + R.m[6 + c] * vM.m[6 + c]) * glob_scale;
  float vR[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) vR[r * 3 + c] = vM.m[r * 3 + c] * s[c];
  const float sn = rsqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float w = q[0] * sn, x = q[1] * sn, y = q[2] * sn, z = q[3] * sn;
#define GOL_VR(r, c) vR[(r) * 3 + (c)]
  g.quat[0] = 2.f * (x * (GOL_VR(2, 1) - GOL_VR(1, 2)) + y * (GOL_VR(0, 2) - GOL_VR(2, 0)) + z * (GOL_VR(1, 0) - GOL_VR(0, 1)));
  g.quat[1] = 2.f * (-2.f * x * (GOL_VR(1, 1) + GOL_VR(2, 2)) + y * (GOL_VR(1, 0) + GOL_VR(0, 1)) + z * (GOL_VR(2, 0) + GOL_VR(0, 2)) +
                     w * (GOL_VR(2, 1) - GOL_VR(1, 2)));
  g.quat[2] = 2.f * (x * (GOL_VR(1, 0) + GOL_VR(0, 1)) - 2.f * y * (GOL_VR(0, 0) + GOL_VR(2, 2)) + z * (GOL_VR(2, 1) + GOL_VR(1, 2)) +
                     w * (GOL_VR(0, 2) - GOL_VR(2, 0)));
  g.quat[3] = 2.f * (x * (GOL_VR(2, 0) + GOL_VR(0, 2)) + y * (GOL_VR(2, 1) + GOL_VR(1, 2)) - 2.f * z * (GOL_VR(0, 0) + GOL_VR(1, 1)) +
                     w * (GOL_VR(1, 0) - GOL_VR(0, 1)));
#undef GOL_VR
  return g;
}

}  // namespace gol_proj
