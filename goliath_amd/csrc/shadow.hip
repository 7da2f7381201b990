// shadow.hip -- URHand shadow-map lookup with 3x3 PCF, gfx950.
//
// Replaces the per-texel part of get_shadow_map (/root/reference/ca_code/utils/shadowmap.py:30-96): project every
// uv texel into each light's depth camera (project_points_multi, ca_code/utils/geom.py:599-631), compare its depth
// with a 3x3 Gaussian-weighted neighbourhood of nearest-sampled depth-map values (18 F.grid_sample calls on
// [B*L,1,h,w] tensors in the reference), blend with the soft back-face term, and -- optionally -- apply the
// exp(-x/8) of the caller (ca_code/models/urhand.py:416,504).  The depth image itself comes from the mesh
// rasteriser (drtk in the reference; third-party, not part of this library).
//   one lane = one texel; the L lights of a batch element are the inner loop, so positions / normals are read once
//   (the reference materialises them L times: [B*L,3,S,S]); light matrices are wave-uniform scalar loads.
#include "gol_common.h"

namespace {

struct ShadowDims {
  int B, L, HW, dh, dw;
  float fx, fy, cx, cy, exp_scale, inv_exp_scale;
  float wgt[9];  // exp(-((x-1)^2 + (y-1)^2) / (2 sigma^2)), index 3*x + y (shadowmap.py:73-77)
};

// F.grid_sample(mode="nearest", align_corners=False, padding_mode="zeros") of one coordinate
__device__ __forceinline__ int nearest_index(float g, int size) {
  const float u = ((g + 1.f) * (float)size - 1.f) / 2.f;  // grid_sampler_unnormalize
  return (int)nearbyintf(u);                               // round half to even, like std::nearbyint
}

__global__ __launch_bounds__(256) void shadow_pcf_kernel(const ShadowDims p, const float* __restrict__ depth,
                                                         const float* __restrict__ Rt, const float* __restrict__ postex,
                                                         const float* __restrict__ nml, float* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (n >= p.HW) return;
  const float px = postex[((size_t)b * 3 + 0) * p.HW + n], py = postex[((size_t)b * 3 + 1) * p.HW + n],
              pz = postex[((size_t)b * 3 + 2) * p.HW + n];
  float nx = 0.f, ny = 0.f, nz = 0.f;
  if (nml) {
    nx = nml[((size_t)b * 3 + 0) * p.HW + n]; ny = nml[((size_t)b * 3 + 1) * p.HW + n];
    nz = nml[((size_t)b * 3 + 2) * p.HW + n];
  }
  const float dx = 2.0f / (float)p.dw, dy = 2.0f / (float)p.dh;
  for (int l = 0; l < p.L; ++l) {
    const size_t bl = (size_t)b * p.L + l;
    const float* M = Rt + bl * 12;  // wave-uniform
    const float X = M[0] * px + M[1] * py + M[2] * pz + M[3];
    const float Y = M[4] * px + M[5] * py + M[6] * pz + M[7];
    const float Z = M[8] * px + M[9] * py + M[10] * pz + M[11];
    // p_pix = p_cam @ K^T, then / depth  (geom.py:619-622)
    const float iz = 1.f / Z;  // one exact division per (texel, light); everything below multiplies
    const float u = (p.fx * X + p.cx * Z) * iz, v = (p.fy * Y + p.cy * Z) * iz;
    const float gx = (u - (float)p.dw / 2.0f - 0.5f) / ((float)p.dw / 2.0f);  // shadowmap.py:55-56
    const float gy = (v - (float)p.dh / 2.0f - 0.5f) / ((float)p.dh / 2.0f);
    const float* D = depth + bl * (size_t)p.dh * p.dw;
    float vsum = 0.f, ssum = 0.f;
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 3; ++y) {
        const int ix = nearest_index(gx + dx * (float)(x - 1), p.dw), iy = nearest_index(gy + dy * (float)(y - 1), p.dh);
        const bool in = ix >= 0 && ix < p.dw && iy >= 0 && iy < p.dh;
        const float d = D[(size_t)min(max(iy, 0), p.dh - 1) * p.dw + min(max(ix, 0), p.dw - 1)];
        const float dd = in ? d : 0.f;
        // w = sample of (depth > 0).float() is 0 or 1: d / (w + 1e-8) is d itself when w = 1 (1 + 1e-8 == 1 in fp32) and
        // is multiplied by valid = 0 when w = 0, so the division of shadowmap.py:82 drops out exactly
        const float valid = dd > 0.f ? p.wgt[3 * x + y] : 0.f;
        vsum += valid;
        ssum += valid * fmaxf(Z - dd, 0.f);
      }
    float sh = ssum / (vsum + 1e-6f);
    if (nml) {
      // v_dir = normalize(Rt[:, :, -1] - postex); bcull = sigmoid(10 n.v)  (shadowmap.py:58-62)
      const float vx = M[3] - px, vy = M[7] - py, vz = M[11] - pz;
      const float inv = 1.f / fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
      const float nv = (nx * vx + ny * vy + nz * vz) * inv;
      const float bc = 1.f / (1.f + expf(-10.f * nv));
      sh = bc * sh + (1.f - bc) * 1e3f;
    }
    out[bl * p.HW + n] = p.exp_scale > 0.f ? expf(-sh * p.inv_exp_scale) : sh;
  }
}

}  // namespace

extern "C" int gol_shadow_pcf(int B, int L, int H, int W, int dh, int dw, const float* depth, const float* Rt, float fx,
                              float fy, float cx, float cy, const float* postex, const float* nml, float exp_scale,
                              float* out, void* stream) {
  GOL_REQUIRE(B >= 0 && L >= 0 && H > 0 && W > 0 && dh > 0 && dw > 0, "bad sizes");
  if (B == 0 || L == 0) return GOL_OK;
  GOL_REQUIRE(depth && Rt && postex && out, "null pointer");
  GOL_REQUIRE(B <= 65535, "B > 65535");
  ShadowDims p;
  p.B = B; p.L = L; p.HW = H * W; p.dh = dh; p.dw = dw;
  p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.exp_scale = exp_scale;
  p.inv_exp_scale = exp_scale > 0.f ? 1.f / exp_scale : 0.f;
  const double sigma = 0.3 * ((3 - 1) * 0.5 - 1) + 0.8;  // shadowmap.py:66
  for (int x = 0; x < 3; ++x)
    for (int y = 0; y < 3; ++y)
      p.wgt[3 * x + y] = (float)exp(-((x - 1) * (x - 1) + (y - 1) * (y - 1)) / (2.0 * sigma * sigma));
  shadow_pcf_kernel<<<dim3(gol_cdiv(p.HW, 256), B), 256, 0, (hipStream_t)stream>>>(p, depth, Rt, postex, nml, out);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
