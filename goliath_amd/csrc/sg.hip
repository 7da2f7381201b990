// sg.hip -- spherical-Gaussian specular lobe evaluation for gfx950.
//
// Replaces evaluate_gaussian_{fwd,bwd}_kernel (/root/reference/extensions/sgutils/sg.cu:27-76,
// 78-175).  One lane per (view n, Gaussian d); the per-view light list is wave-uniform, so the
// light loop runs on scalar (SGPR) loads while the per-Gaussian data streams through coalesced
// 12-byte-per-lane vector loads.  HBM-bound for small n_lights (40 B/Gaussian fwd, 56 B bwd),
// VALU-bound (acos + exp + rsqrt per light) beyond ~8 lights.
#include "gol_common.h"

namespace {

constexpr float kTwoPi = 6.28318530718f;        // sg.cu:18
constexpr float kInv2Pi = 0.15915494309f;       // sg.cu:19
constexpr float kSqrt2Pi23 = 3.03352966508f;    // sg.cu:20
constexpr float kInvSqrt2Pi23 = 0.32964899322f; // sg.cu:21

__device__ __forceinline__ float sqr(float v) { return v * v; }

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 ld3(const float* p, size_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

template <int W_TYPE>
__global__ __launch_bounds__(256) void sg_fwd_kernel(
    const float* __restrict__ lobe_dirs, const float* __restrict__ lobe_sigmas,
    const float* __restrict__ light_values, const float* __restrict__ light_pts,
    const float* __restrict__ prim_pts, const int32_t* __restrict__ n_lights,
    float* __restrict__ integral, int D, int L) {
  const int n = blockIdx.y;
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  const size_t e = (size_t)n * D + d;
  const f3 dir = ld3(lobe_dirs, e);
  const f3 pp = ld3(prim_pts, e);
  const float sigma = lobe_sigmas[e];
  const float inv_sigma = 1.f / sigma;
  // lobe normalisation hoisted out of the light loop (one exact division per Gaussian instead of one per light)
  const float norm = W_TYPE == 0 ? 1.f / (sigma * kSqrt2Pi23) : (W_TYPE == 2 ? 1.f / (sigma * kTwoPi) : 1.f);
  const int nL = n_lights[n];
  const float* lv = light_values + (size_t)n * L * 3;
  const float* lp = light_pts + (size_t)n * L * 3;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int l = 0; l < nL; ++l) {
    const float lx = lp[3 * l] - pp.x, ly = lp[3 * l + 1] - pp.y, lz = lp[3 * l + 2] - pp.z;
    const float rn = rsqrtf(lx * lx + ly * ly + lz * lz);
    const float c = fminf(1.f, fmaxf(-1.f, (lx * dir.x + ly * dir.y + lz * dir.z) * rn));
    float w;
    if (W_TYPE == 0) {
      w = __expf(-0.5f * sqr(acosf(c) * inv_sigma)) * norm;
    } else if (W_TYPE == 1) {
      w = __expf(-0.5f * sqr(acosf(c) * inv_sigma));
    } else if (W_TYPE == 2) {
      w = __expf((c - 1.f) * inv_sigma) * norm;
    } else {
      w = __expf((c - 1.f) * inv_sigma);
    }
    sx += lv[3 * l] * w; sy += lv[3 * l + 1] * w; sz += lv[3 * l + 2] * w;
  }
  integral[3 * e] = sx; integral[3 * e + 1] = sy; integral[3 * e + 2] = sz;
}

template <int W_TYPE, bool LIGHT_GRAD>
__global__ __launch_bounds__(256) void sg_bwd_kernel(
    const float* __restrict__ lobe_dirs, const float* __restrict__ lobe_sigmas,
    const float* __restrict__ light_values, const float* __restrict__ light_pts,
    const float* __restrict__ prim_pts, const int32_t* __restrict__ n_lights,
    const float* __restrict__ grad_integral, float* __restrict__ grad_dirs,
    float* __restrict__ grad_sigmas, float* __restrict__ grad_light_values, int D, int L) {
  const int n = blockIdx.y;
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = d < D;
  const size_t e = (size_t)n * D + (live ? d : 0);
  const f3 gi = live ? ld3(grad_integral, e) : f3{0.f, 0.f, 0.f};
  const f3 dir = ld3(lobe_dirs, e);
  const f3 pp = ld3(prim_pts, e);
  const float sigma = lobe_sigmas[e];
  const float s2 = sigma * sigma;
  // per-Gaussian reciprocals: the light loop multiplies (five exact divisions per light otherwise)
  const float inv_sigma = 1.f / sigma, inv_s2 = 1.f / s2, inv_s3 = 1.f / (s2 * sigma), inv_s4 = 1.f / (s2 * s2);
  const float norm0 = 1.f / (sigma * kSqrt2Pi23), norm2 = 1.f / (sigma * kTwoPi);
  const int nL = n_lights[n];
  const float* lv = light_values + (size_t)n * L * 3;
  const float* lp = light_pts + (size_t)n * L * 3;
  float gx = 0.f, gy = 0.f, gz = 0.f, gs = 0.f;
  for (int l = 0; l < nL; ++l) {
    float lx = lp[3 * l] - pp.x, ly = lp[3 * l + 1] - pp.y, lz = lp[3 * l + 2] - pp.z;
    const float rn = rsqrtf(lx * lx + ly * ly + lz * lz);
    lx *= rn; ly *= rn; lz *= rn;
    const float c = lx * dir.x + ly * dir.y + lz * dir.z;
    const float cc = fminf(1.f, fmaxf(-1.f, c));
    const float ex0 = lv[3 * l], ex1 = lv[3 * l + 1], ex2 = lv[3 * l + 2];
    const float dw = gi.x * ex0 + gi.y * ex1 + gi.z * ex2;
    float weight, dc;
    if (W_TYPE == 0 || W_TYPE == 1) {
      const float angle = acosf(cc);
      const float ex = __expf(-0.5f * sqr(angle * inv_sigma));
      // sg.cu:129,139: d acos/dc = -1/sqrt(1-c^2) inside (-1,1), the constant -20 outside
      const float dacos = (c > -1.f && c < 1.f) ? -rsqrtf(1.f - c * c) : -20.f;
      if (W_TYPE == 0) {
        weight = ex * norm0;
        gs += dw * ((ex * kInvSqrt2Pi23 * (sqr(angle) - s2)) * inv_s4);
        dc = dw * -((kInvSqrt2Pi23 * angle * ex) * inv_s3) * dacos;
      } else {
        weight = ex;
        gs += dw * ((ex * sqr(angle)) * inv_s3);
        dc = dw * -((angle * ex) * inv_s2) * dacos;
      }
    } else {
      const float ex = __expf((cc - 1.f) * inv_sigma);
      if (W_TYPE == 2) {
        weight = ex * norm2;
        gs += dw * ((ex * kInv2Pi * ((1.f - cc) - sigma)) * inv_s3);
        dc = dw * kInv2Pi * ex * inv_s2;
      } else {
        weight = ex;
        gs += dw * ((ex * (1.f - cc) * inv_s2));
        dc = dw * ex * inv_sigma;
      }
    }
    gx += dc * lx; gy += dc * ly; gz += dc * lz;
    if (LIGHT_GRAD) {
      // one atomic per wave per light component instead of one per lane (sg.cu:165-169)
      float a0 = gol_wave_sum_to_lane63(live ? gi.x * weight : 0.f);
      float a1 = gol_wave_sum_to_lane63(live ? gi.y * weight : 0.f);
      float a2 = gol_wave_sum_to_lane63(live ? gi.z * weight : 0.f);
      if ((threadIdx.x & 63) == 63) {
        float* g = grad_light_values + ((size_t)n * L + l) * 3;
        atomicAdd(g + 0, a0); atomicAdd(g + 1, a1); atomicAdd(g + 2, a2);
      }
    }
  }
  if (live) {
    grad_sigmas[e] = gs;
    grad_dirs[3 * e] = gx; grad_dirs[3 * e + 1] = gy; grad_dirs[3 * e + 2] = gz;
  }
}

}  // namespace

extern "C" int gol_sg_eval_fwd(int N, int D, int L, const float* lobe_dirs, const float* lobe_sigmas,
                               const float* light_values, const float* light_pts, const float* prim_pts,
                               const int32_t* n_lights, float* integral, int w_type, void* stream) {
  GOL_REQUIRE(N >= 0 && D >= 0 && L >= 0, "negative size");
  GOL_REQUIRE(w_type >= 0 && w_type <= 3, "w_type must be 0..3");
  if (N == 0 || D == 0) return GOL_OK;
  GOL_REQUIRE(lobe_dirs && lobe_sigmas && prim_pts && n_lights && integral, "null pointer");
  GOL_REQUIRE(L == 0 || (light_values && light_pts), "null light pointer");
  GOL_REQUIRE(N <= 65535, "N (views) > 65535");
  dim3 grid(gol_cdiv(D, 256), N), block(256);
  hipStream_t s = (hipStream_t)stream;
#define GOL_SG_FWD(W) sg_fwd_kernel<W><<<grid, block, 0, s>>>(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, integral, D, L)
  switch (w_type) {
    case 0: GOL_SG_FWD(0); break;
    case 1: GOL_SG_FWD(1); break;
    case 2: GOL_SG_FWD(2); break;
    default: GOL_SG_FWD(3); break;
  }
#undef GOL_SG_FWD
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_sg_eval_bwd(int N, int D, int L, const float* lobe_dirs, const float* lobe_sigmas,
                               const float* light_values, const float* light_pts, const float* prim_pts,
                               const int32_t* n_lights, const float* grad_integral, float* grad_dirs,
                               float* grad_sigmas, float* grad_light_values, int w_type, void* stream) {
  GOL_REQUIRE(N >= 0 && D >= 0 && L >= 0, "negative size");
  GOL_REQUIRE(w_type >= 0 && w_type <= 3, "w_type must be 0..3");
  if (N == 0 || D == 0) return GOL_OK;
  GOL_REQUIRE(lobe_dirs && lobe_sigmas && prim_pts && n_lights && grad_integral && grad_dirs && grad_sigmas,
              "null pointer");
  GOL_REQUIRE(L == 0 || (light_values && light_pts), "null light pointer");
  GOL_REQUIRE(N <= 65535, "N (views) > 65535");
  dim3 grid(gol_cdiv(D, 256), N), block(256);
  hipStream_t s = (hipStream_t)stream;
#define GOL_SG_BWD(W, G) sg_bwd_kernel<W, G><<<grid, block, 0, s>>>(lobe_dirs, lobe_sigmas, light_values, light_pts, prim_pts, n_lights, grad_integral, grad_dirs, grad_sigmas, grad_light_values, D, L)
  const bool lg = grad_light_values != nullptr;
  switch (w_type) {
    case 0: if (lg) GOL_SG_BWD(0, true); else GOL_SG_BWD(0, false); break;
    case 1: if (lg) GOL_SG_BWD(1, true); else GOL_SG_BWD(1, false); break;
    case 2: if (lg) GOL_SG_BWD(2, true); else GOL_SG_BWD(2, false); break;
    default: if (lg) GOL_SG_BWD(3, true); else GOL_SG_BWD(3, false); break;
  }
#undef GOL_SG_BWD
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
