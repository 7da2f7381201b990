// imgloss.hip -- masked L1 image loss, forward + backward in one pass each, gfx950.
//
// Replaces the ATen chain of rgb_l1 (/root/reference/ca_code/loss/__init__.py:391-411):
//   ((pred - target) * mask).abs().mean()     sub, mul, abs, mean  (+ sign, mul, mul, div in autograd)
// = 8 passes over the 2048x1334x3 image per view; here the forward reads pred/target(/mask) once and
// the backward reads them once and writes the gradient once ("losses on the image").
// Purely HBM-bound streaming kernels, 16 B per lane.
#include "gol_common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* s_part) {
  v = gol_wave_sum_to_lane63(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) s_part[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_part[w];
  return t;  // valid in thread 0
}

// pred/target [B,C,HW]; mask NULL, [B,1,HW] (mask_c = 1) or [B,C,HW] (mask_c = C).  HW % 4 == 0 fast path.
template <bool BWD>
__global__ __launch_bounds__(256) void l1_kernel(int C, int HW, int mask_c, const float* __restrict__ pred,
                                                 const float* __restrict__ target, const float* __restrict__ mask,
                                                 float* __restrict__ partial, const float* __restrict__ g_loss,
                                                 float inv_n, float* __restrict__ g_pred) {
  __shared__ float s_part[4];
  const int bc = blockIdx.y;          // b * C + c
  const int b = bc / C;
  const size_t base = (size_t)bc * HW;
  const float* mrow = mask ? mask + (size_t)(mask_c == 1 ? b : bc) * HW : nullptr;
  const float scale = BWD ? g_loss[0] * inv_n : 0.f;
  float acc = 0.f;
  for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < HW; i += gridDim.x * 1024) {
    float p[4], t[4], m[4] = {1.f, 1.f, 1.f, 1.f};
    if (i + 3 < HW && (HW & 3) == 0) {
      const float4 pv = *reinterpret_cast<const float4*>(pred + base + i);
      const float4 tv = *reinterpret_cast<const float4*>(target + base + i);
      p[0] = pv.x; p[1] = pv.y; p[2] = pv.z; p[3] = pv.w; t[0] = tv.x; t[1] = tv.y; t[2] = tv.z; t[3] = tv.w;
      if (mrow) { const float4 mv = *reinterpret_cast<const float4*>(mrow + i); m[0] = mv.x; m[1] = mv.y; m[2] = mv.z; m[3] = mv.w; }
      if (BWD) {
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float d = (p[k] - t[k]) * m[k]; g[k] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * m[k] * scale; }
        *reinterpret_cast<float4*>(g_pred + base + i) = make_float4(g[0], g[1], g[2], g[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += fabsf((p[k] - t[k]) * m[k]);
      }
    } else {
      for (int k = 0; k < 4 && i + k < HW; ++k) {
        const float mk = mrow ? mrow[i + k] : 1.f;
        const float d = (pred[base + i + k] - target[base + i + k]) * mk;
        if (BWD) g_pred[base + i + k] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * mk * scale;
        else acc += fabsf(d);
      }
    }
  }
  if (!BWD) {
    const float tsum = block_sum(acc, s_part);
    if (threadIdx.x == 0) partial[(size_t)bc * gridDim.x + blockIdx.x] = tsum;
  }
}

}  // namespace

// partial[B*C*blocks_x] per-workgroup sums (caller reduces: deterministic); blocks_x = gol_l1_blocks(HW)
extern "C" int gol_l1_blocks(int HW) { int b = gol_cdiv(HW, 4096); return b < 1 ? 1 : (b > 64 ? 64 : b); }

extern "C" int gol_l1_fwd(int B, int C, int HW, int mask_c, const float* pred, const float* target, const float* mask,
                          float* partial, void* stream) {
  GOL_REQUIRE(B >= 0 && C > 0 && HW >= 0, "bad sizes");
  if (B == 0 || HW == 0) return GOL_OK;
  GOL_REQUIRE(pred && target && partial, "null pointer");
  GOL_REQUIRE(!mask || mask_c == 1 || mask_c == C, "mask must have 1 or C channels");
  GOL_REQUIRE((long long)B * C <= 65535, "B*C > 65535");
  l1_kernel<false><<<dim3(gol_l1_blocks(HW), B * C), 256, 0, (hipStream_t)stream>>>(C, HW, mask_c, pred, target, mask,
                                                                                  partial, nullptr, 0.f, nullptr);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

// g_loss: device scalar (upstream gradient of the loss); g_pred[B,C,HW] written in full
extern "C" int gol_l1_bwd(int B, int C, int HW, int mask_c, const float* pred, const float* target, const float* mask,
                          const float* g_loss, float* g_pred, void* stream) {
  GOL_REQUIRE(B >= 0 && C > 0 && HW >= 0, "bad sizes");
  if (B == 0 || HW == 0) return GOL_OK;
  GOL_REQUIRE(pred && target && g_loss && g_pred, "null pointer");
  GOL_REQUIRE(!mask || mask_c == 1 || mask_c == C, "mask must have 1 or C channels");
  GOL_REQUIRE((long long)B * C <= 65535, "B*C > 65535");
  const float inv_n = 1.f / ((float)B * (float)C * (float)HW);
  l1_kernel<true><<<dim3(gol_l1_blocks(HW), B * C), 256, 0, (hipStream_t)stream>>>(C, HW, mask_c, pred, target, mask,
                                                                                 nullptr, g_loss, inv_n, g_pred);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
