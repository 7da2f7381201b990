// gol_common.h -- device helpers shared by the gfx950 kernels of libgoliath_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/goliath_hip.h"

// ---- gsplat 0.1.11 constants; one place, named ------------------
#define GOL_BLUR 0.3f
#define GOL_FOV_CLAMP 1.3f
#define GOL_EIG_FLOOR 0.1f
#define GOL_RADIUS_SIGMAS 3.0f
#define GOL_Z_EPS 1e-6f
#define GOL_ALPHA_CAP_FWD 0.999f
#define GOL_ALPHA_CAP_BWD 0.99f /* upstream backward.cu clamps at 0.99, forward at 0.999 */
#define GOL_ALPHA_FLOOR (1.f / 255.f)
#define GOL_T_STOP 1e-4f
#define GOL_COMP_EPS 1e-6f

#define GOL_WAVE 64

void gol_set_error(const char* fmt, ...);

#define GOL_REQUIRE(cond, msg)                                         \
  do {                                                                 \
    if (!(cond)) {                                                     \
      gol_set_error("%s: %s", __func__, msg);                          \
      return GOL_ERR_INVALID_ARG;                                      \
    }                                                                  \
  } while (0)

#define GOL_CHECK_LAUNCH()                                             \
  do {                                                                 \
    hipError_t e_ = hipGetLastError();                                 \
    if (e_ != hipSuccess) {                                            \
      gol_set_error("%s: %s", __func__, hipGetErrorString(e_));        \
      return GOL_ERR_LAUNCH;                                           \
    }                                                                  \
  } while (0)

static inline int gol_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- wave64 reductions through DPP (no LDS traffic) ---------------------------------------
// Sum over the 64 lanes; the total is valid in lane 63 (row_shr 1,2,3 + row_bcast 15/31,
// the classic GCN/CDNA reduction ladder; bound_ctrl=0 makes out-of-row sources read 0).
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float gol_dpp_mov0(float v) {
  // old = 0 with bound_ctrl so disabled / out-of-range lanes contribute 0
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false));
}

// returns the wave-wide sum in lane 63 (other lanes hold partial sums)
__device__ __forceinline__ float gol_wave_sum_to_lane63(float v) {
  v += gol_dpp_mov0<0x111>(v);              // row_shr:1
  v += gol_dpp_mov0<0x112>(v);              // row_shr:2
  v += gol_dpp_mov0<0x114>(v);              // row_shr:4  (lanes 4..15 of a row now hold 8-sums..)
  v += gol_dpp_mov0<0x118>(v);              // row_shr:8  -> lane 15 of each row = row sum
  v += gol_dpp_mov0<0x142, 0xa>(v);         // row_bcast:15 into rows 1 and 3
  v += gol_dpp_mov0<0x143, 0xc>(v);         // row_bcast:31 into rows 2 and 3
  return v;
}

__device__ __forceinline__ float gol_readlane63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float gol_fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float gol_rcp(float x) { return __frcp_rn(x); }
