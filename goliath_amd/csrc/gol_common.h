// gol_common.h -- device helpers shared by the gfx950 kernels of libgoliath_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
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

// v + (v shifted right by 8 lanes inside its row of 16, 0 shifted in) as ONE v_add_f32_dpp.  Written as assembly: when
// only lane 15 of a row consumes the result the compiler sinks the add into the consumer's branch and leaves a
// v_mov 0 + v_mov_dpp + v_add triple behind (the DPP combiner works inside one basic block).  The s_nop covers the
// two wait states a DPP read needs after a VALU write of its source (the hazard recognizer does not look into asm).
__device__ __forceinline__ float gol_add_row_shr8(float v) {
  float r;
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
  return r;
}
// rows 1 and 3 add lane 15 of the row before them / rows 2 and 3 add lane 31 (the last two steps of a 64-lane sum):
// one v_add_f32_dpp each (masked-off rows keep their value) instead of v_mov 0 + v_mov_dpp + v_add
__device__ __forceinline__ float gol_add_row_bcast15(float v) {
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
  return v;
}
__device__ __forceinline__ float gol_add_row_bcast31(float v) {
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
  return v;
}

// sums of the four 16-lane rows, valid in lanes 15, 31, 47, 63 (4 fused v_add_f32_dpp)
__device__ __forceinline__ float gol_row_sum_to_lane15(float v) {
  v += gol_dpp_mov0<0x111>(v);  // row_shr:1
  v += gol_dpp_mov0<0x112>(v);  // row_shr:2
  v += gol_dpp_mov0<0x114>(v);  // row_shr:4
  return gol_add_row_shr8(v);   // row_shr:8
}

// Four wave-wide sums for the price of ~1.5: gfx950's v_permlane32_swap / v_permlane16_swap fold
// two registers into one per step (upper half of x <-> lower half of y, then odd rows <-> even
// rows), so 64-lane sums of (a, b, c, d) cost 3 swaps + 3 adds + 4 DPP row adds = 10 instructions
// instead of 4 x 6.  Results: lane 15 = sum(a), lane 31 = sum(b), lane 47 = sum(c), lane 63 = sum(d).
__device__ __forceinline__ float gol_swap32_sum(float x, float y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y),
                                                  false, false);
  const unsigned r0 = r[0], r1 = r[1];  // (bit_cast straight from a vector element miscompiles)
  return __uint_as_float(r0) + __uint_as_float(r1);  // [x_lo+x_hi | y_lo+y_hi]
}
__device__ __forceinline__ float gol_swap16_sum(float x, float y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y),
                                                  false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __uint_as_float(r0) + __uint_as_float(r1);  // rows [x01, y01, x23, y23]
}
__device__ __forceinline__ float gol_wave_sum4(float a, float b, float c, float d) {
  const float ac = gol_swap32_sum(a, c);   // lower 32 lanes: a, upper 32: c
  const float bd = gol_swap32_sum(b, d);   // lower: b, upper: d
  const float q = gol_swap16_sum(ac, bd);  // rows: a, b, c, d
  return gol_row_sum_to_lane15(q);
}

// element at a 32-bit BYTE offset from a (wave-uniform) base pointer: the SGPR-base + VGPR-offset addressing form
// (64-bit per-lane address arithmetic costs several quarter-rate vector instructions per access)
template <typename T>
__device__ __forceinline__ T* gol_at(T* base, unsigned byte_off) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(base)) + byte_off);
}

// lane mask of a predicate as a scalar (s_and with exec); HIP's __ballot(int) goes through an int (v_cndmask + v_cmp_ne)
__device__ __forceinline__ unsigned long long gol_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// returns the wave-wide sum in lane 63 (other lanes hold partial sums)
__device__ __forceinline__ float gol_wave_sum_to_lane63(float v) {
  v += gol_dpp_mov0<0x111>(v);              // row_shr:1
  v += gol_dpp_mov0<0x112>(v);              // row_shr:2
  v += gol_dpp_mov0<0x114>(v);              // row_shr:4  (lanes 4..15 of a row now hold 8-sums..)
  v = gol_add_row_shr8(v);                  // row_shr:8  -> lane 15 of each row = row sum
  v = gol_add_row_bcast15(v);               // row_bcast:15 into rows 1 and 3
  return gol_add_row_bcast31(v);            // row_bcast:31 into rows 2 and 3
}

__device__ __forceinline__ float gol_readlane63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- alpha >= 1/255 reach test (shared by binning and raster) -----------------------------------
// A Gaussian (centre g, conic a,b,c, opacity op) contributes at pixel p only if
//   op * exp(-sigma(p)) >= 1/255  <=>  sigma(p) = 0.5 d^T C d <= ln(255 op) =: tau     
// gol_alpha_tau returns a slightly inflated tau (conservative against rounding; < 0 means "never").
__device__ __forceinline__ float gol_alpha_tau(float op) {
  _Pragma("clang fp contract(on)")
  const float k = 255.f * op;
  return (k > 1.f) ? __logf(k) * 1.001f + 1e-3f : -1.f;
}
// minimum of sigma over the axis-aligned rectangle [x0,x1] x [y0,y1] (exact: the quadratic is convex,
// so the minimum is 0 inside or lies on one of the four edges)
// (ia = 1 / a and ic = 1 / c are per-Gaussian: callers testing many rectangles pass them in)
__device__ __forceinline__ float gol_min_sigma_rect(float gx, float gy, float a, float b, float c, float ia, float ic,
                                                    float x0, float x1, float y0, float y1) {
  if (gx >= x0 && gx <= x1 && gy >= y0 && gy <= y1) return 0.f;
  float m = 3.0e38f;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float dx = (e ? x1 : x0) - gx;
    const float dy = fminf(fmaxf(-b * dx * ic, y0 - gy), y1 - gy);
    m = fminf(m, 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy);
    const float ey = (e ? y1 : y0) - gy;
    const float ex = fminf(fmaxf(-b * ey * ia, x0 - gx), x1 - gx);
    m = fminf(m, 0.5f * (a * ex * ex + c * ey * ey) + b * ex * ey);
  }
  return m;
}

// ---- packed per-Gaussian raster record (GOL_SPLAT_RECORD floats = 64 bytes = one HBM sector) ------------------------
// Everything the rasterizer stages for a list entry, written once per (view, Gaussian) by the projection (or by
// gol_splat_pack for the gsplat-compatible operators) and fetched with four 16-byte loads from ONE aligned 64-byte line
// -- the five separate attribute arrays of rounds 1-2 cost up to five sectors per entry (PMC: 109 B fetched for 40 B
// used).  The conic is stored the way the pixel loops want it: pre-multiplied by log2(e) (alpha = opacity * 2^-sigma' is
// one v_exp_f32) and the diagonal terms by the 1/2 of sigma = (a dx^2 + c dy^2) / 2 + b dx dy as well; GOL_UN_A / GOL_UN_B
// bring the true conic back.  The exact-math test build stores it unscaled.
//   [0] x  [1] y  [2] a'  [3] b' | [4] c'  [5] opacity  [6] r  [7] g | [8] b  [9] extra (depth)  [10] tau  [11] 1/a |
//   [12] 1/c  [13] exact (1 = the ellipse tests are valid, 0 = degenerate conic: "reaches everything")  [14-15] pad
// tau = gol_alpha_tau(opacity) (< 0: alpha < 1/255 everywhere), 1/a and 1/c of the TRUE conic: the per-entry part of the
// alpha >= 1/255 reach test, hoisted out of the staging loops of both raster passes.
#ifndef GOL_EXACT_MATH
#define GOL_SC_A (0.5f * 1.4426950408889634f)
#define GOL_SC_B 1.4426950408889634f
#define GOL_UN_A (2.f * 0.6931471805599453f)
#define GOL_UN_B 0.6931471805599453f
#else
#define GOL_SC_A 1.f
#define GOL_SC_B 1.f
#define GOL_UN_A 1.f
#define GOL_UN_B 1.f
#endif

__device__ __forceinline__ void gol_record_write(float* __restrict__ rec, float x, float y, float ca, float cb, float cc,
                                                 float op, float r, float g, float b, float extra) {
  _Pragma("clang fp contract(on)")   // written from two kernels (projection, shading epilogue): same bits from both
  const bool exact = (ca * cc - cb * cb > 0.f) && ca > 0.f && cc > 0.f;
  float4* R = reinterpret_cast<float4*>(rec);
  R[0] = make_float4(x, y, ca * GOL_SC_A, cb * GOL_SC_B);
  R[1] = make_float4(cc * GOL_SC_A, op, r, g);
  R[2] = make_float4(b, extra, gol_alpha_tau(op), exact ? 1.f / ca : 0.f);
  R[3] = make_float4(exact ? 1.f / cc : 0.f, exact ? 1.f : 0.f, 0.f, 0.f);
}

__device__ __forceinline__ float gol_fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float gol_rcp(float x) { return __frcp_rn(x); }
