// abi.hip -- library identification and error reporting for libgoliath_hip.so.
#include <cstdarg>
#include <cstdio>

#include "gol_common.h"

static thread_local char g_err[512] = "";

void gol_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gol_version(void) { return "goliath_hip 0.1.0 gfx950"; }
extern "C" const char* gol_last_error(void) { return g_err; }

// ---- self-test hook for the wave64 cross-lane primitives (tests/test_gpu_primitives.py) ----------
__global__ void selftest_wave_sum4_kernel(const float* __restrict__ in, float* __restrict__ out) {
  const int l = threadIdx.x;
  const float r = gol_wave_sum4(in[l], in[64 + l], in[128 + l], in[192 + l]);
  out[l] = r;
  out[64 + l] = gol_wave_sum_to_lane63(in[l]);
}
extern "C" int gol_selftest_wave_sum4(const float* in256, float* out128, void* stream) {
  selftest_wave_sum4_kernel<<<1, 64, 0, (hipStream_t)stream>>>(in256, out128);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
