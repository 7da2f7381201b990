// mfma_valu_probe.hip -- do fp32 MFMAs (v_mfma_f32_16x16x4_f32) and plain fp32 VALU work overlap on a SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_probe.hip -o tools/bin/mfma_valu_probe
// Each kernel runs, per loop iteration, M independent-accumulator MFMAs and V v_fma_f32, either from the SAME wave
// (interleaved) or split between waves of a SIMD (even waves MFMA, odd waves VALU).  Reported: ns per iteration, so
// t(M, V) can be compared with t(M, 0) + t(0, V) (no overlap) and max of the two (full overlap).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M, int V, bool SPLIT>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = threadIdx.x * 1e-3f, b = 1.f - a;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  const bool do_m = !SPLIT || ((blockIdx.x >> 8) & 1) == 0, do_v = !SPLIT || ((blockIdx.x >> 8) & 1) == 1;
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int k = 0; k < V; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(a), "v"(b));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  out[blockIdx.x * 256 + threadIdx.x] = s + wave;
}

template <int M, int V, bool SPLIT>
static float run(const char* name, float* out) {
  // SPLIT: twice the workgroups (half of them MFMA-only, half VALU-only), so every SIMD hosts both kinds
  const int iters = 4000, blocks = 256 * (SPLIT ? 4 : 2);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<M, V, SPLIT><<<blocks, 256>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<M, V, SPLIT><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %8.3f ms  %7.1f ns / iteration\n", name, ms, ms * 1e6 / iters);
  return ms;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  run<8, 0, false>("8 MFMA 16x16x4 f32 per iteration, 2 waves/SIMD", out);
  run<0, 128, false>("128 v_fma_f32 per iteration, 2 waves/SIMD", out);
  run<8, 128, false>("8 MFMA + 128 v_fma in the SAME wave, 2 waves/SIMD", out);
  run<8, 128, true>("8 MFMA waves + 128 v_fma waves (2 + 2 waves/SIMD)", out);
  run<8, 0, true>("split layout, MFMA waves only work (V waves idle)", out);
  run<0, 128, true>("split layout, VALU waves only work (M waves idle)", out);
  (void)hipFree(out);
  return 0;
}
