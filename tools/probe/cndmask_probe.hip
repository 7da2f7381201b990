// cndmask_probe.hip -- what does a select cost?  tools/probe/valu_probe.hip measured v_cndmask_b32 at ~23 cycles per
// wave-instruction; this probe separates the encodings and operand patterns.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/cndmask_probe.hip -o tools/bin/cndmask_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 0.999f, c = 1.5f;
  unsigned long long mask = 0x5555aaaa5555aaaaull ^ (unsigned long long)blockIdx.x;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // VOP2, implicit vcc, dst == src0 (the old probe)
      REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
    } else if (KIND == 1) {  // VOP3 with an ordinary SGPR pair as the mask
      REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                        "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(mask));)
    } else if (KIND == 2) {  // compare + select pairs (the pattern compiled code has): 4 cmp + 4 cndmask per group
      REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                        "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(m) : "vcc");)
    } else if (KIND == 3) {  // the arithmetic alternative: multiply by a 0/1 float
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                        "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (KIND == 4) {  // select with constant 0 operand (v_cndmask dst, 0, src, mask)
      REP8(asm volatile("v_cndmask_b32_e64 %0, 0, %0, %8\n v_cndmask_b32_e64 %1, 0, %1, %8\n v_cndmask_b32_e64 %2, 0, %2, %8\n v_cndmask_b32_e64 %3, 0, %3, %8\n"
                        "v_cndmask_b32_e64 %4, 0, %4, %8\n v_cndmask_b32_e64 %5, 0, %5, %8\n v_cndmask_b32_e64 %6, 0, %6, %8\n v_cndmask_b32_e64 %7, 0, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(mask));)
    } else if (KIND == 5) {  // v_and_b32 with a per-lane 0 / ~0 mask register (bitwise select-to-zero)
      REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                        "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
static void run(const char* name, float* out) {
  const int iters = 2000, w = 4, blocks = 256 * w;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<KIND><<<blocks, 256>>>(out, 10);
  (void)hipEventRecord(e0);
  probe<KIND><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %8.3f ms  -> %6.2f cycles per wave-instruction per SIMD (4 waves/SIMD, 2.4 GHz)\n", name, ms,
         ms * 1e-3 * 2.4e9 / ((double)iters * 64 * w));
}

int main() {
  float* out;
  (void)hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  run<0>("v_cndmask_b32 (VOP2, vcc), dst == src0", out);
  run<1>("v_cndmask_b32_e64, SGPR-pair mask", out);
  run<2>("v_cmp_gt_f32 + v_cndmask_b32 pairs (per instruction)", out);
  run<3>("v_mul_f32", out);
  run<4>("v_cndmask_b32_e64 dst, 0, src, SGPR-pair mask", out);
  run<5>("v_and_b32", out);
  (void)hipFree(out);
  return 0;
}
