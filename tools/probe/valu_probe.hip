// valu_probe.hip -- issue cost of the VALU instruction classes the rasterizer is made of, on the GPU at hand.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/valu_probe.hip -o tools/bin/valu_probe ; run on the GPU box.
// Each kernel runs a dependent-free stream of ONE instruction class over 8 independent register sets; the grid fills
// every SIMD with W waves (W = 1, 2, 4, 8).  Reported: cycles per wave-instruction per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
  const float m = 0.999f, c = 1e-6f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (KIND == 1) {  // v_pk_fma_f32 (two floats per lane per instruction)
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
                        : "v"(*(const double*)&b0), "v"(*(const double*)&b2));)
    } else if (KIND == 2) {  // v_exp_f32
      REP8(asm volatile("v_exp_f32 %0, %8\n v_exp_f32 %1, %8\n v_exp_f32 %2, %8\n v_exp_f32 %3, %8\n"
                        "v_exp_f32 %4, %8\n v_exp_f32 %5, %8\n v_exp_f32 %6, %8\n v_exp_f32 %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (KIND == 3) {  // v_cndmask_b32 with an SGPR-pair mask
      REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
    } else if (KIND == 4) {  // v_cmp_gt_f32 into an SGPR pair
      REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
                        "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
    } else if (KIND == 5) {  // v_add_f32 with a DPP row shift
      REP8(asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 6) {  // v_permlane32_swap
      REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                        "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 7) {  // v_mul_f32 (plain)
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                        "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (KIND == 9) {  // v_cmp_lt_u64 (the key compare of the tile sort)
      REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %2\n v_cmp_lt_u64 vcc, %2, %3\n v_cmp_lt_u64 vcc, %3, %0\n"
                        "v_cmp_lt_u64 vcc, %0, %2\n v_cmp_lt_u64 vcc, %1, %3\n v_cmp_lt_u64 vcc, %2, %0\n v_cmp_lt_u64 vcc, %3, %1\n"
                        : : "v"(*(double*)&a0), "v"(*(double*)&a2), "v"(*(double*)&a4), "v"(*(double*)&a6) : "vcc");)
    } else if (KIND == 10) {  // v_cmp_lt_u32
      REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0\n"
                        "v_cmp_lt_u32 vcc, %0, %2\n v_cmp_lt_u32 vcc, %1, %3\n v_cmp_lt_u32 vcc, %2, %0\n v_cmp_lt_u32 vcc, %3, %1\n"
                        : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
    } else if (KIND == 11) {  // ds_bpermute_b32 (the cross-lane exchange of the tile sort)
      REP8(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                        "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
    } else if (KIND == 8) {  // v_rcp_f32
      REP8(asm volatile("v_rcp_f32 %0, %8\n v_rcp_f32 %1, %8\n v_rcp_f32 %2, %8\n v_rcp_f32 %3, %8\n"
                        "v_rcp_f32 %4, %8\n v_rcp_f32 %5, %8\n v_rcp_f32 %6, %8\n v_rcp_f32 %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

template <int KIND>
static void run(const char* name, float* out, int cus, double ghz) {
  const int iters = 2000;              // x 64 instructions per iteration
  for (int w : {1, 2, 4, 8}) {
    const int blocks = cus * w;        // 256 threads = 4 waves = one per SIMD; w blocks per CU -> w waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<blocks, 256>>>(out, 10);
    hipEventRecord(e0);
    probe<KIND><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * 64 * w;
    printf("%-22s waves/SIMD %d: %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, w, ms,
           ms * 1e-3 * ghz * 1e9 / inst_per_simd, ghz);
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  printf("%s: %d CUs, clockRate %.2f GHz\n", p.name, cus, ghz);
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  run<0>("v_fma_f32", out, cus, ghz);
  run<7>("v_mul_f32", out, cus, ghz);
  run<1>("v_pk_fma_f32", out, cus, ghz);
  run<2>("v_exp_f32", out, cus, ghz);
  run<8>("v_rcp_f32", out, cus, ghz);
  run<3>("v_cndmask_b32", out, cus, ghz);
  run<4>("v_cmp_gt_f32", out, cus, ghz);
  run<5>("v_add_f32_dpp", out, cus, ghz);
  run<6>("v_permlane32_swap", out, cus, ghz);
  run<9>("v_cmp_lt_u64", out, cus, ghz);
  run<10>("v_cmp_lt_u32", out, cus, ghz);
  run<11>("ds_bpermute_b32", out, cus, ghz);
  return 0;
}
