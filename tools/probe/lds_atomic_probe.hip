// lds_atomic_probe.hip -- cost of ds_add_f32 (no return) as a function of how many lanes of a wave hit the same address.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probe/lds_atomic_probe.hip -o tools/bin/lds_atomic_probe
// Question it answers (MVP backward): is accumulating a wave's template gradients with per-lane LDS atomics
// (64 lanes on ~4-16 distinct voxels) cheaper than the cross-lane reductions (~51 cycles per 4 values)?
// Reported: cycles per wave-instruction per SIMD (4 waves per SIMD resident, 2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int DISTINCT, int STRIDE, typename T, bool RTN>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ T s[4][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 1024; i += 64) s[wave][i] = (T)0;
  __syncthreads();
  // lane -> one of DISTINCT addresses (STRIDE floats apart); 8 independent address sets, one per unrolled instruction
  T* base = &s[wave][(lane % DISTINCT) * STRIDE];
  const T v = (T)(1 + lane);
  T got = (T)0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (RTN) got += atomicAdd(base + ((u * 37) & 63), v);
      else atomicAdd(base + ((u * 37) & 63), v);
    }
  }
  __syncthreads();
  float acc = (float)got;
  for (int i = lane; i < 1024; i += 64) acc += s[wave][i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int DISTINCT, int STRIDE, typename T = float, bool RTN = false>
static void run(const char* name, float* out) {
  const int iters = 2000, blocks = 256 * 4;  // 4 workgroups of 4 waves per CU -> 4 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<DISTINCT, STRIDE, T, RTN><<<blocks, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<DISTINCT, STRIDE, T, RTN><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)blocks * 4 / 1024.0 * iters * 8;  // wave-instructions per SIMD
  printf("%-44s %8.3f ms  %7.1f cycles / wave-instruction / SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  run<64, 1>("ds_add_f32  64 distinct, consecutive", out);
  run<32, 1>("ds_add_f32  32 distinct (2 lanes each)", out);
  run<16, 1>("ds_add_f32  16 distinct (4 lanes each)", out);
  run<8, 1>("ds_add_f32   8 distinct (8 lanes each)", out);
  run<4, 1>("ds_add_f32   4 distinct (16 lanes each)", out);
  run<1, 1>("ds_add_f32   1 address (64 lanes)", out);
  run<16, 4>("ds_add_f32  16 distinct, stride 4 floats", out);
  run<4, 4>("ds_add_f32   4 distinct, stride 4 floats", out);
  run<16, 32>("ds_add_f32  16 distinct, same bank", out);
  run<64, 1, int>("ds_add_u32  64 distinct", out);
  run<16, 1, int>("ds_add_u32  16 distinct (4 lanes each)", out);
  run<1, 1, int>("ds_add_u32   1 address", out);
  run<64, 1, int, true>("ds_add_rtn_u32  64 distinct", out);
  run<16, 1, int, true>("ds_add_rtn_u32  16 distinct (4 lanes each)", out);
  run<1, 1, int, true>("ds_add_rtn_u32   1 address", out);
  run<16, 1, float, true>("ds_add_rtn_f32  16 distinct (4 lanes each)", out);
  hipFree(out);
  return 0;
}
