// lds_atomic_pattern_probe.hip -- cost of ds_add_u32 (no return) for the address patterns of the MVP backward's fixed-point
// scatter table: how do same-word and same-bank collisions price a wave-instruction?
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_atomic_pattern_probe.hip -o /tmp/lds_pat
// Reported: CU-cycles per wave-instruction (16 waves per CU resident, all issuing; 2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void probe(const int* __restrict__ pat, int nact, float* out, int iters) {
  __shared__ int s[4][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 1024; i += 64) s[wave][i] = 0;
  __syncthreads();
  int a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = pat[u * 64 + lane];
  if (lane < nact) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) __hip_atomic_fetch_add(&s[wave][a[u]], lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  __syncthreads();
  float acc = 0.f;
  for (int i = lane; i < 1024; i += 64) acc += (float)s[wave][i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void probe64(const int* __restrict__ pat, int nact, float* out, int iters) {
  __shared__ unsigned long long s[4][512];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 512; i += 64) s[wave][i] = 0;
  __syncthreads();
  int a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = pat[u * 64 + lane];
  if (lane < nact) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        __hip_atomic_fetch_add(&s[wave][a[u]], (unsigned long long)(lane + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  __syncthreads();
  float acc = 0.f;
  for (int i = lane; i < 512; i += 64) acc += (float)s[wave][i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static void run64(const char* name, const int* h, int nact, int* dpat, float* out) {
  hipMemcpy(dpat, h, 512 * sizeof(int), hipMemcpyHostToDevice);
  const int iters = 2000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe64<<<blocks, 256>>>(dpat, nact, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe64<<<blocks, 256>>>(dpat, nact, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_cu = (double)blocks * 4 / 256.0 * iters * 8;
  printf("u64  %-59s %8.3f ms  %7.1f CU-cycles / wave-instruction\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_cu);
}

static void run(const char* name, const int* h, int nact, int* dpat, float* out) {
  hipMemcpy(dpat, h, 512 * sizeof(int), hipMemcpyHostToDevice);
  const int iters = 2000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<<<blocks, 256>>>(dpat, nact, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<<<blocks, 256>>>(dpat, nact, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_cu = (double)blocks * 4 / 256.0 * iters * 8;
  printf("%-64s %8.3f ms  %7.1f CU-cycles / wave-instruction\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_cu);
}

int main() {
  int* dpat; float* out;
  hipMalloc(&dpat, 512 * sizeof(int));
  hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  int h[512];
  auto fill = [&](auto f) { for (int u = 0; u < 8; ++u) for (int l = 0; l < 64; ++l) h[u * 64 + l] = f(u, l); };
  fill([](int u, int l) { return (l + u * 37) & 63; });                  run("64 lanes, 64 distinct consecutive words", h, 64, dpat, out);
  fill([](int u, int l) { return ((l + u * 37) & 63) * 2; });            run("64 lanes, distinct words, stride 2 (2 per bank.. 4)", h, 64, dpat, out);
  fill([](int u, int l) { return ((l + u * 37) & 63) * 4; });            run("64 lanes, distinct words, stride 4", h, 64, dpat, out);
  fill([](int u, int l) { return ((l + u * 37) & 63) * 8; });            run("64 lanes, distinct words, stride 8", h, 64, dpat, out);
  fill([](int u, int l) { return ((l + u * 37) & 31); });                run("64 lanes, 32 words x 2 lanes, 32 banks", h, 64, dpat, out);
  fill([](int u, int l) { return ((l + u * 37) & 15); });                run("64 lanes, 16 words x 4 lanes", h, 64, dpat, out);
  fill([](int u, int l) { return ((l + u * 37) & 31); });                run("32 lanes, 32 distinct words", h, 32, dpat, out);
  fill([](int u, int l) { return ((l + u * 37) & 15); });                run("16 lanes, 16 distinct words", h, 16, dpat, out);
  srand(1);
  fill([](int u, int l) { return rand() % 96; });                        run("64 lanes, random among 96 words", h, 64, dpat, out);
  fill([](int u, int l) { return rand() % 96; });                        run("53 lanes, random among 96 words", h, 53, dpat, out);
  fill([](int u, int l) { return rand() % 96; });                        run("27 lanes, random among 96 words", h, 27, dpat, out);
  fill([](int u, int l) { return rand() % 32; });                        run("53 lanes, random among 32 words", h, 53, dpat, out);
  fill([](int u, int l) { return (rand() % 24) * 4 + (l & 3); });        run("53 lanes, random voxel of 24, channel = lane & 3", h, 53, dpat, out);
  fill([](int u, int l) { return (l + u * 37) & 63; });                  run64("64 lanes, 64 distinct consecutive qwords", h, 64, dpat, out);
  fill([](int u, int l) { return (l + u * 37) & 31; });                  run64("64 lanes, 32 qwords x 2 lanes", h, 64, dpat, out);
  fill([](int u, int l) { return (l + u * 37) & 31; });                  run64("32 lanes, 32 distinct qwords", h, 32, dpat, out);
  fill([](int u, int l) { return rand() % 48; });                        run64("64 lanes, random among 48 qwords", h, 64, dpat, out);
  fill([](int u, int l) { return rand() % 48; });                        run64("53 lanes, random among 48 qwords", h, 53, dpat, out);
  fill([](int u, int l) { return rand() % 48; });                        run64("27 lanes, random among 48 qwords", h, 27, dpat, out);
  hipFree(dpat); hipFree(out);
  return 0;
}
