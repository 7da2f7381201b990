// global_atomic_probe.hip -- throughput of RETURNING device-scope atomicAdd(int) as a function of how many distinct
// addresses the operations share (the tile cursors of the binning scatter: ~1.3 M operations on ~3000 hot addresses per view).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/global_atomic_probe.hip -o tools/bin/gatomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>

// every lane issues `per_lane` atomics; address = hash(lane id, k) % n_addr (stride ints apart); 4 in flight per lane
template <bool RTN>
__global__ __launch_bounds__(256) void probe(int* cur, int n_addr, int stride, int per_lane, int* sink) {
  const unsigned g = blockIdx.x * 256 + threadIdx.x;
  int acc = 0;
  for (int k = 0; k < per_lane; k += 4) {
    int r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned h = (g * 2654435761u + (unsigned)(k + u) * 40503u) >> 7;
      int* p = cur + (size_t)(h % (unsigned)n_addr) * stride;
      if (RTN) r[u] = atomicAdd(p, 1); else { atomicAdd(p, 1); r[u] = 0; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += r[u];
  }
  if (acc == 0x7fffffff) sink[0] = acc;
}

template <bool RTN>
static void run(const char* name, int* cur, int n_addr, int stride, int* sink) {
  const int blocks = 2048, per_lane = 4;  // 2048 * 256 * 4 = 2.1 M operations
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<RTN><<<blocks, 256>>>(cur, n_addr, stride, per_lane, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) probe<RTN><<<blocks, 256>>>(cur, n_addr, stride, per_lane, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  const double ops = 5.0 * blocks * 256 * per_lane;
  printf("%-56s %8.1f us per 2.1 M ops   %7.2f G ops/s\n", name, ms * 1e3 / 5, ops / (ms * 1e-3) / 1e9);
}

int main() {
  int *cur, *sink;
  hipMalloc(&cur, (size_t)(1 << 22) * 16 * sizeof(int)); hipMemset(cur, 0, (size_t)(1 << 22) * 16 * sizeof(int));
  hipMalloc(&sink, 64);
  run<true>("returning, 1 address", cur, 1, 1, sink);
  run<true>("returning, 64 addresses (consecutive ints)", cur, 64, 1, sink);
  run<true>("returning, 3000 addresses, stride 2 ints", cur, 3000, 2, sink);
  run<true>("returning, 24000 addresses, stride 2 ints", cur, 24000, 2, sink);
  run<true>("returning, 24000 addresses, stride 16 ints (own line)", cur, 24000, 16, sink);
  run<true>("returning, 86000 addresses, stride 2 ints", cur, 86000, 2, sink);
  run<true>("returning, 1 M addresses, stride 2 ints", cur, 1 << 20, 2, sink);
  run<true>("returning, 4 M addresses, stride 16 ints", cur, 1 << 22, 16, sink);
  run<false>("no return, 3000 addresses, stride 2 ints", cur, 3000, 2, sink);
  run<false>("no return, 24000 addresses, stride 2 ints", cur, 24000, 2, sink);
  run<false>("no return, 1 M addresses, stride 2 ints", cur, 1 << 20, 2, sink);
  return 0;
}
