// gather_probe.hip -- calibration of rocprofv3's FETCH_SIZE for the access patterns of this repo's gather kernels
// (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern").  Kernels, each with an exactly known byte count:
//   stream16        coalesced 16 B / lane reads of a 1 GiB buffer                           (the guide's calibration case)
//   gather64        one 64-byte record per lane at a random, 64-byte aligned position of a TABLE_MB table, read as 4 x 16 B
//                   (shade.hip's env footprint records, raster.hip's splat records)
//   gather128       one 128-byte aligned pair of records per lane (8 x 16 B)
//   gather16        one random, 16-byte aligned 16 B texel per lane
// Tables of 1 GiB (beyond L2 + the 256 MiB Infinity Cache) and 32 MiB (Infinity-Cache resident).  Prints the time and the
// useful bytes of each launch; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and divide (tools/gather_calibration.py).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/gather_probe.hip -o tools/bin/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void stream16(const float4* __restrict__ src, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) sink[0] = acc;
}

// BYTES per lookup (16, 64 or 128), aligned to BYTES; `records` = table size / BYTES; `per_lane` lookups per lane
template <int BYTES>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ table, uint32_t records, int per_lane, float* sink) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (int k = 0; k < per_lane; ++k) {
    const uint32_t r = hash32(g * 9781u + (uint32_t)k * 6271u + 17u) % records;
    const float4* p = table + (size_t)r * (BYTES / 16);
#pragma unroll
    for (int j = 0; j < BYTES / 16; ++j) {
      const float4 v = p[j];
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <typename F>
static void timed(const char* name, double bytes, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 3; ++it) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s useful %9.1f MB per launch  %8.1f us  %7.1f GB/s useful\n", name, bytes / 1e6, ms * 1e3 / 3, bytes / (ms / 3 * 1e-3) / 1e9);
}

int main() {
  const size_t big = (size_t)1 << 30, small = (size_t)32 << 20;
  float4 *tb, *ts; float* sink;
  hipMalloc(&tb, big); hipMalloc(&ts, small); hipMalloc(&sink, 64);
  hipMemset(tb, 0, big); hipMemset(ts, 0, small);
  const int blocks = 4096, per_lane = 4;                    // 4096 * 256 * 4 = 4.19 M lookups per launch
  const double lookups = (double)blocks * 256 * per_lane;
  timed("stream16 (1 GiB, coalesced 16 B/lane)", (double)big, [&] { stream16<<<2048, 256>>>(tb, big / 16, sink); });
  timed("gather16  from 1 GiB", lookups * 16, [&] { gather<16><<<blocks, 256>>>(tb, (uint32_t)(big / 16), per_lane, sink); });
  timed("gather64  from 1 GiB", lookups * 64, [&] { gather<64><<<blocks, 256>>>(tb, (uint32_t)(big / 64), per_lane, sink); });
  timed("gather128 from 1 GiB", lookups * 128, [&] { gather<128><<<blocks, 256>>>(tb, (uint32_t)(big / 128), per_lane, sink); });
  timed("gather64  from 32 MiB", lookups * 64, [&] { gather<64><<<blocks, 256>>>(ts, (uint32_t)(small / 64), per_lane, sink); });
  timed("gather128 from 32 MiB", lookups * 128, [&] { gather<128><<<blocks, 256>>>(ts, (uint32_t)(small / 128), per_lane, sink); });
  return 0;
}
