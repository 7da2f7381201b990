// binning.hip -- tile binning and per-tile depth sort of Gaussian/tile intersections, gfx950.
//
// Replaces gsplat 0.1.11's cumsum -> map_gaussian_to_intersects -> torch.sort(int64 keys) ->
// get_tile_bin_edges chain (the reference reaches it through
// /root/reference/ca_code/utils/render_gsplat.py:65-78).  MI355X-first redesign:
//   * no global 64-bit sort and no host sync on the intersection count: intersections are
//     scattered straight into their tile's segment (count -> scan over T tiles -> scatter with a
//     per-tile cursor), then every tile sorts ITS list inside LDS (one workgroup per tile);
//   * the sort key is (depth bits << 32 | gaussian id): the same front-to-back order as gsplat's
//     (tile << 32 | depth) global sort, with ties broken deterministically by Gaussian id.
// Traffic: 8 B written + 8 B read + 4 B written per intersection, everything else stays in LDS.
#include "gol_common.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__device__ __forceinline__ void tile_bbox(float cx, float cy, float radius, int tiles_x, int tiles_y,
                                          float inv_block, int& x0, int& x1, int& y0, int& y1) {
  const float tcx = cx * inv_block, tcy = cy * inv_block, tr = radius * inv_block;
  x0 = clampi((int)(tcx - tr), 0, tiles_x);
  x1 = clampi((int)(tcx + tr + 1.f), 0, tiles_x);
  y0 = clampi((int)(tcy - tr), 0, tiles_y);
  y1 = clampi((int)(tcy + tr + 1.f), 0, tiles_y);
}

// pass 1: per-tile intersection counts
__global__ __launch_bounds__(256) void count_kernel(int N, const float* __restrict__ xys,
                                                     const int32_t* __restrict__ radii, int tiles_x,
                                                     int tiles_y, float inv_block,
                                                     int32_t* __restrict__ tile_count) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const size_t e = (size_t)b * N + i;
  const int r = radii[e];
  if (r <= 0) return;
  const float2 c = *reinterpret_cast<const float2*>(xys + 2 * e);
  int x0, x1, y0, y1;
  tile_bbox(c.x, c.y, (float)r, tiles_x, tiles_y, inv_block, x0, x1, y0, y1);
  int32_t* tc = tile_count + (size_t)b * tiles_x * tiles_y;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) atomicAdd(tc + y * tiles_x + x, 1);
}

// pass 2: exclusive scan of the T tile counts of one view (one 1024-thread workgroup per view);
// writes tile_bins[t] = (start, start): .y is the scatter cursor and ends up as the end offset.
__global__ __launch_bounds__(1024) void scan_kernel(int T, const int32_t* __restrict__ tile_count,
                                                    int32_t* __restrict__ tile_bins,
                                                    int32_t* __restrict__ n_isect) {
  __shared__ int32_t wave_tot[16];
  __shared__ int32_t carry_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int32_t* cnt = tile_count + (size_t)b * T;
  int2* bins = reinterpret_cast<int2*>(tile_bins) + (size_t)b * T;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < T; base += 1024) {
    const int t = base + tid;
    const int v = t < T ? cnt[t] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int wave_off = 0, chunk_tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int w = wave_tot[k];
      if (k < wv) wave_off += w;
      chunk_tot += w;
    }
    const int carry = carry_s;
    if (t < T) {
      const int start = carry + wave_off + incl - v;
      bins[t] = make_int2(start, start);
    }
    __syncthreads();
    if (tid == 0) carry_s = carry + chunk_tot;
    __syncthreads();
  }
  if (tid == 0) n_isect[b] = carry_s;
}

// pass 3: scatter (depth bits, id) into the tile segments
__global__ __launch_bounds__(256) void scatter_kernel(int N, const float* __restrict__ xys,
                                                       const float* __restrict__ depths,
                                                       const int32_t* __restrict__ radii, int tiles_x,
                                                       int tiles_y, float inv_block, int64_t capacity,
                                                       int32_t* __restrict__ tile_bins,
                                                       uint64_t* __restrict__ isect_keys) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const size_t e = (size_t)b * N + i;
  const int r = radii[e];
  if (r <= 0) return;
  const float2 c = *reinterpret_cast<const float2*>(xys + 2 * e);
  int x0, x1, y0, y1;
  tile_bbox(c.x, c.y, (float)r, tiles_x, tiles_y, inv_block, x0, x1, y0, y1);
  const uint64_t key = ((uint64_t)__float_as_uint(depths[e]) << 32) | (uint32_t)i;
  int32_t* bins = tile_bins + (size_t)b * tiles_x * tiles_y * 2;
  uint64_t* keys = isect_keys + (size_t)b * capacity;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      const int slot = atomicAdd(bins + 2 * (y * tiles_x + x) + 1, 1);
      if (slot < capacity) keys[slot] = key;
    }
}

// pass 4: one workgroup per tile sorts its list.  Bitonic network in the "all-ascending" form
// (first sub-step of each stage pairs i with i ^ (2k-1)), so lists of any length work without
// padding: a partner index >= n stands for +inf and the exchange is skipped.
constexpr int kSortLds = 4096;  // keys staged in LDS (32 KiB); longer lists sort in global memory

template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr keys, int n, int tid, int nthreads) {
  int P = 1;
  while (P < n) P <<= 1;
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const bool flip = (j == (k >> 1));
      for (int t = tid; t < (P >> 1); t += nthreads) {
        // t-th pair of this sub-step
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int hi = flip ? (lo ^ (k - 1)) : (lo | j);
        if (hi < n) {
          const uint64_t a = keys[lo], c = keys[hi];
          if (a > c) { keys[lo] = c; keys[hi] = a; }
        }
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void sort_kernel(int T, int64_t capacity, int32_t* __restrict__ tile_bins,
                                                   uint64_t* __restrict__ isect_keys,
                                                   int32_t* __restrict__ sorted_ids) {
  __shared__ uint64_t lds_keys[kSortLds];
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  int2* binp = reinterpret_cast<int2*>(tile_bins) + (size_t)b * T + t;
  int2 bin = *binp;
  __syncthreads();  // everyone has read the bin before thread 0 may clamp it
  // clamp to capacity (overflow is reported through n_isect; keep the bins self-consistent)
  int start = bin.x, end = bin.y;
  if (start > capacity) start = (int)capacity;
  if (end > capacity) end = (int)capacity;
  if (tid == 0 && (start != bin.x || end != bin.y)) *binp = make_int2(start, end);
  const int n = end - start;
  if (n <= 0) return;
  uint64_t* keys = isect_keys + (size_t)b * capacity + start;
  int32_t* out = sorted_ids + (size_t)b * capacity + start;
  if (n == 1) {
    if (tid == 0) out[0] = (int32_t)(uint32_t)keys[0];
    return;
  }
  if (n <= kSortLds) {
    for (int i = tid; i < n; i += 256) lds_keys[i] = keys[i];
    __syncthreads();
    bitonic_sort(lds_keys, n, tid, 256);
    for (int i = tid; i < n; i += 256) out[i] = (int32_t)(uint32_t)lds_keys[i];
  } else {
    // rare: a tile covered by > 4096 Gaussians; same network straight on global memory
    // (one workgroup = one CU, so __syncthreads() orders its own global stores and loads)
    __syncthreads();
    bitonic_sort(keys, n, tid, 256);
    for (int i = tid; i < n; i += 256) out[i] = (int32_t)(uint32_t)keys[i];
  }
}

}  // namespace

extern "C" int gol_bin_sort(int B, int N, const float* xys, const float* depths, const int32_t* radii,
                            int img_h, int img_w, int block, int64_t capacity, int32_t* tile_count,
                            int count_done, int32_t* tile_bins, uint64_t* isect_keys, int32_t* sorted_ids,
                            int32_t* n_isect, void* stream) {
  GOL_REQUIRE(B >= 0 && N >= 0, "negative size");
  GOL_REQUIRE(block > 1 && block <= 16, "block_width must be between 2 and 16");
  GOL_REQUIRE(img_h > 0 && img_w > 0, "empty image");
  GOL_REQUIRE(capacity >= 0 && capacity < (1ll << 31), "capacity out of range");
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(B <= 65535, "B > 65535");
  GOL_REQUIRE(tile_count && tile_bins && n_isect, "null workspace");
  GOL_REQUIRE(N == 0 || (xys && depths && radii), "null input");
  GOL_REQUIRE(capacity == 0 || (isect_keys && sorted_ids), "null intersection buffers");
  hipStream_t s = (hipStream_t)stream;
  const int tiles_x = (img_w + block - 1) / block, tiles_y = (img_h + block - 1) / block;
  const int T = tiles_x * tiles_y;
  const float inv_block = 1.f / (float)block;
  dim3 ggrid(gol_cdiv(N > 0 ? N : 1, 256), B);
  if (!count_done) {
    if (hipMemsetAsync(tile_count, 0, sizeof(int32_t) * (size_t)B * T, s) != hipSuccess) {
      gol_set_error("gol_bin_sort: hipMemsetAsync failed");
      return GOL_ERR_LAUNCH;
    }
    if (N > 0) count_kernel<<<ggrid, 256, 0, s>>>(N, xys, radii, tiles_x, tiles_y, inv_block, tile_count);
  }
  scan_kernel<<<B, 1024, 0, s>>>(T, tile_count, tile_bins, n_isect);
  if (N > 0 && capacity > 0) {
    scatter_kernel<<<ggrid, 256, 0, s>>>(N, xys, depths, radii, tiles_x, tiles_y, inv_block, capacity, tile_bins,
                                         isect_keys);
    dim3 sgrid(T, B);
    sort_kernel<<<sgrid, 256, 0, s>>>(T, capacity, tile_bins, isect_keys, sorted_ids);
  }
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
