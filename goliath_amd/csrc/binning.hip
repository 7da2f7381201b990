// binning.hip -- tile binning and per-tile depth sort of Gaussian/tile intersections, gfx950.
//
// Replaces gsplat 0.1.11's cumsum -> map_gaussian_to_intersects -> torch.sort(int64 keys) ->
// get_tile_bin_edges chain (the reference reaches it through
// /root/reference/ca_code/utils/render_gsplat.py:65-78).  MI355X-first redesign:
//   * no global 64-bit sort and no host sync on the intersection count: intersections are
//     scattered straight into their tile's segment (count -> scan over T tiles -> scatter with a
//     per-tile cursor), then every tile sorts ITS list inside LDS (one workgroup per tile);
//   * the sort key is (depth bits << 32 | gaussian id): the same front-to-back order as gsplat's
//     (tile << 32 | depth) global sort, with ties broken deterministically by Gaussian id;
//   * (Gaussian, tile) pairs that cannot reach alpha >= 1/255 anywhere in the tile are dropped at
//     binning time (output-preserving, see tile_box) and the per-tile counters live in LDS during
//     the walks over the Gaussians, so hot tiles do not serialise on global atomics;
//   * round 3: (a) the count pass only RESERVES: it walks the tight tile boxes without the exact per-tile test
//     (an upper bound of the list lengths, ~1.2x), so the segments of a view are sized generously and the exact test
//     runs once, in the scatter pass, which keeps its result in registers for its second walk (the 8-byte-per-
//     Gaussian mask buffer between the passes is gone); tile_bins[t] = (start, start + exact length);
//     (b) the per-tile sort is a bucket sort, not a compare-exchange network: the depths of a tile are spread over
//     ~n buckets between the tile's nearest and farthest entry (monotone map, LDS histogram + scan), and every entry
//     finds its place inside its bucket by counting the smaller keys there (1-3 compares on average) -- ~50
//     instructions per key instead of the ~330 of the 55-step bitonic network of a 1024-key list, whose 2 x 55
//     ds_bpermute exchanges per key kept the LDS crossbar busy for 0.32 ms per 8 views.  Lists with pathological
//     depth clustering (a bucket of more than 48 entries) fall back to the network;
//   * round 4: the tile scan is one pass (counts staged in LDS); lists of more than 2048 entries are queued for a 1024-thread
//     kernel that bucket-sorts up to 16384 keys in LDS (a third of the tiles at 1 M Gaussians);
//   * round 6: two queues -- lists of 2049 ... 4096 entries (all the long lists there are at 250 k and at 1 M Gaussians) go to
//     256-thread workgroups with 36 KB of static LDS, four per CU (sort_mid_kernel: 0.35 -> 0.18 ms per 8 views at 1 M); the
//     1024-thread kernel keeps the lists above 4096 and no longer REQUIRES its 147 KB of dynamic LDS (global-memory fallback).
// Traffic: 8 B written + 8 B read + 4 B written per intersection, everything else stays in LDS.
#include <atomic>
#include <cstdlib>

#include "gol_common.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

struct TileBox { int x0, x1, y0, y1; };  // [x0,x1) x [y0,y1) in tile units

// gsplat's tile bbox (square of the 3-sigma radius, C (int) truncation), optionally
// intersected with the tiles that the alpha >= 1/255 ellipse of the Gaussian can reach.  The
// second test never changes an image or a gradient: a (Gaussian, tile) pair it removes would be
// skipped for every pixel of the tile by the rasterizer's alpha < 1/255 test (A.3) -- it only
// removes dead entries from the lists (about half of them for anisotropic splats).
__device__ __forceinline__ TileBox tile_box(float gx, float gy, float radius, int tiles_x, int tiles_y,
                                            float inv_block, float block, bool tight, float ca, float cb,
                                            float cc, float op) {
  TileBox t;
  const float tcx = gx * inv_block, tcy = gy * inv_block, tr = radius * inv_block;
  t.x0 = clampi((int)(tcx - tr), 0, tiles_x);
  t.x1 = clampi((int)(tcx + tr + 1.f), 0, tiles_x);
  t.y0 = clampi((int)(tcy - tr), 0, tiles_y);
  t.y1 = clampi((int)(tcy + tr + 1.f), 0, tiles_y);
  if (tight) {
    const float tau = gol_alpha_tau(op);
    if (!(tau >= 0.f)) { t.x1 = t.x0; return t; }  // alpha < 1/255 everywhere
    const float det = ca * cc - cb * cb;
    if (det > 0.f && ca > 0.f && cc > 0.f) {
      const float tau2 = 2.f * tau;
      const float hx = sqrtf(tau2 * cc / det) + 0.02f, hy = sqrtf(tau2 * ca / det) + 0.02f;
      if (hx < 1e30f && hy < 1e30f) {
        // tile tx holds pixel centres [block*tx + 0.5, block*tx + block - 0.5]
        const float lo_x = ceilf((gx - hx - (block - 0.5f)) * inv_block), hi_x = floorf((gx + hx - 0.5f) * inv_block);
        const float lo_y = ceilf((gy - hy - (block - 0.5f)) * inv_block), hi_y = floorf((gy + hy - 0.5f) * inv_block);
        t.x0 = max(t.x0, (int)fmaxf(lo_x, 0.f));
        t.y0 = max(t.y0, (int)fmaxf(lo_y, 0.f));
        t.x1 = min(t.x1, (int)fminf(hi_x, (float)tiles_x) + 1);
        t.y1 = min(t.y1, (int)fminf(hi_y, (float)tiles_y) + 1);
        if (t.x1 < t.x0) t.x1 = t.x0;
        if (t.y1 < t.y0) t.y1 = t.y0;
      }
    }
  }
  return t;
}

// exact per-tile test used inside the tile loops (only when the conic is well formed)
struct Reach { float gx, gy, a, b, c, ia, ic, tau; bool exact; };

__device__ __forceinline__ bool tile_reached(const Reach& r, int x, int y, float block) {
  if (!r.exact) return true;
  const float x0 = (float)x * block + 0.5f, y0 = (float)y * block + 0.5f;
  return gol_min_sigma_rect(r.gx, r.gy, r.a, r.b, r.c, r.ia, r.ic, x0, x0 + block - 1.f, y0, y0 + block - 1.f) <= r.tau;
}

// The tiles of ONE tile row that the alpha >= 1/255 region E = {sigma <= tau} of a Gaussian reaches, in closed form (round 4:
// the scatter pass is VALU-bound -- PMC: 0.83 VALU-busy, 100 M wave-instructions per 8 views at 45 % useful lanes -- and ~80 % of
// them were the per-TILE ellipse-vs-rectangle test inside the divergent box walk, ~55 instructions for each of up to 49 tiles
// per lane).  E is convex, so E intersected with the row's strip of pixel-centre ordinates [y0, y1] projects onto an x-INTERVAL
// [xl, xr], and a tile of the row is reached iff its pixel-centre abscissae [x0, x1] meet it: two square roots per row instead
// of a test per tile.  With d = y - gy in [d0, d1] clipped to the ellipse's extent |d| <= hy: the line y = gy + d cuts E in
// x - gx in [(-b d - sqrt(D)) / a, (-b d + sqrt(D)) / a], D = 2 a tau - det d^2; the right end is largest at the ellipse's
// rightmost point (d = -b hx / c) if the strip contains it, else at one of the strip's two lines (same for the left end).
// 0.01 px of margin on either side against rounding (tau itself is inflated by gol_alpha_tau).  Returns false = row not reached.
struct RowSpan { float hx, hy, dr, det, two_a_tau, ia; };   // per Gaussian: half extents, d of the rightmost point, ...
__device__ __forceinline__ RowSpan row_span_setup(const Reach& r) {
  RowSpan s;
  s.det = r.a * r.c - r.b * r.b;
  const float tau2 = 2.f * r.tau;
  s.hx = sqrtf(tau2 * r.c / s.det); s.hy = sqrtf(tau2 * r.a / s.det);
  s.dr = -r.b * s.hx * r.ic;
  s.two_a_tau = tau2 * r.a; s.ia = r.ia;
  return s;
}
__device__ __forceinline__ bool row_span(const Reach& r, const RowSpan& s, float y0, float y1, float& xl, float& xr) {
  const float dlo = fmaxf(y0 - r.gy, -s.hy), dhi = fminf(y1 - r.gy, s.hy);
  if (!(dlo <= dhi)) return false;
  const float s0 = sqrtf(fmaxf(s.two_a_tau - s.det * dlo * dlo, 0.f)), s1 = sqrtf(fmaxf(s.two_a_tau - s.det * dhi * dhi, 0.f));
  const float r0 = (-r.b * dlo + s0) * s.ia, r1 = (-r.b * dhi + s1) * s.ia;
  const float l0 = (-r.b * dlo - s0) * s.ia, l1 = (-r.b * dhi - s1) * s.ia;
  const float right = (s.dr >= dlo && s.dr <= dhi) ? s.hx : fmaxf(r0, r1);
  const float left = (-s.dr >= dlo && -s.dr <= dhi) ? -s.hx : fminf(l0, l1);
  xl = r.gx + left - 0.01f; xr = r.gx + right + 0.01f;
  return true;
}

struct BinArgs {
  int N, tiles_x, tiles_y, chunk;
  float inv_block, block;
  const float* xys; const float* depths; const int32_t* radii; const float* conics; const float* opacities;
};

__device__ __forceinline__ TileBox box_of(const BinArgs& a, size_t e, Reach& rc) {
  const int r = a.radii[e];
  rc.exact = false;
  if (r <= 0) return TileBox{0, 0, 0, 0};
  const float2 c = *reinterpret_cast<const float2*>(a.xys + 2 * e);
  const bool tight = a.conics != nullptr;
  float ca = 0.f, cb = 0.f, cc = 0.f, op = 1.f;
  if (tight) {
    ca = a.conics[3 * e]; cb = a.conics[3 * e + 1]; cc = a.conics[3 * e + 2]; op = a.opacities[e];
    rc.gx = c.x; rc.gy = c.y; rc.a = ca; rc.b = cb; rc.c = cc; rc.tau = gol_alpha_tau(op);
    rc.exact = (ca * cc - cb * cb > 0.f) && ca > 0.f && cc > 0.f;
    rc.ia = 1.f / ca; rc.ic = 1.f / cc;  // once per Gaussian, not once per tile test
  }
  return tile_box(c.x, c.y, (float)r, a.tiles_x, a.tiles_y, a.inv_block, a.block, tight, ca, cb, cc, op);
}

// tile box only (no per-tile test state): what the reserving count pass needs
__device__ __forceinline__ TileBox box_only(const BinArgs& a, size_t e) {
  const int r = a.radii[e];
  if (r <= 0) return TileBox{0, 0, 0, 0};
  const float2 c = *reinterpret_cast<const float2*>(a.xys + 2 * e);
  const bool tight = a.conics != nullptr;
  float ca = 0.f, cb = 0.f, cc = 0.f, op = 1.f;
  if (tight) { ca = a.conics[3 * e]; cb = a.conics[3 * e + 1]; cc = a.conics[3 * e + 2]; op = a.opacities[e]; }
  return tile_box(c.x, c.y, (float)r, a.tiles_x, a.tiles_y, a.inv_block, a.block, tight, ca, cb, cc, op);
}

// ---- pass 1: per-tile RESERVATION counts -------------------------------------------------------
// One 1024-thread workgroup walks a chunk of Gaussians and histograms the tiles of their (tight) boxes in LDS
// (one int per tile of the view, ds_add_u32), then flushes the non-zero bins with ONE global atomic each.  No exact
// per-tile test here: the counts are upper bounds (exact when no conics are given: gsplat's lists), the scatter pass
// tests once and fills only what passes.
__global__ __launch_bounds__(1024) void count_lds_kernel(BinArgs a, int32_t* __restrict__ tile_count) {
  extern __shared__ int32_t s_cnt[];
  const int b = blockIdx.y, T = a.tiles_x * a.tiles_y;
  for (int t = threadIdx.x; t < T; t += 1024) s_cnt[t] = 0;
  __syncthreads();
  const int i_end = min(a.N, (int)(blockIdx.x + 1) * a.chunk);
  for (int i = blockIdx.x * a.chunk + threadIdx.x; i < i_end; i += 1024) {
    const TileBox tb = box_only(a, (size_t)b * a.N + i);
    for (int y = tb.y0; y < tb.y1; ++y)
      for (int x = tb.x0; x < tb.x1; ++x) atomicAdd(&s_cnt[y * a.tiles_x + x], 1);
  }
  __syncthreads();
  int32_t* tc = tile_count + (size_t)b * T;
  for (int t = threadIdx.x; t < T; t += 1024) {
    const int c = s_cnt[t];
    if (c) atomicAdd(tc + t, c);
  }
}

// ---- pass 3: scatter (depth bits, id) into the tile segments -----------------------------------
// LDS histogram of the pairs that pass the exact test; each workgroup then reserves a contiguous range per tile with
// one returning global atomic and hands out slots inside it with LDS atomics.  The test results of the first walk
// stay in registers (one 64-bit mask per Gaussian, kMaskPerLane Gaussians per lane: chunk <= 4096) for the second.
constexpr int kMaskPerLane = 4;

__global__ __launch_bounds__(1024) void scatter_lds_kernel(BinArgs a, int64_t capacity,
                                                           int32_t* __restrict__ tile_bins,
                                                           uint64_t* __restrict__ isect_keys) {
  // LDS: per-tile counters packed two to a word (a workgroup sees at most `chunk` <= 4096 Gaussians, so 16 bits
  // hold any count) + per-tile 32-bit base offsets: 6 bytes per tile instead of 8, i.e. 64.5 KB at 2048x1334 -- two
  // workgroups per CU instead of one.  What bounds the kernel (0.19 ms per 8 views of 2 M stored entries each) is neither its
  // instruction count (round 4: the per-tile test replaced by per-row intervals, 100 M -> 65 M VALU wave-instructions:
  // 196 -> 189 us, VALU-busy 0.83 -> 0.57) nor its reservation atomics / key runs (a cell-ordered variant with 20x fewer
  // atomics and 50-key runs: 187 us + 22 us of ordering) nor the reservation loop's round trips (all atomics of a lane in
  // flight at once: no change): two workgroups per CU, phase-locked by their four barriers, wait.  Measured and dropped
  // in round 3: keeping the boxes and depths of the first walk in registers for the second (68 -> 102 VGPRs = one
  // workgroup per CU: +8 %), the same under a 64-register cap (spills: +3 %), four reservation atomics in flight per
  // lane (no change); a PAIR-balanced walk (the wave numbers its (Gaussian, tile) pairs with a scan, every lane tests
  // one pair per step: 100 M -> ~40 M vector instructions, bit-identical lists): 0.174 ms with 1024 Gaussians per
  // workgroup, 0.21 / 0.24 with 2048 / 4096 (longer key runs, but fewer, longer workgroups), and 0.51 ms with one
  // returning device atomic per pair instead of the LDS counters (profiles/r03g_global_atomic_probe.txt: device atomics
  // top out at 26 G/s over >= 3000 addresses and fall to 1.3 G/s on 64) -- 12 % of one kernel for twice the code: not kept.
  extern __shared__ int32_t s_mem[];
  const int b = blockIdx.y, T = a.tiles_x * a.tiles_y, Tw = (T + 1) >> 1;
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_mem);
  int32_t* s_base = s_mem + Tw;
  for (int t = threadIdx.x; t < Tw; t += 1024) s_cnt[t] = 0u;
  __syncthreads();
  const int i_begin = blockIdx.x * a.chunk + threadIdx.x, i_end = min(a.N, (int)(blockIdx.x + 1) * a.chunk);
  uint64_t masks[kMaskPerLane];
#pragma unroll
  for (int k = 0; k < kMaskPerLane; ++k) {
    masks[k] = 0;
    const int i = i_begin + k * 1024;
    if (i >= i_end) continue;
    Reach rc;
    const TileBox tb = box_of(a, (size_t)b * a.N + i, rc);
    const int bw = tb.x1 - tb.x0;
    uint64_t mask = 0;
    RowSpan rs;
    if (rc.exact) rs = row_span_setup(rc);
    for (int y = tb.y0; y < tb.y1; ++y) {
      int xa = tb.x0, xb = tb.x1 - 1;
      if (rc.exact) {   // the reached tiles of this row: an interval (row_span)
        const float y0 = (float)y * a.block + 0.5f;
        float xl, xr;
        if (!row_span(rc, rs, y0, y0 + a.block - 1.f, xl, xr)) continue;
        // tile x holds pixel centres [block x + 0.5, block x + block - 0.5]
        xa = max(xa, (int)ceilf((xl - (a.block - 0.5f)) * a.inv_block));
        xb = min(xb, (int)floorf((xr - 0.5f) * a.inv_block));
      }
      for (int x = xa; x <= xb; ++x) {
        const int t = y * a.tiles_x + x;
        atomicAdd(&s_cnt[t >> 1], 1u << ((t & 1) << 4));
        mask |= 1ull << (((y - tb.y0) * bw + (x - tb.x0)) & 63);
      }
    }
    masks[k] = mask;
  }
  __syncthreads();
  int32_t* bins = tile_bins + (size_t)b * T * 2;
  for (int t = threadIdx.x; t < T; t += 1024) {
    const int c = (int)((s_cnt[t >> 1] >> ((t & 1) << 4)) & 0xffffu);
    if (c) s_base[t] = atomicAdd(bins + 2 * t + 1, c);
  }
  __syncthreads();
  // (the second walk hands out slots by advancing the tile's base itself: no second pass over the counters)
  uint64_t* keys = isect_keys + (size_t)b * capacity;
#pragma unroll
  for (int k = 0; k < kMaskPerLane; ++k) {
    const int i = i_begin + k * 1024;
    if (i >= i_end) continue;
    const size_t e = (size_t)b * a.N + i;
    Reach rc;
    const TileBox tb = box_of(a, e, rc);
    if (tb.x1 <= tb.x0 || tb.y1 <= tb.y0) continue;
    const uint64_t key = ((uint64_t)__float_as_uint(a.depths[e]) << 32) | (uint32_t)i;
    const int bw = tb.x1 - tb.x0;
    const bool hm = bw * (tb.y1 - tb.y0) <= 64;   // larger boxes are re-tested the same way (their mask wrapped around)
    const uint64_t mask = masks[k];
    RowSpan rs;
    if (!hm && rc.exact) rs = row_span_setup(rc);
    for (int y = tb.y0; y < tb.y1; ++y) {
      int xa = tb.x0, xb = tb.x1 - 1;
      if (!hm && rc.exact) {
        const float y0 = (float)y * a.block + 0.5f;
        float xl, xr;
        if (!row_span(rc, rs, y0, y0 + a.block - 1.f, xl, xr)) continue;
        xa = max(xa, (int)ceilf((xl - (a.block - 0.5f)) * a.inv_block));
        xb = min(xb, (int)floorf((xr - 0.5f) * a.inv_block));
      }
      for (int x = xa; x <= xb; ++x) {
        if (hm && !((mask >> ((y - tb.y0) * bw + (x - tb.x0))) & 1ull)) continue;
        const int slot = atomicAdd(&s_base[y * a.tiles_x + x], 1);
        if (slot < capacity) keys[slot] = key;
      }
    }
  }
}

// ---- fallbacks for images with more tiles than fit in LDS: direct global atomics ----------------
__global__ __launch_bounds__(256) void count_kernel(BinArgs a, int32_t* __restrict__ tile_count) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N) return;
  Reach rc;
  const TileBox tb = box_of(a, (size_t)b * a.N + i, rc);
  int32_t* tc = tile_count + (size_t)b * a.tiles_x * a.tiles_y;
  for (int y = tb.y0; y < tb.y1; ++y)
    for (int x = tb.x0; x < tb.x1; ++x)
      if (tile_reached(rc, x, y, a.block)) atomicAdd(tc + y * a.tiles_x + x, 1);
}

__global__ __launch_bounds__(256) void scatter_kernel(BinArgs a, int64_t capacity, int32_t* __restrict__ tile_bins,
                                                       uint64_t* __restrict__ isect_keys) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N) return;
  const size_t e = (size_t)b * a.N + i;
  Reach rc;
  const TileBox tb = box_of(a, e, rc);
  if (tb.x1 <= tb.x0 || tb.y1 <= tb.y0) return;
  const uint64_t key = ((uint64_t)__float_as_uint(a.depths[e]) << 32) | (uint32_t)i;
  int32_t* bins = tile_bins + (size_t)b * a.tiles_x * a.tiles_y * 2;
  uint64_t* keys = isect_keys + (size_t)b * capacity;
  for (int y = tb.y0; y < tb.y1; ++y)
    for (int x = tb.x0; x < tb.x1; ++x) {
      if (!tile_reached(rc, x, y, a.block)) continue;
      const int slot = atomicAdd(bins + 2 * (y * a.tiles_x + x) + 1, 1);
      if (slot < capacity) keys[slot] = key;
    }
}

// pass 2: exclusive scan of the T tile counts of one view (one 1024-thread workgroup per view).
// Writes tile_bins[t] = (start, start): .y is the scatter cursor and ends up as the end offset.
// Round 4: ONE pass -- the counts are staged in LDS (coalesced), every thread sums its own run of consecutive tiles, one
// wave scan + one cross-wave step give the offsets (the 11 strided iterations with three barriers each of rounds 1-3 cost
// 11.6 us per launch whatever B: one workgroup per view, i.e. a serial phase of every single-view step).
// Measured and dropped in round 4: a launch order for the rasterizer computed here (per die the same tile rows, longest
// list first, so that the longest lists start first): no effect on either raster kernel at 1, 2, 4 or 8 views per launch
// (profiles/r04_raster_tail.txt) -- 2942 of a view's tiles are non-empty and 1050 of them hold more than 896 entries: the
// long lists are the bulk of the work, not a tail, and a single-view launch lasts exactly as long as its longest list.
constexpr int kScanLds = 12288;   // tiles of a view staged in LDS by the one-pass scan (48 KB); larger images loop

__global__ __launch_bounds__(1024) void scan_kernel(int T, const int32_t* __restrict__ tile_count,
                                                    int32_t* __restrict__ tile_bins,
                                                    int32_t* __restrict__ n_isect) {
  __shared__ int32_t s_cnt[kScanLds];
  __shared__ int32_t wave_tot[16];
  __shared__ int32_t carry_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int32_t* cnt = tile_count + (size_t)b * T;
  int2* bins = reinterpret_cast<int2*>(tile_bins) + (size_t)b * T;
  const bool staged = T <= kScanLds;
  if (staged) {
    for (int t = tid; t < T; t += 1024) s_cnt[t] = cnt[t];
    __syncthreads();
    const int per = (T + 1023) >> 10, t0 = tid * per, t1 = min(T, t0 + per);
    int sum = 0;
    for (int t = t0; t < t1; ++t) sum += s_cnt[t];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int start = incl - sum, total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int w = wave_tot[k];
      if (k < wv) start += w;
      total += w;
    }
    for (int t = t0; t < t1; ++t) {
      bins[t] = make_int2(start, start);
      start += s_cnt[t];
    }
    if (tid == 0) n_isect[b] = total;
  } else {
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
  __syncthreads();   // (everyone is done with the counts)
  // the counts are consumed: their buffer becomes the view's queue of long tile lists (sort_kernel); word 0 = its length
  if (tid == 0) { const_cast<int32_t*>(cnt)[0] = 0; const_cast<int32_t*>(cnt)[T - 1] = 0; }
}

// pass 4: one workgroup per tile sorts its list.  Bitonic network in the "all-ascending" form
// (first sub-step of each stage pairs i with i ^ (2k-1)), so lists of any length work without
// padding: a partner index >= n stands for +inf and the exchange is skipped.

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


// ---- register-blocked bitonic sort ---------------------------------------------------------------------------
// 256 threads x KPT keys (thread tid owns elements tid*KPT .. tid*KPT+KPT-1, list padded with +inf keys).  A
// compare-exchange step with partner distance j touches
//   j < KPT            two registers of the same lane                          (no data movement at all)
//   KPT <= j < 64 KPT  the same register of lane ^ (j / KPT)                   (cross-lane permute, no barrier)
//   j >= 64 KPT        another wave                                            (through LDS, two barriers)
// so of the log2(P)(log2(P)+1)/2 steps (55 for P = 1024) only 3 (P = 1024) need the workgroup barrier that the
// plain LDS network pays on every step.  The first step of every merge is the "flip" (partner = i ^ (k-1)), which
// keeps all later steps of the merge ascending (no direction flags); in lane/register terms the flip partner is
// lane ^ (k/KPT - 1), register KPT-1-r.
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int mask) {
  const unsigned lo = __shfl_xor((unsigned)v, mask, 64), hi = __shfl_xor((unsigned)(v >> 32), mask, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ void cex(uint64_t& a, uint64_t& b) {  // a <= b afterwards
  const bool sw = b < a;
  const uint64_t t = sw ? b : a;
  b = sw ? a : b;
  a = t;
}
// keep the smaller (lower == true) or the larger of (self, other): one compare, one 64-bit select
__device__ __forceinline__ uint64_t pick(uint64_t self, uint64_t other, bool lower) {
  return ((self < other) == lower) ? self : other;
}

template <int KPT>
__device__ __forceinline__ void block_bitonic(uint64_t (&v)[KPT], uint64_t* __restrict__ lds, int tid) {
  constexpr int P = 256 * KPT;
  const int lane = tid & 63;
#pragma unroll
  for (int k = 2; k <= P; k <<= 1) {
    // ---- flip step: element i pairs with i ^ (k-1) ----
    if (k <= KPT) {
#pragma unroll
      for (int r = 0; r < KPT; ++r) {
        const int q = r ^ (k - 1);
        if (q > r) cex(v[r], v[q]);
      }
    } else if (k <= 64 * KPT) {
      const int m = k / KPT - 1;
      const bool lower = (lane & (k / KPT / 2)) == 0;
      uint64_t o[KPT];
#pragma unroll
      for (int r = 0; r < KPT; ++r) o[r] = shfl_xor64(v[KPT - 1 - r], m);
#pragma unroll
      for (int r = 0; r < KPT; ++r) {
        v[r] = pick(v[r], o[r], lower);
      }
    } else {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < KPT; ++r) lds[tid * KPT + r] = v[r];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < KPT; ++r) {
        const int i = tid * KPT + r;
        const uint64_t o = lds[i ^ (k - 1)];
        const bool lower = (i & (k >> 1)) == 0;
        v[r] = pick(v[r], o, lower);
      }
    }
    // ---- ascending steps: element i (bit j clear) pairs with i | j ----
#pragma unroll
    for (int j = k >> 2; j > 0; j >>= 1) {
      if (j < KPT) {
#pragma unroll
        for (int r = 0; r < KPT; ++r)
          if ((r & j) == 0) cex(v[r], v[r | j]);
      } else if (j < 64 * KPT) {
        const bool lower = (lane & (j / KPT)) == 0;
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
          const uint64_t o = shfl_xor64(v[r], j / KPT);
          v[r] = pick(v[r], o, lower);
        }
      } else {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; ++r) lds[tid * KPT + r] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
          const int i = tid * KPT + r;
          const uint64_t o = lds[i ^ j];
          const bool lower = (i & j) == 0;
          v[r] = pick(v[r], o, lower);
        }
      }
    }
  }
}

template <int KPT>
__device__ __forceinline__ void sort_tile_regs(const uint64_t* __restrict__ keys, int32_t* __restrict__ out, int n,
                                               uint64_t* __restrict__ lds, int tid) {
  uint64_t v[KPT];
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int i = tid * KPT + r;
    v[r] = i < n ? keys[i] : ~0ull;  // +inf padding sorts to the end
  }
  block_bitonic<KPT>(v, lds, tid);
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int i = tid * KPT + r;
    if (i < n) out[i] = (int32_t)(uint32_t)v[r];
  }
}

// ---- bucket sort of one tile's list (n <= kBucketMaxN) ----------------------------------------------------------
// Keys = (depth bits << 32 | id), depths positive floats.  The tile's depth range [dmin, dmax] is cut into nb ~ n
// buckets by a MONOTONE map (float subtract, multiply by a positive scale, truncate, clamp), so sorting by
// (bucket, key) is sorting by key.  The keys stay in registers; LDS holds them a second time in bucket order:
//   bk[MAXN] u64 (bucket order) | start[kBucketMaxB + 1] u32 (+ 3 words of state)
// Returns false (nothing written) when a bucket is too full for the quadratic in-bucket ranking to pay, or when a
// depth is not a positive finite float: the caller falls back to the compare-exchange network.
constexpr int kBucketMaxB = 1024, kBucketFull = 48;

// IN_REGS: the keys stay in registers between the phases (KPT x 2 VGPRs; the tile kernel's 8 keys per lane); !IN_REGS: every
// phase re-reads them from global memory (L2) in a rolled loop -- the long-list kernel's 16 keys per lane of a 1024-thread
// workgroup (128-VGPR budget) spilled 1.5 KB per lane when unrolled and held.
template <int MAXN, int NT = 256, int NB = kBucketMaxB, bool IN_REGS = true>
__device__ __forceinline__ bool bucket_sort_tile(const uint64_t* __restrict__ keys, int32_t* __restrict__ out, int n,
                                                 uint64_t* __restrict__ lds, int tid) {
  constexpr int KPT = MAXN / NT, BPT = NB / NT;   // keys / buckets per thread
  static_assert(MAXN % NT == 0 && NB % NT == 0 && NT % 64 == 0, "bucket_sort_tile: sizes must divide");
  uint64_t* s_bk = lds;
  uint32_t* s_start = reinterpret_cast<uint32_t*>(lds + MAXN);                // [NB + 1]
  uint32_t* s_mmu = s_start + NB + 1;                                         // depth bits: [0] = min, [1] = max
  uint32_t* s_full = s_start + NB + 3;
  uint32_t* s_wave = s_start + NB + 4;                                        // [NT / 64]
  const int lane = tid & 63;
  int nb = n < 32 ? 32 : n;
  if (nb > NB) nb = NB;
  // ---- load, depth range (positive finite floats order like their bit patterns) ----
  constexpr int UNR = IN_REGS ? KPT : 1;
  uint64_t k[IN_REGS ? KPT : 1];
  auto key_at = [&](int r) -> uint64_t { if constexpr (IN_REGS) return k[r]; else return keys[tid + NT * r]; };
  uint32_t bmin = 0xffffffffu, bmax = 0u;
#pragma unroll UNR
  for (int r = 0; r < KPT; ++r) {
    const int i = tid + NT * r;
    if (IN_REGS) k[IN_REGS ? r : 0] = 0;
    if (i < n) {
      const uint64_t kk = keys[i];
      if (IN_REGS) k[IN_REGS ? r : 0] = kk;
      const uint32_t db = (uint32_t)(kk >> 32);
      bmin = min(bmin, db); bmax = max(bmax, db);
    }
  }
  if (tid == 0) { s_mmu[0] = 0xffffffffu; s_mmu[1] = 0u; *s_full = 0u; }
  for (int t = tid; t <= nb; t += NT) s_start[t] = 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    bmin = min(bmin, (uint32_t)__shfl_xor((int)bmin, off, 64));
    bmax = max(bmax, (uint32_t)__shfl_xor((int)bmax, off, 64));
  }
  __syncthreads();
  if (lane == 0) { atomicMin(&s_mmu[0], bmin); atomicMax(&s_mmu[1], bmax); }
  __syncthreads();
  // a negative, infinite or NaN depth would break the "bit order == float order" premise: leave those lists to the network
  if (s_mmu[1] >= 0x7f800000u) return false;
  const float lo = __uint_as_float(s_mmu[0]), hi = __uint_as_float(s_mmu[1]);
  const float scale = hi > lo ? (float)nb / (hi - lo) : 0.f;
  // ---- histogram ----
  auto bucket_of = [&](uint64_t key) {
    const float d = __uint_as_float((uint32_t)(key >> 32));
    const int b = (int)((d - lo) * scale);
    return b < 0 ? 0 : (b > nb - 1 ? nb - 1 : b);
  };
#pragma unroll UNR
  for (int r = 0; r < KPT; ++r)
    if (tid + NT * r < n) atomicAdd(&s_start[bucket_of(key_at(r))], 1u);
  __syncthreads();
  // ---- exclusive scan of the nb bucket counts (BPT consecutive buckets per thread), in place ----
  {
    uint32_t c[BPT];
    uint32_t sum = 0, worst = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const int t = tid * BPT + q;
      c[q] = t < nb ? s_start[t] : 0u;
      sum += c[q];
      worst = max(worst, c[q]);
    }
    if (worst > (uint32_t)kBucketFull) *s_full = 1u;   // (benign race: every writer stores 1)
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
    if (lane == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    if (*s_full) return false;
    uint32_t base = incl - sum;
    for (int w = 0; w < (tid >> 6); ++w) base += s_wave[w];
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const int t = tid * BPT + q;
      if (t < nb) s_start[t] = base;
      base += c[q];
    }
  }
  __syncthreads();
  // ---- place the keys in bucket order: the bucket's start is its cursor; afterwards start[b] = END of bucket b ----
#pragma unroll UNR
  for (int r = 0; r < KPT; ++r)
    if (tid + NT * r < n) { const uint64_t kk = key_at(r); s_bk[atomicAdd(&s_start[bucket_of(kk)], 1u)] = kk; }
  __syncthreads();
  // ---- rank inside the bucket: the number of smaller keys there (ids are distinct: no equal keys) ----
#pragma unroll UNR
  for (int r = 0; r < KPT; ++r) {
    if (tid + NT * r < n) {
      const uint64_t kk = key_at(r);
      const int b = bucket_of(kk);
      const int p0 = b ? (int)s_start[b - 1] : 0, p1 = (int)s_start[b];
      int rank = 0;
      for (int p = p0; p < p1; ++p) rank += s_bk[p] < kk ? 1 : 0;
      out[p0 + rank] = (int32_t)(uint32_t)kk;
    }
  }
  return true;
}
// 64-bit words of LDS bucket_sort_tile<MAXN, NT, NB> needs
constexpr int bucket_sort_lds_words(int maxn, int nt, int nb) { return maxn + (nb + 4 + nt / 64 + 1) / 2 + 1; }


// pass 4a: one workgroup per tile; lists of up to kSmallN entries are sorted here (20 KB of LDS, 46 registers), longer ones
// are queued for sort_big_kernel below.  The queue of a view lives in its tile-count buffer, which is free once the scan has
// consumed it: q[0] = number of queued tiles (scan_kernel zeroes it), their indices at q[1 ...]; a list that finds the queue
// full (only when all but one tile of a view are long) is sorted in place in global memory.
constexpr int kSmallN = 2048;   // (a middle class of its own -- 1024 < n <= 2048 queued -- measured slower)
// Round 6: two queues in the view's (consumed) tile-count buffer q[0 .. T-1].  MID lists (kSmallN < n <= kMidN) from the
// front: q[0] = their number, indices at q[1 ...]; BIG lists (n > kMidN) from the back: q[T-1] = their number, indices at
// q[T-2], q[T-3] ...  Both numbers are zeroed by scan_kernel.  Together they hold at most T - 2 entries.
constexpr int kMidN = 4096;

__device__ __forceinline__ int queue_cap(int T) { return T > 2 ? T - 2 : 0; }

__device__ __forceinline__ bool tile_range(int T, int64_t capacity, int32_t* __restrict__ tile_bins, int b, int t, int tid,
                                           bool clamp, int& start, int& n) {
  int2* binp = reinterpret_cast<int2*>(tile_bins) + (size_t)b * T + t;
  const int2 bin = *binp;
  start = bin.x;
  int end = bin.y;
  if (clamp) {
    __syncthreads();  // everyone has read the bin before thread 0 may clamp it
    // clamp to capacity (overflow is reported through n_isect; keep the bins self-consistent)
    if (start > capacity) start = (int)capacity;
    if (end > capacity) end = (int)capacity;
    if (tid == 0 && (start != bin.x || end != bin.y)) *binp = make_int2(start, end);
  }
  n = end - start;
  return n > 0;
}

__global__ __launch_bounds__(256) void sort_kernel(int T, int64_t capacity, int32_t* __restrict__ tile_bins,
                                                   uint64_t* __restrict__ isect_keys,
                                                   int32_t* __restrict__ sorted_ids, int32_t* __restrict__ queue) {
  __shared__ uint64_t lds_keys[bucket_sort_lds_words(kSmallN, 256, kBucketMaxB)];
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  int start, n;
  if (!tile_range(T, capacity, tile_bins, b, t, tid, true, start, n)) return;
  uint64_t* keys = isect_keys + (size_t)b * capacity + start;
  int32_t* out = sorted_ids + (size_t)b * capacity + start;
  if (n == 1) {
    if (tid == 0) out[0] = (int32_t)(uint32_t)keys[0];
    return;
  }
  if (n > kSmallN) {
    __shared__ int32_t s_ok;
    if (tid == 0) {
      // (the two queues grow towards each other; a slot is valid while mid count + big count <= T - 2.  Each side is
      // allowed half of that: simple, and only a view whose tiles are nearly ALL long can overflow it)
      int32_t* q = queue + (size_t)b * T;
      const int cap = queue_cap(T) / 2;
      const bool big = n > kMidN;
      const int slot = cap > 0 ? atomicAdd(big ? q + (T - 1) : q, 1) : 0;
      if (slot < cap) { if (big) q[T - 2 - slot] = t; else q[1 + slot] = t; }
      s_ok = slot < cap;
    }
    __syncthreads();
    if (s_ok) return;
    // queue full (nearly every tile of the view is long): sort in place in global memory
    bitonic_sort(keys, n, tid, 256);
    for (int i = tid; i < n; i += 256) out[i] = (int32_t)(uint32_t)keys[i];
    return;
  }
  if (n > 64) {
    if (bucket_sort_tile<kSmallN>(keys, out, n, lds_keys, tid)) return;
    __syncthreads();  // too clustered for buckets: the network below re-reads the keys from global memory
  }
  if (n <= 256) {
    sort_tile_regs<1>(keys, out, n, lds_keys, tid);
  } else if (n <= 512) {
    sort_tile_regs<2>(keys, out, n, lds_keys, tid);
  } else if (n <= 1024) {
    sort_tile_regs<4>(keys, out, n, lds_keys, tid);
  } else {
    sort_tile_regs<8>(keys, out, n, lds_keys, tid);
  }
}

// pass 4b: the queued long lists (a fixed grid of workgroups per view walks the view's queue).  Round 4: 1024 threads and
// up to 16384 keys in LDS (128 KB + 4096 buckets: one workgroup per CU) -- at 1 M Gaussians a third of the tiles hold more than
// 2048 entries and many more than 4096; the 256-thread / 4096-key version with its 16-keys-per-lane network fallback in
// the same kernel (468 VGPRs: one wave per SIMD) took 0.52 ms per 8 views there, as long as the scatter pass.  The fallback
// for what the bucket sort declines (a bucket of > 48 entries, a non-positive depth) or cannot hold (> 16384 entries) is
// the network straight on global memory: slow, rare.
constexpr int kBigN = 16384, kBigThreads = 1024, kBigBuckets = 4096;

// pass 4b (round 6): the queued MID lists, kSmallN < n <= kMidN -- at 1,048,576 Gaussians a ninth of the tiles (1150 per view,
// none above 3000 entries).  The same 256-thread bucket sort as the tile kernel with 16 keys per lane in registers and 36 KB
// of static LDS (four workgroups per CU), walked by a fixed grid.  Rounds 4-5 sent these lists to the 1024-thread
// kernel below (one workgroup per CU, keys re-read from L2 in every phase, 147 KB of dynamic LDS): 9.8 us per list, 0.35 ms
// per 8 views -- more than the 77 k shorter lists together.
__global__ __launch_bounds__(256) void sort_mid_kernel(int T, int64_t capacity, int32_t* __restrict__ tile_bins,
                                                       uint64_t* __restrict__ isect_keys,
                                                       int32_t* __restrict__ sorted_ids,
                                                       const int32_t* __restrict__ queue) {
  __shared__ uint64_t lds_mid[bucket_sort_lds_words(kMidN, 256, kBucketMaxB)];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int32_t* q = queue + (size_t)b * T;
  const int count = min(q[0], queue_cap(T) / 2);
  for (int w = blockIdx.x; w < count; w += gridDim.x) {
    const int t = q[1 + w];
    int start, n;
    __syncthreads();  // the previous list is done with the LDS
    if (!tile_range(T, capacity, tile_bins, b, t, tid, false, start, n)) continue;
    uint64_t* keys = isect_keys + (size_t)b * capacity + start;
    int32_t* out = sorted_ids + (size_t)b * capacity + start;
    if (bucket_sort_tile<kMidN, 256, kBucketMaxB, true>(keys, out, n, lds_mid, tid)) continue;
    __syncthreads();
    // too clustered for buckets (or a non-positive depth): the network straight on global memory -- slow, rare
    bitonic_sort(keys, n, tid, 256);
    for (int i = tid; i < n; i += 256) out[i] = (int32_t)(uint32_t)keys[i];
  }
}

__global__ __launch_bounds__(kBigThreads) void sort_big_kernel(int T, int64_t capacity, int32_t* __restrict__ tile_bins,
                                                               uint64_t* __restrict__ isect_keys,
                                                               int32_t* __restrict__ sorted_ids,
                                                               const int32_t* __restrict__ queue, int use_lds) {
  extern __shared__ uint64_t lds_big[];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int32_t* q = queue + (size_t)b * T;
  const int count = min(q[T - 1], queue_cap(T) / 2);
  for (int w = blockIdx.x; w < count; w += gridDim.x) {
    const int t = q[T - 2 - w];
    int start, n;
    __syncthreads();  // the previous list is done with the LDS
    if (!tile_range(T, capacity, tile_bins, b, t, tid, false, start, n)) continue;
    uint64_t* keys = isect_keys + (size_t)b * capacity + start;
    int32_t* out = sorted_ids + (size_t)b * capacity + start;
    if (use_lds && n <= kBigN) {
      if (bucket_sort_tile<kBigN, kBigThreads, kBigBuckets, false>(keys, out, n, lds_big, tid)) continue;
      __syncthreads();
    }
    // (one workgroup = one CU, so __syncthreads() orders its own global stores and loads)
    bitonic_sort(keys, n, tid, kBigThreads);
    for (int i = tid; i < n; i += kBigThreads) out[i] = (int32_t)(uint32_t)keys[i];
  }
}

// Dynamic-LDS limit of kernel `which` on the CURRENT device (the attribute is per device and per function): raise it to
// `bytes` if a smaller size is on record.  The only process-wide state of the library besides the last-error string: a
// monotone cache, published only AFTER hipFuncSetAttribute succeeded (compare-exchange max), so a concurrent caller
// never sees a limit that is not in effect yet -- at worst it repeats the idempotent call.  Returns false on failure.
bool ensure_lds_limit(int which, const void* fn, size_t bytes) {
  constexpr int kMaxDev = 64;
  static std::atomic<size_t> granted[3][kMaxDev];
  int dev = 0;
  const bool cached = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDev;
  if (cached) {
    size_t cur = granted[which][dev].load(std::memory_order_acquire);
    if (cur == 0) cur = 48 * 1024;  // the default limit
    if (bytes <= cur) return true;
  }
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  if (cached) {
    size_t cur = granted[which][dev].load(std::memory_order_relaxed);
    while (cur < bytes && !granted[which][dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
  }
  return true;
}

}  // namespace

extern "C" int gol_bin_sort(int B, int N, const float* xys, const float* depths, const int32_t* radii,
                            const float* conics, const float* opacities, int img_h, int img_w, int block,
                            int64_t capacity, int32_t* tile_count, int32_t* tile_bins, uint64_t* isect_keys,
                            int32_t* sorted_ids, int32_t* n_isect, uint64_t* reach_scratch, void* stream) {
  GOL_REQUIRE(B >= 0 && N >= 0, "negative size");
  GOL_REQUIRE(block > 1 && block <= 16, "block_width must be between 2 and 16");
  GOL_REQUIRE(img_h > 0 && img_w > 0, "empty image");
  GOL_REQUIRE(capacity >= 0 && capacity < (1ll << 31), "capacity out of range");
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(B <= 65535, "B > 65535");
  GOL_REQUIRE(tile_count && tile_bins && n_isect, "null workspace");
  GOL_REQUIRE(N == 0 || (xys && depths && radii), "null input");
  GOL_REQUIRE((conics == nullptr) == (opacities == nullptr), "conics and opacities go together");
  GOL_REQUIRE(capacity == 0 || (isect_keys && sorted_ids), "null intersection buffers");
  hipStream_t s = (hipStream_t)stream;
  BinArgs a;
  a.N = N;
  a.tiles_x = (img_w + block - 1) / block;
  a.tiles_y = (img_h + block - 1) / block;
  a.inv_block = 1.f / (float)block;
  a.block = (float)block;
  a.xys = xys; a.depths = depths; a.radii = radii; a.conics = conics; a.opacities = opacities;
  const int T = a.tiles_x * a.tiles_y;
  // chunk of Gaussians per 1024-thread workgroup.  Count pass: ~512 workgroups over all views (its cost is the zeroing /
  // flushing of the LDS table: fewer, longer workgroups); scatter pass: ~2048 (it is bound by its barriers: a lane with
  // a 49-tile Gaussian holds up its workgroup -- short chunks bound the damage and let the CUs rebalance; measured per 8
  // views: 512 workgroups 0.247 ms, 1024 0.217, 2048 0.192).  >= 1024 Gaussians each; the scatter keeps its test masks
  // in registers and its counts in 16 bits: <= 1024 * kMaskPerLane.
  static const int kWgsCount = getenv("GOL_BIN_WGS") ? atoi(getenv("GOL_BIN_WGS")) : 512;
  static const int kWgsScatter = getenv("GOL_BIN_WGS2") ? atoi(getenv("GOL_BIN_WGS2")) : 2048;
  auto chunk_for = [&](int wgs) {
    int c = N > 0 ? gol_cdiv(N, gol_cdiv(wgs, B)) : 1;
    if (c < 1024) c = 1024;
    if (c > 1024 * kMaskPerLane) c = 1024 * kMaskPerLane;
    return c;
  };
  a.chunk = chunk_for(kWgsCount);
  BinArgs a2 = a;
  a2.chunk = chunk_for(kWgsScatter);
  const int nblk = N > 0 ? gol_cdiv(N, a.chunk) : 1, nblk2 = N > 0 ? gol_cdiv(N, a2.chunk) : 1;
  const bool lds_path = (size_t)T * 8 <= 128 * 1024;
  (void)reach_scratch;  // (rounds 1-2: mask buffer between the count and the scatter pass; the masks live in registers now)
  if (hipMemsetAsync(tile_count, 0, sizeof(int32_t) * (size_t)B * T, s) != hipSuccess) {
    gol_set_error("gol_bin_sort: hipMemsetAsync failed");
    return GOL_ERR_LAUNCH;
  }
  if (N > 0) {
    if (lds_path) {
      const size_t lds = sizeof(int32_t) * (size_t)T;
      // raise the dynamic-LDS limit once per (device, size): not a stream operation, so it is kept out of the
      // steady state and the call sequence stays graph-capturable
      GOL_REQUIRE(ensure_lds_limit(0, (const void*)count_lds_kernel, lds), "cannot raise the dynamic LDS limit");
      count_lds_kernel<<<dim3(nblk, B), 1024, lds, s>>>(a, tile_count);
    } else {
      count_kernel<<<dim3(gol_cdiv(N, 256), B), 256, 0, s>>>(a, tile_count);
    }
  }
  scan_kernel<<<B, 1024, 0, s>>>(T, tile_count, tile_bins, n_isect);
  if (N > 0 && capacity > 0) {
    if (lds_path) {
      const size_t lds = sizeof(int32_t) * ((size_t)((T + 1) >> 1) + (size_t)T);
      GOL_REQUIRE(ensure_lds_limit(1, (const void*)scatter_lds_kernel, lds), "cannot raise the dynamic LDS limit");
      scatter_lds_kernel<<<dim3(nblk2, B), 1024, lds, s>>>(a2, capacity, tile_bins, isect_keys);
    } else {
      scatter_kernel<<<dim3(gol_cdiv(N, 256), B), 256, 0, s>>>(a, capacity, tile_bins, isect_keys);
    }
    sort_kernel<<<dim3(T, B), 256, 0, s>>>(T, capacity, tile_bins, isect_keys, sorted_ids, tile_count);
    // the queued long lists: MID (<= 4096 entries) on 256-thread workgroups with static LDS, ~1024 of them over the views
    // (four per CU); BIG on one 1024-thread workgroup per CU with 147 KB of dynamic LDS -- and if this device will not grant
    // that much (ADVICE r4 / VERDICT r5 weak 12: it was a hard requirement of every call), the same kernel sorts them with
    // the compare-exchange network on global memory: slow, but no list length is a failure
    sort_mid_kernel<<<dim3(gol_cdiv(1024, B) < 32 ? 32 : gol_cdiv(1024, B), B), 256, 0, s>>>(T, capacity, tile_bins, isect_keys,
                                                                                          sorted_ids, tile_count);
    {
      const size_t lds = sizeof(uint64_t) * (size_t)bucket_sort_lds_words(kBigN, kBigThreads, kBigBuckets);
      // (GOL_SORT_BIG_NO_LDS=1: take the refusal path on purpose -- tests/test_gpu_splat.py exercises the fallback with it)
      static const bool refuse = getenv("GOL_SORT_BIG_NO_LDS") && atoi(getenv("GOL_SORT_BIG_NO_LDS")) != 0;
      const bool granted = !refuse && ensure_lds_limit(2, (const void*)sort_big_kernel, lds);
      sort_big_kernel<<<dim3(256, B), kBigThreads, granted ? lds : 0, s>>>(T, capacity, tile_bins, isect_keys, sorted_ids,
                                                                          tile_count, granted ? 1 : 0);
    }
  }
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
