// mvp.hip -- Mixture-of-Volumetric-Primitives ray marcher (forward + backward), leaf/node AABBs and
// camera ray generation for gfx950.
//
// Replaces (/root/reference/extensions):
//   utils/utils_kernel.cu:11-51                         compute_raydirs_forward_kernel
//   mvpraymarch/bvh.cu:157-201, primtransf.h:12-63      compute_aabb_kernel (fixed-order tree)
//   mvpraymarch/mvpraymarch_subset_kernel.h:7-228       raymarch_subset_{forward,backward}_kernel with
//       PrimTransfSRT (primtransf.h:99-179), PrimSamplerTW<false, GridSamplerChlast> (primsampler.h,
//       utils.h:523-770), PrimAccumAdditive (primaccum.h:63-98), PrimSplatterTW (primsplatter.h),
//       RaySubsetFixedBVH<false,512,true> (utils.h:949-1045)
// The reference is written around 32-lane warps (8x4-pixel footprints, __any_sync / __shfl_down_sync
// with literal 32s).  This is a wave64 redesign:
//   * a wave owns an 8x8-pixel footprint and builds ONE hit list for it (in LDS) with a wave-synchronous, stackless
//     walk of the implicit heap: node AABBs and box transforms are wave-uniform (scalar loads), only the ray tests are
//     per lane.  The wave's two halves (lanes 0-31 = rows 0-3, lanes 32-63 = rows 4-7) are exactly two of the
//     reference's 8x4-pixel warps (its (8, 16) thread block, mvpraymarch.py:334); every list entry carries which of the
//     two keeps it, so that each half evaluates precisely the boxes the reference warp's own list holds -- the first 512
//     boxes, in DFS order, that any of ITS 32 rays hits (utils.h:993-1012) -- also in scenes that exceed that cap;
//   * the list is annotated with the range of march iterations in which ANY lane of the wave can be
//     inside the box.  The reference re-tests every hit box at every step (steps x boxes transform
//     evaluations); here a box costs work only while some ray of the wave is inside it (~20x fewer
//     evaluations at the BASELINE config) -- output-preserving, because a skipped evaluation is one
//     whose valid() test fails for every lane;
//   * backward: the 15 primitive-transform gradients are reduced over the wave four-at-a-time with
//     the v_permlane*_swap ladder (one atomic per wave per scalar, like the reference's
//     fastAtomicAdd per warp); template gradients are scattered with hardware f32 atomics.
#include "gol_common.h"

namespace {

constexpr int kTgSlots = 48;  // hit-list slots with an LDS accumulator for the transform gradients (backward)
constexpr int kMaxHits = 512;  // per 8x4-pixel half of the wave = per 32-lane warp of the reference (utils.h:993-1012)
// entries of the wave's merged list (the union of what its two halves keep: at most 2 x 512, in practice the halves -- four
// image rows apart -- keep nearly the same boxes).  896 entries x 8 bytes x 4 waves + the backward's gradient slots = 40 KB
// = four workgroups per CU, what the backward's registers allow anyway.  Entries the merged list has no room for are dropped
// for both halves (only in scenes beyond the reference's own cap whose halves differ by more than 384 boxes).
constexpr int kListCap = 896;
constexpr int kKeepA = 0x40000000, kKeepB = (int)0x80000000u, kBoxMask = 0x3fffffff;   // flags in the list entry: kept by half A (rows 0-3) / B

// iteration window [lo, hi] of a list entry, packed: lo clamped to [0, 65535] in the low half, hi in the high half with
// 65535 = unbounded; an empty window is (65535, 0).  Clamping only ever widens a window (a box evaluated in vain fails
// valid() for every lane).
__device__ __forceinline__ unsigned pack_window(int lo, int hi) {
  if (hi < lo || hi < 0) return 65535u;
  const unsigned l = (unsigned)min(max(lo, 0), 65535), h = (unsigned)min(hi, 65535);
  return l | (h << 16);
}
__device__ __forceinline__ bool window_active(unsigned w, int iter) {
  const int lo = (int)(w & 0xffffu), hi = (int)(w >> 16);
  return lo <= iter && (iter <= hi || hi == 65535);
}
// does this lane's half keep list entry e?
__device__ __forceinline__ bool half_keeps(int e, int lane) { return (e & (lane < 32 ? kKeepA : kKeepB)) != 0; }

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const float* __restrict__ p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float min3(V3 a) { return fminf(fminf(a.x, a.y), a.z); }
__device__ __forceinline__ float max3(V3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
__device__ __forceinline__ V3 vmin(V3 a, V3 b) { return V3{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
__device__ __forceinline__ V3 vmax(V3 a, V3 b) { return V3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }

// ---- compute_raydirs ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void raydirs_kernel(int N, int H, int W, const float* __restrict__ viewpos,
                                                      const float* __restrict__ viewrot,
                                                      const float* __restrict__ focal,
                                                      const float* __restrict__ princpt,
                                                      const float* __restrict__ pixelcoords, float volradius,
                                                      float* __restrict__ rayposim, float* __restrict__ raydirim,
                                                      float* __restrict__ tminmaxim) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  const int h = p / W, w = p - h * W;
  const size_t r = (size_t)n * H * W + p;
  const V3 rp = ld3(viewpos + 3 * n) * (1.f / volradius);
  const V3 v0 = ld3(viewrot + 9 * n), v1 = ld3(viewrot + 9 * n + 3), v2 = ld3(viewrot + 9 * n + 6);
  float px = (float)w, py = (float)h;
  if (pixelcoords) { px = pixelcoords[2 * r]; py = pixelcoords[2 * r + 1]; }
  const float u = (px - princpt[2 * n]) / focal[2 * n], v = (py - princpt[2 * n + 1]) / focal[2 * n + 1];
  V3 d = v0 * u + v1 * v + v2;
  d = d * (1.f / sqrtf(dot(d, d)));
  const V3 t1 = v3((-1.f - rp.x) / d.x, (-1.f - rp.y) / d.y, (-1.f - rp.z) / d.z);
  const V3 t2 = v3((1.f - rp.x) / d.x, (1.f - rp.y) / d.y, (1.f - rp.z) / d.z);
  const float tmin = max3(vmin(t1, t2)), tmax = min3(vmax(t1, t2));
  rayposim[3 * r] = rp.x; rayposim[3 * r + 1] = rp.y; rayposim[3 * r + 2] = rp.z;
  raydirim[3 * r] = d.x; raydirim[3 * r + 1] = d.y; raydirim[3 * r + 2] = d.z;
  *reinterpret_cast<float2*>(tminmaxim + 2 * r) = make_float2(fmaxf(tmin, 0.f), tmax);
}

// ---- AABBs of the fixed-order tree ------------------------------------------------------------------
// Leaves: one lane per primitive.  Inner nodes: level by level, bottom-up, inside ONE workgroup per
// view (the tree of a view is tiny: 2K-1 nodes), so no atomics, flags or cudaMalloc (cf. bvh.cu:261-293).
__global__ __launch_bounds__(1024) void aabb_kernel(int K, const float* __restrict__ primpos,
                                                    const float* __restrict__ primrot,
                                                    const float* __restrict__ primscale,
                                                    float* __restrict__ nodeaabb) {
  const int n = blockIdx.x;
  float* A = nodeaabb + (size_t)n * (2 * K - 1) * 6;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const size_t e = (size_t)n * K + k;
    const V3 pt = ld3(primpos + 3 * e), pr0 = ld3(primrot + 9 * e), pr1 = ld3(primrot + 9 * e + 3),
             pr2 = ld3(primrot + 9 * e + 6), ps = ld3(primscale + 3 * e);
    V3 mn = v3(INFINITY, INFINITY, INFINITY), mx = v3(-INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      V3 p = v3(((c & 1) ? 1.f : -1.f) / ps.x, ((c & 2) ? 1.f : -1.f) / ps.y, ((c & 4) ? 1.f : -1.f) / ps.z);
      p = v3(dot(p, pr0), dot(p, pr1), dot(p, pr2)) + pt;
      mn = vmin(mn, p); mx = vmax(mx, p);
    }
    float* a = A + (size_t)(K - 1 + k) * 6;
    a[0] = mn.x; a[1] = mn.y; a[2] = mn.z; a[3] = mx.x; a[4] = mx.y; a[5] = mx.z;
  }
  // heap levels bottom-up: nodes [lo, hi) of a level only depend on deeper levels
  int hi = K - 1;  // inner nodes are 0 .. K-2
  while (hi > 0) {
    __syncthreads();
    // deepest unfinished level: nodes whose children are both >= hi
    const int lo = (hi - 1 + 1) / 2;  // smallest node with 2*node+1 >= hi  <=>  node >= (hi-1)/2 rounded up
    for (int node = lo + threadIdx.x; node < hi; node += blockDim.x) {
      const float* l = A + (size_t)(2 * node + 1) * 6;
      const float* r = A + (size_t)(2 * node + 2) * 6;
      float* a = A + (size_t)node * 6;
#pragma unroll
      for (int c = 0; c < 3; ++c) { a[c] = fminf(l[c], r[c]); a[3 + c] = fmaxf(l[3 + c], r[3 + c]); }
    }
    hi = lo;
  }
}

// ---- shared march machinery ---------------------------------------------------------------------------
struct Xform { V3 xmt, pr0, pr1, pr2, rxmt, ps; };

__device__ __forceinline__ V3 xform_fwd(Xform& t, const float* __restrict__ primpos, const float* __restrict__ primrot,
                                        const float* __restrict__ primscale, int k, V3 x) {
  const V3 pt = ld3(primpos + 3 * k);
  t.pr0 = ld3(primrot + 9 * k); t.pr1 = ld3(primrot + 9 * k + 3); t.pr2 = ld3(primrot + 9 * k + 6);
  t.ps = ld3(primscale + 3 * k);
  t.xmt = x - pt;
  t.rxmt = t.pr0 * t.xmt.x + t.pr1 * t.xmt.y + t.pr2 * t.xmt.z;
  return t.rxmt * t.ps;
}

__device__ __forceinline__ bool valid_pos(V3 p) {
  return p.x > -1.f && p.x < 1.f && p.y > -1.f && p.y < 1.f && p.z > -1.f && p.z < 1.f;
}

// ray vs oriented unit box (utils.h:976-993): parametric entry/exit
__device__ __forceinline__ bool box_hit(const float* __restrict__ primpos, const float* __restrict__ primrot,
                                        const float* __restrict__ primscale, int k, V3 raypos, V3 raydir,
                                        float& trmin, float& trmax) {
  const V3 pt = ld3(primpos + 3 * k), pr0 = ld3(primrot + 9 * k), pr1 = ld3(primrot + 9 * k + 3),
           pr2 = ld3(primrot + 9 * k + 6), ps = ld3(primscale + 3 * k);
  const V3 xmt = raypos - pt;
  const V3 r0 = (pr0 * xmt.x + pr1 * xmt.y + pr2 * xmt.z) * ps;
  const V3 rd = (pr0 * raydir.x + pr1 * raydir.y + pr2 * raydir.z) * ps;
  const V3 ird = v3(1.f / rd.x, 1.f / rd.y, 1.f / rd.z);
  const V3 t0 = (v3(-1.f, -1.f, -1.f) - r0) * ird, t1 = (v3(1.f, 1.f, 1.f) - r0) * ird;
  trmin = max3(vmin(t0, t1));
  trmax = min3(vmax(t0, t1));
  return trmin <= trmax;
}

__device__ __forceinline__ bool aabb_hit(const float* __restrict__ a, V3 raypos, V3 ird) {
  const V3 t0 = (ld3(a) - raypos) * ird, t1 = (ld3(a + 3) - raypos) * ird;
  return max3(vmin(t0, t1)) <= min3(vmax(t0, t1));
}

// next node of a left-first DFS of the implicit heap after finishing the subtree of `node`
__device__ __forceinline__ int heap_next(int node) {
  while (node != 0 && !(node & 1)) node = (node - 1) >> 1;  // climb while we are a right child
  return node == 0 ? -1 : node + 1;                           // right sibling
}

__device__ __forceinline__ int sat_floor_to_int(float v) {
  const float f = floorf(v);
  if (!(f == f)) return 0;
  if (f >= 2147483520.f) return 2147483647;
  if (f <= -2147483648.f) return -2147483647 - 1;
  return (int)f;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return __builtin_amdgcn_readfirstlane(v);
}

struct Ray {
  V3 pos, dir;
  float t, rt1;
  bool live;
};

struct Tri { int idx[8]; float w[8]; float fx, fy, fz; int x0, y0, z0; bool allv; };

// utils.h:523-560 (align_corners=True coordinates, zero padding): corner indices (-1 = outside) + weights, for a
// position strictly inside the box (valid_pos: every sample the march evaluates).  Then (pos + 1) / 2 is in (0, 1) and
// the floor of the voxel coordinate in [0, W-1], so the lower corner always exists and the upper one of an axis is
// missing only when the coordinate rounded up to W-1 exactly: three compares decide all eight corners (the general
// form tested 6 bounds per corner).  Executed by every lane (the clamp keeps the integers defined for the positions of
// lanes that do not evaluate; their indices are never used).
__device__ __forceinline__ void tri_setup(Tri& q, int D, int H, int W, V3 pos) {
  const float ix = fmaxf(-100.f, fminf(100.f, (pos.x + 1.f) * 0.5f)) * (float)(W - 1);
  const float iy = fmaxf(-100.f, fminf(100.f, (pos.y + 1.f) * 0.5f)) * (float)(H - 1);
  const float iz = fmaxf(-100.f, fminf(100.f, (pos.z + 1.f) * 0.5f)) * (float)(D - 1);
  const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
  const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
  q.fx = ix - fx0; q.fy = iy - fy0; q.fz = iz - fz0; q.x0 = x0; q.y0 = y0; q.z0 = z0;
  const bool vx = x0 + 1 < W, vy = y0 + 1 < H, vz = z0 + 1 < D;
  q.allv = vx && vy && vz;  // all eight corners exist (always, but for a coordinate that rounded up to the last voxel)
  const int sy = W, sz = H * W;
  const int b = z0 * sz + y0 * sy + x0;
  q.idx[0] = b;
  q.idx[1] = vx ? b + 1 : -1;
  q.idx[2] = vy ? b + sy : -1;
  q.idx[3] = (vx && vy) ? b + sy + 1 : -1;
  q.idx[4] = vz ? b + sz : -1;
  q.idx[5] = (vx && vz) ? b + sz + 1 : -1;
  q.idx[6] = (vy && vz) ? b + sz + sy : -1;
  q.idx[7] = (vx && vy && vz) ? b + sz + sy + 1 : -1;
  const float gx = 1.f - q.fx, gy = 1.f - q.fy, gz = 1.f - q.fz;
  const float w00 = gx * gy, w10 = q.fx * gy, w01 = gx * q.fy, w11 = q.fx * q.fy;
  q.w[0] = w00 * gz; q.w[1] = w10 * gz; q.w[2] = w01 * gz; q.w[3] = w11 * gz;
  q.w[4] = w00 * q.fz; q.w[5] = w10 * q.fz; q.w[6] = w01 * q.fz; q.w[7] = w11 * q.fz;
}

// The same for an ARBITRARY position (warp fields, algo 1: the template is sampled at the warped position, which may
// leave the box): every corner tested against all six bounds (utils.h:523-560).
__device__ __forceinline__ void tri_setup_any(Tri& q, int D, int H, int W, V3 pos) {
  const float ix = fmaxf(-100.f, fminf(100.f, (pos.x + 1.f) * 0.5f)) * (float)(W - 1);
  const float iy = fmaxf(-100.f, fminf(100.f, (pos.y + 1.f) * 0.5f)) * (float)(H - 1);
  const float iz = fmaxf(-100.f, fminf(100.f, (pos.z + 1.f) * 0.5f)) * (float)(D - 1);
  const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
  const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
  q.fx = ix - fx0; q.fy = iy - fy0; q.fz = iz - fz0; q.x0 = x0; q.y0 = y0; q.z0 = z0;
  const bool vx[2] = {x0 >= 0 && x0 < W, x0 + 1 >= 0 && x0 + 1 < W};
  const bool vy[2] = {y0 >= 0 && y0 < H, y0 + 1 >= 0 && y0 + 1 < H};
  const bool vz[2] = {z0 >= 0 && z0 < D, z0 + 1 >= 0 && z0 + 1 < D};
  const int sy = W, sz = H * W;
  const int b = z0 * sz + y0 * sy + x0;
  q.allv = true;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const bool v = vx[c & 1] && vy[(c >> 1) & 1] && vz[(c >> 2) & 1];
    q.idx[c] = v ? b + (c & 1) + ((c >> 1) & 1) * sy + ((c >> 2) & 1) * sz : -1;
    q.allv = q.allv && v;
  }
  const float gx = 1.f - q.fx, gy = 1.f - q.fy, gz = 1.f - q.fz;
  const float w00 = gx * gy, w10 = q.fx * gy, w01 = gx * q.fy, w11 = q.fx * q.fy;
  q.w[0] = w00 * gz; q.w[1] = w10 * gz; q.w[2] = w01 * gz; q.w[3] = w11 * gz;
  q.w[4] = w00 * q.fz; q.w[5] = w10 * q.fz; q.w[6] = w01 * q.fz; q.w[7] = w11 * q.fz;
}
__device__ __forceinline__ bool tri_any(const Tri& q) {
  bool v = false;
#pragma unroll
  for (int c = 0; c < 8; ++c) v = v || q.idx[c] >= 0;
  return v;
}

// warped sample position y1 = trilinear(warp field of the box [WD,WH,WW,3], y0) for y0 strictly inside the box
// (primsampler.h:53-56)
__device__ __forceinline__ V3 warp_sample(const float* __restrict__ wp, int WD, int WH, int WW, V3 y0, Tri& qw) {
  tri_setup(qw, WD, WH, WW, y0);
  V3 r = v3(0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (qw.idx[c] >= 0) {
      const float* v = gol_at(wp, (unsigned)qw.idx[c] * 12u);
      r.x += v[0] * qw.w[c]; r.y += v[1] * qw.w[c]; r.z += v[2] * qw.w[c];
    }
  }
  return r;
}

// Builds the wave's hit list (LDS) + per-box iteration windows, and positions the ray at its first
// sample.  Returns the number of hit boxes (wave-uniform).
__device__ __forceinline__ int build_hits(int K, const float* __restrict__ nodeaabb, const float* __restrict__ primpos,
                                          const float* __restrict__ primrot, const float* __restrict__ primscale,
                                          float stepsize, float tmin, float tmax, Ray& ray, int* __restrict__ s_list,
                                          unsigned* __restrict__ s_win) {
  const V3 ird = v3(1.f / ray.dir.x, 1.f / ray.dir.y, 1.f / ray.dir.z);
  float rt0 = INFINITY, rt1 = -INFINITY;
  int num = 0, num_a = 0, num_b = 0;   // merged list / boxes kept by half A / by half B
  int node = K > 1 ? 1 : 0;  // the reference never tests the root box itself (utils.h:1016-1021)
  if (K == 1) node = 0;
  while (node != -1) {
    if (node >= K - 1) {
      const int k = node - (K - 1);
      float a, b;
      const bool hit = ray.live && box_hit(primpos, primrot, primscale, k, ray.pos, ray.dir, a, b);
      if (hit) { rt0 = fminf(rt0, a); rt1 = fmaxf(rt1, b); }
      // (the reference: hit = __any_sync over the 32-lane warp, appended while that warp's list has < 512 entries)
      const unsigned long long hb = gol_ballot(hit);
      const bool keep_a = (hb & 0xffffffffull) != 0ull && num_a < kMaxHits;
      const bool keep_b = (hb >> 32) != 0ull && num_b < kMaxHits;
      num_a += keep_a ? 1 : 0;
      num_b += keep_b ? 1 : 0;
      if ((keep_a || keep_b) && num < kListCap) {
        if ((threadIdx.x & 63) == 0) s_list[num] = k | (keep_a ? kKeepA : 0) | (keep_b ? kKeepB : 0);
        ++num;
      }
      node = heap_next(node);
    } else {
      const bool hit = ray.live && aabb_hit(nodeaabb + (size_t)node * 6, ray.pos, ird);
      node = (gol_ballot(hit) != 0ull) ? 2 * node + 1 : heap_next(node);
    }
  }
  __builtin_amdgcn_wave_barrier();  // list written by lane 0, read by the whole wave below
  rt0 = fmaxf(rt0, tmin);
  ray.rt1 = fminf(rt1, tmax);
  // first sample: whole steps from tmin up to just before the first box (subset_kernel.h:72-77)
  ray.t = tmin;
  ray.pos = ray.pos + ray.dir * tmin;
  const int incs = sat_floor_to_int((rt0 - ray.t) / stepsize);
  ray.t += (float)incs * stepsize;
  ray.pos = ray.pos + ray.dir * (float)incs * stepsize;
  // iteration windows: iteration i samples t_i ~ ray.t + i*stepsize; a lane can be inside box s only for t in [a, b]
  // -> i in [floor((a - t)/step - m), ceil((b - t)/step + m)].  The margin m (in steps) covers what separates the exact
  // line used here from the marched position: t and pos are accumulated by repeated float additions (error <= n ulp
  // after n steps: 1.2e-7 |pos| n / step ~ 2e-5 n steps at the BASELINE step size) plus the rounding of box_hit itself;
  // m = 0.05 + drift n with drift >= 1e-4 scaled by the ray's own |pos| / stepsize (below) leaves a factor > 4.  (A full step of margin on either side, as before, evaluated every box for
  // ~2 of ~11 iterations in vain.)  Window of the wave = union over its lanes.
  const float inv_step = 1.f / stepsize;
  const V3 p0 = ray.pos - ray.dir * ray.t;  // ray origin again (positions are affine in t)
  // drift per step, in steps: ray.pos and ray.t are accumulated by float additions, each off by <= ulp/2 of the LARGEST
  // coordinate involved, i.e. <= 1.2e-7 max|pos| / stepsize steps per step (x4 for the three coordinates + t): 1e-4 at
  // the BASELINE geometry (|pos| ~ 3, step 1/64), and growing with |pos| / stepsize -- world-space units, a small dt
  const V3 pe = ray.pos + ray.dir * (ray.rt1 - ray.t);
  const float pmax = fmaxf(fmaxf(fmaxf(fabsf(ray.pos.x), fabsf(ray.pos.y)), fmaxf(fabsf(ray.pos.z), fabsf(pe.x))),
                           fmaxf(fmaxf(fabsf(pe.y), fabsf(pe.z)), fmaxf(fabsf(ray.t), fabsf(ray.rt1))));
  const float drift = fmaxf(1e-4f, 4.8e-7f * pmax * inv_step);
  for (int s = 0; s < num; ++s) {
    const int e = __builtin_amdgcn_readfirstlane(s_list[s]);
    const int k = e & kBoxMask;
    float a, b;
    int lo = 2147483647, hi = -2147483647;
    if (ray.live && half_keeps(e, threadIdx.x & 63) && box_hit(primpos, primrot, primscale, k, p0, ray.dir, a, b)) {
      const float xa = (a - ray.t) * inv_step, xb = (b - ray.t) * inv_step;
      const float m = 0.05f + drift * fmaxf(fabsf(xa), fabsf(xb));
      const float fl = floorf(xa - m), fh = ceilf(xb + m);
      if (fl < 2.0e9f && fh > -2.0e9f) {
        lo = (int)fmaxf(fl, -2.0e9f);
        hi = (int)fminf(fh, 2.0e9f);
      }
    }
    lo = wave_min_i(lo);
    hi = wave_max_i(hi);
    if ((threadIdx.x & 63) == 0) s_win[s] = pack_window(lo, hi);
  }
  __builtin_amdgcn_wave_barrier();
  return num;
}

struct MarchArgs {
  int N, H, W, K, TD, TH, TW;
  float stepsize, fadescale, fadeexp;
  const float* raypos; const float* raydir; const float* tminmax; const float* nodeaabb;
  const float* primpos; const float* primrot; const float* primscale; const float* tplate;
  int group = 1;          // `group` consecutive ray images share one primitive set / template (light-batched shadow march)
  int alpha_only = 0;     // template has ONE channel (alpha); colour channels read as 0
  const float* warp = nullptr;  // algo 1: per-box warp fields [N,K,WD,WH,WW,3] (kernels instantiated with WARP only)
  int WD = 0, WH = 0, WW = 0;
};

__device__ __forceinline__ bool load_ray(const MarchArgs& a, int n, Ray& ray, float& tmin, float& tmax, size_t& r,
                                         int& wave) {
  const int tid = threadIdx.x, lane = tid & 63;
  wave = tid >> 6;
  const int lx = ((wave & 1) << 3) + (lane & 7), ly = ((wave >> 1) << 3) + (lane >> 3);
  const int w = blockIdx.x * 16 + lx, h = blockIdx.y * 16 + ly;
  ray.live = (w < a.W) && (h < a.H);
  r = ((size_t)n * a.H + min(h, a.H - 1)) * a.W + min(w, a.W - 1);
  ray.pos = ld3(a.raypos + 3 * r);
  ray.dir = ld3(a.raydir + 3 * r);
  const float2 tm = *reinterpret_cast<const float2*>(a.tminmax + 2 * r);
  tmin = tm.x; tmax = tm.y;
  return ray.live;
}


// |y|^e and |y|^(e-1) of the fade term exp(-fadescale * sum |y_i|^fadeexp) (primsampler.h:52-56).  The reference's models
// use fadeexp = 8 (hand_mvp.py / render_raymarcher.py): three squarings; any other exponent is exp2(e * log2|y|) on the
// transcendental unit, which is what CUDA's __powf (the reference's -use_fast_math build) computes -- HIP's __powf
// expands to the ~180-instruction double-float pow.  log2(0) = -inf gives 0 for e > 0.  `e8` is kernel-uniform.
__device__ __forceinline__ float fast_pow(float ax, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(ax)); }
__device__ __forceinline__ float fade_pow(float ax, float e, bool e8) {
  if (e8) { const float x2 = ax * ax, x4 = x2 * x2; return x4 * x4; }
  return fast_pow(ax, e);
}
__device__ __forceinline__ float fade_pow_m1(float ax, float e, bool e8) {
  if (e8) { const float x2 = ax * ax, x4 = x2 * x2; return x4 * x2 * ax; }
  return fast_pow(ax, e - 1.f);
}

template <bool SHADOW, bool WARP>
__global__ __launch_bounds__(256) void march_fwd_kernel(MarchArgs a, float* __restrict__ rayrgba,
                                                        float* __restrict__ raysat, float* __restrict__ shadow) {
  __shared__ int s_list[4][kListCap];
  __shared__ unsigned s_win[4][kListCap];
  const int n = blockIdx.z;
  const int lane = threadIdx.x & 63;
  Ray ray; float tmin, tmax; size_t r; int wave;
  load_ray(a, n, ray, tmin, tmax, r, wave);
  const size_t vox = (size_t)a.TD * a.TH * a.TW;
  const int pn = n / a.group;  // primitive set of this ray image
  const float* primpos = a.primpos + (size_t)pn * a.K * 3;
  const float* primrot = a.primrot + (size_t)pn * a.K * 9;
  const float* primscale = a.primscale + (size_t)pn * a.K * 3;
  const float4* tplate = reinterpret_cast<const float4*>(a.tplate) + (size_t)pn * a.K * vox;
  const float* tplate_a = a.tplate + (size_t)pn * a.K * vox;  // alpha-only layout
  float* shadow_n = SHADOW ? shadow + (size_t)n * a.K * vox * 2 : nullptr;
  const int num = build_hits(a.K, a.nodeaabb + (size_t)pn * (2 * a.K - 1) * 6, primpos, primrot, primscale, a.stepsize,
                             tmin, tmax, ray, s_list[wave], s_win[wave]);

  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, sat0 = -1.f, sat1 = -1.f, sat2 = -1.f;
  bool sat = false;
  int iter = 0;
  // !__all(t > rt1 + 1e-5 || done)   (subset_kernel.h:81)
  while (gol_ballot(ray.live && !(ray.t > ray.rt1 + 1e-5f || sat)) != 0ull) {
    // the boxes whose window contains this iteration, 64 list entries per ballot (lane j tests entry base + j), then a
    // scalar walk over the set bits in list order: the per-step cost follows the 2-3 boxes the wave is inside, not the
    // length of the hit list
    for (int base = 0; base < num; base += 64) {
    const int s_me = base + lane;
    unsigned long long active = gol_ballot(s_me < num && window_active(s_win[wave][s_me], iter));
    while (active) {
      const int s = base + __builtin_ctzll(active);
      active &= active - 1;
      const int e = __builtin_amdgcn_readfirstlane(s_list[wave][s]);  // wave-uniform: the box transform comes through scalar loads
      const int k = e & kBoxMask;
      Xform xf;
      const V3 y0 = xform_fwd(xf, primpos, primrot, primscale, k, ray.pos);
      // (half_keeps: a lane only evaluates boxes of ITS half's list; below the cap that is implied by valid_pos)
      const bool ev = ray.live && half_keeps(e, lane) && valid_pos(y0) && !sat && ray.t < ray.rt1 + 1e-5f;
      // shadow splat state of this lane (filled inside the branch, scattered wave-wide after it)
      int sh_idx[SHADOW ? 8 : 1], sh_key = -1;
      float sh_w[SHADOW ? 8 : 1], sh_vis = 0.f;
      Tri q;  // (every lane; only evaluating lanes use theirs)
      if (WARP) {
        V3 y1 = y0;
        Tri qw;
        if (ev) y1 = warp_sample(a.warp + ((size_t)pn * a.K + k) * (size_t)a.WD * a.WH * a.WW * 3, a.WD, a.WH, a.WW, y0, qw);
        tri_setup_any(q, a.TD, a.TH, a.TW, y1);
      } else {
        tri_setup(q, a.TD, a.TH, a.TW, y0);
      }
      const bool all_corners = gol_ballot(ev && !q.allv) == 0ull;
      if (ev) {
        const bool e8 = a.fadeexp == 8.f;
        const float fade = __expf(-a.fadescale * (fade_pow(fabsf(y0.x), a.fadeexp, e8) + fade_pow(fabsf(y0.y), a.fadeexp, e8) +
                                                  fade_pow(fabsf(y0.z), a.fadeexp, e8)));
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (a.alpha_only) {  // (kernel-uniform) shadow march: 4 bytes per voxel instead of 16
          const float* tp = tplate_a + (size_t)k * vox;
#pragma unroll
          for (int c = 0; c < 8; ++c)
            if (q.idx[c] >= 0) s3 += *gol_at(tp, (unsigned)q.idx[c] * 4u) * q.w[c];
        } else {
          const float4* tp = tplate + (size_t)k * vox;
          if (all_corners) {  // (wave-uniform) the common case: eight unguarded loads, no exec-mask regions
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const float4 v = *gol_at(tp, (unsigned)q.idx[c] * 16u);  // uniform box base + 32-bit byte offset
              s0 += v.x * q.w[c]; s1 += v.y * q.w[c]; s2 += v.z * q.w[c]; s3 += v.w * q.w[c];
            }
          } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              if (q.idx[c] >= 0) {
                const float4 v = *gol_at(tp, (unsigned)q.idx[c] * 16u);
                s0 += v.x * q.w[c]; s1 += v.y * q.w[c]; s2 += v.z * q.w[c]; s3 += v.w * q.w[c];
              }
            }
          }
        }
        s3 *= fade;
        if (SHADOW && (!WARP || tri_any(q))) {  // (a warped sample outside the template splats nothing: sh_key stays -1)
          sh_vis = 1.f - acc3;
          sh_key = ((q.z0 + 1) * (a.TH + 1) + (q.y0 + 1)) * (a.TW + 1) + (q.x0 + 1);
#pragma unroll
          for (int c = 0; c < 8; ++c) { sh_idx[c] = q.idx[c]; sh_w[c] = q.w[c]; }
        }
        // PrimAccumAdditive::forward_prim (primaccum.h:63-79)
        const float newalpha = acc3 + s3 * a.stepsize;
        const float contrib = fminf(newalpha, 1.f) - acc3;
        acc0 += s0 * contrib; acc1 += s1 * contrib; acc2 += s2 * contrib; acc3 += contrib;
        if (newalpha >= 1.f) {
          if (!sat) { sat0 = s0; sat1 = s1; sat2 = s2; }
          sat = true;
        }
      }
      if (SHADOW) {
        // PrimSplatterTW (primsplatter.h): (w * visibility, w) into the 8 corner voxels.  Like the template gradient of
        // the backward: group the wave's lanes by voxel cell, reduce each group, and let lanes 15|31 (corner c) and 47|63
        // (corner c+1) issue ONE pair of atomics per (cell, corner) -- memory-side float atomics per lane are the cost.
        float* sp = shadow_n + (size_t)k * vox * 2;
        unsigned long long todo = gol_ballot(WARP ? (ev && sh_key >= 0) : ev);
        while (todo) {
          const int leader = __builtin_ctzll(todo);
          const int key = __builtin_amdgcn_readlane(sh_key, leader);
          const bool mine = ev && (sh_key == key);
          todo &= ~gol_ballot(mine);
#pragma unroll
          for (int c = 0; c < 8; c += 2) {
            const int i0 = __builtin_amdgcn_readlane(sh_idx[c], leader), i1 = __builtin_amdgcn_readlane(sh_idx[c + 1], leader);
            if (i0 < 0 && i1 < 0) continue;
            const float w0 = mine ? sh_w[c] : 0.f, w1 = mine ? sh_w[c + 1] : 0.f;
            const float rs = gol_wave_sum4(w0 * sh_vis, w0, w1 * sh_vis, w1);  // lanes 15, 31 | 47, 63
            const int j = lane >> 4, idx = j < 2 ? i0 : i1;
            if ((lane & 15) == 15 && idx >= 0) atomicAdd(sp + (size_t)idx * 2 + (j & 1), rs);
          }
        }
      }
    }
    }
    ray.t += a.stepsize;
    ray.pos = ray.pos + ray.dir * a.stepsize;
    ++iter;
  }
  if (ray.live && rayrgba) {
    *reinterpret_cast<float4*>(rayrgba + 4 * r) = make_float4(acc0, acc1, acc2, acc3);
    if (raysat) { raysat[3 * r] = sat0; raysat[3 * r + 1] = sat1; raysat[3 * r + 2] = sat2; }
  }
}

template <bool WARP>
__global__ __launch_bounds__(256) void march_bwd_kernel(MarchArgs a, const float* __restrict__ raysat_im,
                                                        const float* __restrict__ grad_rayrgba,
                                                        float* __restrict__ grad_primpos,
                                                        float* __restrict__ grad_primrot,
                                                        float* __restrict__ grad_primscale,
                                                        float* __restrict__ grad_tplate,
                                                        float* __restrict__ grad_warp) {
  __shared__ int s_list[4][kListCap];
  __shared__ unsigned s_win[4][kListCap];
  // per-wave accumulators of the 15 transform-gradient sums of the first kTgSlots boxes of the hit list: every wave that
  // crosses a box adds to the SAME 15 global addresses, and memory-side float atomics to one address serialise -- so
  // they are issued once per (wave, box) at the end instead of once per (wave, box, step)
  __shared__ float s_tg[4][kTgSlots][16];
  const int n = blockIdx.z;
  const int lane = threadIdx.x & 63;
  Ray ray; float tmin, tmax; size_t r; int wave;
  load_ray(a, n, ray, tmin, tmax, r, wave);
  for (int i = lane; i < kTgSlots * 16; i += 64) (&s_tg[wave][0][0])[i] = 0.f;
  const size_t vox = (size_t)a.TD * a.TH * a.TW;
  const float* primpos = a.primpos + (size_t)n * a.K * 3;
  const float* primrot = a.primrot + (size_t)n * a.K * 9;
  const float* primscale = a.primscale + (size_t)n * a.K * 3;
  const float4* tplate = reinterpret_cast<const float4*>(a.tplate) + (size_t)n * a.K * vox;
  float* g_tplate = grad_tplate + (size_t)n * a.K * vox * 4;
  float* g_pos = grad_primpos + (size_t)n * a.K * 3;
  float* g_rot = grad_primrot + (size_t)n * a.K * 9;
  float* g_scale = grad_primscale + (size_t)n * a.K * 3;
  const int num = build_hits(a.K, a.nodeaabb + (size_t)n * (2 * a.K - 1) * 6, primpos, primrot, primscale, a.stepsize,
                             tmin, tmax, ray, s_list[wave], s_win[wave]);

  // PrimAccumAdditive::read (primaccum.h:58-61)
  const float4 dL = ray.live ? *reinterpret_cast<const float4*>(grad_rayrgba + 4 * r) : make_float4(0.f, 0.f, 0.f, 0.f);
  float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f, rs3 = 0.f;  // (raysat, 1) when the ray saturated, else 0
  if (ray.live && raysat_im[3 * r] > -1.f) { rs0 = raysat_im[3 * r]; rs1 = raysat_im[3 * r + 1]; rs2 = raysat_im[3 * r + 2]; rs3 = 1.f; }
  float accw = 0.f;
  bool sat = false;
  int iter = 0;
  const float fe = a.fadeexp, fs = a.fadescale;
  while (gol_ballot(ray.live && ray.t < ray.rt1 + 1e-5f && !sat) != 0ull) {
    for (int base = 0; base < num; base += 64) {  // (see the forward kernel)
    const int s_me = base + lane;
    unsigned long long active = gol_ballot(s_me < num && window_active(s_win[wave][s_me], iter));
    while (active) {
      const int s = base + __builtin_ctzll(active);
      active &= active - 1;
      const int e = __builtin_amdgcn_readfirstlane(s_list[wave][s]);
      const int k = e & kBoxMask;
      Xform xf;
      const V3 y0 = xform_fwd(xf, primpos, primrot, primscale, k, ray.pos);
      const bool ev = ray.live && half_keeps(e, lane) && valid_pos(y0) && !sat && ray.t < ray.rt1 + 1e-5f;
      if (gol_ballot(ev) == 0ull) continue;
      V3 dLy = v3(0.f, 0.f, 0.f);
      float sd0 = 0.f, sd1 = 0.f, sd2 = 0.f, sd3 = 0.f;
      // corner indices / weights of every lane (plain arithmetic; only lanes with `ev` use theirs): kept in `q` for the
      // scatter below instead of being copied out of the branch
      Tri q;
      Tri qw;  // WARP: the warp field's cell at y0
      bool evs = ev;  // lanes taking part in the template-gradient scatter
      if (WARP) {
        V3 y1 = y0;
        if (ev) y1 = warp_sample(a.warp + ((size_t)n * a.K + k) * (size_t)a.WD * a.WH * a.WW * 3, a.WD, a.WH, a.WW, y0, qw);
        tri_setup_any(q, a.TD, a.TH, a.TW, y1);
        evs = ev && tri_any(q);  // a warped sample outside the template has no corner to give a gradient to
      } else {
        tri_setup(q, a.TD, a.TH, a.TW, y0);
      }
      const int cellkey = ((q.z0 + 1) * (a.TH + 1) + (q.y0 + 1)) * (a.TW + 1) + (q.x0 + 1);
      if (ev) {
        const float ax = fabsf(y0.x), ay = fabsf(y0.y), az = fabsf(y0.z);
        const bool e8 = fe == 8.f;
        const float fade = __expf(-fs * (fade_pow(ax, fe, e8) + fade_pow(ay, fe, e8) + fade_pow(az, fe, e8)));
        const float4* tp = tplate + (size_t)k * vox;
        float4 cv[8];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          cv[c] = (q.idx[c] >= 0) ? *gol_at(tp, (unsigned)q.idx[c] * 16u) : make_float4(0.f, 0.f, 0.f, 0.f);
          s0 += cv[c].x * q.w[c]; s1 += cv[c].y * q.w[c]; s2 += cv[c].z * q.w[c]; s3 += cv[c].w * q.w[c];
        }
        s3 *= fade;
        // PrimAccumAdditive::forwardbackward_prim (primaccum.h:81-98)
        const float al = s3 * a.stepsize;
        sat = sat || (accw + al >= 1.f);
        const float weight = sat ? (1.f - accw) : al;
        float d0 = weight * dL.x, d1 = weight * dL.y, d2 = weight * dL.z;
        float d3 = sat ? 0.f
                       : a.stepsize * ((s0 - rs0) * dL.x + (s1 - rs1) * dL.y + (s2 - rs2) * dL.z + (1.f - rs3) * dL.w);
        accw += weight;
        // PrimSamplerTW::backward (primsampler.h:70-92)
        const float kf = -(fs * fe);
        dLy = v3(kf * fade_pow_m1(ax, fe, e8) * (y0.x > 0.f ? 1.f : -1.f), kf * fade_pow_m1(ay, fe, e8) * (y0.y > 0.f ? 1.f : -1.f),
                 kf * fade_pow_m1(az, fe, e8) * (y0.z > 0.f ? 1.f : -1.f)) * (s3 * d3);
        d3 *= fade;
        // trilinear backward (utils.h:619-770): position gradient here, template scatter below
        float gix = 0.f, giy = 0.f, giz = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (q.idx[c] >= 0) {
            const float dp = cv[c].x * d0 + cv[c].y * d1 + cv[c].z * d2 + cv[c].w * d3;
            const int dx = c & 1, dy = (c >> 1) & 1, dz = (c >> 2) & 1;
            const float wx = dx ? q.fx : 1.f - q.fx, wy = dy ? q.fy : 1.f - q.fy, wz = dz ? q.fz : 1.f - q.fz;
            gix += (dx ? dp : -dp) * wy * wz;
            giy += (dy ? dp : -dp) * wx * wz;
            giz += (dz ? dp : -dp) * wx * wy;
          }
        }
        V3 dLy1 = v3(gix * 0.5f * (float)(a.TW - 1), giy * 0.5f * (float)(a.TH - 1), giz * 0.5f * (float)(a.TD - 1));
        if (WARP) {
          // warp sampler backward at y0 (the chain of the reference's PyTorch fixture, mvpraymarch.py:621-626): the field's
          // gradient (plain per-lane atomics: algo 1 has no caller in the reference's models, it is here for the fixture)
          // and d y1 / d y0 applied to dLy1
          const float* wp = a.warp + ((size_t)n * a.K + k) * (size_t)a.WD * a.WH * a.WW * 3;
          float* gw = grad_warp + ((size_t)n * a.K + k) * (size_t)a.WD * a.WH * a.WW * 3;
          float wix = 0.f, wiy = 0.f, wiz = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            if (qw.idx[c] >= 0) {
              const float* v = gol_at(wp, (unsigned)qw.idx[c] * 12u);
              float* g = gol_at(gw, (unsigned)qw.idx[c] * 12u);
              atomicAdd(g, qw.w[c] * dLy1.x); atomicAdd(g + 1, qw.w[c] * dLy1.y); atomicAdd(g + 2, qw.w[c] * dLy1.z);
              const float dp = v[0] * dLy1.x + v[1] * dLy1.y + v[2] * dLy1.z;
              const int dx = c & 1, dy = (c >> 1) & 1, dz = (c >> 2) & 1;
              const float wx = dx ? qw.fx : 1.f - qw.fx, wy = dy ? qw.fy : 1.f - qw.fy, wz = dz ? qw.fz : 1.f - qw.fz;
              wix += (dx ? dp : -dp) * wy * wz;
              wiy += (dy ? dp : -dp) * wx * wz;
              wiz += (dz ? dp : -dp) * wx * wy;
            }
          }
          dLy1 = v3(wix * 0.5f * (float)(a.WW - 1), wiy * 0.5f * (float)(a.WH - 1), wiz * 0.5f * (float)(a.WD - 1));
        }
        dLy = dLy + dLy1;
        sd0 = d0; sd1 = d1; sd2 = d2; sd3 = d3;
      }
      // Template-gradient scatter.  The 64 rays of a wave sample only a handful of distinct voxel cells
      // of this box at this step, and every device-scope float atomic is a fabric transaction on
      // MI355X: group the lanes by cell (ballot + readlane), reduce each group's 8 corners x 4
      // channels over the wave (v_permlane swaps) and issue ONE atomic per (cell, corner, channel)
      // instead of one per lane (mvpraymarch utils.h:83-113 issues one per lane).
      {
        float* gt = g_tplate + (size_t)k * vox * 4;
        unsigned long long todo = gol_ballot(evs);
        while (todo) {
          const int leader = __builtin_ctzll(todo);
          const int key = __builtin_amdgcn_readlane(cellkey, leader);
          const bool mine = evs && (cellkey == key);
          const unsigned long long grp = gol_ballot(mine);
          todo &= ~grp;
          // always reduce (even a one-lane group).  The group's lanes are selected ONCE, on the four sample gradients
          // (the corner weights stay unmasked), and lanes 15, 31, 47, 63 -- which hold the four channel sums of a corner --
          // issue that corner's atomics themselves: moving the second corner of a pair to lanes 14, 30, .. so that one
          // instruction covers both cost six vector instructions per pair for one atomic instruction saved
          // (profiles/r03g_mvp_bwd_probes.txt: all device atomics of this kernel together are 0.3 of its 11.4 ms).
          const float m0 = mine ? sd0 : 0.f, m1 = mine ? sd1 : 0.f, m2 = mine ? sd2 : 0.f, m3 = mine ? sd3 : 0.f;
          // phase 1: the eight corners' reduction chains in ONE basic block (the scheduler interleaves them; with the
          // atomic's exec-mask region after every corner they ran one after the other, each a ~60-cycle dependent chain)
          float rc[8];
          int ic[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            ic[c] = __builtin_amdgcn_readlane(q.idx[c], leader);
            rc[c] = gol_wave_sum4(q.w[c] * m0, q.w[c] * m1, q.w[c] * m2, q.w[c] * m3);   // lanes 15, 31, 47, 63
          }
          // phase 2: one exec-mask region, the (wave-uniform) validity of a corner is a scalar branch inside it
          if ((lane & 15) == 15) {
            float* gl = gol_at(gt, (unsigned)(lane >> 4) * 4u);
#pragma unroll
            for (int c = 0; c < 8; ++c)
              if (ic[c] >= 0) atomicAdd(gol_at(gl, (unsigned)ic[c] * 16u), rc[c]);
          }
        }
      }
      // PrimTransfSRT::backward (primtransf.h:155-179): 15 wave sums, one atomic each
      const V3 gs = ev ? xf.rxmt * dLy : v3(0.f, 0.f, 0.f);
      const V3 d = ev ? dLy * xf.ps : v3(0.f, 0.f, 0.f);
      const float gp0 = -dot(xf.pr0, d), gp1 = -dot(xf.pr1, d), gp2 = -dot(xf.pr2, d);
      const float r0 = gol_wave_sum4(gs.x, gs.y, gs.z, gp0);
      const float r1 = gol_wave_sum4(gp1, gp2, xf.xmt.x * d.x, xf.xmt.x * d.y);
      const float r2 = gol_wave_sum4(xf.xmt.x * d.z, xf.xmt.y * d.x, xf.xmt.y * d.y, xf.xmt.y * d.z);
      const float r3 = gol_wave_sum4(xf.xmt.z * d.x, xf.xmt.z * d.y, xf.xmt.z * d.z, 0.f);
      if ((lane & 15) == 15) {
        const int j = lane >> 4;  // 0..3: which of the four sums this lane holds
        // r0: scale.x scale.y scale.z pos.x | r1: pos.y pos.z rot[0] rot[1] | r2: rot[2..5] | r3: rot[6..8]
        if (s < kTgSlots) {  // wave-private LDS slot (in-order LDS ops of one wave: plain read-modify-write)
          float* t = &s_tg[wave][s][j];
          t[0] += r0; t[4] += r1; t[8] += r2; t[12] += r3;
        } else {
          float* p0 = (j < 3) ? (g_scale + 3 * k + j) : (g_pos + 3 * k);
          atomicAdd(p0, r0);
          float* p1 = (j < 2) ? (g_pos + 3 * k + 1 + j) : (g_rot + 9 * k + (j - 2));
          atomicAdd(p1, r1);
          atomicAdd(g_rot + 9 * k + 2 + j, r2);
          if (j < 3) atomicAdd(g_rot + 9 * k + 6 + j, r3);
        }
      }
    }
    }
    ray.t += a.stepsize;
    ray.pos = ray.pos + ray.dir * a.stepsize;
    ++iter;
  }
  // flush: 4 boxes per wave-instruction, lane = (box, component); component order = [scale 0-2 | pos 3-5 | rot 6-14 | -]
  __builtin_amdgcn_wave_barrier();
  const int nflush = min(num, kTgSlots);
  for (int s0 = 0; s0 < nflush; s0 += 4) {
    const int s = s0 + (lane >> 4), i = lane & 15;
    if (s < nflush && i < 15) {
      const float v = s_tg[wave][s][i];
      if (v != 0.f) {
        const int k = s_list[wave][s] & kBoxMask;
        float* dst = i < 3 ? g_scale + 3 * k + i : (i < 6 ? g_pos + 3 * k + (i - 3) : g_rot + 9 * k + (i - 6));
        atomicAdd(dst, v);
      }
    }
  }
}

int check_march(int N, int H, int W, int K, int TD, int TH, int TW, float stepsize) {
  GOL_REQUIRE(N >= 0 && H >= 0 && W >= 0 && K >= 1 && K <= kBoxMask, "bad sizes");
  GOL_REQUIRE(TD > 0 && TH > 0 && TW > 0, "bad template size");
  GOL_REQUIRE(stepsize > 0.f, "stepsize must be positive");
  GOL_REQUIRE(N <= 65535, "N > 65535");
  return GOL_OK;
}

}  // namespace

extern "C" int gol_raydirs_fwd(int N, int H, int W, const float* viewpos, const float* viewrot, const float* focal,
                               const float* princpt, const float* pixelcoords, float volradius, float* raypos,
                               float* raydir, float* tminmax, void* stream) {
  GOL_REQUIRE(N >= 0 && H >= 0 && W >= 0, "negative size");
  if (N == 0 || H == 0 || W == 0) return GOL_OK;
  GOL_REQUIRE(viewpos && viewrot && focal && princpt && raypos && raydir && tminmax, "null pointer");
  GOL_REQUIRE(N <= 65535, "N > 65535");
  raydirs_kernel<<<dim3(gol_cdiv((long long)H * W, 256), N), 256, 0, (hipStream_t)stream>>>(
      N, H, W, viewpos, viewrot, focal, princpt, pixelcoords, volradius, raypos, raydir, tminmax);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_mvp_aabb(int N, int K, const float* primpos, const float* primrot, const float* primscale,
                            float* nodeaabb, void* stream) {
  GOL_REQUIRE(N >= 0 && K >= 1, "bad sizes");
  if (N == 0) return GOL_OK;
  GOL_REQUIRE(primpos && primrot && primscale && nodeaabb, "null pointer");
  aabb_kernel<<<N, 1024, 0, (hipStream_t)stream>>>(K, primpos, primrot, primscale, nodeaabb);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_mvp_march_fwd(int N, int H, int W, int K, const float* raypos, const float* raydir, float stepsize,
                                 const float* tminmax, const float* nodeaabb, const float* primpos,
                                 const float* primrot, const float* primscale, const float* tplate, int TD, int TH,
                                 int TW, float fadescale, float fadeexp, float* rayrgba, float* raysat, float* shadow,
                                 void* stream) {
  int rc = check_march(N, H, W, K, TD, TH, TW, stepsize);
  if (rc != GOL_OK) return rc;
  if (N == 0 || H == 0 || W == 0) return GOL_OK;
  GOL_REQUIRE(raypos && raydir && tminmax && nodeaabb && primpos && primrot && primscale && tplate && rayrgba,
              "null pointer");
  MarchArgs a{N, H, W, K, TD, TH, TW, stepsize, fadescale, fadeexp, raypos, raydir, tminmax, nodeaabb,
              primpos, primrot, primscale, tplate};
  dim3 grid(gol_cdiv(W, 16), gol_cdiv(H, 16), N);
  if (shadow) march_fwd_kernel<true, false><<<grid, 256, 0, (hipStream_t)stream>>>(a, rayrgba, raysat, shadow);
  else march_fwd_kernel<false, false><<<grid, 256, 0, (hipStream_t)stream>>>(a, rayrgba, raysat, shadow);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_mvp_march_warp_fwd(int N, int H, int W, int K, const float* raypos, const float* raydir,
                                      float stepsize, const float* tminmax, const float* nodeaabb, const float* primpos,
                                      const float* primrot, const float* primscale, const float* tplate, int TD, int TH,
                                      int TW, const float* warp, int WD, int WH, int WW, float fadescale, float fadeexp,
                                      float* rayrgba, float* raysat, float* shadow, void* stream) {
  int rc = check_march(N, H, W, K, TD, TH, TW, stepsize);
  if (rc != GOL_OK) return rc;
  GOL_REQUIRE(WD > 0 && WH > 0 && WW > 0, "bad warp field size");
  if (N == 0 || H == 0 || W == 0) return GOL_OK;
  GOL_REQUIRE(raypos && raydir && tminmax && nodeaabb && primpos && primrot && primscale && tplate && warp && rayrgba,
              "null pointer");
  MarchArgs a{N, H, W, K, TD, TH, TW, stepsize, fadescale, fadeexp, raypos, raydir, tminmax, nodeaabb,
              primpos, primrot, primscale, tplate};
  a.warp = warp; a.WD = WD; a.WH = WH; a.WW = WW;
  dim3 grid(gol_cdiv(W, 16), gol_cdiv(H, 16), N);
  if (shadow) march_fwd_kernel<true, true><<<grid, 256, 0, (hipStream_t)stream>>>(a, rayrgba, raysat, shadow);
  else march_fwd_kernel<false, true><<<grid, 256, 0, (hipStream_t)stream>>>(a, rayrgba, raysat, shadow);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_mvp_shadow_march(int N, int group, int H, int W, int K, const float* raypos, const float* raydir,
                                    float stepsize, const float* tminmax, const float* nodeaabb, const float* primpos,
                                    const float* primrot, const float* primscale, const float* tplate, int alpha_only,
                                    int TD, int TH, int TW, float fadescale, float fadeexp, float* rayrgba, float* shadow,
                                    void* stream) {
  int rc = check_march(N, H, W, K, TD, TH, TW, stepsize);
  if (rc != GOL_OK) return rc;
  GOL_REQUIRE(group >= 1 && N % group == 0, "N must be a multiple of group");
  if (N == 0 || H == 0 || W == 0) return GOL_OK;
  GOL_REQUIRE(raypos && raydir && tminmax && nodeaabb && primpos && primrot && primscale && tplate && shadow,
              "null pointer");
  MarchArgs a{N, H, W, K, TD, TH, TW, stepsize, fadescale, fadeexp, raypos, raydir, tminmax, nodeaabb,
              primpos, primrot, primscale, tplate, group, alpha_only ? 1 : 0};
  dim3 grid(gol_cdiv(W, 16), gol_cdiv(H, 16), N);
  march_fwd_kernel<true, false><<<grid, 256, 0, (hipStream_t)stream>>>(a, rayrgba, nullptr, shadow);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_mvp_march_bwd(int N, int H, int W, int K, const float* raypos, const float* raydir, float stepsize,
                                 const float* tminmax, const float* nodeaabb, const float* primpos,
                                 const float* primrot, const float* primscale, const float* tplate, int TD, int TH,
                                 int TW, float fadescale, float fadeexp, const float* raysat,
                                 const float* grad_rayrgba, float* grad_primpos, float* grad_primrot,
                                 float* grad_primscale, float* grad_tplate, void* stream) {
  int rc = check_march(N, H, W, K, TD, TH, TW, stepsize);
  if (rc != GOL_OK) return rc;
  if (N == 0 || H == 0 || W == 0) return GOL_OK;
  GOL_REQUIRE(raypos && raydir && tminmax && nodeaabb && primpos && primrot && primscale && tplate, "null pointer");
  GOL_REQUIRE(raysat && grad_rayrgba && grad_primpos && grad_primrot && grad_primscale && grad_tplate, "null pointer");
  MarchArgs a{N, H, W, K, TD, TH, TW, stepsize, fadescale, fadeexp, raypos, raydir, tminmax, nodeaabb,
              primpos, primrot, primscale, tplate};
  dim3 grid(gol_cdiv(W, 16), gol_cdiv(H, 16), N);
  march_bwd_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(a, raysat, grad_rayrgba, grad_primpos, grad_primrot,
                                                                  grad_primscale, grad_tplate, nullptr);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_mvp_march_warp_bwd(int N, int H, int W, int K, const float* raypos, const float* raydir,
                                      float stepsize, const float* tminmax, const float* nodeaabb, const float* primpos,
                                      const float* primrot, const float* primscale, const float* tplate, int TD, int TH,
                                      int TW, const float* warp, int WD, int WH, int WW, float fadescale, float fadeexp,
                                      const float* raysat, const float* grad_rayrgba, float* grad_primpos,
                                      float* grad_primrot, float* grad_primscale, float* grad_tplate, float* grad_warp,
                                      void* stream) {
  int rc = check_march(N, H, W, K, TD, TH, TW, stepsize);
  if (rc != GOL_OK) return rc;
  GOL_REQUIRE(WD > 0 && WH > 0 && WW > 0, "bad warp field size");
  if (N == 0 || H == 0 || W == 0) return GOL_OK;
  GOL_REQUIRE(raypos && raydir && tminmax && nodeaabb && primpos && primrot && primscale && tplate && warp, "null pointer");
  GOL_REQUIRE(raysat && grad_rayrgba && grad_primpos && grad_primrot && grad_primscale && grad_tplate && grad_warp,
              "null pointer");
  MarchArgs a{N, H, W, K, TD, TH, TW, stepsize, fadescale, fadeexp, raypos, raydir, tminmax, nodeaabb,
              primpos, primrot, primscale, tplate};
  a.warp = warp; a.WD = WD; a.WH = WH; a.WW = WW;
  dim3 grid(gol_cdiv(W, 16), gol_cdiv(H, 16), N);
  march_bwd_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(a, raysat, grad_rayrgba, grad_primpos, grad_primrot,
                                                                 grad_primscale, grad_tplate, grad_warp);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
