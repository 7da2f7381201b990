// meshraster.hip -- z-buffer rasterizer of small triangle meshes (index / depth / barycentric images), gfx950.
//
// Stand-in for the two drtk calls of the reference's mesh render layer,
//   /root/reference/ca_code/utils/render_drtk.py:44-46   index_img = rasterize(v_pix, vi, h, w)
//                                                        depth_img, bary_img = render(v_pix, vi, index_img)
// whose only consumer on the hot path is the shadow map: get_shadow_map renders the hand mesh from every light and
// keeps `depth_img` (ca_code/utils/shadowmap.py:39-50; 32 lights x 1024^2 for the OLAT sweep of BASELINE config 4).
// drtk is third-party and absent from the reference tree (requirements.txt:6, unpinned), so the conventions are stated,
// not copied: pixel (i, j) is sampled at its centre (j + 0.5, i + 0.5) in v_pix units; a face covers a sample when all
// three edge functions are >= 0 (either winding); among the covering faces the smallest positive depth wins, ties go to
// the lower face index; depth and barycentrics are perspective-correct (linear in 1 / z).
// Two kernels: per-face setup (edge equations + tile bounds, 64 B per face, plus the tile bounding box of the whole
// mesh per view), then one 256-thread workgroup per 16x16 tile: tiles outside the mesh's box only clear their pixels
// (a hand seen from a light 1.1 m away covers a few percent of the 1024^2 light camera); the others compact the faces
// whose bounds touch the tile into LDS, 1024 faces per round, from the faces' 8-byte packed tile bounds (a separate
// array: the test reads 8 bytes per face, not the 64-byte record), and every lane walks that short list for its own
// pixel with wave-uniform (scalar) loads of the face records.  No global atomics besides the four per workgroup of
// the box.  (Round 2 tested 256 faces per round from the 64-byte records: 20 rounds of two barriers and 327 KB of L2
// reads per covered tile for the 5120-face hand.)
#include "gol_common.h"

namespace {

struct FaceRec {
  // barycentrics relative to vertex a (anchored form: the edge functions evaluated in absolute pixel coordinates lose
  // ~1e-4 of their value to cancellation in a 1024^2 image -- 0.2 mm of depth at 1 m, visible in the shadow test
  // exp(-(d1 - d2) / 8)):  b1 = e[3] (x - ax) + e[4] (y - ay),  b2 = e[6] (x - ax) + e[7] (y - ay),  b0 = 1 - b1 - b2;
  // e[0], e[1] = (ax, ay); e[2], e[5], e[8] unused
  float e[9];
  float iz[3];     // 1 / z of the three vertices
  int32_t tx0, tx1, ty0, ty1;  // inclusive tile bounds, tx1 < tx0 = culled
};

// per-view tile box of the mesh: {min tx, min ty, -max tx, -max ty} (all reduced with atomicMin; initialised to INT_MAX)
__global__ void box_init_kernel(int n, int32_t* __restrict__ box) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) box[i] = 2147483647;
}

__global__ __launch_bounds__(256) void face_setup_kernel(int V, int F, int H, int W, const float* __restrict__ v_pix,
                                                         const int32_t* __restrict__ vi, FaceRec* __restrict__ rec,
                                                         int32_t* __restrict__ box, uint2* __restrict__ bounds) {
  const int f = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  FaceRec r;
  r.tx0 = 1; r.tx1 = 0; r.ty0 = 1; r.ty1 = 0;
  int i0 = -1, i1 = -1, i2 = -1;
  if (f < F) { i0 = vi[3 * f]; i1 = vi[3 * f + 1]; i2 = vi[3 * f + 2]; }
  const bool ok_idx = f < F && i0 >= 0 && i0 < V && i1 >= 0 && i1 < V && i2 >= 0 && i2 < V;
  if (ok_idx) {
    const float* p = v_pix + (size_t)b * V * 3;
    const float ax = p[3 * i0], ay = p[3 * i0 + 1], az = p[3 * i0 + 2];
    const float bx = p[3 * i1], by = p[3 * i1 + 1], bz = p[3 * i1 + 2];
    const float cx = p[3 * i2], cy = p[3 * i2 + 1], cz = p[3 * i2 + 2];
    const float area = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
    const bool fin = isfinite(ax) && isfinite(ay) && isfinite(bx) && isfinite(by) && isfinite(cx) && isfinite(cy);
    if (fin && az > 0.f && bz > 0.f && cz > 0.f && area != 0.f) {
      const float ia = 1.f / area;
      // barycentric of vertex b: area(p, c, a) / area, of vertex c: area(p, a, b) / area; both vanish at a
      r.e[0] = ax; r.e[1] = ay; r.e[2] = 0.f;
      r.e[3] = (cy - ay) * ia; r.e[4] = (ax - cx) * ia; r.e[5] = 0.f;
      r.e[6] = (ay - by) * ia; r.e[7] = (bx - ax) * ia; r.e[8] = 0.f;
      r.iz[0] = 1.f / az; r.iz[1] = 1.f / bz; r.iz[2] = 1.f / cz;
      // pixels whose centre can be inside: centre x = j + 0.5 in [min, max]
      const float x0 = fminf(ax, fminf(bx, cx)), x1 = fmaxf(ax, fmaxf(bx, cx));
      const float y0 = fminf(ay, fminf(by, cy)), y1 = fmaxf(ay, fmaxf(by, cy));
      const int j0 = max(0, (int)ceilf(x0 - 0.5f)), j1 = min(W - 1, (int)floorf(x1 - 0.5f));
      const int k0 = max(0, (int)ceilf(y0 - 0.5f)), k1 = min(H - 1, (int)floorf(y1 - 0.5f));
      if (j0 <= j1 && k0 <= k1 && x1 >= 0.f && y1 >= 0.f && x0 <= (float)W && y0 <= (float)H) {
        r.tx0 = j0 >> 4; r.tx1 = j1 >> 4; r.ty0 = k0 >> 4; r.ty1 = k1 >> 4;
      }
    }
  }
  if (f < F) {
    rec[(size_t)b * F + f] = r;
    // tile bounds, 16 bits each: (tx0 | tx1 << 16, ty0 | ty1 << 16); culled faces keep tx1 < tx0
    bounds[(size_t)b * F + f] = make_uint2((uint32_t)r.tx0 | ((uint32_t)r.tx1 << 16), (uint32_t)r.ty0 | ((uint32_t)r.ty1 << 16));
  }
  // workgroup box -> one atomicMin per component
  __shared__ int32_t s_box[4];
  if (threadIdx.x < 4) s_box[threadIdx.x] = 2147483647;
  __syncthreads();
  if (r.tx0 <= r.tx1) {
    atomicMin(&s_box[0], r.tx0); atomicMin(&s_box[1], r.ty0); atomicMin(&s_box[2], -r.tx1); atomicMin(&s_box[3], -r.ty1);
  }
  __syncthreads();
  if (threadIdx.x < 4 && s_box[threadIdx.x] != 2147483647) atomicMin(&box[4 * b + threadIdx.x], s_box[threadIdx.x]);
}

// "no face" everywhere (index -1, depth 0, barycentrics 0): a streaming fill with 16-byte stores.  The raster workgroups
// then only write the pixels a face covers.
template <typename I4, typename F4>
__global__ __launch_bounds__(256) void mesh_clear_kernel(size_t n, I4* __restrict__ index_img, F4* __restrict__ depth_img,
                                                         size_t nb, F4* __restrict__ bary_img, I4 none, F4 zero) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { index_img[i] = none; depth_img[i] = zero; }
  if (bary_img)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) bary_img[i] = zero;
}

// in-box tile counts of the views -> exclusive prefix (prefix[B] = total): the raster kernel's work list.  One workgroup.
__global__ __launch_bounds__(1024) void tile_prefix_kernel(int B, const int32_t* __restrict__ box, int32_t* __restrict__ prefix) {
  __shared__ int32_t s_wave[16];
  __shared__ int32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < B; base += 1024) {
    const int b = base + tid;
    int v = 0;
    if (b < B) {
      const int w = -box[4 * b + 2] - box[4 * b] + 1, h = -box[4 * b + 3] - box[4 * b + 1] + 1;  // (empty box: INT_MAX mins)
      v = (box[4 * b] != 2147483647 && w > 0 && h > 0) ? w * h : 0;
    }
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int wave_off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int w = s_wave[k]; if (k < wv) wave_off += w; tot += w; }
    const int carry = s_carry;
    if (b < B) prefix[b] = carry + wave_off + incl - v;
    __syncthreads();
    if (tid == 0) s_carry = carry + tot;
    __syncthreads();
  }
  if (tid == 0) prefix[B] = s_carry;
}

constexpr int kFaceRound = 1024;  // faces tested per compaction round (all of them may hit: the list holds a full round)
constexpr int kRasterWgs = 4096;  // persistent workgroups of the raster kernel

// One 256-thread workgroup per IN-BOX 16x16 tile, taken from the implicit work list prefix[] (entry e -> view b with
// prefix[b] <= e < prefix[b + 1], tile e - prefix[b] of the view's box, row-major): a fixed grid strides over the list.
// Launching one workgroup per tile of every image made the kernel dispatch-bound: 131 k workgroups of which ~3 k had
// work took 0.15 ms at the ~1 workgroup/ns the dispatcher sustains (profiles/r03g_urhand_kernel_trace.txt).
__global__ __launch_bounds__(256) void mesh_raster_kernel(int B, int F, int H, int W, const FaceRec* __restrict__ rec,
                                                          const uint2* __restrict__ bounds,
                                                          const int32_t* __restrict__ box,
                                                          const int32_t* __restrict__ prefix,
                                                          int32_t* __restrict__ index_img, float* __restrict__ depth_img,
                                                          float* __restrict__ bary_img) {
  __shared__ int32_t s_face[kFaceRound];
  __shared__ float4 s_rec[3][256];
  __shared__ int32_t s_count;
  __shared__ int32_t s_view;
  const int tid = threadIdx.x;
  const int total = prefix[B];
  for (int e = blockIdx.x; e < total; e += gridDim.x) {
    // view of entry e: 256 views per step, lane = view
    __syncthreads();
    for (int vb = 0; vb < B; vb += 256) {
      const int b = vb + tid;
      if (b < B && prefix[b] <= e && e < prefix[b + 1]) s_view = b;
    }
    __syncthreads();
    const int b = s_view;
    const int bx0 = box[4 * b], by0 = box[4 * b + 1], bw = -box[4 * b + 2] - bx0 + 1;
    const int l = e - prefix[b];
    const int tx = bx0 + l % bw, ty = by0 + l / bw;
    const FaceRec* R = rec + (size_t)b * F;
    const uint2* Bd = bounds + (size_t)b * F;
    const int j = tx * 16 + (tid & 15), i = ty * 16 + (tid >> 4);
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    float best_iz = 0.f;   // 1 / depth of the nearest covering face (larger = nearer)
    int best = -1;
    float bb0 = 0.f, bb1 = 0.f, bb2 = 0.f;
    for (int base = 0; base < F; base += kFaceRound) {
      if (tid == 0) s_count = 0;
      __syncthreads();
      // compaction (any order: the z-test below breaks depth ties by face index)
#pragma unroll
      for (int u = 0; u < kFaceRound / 256; ++u) {
        const int f = base + u * 256 + tid;
        if (f < F) {
          const uint2 bd = Bd[f];
          const int x0 = (int)(bd.x & 0xffffu), x1 = (int)(bd.x >> 16), y0 = (int)(bd.y & 0xffffu), y1 = (int)(bd.y >> 16);
          if (x0 <= tx && tx <= x1 && y0 <= ty && ty <= y1) s_face[atomicAdd(&s_count, 1)] = f;
        }
      }
      __syncthreads();
      const int n = s_count;
      // the hit faces' records go through LDS, 256 at a time (one record per thread, coalesced 16-byte loads): walking the
      // list with one dependent scalar load per face cost ~0.5 us per face and lane-loop iteration
      for (int c0 = 0; c0 < n; c0 += 256) {
        const int cn = min(256, n - c0);
        if (tid < cn) {
          const float4* src = reinterpret_cast<const float4*>(R + s_face[c0 + tid]);
          const float4 r0 = src[0], r1 = src[1], r2 = src[2];  // e[0..8], iz[0..2]
          s_rec[0][tid] = make_float4(r0.x, r0.y, r0.w, r1.x);  // ax ay e3 e4
          s_rec[1][tid] = make_float4(r1.z, r1.w, r2.y, r2.z);  // e6 e7 iz0 iz1
          s_rec[2][tid] = make_float4(r2.w, __int_as_float(s_face[c0 + tid]), 0.f, 0.f);  // iz2, face index
        }
        __syncthreads();
        for (int k = 0; k < cn; ++k) {
          const float4 q0 = s_rec[0][k], q1 = s_rec[1][k], q2 = s_rec[2][k];  // same address in every lane: LDS broadcast
          const float dx = px - q0.x, dy = py - q0.y;
          const float b1 = q0.z * dx + q0.w * dy;
          const float b2 = q1.x * dx + q1.y * dy;
          const float b0 = 1.f - b1 - b2;
          if (b0 >= 0.f && b1 >= 0.f && b2 >= 0.f) {
            const float w0 = b0 * q1.z, w1 = b1 * q1.w, w2 = b2 * q2.x;
            const float iz = w0 + w1 + w2;  // 1 / depth at the sample
            const int fi = __float_as_int(q2.y);
            if (iz > best_iz || (iz == best_iz && fi < best)) {  // nearer; exact ties go to the lower face index
              best_iz = iz; best = fi;
              const float z = 1.f / iz;
              bb0 = w0 * z; bb1 = w1 * z; bb2 = w2 * z;
            }
          }
        }
        __syncthreads();
      }
    }
    if (i < H && j < W && best >= 0) {  // (mesh_clear_kernel wrote "no face" everywhere)
      const size_t p = ((size_t)b * H + i) * W + j;
      index_img[p] = best;
      depth_img[p] = 1.f / best_iz;
      if (bary_img) {
        const size_t hw = (size_t)H * W, qq = (size_t)b * 3 * hw + (size_t)i * W + j;
        bary_img[qq] = bb0; bary_img[qq + hw] = bb1; bary_img[qq + 2 * hw] = bb2;
      }
    }
  }
}

}  // namespace

extern "C" int64_t gol_mesh_raster_workspace_bytes(int B, int F) {
  // face records | per-view tile box (16 bytes per view) | packed tile bounds | in-box tile prefix
  return (int64_t)B * F * (int64_t)sizeof(FaceRec) + (int64_t)B * 4 * (int64_t)sizeof(int32_t) +
         (int64_t)B * F * (int64_t)sizeof(uint2) + ((int64_t)B + 1) * (int64_t)sizeof(int32_t);
}

extern "C" int gol_mesh_raster(int B, int V, int F, int H, int W, const float* v_pix, const int32_t* vi,
                               int32_t* index_img, float* depth_img, float* bary_img, void* workspace, void* stream) {
  GOL_REQUIRE(B >= 0 && V >= 0 && F >= 0 && H > 0 && W > 0, "bad size");
  GOL_REQUIRE(B <= 65535 && H <= 16 * 65535 && W <= 16 * 65535, "size out of range");  // (tile bounds are packed in 16 bits)
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(index_img && depth_img, "null output");
  GOL_REQUIRE(workspace != nullptr, "null workspace");
  GOL_REQUIRE(F == 0 || (v_pix && vi), "null input");
  hipStream_t s = (hipStream_t)stream;
  FaceRec* rec = reinterpret_cast<FaceRec*>(workspace);
  int32_t* box = reinterpret_cast<int32_t*>(rec + (size_t)B * F);  // [B,4] behind the face records
  uint2* bounds = reinterpret_cast<uint2*>(box + (size_t)B * 4);   // [B,F] behind the boxes (16 B per view: stays 8-aligned)
  int32_t* prefix = reinterpret_cast<int32_t*>(bounds + (size_t)B * F);  // [B+1] in-box tile counts, exclusive prefix
  box_init_kernel<<<gol_cdiv(4 * B, 256), 256, 0, s>>>(4 * B, box);
  if (F > 0) face_setup_kernel<<<dim3(gol_cdiv(F, 256), B), 256, 0, s>>>(V, F, H, W, v_pix, vi, rec, box, bounds);
  // "no face" everywhere by a streaming fill (16-byte stores when the element counts and addresses allow), then the covered
  // pixels of the in-box tiles
  const size_t npix = (size_t)B * H * W;
  const bool aligned = npix % 4 == 0 && ((uintptr_t)index_img | (uintptr_t)depth_img | (uintptr_t)bary_img) % 16 == 0;
  if (aligned)
    mesh_clear_kernel<int4, float4><<<2048, 256, 0, s>>>(npix / 4, reinterpret_cast<int4*>(index_img),
                                                        reinterpret_cast<float4*>(depth_img), 3 * npix / 4,
                                                        reinterpret_cast<float4*>(bary_img), make_int4(-1, -1, -1, -1),
                                                        make_float4(0.f, 0.f, 0.f, 0.f));
  else
    mesh_clear_kernel<int32_t, float><<<2048, 256, 0, s>>>(npix, index_img, depth_img, 3 * npix, bary_img, -1, 0.f);
  if (F > 0) {
    tile_prefix_kernel<<<1, 1024, 0, s>>>(B, box, prefix);
    mesh_raster_kernel<<<kRasterWgs, 256, 0, s>>>(B, F, H, W, rec, bounds, box, prefix, index_img, depth_img, bary_img);
  }
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
