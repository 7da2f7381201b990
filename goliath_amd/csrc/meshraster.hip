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
// whose bounds touch the tile into LDS 256 at a time and every lane walks that short list for its own pixel with
// wave-uniform (scalar) loads of the face records.  No global atomics besides the four per workgroup of the box.
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
                                                         int32_t* __restrict__ box) {
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
  if (f < F) rec[(size_t)b * F + f] = r;
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

__global__ __launch_bounds__(256) void mesh_raster_kernel(int F, int H, int W, const FaceRec* __restrict__ rec,
                                                          const int32_t* __restrict__ box,
                                                          int32_t* __restrict__ index_img, float* __restrict__ depth_img,
                                                          float* __restrict__ bary_img) {
  __shared__ int32_t s_face[256];
  __shared__ int32_t s_count;
  const int b = blockIdx.z, tx = blockIdx.x, ty = blockIdx.y, tid = threadIdx.x;
  const int j = tx * 16 + (tid & 15), i = ty * 16 + (tid >> 4);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const FaceRec* R = rec + (size_t)b * F;
  float best_iz = 0.f;   // 1 / depth of the nearest covering face (larger = nearer)
  int best = -1;
  float bb0 = 0.f, bb1 = 0.f, bb2 = 0.f;
  const bool in_box = F > 0 && box[4 * b] <= tx && tx <= -box[4 * b + 2] && box[4 * b + 1] <= ty && ty <= -box[4 * b + 3];
  for (int base = 0; in_box && base < F; base += 256) {
    if (tid == 0) s_count = 0;
    __syncthreads();
    const int f = base + tid;
    bool hit = false;
    if (f < F) {
      const FaceRec& r = R[f];
      hit = r.tx0 <= tx && tx <= r.tx1 && r.ty0 <= ty && ty <= r.ty1;
    }
    // compaction (any order: the z-test below breaks depth ties by face index)
    if (hit) s_face[atomicAdd(&s_count, 1)] = f;
    __syncthreads();
    const int n = s_count;
    for (int k = 0; k < n; ++k) {
      const int fi = __builtin_amdgcn_readfirstlane(s_face[k]);
      const FaceRec& r = R[fi];  // wave-uniform address: scalar loads
      const float dx = px - r.e[0], dy = py - r.e[1];
      const float b1 = r.e[3] * dx + r.e[4] * dy;
      const float b2 = r.e[6] * dx + r.e[7] * dy;
      const float b0 = 1.f - b1 - b2;
      if (b0 >= 0.f && b1 >= 0.f && b2 >= 0.f) {
        const float w0 = b0 * r.iz[0], w1 = b1 * r.iz[1], w2 = b2 * r.iz[2];
        const float iz = w0 + w1 + w2;  // 1 / depth at the sample
        if (iz > best_iz || (iz == best_iz && fi < best)) {  // nearer; exact ties go to the lower face index
          best_iz = iz; best = fi;
          const float z = 1.f / iz;
          bb0 = w0 * z; bb1 = w1 * z; bb2 = w2 * z;
        }
      }
    }
    __syncthreads();
  }
  if (i < H && j < W) {
    const size_t p = ((size_t)b * H + i) * W + j;
    index_img[p] = best;
    depth_img[p] = best >= 0 ? 1.f / best_iz : 0.f;
    if (bary_img) {
      const size_t hw = (size_t)H * W, q = (size_t)b * 3 * hw + (size_t)i * W + j;
      bary_img[q] = bb0; bary_img[q + hw] = bb1; bary_img[q + 2 * hw] = bb2;
    }
  }
}

}  // namespace

extern "C" int64_t gol_mesh_raster_workspace_bytes(int B, int F) {
  return (int64_t)B * F * (int64_t)sizeof(FaceRec) + (int64_t)B * 4 * (int64_t)sizeof(int32_t);
}

extern "C" int gol_mesh_raster(int B, int V, int F, int H, int W, const float* v_pix, const int32_t* vi,
                               int32_t* index_img, float* depth_img, float* bary_img, void* workspace, void* stream) {
  GOL_REQUIRE(B >= 0 && V >= 0 && F >= 0 && H > 0 && W > 0, "bad size");
  GOL_REQUIRE(B <= 65535 && H <= 16 * 65535 && W <= 16 * 65535, "size out of range");
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(index_img && depth_img, "null output");
  GOL_REQUIRE(workspace != nullptr, "null workspace");
  GOL_REQUIRE(F == 0 || (v_pix && vi), "null input");
  hipStream_t s = (hipStream_t)stream;
  FaceRec* rec = reinterpret_cast<FaceRec*>(workspace);
  int32_t* box = reinterpret_cast<int32_t*>(rec + (size_t)B * F);  // [B,4] behind the face records
  box_init_kernel<<<gol_cdiv(4 * B, 256), 256, 0, s>>>(4 * B, box);
  if (F > 0) face_setup_kernel<<<dim3(gol_cdiv(F, 256), B), 256, 0, s>>>(V, F, H, W, v_pix, vi, rec, box);
  mesh_raster_kernel<<<dim3(gol_cdiv(W, 16), gol_cdiv(H, 16), B), 256, 0, s>>>(F, H, W, rec, box, index_img, depth_img, bary_img);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
