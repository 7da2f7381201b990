// render.hip -- the whole render direction of a batch of views as ONE C-ABI call.
//
// Replaces, per call, what the reference issues per view from Python: project_gaussians -> rasterize_gaussians(colour)
// -> rasterize_gaussians(depth) with their host syncs (/root/reference/ca_code/utils/render_gsplat.py:49-104, called in
// a loop over the views by AutoEncoder.render, ca_code/models/rgca.py:112-151).  gol_render_fwd enqueues
//   projection (+ packed raster records) -> reserving tile count -> scan -> scatter -> per-tile sort -> colour + depth
//   raster (+ the fused epilogue: alpha, depth / clamp(alpha), optional masked L1 against a target)
// and gol_render_bwd the raster backward + projection backward, on the given stream, out of ONE caller-provided
// workspace whose layout gol_render_layout computes.  Nothing is allocated here; the only host-side work of a direction
// is this one call -- an eager training loop then issues a step in ~1 ms of host time (rounds 1-2: three ABI calls and
// ~25 tensor allocations per direction).  The kernels are the ones of project.hip / binning.hip / raster.hip.
#include <cstring>

#include "gol_common.h"

namespace {
inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
template <typename T>
inline T* at(void* ws, int64_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off); }
}  // namespace

static int render_layout(int B, int N, int img_h, int img_w, int64_t capacity, int with_l1, bool projected, gol_render_ws* L) {
  GOL_REQUIRE(B >= 0 && N >= 0 && img_h > 0 && img_w > 0 && capacity >= 0 && L != nullptr, "bad argument");
  const int64_t T = (int64_t)((img_w + 15) / 16) * ((img_h + 15) / 16), P = (int64_t)img_h * img_w;
  const int64_t BN = (int64_t)B * N, cap = capacity > 0 ? capacity : 1;
  int64_t o = 0;
  auto take = [&](int64_t bytes) { const int64_t at_ = o; o = align256(o + bytes); return at_; };
  L->cov3d = -1;   // not stored: gol_project_bwd recomputes Sigma from the scales / quaternions (24 B per Gaussian less each way)
  // (projected: the projection's outputs live in the caller's gol_shade_proj buffers, written by the shading kernel)
  L->xys = projected ? -1 : take(BN * 2 * 4);
  L->depths = projected ? -1 : take(BN * 4);
  L->radii = projected ? -1 : take(BN * 4);
  L->conics = projected ? -1 : take(BN * 3 * 4);
  L->comp = projected ? -1 : take(BN * 4);
  L->nth = -1;     // gsplat's num_tiles_hit is not needed on this path
  L->opac_eff = projected ? -1 : take(BN * 4);
  L->records = projected ? -1 : take(BN * GOL_SPLAT_RECORD * 4);
  L->tile_count = take((int64_t)B * T * 4);
  L->tile_bins = take((int64_t)B * T * 2 * 4);
  L->keys = take((int64_t)B * cap * 8);
  L->sorted_ids = take((int64_t)B * cap * 4);
  L->n_isect = take((int64_t)B * 4);
  L->final_T = take((int64_t)B * P * 4);
  L->final_idx = take((int64_t)B * P * 4);
  L->l1_sign = with_l1 ? take((int64_t)B * P) : -1;
  L->total = o;
  return GOL_OK;
}

extern "C" int gol_render_layout(int B, int N, int img_h, int img_w, int64_t capacity, int with_l1, gol_render_ws* L) {
  return render_layout(B, N, img_h, img_w, capacity, with_l1, false, L);
}

extern "C" int gol_render_layout_projected(int B, int N, int img_h, int img_w, int64_t capacity, int with_l1,
                                           gol_render_ws* L) {
  return render_layout(B, N, img_h, img_w, capacity, with_l1, true, L);
}

// binning + raster of Gaussians that are already projected (by gol_shade_project_fwd, or by gol_project_fwd)
static int render_fwd_from(int B, int N, int img_h, int img_w, const float* xys, const float* depths, const int32_t* radii,
                           const float* conics, const float* opac_eff, const float* records, const float* background,
                           int with_depth, float norm_lo, int64_t capacity, void* ws, const gol_render_ws* L,
                           float* out_img, float* out_depth, float* out_alpha, float* out_depth_norm,
                           const float* l1_target, const float* l1_mask, int l1_mask_c, float* l1_partial, float* l1_out,
                           float l1_scale, void* stream) {
  int rc = gol_bin_sort(B, N, xys, depths, radii, conics, opac_eff, img_h, img_w, 16, capacity,
                        at<int32_t>(ws, L->tile_count), at<int32_t>(ws, L->tile_bins), at<uint64_t>(ws, L->keys),
                        at<int32_t>(ws, L->sorted_ids), at<int32_t>(ws, L->n_isect), nullptr, stream);
  if (rc != GOL_OK) return rc;
  return gol_rasterize_fwd(B, N, img_h, img_w, 16, 1, at<int32_t>(ws, L->tile_bins), at<int32_t>(ws, L->sorted_ids),
                           capacity, records, with_depth ? 1 : 0, background, out_img,
                           with_depth ? out_depth : nullptr, at<float>(ws, L->final_T), at<int32_t>(ws, L->final_idx),
                           out_alpha, with_depth ? out_depth_norm : nullptr, norm_lo, l1_target, l1_mask, l1_mask_c,
                           l1_target ? at<uint8_t>(ws, L->l1_sign) : nullptr, l1_partial, l1_out, l1_scale, 0,
                           stream);
}

// the raster backward into zeroed 64-byte gradient records
static int render_bwd_to_records(int B, int N, int img_h, int img_w, const float* records, const float* background,
                                 int64_t capacity, void* ws, const gol_render_ws* L, const float* v_img,
                                 const float* v_depth, const float* v_alpha, int use_l1_sign, const float* l1_mask,
                                 int l1_mask_c, const float* v_img_scale, float v_img_scale_mul, float* grad_records,
                                 void* stream) {
  hipStream_t s = (hipStream_t)stream;
  // one zeroed buffer of 64-byte gradient records per Gaussian (GOL_GRAD_RECORD): [rgb | opacity | xy | conic | depth | pad]
  if (hipMemsetAsync(grad_records, 0, sizeof(float) * (size_t)B * N * GOL_GRAD_RECORD, s) != hipSuccess) {
    gol_set_error("gol_render_bwd: hipMemsetAsync failed");
    return GOL_ERR_LAUNCH;
  }
  float* g = grad_records;
  const bool use_depth = v_depth != nullptr;
  return gol_rasterize_bwd(B, N, img_h, img_w, 16, 1, at<int32_t>(ws, L->tile_bins), at<int32_t>(ws, L->sorted_ids),
                           capacity, records, use_depth ? 1 : 0, background,
                           at<float>(ws, L->final_T), at<int32_t>(ws, L->final_idx), v_img, v_depth, v_alpha, g + 4, g + 6,
                           g, use_depth ? g + 9 : nullptr, g + 3, GOL_GRAD_RECORD,
                           use_l1_sign ? at<uint8_t>(ws, L->l1_sign) : nullptr, use_l1_sign ? l1_mask : nullptr,
                           use_l1_sign ? l1_mask_c : 0, v_img_scale, v_img_scale_mul, 0, stream);
}

extern "C" int gol_render_fwd(int B, int N, int img_h, int img_w, float glob_scale, float clip_thresh, const float* means,
                              const float* scales, const float* quats, const float* opacity, const float* colors,
                              const float* viewmats, const float* intrins, const float* background, int with_depth,
                              float norm_lo, int64_t capacity, void* workspace, const gol_render_ws* L, float* out_img,
                              float* out_depth, float* out_alpha, float* out_depth_norm, const float* l1_target,
                              const float* l1_mask, int l1_mask_c, float* l1_partial, float* l1_out, float l1_scale,
                              void* stream) {
  GOL_REQUIRE(workspace != nullptr && L != nullptr, "null workspace / layout");
  GOL_REQUIRE(!l1_target || L->l1_sign >= 0, "layout was computed without the fused L1");
  if (B == 0) return GOL_OK;
  void* ws = workspace;
  GOL_REQUIRE(!l1_out || l1_target, "l1_out needs l1_target");
  int rc = gol_project_fwd(B, N, means, scales, glob_scale, quats, viewmats, intrins, img_h, img_w, 16, clip_thresh,
                           nullptr, at<float>(ws, L->xys), at<float>(ws, L->depths),
                           at<int32_t>(ws, L->radii), at<float>(ws, L->conics), at<float>(ws, L->comp),
                           nullptr, opacity, at<float>(ws, L->opac_eff), colors,
                           at<float>(ws, L->records), stream);
  if (rc != GOL_OK) return rc;
  return render_fwd_from(B, N, img_h, img_w, at<float>(ws, L->xys), at<float>(ws, L->depths), at<int32_t>(ws, L->radii),
                         at<float>(ws, L->conics), at<float>(ws, L->opac_eff), at<float>(ws, L->records), background,
                         with_depth, norm_lo, capacity, ws, L, out_img, out_depth, out_alpha, out_depth_norm, l1_target,
                         l1_mask, l1_mask_c, l1_partial, l1_out, l1_scale, stream);
}

extern "C" int gol_render_fwd_projected(int B, int N, const gol_shade_proj* proj, const float* background, int with_depth,
                                        float norm_lo, int64_t capacity, void* workspace, const gol_render_ws* L,
                                        float* out_img, float* out_depth, float* out_alpha, float* out_depth_norm,
                                        const float* l1_target, const float* l1_mask, int l1_mask_c, float* l1_partial,
                                        float* l1_out, float l1_scale, void* stream) {
  GOL_REQUIRE(workspace != nullptr && L != nullptr && proj != nullptr, "null workspace / layout / projection");
  GOL_REQUIRE(!l1_target || L->l1_sign >= 0, "layout was computed without the fused L1");
  GOL_REQUIRE(!l1_out || l1_target, "l1_out needs l1_target");
  if (B == 0) return GOL_OK;
  return render_fwd_from(B, N, proj->img_h, proj->img_w, proj->xys, proj->depths, proj->radii, proj->conics,
                         proj->opac_eff, proj->records, background, with_depth, norm_lo, capacity, workspace, L, out_img,
                         out_depth, out_alpha, out_depth_norm, l1_target, l1_mask, l1_mask_c, l1_partial, l1_out,
                         l1_scale, stream);
}

extern "C" int gol_render_bwd(int B, int N, int img_h, int img_w, float glob_scale, const float* means,
                              const float* scales, const float* quats, const float* opacity, const float* viewmats,
                              const float* intrins, const float* background, int64_t capacity, void* workspace,
                              const gol_render_ws* L, const float* v_img, const float* v_depth, const float* v_alpha,
                              int use_l1_sign, const float* l1_mask, int l1_mask_c, const float* v_img_scale,
                              float v_img_scale_mul, float* grad_records, float* v_mean, float* v_scale, float* v_quat, float* v_opacity,
                              float* v_colors, void* stream) {
  GOL_REQUIRE(workspace != nullptr && L != nullptr && grad_records != nullptr, "null workspace / layout / gradient records");
  GOL_REQUIRE(!use_l1_sign || L->l1_sign >= 0, "layout was computed without the fused L1");
  if (B == 0 || N == 0) return GOL_OK;
  void* ws = workspace;
  const float* g = grad_records;
  const bool use_depth = v_depth != nullptr;
  int rc = render_bwd_to_records(B, N, img_h, img_w, at<float>(ws, L->records), background, capacity, ws, L, v_img, v_depth,
                                 v_alpha, use_l1_sign, l1_mask, l1_mask_c, v_img_scale, v_img_scale_mul, grad_records,
                                 stream);
  if (rc != GOL_OK) return rc;
  return gol_project_bwd_records(B, N, means, scales, glob_scale, quats, viewmats, intrins, at<int32_t>(ws, L->radii),
                                 at<float>(ws, L->conics), at<float>(ws, L->comp), opacity, g, use_depth ? 1 : 0, v_mean,
                                 v_scale, v_quat, v_opacity, v_colors, stream);
}

extern "C" int gol_render_bwd_projected(int B, int N, const gol_shade_proj* proj, const float* background, int64_t capacity,
                                        void* workspace, const gol_render_ws* L, const float* v_img, const float* v_depth,
                                        const float* v_alpha, int use_l1_sign, const float* l1_mask, int l1_mask_c,
                                        const float* v_img_scale, float v_img_scale_mul, float* grad_records,
                                        void* stream) {
  GOL_REQUIRE(workspace != nullptr && L != nullptr && proj != nullptr && grad_records != nullptr,
              "null workspace / layout / projection / gradient records");
  GOL_REQUIRE(!use_l1_sign || L->l1_sign >= 0, "layout was computed without the fused L1");
  if (B == 0 || N == 0) return GOL_OK;
  return render_bwd_to_records(B, N, proj->img_h, proj->img_w, proj->records, background, capacity, workspace, L, v_img,
                               v_depth, v_alpha, use_l1_sign, l1_mask, l1_mask_c, v_img_scale, v_img_scale_mul,
                               grad_records, stream);
}
