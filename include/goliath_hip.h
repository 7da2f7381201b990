/*
 * goliath_hip.h -- C ABI of libgoliath_hip.so, the MI355X (gfx950) implementation of the
 * Relightable-Gaussian-Codec-Avatar render hot path of facebookresearch/goliath.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 / int32 data unless it says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream);
 *   - functions only enqueue work (no allocation, no host sync) and return GOL_OK or a negative
 *     gol_status; outputs are caller-owned and pre-allocated, exactly like the reference's
 *     pybind entry points (sg.cu:177-283, mvpraymarch.cpp:107-409, utils.cpp:46-137);
 *   - "[B,N,3]" = row-major, B views (batch elements) x N Gaussians;
 *   - re-entrant given distinct streams and distinct output/scratch buffers.  Process-wide state is limited to
 *     the thread-local last-error text and a monotone per-device cache of the dynamic-LDS limits already
 *     granted to the binning kernels (atomics; a race only repeats an idempotent hipFuncSetAttribute).
 *
 * Each entry cites the reference interface it replaces (paths relative to /root/reference,
 * "gsplat:" = the third-party gsplat==0.1.11 the reference calls).
 */
#ifndef GOLIATH_HIP_H
#define GOLIATH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GOL_OK = 0,
  GOL_ERR_INVALID_ARG = -1,   /* bad size / null pointer / unsupported option */
  GOL_ERR_LAUNCH = -2,        /* hipGetLastError() != hipSuccess after a launch */
  GOL_ERR_UNSUPPORTED = -3
} gol_status;

/* Library / build identification ("goliath_hip <ver> gfx950"). */
const char* gol_version(void);
/* Text of the last error on this thread (static storage). */
const char* gol_last_error(void);
/* Self-test of the wave64 reduction primitives: in256 = 4 x 64 floats (a,b,c,d per lane);
 * out128[0..63] = per-lane result of the 4-way swap reduction (lanes 15/31/47/63 = sums of
 * a/b/c/d), out128[64..127] = per-lane result of the DPP ladder (lane 63 = sum of a). */
int gol_selftest_wave_sum4(const float* in256, float* out128, void* stream);

/* ------------------------------------------------------------------------------------------
 * sgutils -- spherical-Gaussian specular lobe evaluation.
 * Replaces sgutilslib.evaluate_gaussian_fwd / _bwd (extensions/sgutils/sg.cu:177-226, 228-278;
 * kernels sg.cu:27-76, 78-175), called from extensions/sgutils/sgutils.py:27-29, 50-62.
 *   lobe_dirs[N,D,3] lobe_sigmas[N,D] light_values[N,L,3] light_pts[N,L,3] prim_pts[N,D,3]
 *   n_lights[N] int32, w_type in {0,1,2,3}.
 * fwd writes integral[N,D,3] (every element).  bwd writes grad_dirs[N,D,3], grad_sigmas[N,D]
 * and, when grad_light_values != NULL, ACCUMULATES into grad_light_values[N,L,3] (caller zeroes).
 * ---------------------------------------------------------------------------------------- */
int gol_sg_eval_fwd(int N, int D, int L, const float* lobe_dirs, const float* lobe_sigmas,
                    const float* light_values, const float* light_pts, const float* prim_pts,
                    const int32_t* n_lights, float* integral, int w_type, void* stream);
int gol_sg_eval_bwd(int N, int D, int L, const float* lobe_dirs, const float* lobe_sigmas,
                    const float* light_values, const float* light_pts, const float* prim_pts,
                    const int32_t* n_lights, const float* grad_integral, float* grad_dirs,
                    float* grad_sigmas, float* grad_light_values, int w_type, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gaussian projection (EWA).  Replaces gsplat: project_gaussians forward/backward
 * (call site ca_code/utils/render_gsplat.py:49-63; semantics SURVEY.md A.1, A.5).
 * Batched over B views: viewmats[B,12] (row-major 3x4 world->camera), intrins[B,4]=(fx,fy,cx,cy)
 * live on the device so no host sync is needed to read K (cf. rgca.py:123-126 .item() x4).
 * Outputs for culled Gaussians are zero (radii = num_tiles_hit = 0).  cov3d and num_tiles_hit may be NULL (not written);
 * gol_project_bwd with cov3d == NULL recomputes Sigma from the scales and quaternions (the forward's own arithmetic).
 * Optional fused extras (pass NULL to skip):
 *   opacities[B,N] -> opac_eff[B,N] = opacity * compensation   (render_gsplat.py:72)
 *   colors[B,N,3] (+ opacities) -> records[B,N,GOL_SPLAT_RECORD]: the rasterizer's packed per-Gaussian records
 *   (see gol_rasterize_fwd) with the depth as the 4th channel -- the fused path never assembles them separately.
 * ---------------------------------------------------------------------------------------- */
int gol_project_fwd(int B, int N, const float* means3d, const float* scales, float glob_scale,
                    const float* quats, const float* viewmats, const float* intrins, int img_h,
                    int img_w, int block, float clip_thresh, float* cov3d, float* xys,
                    float* depths, int32_t* radii, float* conics, float* compensation,
                    int32_t* num_tiles_hit, const float* opacities, float* opac_eff, const float* colors,
                    float* records, void* stream);
/* v_* inputs may be NULL (treated as zero).  If opacities != NULL the op also differentiates
 * opac_eff = opacity*compensation: v_opac_eff[B,N] in, v_opacity[B,N] out.
 * grad_stride = 0: v_xy[B,N,2], v_depth[B,N], v_conic[B,N,3], v_opac_eff[B,N] are dense arrays;
 * grad_stride = s > 0: they are fields of per-Gaussian records of s floats (element e at pointer + e*s), e.g. the
 * GOL_GRAD_RECORD-float records gol_rasterize_bwd accumulates into. */
int gol_project_bwd(int B, int N, const float* means3d, const float* scales, float glob_scale,
                    const float* quats, const float* viewmats, const float* intrins,
                    const float* cov3d, const int32_t* radii, const float* conics,
                    const float* compensation, const float* v_xy, const float* v_depth,
                    const float* v_conic, const float* v_compensation, const float* opacities,
                    const float* v_opac_eff, int grad_stride, float* v_mean3d, float* v_scale, float* v_quat,
                    float* v_opacity, void* stream);
/* The same with its upstream gradients taken from the rasterizer's per-Gaussian gradient records (grad_records
 * [B,N,GOL_GRAD_RECORD]: rgb | opacity | xy | conic a b c | depth | pad, gol_rasterize_bwd) and cov3d recomputed; v_colors
 * [B,N,3] (may be NULL) additionally receives the records' colour gradient as a dense array -- the consumer (the shading
 * tail's backward) then reads 12 contiguous bytes per Gaussian instead of a strided view of the records. */
int gol_project_bwd_records(int B, int N, const float* means3d, const float* scales, float glob_scale,
                            const float* quats, const float* viewmats, const float* intrins, const int32_t* radii,
                            const float* conics, const float* compensation, const float* opacities,
                            const float* grad_records, int with_depth, float* v_mean3d, float* v_scale, float* v_quat,
                            float* v_opacity, float* v_colors, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tile binning + per-tile depth sort.  Replaces gsplat: cumsum + map_gaussian_to_intersects +
 * torch.sort(int64) + get_tile_bin_edges without the host sync on the
 * intersection count.  Order inside a tile: ascending (depth bits, gaussian id) -- a
 * deterministic instance of gsplat's unspecified tie order.
 *   conics[B,N,3], opacities[B,N]  optional (both or neither): when given, (Gaussian, tile) pairs
 *                   whose alpha >= 1/255 ellipse cannot reach the tile are not stored.  This is
 *                   output-preserving (the rasterizer would skip them at every pixel, A.3).
 *   capacity        max intersections stored per view
 *   tile_count[B,T] int32 scratch
 *   tile_bins[B,T,2] out: [start,end) into the view's segment of sorted_ids
 *   isect_keys[B,capacity] uint64 scratch
 *   sorted_ids[B,capacity] out: Gaussian ids, tile by tile, front to back
 *   n_isect[B] out: number of list slots the view needs (> capacity means overflow: the excess
 *                   intersections were dropped and the render is incomplete).  Without conics this is gsplat's exact
 *                   intersection count; with conics it is an upper bound (the tight tile boxes, ~1.2x the pruned
 *                   lists: the count pass reserves, the scatter pass tests) and a view's lists have slack between them
 *   reach_scratch   unused since round 3 (pass NULL; kept for ABI stability)
 * ---------------------------------------------------------------------------------------- */
int gol_bin_sort(int B, int N, const float* xys, const float* depths, const int32_t* radii,
                 const float* conics, const float* opacities, int img_h, int img_w, int block,
                 int64_t capacity, int32_t* tile_count, int32_t* tile_bins, uint64_t* isect_keys,
                 int32_t* sorted_ids, int32_t* n_isect, uint64_t* reach_scratch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tile rasterizer.  Replaces gsplat: rasterize_forward / rasterize_backward, 3-channel
 * specialisation (call sites render_gsplat.py:65-78, 91-104; semantics SURVEY.md A.3, A.4).
 * Gaussian attributes come as PACKED RECORDS records[B,N,GOL_SPLAT_RECORD] (64 bytes = one HBM sector per Gaussian):
 *   [0] x [1] y [2] a' [3] b' | [4] c' [5] opacity [6] r [7] g | [8] b [9] extra [10] tau [11] 1/a | [12] 1/c [13] exact [14-15] pad
 *   (a', b', c' = the conic scaled for the pixel loops, tau / 1/a / 1/c / exact = the per-Gaussian part of the
 *   alpha >= 1/255 reach test; goliath_amd/csrc/gol_common.h) written by gol_project_fwd (fused path) or by
 *   gol_splat_pack from xys[B,N,2], conics[B,N,3], colors[B,N,3], extra[B,N] (NULL = 0), opacities[B,N] (the
 *   gsplat-compatible operators): a list entry costs one aligned 64-byte fetch instead of up to five sectors of five arrays.
 * One launch composites the colour and, with with_extra != 0, the 4th channel "extra" of the records
 * (the depth pass of render_gsplat.py:91-104 fused into the colour pass; its background is 0).
 *   planar = 0: out_img / v_out_img are [B,H,W,3] (gsplat); planar = 1: [B,3,H,W] (what
 *   AutoEncoder.render stacks, rgca.py:139 -- saves the permute copy in the backward).
 *   background[3]; out_extra[B,H,W]; final_Ts[B,H,W]; final_idx[B,H,W] int32:
 *   planar = 0: index into the view's sorted_ids segment of the last contributing Gaussian, 0 if none (gsplat's);
 *   planar = 1: the index gol_rasterize_bwd starts the pixel's walk from -- an upper bound of the above that excludes
 *   the entry the pixel stopped at (stop index - 1; end of the tile's list for a pixel that never reached T <= 1e-4).
 * fwd optional fused epilogue of AutoEncoder.render (rgca.py:137,144-145), NULL = off: out_alpha[B,H,W] = 1 - final_T and
 *   out_extra_norm[B,H,W] = extra image / clamp(1 - final_T, norm_lo, 1)  (the reference divides depth by alpha.clamp(0.05, 1));
 *   with out_extra_norm given, out_extra (the un-normalised image) may be NULL -- one image write less.
 * bwd ACCUMULATES into v_xy[B,N,2] v_conic[B,N,3] v_colors[B,N,3] v_opacity[B,N]
 * (and v_extra[B,N]) which the caller zeroes; v_out_alpha / v_out_extra may be NULL.
 * grad_stride = 0: the five gradient outputs are dense arrays (gsplat's layout);
 * grad_stride = GOL_GRAD_RECORD: they are fields of ONE zeroed buffer rec[B,N,GOL_GRAD_RECORD] of 64-byte records
 *   [rgb 0-2 | opacity 3 | xy 4-5 | conic 6-8 | extra 9 | pad], i.e. v_colors = rec, v_opacity = rec+3, v_xy = rec+4,
 *   v_conic = rec+6, v_extra = rec+9 (checked): the float atomics of a Gaussian then hit one cache line and 16
 *   consecutive lanes issue them together -- the memory-side atomic units see 1/10 of the requests.
 * Optional fused masked L1 loss (rgb_l1, ca_code/loss/__init__.py:391-411), planar layout only: with l1_target[B,3,H,W]
 *   (l1_mask[B,l1_mask_c,H,W], l1_mask_c = 1 or 3, or NULL) the forward also writes l1_sign[B,H,W], ONE BYTE per pixel
 *   holding sign((rgb_c - target_c) * mask_c) + 1 of channel c in bits 2c..2c+1, and l1_partial[B, tiles] = per-tile sums
 *   of |(rgb - target) * mask| (tiles = ceil(W/16) * ceil(H/16); loss = sum(l1_partial) / (B*3*H*W)).
 *   l1_out (NULL = the caller adds l1_partial up): l1_out[0] = l1_scale * sum(l1_partial), by a one-workgroup kernel
 *   enqueued behind the raster launch (fixed summation order: deterministic).
 *   The backward's upstream image gradient is  v_out_img (NULL = 0)  +  (code_c - 1) * mask_c * v_img_scale[0]  of
 *   v_sign[B,H,W] (NULL = none; v_sign_mask[B,v_sign_mask_c,H,W] or NULL = 1; v_img_scale device scalar, NULL = 1, times the
 *   host scalar v_img_scale_mul), so passing v_sign = l1_sign, v_sign_mask = l1_mask, v_img_scale = d loss_total / d l1
 *   (the autograd gradient of the loss value, as it is) and v_img_scale_mul = 1 / (B*3*H*W) back-propagates the
 *   loss without the two extra passes over the image a separate loss kernel needs; at least one of v_out_img / v_sign.
 * pixels_per_lane (fwd and bwd): the wave footprint.  2 = two waves per 16x16 tile, 16x8 pixels each, two pixels per lane
 *   (fewest instructions per pixel: launches that fill the chip); 1 = four waves per tile, 8x8 pixels each (~9-17 % more
 *   instructions, but a tile's work is spread over more SIMDs: launches of one or two views, whose workgroups are all
 *   resident at once and which last until the most loaded SIMD is done);
 *   0 = the forward chooses by B (gol_raster_plan), the backward takes 2.  Same images, final_T / final_idx and gradients
 *   either way (a footprint only skips entries none of its pixels can take), up to the summation order of the gradient atomics.
 *   In the fused path (planar = 1) final_idx / l1_sign of pixels in tiles WITHOUT list entries are not written (the
 *   backward never reads them; final_T = 1 is).
 * ---------------------------------------------------------------------------------------- */
#define GOL_GRAD_RECORD 16
#define GOL_SPLAT_RECORD 16
int gol_splat_pack(int B, int N, const float* xys, const float* conics, const float* colors, const float* extra,
                   const float* opacities, float* records, void* stream);
int gol_rasterize_fwd(int B, int N, int img_h, int img_w, int block, int planar, const int32_t* tile_bins,
                      const int32_t* sorted_ids, int64_t capacity, const float* records, int with_extra,
                      const float* background, float* out_img,
                      float* out_extra, float* final_Ts, int32_t* final_idx, float* out_alpha,
                      float* out_extra_norm, float norm_lo, const float* l1_target, const float* l1_mask, int l1_mask_c,
                      uint8_t* l1_sign, float* l1_partial, float* l1_out, float l1_scale, int pixels_per_lane,
                      void* stream);
int gol_rasterize_bwd(int B, int N, int img_h, int img_w, int block, int planar, const int32_t* tile_bins,
                      const int32_t* sorted_ids, int64_t capacity, const float* records, int with_extra,
                      const float* background, const float* final_Ts,
                      const int32_t* final_idx, const float* v_out_img, const float* v_out_extra,
                      const float* v_out_alpha, float* v_xy, float* v_conic, float* v_colors,
                      float* v_extra, float* v_opacity, int grad_stride, const uint8_t* v_sign, const float* v_sign_mask,
                      int v_sign_mask_c, const float* v_img_scale, float v_img_scale_mul, int pixels_per_lane,
                      void* stream);
/* Diagnostic (bench.py's algorithmic roofline of the raster kernels): counts[B,2] (uint64) = per view the number of
 * (pixel, list entry) pairs up to the pixel's final_idx ("tested") and of those with alpha >= 1/255 ("taken" = composited),
 * from the state a planar gol_rasterize_fwd left (tile_bins, sorted_ids, records, final_idx). */
int gol_raster_count_pairs(int B, int N, int img_h, int img_w, const int32_t* tile_bins, const int32_t* sorted_ids,
                           int64_t capacity, const float* records, const int32_t* final_idx, uint64_t* counts, void* stream);
/* What pixels_per_lane = 0 means for the forward of a launch of B views (1 for B <= 2, else 2). */
int gol_raster_plan(int B, int* fwd_pixels_per_lane);

/* ------------------------------------------------------------------------------------------
 * One render direction of a batch of views as ONE call (csrc/render.hip): what AutoEncoder.render issues per view from
 * Python -- project_gaussians -> rasterize_gaussians(colour) -> rasterize_gaussians(depth), ca_code/utils/render_gsplat.py:49-104,
 * in the view loop of ca_code/models/rgca.py:112-151 -- becomes gol_render_fwd = gol_project_fwd (+ packed records) +
 * gol_bin_sort + gol_rasterize_fwd (planar images, fused alpha / depth-normalisation / optional L1 epilogue) and
 * gol_render_bwd = gol_rasterize_bwd (64-byte gradient records, zeroed here) + gol_project_bwd, enqueued on `stream` out
 * of ONE caller-provided workspace.  gol_render_layout fills the byte offsets of the workspace's sub-buffers
 * (all 256-byte aligned; `total` = bytes to allocate) -- the caller reads n_isect[B] (int32, overflow check as for
 * gol_bin_sort), radii[B,N], final_T / final_idx[B,H,W], sorted_ids[B,capacity], tile_bins[B,T,2] there.
 * The forward leaves everything the backward needs in the workspace: keep it (and the inputs) until gol_render_bwd.
 *   out_img[B,3,H,W]; out_alpha[B,H,W] = 1 - final_T; with_depth: out_depth_norm[B,H,W] = depth / clamp(alpha, norm_lo, 1)
 *   and optionally out_depth[B,H,W] (NULL = skip); l1_target / l1_mask / l1_partial[B,T] / l1_out (NULL = partial sums
 *   only) / l1_scale as for gol_rasterize_fwd.
 *   bwd: v_img[B,3,H,W] / v_depth[B,H,W] (w.r.t. the UN-normalised depth image) / v_alpha[B,H,W] may be NULL;
 *   use_l1_sign != 0 adds the fused L1's gradient (v_img_scale = device scalar d loss / d l1, v_img_scale_mul = 1 / (B*3*H*W));
 *   grad_records[B,N,GOL_GRAD_RECORD] scratch; v_colors[B,N,3] (may be NULL): d loss / d colour as a dense array (it is
 *   also the first three floats of every record).
 * ---------------------------------------------------------------------------------------- */
typedef struct gol_render_ws {
  int64_t cov3d, xys, depths, radii, conics, comp, nth, opac_eff, records, tile_count, tile_bins, keys, sorted_ids,
      n_isect, final_T, final_idx, l1_sign, total;
} gol_render_ws;
int gol_render_layout(int B, int N, int img_h, int img_w, int64_t capacity, int with_l1, gol_render_ws* layout);
int gol_render_fwd(int B, int N, int img_h, int img_w, float glob_scale, float clip_thresh, const float* means,
                   const float* scales, const float* quats, const float* opacity, const float* colors,
                   const float* viewmats, const float* intrins, const float* background, int with_depth, float norm_lo,
                   int64_t capacity, void* workspace, const gol_render_ws* layout, float* out_img, float* out_depth,
                   float* out_alpha, float* out_depth_norm, const float* l1_target, const float* l1_mask, int l1_mask_c,
                   float* l1_partial, float* l1_out, float l1_scale, void* stream);
int gol_render_bwd(int B, int N, int img_h, int img_w, float glob_scale, const float* means, const float* scales,
                   const float* quats, const float* opacity, const float* viewmats, const float* intrins,
                   const float* background, int64_t capacity, void* workspace, const gol_render_ws* layout,
                   const float* v_img, const float* v_depth, const float* v_alpha, int use_l1_sign, const float* l1_mask,
                   int l1_mask_c, const float* v_img_scale, float v_img_scale_mul, float* grad_records, float* v_mean,
                   float* v_scale, float* v_quat, float* v_opacity, float* v_colors, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused RGCA shading tail.  Replaces the chain of ATen kernels in PrimDecoder.forward after the
 * two transposed-conv decoders (ca_code/models/rgca.py:505-588, training extra :590-618), incl.
 * the specular term: point lights = evaluate_gaussian w_type 0 (extensions/sgutils/sg.cu:27-175,
 * rgca.py:559-570) or environment = dir2uv + mipmap_grid_sample (ca_code/utils/envmap.py:284-292,
 * ca_code/utils/mipmap_sampler.py:13-69, rgca.py:548-556).
 * The decoder outputs are read ONCE in their native planar NCHW layout (coalesced per channel
 * plane): no permute().view() copies of the 125-channel tensor.
 * All pointers device memory; structs themselves are host memory, read before the call returns.
 * ---------------------------------------------------------------------------------------- */
#define GOL_MAX_MIPS 8
typedef struct {
  int32_t B, N;              /* views, Gaussians per view (N = S*S)                              */
  int32_t n_color_coef;      /* SH coefficients carried per colour channel: (n_color_sh+1)^2 = 16 */
  int32_t n_mono_coef;       /* monochrome SH coefficients: (n_diff_sh+1)^2 - n_color_coef = 65   */
  const float* f_vnocond;    /* [B, 3*n_color_coef + n_mono_coef + 12, N]  (rgca.py:494-495)      */
  const float* f_vcond;      /* [B, 4, N]                                   (rgca.py:498-503)      */
  const float* postex;       /* [B, 3, N] uv position map                   (rgca.py:483-484)      */
  const float* tn;           /* [B, 3, N] normalised uv normal map          (rgca.py:488-491)      */
  const float* albedo;       /* [N, 3]                                      (rgca.py:462-464)      */
  const float* light_sh;     /* [B, 3, n_color_coef + n_mono_coef]          (rgca.py:187-191)      */
  const float* light_sh_rand;/* same shape or NULL: training-only random light (rgca.py:590-616)  */
  const float* campos;       /* [B, 3] head-relative camera position                              */
  /* specular, point lights (used when n_mips == 0) */
  int32_t L;
  const float* light_intensity; /* [B, L, 3] */
  const float* light_pos;       /* [B, L, 3] */
  const int32_t* n_lights;      /* [B]       */
  /* specular, prefiltered environment (n_mips > 0) */
  int32_t n_mips;
  const float* mips[GOL_MAX_MIPS];  /* level i: [B, 3, mip_h[i], mip_w[i]] */
  int32_t mip_h[GOL_MAX_MIPS], mip_w[GOL_MAX_MIPS];
  const float* lightrot;        /* [B, 3, 3] ROTATIONS (orthonormal rows): the lookup direction lightrot x reflect(view, normal)
                                 * must be a unit vector -- the polar angle is formed as atan2(|r_xz|, r_y) and its derivative as
                                 * -1 / |r_xz|, which equal the reference's acos(r_y) (envmap.py:289) and its derivative only
                                 * for |r| = 1; the host side checks it (shade.py) */
  /* optional: the same levels repacked by gol_envmap_pack to [B, h, w, 16] footprint records (a bilinear lookup = one
   * aligned 64-byte fetch); all levels or none (NULL) */
  const float* mips_packed[GOL_MAX_MIPS];
  float primscale_min, primscale_max; /* rgca.py:47 (0.1, 20) */
  /* 1: ONE pyramid lights all B views -- every level is [1,3,h,w] (packed: [1,h,w,16]) and only `lightrot` differs per
   * view.  This is what the reference's relight driver feeds: EnvSpinDecorator.mipmap() expands one registered pyramid over
   * the batch (ca_code/utils/light_decorator.py:96-100) and rotates the lookup, not the map (:112-118, rgca.py:548-550).
   * 0: one pyramid per view, levels [B,...] as above. */
  int32_t mips_shared;
  /* every env-map sample (value and its u / v derivatives) is multiplied by this; 0 is read as 1.  The relight driver scales
   * the registered pyramid by 2 pi norm_scale[0] per FRAME (light_decorator.py:147-149): with the factor here the unscaled
   * pyramid is packed once per environment and nothing is re-packed when the spin index changes. */
  float mips_scale;
} gol_shade_in;

typedef struct {  /* every field [B,N,k] row-major like the reference's preds (rgca.py:574-588) */
  float* color;            /* 3 */
  float* opacity;          /* 1 */
  float* primpos;          /* 3 */
  float* primqvec;         /* 4 */
  float* primscale;        /* 3 clamped */
  float* primscale_preclip;/* 3 */
  float* sigma;            /* 1 */
  float* spec_vis;         /* 1 */
  float* spec_nml;         /* 3 */
  float* spec_dnml;        /* 3 */
  float* diff_color;       /* 3 */
  float* spec_color;       /* 3 */
  float* primnmlbase;      /* 3 */
  float* color_rand;       /* 3, or NULL when light_sh_rand == NULL */
  float* diff_sum;         /* 3: sum_k sh_k * L_k before albedo (saved for the backward) */
  float* env_saved;        /* 9, optional (NULL = off), env mode only: the env-map sample (value rgb, d/du rgb, d/dv rgb)
                              saved so the backward does not repeat the 8 texel gathers per Gaussian */
} gol_shade_out;

typedef struct {  /* upstream gradients, same shapes as gol_shade_out; any may be NULL (= 0) */
  const float *color, *opacity, *primpos, *primqvec, *primscale, *primscale_preclip, *sigma, *spec_vis,
      *spec_nml, *spec_dnml, *diff_color, *spec_color, *primnmlbase, *color_rand;
} gol_shade_out_grad;

typedef struct {  /* written in full */
  float* f_vnocond;        /* [B, C, N] */
  float* f_vcond;          /* [B, 4, N] */
  float* postex;           /* [B, 3, N] */
  float* tn;               /* [B, 3, N] */
  float* albedo_per_view;  /* [B, N, 3]  (sum over B = gradient of the shared albedo) */
  float* albedo;           /* [N, 3] or NULL: that sum, written by a second small kernel of the same call */
} gol_shade_in_grad;

/* src[B,3,h,w] (the layout of EnvSpinDecorator's mip pyramid, light_decorator.py:100-140) ->
 * dst[B,h,w,16]: record (y, x) = the four taps (y,x) (y,x+1) (y+1,x) (y+1,x+1) of the bilinear footprint whose top-left
 * texel it is, each as (r, g, b, 0), taps beyond the border zero -- 64 bytes, one HBM sector per lookup (4x the memory
 * of the map; packed once per environment, the host side caches it). */
int gol_envmap_pack(int B, int h, int w, const float* src, float* dst, void* stream);
int gol_shade_fwd(const gol_shade_in* in, const gol_shade_out* out, void* stream);
/* `saved` = the gol_shade_out of the forward (reads color_rand, diff_sum, env_saved). */
int gol_shade_bwd(const gol_shade_in* in, const gol_shade_out* saved, const gol_shade_out_grad* g,
                  const gol_shade_in_grad* gin, void* stream);

/* Shading WITH the projection fused in (north_star: "streaming stages fused"; rgca.py:505-588 -> render_gsplat.py:49-63).
 * The cameras of the B views are known before the decoder runs (AutoEncoder.forward builds headrel_Rt first,
 * rgca.py:175-195), so the shading kernel projects the Gaussians it has just produced while their position, quaternion,
 * clamped scale, opacity and colour are still in registers, and writes what binning and the rasterizer read; the render
 * direction then starts at the tile count (gol_render_fwd_projected) and ends at the gradient records
 * (gol_render_bwd_projected), which gol_shade_project_bwd pushes through the projection's vjp in its own prologue.
 * gol_project_fwd / gol_project_bwd stay for the gsplat-compatible operators and for Gaussians that do not come out of
 * the shading tail.  Same arithmetic (csrc/gol_project.h) either way. */
typedef struct {
  const float* viewmats;     /* [B,12] world -> camera, row-major 3x4                                   */
  const float* intrins;      /* [B,4]  fx, fy, cx, cy                                                   */
  int32_t img_h, img_w;
  float glob_scale, clip_thresh;
  /* written by gol_shade_project_fwd, read by binning / raster / gol_shade_project_bwd: shapes of gol_project_fwd's outputs */
  float* xys;                /* [B,N,2] */
  float* depths;             /* [B,N]   */
  int32_t* radii;            /* [B,N]   */
  float* conics;             /* [B,N,3] */
  float* comp;               /* [B,N]   */
  float* opac_eff;           /* [B,N]   opacity x compensation (render_gsplat.py:72)                    */
  float* records;            /* [B,N,GOL_SPLAT_RECORD] the rasterizer's packed records (colour + depth) */
} gol_shade_proj;
int gol_shade_project_fwd(const gol_shade_in* in, const gol_shade_out* out, const gol_shade_proj* proj, void* stream);
/* The render direction of Gaussians gol_shade_project_fwd has projected: gol_render_fwd minus its first kernel,
 * gol_render_bwd minus its last (the workspace then holds only tile lists and per-pixel state:
 * gol_render_layout_projected; the layout's xys ... records offsets are -1). */
int gol_render_layout_projected(int B, int N, int img_h, int img_w, int64_t capacity, int with_l1, gol_render_ws* layout);
int gol_render_fwd_projected(int B, int N, const gol_shade_proj* proj, const float* background, int with_depth,
                             float norm_lo, int64_t capacity, void* workspace, const gol_render_ws* layout,
                             float* out_img, float* out_depth, float* out_alpha, float* out_depth_norm,
                             const float* l1_target, const float* l1_mask, int l1_mask_c, float* l1_partial, float* l1_out,
                             float l1_scale, void* stream);
int gol_render_bwd_projected(int B, int N, const gol_shade_proj* proj, const float* background, int64_t capacity,
                             void* workspace, const gol_render_ws* layout, const float* v_img, const float* v_depth,
                             const float* v_alpha, int use_l1_sign, const float* l1_mask, int l1_mask_c,
                             const float* v_img_scale, float v_img_scale_mul, float* grad_records, void* stream);
/* grad_records[B,N,GOL_GRAD_RECORD] = the output of gol_render_bwd_projected; its colour / opacity / position /
 * quaternion / scale gradients are ADDED to the matching fields of `g` (other consumers of those outputs). */
int gol_shade_project_bwd(const gol_shade_in* in, const gol_shade_out* saved, const gol_shade_out_grad* g,
                          const gol_shade_proj* proj, const float* grad_records, int with_depth,
                          const gol_shade_in_grad* gin, void* stream);

/* ------------------------------------------------------------------------------------------
 * Mixture-of-Volumetric-Primitives ray marcher + its helpers (BASELINE config 5).
 * Replaces utilslib.compute_raydirs_forward (extensions/utils/utils.cpp:46-82, kernel
 * utils_kernel.cu:11-51) and mvpraymarchlib.compute_aabb / raymarch_forward / raymarch_backward
 * (extensions/mvpraymarch/mvpraymarch.cpp:145-400; kernels mvpraymarch_subset_kernel.h:7-228,
 * bvh.cu:157-201) for the configuration every model in the reference uses: algo 0 (no warp field),
 * fixed-order BVH (implicit heap, children 2i+1 / 2i+2), channels-last template, additive
 * accumulation.  Unlike the reference (stream 0, no device guard) everything runs on `stream`.
 *   raypos/raydir[N,H,W,3] tminmax[N,H,W,2] primpos[N,K,3] primrot[N,K,3,3] primscale[N,K,3]
 *   tplate[N,K,TD,TH,TW,4] nodeaabb[N,2K-1,2,3]
 * march_fwd writes rayrgba[N,H,W,4] (every element) and, when non-NULL, raysat[N,H,W,3]; when
 * shadow[N,K,TD,TH,TW,2] is non-NULL it is ACCUMULATED into (primsplatter.h:29-36).
 * march_bwd ACCUMULATES into grad_primpos/rot/scale and grad_tplate (caller zeroes them,
 * mvpraymarch.py:258-264).
 * ---------------------------------------------------------------------------------------- */
int gol_raydirs_fwd(int N, int H, int W, const float* viewpos, const float* viewrot, const float* focal,
                    const float* princpt, const float* pixelcoords /* NULL = (w,h) grid */, float volradius,
                    float* raypos, float* raydir, float* tminmax, void* stream);
int gol_mvp_aabb(int N, int K, const float* primpos, const float* primrot, const float* primscale,
                 float* nodeaabb, void* stream);
int gol_mvp_march_fwd(int N, int H, int W, int K, const float* raypos, const float* raydir, float stepsize,
                      const float* tminmax, const float* nodeaabb, const float* primpos,
                      const float* primrot, const float* primscale, const float* tplate, int TD, int TH,
                      int TW, float fadescale, float fadeexp, float* rayrgba, float* raysat, float* shadow,
                      void* stream);
int gol_mvp_march_bwd(int N, int H, int W, int K, const float* raypos, const float* raydir, float stepsize,
                      const float* tminmax, const float* nodeaabb, const float* primpos,
                      const float* primrot, const float* primscale, const float* tplate, int TD, int TH,
                      int TW, float fadescale, float fadeexp, const float* raysat, const float* grad_rayrgba,
                      float* grad_primpos, float* grad_primrot, float* grad_primscale, float* grad_tplate,
                      void* stream);

/* algo 1 of the reference operator (mvpraymarch_kernel.cu:92-97, PrimSamplerTW<true>, primsampler.h:53-61, 83-87): every
 * box carries a warp field warp[N,K,WD,WH,WW,3] (channels-last, the layout raymarch_forward receives with chlast = true,
 * mvpraymarch.py:686); a sample at box position y0 reads its template at y1 = trilinear(warp_k, y0) -- y1 may leave the
 * box: missing corners read 0 -- while the fade term stays a function of y0.  Backward: additionally grad_warp
 * [N,K,WD,WH,WW,3] (ACCUMULATED, caller zeroes) and d y1 / d y0 in the chain to the box transform.  The gradient chain is
 * the one of the reference's own PyTorch fixture (mvpraymarch.py:603-626, pinned by tests/golden/mvp_golden.npz set "w");
 * the CUDA sampler hands the already warped position to its backward (primsampler.h:64 overwrites y0, :71-86 use it),
 * which differs from that fixture as soon as the warp is not the identity.  No model of the reference emits a warp. */
int gol_mvp_march_warp_fwd(int N, int H, int W, int K, const float* raypos, const float* raydir, float stepsize,
                           const float* tminmax, const float* nodeaabb, const float* primpos,
                           const float* primrot, const float* primscale, const float* tplate, int TD, int TH,
                           int TW, const float* warp, int WD, int WH, int WW, float fadescale, float fadeexp,
                           float* rayrgba, float* raysat, float* shadow, void* stream);
int gol_mvp_march_warp_bwd(int N, int H, int W, int K, const float* raypos, const float* raydir, float stepsize,
                           const float* tminmax, const float* nodeaabb, const float* primpos,
                           const float* primrot, const float* primscale, const float* tplate, int TD, int TH,
                           int TW, const float* warp, int WD, int WH, int WW, float fadescale, float fadeexp,
                           const float* raysat, const float* grad_rayrgba, float* grad_primpos, float* grad_primrot,
                           float* grad_primscale, float* grad_tplate, float* grad_warp, void* stream);

/* Light-batched shadow march of the teacher model (ca_code/models/hand_teacher_mvp.py:271-358, no_grad): N = B*L ray
 * images (L = `group` lights per frame, consecutive), but the primitive transforms, the AABB tree and the template exist
 * once per FRAME: primpos/primrot/primscale [B,K,.], nodeaabb [B,2K-1,2,3], tplate [B,K,TD,TH,TW,4] -- or, with
 * alpha_only, [B,K,TD,TH,TW] (the reference fills the colour channels with the constant 255 and never looks at the
 * marched colour).  The reference materialises L copies of all of them (expand().reshape(), :273-349).
 * shadow [N,K,TD,TH,TW,2] is ACCUMULATED (caller zeroes), exactly like raymarch_forward's `shadow` argument;
 * rayrgba [N,H,W,4] may be NULL. */
int gol_mvp_shadow_march(int N, int group, int H, int W, int K, const float* raypos, const float* raydir,
                         float stepsize, const float* tminmax, const float* nodeaabb, const float* primpos,
                         const float* primrot, const float* primscale, const float* tplate, int alpha_only,
                         int TD, int TH, int TW, float fadescale, float fadeexp, float* rayrgba, float* shadow,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * URHand per-texel-per-light UV feature loops (BASELINE config 4).  Replace the broadcast PyTorch
 * expressions of ConvTeacherDecoder.forward: ca_code/models/urhand.py:419-445 (Lambert + Phong^p
 * features) and :508-567 (GGX/Schlick features + physically based texture).  All images planar:
 *   p_uv[B,3,HW] nml[B,3,HW] cam_pos[B,3] light_pos[B,L,3] light_intensity[B,L]
 *   shadow_map[B,L,HW] or NULL; GGX only: roughness[B,HW] tex_mean[B,3,HW] (0..255), fresnel
 *   pow[n_pow] = spec_powers (urhand.py:277: 1, 16, 32)
 * phong: diff[B,HW] = diff_feature_raw, spec[B,n_pow,HW] = spec_feature_raw.
 * ggx:   feat[B,1+n_pow,HW] = feat_p (urhand.py:558), rgb[B,3,HW] = phys_tex before the global
 *        scale of :567.  Backward writes g_* in full (no accumulation); lights, camera and the
 *        shadow map receive no gradient (data / computed under no_grad in the reference).
 * ---------------------------------------------------------------------------------------- */
#define GOL_UV_MAX_POW 4
typedef struct {
  int32_t B, L, HW, n_pow;
  float pow[GOL_UV_MAX_POW];
  float fresnel;
  const float *p_uv, *nml, *cam_pos, *light_pos, *light_intensity, *shadow_map, *roughness, *tex_mean;
} gol_uvlight_in;
int gol_uvlight_phong_fwd(const gol_uvlight_in* in, float* diff, float* spec, void* stream);
int gol_uvlight_phong_bwd(const gol_uvlight_in* in, const float* u_diff, const float* u_spec, float* g_p_uv,
                          float* g_nml, void* stream);
int gol_uvlight_ggx_fwd(const gol_uvlight_in* in, float* feat, float* rgb, void* stream);
int gol_uvlight_ggx_bwd(const gol_uvlight_in* in, const float* u_feat, const float* u_rgb, float* g_p_uv,
                        float* g_nml, float* g_roughness, float* g_tex, void* stream);

/* ------------------------------------------------------------------------------------------
 * Masked L1 image loss ("next" row).  Replaces the ATen chain of rgb_l1
 * (ca_code/loss/__init__.py:391-411): ((pred - target) * mask).abs().mean().
 *   pred, target [B,C,HW]; mask NULL, [B,1,HW] (mask_c = 1) or [B,C,HW] (mask_c = C)
 * fwd writes partial[B*C*gol_l1_blocks(HW)] per-workgroup sums (loss = sum(partial)/(B*C*HW));
 * bwd writes g_pred = sign((pred-target)*mask) * mask * g_loss[0] / (B*C*HW)  (g_loss: device scalar).
 * ---------------------------------------------------------------------------------------- */
int gol_l1_blocks(int HW);
int gol_l1_fwd(int B, int C, int HW, int mask_c, const float* pred, const float* target, const float* mask,
               float* partial, void* stream);
int gol_l1_bwd(int B, int C, int HW, int mask_c, const float* pred, const float* target, const float* mask,
               const float* g_loss, float* g_pred, void* stream);

/* ------------------------------------------------------------------------------------------
 * Light-contracted last decoder layer ("next" row).  Replaces
 *   ConvTranspose2dWNUB(16 -> C_out, 4, 2, 1)  (ca_code/models/rgca.py:427,455; ca_code/nn/layers.py:331-397)
 * followed by the SH contraction of rgca.py:506-514,528-530, for weights the host has already contracted
 * with the light (goliath_amd/tail.py):
 *   out[b,ch,oy,ox] = sum_{ci,ky,kx} x[b,ci,iy,ix] weff[b',ci,ch,ky,kx]   (oy = 2 iy - 1 + ky, b' = wB==1 ? 0 : b)
 *                     + (ch < E ? sum_k lc[k,b,ch] bias[k,oy,ox] : bias[nd + ch - E, oy, ox])
 *   x [B,16,h,w]; weff [wB,16,CH,4,4]; lc [nd,B,E] (E = 0|3|6 contracted channels, nd SH planes; plane-major so a plane's B*E
 *   coefficients are one contiguous scalar load);
 *   bias [nd + CH - E, 2h, 2w]; out [B,CH,2h,2w].
 * bwd (any of g_x / g_weff / g_bias may be NULL = not wanted):
 *   g_x [B,16,h,w] written; needs weff_t = weff permuted to [wB,CH,4,4,16];
 *   g_weff [wB,16,CH,4,4] ACCUMULATES (caller zeroes), fp32 MFMA; needs x; w_scratch (optional, see below) avoids atomics;
 *   g_bias [nd + CH - E, 2h, 2w] written.
 * ---------------------------------------------------------------------------------------- */
int gol_tail_conv_fwd(int B, int Ci, int h, int w, int CH, int E, int nd, int wB, const float* x, const float* weff,
                      const float* lc, const float* bias, float* out, void* stream);
int gol_tail_conv_bwd(int B, int Ci, int h, int w, int CH, int E, int nd, int wB, const float* x, const float* weff_t,
                      const float* lc, const float* g_out, float* g_x, float* g_weff, float* g_bias, float* w_scratch,
                      void* stream);
/* floats of optional scratch for g_weff: with it the per-workgroup partial sums are written out and reduced by a second
 * kernel instead of being added with float atomics (NULL = atomics). */
long long gol_tail_conv_bwd_scratch_floats(int B, int h, int w, int CH);

/* ------------------------------------------------------------------------------------------
 * SSIM image loss ("next" row).  Replaces `ssim` / `_ssim`
 * (ca_code/utils/ssim.py:25-65) as used by rgb_ssim (ca_code/loss/__init__.py:478-494): 11x11 Gaussian
 * window (sigma 1.5, zero padding), C1 = 0.01^2, C2 = 0.03^2, masked mean.
 *   img1 (target), img2 (prediction: the differentiated argument) [B,C,H,W]; mask NULL, [B,1,H,W] or [B,C,H,W]
 * fwd writes partial[B*C*gol_ssim_blocks(H,W)] = per-workgroup sums of ssim_map*mask (caller: sum / denominator)
 *     and, when dmap != NULL, dmap[3,B,C,H,W] = mask * d ssim / d (mu2, E[img2^2], E[img1*img2]) for the backward;
 * bwd writes g_img2 = g_scale[0] * d(sum of ssim_map*mask)/d img2   (g_scale: device scalar, e.g. -g_loss/denominator).
 * ---------------------------------------------------------------------------------------- */
int gol_ssim_blocks(int H, int W);
int gol_ssim_fwd(int B, int C, int H, int W, int mask_c, const float* img1, const float* img2, const float* mask,
                 float* partial, float* dmap, void* stream);
int gol_ssim_bwd(int B, int C, int H, int W, const float* img1, const float* img2, const float* dmap,
                 const float* g_scale, float* g_img2, void* stream);

/* ------------------------------------------------------------------------------------------
 * URHand shadow-map lookup with 3x3 PCF.  Replaces the per-texel part of get_shadow_map
 * (ca_code/utils/shadowmap.py:30-96; projection ca_code/utils/geom.py:599-631): forward only (the reference
 * calls it under no_grad, ca_code/models/urhand.py:404,492).
 *   depth [B*L,dh,dw] light-camera depth images (0 = empty; from the mesh rasteriser), Rt [B*L,3,4] = [R | t] with
 *   p_cam = R p + t, intrinsics fx,fy,cx,cy (the reference uses 1000,1000,dw/2,dh/2), postex [B,3,H,W] texel
 *   positions, nml [B,3,H,W] or NULL (no back-face term) -> out [B*L,1,H,W] = in_shadow, or exp(-in_shadow/exp_scale)
 *   when exp_scale > 0 (urhand.py:416 uses 8).
 * ---------------------------------------------------------------------------------------- */
int gol_shadow_pcf(int B, int L, int H, int W, int dh, int dw, const float* depth, const float* Rt, float fx, float fy,
                   float cx, float cy, const float* postex, const float* nml, float exp_scale, float* out,
                   void* stream);

/* ------------------------------------------------------------------------------------------
 * Image tail of AutoEncoder.forward between the rasterizer and the losses, one pass each way over
 * rgb[B,3,H,W]:  out = blur( cal(rgb) + (1 - alpha) * bg * bg_scale )
 *   cal      CalV5.forward, ca_code/nn/color_cal.py:211-241, as cal_M[B,3,3] / cal_b[B,3] (both NULL = identity):
 *            cal(rgb)[c] = sum_j cal_M[b,c,j] * rgb[j] + cal_b[b,c]  (grey cameras: three identical rows)
 *   alpha[B,H,W], bg[B,3,H,W] (both NULL = no composite), bg_scale[B] or NULL: ca_code/models/rgca.py:226-230
 *   blur_w[B,3] (NULL = no blur): LearnableBlur.forward, ca_code/nn/dof_cal.py:44-56, softmaxed weights of
 *            (identity, gaussian_blur 3x3, gaussian_blur 7x7), torchvision semantics (reflect padding, H, W >= 4)
 * bwd: g_rgb[B,3,H,W] is written; `partials` (gol_imgtail_partial_floats(B,H,W) floats, [B, tiles, 16]) receives
 *   per-workgroup partial sums of the parameter gradients -- columns 0..2 blur_w, 3..5 cal_b, 6..14 cal_M (row-major)
 *   -- to be summed over `tiles` by the caller (no float atomics).  No gradient to alpha / bg (alpha is detached in the
 *   reference, rgca.py:137).
 * ---------------------------------------------------------------------------------------- */
int64_t gol_imgtail_partial_floats(int B, int H, int W);
int gol_imgtail_fwd(int B, int H, int W, const float* rgb, const float* alpha, const float* bg,
                    const float* bg_scale, const float* cal_M, const float* cal_b, const float* blur_w,
                    float* out, void* stream);
int gol_imgtail_bwd(int B, int H, int W, const float* rgb, const float* alpha, const float* bg,
                    const float* bg_scale, const float* cal_M, const float* cal_b, const float* blur_w,
                    const float* g_out, float* g_rgb, float* partials, void* stream);

/* ------------------------------------------------------------------------------------------
 * Triangle-mesh z-buffer rasterizer: the index / depth / barycentric images the reference obtains from the third-party
 * drtk (drtk.rasterize + drtk.render, ca_code/utils/render_drtk.py:44-46); consumer on the hot path: the per-light depth
 * render of the shadow map (ca_code/utils/shadowmap.py:39-50).  Forward only.
 *   v_pix[B,V,3] = (pixel x, pixel y, camera z) per vertex, vi[F,3] int32 vertex indices (shared by the B meshes)
 *   index_img[B,H,W] int32: nearest covering face, -1 = none;  depth_img[B,H,W]: its depth, 0 = none;
 *   bary_img[B,3,H,W] (may be NULL): perspective-correct barycentrics of the sample.
 * Conventions (drtk's source is not in the reference tree): sample at the pixel centre (j + 0.5, i + 0.5); coverage =
 * all edge functions >= 0, either winding; faces with a vertex at z <= 0 are skipped; depth ties -> lower face index.
 * workspace: gol_mesh_raster_workspace_bytes(B, F) bytes of device scratch (face records, packed tile bounds, the
 * per-view tile boxes and their in-box tile prefix).  The images are cleared by a streaming fill; only tiles inside a
 * view's mesh box are rasterized, by a fixed grid of workgroups that strides over them.
 * ---------------------------------------------------------------------------------------- */
int64_t gol_mesh_raster_workspace_bytes(int B, int F);
int gol_mesh_raster(int B, int V, int F, int H, int W, const float* v_pix, const int32_t* vi, int32_t* index_img,
                    float* depth_img, float* bary_img, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GOLIATH_HIP_H */
