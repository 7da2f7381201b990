/*
 * oracle/gsplat_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).
 *
 * Scalar fp32 restatement of the Gaussian-splat rasterizer the reference calls
 * through the third-party package `gsplat`, pinned `gsplat=0.1.11`
 * (/root/reference/requirements.txt:7).  Call sites in the reference:
 *   ca_code/utils/render_gsplat.py:49-63   project_gaussians(...)
 *   ca_code/utils/render_gsplat.py:65-78   rasterize_gaussians(colour)
 *   ca_code/utils/render_gsplat.py:91-104  rasterize_gaussians(depth as 3-ch colour)
 * The gsplat source is NOT under /root/reference and cannot be installed here
 * (no network), and the reference holds no test or golden vector for it:
 *
 *        ****  PARITY UNPINNED  ****
 *
 * The algorithm below restates the published gsplat v0.1.11 CUDA kernels
 * (gsplat/cuda/csrc/{forward.cu,backward.cu,helpers.cuh}) as summarised in
 * SURVEY.md Appendix A.1-A.6; every constant is a named macro so a later
 * correction is one line.  The explicit backward is cross-checked against torch
 * autograd of an independent restatement in tests/test_oracle_gsplat.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* The arithmetic type is a parameter: the library carries the fp32 entry points orc_* (the oracle proper, the
 * reference computes in fp32) and, compiled a second time with -DORC_FP64, fp64 twins orc64_* of the projection and
 * rasterization passes.  The twins exist for ONE purpose: tests/test_oracle_gsplat_fd.py differentiates the forward
 * numerically in fp64 to show that the hand-written backward is the derivative of the forward (away from the named
 * upstream quirks) -- the best available check of a restatement whose upstream source cannot be run here. */
#ifdef ORC_FP64
typedef double real;
#define ORC(name) orc64_##name
#define R_SQRT sqrt
#define R_EXP exp
#define R_MIN fmin
#define R_MAX fmax
#define R_CEIL ceil
#else
typedef float real;
#define ORC(name) orc_##name
#define R_SQRT sqrtf
#define R_EXP expf
#define R_MIN fminf
#define R_MAX fmaxf
#define R_CEIL ceilf
#endif

/* ---- constants of gsplat 0.1.11 ------------------------- */
#define ORC_BLUR          0.3f      /* added to cov2d diagonal            (A.1) */
#define ORC_FOV_CLAMP     1.3f      /* lim = 1.3 * tan_fov                (A.1) */
#define ORC_EIG_FLOOR     0.1f      /* max(0.1, b*b - det)                (A.1) */
#define ORC_RADIUS_SIGMAS 3.0f      /* radius = ceil(3 sqrt(lambda_max))  (A.1) */
#define ORC_Z_EPS         1e-6f     /* 1/(z + 1e-6) in project_pix        (A.1) */
#define ORC_ALPHA_FLOOR   (1.f / 255.f) /*                                (A.3) */
#define ORC_T_STOP        1e-4f     /*                                    (A.3) */
#define ORC_COMP_EPS      1e-6f     /* 0.5 / (comp + 1e-6) in the comp vjp (A.5) */

typedef struct { real m[9]; } mat3; /* row-major m[r*3+c] */

static mat3 mat3_mul(mat3 a, mat3 b) {
  mat3 o;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      real s = 0.f;
      for (int k = 0; k < 3; ++k) s += a.m[r * 3 + k] * b.m[k * 3 + c];
      o.m[r * 3 + c] = s;
    }
  return o;
}
static mat3 mat3_T(mat3 a) {
  mat3 o;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) o.m[r * 3 + c] = a.m[c * 3 + r];
  return o;
}

/* A.1: quaternion (w,x,y,z), normalised inside. */
static mat3 quat_to_rotmat(const real* q) {
  real s = 1.f / R_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  real w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
  mat3 R = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z), 2.f * (x * z + w * y),
             2.f * (x * y + w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x),
             2.f * (x * z - w * y), 2.f * (y * z + w * x), 1.f - 2.f * (x * x + y * y)}};
  return R;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* A.1 tile bbox, C (int) truncation; tiles are [min,max) in tile units. */
static void tile_bbox(real cx, real cy, real radius, int tiles_x, int tiles_y, int block,
                      int* x0, int* x1, int* y0, int* y1) {
  real tcx = cx / (real)block, tcy = cy / (real)block, tr = radius / (real)block;
  *x0 = clampi((int)(tcx - tr), 0, tiles_x);
  *x1 = clampi((int)(tcx + tr + 1.f), 0, tiles_x);
  *y0 = clampi((int)(tcy - tr), 0, tiles_y);
  *y1 = clampi((int)(tcy + tr + 1.f), 0, tiles_y);
}

/* ------------------------------------------------------------------------ *
 * project_gaussians forward (A.1).  viewmat: first 12 floats, row-major 3x4.
 * Outputs are zero for culled Gaussians (the caller passes zeroed buffers,
 * like gsplat's torch.zeros allocations).
 * ------------------------------------------------------------------------ */
void ORC(project_fwd)(int N, const real* means, const real* scales, real glob_scale,
                     const real* quats, const real* viewmat, real fx, real fy, real cx,
                     real cy, int img_h, int img_w, int block, real clip_thresh,
                     real* cov3d, real* xys, real* depths, int32_t* radii, real* conics,
                     real* compensation, int32_t* num_tiles_hit) {
  const int tiles_x = (img_w + block - 1) / block, tiles_y = (img_h + block - 1) / block;
  const real* V = viewmat;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i) {
    radii[i] = 0;
    num_tiles_hit[i] = 0;
    const real* p = means + 3 * i;
    real tx = V[0] * p[0] + V[1] * p[1] + V[2] * p[2] + V[3];
    real ty = V[4] * p[0] + V[5] * p[1] + V[6] * p[2] + V[7];
    real tz = V[8] * p[0] + V[9] * p[1] + V[10] * p[2] + V[11];
    if (tz <= clip_thresh) continue;

    mat3 R = quat_to_rotmat(quats + 4 * i);
    mat3 M = R;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M.m[r * 3 + c] = R.m[r * 3 + c] * (glob_scale * scales[3 * i + c]);
    mat3 S3 = mat3_mul(M, mat3_T(M));
    real* c3 = cov3d + 6 * i;
    c3[0] = S3.m[0]; c3[1] = S3.m[1]; c3[2] = S3.m[2];
    c3[3] = S3.m[4]; c3[4] = S3.m[5]; c3[5] = S3.m[8];

    /* EWA */
    real tan_fovx = 0.5f * (real)img_w / fx, tan_fovy = 0.5f * (real)img_h / fy;
    real lim_x = ORC_FOV_CLAMP * tan_fovx, lim_y = ORC_FOV_CLAMP * tan_fovy;
    real ex = tz * R_MIN(lim_x, R_MAX(-lim_x, tx / tz));
    real ey = tz * R_MIN(lim_y, R_MAX(-lim_y, ty / tz));
    real rz = 1.f / tz, rz2 = rz * rz;
    mat3 J = {{fx * rz, 0.f, -fx * ex * rz2, 0.f, fy * rz, -fy * ey * rz2, 0.f, 0.f, 0.f}};
    mat3 Wm = {{V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]}};
    mat3 T = mat3_mul(J, Wm);
    mat3 Vc = {{c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]}};
    mat3 cov = mat3_mul(mat3_mul(T, Vc), mat3_T(T));
    real c00 = cov.m[0], c01 = cov.m[1], c11 = cov.m[4];
    real det_orig = c00 * c11 - c01 * c01;
    real a = c00 + ORC_BLUR, b = c01, c = c11 + ORC_BLUR;
    real det = a * c - b * b;
    real comp = R_SQRT(R_MAX(0.f, det_orig / det));
    if (det == 0.f) continue;
    real inv_det = 1.f / det;
    real con0 = c * inv_det, con1 = -b * inv_det, con2 = a * inv_det;
    real bb = 0.5f * (a + c);
    real sq = R_SQRT(R_MAX(ORC_EIG_FLOOR, bb * bb - det));
    real v1 = bb + sq, v2 = bb - sq;
    real radius = R_CEIL(ORC_RADIUS_SIGMAS * R_SQRT(R_MAX(v1, v2)));
    conics[3 * i + 0] = con0; conics[3 * i + 1] = con1; conics[3 * i + 2] = con2;

    real rw = 1.f / (tz + ORC_Z_EPS);
    real px = fx * (tx * rw) + cx, py = fy * (ty * rw) + cy;
    int x0, x1, y0, y1;
    tile_bbox(px, py, radius, tiles_x, tiles_y, block, &x0, &x1, &y0, &y1);
    int area = (x1 - x0) * (y1 - y0);
    if (area <= 0) continue;
    num_tiles_hit[i] = area;
    depths[i] = tz;
    radii[i] = (int)radius;
    xys[2 * i] = px; xys[2 * i + 1] = py;
    compensation[i] = comp;
  }
}

#ifndef ORC_FP64
/* ------------------------------------------------------------------------ *
 * Binning (A.2): keys (tile_id<<32)|depth_bits, sorted; ties broken by
 * Gaussian id (a deterministic instance of gsplat's unspecified tie order).
 * Returns the number of intersections written (== sum(num_tiles_hit)).
 * ------------------------------------------------------------------------ */
typedef struct { int64_t key; int32_t id; } isect_t;
static int isect_cmp(const void* a, const void* b) {
  const isect_t* x = (const isect_t*)a; const isect_t* y = (const isect_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return (x->id > y->id) - (x->id < y->id);
}

int64_t orc_bin_sort(int N, const float* xys, const float* depths, const int32_t* radii,
                     int img_h, int img_w, int block, int64_t capacity, int64_t* isect_ids_sorted,
                     int32_t* gaussian_ids_sorted, int32_t* tile_bins /* [tiles,2] zeroed */) {
  const int tiles_x = (img_w + block - 1) / block, tiles_y = (img_h + block - 1) / block;
  isect_t* buf = (isect_t*)malloc(sizeof(isect_t) * (size_t)(capacity > 0 ? capacity : 1));
  int64_t n = 0;
  for (int i = 0; i < N; ++i) {
    if (radii[i] <= 0) continue;
    int x0, x1, y0, y1;
    tile_bbox(xys[2 * i], xys[2 * i + 1], (float)radii[i], tiles_x, tiles_y, block, &x0, &x1, &y0, &y1);
    int32_t dbits; memcpy(&dbits, depths + i, 4);
    for (int ty = y0; ty < y1; ++ty)
      for (int tx = x0; tx < x1; ++tx) {
        if (n < capacity) {
          int64_t tile = (int64_t)ty * tiles_x + tx;
          buf[n].key = (tile << 32) | (int64_t)(uint32_t)dbits;
          buf[n].id = i;
        }
        ++n;
      }
  }
  int64_t m = n < capacity ? n : capacity;
  qsort(buf, (size_t)m, sizeof(isect_t), isect_cmp);
  memset(tile_bins, 0, sizeof(int32_t) * 2 * (size_t)tiles_x * tiles_y);
  for (int64_t k = 0; k < m; ++k) {
    isect_ids_sorted[k] = buf[k].key;
    gaussian_ids_sorted[k] = buf[k].id;
    int32_t tile = (int32_t)(buf[k].key >> 32);
    if (k == 0 || (int32_t)(buf[k - 1].key >> 32) != tile) tile_bins[2 * tile] = (int32_t)k;
    if (k == m - 1 || (int32_t)(buf[k + 1].key >> 32) != tile) tile_bins[2 * tile + 1] = (int32_t)(k + 1);
  }
  free(buf);
  return n;
}

#endif /* !ORC_FP64 */

/* ------------------------------------------------------------------------ *
 * rasterize forward (A.3), C channels.  alpha_cap = 0.999 in gsplat 0.1.11.
 * final_idx is the absolute index into the sorted list (0 when none).
 * ------------------------------------------------------------------------ */
void ORC(rasterize_fwd)(int img_h, int img_w, int block, int C, const int32_t* ids_sorted,
                       const int32_t* tile_bins, const real* xys, const real* conics,
                       const real* colors, const real* opacities, const real* background,
                       real alpha_cap, real* out_img, real* final_Ts, int32_t* final_idx) {
  const int tiles_x = (img_w + block - 1) / block;
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < img_h; ++i) {
    real pix[16];
    for (int j = 0; j < img_w; ++j) {
      int tile = (i / block) * tiles_x + (j / block);
      int lo = tile_bins[2 * tile], hi = tile_bins[2 * tile + 1];
      real px = (real)j + 0.5f, py = (real)i + 0.5f;
      real T = 1.f; int cur = 0;
      for (int c = 0; c < C; ++c) pix[c] = 0.f;
      for (int k = lo; k < hi; ++k) {
        int g = ids_sorted[k];
        real dx = xys[2 * g] - px, dy = xys[2 * g + 1] - py;
        real ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
        real sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
        real alpha = R_MIN(alpha_cap, opacities[g] * R_EXP(-sigma));
        if (sigma < 0.f || alpha < ORC_ALPHA_FLOOR) continue;
        real next_T = T * (1.f - alpha);
        if (next_T <= ORC_T_STOP) break;
        real vis = alpha * T;
        for (int c = 0; c < C; ++c) pix[c] += colors[(size_t)C * g + c] * vis;
        T = next_T; cur = k;
      }
      size_t p = (size_t)i * img_w + j;
      final_Ts[p] = T; final_idx[p] = cur;
      for (int c = 0; c < C; ++c) out_img[p * C + c] = pix[c] + T * background[c];
    }
  }
}

/* ------------------------------------------------------------------------ *
 * rasterize backward (A.4).  alpha_cap_bwd = 0.99 in gsplat 0.1.11 (the
 * forward uses 0.999; upstream quirk).  Gradients accumulate in
 * double per Gaussian to give an order-independent reference sum.
 * ------------------------------------------------------------------------ */
void ORC(rasterize_bwd)(int img_h, int img_w, int block, int C, int N, const int32_t* ids_sorted,
                       const int32_t* tile_bins, const real* xys, const real* conics,
                       const real* colors, const real* opacities, const real* background,
                       const real* final_Ts, const int32_t* final_idx, const real* v_out,
                       const real* v_out_alpha, real alpha_cap_bwd, real* v_xy, real* v_conic,
                       real* v_colors, real* v_opacity) {
  const int tiles_x = (img_w + block - 1) / block;
  double* axy = (double*)calloc((size_t)N * 2, sizeof(double));
  double* aco = (double*)calloc((size_t)N * 3, sizeof(double));
  double* acl = (double*)calloc((size_t)N * C, sizeof(double));
  double* aop = (double*)calloc((size_t)N, sizeof(double));
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < img_h; ++i) {
    real buffer[16];
    for (int j = 0; j < img_w; ++j) {
      int tile = (i / block) * tiles_x + (j / block);
      int lo = tile_bins[2 * tile], hi = tile_bins[2 * tile + 1];
      if (hi <= lo) continue;
      size_t p = (size_t)i * img_w + j;
      real px = (real)j + 0.5f, py = (real)i + 0.5f;
      real T_final = final_Ts[p], T = T_final;
      int bin_final = final_idx[p];
      const real* vo = v_out + p * C;
      real voa = v_out_alpha ? v_out_alpha[p] : 0.f;
      for (int c = 0; c < C; ++c) buffer[c] = 0.f;
      int start = bin_final < hi - 1 ? bin_final : hi - 1;
      for (int k = start; k >= lo; --k) {
        int g = ids_sorted[k];
        real dx = xys[2 * g] - px, dy = xys[2 * g + 1] - py;
        real ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
        real opac = opacities[g];
        real sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
        real vis = R_EXP(-sigma);
        real alpha = R_MIN(alpha_cap_bwd, opac * vis);
        if (sigma < 0.f || alpha < ORC_ALPHA_FLOOR) continue;
        real ra = 1.f / (1.f - alpha);
        T *= ra;
        real fac = alpha * T;
        real v_alpha = 0.f;
        for (int c = 0; c < C; ++c) {
          real col = colors[(size_t)C * g + c];
          
#pragma omp atomic
          acl[(size_t)C * g + c] += (double)(fac * vo[c]);
          v_alpha += (col * T - buffer[c] * ra) * vo[c];
          v_alpha += -T_final * ra * background[c] * vo[c];
          buffer[c] += col * fac;
        }
        v_alpha += T_final * ra * voa;
        real v_sigma = -opac * vis * v_alpha;
        
#pragma omp atomic
          aco[3 * g + 0] += (double)(0.5f * v_sigma * dx * dx);
        
#pragma omp atomic
          aco[3 * g + 1] += (double)(v_sigma * dx * dy);
        
#pragma omp atomic
          aco[3 * g + 2] += (double)(0.5f * v_sigma * dy * dy);
        
#pragma omp atomic
          axy[2 * g + 0] += (double)(v_sigma * (ca * dx + cb * dy));
        
#pragma omp atomic
          axy[2 * g + 1] += (double)(v_sigma * (cb * dx + cc * dy));
        
#pragma omp atomic
          aop[g] += (double)(vis * v_alpha);
      }
    }
  }
  for (int g = 0; g < N; ++g) {
    v_xy[2 * g] = (real)axy[2 * g]; v_xy[2 * g + 1] = (real)axy[2 * g + 1];
    for (int c = 0; c < 3; ++c) v_conic[3 * g + c] = (real)aco[3 * g + c];
    for (int c = 0; c < C; ++c) v_colors[(size_t)C * g + c] = (real)acl[(size_t)C * g + c];
    v_opacity[g] = (real)aop[g];
  }
  free(axy); free(aco); free(acl); free(aop);
}

/* ------------------------------------------------------------------------ *
 * project_gaussians backward (A.5).  Skips radii <= 0 (outputs stay zero).
 * v_quat is the vjp w.r.t. the NORMALISED quaternion components (upstream
 * quat_to_rotmat_vjp does not chain through the in-kernel normalisation);
 * the fov clamp is ignored by the vjp, as upstream.
 * ------------------------------------------------------------------------ */
void ORC(project_bwd)(int N, const real* means, const real* scales, real glob_scale,
                     const real* quats, const real* viewmat, real fx, real fy,
                     const real* cov3d, const int32_t* radii, const real* conics,
                     const real* compensation, const real* v_xy, const real* v_depth,
                     const real* v_conic, const real* v_compensation, real* v_cov2d,
                     real* v_cov3d, real* v_mean3d, real* v_scale, real* v_quat) {
  const real* V = viewmat;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i) {
    if (radii[i] <= 0) continue;
    const real* p = means + 3 * i;
    real tx = V[0] * p[0] + V[1] * p[1] + V[2] * p[2] + V[3];
    real ty = V[4] * p[0] + V[5] * p[1] + V[6] * p[2] + V[7];
    real tz = V[8] * p[0] + V[9] * p[1] + V[10] * p[2] + V[11];
    /* project_pix vjp */
    real rw = 1.f / (tz + ORC_Z_EPS);
    real vpx = fx * v_xy[2 * i], vpy = fy * v_xy[2 * i + 1];
    real vv[3] = {vpx * rw, vpy * rw, -(vpx * tx + vpy * ty) * rw * rw};
    real vm[3];
    for (int c = 0; c < 3; ++c) vm[c] = V[0 + c] * vv[0] + V[4 + c] * vv[1] + V[8 + c] * vv[2];
    real vz = v_depth ? v_depth[i] : 0.f;
    vm[0] += V[8] * vz; vm[1] += V[9] * vz; vm[2] += V[10] * vz;

    /* conic -> cov2d vjp: v_Sigma = -X G X */
    real X0 = conics[3 * i], X1 = conics[3 * i + 1], X2 = conics[3 * i + 2];
    real G0 = v_conic[3 * i], G1 = 0.5f * v_conic[3 * i + 1], G2 = v_conic[3 * i + 2];
    /* XG */
    real a00 = X0 * G0 + X1 * G1, a01 = X0 * G1 + X1 * G2;
    real a10 = X1 * G0 + X2 * G1, a11 = X1 * G1 + X2 * G2;
    real s00 = -(a00 * X0 + a01 * X1), s01 = -(a00 * X1 + a01 * X2);
    real s10 = -(a10 * X0 + a11 * X1), s11 = -(a10 * X1 + a11 * X2);
    real vc2[3] = {s00, s01 + s10, s11};
    /* compensation vjp */
    {
      real comp = compensation[i];
      real inv_det = X0 * X2 - X1 * X1;
      real om = 1.f - comp * comp;
      real vsq = (v_compensation ? v_compensation[i] : 0.f) * 0.5f / (comp + ORC_COMP_EPS);
      vc2[0] += vsq * (om * X0 - ORC_BLUR * inv_det);
      vc2[1] += 2.f * vsq * (om * X1);
      vc2[2] += vsq * (om * X2 - ORC_BLUR * inv_det);
    }
    if (v_cov2d) { v_cov2d[3 * i] = vc2[0]; v_cov2d[3 * i + 1] = vc2[1]; v_cov2d[3 * i + 2] = vc2[2]; }

    /* EWA vjp (unclamped t) */
    real rz = 1.f / tz, rz2 = rz * rz, rz3 = rz2 * rz;
    mat3 J = {{fx * rz, 0.f, -fx * tx * rz2, 0.f, fy * rz, -fy * ty * rz2, 0.f, 0.f, 0.f}};
    mat3 Wm = {{V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]}};
    mat3 T = mat3_mul(J, Wm);
    const real* c3 = cov3d + 6 * i;
    mat3 Vc = {{c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]}};
    mat3 Gc = {{vc2[0], 0.5f * vc2[1], 0.f, 0.5f * vc2[1], vc2[2], 0.f, 0.f, 0.f, 0.f}};
    mat3 vV = mat3_mul(mat3_mul(mat3_T(T), Gc), T);
    /* v_T = G T V^T + G^T T V ; G and V symmetric */
    mat3 GTV = mat3_mul(mat3_mul(Gc, T), Vc);
    mat3 vT; for (int k = 0; k < 9; ++k) vT.m[k] = 2.f * GTV.m[k];
    real vc3[6] = {vV.m[0], vV.m[1] + vV.m[3], vV.m[2] + vV.m[6], vV.m[4], vV.m[5] + vV.m[7], vV.m[8]};
    if (v_cov3d) for (int k = 0; k < 6; ++k) v_cov3d[6 * i + k] = vc3[k];
    mat3 vJ = mat3_mul(vT, mat3_T(Wm)); /* row-major vJ[r][c] */
    real vt[3] = {-fx * rz2 * vJ.m[0 * 3 + 2], -fy * rz2 * vJ.m[1 * 3 + 2],
                   -fx * rz2 * vJ.m[0] + 2.f * fx * tx * rz3 * vJ.m[0 * 3 + 2] -
                       fy * rz2 * vJ.m[4] + 2.f * fy * ty * rz3 * vJ.m[1 * 3 + 2]};
    for (int c = 0; c < 3; ++c) vm[c] += vt[0] * V[0 + c] + vt[1] * V[4 + c] + vt[2] * V[8 + c];
    for (int c = 0; c < 3; ++c) v_mean3d[3 * i + c] = vm[c];

    /* cov3d -> scale, quat vjp */
    mat3 vVs = {{vc3[0], 0.5f * vc3[1], 0.5f * vc3[2], 0.5f * vc3[1], vc3[3], 0.5f * vc3[4],
                 0.5f * vc3[2], 0.5f * vc3[4], vc3[5]}};
    mat3 R = quat_to_rotmat(quats + 4 * i);
    real sc[3] = {glob_scale * scales[3 * i], glob_scale * scales[3 * i + 1], glob_scale * scales[3 * i + 2]};
    mat3 M = R;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M.m[r * 3 + c] = R.m[r * 3 + c] * sc[c];
    mat3 vM = mat3_mul(vVs, M); for (int k = 0; k < 9; ++k) vM.m[k] *= 2.f;
    for (int c = 0; c < 3; ++c) {
      real s = 0.f; for (int r = 0; r < 3; ++r) s += R.m[r * 3 + c] * vM.m[r * 3 + c];
      v_scale[3 * i + c] = s * glob_scale;
    }
    mat3 vR = vM; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) vR.m[r * 3 + c] = vM.m[r * 3 + c] * sc[c];
    const real* q = quats + 4 * i;
    real s = 1.f / R_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    real w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
#define VR(r, c) vR.m[(r) * 3 + (c)] /* row-major dL/dR[r][c] */
    v_quat[4 * i + 0] = 2.f * (x * (VR(2, 1) - VR(1, 2)) + y * (VR(0, 2) - VR(2, 0)) + z * (VR(1, 0) - VR(0, 1)));
    v_quat[4 * i + 1] = 2.f * (-2.f * x * (VR(1, 1) + VR(2, 2)) + y * (VR(1, 0) + VR(0, 1)) +
                               z * (VR(2, 0) + VR(0, 2)) + w * (VR(2, 1) - VR(1, 2)));
    v_quat[4 * i + 2] = 2.f * (x * (VR(1, 0) + VR(0, 1)) - 2.f * y * (VR(0, 0) + VR(2, 2)) +
                               z * (VR(2, 1) + VR(1, 2)) + w * (VR(0, 2) - VR(2, 0)));
    v_quat[4 * i + 3] = 2.f * (x * (VR(2, 0) + VR(0, 2)) + y * (VR(2, 1) + VR(1, 2)) -
                               2.f * z * (VR(0, 0) + VR(1, 1)) + w * (VR(1, 0) - VR(0, 1)));
#undef VR
  }
}
