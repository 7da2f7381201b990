/*
 * oracle/mvp_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).
 *
 * Scalar fp32 restatement of the reference's Mixture-of-Volumetric-Primitives ray marcher
 * (algo 0, channels-last template, additive accumulation, fixed-order BVH) and of compute_raydirs:
 *   forward/backward march   /root/reference/extensions/mvpraymarch/mvpraymarch_subset_kernel.h:7-228
 *   primitive transform      .../primtransf.h:99-179 (PrimTransfSRT)
 *   sampler + fade           .../primsampler.h:44-92, utils.h:523-770 (GridSamplerChlast, zero padding,
 *                            align_corners=True coordinates)
 *   accumulation             .../primaccum.h:63-98 (PrimAccumAdditive: saturation at alpha 1, raysat)
 *   hit list                 .../utils.h:949-1045 (leaf test = ray vs oriented unit box, DFS leaf order
 *                            of the implicit heap children = 2i+1, 2i+2)
 *   shadow splat             .../primsplatter.h:29-36, utils.h:773-880
 *   leaf / node AABBs        .../primtransf.h:12-63, bvh.cu:157-201
 *   ray directions           /root/reference/extensions/utils/utils_kernel.cu:11-51
 * PINNED: tests/test_oracle_mvp.py checks it against golden vectors produced by the reference's own
 * in-tree PyTorch ray marcher (mvpraymarch.py:581-669) and raydirs (utils.py:127-143), see
 * tests/golden/make_mvp_golden.py.
 * Hit list: by default per ray and uncapped -- equal to the CUDA kernels' result whenever no warp footprint
 * collects more than 512 boxes (extra boxes of the warp-wide union fail the per-sample valid() test).
 * orc_mvp_set_footprint(w, h, cap) switches to the kernels' exact list semantics: ONE list per w x h pixel block
 * (the union of the boxes any ray of the block hits, in DFS leaf order, truncated to the first `cap`;
 * utils.h:976-1012 with sync = true).  (8, 4, 512) is the reference's 32-lane warp of its (8, 16) thread block
 * (mvpraymarch.py:334, mvpraymarch_kernel.cu:39); (8, 8, 512) is the wave64 footprint of csrc/mvp.hip.
 */
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } f3;
static f3 f3_(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static f3 ld3(const float* p) { return f3_(p[0], p[1], p[2]); }
static f3 add(f3 a, f3 b) { return f3_(a.x + b.x, a.y + b.y, a.z + b.z); }
static f3 sub(f3 a, f3 b) { return f3_(a.x - b.x, a.y - b.y, a.z - b.z); }
static f3 mul(f3 a, f3 b) { return f3_(a.x * b.x, a.y * b.y, a.z * b.z); }
static f3 scl(f3 a, float s) { return f3_(a.x * s, a.y * s, a.z * s); }
static float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static float min3(f3 a) { return fminf(fminf(a.x, a.y), a.z); }
static float max3(f3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }

/* DFS (left first) leaf order of the implicit heap with K leaves at nodes K-1 .. 2K-2 */
static void dfs_leaf_order(int K, int* order) {
  int* stack = (int*)malloc(sizeof(int) * 128);
  int sp = 0, n = 0, node = 0;
  stack[sp++] = -1;
  while (node != -1) {
    if (node >= K - 1) { order[n++] = node - (K - 1); node = stack[--sp]; }
    else { stack[sp++] = node * 2 + 2; node = node * 2 + 1; }
  }
  free(stack);
}

typedef struct {
  f3 xmt, pr0, pr1, pr2, rxmt, ps;
} xform_t;

/* primtransf.h:119-132 */
static f3 xform_fwd(xform_t* t, const float* primpos, const float* primrot, const float* primscale, int k, f3 x) {
  f3 pt = ld3(primpos + 3 * k);
  t->pr0 = ld3(primrot + 9 * k); t->pr1 = ld3(primrot + 9 * k + 3); t->pr2 = ld3(primrot + 9 * k + 6);
  t->ps = ld3(primscale + 3 * k);
  t->xmt = sub(x, pt);
  t->rxmt = add(add(scl(t->pr0, t->xmt.x), scl(t->pr1, t->xmt.y)), scl(t->pr2, t->xmt.z));
  return mul(t->rxmt, t->ps);
}

static int valid_pos(f3 p) { return p.x > -1.f && p.x < 1.f && p.y > -1.f && p.y < 1.f && p.z > -1.f && p.z < 1.f; }

typedef struct { int i[8]; float w[8]; float ix, iy, iz; int x0, y0, z0; } tri_t;

/* utils.h:523-617 coordinates and corner weights; i[c] = linear voxel index or -1 when out of bounds */
static void tri_setup(tri_t* q, int D, int H, int W, f3 pos) {
  float ix = fmaxf(-100.f, fminf(100.f, (pos.x + 1.f) / 2)) * (W - 1);
  float iy = fmaxf(-100.f, fminf(100.f, (pos.y + 1.f) / 2)) * (H - 1);
  float iz = fmaxf(-100.f, fminf(100.f, (pos.z + 1.f) / 2)) * (D - 1);
  int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
  q->ix = ix; q->iy = iy; q->iz = iz; q->x0 = x0; q->y0 = y0; q->z0 = z0;
  for (int c = 0; c < 8; ++c) {
    int dx = c & 1, dy = (c >> 1) & 1, dz = (c >> 2) & 1;
    int x = x0 + dx, y = y0 + dy, z = z0 + dz;
    float wx = dx ? (ix - x0) : (x0 + 1 - ix), wy = dy ? (iy - y0) : (y0 + 1 - iy), wz = dz ? (iz - z0) : (z0 + 1 - iz);
    q->w[c] = wx * wy * wz;
    q->i[c] = (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) ? (z * H + y) * W + x : -1;
  }
}

/* 3-channel trilinear sample of a channels-last warp field at pos (same sampler as the template's, utils.h:523-617) */
static f3 warp_sample(const float* wp, int D, int H, int W, f3 pos) {
  tri_t q; tri_setup(&q, D, H, W, pos);
  f3 r = f3_(0.f, 0.f, 0.f);
  for (int c = 0; c < 8; ++c) if (q.i[c] >= 0) {
    const float* v = wp + (size_t)q.i[c] * 3;
    r.x += v[0] * q.w[c]; r.y += v[1] * q.w[c]; r.z += v[2] * q.w[c];
  }
  return r;
}

static int sat_floor_to_int(float v) {
  float f = floorf(v);
  if (f != f) return 0;
  if (f >= 2147483520.f) return INT_MAX;   /* cvt.rzi saturates */
  if (f <= -2147483648.f) return INT_MIN;
  return (int)f;
}

typedef struct {
  int n; int* k;
  float rmin, rmax;
} hits_t;

/* utils.h:976-1012 leaf test, per ray */
static void collect_hits(int K, const int* order, f3 raypos, f3 raydir, const float* primpos,
                         const float* primrot, const float* primscale, hits_t* h) {
  h->n = 0; h->rmin = INFINITY; h->rmax = -INFINITY;
  for (int j = 0; j < K; ++j) {
    int k = order[j];
    f3 pt = ld3(primpos + 3 * k), pr0 = ld3(primrot + 9 * k), pr1 = ld3(primrot + 9 * k + 3), pr2 = ld3(primrot + 9 * k + 6);
    f3 ps = ld3(primscale + 3 * k);
    f3 xmt = sub(raypos, pt);
    f3 r0 = mul(add(add(scl(pr0, xmt.x), scl(pr1, xmt.y)), scl(pr2, xmt.z)), ps);
    f3 rd = mul(add(add(scl(pr0, raydir.x), scl(pr1, raydir.y)), scl(pr2, raydir.z)), ps);
    f3 ird = f3_(1.f / rd.x, 1.f / rd.y, 1.f / rd.z);
    f3 t0 = mul(sub(f3_(-1.f, -1.f, -1.f), r0), ird), t1 = mul(sub(f3_(1.f, 1.f, 1.f), r0), ird);
    f3 tmn = f3_(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z));
    f3 tmx = f3_(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z));
    float trmin = max3(tmn), trmax = min3(tmx);
    if (trmin <= trmax) {
      h->rmin = fminf(h->rmin, trmin); h->rmax = fmaxf(h->rmax, trmax);
      h->k[h->n++] = k;
    }
  }
}

/* footprint mode (see the header): 0 = per ray, uncapped */
static int g_fp_w = 0, g_fp_h = 0, g_fp_cap = 0;
void orc_mvp_set_footprint(int w, int h, int cap) { g_fp_w = w; g_fp_h = h; g_fp_cap = cap; }

/* one list per footprint block: L[blk * (cap + 1)] = count, then the box ids; NULL in per-ray mode */
static int* build_block_lists(int N, int H, int W, int K, const int* order, const float* rayposim,
                              const float* raydirim, const float* primpos_, const float* primrot_,
                              const float* primscale_, int* nbx_out, int* nby_out) {
  if (g_fp_w <= 0 || g_fp_h <= 0) return NULL;
  const int nbx = (W + g_fp_w - 1) / g_fp_w, nby = (H + g_fp_h - 1) / g_fp_h, cap = g_fp_cap;
  int* L = (int*)calloc((size_t)N * nbx * nby * (cap + 1), sizeof(int));
#pragma omp parallel
  {
    hits_t h; h.k = (int*)malloc(sizeof(int) * K);
    unsigned char* any = (unsigned char*)malloc((size_t)K);
#pragma omp for schedule(dynamic, 4)
    for (int blk = 0; blk < N * nbx * nby; ++blk) {
      int n = blk / (nbx * nby), by = (blk / nbx) % nby, bx = blk % nbx;
      const float* primpos = primpos_ + (size_t)n * K * 3; const float* primrot = primrot_ + (size_t)n * K * 9;
      const float* primscale = primscale_ + (size_t)n * K * 3;
      memset(any, 0, (size_t)K);
      for (int y = by * g_fp_h; y < (by + 1) * g_fp_h && y < H; ++y)
        for (int x = bx * g_fp_w; x < (bx + 1) * g_fp_w && x < W; ++x) {
          size_t r = ((size_t)n * H + y) * W + x;
          collect_hits(K, order, ld3(rayposim + 3 * r), ld3(raydirim + 3 * r), primpos, primrot, primscale, &h);
          for (int s = 0; s < h.n; ++s) any[h.k[s]] = 1;
        }
      int* out = L + (size_t)blk * (cap + 1);
      int cnt = 0;
      for (int j = 0; j < K && cnt < cap; ++j) if (any[order[j]]) out[1 + cnt++] = order[j];
      out[0] = cnt;
    }
    free(h.k); free(any);
  }
  *nbx_out = nbx; *nby_out = nby;
  return L;
}

/* ---------------------------------------------------------------------------------------------- *
 * forward: rayrgba[N,H,W,4]; raysat[N,H,W,3] (may be NULL); shadow[N,K,TD,TH,TW,2] (may be NULL,
 * accumulated).  template[N,K,TD,TH,TW,4].
 * ---------------------------------------------------------------------------------------------- */
/* warp_ (optional, algo 1): [N,K,WD,WH,WW,3] channels-last warp fields; the template is sampled at the warped
 * position y1 = trilinear(warp_k, y0) (primsampler.h:53-61), which may leave the box: zero padding per corner. */
void orc_mvp_fwd_warp(int N, int H, int W, int K, const float* rayposim, const float* raydirim, float stepsize,
                      const float* tminmaxim, const float* primpos_, const float* primrot_, const float* primscale_,
                      const float* tplate_, int TD, int TH, int TW, const float* warp_, int WD, int WH, int WW,
                      float fadescale, float fadeexp, float* rayrgba, float* raysat, float* shadow_) {
  int* order = (int*)malloc(sizeof(int) * K);
  dfs_leaf_order(K, order);
  const size_t vox = (size_t)TD * TH * TW;
  int nbx = 0, nby = 0;
  int* lists = build_block_lists(N, H, W, K, order, rayposim, raydirim, primpos_, primrot_, primscale_, &nbx, &nby);
#pragma omp parallel
  {
    hits_t h; h.k = (int*)malloc(sizeof(int) * K);
#pragma omp for schedule(dynamic, 16)
    for (int r = 0; r < N * H * W; ++r) {
      int n = r / (H * W);
      const float* primpos = primpos_ + (size_t)n * K * 3; const float* primrot = primrot_ + (size_t)n * K * 9;
      const float* primscale = primscale_ + (size_t)n * K * 3; const float* tplate = tplate_ + (size_t)n * K * vox * 4;
      float* shadow = shadow_ ? shadow_ + (size_t)n * K * vox * 2 : NULL;
      f3 raypos = ld3(rayposim + 3 * (size_t)r), raydir = ld3(raydirim + 3 * (size_t)r);
      float tmin = tminmaxim[2 * (size_t)r], tmax = tminmaxim[2 * (size_t)r + 1];
      collect_hits(K, order, raypos, raydir, primpos, primrot, primscale, &h);  /* the ray's own hits: rmin, rmax */
      int hn = h.n; const int* hk = h.k;
      if (lists) {
        int y = (r / W) % H, x = r % W;
        const int* Lb = lists + ((size_t)(n * nby + y / g_fp_h) * nbx + x / g_fp_w) * (g_fp_cap + 1);
        hn = Lb[0]; hk = Lb + 1;
      }
      float rt0 = fmaxf(h.rmin, tmin), rt1 = fminf(h.rmax, tmax);
      float t = tmin;
      raypos = add(raypos, scl(raydir, tmin));
      int incs = sat_floor_to_int((rt0 - t) / stepsize);
      t += incs * stepsize;
      raypos = add(raypos, scl(scl(raydir, (float)incs), stepsize));
      float acc[4] = {0, 0, 0, 0}, sat3[3] = {-1.f, -1.f, -1.f};
      int sat = 0;
      while (!(t > rt1 + 1e-5f || sat)) {
        for (int s = 0; s < hn; ++s) {
          int k = hk[s];
          xform_t xf;
          f3 y0 = xform_fwd(&xf, primpos, primrot, primscale, k, raypos);
          if (valid_pos(y0) && !sat && t < rt1 + 1e-5f) {
            float fade = expf(-fadescale * (powf(fabsf(y0.x), fadeexp) + powf(fabsf(y0.y), fadeexp) + powf(fabsf(y0.z), fadeexp)));
            f3 y1 = y0;
            if (warp_) y1 = warp_sample(warp_ + ((size_t)n * K + k) * (size_t)WD * WH * WW * 3, WD, WH, WW, y0);
            tri_t q; tri_setup(&q, TD, TH, TW, y1);
            float smp[4] = {0, 0, 0, 0};
            const float* tp = tplate + (size_t)k * vox * 4;
            for (int c = 0; c < 8; ++c) if (q.i[c] >= 0) for (int ch = 0; ch < 4; ++ch) smp[ch] += tp[(size_t)q.i[c] * 4 + ch] * q.w[c];
            smp[3] *= fade;
            if (shadow) {  /* primsplatter.h:29-36: vis = 1 - accumulated alpha before this sample */
              float vis = 1.f - acc[3];
              float* sp = shadow + (size_t)k * vox * 2;
              for (int c = 0; c < 8; ++c) if (q.i[c] >= 0) {
#pragma omp atomic
                sp[(size_t)q.i[c] * 2] += q.w[c] * vis;
#pragma omp atomic
                sp[(size_t)q.i[c] * 2 + 1] += q.w[c];
              }
            }
            /* primaccum.h:63-79 */
            float newalpha = acc[3] + smp[3] * stepsize;
            float contrib = fminf(newalpha, 1.f) - acc[3];
            acc[0] += smp[0] * contrib; acc[1] += smp[1] * contrib; acc[2] += smp[2] * contrib; acc[3] += contrib;
            if (newalpha >= 1.f) { if (!sat) { sat3[0] = smp[0]; sat3[1] = smp[1]; sat3[2] = smp[2]; } sat = 1; }
          }
        }
        t += stepsize;
        raypos = add(raypos, scl(raydir, stepsize));
      }
      for (int c = 0; c < 4; ++c) rayrgba[4 * (size_t)r + c] = acc[c];
      if (raysat) for (int c = 0; c < 3; ++c) raysat[3 * (size_t)r + c] = sat3[c];
    }
    free(h.k);
  }
  free(order); free(lists);
}

void orc_mvp_fwd(int N, int H, int W, int K, const float* rayposim, const float* raydirim, float stepsize,
                 const float* tminmaxim, const float* primpos_, const float* primrot_, const float* primscale_,
                 const float* tplate_, int TD, int TH, int TW, float fadescale, float fadeexp, float* rayrgba,
                 float* raysat, float* shadow_) {
  orc_mvp_fwd_warp(N, H, W, K, rayposim, raydirim, stepsize, tminmaxim, primpos_, primrot_, primscale_, tplate_, TD, TH, TW,
                   NULL, 0, 0, 0, fadescale, fadeexp, rayrgba, raysat, shadow_);
}

/* ---------------------------------------------------------------------------------------------- *
 * backward (forward-direction replay, mvpraymarch_subset_kernel.h:114-228).  Gradients are summed in
 * double for an order-independent reference; grad_* are written (not accumulated).
 * ---------------------------------------------------------------------------------------------- */
/* With a warp field the chain is the one of the reference's own PyTorch fixture (mvpraymarch.py:603-626): fade and its
 * derivative at the box position y0, template gradient at y1, warp-field gradient and d y1 / d y0 at y0.  (The CUDA
 * sampler hands the ALREADY WARPED position to its backward -- primsampler.h:64 overwrites y0, :71-86 use it -- which
 * disagrees with that fixture whenever the warp is not the identity; no model of the reference emits a warp.) */
void orc_mvp_bwd_warp(int N, int H, int W, int K, const float* rayposim, const float* raydirim, float stepsize,
                      const float* tminmaxim, const float* primpos_, const float* primrot_, const float* primscale_,
                      const float* tplate_, int TD, int TH, int TW, const float* warp_, int WD, int WH, int WW,
                      float fadescale, float fadeexp, const float* raysat_, const float* grad_rayrgba,
                      float* grad_primpos, float* grad_primrot, float* grad_primscale, float* grad_tplate,
                      float* grad_warp) {
  int* order = (int*)malloc(sizeof(int) * K);
  dfs_leaf_order(K, order);
  const size_t vox = (size_t)TD * TH * TW;
  double* gT = (double*)calloc((size_t)N * K * vox * 4, sizeof(double));
  double* gP = (double*)calloc((size_t)N * K * 3, sizeof(double));
  double* gR = (double*)calloc((size_t)N * K * 9, sizeof(double));
  double* gS = (double*)calloc((size_t)N * K * 3, sizeof(double));
  const size_t wvox = (size_t)WD * WH * WW;
  double* gW = warp_ ? (double*)calloc((size_t)N * K * wvox * 3, sizeof(double)) : NULL;
  int nbx = 0, nby = 0;
  int* lists = build_block_lists(N, H, W, K, order, rayposim, raydirim, primpos_, primrot_, primscale_, &nbx, &nby);
#pragma omp parallel
  {
    hits_t h; h.k = (int*)malloc(sizeof(int) * K);
#pragma omp for schedule(dynamic, 16)
    for (int r = 0; r < N * H * W; ++r) {
      int n = r / (H * W);
      const float* primpos = primpos_ + (size_t)n * K * 3; const float* primrot = primrot_ + (size_t)n * K * 9;
      const float* primscale = primscale_ + (size_t)n * K * 3; const float* tplate = tplate_ + (size_t)n * K * vox * 4;
      f3 raypos = ld3(rayposim + 3 * (size_t)r), raydir = ld3(raydirim + 3 * (size_t)r);
      float tmin = tminmaxim[2 * (size_t)r], tmax = tminmaxim[2 * (size_t)r + 1];
      const float* dL = grad_rayrgba + 4 * (size_t)r;
      const float* rs = raysat_ + 3 * (size_t)r;
      collect_hits(K, order, raypos, raydir, primpos, primrot, primscale, &h);
      int hn = h.n; const int* hk = h.k;
      if (lists) {
        int y = (r / W) % H, x = r % W;
        const int* Lb = lists + ((size_t)(n * nby + y / g_fp_h) * nbx + x / g_fp_w) * (g_fp_cap + 1);
        hn = Lb[0]; hk = Lb + 1;
      }
      float rt0 = fmaxf(h.rmin, tmin), rt1 = fminf(h.rmax, tmax);
      float t = tmin;
      raypos = add(raypos, scl(raydir, tmin));
      int incs = sat_floor_to_int((rt0 - t) / stepsize);
      t += incs * stepsize;
      raypos = add(raypos, scl(scl(raydir, (float)incs), stepsize));
      float accw = 0.f; int sat = 0;
      while (t < rt1 + 1e-5f && !sat) {
        for (int s = 0; s < hn; ++s) {
          int k = hk[s];
          xform_t xf;
          f3 y0 = xform_fwd(&xf, primpos, primrot, primscale, k, raypos);
          if (!(valid_pos(y0) && !sat && t < rt1 + 1e-5f)) continue;
          float px = powf(fabsf(y0.x), fadeexp), py = powf(fabsf(y0.y), fadeexp), pz = powf(fabsf(y0.z), fadeexp);
          float fade = expf(-fadescale * (px + py + pz));
          f3 y1 = y0;
          const float* wp = warp_ ? warp_ + ((size_t)n * K + k) * wvox * 3 : NULL;
          if (wp) y1 = warp_sample(wp, WD, WH, WW, y0);
          tri_t q; tri_setup(&q, TD, TH, TW, y1);
          float smp[4] = {0, 0, 0, 0};
          const float* tp = tplate + (size_t)k * vox * 4;
          for (int c = 0; c < 8; ++c) if (q.i[c] >= 0) for (int ch = 0; ch < 4; ++ch) smp[ch] += tp[(size_t)q.i[c] * 4 + ch] * q.w[c];
          smp[3] *= fade;
          /* primaccum.h:81-98 */
          float a = smp[3] * stepsize;
          int thissat = accw + a >= 1.f;
          sat = sat || thissat;
          float weight = sat ? (1.f - accw) : a;
          float dsm[4];
          dsm[0] = weight * dL[0]; dsm[1] = weight * dL[1]; dsm[2] = weight * dL[2];
          if (sat) dsm[3] = 0.f;
          else {
            float s0 = rs[0] > -1.f ? rs[0] : 0.f, s1 = rs[0] > -1.f ? rs[1] : 0.f, s2 = rs[0] > -1.f ? rs[2] : 0.f;
            float s3 = rs[0] > -1.f ? 1.f : 0.f;
            dsm[3] = stepsize * ((smp[0] - s0) * dL[0] + (smp[1] - s1) * dL[1] + (smp[2] - s2) * dL[2] + (1.f - s3) * dL[3]);
          }
          accw += weight;
          /* primsampler.h:70-92 */
          float sgx = y0.x > 0.f ? 1.f : -1.f, sgy = y0.y > 0.f ? 1.f : -1.f, sgz = y0.z > 0.f ? 1.f : -1.f;
          f3 dfade = f3_(-(fadescale * fadeexp) * powf(fabsf(y0.x), fadeexp - 1.f) * sgx,
                         -(fadescale * fadeexp) * powf(fabsf(y0.y), fadeexp - 1.f) * sgy,
                         -(fadescale * fadeexp) * powf(fabsf(y0.z), fadeexp - 1.f) * sgz);
          f3 dLy = scl(dfade, smp[3] * dsm[3]);
          dsm[3] *= fade;
          /* utils.h:619-770 trilinear backward: template grads + position grads */
          double* gt = gT + ((size_t)n * K + k) * vox * 4;
          float gix = 0.f, giy = 0.f, giz = 0.f;
          for (int c = 0; c < 8; ++c) if (q.i[c] >= 0) {
            float dp = 0.f;
            for (int ch = 0; ch < 4; ++ch) {
#pragma omp atomic
              gt[(size_t)q.i[c] * 4 + ch] += (double)(q.w[c] * dsm[ch]);
              dp += tp[(size_t)q.i[c] * 4 + ch] * dsm[ch];
            }
            int dx = c & 1, dy = (c >> 1) & 1, dz = (c >> 2) & 1;
            float wx = dx ? (q.ix - q.x0) : (q.x0 + 1 - q.ix), wy = dy ? (q.iy - q.y0) : (q.y0 + 1 - q.iy),
                  wz = dz ? (q.iz - q.z0) : (q.z0 + 1 - q.iz);
            gix += (dx ? 1.f : -1.f) * wy * wz * dp;
            giy += (dy ? 1.f : -1.f) * wx * wz * dp;
            giz += (dz ? 1.f : -1.f) * wx * wy * dp;
          }
          f3 dLy1 = f3_(gix * (TW - 1.f) / 2, giy * (TH - 1.f) / 2, giz * (TD - 1.f) / 2);
          if (wp) {  /* warp sampler backward at y0: field gradient + d y1 / d y0 */
            tri_t qw; tri_setup(&qw, WD, WH, WW, y0);
            double* gw = gW + ((size_t)n * K + k) * wvox * 3;
            float wix = 0.f, wiy = 0.f, wiz = 0.f;
            for (int c = 0; c < 8; ++c) if (qw.i[c] >= 0) {
              const float* v = wp + (size_t)qw.i[c] * 3;
              float comp[3] = {dLy1.x, dLy1.y, dLy1.z};
              for (int ch = 0; ch < 3; ++ch) {
#pragma omp atomic
                gw[(size_t)qw.i[c] * 3 + ch] += (double)(qw.w[c] * comp[ch]);
              }
              float dp = v[0] * dLy1.x + v[1] * dLy1.y + v[2] * dLy1.z;
              int dx = c & 1, dy = (c >> 1) & 1, dz = (c >> 2) & 1;
              float wx = dx ? (qw.ix - qw.x0) : (qw.x0 + 1 - qw.ix), wy = dy ? (qw.iy - qw.y0) : (qw.y0 + 1 - qw.iy),
                    wz = dz ? (qw.iz - qw.z0) : (qw.z0 + 1 - qw.iz);
              wix += (dx ? 1.f : -1.f) * wy * wz * dp;
              wiy += (dy ? 1.f : -1.f) * wx * wz * dp;
              wiz += (dz ? 1.f : -1.f) * wx * wy * dp;
            }
            dLy1 = f3_(wix * (WW - 1.f) / 2, wiy * (WH - 1.f) / 2, wiz * (WD - 1.f) / 2);
          }
          dLy = add(dLy, dLy1);
          /* primtransf.h:155-179 */
          double* gs = gS + ((size_t)n * K + k) * 3; double* gr = gR + ((size_t)n * K + k) * 9;
          double* gp = gP + ((size_t)n * K + k) * 3;
          float v;
#define AT(p, val) v = (val); _Pragma("omp atomic") p += (double)v
          AT(gs[0], xf.rxmt.x * dLy.x); AT(gs[1], xf.rxmt.y * dLy.y); AT(gs[2], xf.rxmt.z * dLy.z);
          f3 d = mul(dLy, xf.ps);
          AT(gr[0], xf.xmt.x * d.x); AT(gr[1], xf.xmt.x * d.y); AT(gr[2], xf.xmt.x * d.z);
          AT(gr[3], xf.xmt.y * d.x); AT(gr[4], xf.xmt.y * d.y); AT(gr[5], xf.xmt.y * d.z);
          AT(gr[6], xf.xmt.z * d.x); AT(gr[7], xf.xmt.z * d.y); AT(gr[8], xf.xmt.z * d.z);
          AT(gp[0], -dot(xf.pr0, d)); AT(gp[1], -dot(xf.pr1, d)); AT(gp[2], -dot(xf.pr2, d));
#undef AT
        }
        t += stepsize;
        raypos = add(raypos, scl(raydir, stepsize));
      }
    }
    free(h.k);
  }
  for (size_t i = 0; i < (size_t)N * K * vox * 4; ++i) grad_tplate[i] = (float)gT[i];
  for (size_t i = 0; i < (size_t)N * K * 3; ++i) { grad_primpos[i] = (float)gP[i]; grad_primscale[i] = (float)gS[i]; }
  for (size_t i = 0; i < (size_t)N * K * 9; ++i) grad_primrot[i] = (float)gR[i];
  if (gW) { for (size_t i = 0; i < (size_t)N * K * wvox * 3; ++i) grad_warp[i] = (float)gW[i]; free(gW); }
  free(gT); free(gP); free(gR); free(gS); free(order); free(lists);
}

void orc_mvp_bwd(int N, int H, int W, int K, const float* rayposim, const float* raydirim, float stepsize,
                 const float* tminmaxim, const float* primpos_, const float* primrot_, const float* primscale_,
                 const float* tplate_, int TD, int TH, int TW, float fadescale, float fadeexp,
                 const float* raysat_, const float* grad_rayrgba, float* grad_primpos, float* grad_primrot,
                 float* grad_primscale, float* grad_tplate) {
  orc_mvp_bwd_warp(N, H, W, K, rayposim, raydirim, stepsize, tminmaxim, primpos_, primrot_, primscale_, tplate_, TD, TH, TW,
                   NULL, 0, 0, 0, fadescale, fadeexp, raysat_, grad_rayrgba, grad_primpos, grad_primrot, grad_primscale,
                   grad_tplate, NULL);
}

/* primtransf.h:12-63 + bvh.cu:157-201, fixed-order tree (sortedobjid[k] = k, heap children 2i+1, 2i+2):
 * nodeaabb[N, 2K-1, 2, 3] */
void orc_mvp_aabb(int N, int K, const float* primpos, const float* primrot, const float* primscale, float* nodeaabb) {
  for (int n = 0; n < N; ++n) {
    float* A = nodeaabb + (size_t)n * (2 * K - 1) * 6;
    for (int k = 0; k < K; ++k) {
      size_t e = (size_t)n * K + k;
      f3 pt = ld3(primpos + 3 * e), pr0 = ld3(primrot + 9 * e), pr1 = ld3(primrot + 9 * e + 3), pr2 = ld3(primrot + 9 * e + 6);
      f3 ps = ld3(primscale + 3 * e);
      f3 mn = f3_(INFINITY, INFINITY, INFINITY), mx = f3_(-INFINITY, -INFINITY, -INFINITY);
      for (int c = 0; c < 8; ++c) {
        f3 p = f3_(((c & 1) ? 1.f : -1.f) / ps.x, ((c & 2) ? 1.f : -1.f) / ps.y, ((c & 4) ? 1.f : -1.f) / ps.z);
        p = add(f3_(dot(p, pr0), dot(p, pr1), dot(p, pr2)), pt);
        mn = f3_(fminf(mn.x, p.x), fminf(mn.y, p.y), fminf(mn.z, p.z));
        mx = f3_(fmaxf(mx.x, p.x), fmaxf(mx.y, p.y), fmaxf(mx.z, p.z));
      }
      float* a = A + (size_t)(K - 1 + k) * 6;
      a[0] = mn.x; a[1] = mn.y; a[2] = mn.z; a[3] = mx.x; a[4] = mx.y; a[5] = mx.z;
    }
    for (int node = K - 2; node >= 0; --node) {
      const float* l = A + (size_t)(2 * node + 1) * 6; const float* r = A + (size_t)(2 * node + 2) * 6;
      float* a = A + (size_t)node * 6;
      for (int c = 0; c < 3; ++c) { a[c] = fminf(l[c], r[c]); a[3 + c] = fmaxf(l[3 + c], r[3 + c]); }
    }
  }
}

/* utils_kernel.cu:11-51; pixelcoords may be NULL (implicit (w,h) grid) */
void orc_raydirs(int N, int H, int W, const float* viewpos, const float* viewrot, const float* focal,
                 const float* princpt, const float* pixelcoords, float volradius, float* rayposim,
                 float* raydirim, float* tminmaxim) {
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        size_t r = ((size_t)n * H + h) * W + w;
        f3 rp = f3_(viewpos[3 * n] / volradius, viewpos[3 * n + 1] / volradius, viewpos[3 * n + 2] / volradius);
        f3 v0 = ld3(viewrot + 9 * n), v1 = ld3(viewrot + 9 * n + 3), v2 = ld3(viewrot + 9 * n + 6);
        float pxc = pixelcoords ? pixelcoords[2 * r] : (float)w, pyc = pixelcoords ? pixelcoords[2 * r + 1] : (float)h;
        float u = (pxc - princpt[2 * n]) / focal[2 * n], v = (pyc - princpt[2 * n + 1]) / focal[2 * n + 1];
        f3 d = add(add(scl(v0, u), scl(v1, v)), v2);
        d = scl(d, 1.f / sqrtf(dot(d, d)));
        f3 t1 = f3_((-1.f - rp.x) / d.x, (-1.f - rp.y) / d.y, (-1.f - rp.z) / d.z);
        f3 t2 = f3_((1.f - rp.x) / d.x, (1.f - rp.y) / d.y, (1.f - rp.z) / d.z);
        float tmin = fmaxf(fminf(t1.x, t2.x), fmaxf(fminf(t1.y, t2.y), fminf(t1.z, t2.z)));
        float tmax = fminf(fmaxf(t1.x, t2.x), fminf(fmaxf(t1.y, t2.y), fmaxf(t1.z, t2.z)));
        rayposim[3 * r] = rp.x; rayposim[3 * r + 1] = rp.y; rayposim[3 * r + 2] = rp.z;
        raydirim[3 * r] = d.x; raydirim[3 * r + 1] = d.y; raydirim[3 * r + 2] = d.z;
        tminmaxim[2 * r] = fmaxf(tmin, 0.f); tminmaxim[2 * r + 1] = tmax;
      }
}
