/*
 * oracle/sg_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).
 *
 * Scalar fp32 restatement of the reference's spherical-Gaussian specular lobe
 * evaluation:
 *   forward   /root/reference/extensions/sgutils/sg.cu:27-76
 *   backward  /root/reference/extensions/sgutils/sg.cu:78-175
 * including its quirks: d acos/dc replaced by -20 at
 * |c| >= 1 (sg.cu:129,139), no gradient to prim_pts / light_pts
 * (sgutils.py:30), grad_light_values accumulated only when requested.
 * The reference compiles with -use_fast_math (extensions/sgutils/setup.py:31);
 * this oracle uses libm expf/acosf, so device parity is to ~1e-6 relative.
 *
 * PARITY UNPINNED: the reference has no test or golden vector for sgutils
 * (SURVEY.md section 4); the explicit backward is cross-checked against torch
 * autograd in tests/test_oracle_sg.py.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static const float TWOPI = 6.28318530718f;       /* sg.cu:18 */
static const float INV2PI = 0.15915494309f;      /* sg.cu:19 */
static const float SQRT2PI23 = 3.03352966508f;   /* sg.cu:20 */
static const float INVSQRT2PI23 = 0.32964899322f;/* sg.cu:21 */

static float sq(float v) { return v * v; }
static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* sg.cu:27-76 */
void orc_sg_fwd(int N, int D, int L, const float* lobe_dirs, const float* lobe_sigmas,
                const float* light_values, const float* light_pts, const float* prim_pts,
                const int32_t* n_lights, int w_type, float* integral) {
  for (int n = 0; n < N; ++n) {
    int nL = n_lights[n];
    for (int d = 0; d < D; ++d) {
      size_t e = (size_t)n * D + d;
      const float* dir = lobe_dirs + 3 * e;
      const float* pp = prim_pts + 3 * e;
      float sigma = lobe_sigmas[e];
      float sum[3] = {0.f, 0.f, 0.f};
      for (int l = 0; l < nL; ++l) {
        const float* env = light_values + 3 * ((size_t)n * L + l);
        const float* lp = light_pts + 3 * ((size_t)n * L + l);
        float ld[3] = {lp[0] - pp[0], lp[1] - pp[1], lp[2] - pp[2]};
        float nrm = sqrtf(ld[0] * ld[0] + ld[1] * ld[1] + ld[2] * ld[2]);
        ld[0] /= nrm; ld[1] /= nrm; ld[2] /= nrm;
        float c = clampf(ld[0] * dir[0] + ld[1] * dir[1] + ld[2] * dir[2], -1.f, 1.f);
        float angle = acosf(c), w = 0.f;
        switch (w_type) {
          case 0: w = expf(-0.5f * sq(angle / sigma)) / (sigma * SQRT2PI23); break;
          case 1: w = expf(-0.5f * sq(angle / sigma)); break;
          case 2: w = expf((c - 1.f) / sigma) / (sigma * TWOPI); break;
          case 3: w = expf((c - 1.f) / sigma); break;
        }
        sum[0] += env[0] * w; sum[1] += env[1] * w; sum[2] += env[2] * w;
      }
      integral[3 * e] = sum[0]; integral[3 * e + 1] = sum[1]; integral[3 * e + 2] = sum[2];
    }
  }
}

/* sg.cu:78-175.  grad_light_values may be NULL; otherwise it is accumulated
 * into (caller zeroes it, sgutils.py:43-47). */
void orc_sg_bwd(int N, int D, int L, const float* lobe_dirs, const float* lobe_sigmas,
                const float* light_values, const float* light_pts, const float* prim_pts,
                const int32_t* n_lights, const float* grad_integral, int w_type,
                float* grad_dirs, float* grad_sigmas, float* grad_light_values) {
  for (int n = 0; n < N; ++n) {
    int nL = n_lights[n];
    for (int d = 0; d < D; ++d) {
      size_t e = (size_t)n * D + d;
      const float* gi = grad_integral + 3 * e;
      const float* dir = lobe_dirs + 3 * e;
      const float* pp = prim_pts + 3 * e;
      float sigma = lobe_sigmas[e];
      float gdir[3] = {0.f, 0.f, 0.f}, gsig = 0.f;
      for (int l = 0; l < nL; ++l) {
        const float* env = light_values + 3 * ((size_t)n * L + l);
        const float* lp = light_pts + 3 * ((size_t)n * L + l);
        float ld[3] = {lp[0] - pp[0], lp[1] - pp[1], lp[2] - pp[2]};
        float nrm = sqrtf(ld[0] * ld[0] + ld[1] * ld[1] + ld[2] * ld[2]);
        ld[0] /= nrm; ld[1] /= nrm; ld[2] /= nrm;
        float c = ld[0] * dir[0] + ld[1] * dir[1] + ld[2] * dir[2];
        float cc = clampf(c, -1.f, 1.f);
        float angle = acosf(cc);
        float weight = 0.f, dc = 0.f, da = 0.f, dw = 0.f, ex = 0.f;
        float dacos = (c > -1.f && c < 1.f) ? (-1.f / sqrtf(1.f - sq(c))) : -20.f;
        dw = gi[0] * env[0] + gi[1] * env[1] + gi[2] * env[2];
        switch (w_type) {
          case 0:
            ex = expf(-0.5f * sq(angle / sigma));
            weight = ex / (sigma * SQRT2PI23);
            gsig += dw * ((ex * INVSQRT2PI23 * (sq(angle) - sq(sigma))) / (sq(sigma) * sq(sigma)));
            da = dw * -((INVSQRT2PI23 * angle * ex) / (sq(sigma) * sigma));
            dc = da * dacos;
            break;
          case 1:
            ex = expf(-0.5f * sq(angle / sigma));
            weight = ex;
            gsig += dw * ((ex * sq(angle)) / (sigma * sq(sigma)));
            da = dw * -((angle * ex) / sq(sigma));
            dc = da * dacos;
            break;
          case 2:
            ex = expf((cc - 1.f) / sigma);
            weight = ex / (sigma * TWOPI);
            gsig += dw * ((ex * INV2PI * ((1.f - cc) - sigma)) / (sigma * sq(sigma)));
            dc = dw * INV2PI * ex / sq(sigma);
            break;
          case 3:
            ex = expf((cc - 1.f) / sigma);
            weight = ex;
            gsig += dw * ((ex * (1.f - cc) / sq(sigma)));
            dc = dw * ex / sigma;
            break;
        }
        gdir[0] += dc * ld[0]; gdir[1] += dc * ld[1]; gdir[2] += dc * ld[2];
        if (grad_light_values) {
          float* g = grad_light_values + 3 * ((size_t)n * L + l);
          g[0] += gi[0] * weight; g[1] += gi[1] * weight; g[2] += gi[2] * weight;
        }
      }
      grad_sigmas[e] = gsig;
      grad_dirs[3 * e] = gdir[0]; grad_dirs[3 * e + 1] = gdir[1]; grad_dirs[3 * e + 2] = gdir[2];
    }
  }
}
