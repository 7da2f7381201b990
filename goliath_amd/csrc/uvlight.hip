// uvlight.hip -- URHand per-texel-per-light UV feature loops, forward + backward, gfx950.
//
// Replaces the broadcast PyTorch expressions of ConvTeacherDecoder.forward
//   /root/reference/ca_code/models/urhand.py:419-445   Lambert + Phong^{1,16,32} features
//   /root/reference/ca_code/models/urhand.py:508-567   GGX/Schlick features + physically based texture
// which materialise [B,L,3,S,S] tensors (12 MB x L each at S=1024) a dozen times per call.  Here one
// lane owns one texel, the light list is wave-uniform (scalar loads), every per-light quantity lives
// in registers, and the only HBM traffic is the planar inputs once (p_uv, nml, roughness, tex_mean,
// the L shadow planes) and the planar outputs once: HBM-bound for small L, VALU-bound at L = 32.
// Backward recomputes the per-light terms (no saved intermediates).
#include "gol_common.h"

namespace {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kEps = 1e-12f;  // F.normalize eps
constexpr float kLn2 = 0.69314718056f;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 ld3(const float* __restrict__ p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 ldplanar(const float* __restrict__ p, size_t hw) { return V3{p[0], p[hw], p[2 * hw]}; }
__device__ __forceinline__ void stplanar(float* __restrict__ p, size_t hw, V3 v) { p[0] = v.x; p[hw] = v.y; p[2 * hw] = v.z; }

struct Unit { V3 u; float inv_len; };  // u = x / max(|x|, eps)
__device__ __forceinline__ Unit normalize(V3 x) {
  const float il = __builtin_amdgcn_rsqf(fmaxf(dot(x, x), kEps * kEps));  // == 1 / max(|x|, eps): one v_rsq_f32
  return Unit{x * il, il};
}
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// gradient of normalize: (g - u (u.g)) / |x|
__device__ __forceinline__ V3 normalize_bwd(const Unit& n, V3 g) { return (g - n.u * dot(n.u, g)) * n.inv_len; }

__device__ __forceinline__ bool in01(float x, float lo, float hi) { return x >= lo && x <= hi; }

// q = min(s^p, 1) and dq/ds for s >= 0, with l2s = log2(s) shared by all the powers of a light:
// s^p = exp2(p log2 s) on the transcendental unit (v_log_f32 / v_exp_f32) instead of the generic powf;
// log2(0) = -inf gives 0^p = 0 for p > 0 like torch.pow.
__device__ __forceinline__ float pow_cap(float s, float l2s, float p, float& dq) {
  const float sp = (p == 1.f) ? s : __builtin_amdgcn_exp2f(p * l2s);
  const float spm1 = (p == 1.f) ? 1.f : __builtin_amdgcn_exp2f((p - 1.f) * l2s);
  dq = (sp <= 1.f) ? p * spm1 : 0.f;
  return fminf(sp, 1.f);
}

// ---------------------------------------------------------------------------------------------- Phong
template <bool BWD>
__global__ __launch_bounds__(256) void phong_kernel(const gol_uvlight_in in, float* __restrict__ diff_out,
                                                    float* __restrict__ spec_out, const float* __restrict__ u_diff,
                                                    const float* __restrict__ u_spec, float* __restrict__ g_p,
                                                    float* __restrict__ g_n) {
  const int b = blockIdx.y;
  const size_t hw = (size_t)in.HW;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= hw) return;
  const int L = in.L, P = in.n_pow;
  const V3 p = ldplanar(in.p_uv + (size_t)b * 3 * hw + t, hw);
  const V3 n = ldplanar(in.nml + (size_t)b * 3 * hw + t, hw);
  const Unit V = normalize(ld3(in.cam_pos + 3 * b) - p);
  const V3 view = V.u * -1.f;
  const float a = dot(view, n);
  const V3 ref = view - n * (2.f * a);
  float lint = 0.f;
  for (int l = 0; l < L; ++l) lint += in.light_intensity[(size_t)b * L + l];
  const float inv = 1.f / (lint + 1e-6f);

  float D = 0.f, S[GOL_UV_MAX_POW] = {0.f, 0.f, 0.f, 0.f};
  float uD = 0.f, uS[GOL_UV_MAX_POW] = {0.f, 0.f, 0.f, 0.f};
  V3 gp = v3(0, 0, 0), gn = v3(0, 0, 0), gref = v3(0, 0, 0);
  if (BWD) {
    uD = u_diff[(size_t)b * hw + t];
    for (int k = 0; k < P; ++k) uS[k] = u_spec[((size_t)b * P + k) * hw + t];
  }
  for (int l = 0; l < L; ++l) {
    const Unit Lv = normalize(ld3(in.light_pos + ((size_t)b * L + l) * 3) - p);
    const float sh = in.shadow_map ? in.shadow_map[((size_t)b * L + l) * hw + t] : 1.f;
    const float w = in.light_intensity[(size_t)b * L + l] * sh;
    const float dx = dot(n, Lv.u), sx = dot(ref, Lv.u);
    const float s = fmaxf(sx, 0.f);
    const float l2s = __builtin_amdgcn_logf(s);  // v_log_f32 = log2
    if (!BWD) {
      D += fminf(fmaxf(dx, 0.f), 1.f) * w;
      for (int k = 0; k < P; ++k) { float dq; S[k] += pow_cap(s, l2s, in.pow[k], dq) * w; }
    } else {
      const float gd = in01(dx, 0.f, 1.f) ? uD * inv * w : 0.f;
      float gs = 0.f;
      for (int k = 0; k < P; ++k) { float dq; pow_cap(s, l2s, in.pow[k], dq); gs += uS[k] * dq; }
      gs = (sx >= 0.f) ? gs * inv * w : 0.f;
      gn = gn + Lv.u * gd;
      gref = gref + Lv.u * gs;
      const V3 gL = n * gd + ref * gs;
      gp = gp - normalize_bwd(Lv, gL);
    }
  }
  if (!BWD) {
    diff_out[(size_t)b * hw + t] = inv * D;
    for (int k = 0; k < P; ++k) spec_out[((size_t)b * P + k) * hw + t] = inv * S[k];
  } else {
    // ref = view - 2 (view.n) n ; view = -V
    const float ng = dot(n, gref);
    const V3 gview = gref - n * (2.f * ng);
    gn = gn - (view * ng + gref * a) * 2.f;
    gp = gp - normalize_bwd(V, gview * -1.f);
    stplanar(g_p + (size_t)b * 3 * hw + t, hw, gp);
    stplanar(g_n + (size_t)b * 3 * hw + t, hw, gn);
  }
}

// ---------------------------------------------------------------------------------------------- GGX
template <bool BWD>
__global__ __launch_bounds__(256) void ggx_kernel(const gol_uvlight_in in, float* __restrict__ feat_out,
                                                  float* __restrict__ rgb_out, const float* __restrict__ u_feat,
                                                  const float* __restrict__ u_rgb, float* __restrict__ g_p,
                                                  float* __restrict__ g_n, float* __restrict__ g_r,
                                                  float* __restrict__ g_tex) {
  const int b = blockIdx.y;
  const size_t hw = (size_t)in.HW;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= hw) return;
  const int L = in.L, P = in.n_pow;
  const V3 p = ldplanar(in.p_uv + (size_t)b * 3 * hw + t, hw);
  const V3 N0 = ldplanar(in.nml + (size_t)b * 3 * hw + t, hw);
  const V3 tex = ldplanar(in.tex_mean + (size_t)b * 3 * hw + t, hw);
  const float r = in.roughness[(size_t)b * hw + t];
  const Unit V = normalize(ld3(in.cam_pos + 3 * b) - p);
  const float nov0 = dot(V.u, N0);
  const float sg = nov0 > 0.f ? 1.f : (nov0 < 0.f ? -1.f : 0.f);
  const V3 N = N0 * sg;
  const float nov = dot(N, V.u);
  const float al2 = r * r * r * r;
  const float kk = (r * r + 2.f * r + 1.f) * 0.125f;
  const float F0 = in.fresnel;
  const float nom1 = nov * (1.f - kk) + kk;
  float lint = 0.f;
  for (int l = 0; l < L; ++l) lint += in.light_intensity[(size_t)b * L + l];
  const float inv = 1.f / (lint + 1e-6f);
  const float four_pi = 4.f * kPi, invL = 1.f / (float)L;
  const V3 albedo = tex * (1.f / (255.f * kPi));

  float feat[1 + GOL_UV_MAX_POW] = {0.f, 0.f, 0.f, 0.f, 0.f};
  V3 rgb = v3(0, 0, 0);
  float uF[1 + GOL_UV_MAX_POW] = {0.f, 0.f, 0.f, 0.f, 0.f};
  V3 uR = v3(0, 0, 0);
  V3 gp = v3(0, 0, 0), gN0 = v3(0, 0, 0), gN = v3(0, 0, 0), gV = v3(0, 0, 0), gtex = v3(0, 0, 0);
  float g_al2 = 0.f, g_k = 0.f, g_nov = 0.f;
  if (BWD) {
    for (int k = 0; k <= P; ++k) uF[k] = u_feat[((size_t)b * (P + 1) + k) * hw + t];
    uR = ldplanar(u_rgb + (size_t)b * 3 * hw + t, hw);
  }
  for (int l = 0; l < L; ++l) {
    const Unit Lv = normalize(ld3(in.light_pos + ((size_t)b * L + l) * 3) - p);
    const Unit H = normalize((Lv.u + V.u) * 0.5f);
    const float I = in.light_intensity[(size_t)b * L + l];
    const float sh = in.shadow_map ? in.shadow_map[((size_t)b * L + l) * hw + t] : 1.f;
    const float nol_x = dot(N, Lv.u), noh_x = dot(N, H.u), voh_x = dot(V.u, H.u);
    const float nol = fminf(fmaxf(nol_x, 1e-6f), 1.f), noh = fminf(fmaxf(noh_x, 1e-6f), 1.f),
                voh = fminf(fmaxf(voh_x, 1e-6f), 1.f);
    const float fmi = (-5.55473f * voh - 6.98316f) * voh;
    const float e2 = __builtin_amdgcn_exp2f(fmi);
    const float f0 = F0 + (1.f - F0) * e2;
    const float frac = f0 * al2;
    const float nom0 = noh * noh * (al2 - 1.f) + 1.f;
    const float nom2 = nol * (1.f - kk) + kk;
    const float nomx = four_pi * nom0 * nom0 * nom1 * nom2;
    const float nom = fminf(fmaxf(nomx, 1e-6f), four_pi);
    const float inom = rcp(nom);
    const float spec = frac * inom;
    const float l2spec = __builtin_amdgcn_logf(spec);
    const float dcx = dot(N0, Lv.u);
    const float dcos = fminf(fmaxf(dcx, 0.f), 1.f), cosine = fmaxf(dcx, 0.f);
    const float lit = dcos > 0.f ? 1.f : 0.f;
    const float wsh = inv * I * sh;
    if (!BWD) {
      feat[0] += dcos * wsh;
      for (int k = 0; k < P; ++k) { float dq; feat[1 + k] += 10.f * pow_cap(spec, l2spec, in.pow[k], dq) * wsh * lit; }
      const float c = four_pi * invL * I * cosine;
      rgb = rgb + (albedo + v3(spec, spec, spec)) * c;
    } else {
      const float c = four_pi * invL * I;
      const float uRs = uR.x + uR.y + uR.z;
      float g_spec = uRs * c * cosine;
      for (int k = 0; k < P; ++k) { float dq; pow_cap(spec, l2spec, in.pow[k], dq); g_spec += uF[1 + k] * 10.f * wsh * lit * dq; }
      const float g_cos = (dcx >= 0.f) ? c * (dot(uR, albedo) + uRs * spec) : 0.f;
      gtex = gtex + uR * (c * cosine * (1.f / (255.f * kPi)));
      const float g_dcx = (in01(dcx, 0.f, 1.f) ? uF[0] * wsh : 0.f) + g_cos;
      gN0 = gN0 + Lv.u * g_dcx;
      V3 gL = N0 * g_dcx;
      // spec = frac / nom
      const float g_frac = g_spec * inom;
      const float g_nom = in01(nomx, 1e-6f, four_pi) ? -g_spec * frac * inom * inom : 0.f;
      g_al2 += g_frac * f0;
      const float g_voh = in01(voh_x, 1e-6f, 1.f)
                              ? g_frac * al2 * (1.f - F0) * kLn2 * e2 * (2.f * -5.55473f * voh - 6.98316f)
                              : 0.f;
      const float g_nom0 = g_nom * four_pi * 2.f * nom0 * nom1 * nom2;
      const float g_nom1 = g_nom * four_pi * nom0 * nom0 * nom2;
      const float g_nom2 = g_nom * four_pi * nom0 * nom0 * nom1;
      const float g_noh = in01(noh_x, 1e-6f, 1.f) ? g_nom0 * 2.f * noh * (al2 - 1.f) : 0.f;
      g_al2 += g_nom0 * noh * noh;
      g_nov += g_nom1 * (1.f - kk);
      g_k += g_nom1 * (1.f - nov) + g_nom2 * (1.f - nol);
      const float g_nol = in01(nol_x, 1e-6f, 1.f) ? g_nom2 * (1.f - kk) : 0.f;
      gN = gN + Lv.u * g_nol + H.u * g_noh;
      gL = gL + N * g_nol;
      const V3 gH = N * g_noh + V.u * g_voh;
      gV = gV + H.u * g_voh;
      const V3 gHraw = normalize_bwd(H, gH) * 0.5f;
      gL = gL + gHraw;
      gV = gV + gHraw;
      gp = gp - normalize_bwd(Lv, gL);
    }
  }
  if (!BWD) {
    for (int k = 0; k <= P; ++k) feat_out[((size_t)b * (P + 1) + k) * hw + t] = feat[k];
    stplanar(rgb_out + (size_t)b * 3 * hw + t, hw, rgb);
  } else {
    gN = gN + V.u * g_nov;
    gV = gV + N * g_nov;
    gN0 = gN0 + gN * sg;
    gp = gp - normalize_bwd(V, gV);
    stplanar(g_p + (size_t)b * 3 * hw + t, hw, gp);
    stplanar(g_n + (size_t)b * 3 * hw + t, hw, gN0);
    g_r[(size_t)b * hw + t] = g_al2 * 4.f * r * r * r + g_k * (r + 1.f) * 0.25f;
    stplanar(g_tex + (size_t)b * 3 * hw + t, hw, gtex);
  }
}

int check(const gol_uvlight_in* in, bool ggx) {
  GOL_REQUIRE(in != nullptr, "null gol_uvlight_in");
  GOL_REQUIRE(in->B >= 0 && in->L >= 0 && in->HW >= 0, "negative size");
  GOL_REQUIRE(in->B <= 65535, "B > 65535");
  GOL_REQUIRE(in->n_pow >= 0 && in->n_pow <= GOL_UV_MAX_POW, "n_pow out of range");
  if (in->B == 0 || in->HW == 0) return GOL_OK;
  GOL_REQUIRE(in->p_uv && in->nml && in->cam_pos, "null input");
  GOL_REQUIRE(in->L == 0 || (in->light_pos && in->light_intensity), "null light arrays");
  GOL_REQUIRE(!ggx || (in->roughness && in->tex_mean), "GGX needs roughness and tex_mean");
  return GOL_OK;
}

}  // namespace

#define GOL_UV_GRID dim3 grid(gol_cdiv(in->HW, 256), in->B); hipStream_t s = (hipStream_t)stream

extern "C" int gol_uvlight_phong_fwd(const gol_uvlight_in* in, float* diff, float* spec, void* stream) {
  int rc = check(in, false);
  if (rc != GOL_OK) return rc;
  if (in->B == 0 || in->HW == 0) return GOL_OK;
  GOL_REQUIRE(diff && (spec || in->n_pow == 0), "null output");
  GOL_UV_GRID;
  phong_kernel<false><<<grid, 256, 0, s>>>(*in, diff, spec, nullptr, nullptr, nullptr, nullptr);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_uvlight_phong_bwd(const gol_uvlight_in* in, const float* u_diff, const float* u_spec,
                                     float* g_p_uv, float* g_nml, void* stream) {
  int rc = check(in, false);
  if (rc != GOL_OK) return rc;
  if (in->B == 0 || in->HW == 0) return GOL_OK;
  GOL_REQUIRE(u_diff && (u_spec || in->n_pow == 0) && g_p_uv && g_nml, "null pointer");
  GOL_UV_GRID;
  phong_kernel<true><<<grid, 256, 0, s>>>(*in, nullptr, nullptr, u_diff, u_spec, g_p_uv, g_nml);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_uvlight_ggx_fwd(const gol_uvlight_in* in, float* feat, float* rgb, void* stream) {
  int rc = check(in, true);
  if (rc != GOL_OK) return rc;
  if (in->B == 0 || in->HW == 0) return GOL_OK;
  GOL_REQUIRE(feat && rgb, "null output");
  GOL_REQUIRE(in->L > 0, "GGX texture is a mean over lights: L must be > 0");
  GOL_UV_GRID;
  ggx_kernel<false><<<grid, 256, 0, s>>>(*in, feat, rgb, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_uvlight_ggx_bwd(const gol_uvlight_in* in, const float* u_feat, const float* u_rgb, float* g_p_uv,
                                   float* g_nml, float* g_roughness, float* g_tex, void* stream) {
  int rc = check(in, true);
  if (rc != GOL_OK) return rc;
  if (in->B == 0 || in->HW == 0) return GOL_OK;
  GOL_REQUIRE(u_feat && u_rgb && g_p_uv && g_nml && g_roughness && g_tex, "null pointer");
  GOL_REQUIRE(in->L > 0, "L must be > 0");
  GOL_UV_GRID;
  ggx_kernel<true><<<grid, 256, 0, s>>>(*in, nullptr, nullptr, u_feat, u_rgb, g_p_uv, g_nml, g_roughness, g_tex);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
