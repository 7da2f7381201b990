// shade.hip -- fused RGCA shading tail (SH diffuse + Gaussian activations + SG / env-map specular),
// forward + backward, gfx950.
//
// Replaces the chain of ATen kernels of PrimDecoder.forward after the decoders
// (/root/reference/ca_code/models/rgca.py:505-588, training extra :590-618) including the specular
// term (extensions/sgutils/sg.cu:27-175 with w_type 0, or ca_code/utils/envmap.py:284-292 +
// ca_code/utils/mipmap_sampler.py:13-69).  This is THE HBM-bound stage of the frame: 129 decoder
// channels (516 B) per Gaussian in, 34 floats out.  Design:
//   * the decoder tensors are consumed in their native planar [B,C,N] layout: each channel plane is
//     read once with 16-byte-per-lane coalesced loads (V = 4 consecutive Gaussians per lane), the SH
//     dot products are streaming FMAs against wave-uniform light coefficients (scalar loads);
//   * outputs are written in the [B,N,k] layout the reference's preds/losses use, as float4 stores
//     (4 Gaussians x k floats are contiguous);
//   * backward never re-reads the 113 SH channels: d(loss)/d(f_vnocond) of an SH channel is just
//     upstream x light coefficient, so it is write-only (452 B) -- the re-read the PyTorch graph
//     would do (another 452 B) is gone.  Only the 12 geometry channels + f_vcond are re-read.
#include <cstring>

#include "gol_project.h"

namespace {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kSqrt2Pi23 = 3.03352966508f;     // sg.cu:20
constexpr float kInvSqrt2Pi23 = 0.32964899322f;  // sg.cu:21
constexpr float kNormEps = 1e-12f;               // torch F.normalize default eps

template <int V>
__device__ __forceinline__ void ldv(const float* __restrict__ p, float (&x)[V]) {
  if constexpr (V == 4) {
    // streamed once: non-temporal so the planes do not evict the env-map texels from L2
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
  } else if constexpr (V == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 t = __builtin_nontemporal_load(reinterpret_cast<const f2*>(p));
    x[0] = t.x; x[1] = t.y;
  } else {
#pragma unroll
    for (int v = 0; v < V; ++v) x[v] = p[v];
  }
}
template <int V>
__device__ __forceinline__ void stv(float* __restrict__ p, const float (&x)[V]) {
  if constexpr (V == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
  } else {
#pragma unroll
    for (int v = 0; v < V; ++v) p[v] = x[v];
  }
}
// [.,N,K] row-major: the K values of V consecutive Gaussians are V*K contiguous floats
template <int V, int K>
__device__ __forceinline__ void ld_aos(const float* __restrict__ base, size_t g0, float (&x)[K][V]) {
  float buf[V * K];
  const float* p = base + g0 * K;
  if constexpr ((V * K) % 4 == 0 && V == 4) {
#pragma unroll
    for (int j = 0; j < V * K / 4; ++j) {
      const float4 t = reinterpret_cast<const float4*>(p)[j];
      buf[4 * j] = t.x; buf[4 * j + 1] = t.y; buf[4 * j + 2] = t.z; buf[4 * j + 3] = t.w;
    }
  } else if constexpr (V == 2) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float2 t = reinterpret_cast<const float2*>(p)[j];
      buf[2 * j] = t.x; buf[2 * j + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < V * K; ++j) buf[j] = p[j];
  }
#pragma unroll
  for (int v = 0; v < V; ++v)
#pragma unroll
    for (int k = 0; k < K; ++k) x[k][v] = buf[v * K + k];
}
template <int V, int K>
__device__ __forceinline__ void st_aos(float* __restrict__ base, size_t g0, const float (&x)[K][V]) {
  float buf[V * K];
#pragma unroll
  for (int v = 0; v < V; ++v)
#pragma unroll
    for (int k = 0; k < K; ++k) buf[v * K + k] = x[k][v];
  float* p = base + g0 * K;
  if constexpr ((V * K) % 4 == 0 && V == 4) {
#pragma unroll
    for (int j = 0; j < V * K / 4; ++j)
      reinterpret_cast<float4*>(p)[j] = make_float4(buf[4 * j], buf[4 * j + 1], buf[4 * j + 2], buf[4 * j + 3]);
  } else if constexpr (V == 2) {
#pragma unroll
    for (int j = 0; j < K; ++j) reinterpret_cast<float2*>(p)[j] = make_float2(buf[2 * j], buf[2 * j + 1]);
  } else {
#pragma unroll
    for (int j = 0; j < V * K; ++j) p[j] = buf[j];
  }
}
// optional upstream gradient: NULL pointer = zeros
template <int V, int K>
__device__ __forceinline__ void ld_aos_opt(const float* __restrict__ base, size_t g0, float (&x)[K][V]) {
  if (base) {
    ld_aos<V, K>(base, g0, x);
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int v = 0; v < V; ++v) x[k][v] = 0.f;
  }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // torch threshold=20

// ---- environment lookup ---------------------------------------------------------------------
struct Bilinear {
  float val[3];
  float d_ix[3], d_iy[3];  // d val / d (unnormalised pixel coordinate), 0 outside the clamp range
  float mx, my;            // coordinate-clamp gradient masks (torch clip_coordinates_set_grad)
};

// F.grid_sample(mode=bilinear, padding_mode=border, align_corners=False) of a [3,h,w] image at
// normalised (u,v) in [-1,1]
template <bool PACKED>
__device__ __forceinline__ Bilinear bilinear_border(const float* __restrict__ img, int h, int w, float u, float v) {
  Bilinear r;
  float ix = ((u + 1.f) * (float)w - 1.f) * 0.5f, iy = ((v + 1.f) * (float)h - 1.f) * 0.5f;
  r.mx = (ix > 0.f && ix < (float)(w - 1)) ? 1.f : 0.f;
  r.my = (iy > 0.f && iy < (float)(h - 1)) ? 1.f : 0.f;
  ix = fminf(fmaxf(ix, 0.f), (float)(w - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(h - 1));
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx0, wx0 = 1.f - wx1, wy1 = iy - fy0, wy0 = 1.f - wy1;
  const bool bx = x1 < w, by = y1 < h;
  float t00[3], t01[3], t10[3], t11[3];
  if constexpr (PACKED) {
    // [h,w,16] FOOTPRINT records: record (y0, x0) holds the four taps of the bilinear footprint whose top-left texel
    // it is (taps beyond the border already zero), 64 bytes on a 64-byte boundary -- a lookup is ONE sector, whatever
    // its position.  ([h,w,4] texels cost 2.5 sectors per lookup on average: the footprint spans two rows, and the
    // lookups of neighbouring Gaussians are unrelated, so nothing of the rest of a sector is ever used.)
    const float4* q = reinterpret_cast<const float4*>(img) + (size_t)(y0 * w + x0) * 4;
    const float4 a = q[0], b2 = q[1], c2 = q[2], d2 = q[3];
    t00[0] = a.x; t00[1] = a.y; t00[2] = a.z; t01[0] = b2.x; t01[1] = b2.y; t01[2] = b2.z;
    t10[0] = c2.x; t10[1] = c2.y; t10[2] = c2.z; t11[0] = d2.x; t11[1] = d2.y; t11[2] = d2.z;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* p = img + (size_t)c * h * w;
      t00[c] = p[y0 * w + x0];
      t01[c] = bx ? p[y0 * w + x1] : 0.f;
      t10[c] = by ? p[y1 * w + x0] : 0.f;
      t11[c] = (bx && by) ? p[y1 * w + x1] : 0.f;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    r.val[c] = wy0 * (wx0 * t00[c] + wx1 * t01[c]) + wy1 * (wx0 * t10[c] + wx1 * t11[c]);
    r.d_ix[c] = wy0 * (t01[c] - t00[c]) + wy1 * (t11[c] - t10[c]);
    r.d_iy[c] = wx0 * (t10[c] - t00[c]) + wx1 * (t11[c] - t01[c]);
  }
  return r;
}

struct EnvSample {
  float val[3];      // lerp of the two mip levels (before clamp(max=1))
  float d_u[3], d_v[3];
};

// the address half of bilinear_border<true>: which footprint record, with which weights
struct Footprint {
  const float4* q;         // the [h,w,16] record of the top-left texel
  float wx1, wy1, mx, my;  // tap weights (wx0 = 1 - wx1, ...), coordinate-clamp gradient masks
};
__device__ __forceinline__ Footprint footprint_of(const float* __restrict__ img, int h, int w, float u, float v) {
  Footprint f;
  float ix = ((u + 1.f) * (float)w - 1.f) * 0.5f, iy = ((v + 1.f) * (float)h - 1.f) * 0.5f;
  f.mx = (ix > 0.f && ix < (float)(w - 1)) ? 1.f : 0.f;
  f.my = (iy > 0.f && iy < (float)(h - 1)) ? 1.f : 0.f;
  ix = fminf(fmaxf(ix, 0.f), (float)(w - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(h - 1));
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  f.wx1 = ix - fx0; f.wy1 = iy - fy0;
  f.q = reinterpret_cast<const float4*>(img) + (size_t)((int)fy0 * w + (int)fx0) * 4;
  return f;
}
// ... and the arithmetic half, on the four taps already in registers (same expressions as bilinear_border)
__device__ __forceinline__ Bilinear footprint_eval(const Footprint& f, const float4& a, const float4& b2, const float4& c2,
                                                   const float4& d2) {
  Bilinear r;
  r.mx = f.mx; r.my = f.my;
  const float wx1 = f.wx1, wx0 = 1.f - wx1, wy1 = f.wy1, wy0 = 1.f - wy1;
  const float t00[3] = {a.x, a.y, a.z}, t01[3] = {b2.x, b2.y, b2.z}, t10[3] = {c2.x, c2.y, c2.z}, t11[3] = {d2.x, d2.y, d2.z};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    r.val[c] = wy0 * (wx0 * t00[c] + wx1 * t01[c]) + wy1 * (wx0 * t10[c] + wx1 * t11[c]);
    r.d_ix[c] = wy0 * (t01[c] - t00[c]) + wy1 * (t11[c] - t10[c]);
    r.d_iy[c] = wx0 * (t10[c] - t00[c]) + wx1 * (t11[c] - t01[c]);
  }
  return r;
}

// mipmap_grid_sample (mipmap_sampler.py:13-69): level selection has no gradient.
__device__ __forceinline__ EnvSample env_lookup(const gol_shade_in& in, int b, float u, float v, float level) {
  EnvSample e;
  const int q = in.n_mips;
  int d1 = 0;
  float a = 0.f;
  if (q > 1) {
    const float lam = fminf(fmaxf(level, 0.f), (float)(q - 1) - 1e-6f);
    const float fl = floorf(lam);
    d1 = (int)fl;
    a = lam - fl;
  }
  const bool packed = in.mips_packed[0] != nullptr;
  if (in.mips_shared) b = 0;  // one pyramid for all views; only lightrot is per view
  const int d2 = min(d1 + 1, q - 1);
  // the level index differs per lane: indexing the kernel-argument arrays with it makes the compiler FETCH mip_h / mip_w /
  // the level pointer from the argument segment with vector loads -- two more dependent memory round trips in front of the
  // lookup.  A select chain over the (wave-uniform, SGPR-resident) entries costs a few VALU instead.
  int h0 = in.mip_h[0], w0 = in.mip_w[0], h1 = h0, w1 = w0;
  const float *m0 = packed ? in.mips_packed[0] : in.mips[0], *m1 = m0;
#pragma unroll
  for (int l = 1; l < GOL_MAX_MIPS; ++l) {
    const float* ml = packed ? in.mips_packed[l] : in.mips[l];
    if (d1 == l) { h0 = in.mip_h[l]; w0 = in.mip_w[l]; m0 = ml; }
    if (d2 == l) { h1 = in.mip_h[l]; w1 = in.mip_w[l]; m1 = ml; }
  }
  Bilinear s0, s1;
  if (packed) {
    // both levels' record addresses first, then the eight 16-byte loads back to back: ONE memory round trip per lookup
    // instead of two dependent ones (round 6; the two bilinear_border calls each waited for their own four loads)
    const Footprint f0 = footprint_of(m0 + (size_t)b * 16 * h0 * w0, h0, w0, u, v);
    const Footprint f1 = footprint_of(m1 + (size_t)b * 16 * h1 * w1, h1, w1, u, v);
    const float4 a0 = f0.q[0], a1 = f0.q[1], a2 = f0.q[2], a3 = f0.q[3];
    const float4 b0 = f1.q[0], b1 = f1.q[1], b2 = f1.q[2], b3 = f1.q[3];
    s0 = footprint_eval(f0, a0, a1, a2, a3);
    s1 = footprint_eval(f1, b0, b1, b2, b3);
  } else {
    s0 = bilinear_border<false>(m0 + (size_t)b * 3 * h0 * w0, h0, w0, u, v);
    s1 = s0;
    if (q > 1) s1 = bilinear_border<false>(m1 + (size_t)b * 3 * h1 * w1, h1, w1, u, v);
  }
  const float msc = in.mips_scale != 0.f ? in.mips_scale : 1.f;  // the driver's per-frame scale of the whole pyramid
  if (q > 1) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      e.val[c] = s0.val[c] + a * (s1.val[c] - s0.val[c]);  // th.lerp
      const float du0 = s0.d_ix[c] * s0.mx * (0.5f * (float)w0), du1 = s1.d_ix[c] * s1.mx * (0.5f * (float)w1);
      const float dv0 = s0.d_iy[c] * s0.my * (0.5f * (float)h0), dv1 = s1.d_iy[c] * s1.my * (0.5f * (float)h1);
      e.d_u[c] = du0 + a * (du1 - du0);
      e.d_v[c] = dv0 + a * (dv1 - dv0);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      e.val[c] = s0.val[c];
      e.d_u[c] = s0.d_ix[c] * s0.mx * (0.5f * (float)w0);
      e.d_v[c] = s0.d_iy[c] * s0.my * (0.5f * (float)h0);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { e.val[c] *= msc; e.d_u[c] *= msc; e.d_v[c] *= msc; }
  return e;
}

// ---- per-Gaussian shading state shared by forward and backward --------------------------------
struct Geo {
  float pos[3], qn, q[4], sp[3], opac, sigma, e01;  // e01 = 0.1*exp(rough) before the 0.01 floor
  float vis, m[3], mn, n[3];                         // m = dnml + nmlbase, mn = |m|
  float dvec[3], dn, view[3], vdn, ref[3];           // dvec = pos - campos, dn = |dvec|
};

__device__ __forceinline__ Geo make_geo(const float (&g)[12], const float (&fc)[4], const float (&pb)[3],
                                        const float (&nb)[3], const float* __restrict__ cam) {
  Geo s;
#pragma unroll
  for (int k = 0; k < 3; ++k) s.pos[k] = g[k] + pb[k];
  s.qn = sqrtf(g[3] * g[3] + g[4] * g[4] + g[5] * g[5] + g[6] * g[6]);
  const float iq = 1.f / fmaxf(s.qn, kNormEps);
#pragma unroll
  for (int k = 0; k < 4; ++k) s.q[k] = g[3 + k] * iq;
#pragma unroll
  for (int k = 0; k < 3; ++k) s.sp[k] = softplusf(g[7 + k]);
  s.opac = sigmoidf(g[10]);
  s.e01 = 0.1f * expf(g[11]);
  s.sigma = fmaxf(s.e01, 0.01f);
  s.vis = sigmoidf(fc[0]);
#pragma unroll
  for (int k = 0; k < 3; ++k) s.m[k] = fc[1 + k] + nb[k];
  s.mn = sqrtf(s.m[0] * s.m[0] + s.m[1] * s.m[1] + s.m[2] * s.m[2]);
  const float im = 1.f / fmaxf(s.mn, kNormEps);
#pragma unroll
  for (int k = 0; k < 3; ++k) s.n[k] = s.m[k] * im;
#pragma unroll
  for (int k = 0; k < 3; ++k) s.dvec[k] = s.pos[k] - cam[k];
  s.dn = sqrtf(s.dvec[0] * s.dvec[0] + s.dvec[1] * s.dvec[1] + s.dvec[2] * s.dvec[2]);
  const float id = 1.f / fmaxf(s.dn, kNormEps);
#pragma unroll
  for (int k = 0; k < 3; ++k) s.view[k] = s.dvec[k] * id;
  s.vdn = s.view[0] * s.n[0] + s.view[1] * s.n[1] + s.view[2] * s.n[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) s.ref[k] = s.view[k] - 2.f * s.vdn * s.n[k];
  return s;
}

// specular radiance before the spec_vis factor; ENV: min(sample, 1) (rgca.py:556)
template <bool ENV>
__device__ __forceinline__ void spec_forward(const gol_shade_in& in, int b, const Geo& s, float (&spec)[3],
                                             EnvSample* keep = nullptr) {
  if constexpr (ENV) {
    const float* R = in.lightrot + 9 * b;
    const float rx = R[0] * s.ref[0] + R[1] * s.ref[1] + R[2] * s.ref[2];
    const float ry = R[3] * s.ref[0] + R[4] * s.ref[1] + R[5] * s.ref[2];
    const float rz = R[6] * s.ref[0] + R[7] * s.ref[1] + R[8] * s.ref[2];
    const float u = atan2f(rx, rz) * (1.f / kPi);
    // polar angle = acos(r_y) (envmap.py:289) evaluated as atan2(sqrt(r_x^2 + r_z^2), r_y): the same angle for a unit vector,
    // without acos' loss of the rounding of r_y near the poles (acos'(y) = -1 / sqrt(1 - y^2) amplifies it without bound)
    const float v = 2.f * atan2f(sqrtf(rx * rx + rz * rz), ry) * (1.f / kPi) - 1.f;
    const EnvSample e = env_lookup(in, b, u, v, 5.f * s.sigma);
    if (keep) *keep = e;
#pragma unroll
    for (int c = 0; c < 3; ++c) spec[c] = fminf(e.val[c], 1.f);
  } else {
    // evaluate_gaussian(normalize(ref), sigma, ...), w_type 0 (sgutils.py:75-76, sg.cu:49-72)
    const float rn = 1.f / fmaxf(sqrtf(s.ref[0] * s.ref[0] + s.ref[1] * s.ref[1] + s.ref[2] * s.ref[2]), kNormEps);
    const float lx = s.ref[0] * rn, ly = s.ref[1] * rn, lz = s.ref[2] * rn;
    const int nL = in.n_lights[b];
    const float* lv = in.light_intensity + (size_t)b * in.L * 3;
    const float* lp = in.light_pos + (size_t)b * in.L * 3;
    const float inv_sigma = 1.f / s.sigma, norm = 1.f / (s.sigma * kSqrt2Pi23);
    spec[0] = spec[1] = spec[2] = 0.f;
    for (int l = 0; l < nL; ++l) {
      const float dx = lp[3 * l] - s.pos[0], dy = lp[3 * l + 1] - s.pos[1], dz = lp[3 * l + 2] - s.pos[2];
      const float r = rsqrtf(dx * dx + dy * dy + dz * dz);
      const float c = fminf(1.f, fmaxf(-1.f, (dx * lx + dy * ly + dz * lz) * r));
      const float ang = acosf(c) * inv_sigma;
      const float w = __expf(-0.5f * ang * ang) * norm;
      spec[0] += lv[3 * l] * w; spec[1] += lv[3 * l + 1] * w; spec[2] += lv[3 * l + 2] * w;
    }
  }
}

// Forward, three phases per 256-Gaussian workgroup (round 6; the mirror image of the backward below):
//   phase G  one lane per Gaussian, FIRST: the 22 geometry / f_vcond / base planes, the shading state, the env-map lookup
//            (two mip levels = eight 16-byte loads issued together) and every output that does not need the SH sums --
//            23 of the 37 output floats and, PROJ, the EWA projection (gol_project.h: the arithmetic of gol_project_fwd on
//            the position / quaternion / clamped scale / opacity still in registers).  The two dependent memory round
//            trips of a lookup (planes -> address -> records) sit at the START of the wave's life, under the plane
//            streams of the CU's other workgroups, and 13 floats per Gaussian are carried over phase S;
//   phase S  wave w streams SH planes w, w+4, ... of the block's 256 Gaussians (lane -> 4 consecutive Gaussians, 16-byte
//            non-temporal loads, 1 KiB contiguous per wave-instruction; light coefficients through SGPRs) into partial sums
//            that meet in LDS;
//   phase F  colour = max(max(albedo * D, 0) + spec * vis, 0) and, PROJ, the 64-byte raster record binning and the
//            rasterizer read (so the render direction starts at the tile count, gol_render_fwd_projected, and the 56 bytes
//            per Gaussian a projection kernel would read back are never fetched).
// Rounds 1-5 streamed the SH planes first with 4 Gaussians per lane and did the geometry of all four at the END of the
// wave: 244 VGPRs (2 waves per SIMD) and eight serialised lookup round trips per lane with nothing left to hide them.
template <bool ENV, bool RAND, bool PROJ, bool VEC4>
__global__ __launch_bounds__(256, 4) void shade_fwd_kernel(const gol_shade_in in, const gol_shade_out out,
                                                        const gol_shade_proj pj) {
  __shared__ __attribute__((aligned(16))) float s_D[4][RAND ? 6 : 3][256];  // per-wave partial SH sums (light, random light)
  const int b = blockIdx.y;
  const int blk0 = blockIdx.x * 256;
  const int N = in.N;
  const int ncol = in.n_color_coef, nmono = in.n_mono_coef, ncoef = ncol + nmono, nd = 3 * ncol + nmono;
  const float* Fv = in.f_vnocond + (size_t)b * (nd + 12) * N;
  const int i = blk0 + threadIdx.x;
  const bool live = i < N;
  const size_t g0 = (size_t)b * N + i;

  // ---- phase G --------------------------------------------------------------------------------------------------------
  float c_spv[3] = {0.f, 0.f, 0.f}, c_alb[3] = {0.f, 0.f, 0.f};   // carried: spec * vis, albedo
  float c_rec[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};             // carried (PROJ): xy, conic, effective opacity, depth
  if (live) {
    float g[12], fc[4], pb[3], nb[3];
#pragma unroll
    for (int j = 0; j < 12; ++j) g[j] = __builtin_nontemporal_load(Fv + (size_t)(nd + j) * N + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) fc[j] = __builtin_nontemporal_load(in.f_vcond + ((size_t)b * 4 + j) * N + i);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      pb[j] = __builtin_nontemporal_load(in.postex + ((size_t)b * 3 + j) * N + i);
      nb[j] = __builtin_nontemporal_load(in.tn + ((size_t)b * 3 + j) * N + i);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) c_alb[c] = in.albedo[(size_t)i * 3 + c];
    const Geo s = make_geo(g, fc, pb, nb, in.campos + 3 * b);
    float spec[3];
    EnvSample es;
    spec_forward<ENV>(in, b, s, spec, ENV ? &es : nullptr);
    float o_sc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      c_spv[c] = spec[c] * s.vis;
      o_sc[c] = fminf(fmaxf(s.sp[c], in.primscale_min), in.primscale_max);
    }
    // [B,N,k] rows: lane i writes k consecutive floats, the wave 64 k of them contiguously
    float* p;
    out.opacity[g0] = s.opac;
    out.sigma[g0] = s.sigma;
    out.spec_vis[g0] = s.vis;
    p = out.primpos + g0 * 3; p[0] = s.pos[0]; p[1] = s.pos[1]; p[2] = s.pos[2];
    *reinterpret_cast<float4*>(out.primqvec + g0 * 4) = make_float4(s.q[0], s.q[1], s.q[2], s.q[3]);
    p = out.primscale + g0 * 3; p[0] = o_sc[0]; p[1] = o_sc[1]; p[2] = o_sc[2];
    p = out.primscale_preclip + g0 * 3; p[0] = s.sp[0]; p[1] = s.sp[1]; p[2] = s.sp[2];
    p = out.spec_nml + g0 * 3; p[0] = s.n[0]; p[1] = s.n[1]; p[2] = s.n[2];
    p = out.spec_dnml + g0 * 3; p[0] = fc[1]; p[1] = fc[2]; p[2] = fc[3];
    p = out.spec_color + g0 * 3; p[0] = c_spv[0]; p[1] = c_spv[1]; p[2] = c_spv[2];
    p = out.primnmlbase + g0 * 3; p[0] = nb[0]; p[1] = nb[1]; p[2] = nb[2];
    if constexpr (ENV) {
      if (out.env_saved) {
        p = out.env_saved + g0 * 9;
#pragma unroll
        for (int c = 0; c < 3; ++c) { p[c] = es.val[c]; p[3 + c] = es.d_u[c]; p[6 + c] = es.d_v[c]; }
      }
    }
    if constexpr (PROJ) {
      const gol_proj::View view = gol_proj::view_of(pj.viewmats, pj.intrins, b, pj.img_h, pj.img_w, 16, pj.clip_thresh);
      const float sc[3] = {pj.glob_scale * o_sc[0], pj.glob_scale * o_sc[1], pj.glob_scale * o_sc[2]};
      const gol_proj::Projected o = gol_proj::project_point(s.pos, s.q, sc, view);
      const float op_eff = s.opac * o.comp;
      c_rec[0] = o.xy[0]; c_rec[1] = o.xy[1]; c_rec[2] = o.conic[0]; c_rec[3] = o.conic[1]; c_rec[4] = o.conic[2];
      c_rec[5] = op_eff; c_rec[6] = o.depth;
      *reinterpret_cast<float2*>(pj.xys + g0 * 2) = make_float2(o.xy[0], o.xy[1]);
      pj.depths[g0] = o.depth;
      pj.radii[g0] = o.radius;
      p = pj.conics + g0 * 3; p[0] = o.conic[0]; p[1] = o.conic[1]; p[2] = o.conic[2];
      pj.comp[g0] = o.comp;
      pj.opac_eff[g0] = op_eff;
    }
  }

  // ---- phase S --------------------------------------------------------------------------------------------------------
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j0 = blk0 + 4 * lane;  // first of this lane's 4 consecutive Gaussians
  const float* Lsh = in.light_sh + (size_t)b * 3 * ncoef;
  const float* Lr = RAND ? in.light_sh_rand + (size_t)b * 3 * ncoef : nullptr;
  float D[3][4], Dr[3][4];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) { D[c][v] = 0.f; Dr[c][v] = 0.f; }
  // VEC4 (N % 4 == 0: planes 16-byte aligned, a lane's 4 Gaussians in range together): lanes past the end of the view
  // re-read its last 4 Gaussians instead of branching around every load -- their sums land in LDS slots nobody reads
  const int j0c = VEC4 ? min(j0, N - 4) : j0;
  auto plane = [&](int ch, float (&x)[4]) {
    const float* src = Fv + (size_t)ch * N + j0c;
    if (VEC4) {
      ldv<4>(src, x);
    } else {
#pragma unroll
      for (int v = 0; v < 4; ++v) if (j0 + v < N) x[v] = src[v];
    }
  };
  // colour SH: channel c*ncol + k  (rgca.py:508-510); the three colours of one k together: 3 x unroll loads in flight
  constexpr int kUnrollColour = RAND ? 2 : 4, kUnrollMono = RAND ? 4 : 8;  // loads in flight vs the 128-VGPR budget
#pragma unroll kUnrollColour
  for (int k = wave; k < ncol; k += 4) {
    float x[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < 3; ++c) plane(c * ncol + k, x[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float l = Lsh[c * ncoef + k];
      const float lr = RAND ? Lr[c * ncoef + k] : 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) { D[c][v] += x[c][v] * l; if (RAND) Dr[c][v] += x[c][v] * lr; }
    }
  }
  // monochrome SH shared by the three colour channels (rgca.py:511-514)
#pragma unroll kUnrollMono
  for (int k = wave; k < nmono; k += 4) {
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    plane(3 * ncol + k, x);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float l = Lsh[c * ncoef + ncol + k];
      const float lr = RAND ? Lr[c * ncoef + ncol + k] : 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) { D[c][v] += x[v] * l; if (RAND) Dr[c][v] += x[v] * lr; }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    *reinterpret_cast<float4*>(&s_D[wave][c][4 * lane]) = make_float4(D[c][0], D[c][1], D[c][2], D[c][3]);
    if (RAND) *reinterpret_cast<float4*>(&s_D[wave][3 + c][4 * lane]) = make_float4(Dr[c][0], Dr[c][1], Dr[c][2], Dr[c][3]);
  }
  __syncthreads();

  // ---- phase F --------------------------------------------------------------------------------------------------------
  if (!live) return;
  float Ds[3], col[3], diff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    Ds[c] = (s_D[0][c][threadIdx.x] + s_D[1][c][threadIdx.x]) + (s_D[2][c][threadIdx.x] + s_D[3][c][threadIdx.x]);
    diff[c] = c_alb[c] * Ds[c];
    col[c] = fmaxf(fmaxf(diff[c], 0.f) + c_spv[c], 0.f);
  }
  float* p;
  p = out.color + g0 * 3; p[0] = col[0]; p[1] = col[1]; p[2] = col[2];
  p = out.diff_color + g0 * 3; p[0] = diff[0]; p[1] = diff[1]; p[2] = diff[2];
  p = out.diff_sum + g0 * 3; p[0] = Ds[0]; p[1] = Ds[1]; p[2] = Ds[2];
  if (RAND) {
    p = out.color_rand + g0 * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      p[c] = fmaxf((s_D[0][3 + c][threadIdx.x] + s_D[1][3 + c][threadIdx.x]) +
                   (s_D[2][3 + c][threadIdx.x] + s_D[3][3 + c][threadIdx.x]), 0.f);
  }
  if constexpr (PROJ)
    gol_record_write(pj.records + g0 * GOL_SPLAT_RECORD, c_rec[0], c_rec[1], c_rec[2], c_rec[3], c_rec[4], c_rec[5],
                     col[0], col[1], col[2], c_rec[6]);
}

// Backward, two phases per 256-Gaussian workgroup:
//   phase 1  one lane per Gaussian: recompute the shading state, push the upstream gradients through
//            it, write the 12 geometry / 4 f_vcond / base / albedo gradients, park d loss/d(SH sums)
//            (6 floats per Gaussian) in LDS;
//   phase 2  wave w owns SH channels w, w+4, ...: every lane turns 4 consecutive Gaussians' parked
//            sums into one 16-byte store per channel plane (1 KiB contiguous per wave-instruction).
// Phase 2 is 452 of the ~650 B written per Gaussian, so the bulk of the traffic is full-width.
// PROJ: the projection backward runs as the prologue of phase 1 -- the Gaussian's 64-byte gradient record out of the raster
// backward (GOL_GRAD_RECORD: rgb | opacity | xy | conic | depth) is pushed through gol_proj::project_vjp on the recomputed
// position / quaternion / scale / opacity and ADDED to the upstream gradients of color / opacity / primpos / primqvec /
// primscale (which other consumers of those outputs may also feed): gol_project_bwd's 56 B written + 56 B re-read per
// Gaussian and its launch are gone.
// (launch bound: 3 waves per SIMD = 168 VGPRs.  The env variant sits at 160-171 depending on unrelated code around it; at
// 171 the kernel drops to 2 waves per SIMD and runs 10 % longer -- round 6 A/B, 301 -> 332 us per 8 views.)
template <bool ENV, bool RAND, bool VEC4, bool PROJ>
__global__ __launch_bounds__(256, 3) void shade_bwd_kernel(const gol_shade_in in, const gol_shade_out saved,
                                                        const gol_shade_out_grad up, const gol_shade_in_grad gin,
                                                        const gol_shade_proj pj, const float* __restrict__ grad_records,
                                                        int with_depth) {
  constexpr int V = 1;
  __shared__ __attribute__((aligned(16))) float s_g[6][256];  // gD[3], gDr[3] per Gaussian of the block
  const int b = blockIdx.y;
  const int blk0 = blockIdx.x * 256;
  const int i0 = blk0 + threadIdx.x;
  const int N = in.N;
  const int ncol = in.n_color_coef, nmono = in.n_mono_coef, ncoef = ncol + nmono, nd = 3 * ncol + nmono;
  const size_t view0 = (size_t)b * (nd + 12) * N;
  const float* Lsh = in.light_sh + (size_t)b * 3 * ncoef;
  const float* Lr = RAND ? in.light_sh_rand + (size_t)b * 3 * ncoef : nullptr;
  if (i0 < N) {
    const float* F = in.f_vnocond + view0 + i0;
    float* GF = gin.f_vnocond + view0 + i0;
    const size_t g0 = (size_t)b * N + i0;

    float g[12][V], fc[4][V], pb[3][V], nb[3][V], alb[3][V], Dsum[3][V], crand[3][V];
#pragma unroll
    for (int j = 0; j < 12; ++j) ldv<V>(F + (size_t)(nd + j) * N, g[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) ldv<V>(in.f_vcond + ((size_t)b * 4 + j) * N + i0, fc[j]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      ldv<V>(in.postex + ((size_t)b * 3 + j) * N + i0, pb[j]);
      ldv<V>(in.tn + ((size_t)b * 3 + j) * N + i0, nb[j]);
    }
    ld_aos<V, 3>(in.albedo, (size_t)i0, alb);
    ld_aos<V, 3>(saved.diff_sum, g0, Dsum);
    if (RAND) ld_aos<V, 3>(saved.color_rand, g0, crand);

    float u_color[3][V], u_op[1][V], u_pos[3][V], u_q[4][V], u_sc[3][V], u_sp[3][V], u_sig[1][V], u_vis[1][V];
    float u_n[3][V], u_dn[3][V], u_diff[3][V], u_spec[3][V], u_nb[3][V], u_rand[3][V];
    ld_aos_opt<V, 3>(up.color, g0, u_color);
    ld_aos_opt<V, 1>(up.opacity, g0, u_op);
    ld_aos_opt<V, 3>(up.primpos, g0, u_pos);
    ld_aos_opt<V, 4>(up.primqvec, g0, u_q);
    ld_aos_opt<V, 3>(up.primscale, g0, u_sc);
    ld_aos_opt<V, 3>(up.primscale_preclip, g0, u_sp);
    ld_aos_opt<V, 1>(up.sigma, g0, u_sig);
    ld_aos_opt<V, 1>(up.spec_vis, g0, u_vis);
    ld_aos_opt<V, 3>(up.spec_nml, g0, u_n);
    ld_aos_opt<V, 3>(up.spec_dnml, g0, u_dn);
    ld_aos_opt<V, 3>(up.diff_color, g0, u_diff);
    ld_aos_opt<V, 3>(up.spec_color, g0, u_spec);
    ld_aos_opt<V, 3>(up.primnmlbase, g0, u_nb);
    ld_aos_opt<V, 3>(RAND ? up.color_rand : nullptr, g0, u_rand);

    float gD[3][V], gDr[3][V];  // d loss / d (SH sum) for the light and the random light
    float gg[12][V], gfc[4][V], gpb[3][V], gnb[3][V], galb[3][V];
    const float* cam = in.campos + 3 * b;
    constexpr int v = 0;
    {
      float g1[12], f1[4], p3[3], n3[3];
#pragma unroll
      for (int j = 0; j < 12; ++j) g1[j] = g[j][v];
#pragma unroll
      for (int j = 0; j < 4; ++j) f1[j] = fc[j][v];
#pragma unroll
      for (int j = 0; j < 3; ++j) { p3[j] = pb[j][v]; n3[j] = nb[j][v]; }
      const Geo s = make_geo(g1, f1, p3, n3, cam);
      if constexpr (PROJ) {
        const float4* R = reinterpret_cast<const float4*>(grad_records + g0 * GOL_GRAD_RECORD);
        const float4 r0 = R[0];
        u_color[0][v] += r0.x; u_color[1][v] += r0.y; u_color[2][v] += r0.z;
        if (pj.radii[g0] > 0) {
          const float4 r1 = R[1], r2 = R[2];
          gol_proj::ProjUp pu;
          pu.opac_eff = r0.w; pu.xy[0] = r1.x; pu.xy[1] = r1.y; pu.conic[0] = r1.z; pu.conic[1] = r1.w; pu.conic[2] = r2.x;
          pu.depth = with_depth ? r2.y : 0.f;
          pu.comp = 0.f;
          const gol_proj::View view = gol_proj::view_of(pj.viewmats, pj.intrins, b, pj.img_h, pj.img_w, 16, pj.clip_thresh);
          const float sc[3] = {pj.glob_scale * fminf(fmaxf(s.sp[0], in.primscale_min), in.primscale_max),
                               pj.glob_scale * fminf(fmaxf(s.sp[1], in.primscale_min), in.primscale_max),
                               pj.glob_scale * fminf(fmaxf(s.sp[2], in.primscale_min), in.primscale_max)};
          const float X[3] = {pj.conics[3 * g0], pj.conics[3 * g0 + 1], pj.conics[3 * g0 + 2]};
          const gol_proj::ProjGrad pg = gol_proj::project_vjp(s.pos, s.q, sc, pj.glob_scale, view, X, pj.comp[g0], pu, true,
                                                              s.opac, nullptr);
#pragma unroll
          for (int k = 0; k < 3; ++k) { u_pos[k][v] += pg.mean[k]; u_sc[k][v] += pg.scale[k]; }
#pragma unroll
          for (int k = 0; k < 4; ++k) u_q[k][v] += pg.quat[k];
          u_op[0][v] += pg.opacity;
        }
      }
      float spec[3];
      EnvSample e;
      bool have_env = false;
      if constexpr (ENV) {
        if (saved.env_saved) {  // 36 B re-read instead of 8 texel gathers (two mip levels x 4 taps)
          float es[9][V];
          ld_aos<V, 9>(saved.env_saved, g0, es);
#pragma unroll
          for (int c = 0; c < 3; ++c) { e.val[c] = es[c][v]; e.d_u[c] = es[3 + c][v]; e.d_v[c] = es[6 + c][v]; }
#pragma unroll
          for (int c = 0; c < 3; ++c) spec[c] = fminf(e.val[c], 1.f);
          have_env = true;
        }
      }
      if (!have_env) spec_forward<ENV>(in, b, s, spec, ENV ? &e : nullptr);

      // colour composition (rgca.py:572-575): color_out = max(max(diff,0) + spec*vis, 0)
      float g_spec[3], g_sraw[3], g_vis = u_vis[0][v];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float diff = alb[c][v] * Dsum[c][v];
        const float col = fmaxf(diff, 0.f) + spec[c] * s.vis;
        const float gc = (col >= 0.f) ? u_color[c][v] : 0.f;
        const float g_diff = u_diff[c][v] + ((diff >= 0.f) ? gc : 0.f);
        g_spec[c] = u_spec[c][v] + gc;
        g_vis += g_spec[c] * spec[c];
        g_sraw[c] = g_spec[c] * s.vis;
        gD[c][v] = g_diff * alb[c][v];
        galb[c][v] = g_diff * Dsum[c][v];
        gDr[c][v] = RAND ? ((crand[c][v] > 0.f) ? u_rand[c][v] : 0.f) : 0.f;
      }
      gfc[0][v] = g_vis * s.vis * (1.f - s.vis);

      // specular -> reflection direction (and roughness for the SG lobe)
      float g_ref[3] = {0.f, 0.f, 0.f}, g_sigma = u_sig[0][v];
      if constexpr (ENV) {
        const float* R = in.lightrot + 9 * b;
        const float rx = R[0] * s.ref[0] + R[1] * s.ref[1] + R[2] * s.ref[2];
        const float ry = R[3] * s.ref[0] + R[4] * s.ref[1] + R[5] * s.ref[2];
        const float rz = R[6] * s.ref[0] + R[7] * s.ref[1] + R[8] * s.ref[2];
        float g_u = 0.f, g_v = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float gs = (e.val[c] <= 1.f) ? g_sraw[c] : 0.f;  // clamp(max=1)
          g_u += gs * e.d_u[c];
          g_v += gs * e.d_v[c];
        }
        const float den = rx * rx + rz * rz;
        const float iden = den > 0.f ? 1.f / den : 0.f;
        const float grx = g_u * (1.f / kPi) * rz * iden;
        const float grz = -g_u * (1.f / kPi) * rx * iden;
        // d acos(r_y) / d r_y = -1 / sqrt(1 - r_y^2), with 1 - r_y^2 formed as r_x^2 + r_z^2: the difference cancels to a few
        // ulps of 1 near the poles, the sum keeps the relative precision of the two small components
        const float gry = (den > 0.f && ry > -1.f && ry < 1.f) ? g_v * (-2.f / kPi) * rsqrtf(den) : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) g_ref[k] = R[k] * grx + R[3 + k] * gry + R[6 + k] * grz;
      } else {
        // evaluate_gaussian backward, w_type 0 (sg.cu:113-130,160-163), then F.normalize backward
        const float rl = sqrtf(s.ref[0] * s.ref[0] + s.ref[1] * s.ref[1] + s.ref[2] * s.ref[2]);
        const float rn = 1.f / fmaxf(rl, kNormEps);
        const float lx = s.ref[0] * rn, ly = s.ref[1] * rn, lz = s.ref[2] * rn;
        const int nL = in.n_lights[b];
        const float* lv = in.light_intensity + (size_t)b * in.L * 3;
        const float* lp = in.light_pos + (size_t)b * in.L * 3;
        const float sg = s.sigma, s2 = sg * sg;
        const float inv_sg = 1.f / sg, inv_s3 = 1.f / (s2 * sg), inv_s4 = 1.f / (s2 * s2);  // per Gaussian, not per light
        float gx = 0.f, gy = 0.f, gz = 0.f, gs = 0.f;
        for (int l = 0; l < nL; ++l) {
          float dx = lp[3 * l] - s.pos[0], dy = lp[3 * l + 1] - s.pos[1], dz = lp[3 * l + 2] - s.pos[2];
          const float r = rsqrtf(dx * dx + dy * dy + dz * dz);
          dx *= r; dy *= r; dz *= r;
          const float c = dx * lx + dy * ly + dz * lz;
          const float cc = fminf(1.f, fmaxf(-1.f, c));
          const float angle = acosf(cc);
          const float ex = __expf(-0.5f * (angle * inv_sg) * (angle * inv_sg));
          const float dw = g_sraw[0] * lv[3 * l] + g_sraw[1] * lv[3 * l + 1] + g_sraw[2] * lv[3 * l + 2];
          const float dacos = (c > -1.f && c < 1.f) ? -rsqrtf(1.f - c * c) : -20.f;
          gs += dw * ((ex * kInvSqrt2Pi23 * (angle * angle - s2)) * inv_s4);
          const float dc = dw * -((kInvSqrt2Pi23 * angle * ex) * inv_s3) * dacos;
          gx += dc * dx; gy += dc * dy; gz += dc * dz;
        }
        g_sigma += gs;
        const float dotl = gx * lx + gy * ly + gz * lz;
        g_ref[0] = (gx - lx * dotl) * rn; g_ref[1] = (gy - ly * dotl) * rn; g_ref[2] = (gz - lz * dotl) * rn;
      }
      // ref = view - 2 (view.n) n
      const float ndg = s.n[0] * g_ref[0] + s.n[1] * g_ref[1] + s.n[2] * g_ref[2];
      float g_view[3], g_n[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        g_view[k] = g_ref[k] - 2.f * s.n[k] * ndg;
        g_n[k] = u_n[k][v] - 2.f * (s.view[k] * ndg + s.vdn * g_ref[k]);
      }
      // view = normalize(pos - cam)
      const float vdg = s.view[0] * g_view[0] + s.view[1] * g_view[1] + s.view[2] * g_view[2];
      const float idn = 1.f / fmaxf(s.dn, kNormEps);
      // spec_nml = normalize(dnml + nmlbase)
      const float ndn = s.n[0] * g_n[0] + s.n[1] * g_n[1] + s.n[2] * g_n[2];
      const float imn = 1.f / fmaxf(s.mn, kNormEps);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float gp = u_pos[k][v] + (g_view[k] - s.view[k] * vdg) * idn;
        gg[k][v] = gp;
        gpb[k][v] = gp;
        const float gm = (g_n[k] - s.n[k] * ndn) * imn;
        gfc[1 + k][v] = u_dn[k][v] + gm;
        gnb[k][v] = u_nb[k][v] + gm;
      }
      // qvec = normalize(f)
      const float qdg = s.q[0] * u_q[0][v] + s.q[1] * u_q[1][v] + s.q[2] * u_q[2][v] + s.q[3] * u_q[3][v];
      const float iq = 1.f / fmaxf(s.qn, kNormEps);
#pragma unroll
      for (int k = 0; k < 4; ++k) gg[3 + k][v] = (u_q[k][v] - s.q[k] * qdg) * iq;
      // scale = clamp(softplus(x), min, max); preclip = softplus(x)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float pre = s.sp[k];
        const float gpre = u_sp[k][v] + ((pre >= in.primscale_min && pre <= in.primscale_max) ? u_sc[k][v] : 0.f);
        const float x = g1[7 + k];
        gg[7 + k][v] = gpre * (x > 20.f ? 1.f : sigmoidf(x));
      }
      gg[10][v] = u_op[0][v] * s.opac * (1.f - s.opac);
      gg[11][v] = (s.e01 >= 0.01f) ? g_sigma * s.e01 : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) stv<V>(GF + (size_t)(nd + j) * N, gg[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) stv<V>(gin.f_vcond + ((size_t)b * 4 + j) * N + i0, gfc[j]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      stv<V>(gin.postex + ((size_t)b * 3 + j) * N + i0, gpb[j]);
      stv<V>(gin.tn + ((size_t)b * 3 + j) * N + i0, gnb[j]);
    }
    st_aos<V, 3>(gin.albedo_per_view, g0, galb);
#pragma unroll
    for (int c = 0; c < 3; ++c) { s_g[c][threadIdx.x] = gD[c][0]; s_g[3 + c][threadIdx.x] = gDr[c][0]; }
  } else {
#pragma unroll
    for (int c = 0; c < 6; ++c) s_g[c][threadIdx.x] = 0.f;
  }
  __syncthreads();

  // phase 2: write-only gradients of the SH channels = upstream x light coefficient
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j0 = blk0 + 4 * lane;  // first of this lane's 4 Gaussians
  if (j0 >= N) return;
  float gD4[3][4], gR4[3][4];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float4 a = *reinterpret_cast<const float4*>(&s_g[c][4 * lane]);
    gD4[c][0] = a.x; gD4[c][1] = a.y; gD4[c][2] = a.z; gD4[c][3] = a.w;
    if (RAND) {
      const float4 r = *reinterpret_cast<const float4*>(&s_g[3 + c][4 * lane]);
      gR4[c][0] = r.x; gR4[c][1] = r.y; gR4[c][2] = r.z; gR4[c][3] = r.w;
    }
  }
  float* GP = gin.f_vnocond + view0 + j0;
  const int nvalid = min(4, N - j0);
  for (int ch = wave; ch < nd; ch += 4) {
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (ch < 3 * ncol) {
      const int c = ch / ncol, k = ch - c * ncol;
      const float l = Lsh[c * ncoef + k];
      const float lr = RAND ? Lr[c * ncoef + k] : 0.f;
      // c is wave-uniform but not a compile-time constant: select the row without dynamic indexing
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        if (cc == c) {
#pragma unroll
          for (int v = 0; v < 4; ++v) x[v] = gD4[cc][v] * l + (RAND ? gR4[cc][v] * lr : 0.f);
        }
    } else {
      const int k = ch - 3 * ncol;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float l = Lsh[c * ncoef + ncol + k];
        const float lr = RAND ? Lr[c * ncoef + ncol + k] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) x[v] += gD4[c][v] * l + (RAND ? gR4[c][v] * lr : 0.f);
      }
    }
    float* dst = GP + (size_t)ch * N;
    if (VEC4) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 v4 = {x[0], x[1], x[2], x[3]};
      __builtin_nontemporal_store(v4, reinterpret_cast<f4*>(dst));  // written once, consumed by the decoder backward much later
    } else {
      for (int v = 0; v < nvalid; ++v) dst[v] = x[v];
    }
  }
}

// [B,3,h,w] planar -> [B,h,w,16] footprint records (see bilinear_border), one lane per record
// out[i] = sum over the B views of in[b][i]  (the per-view albedo gradients -> the shared parameter's)
__global__ __launch_bounds__(256) void sum_views_kernel(int B, size_t M, const float* __restrict__ in, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += in[(size_t)b * M + i];
  out[i] = acc;
}

__global__ __launch_bounds__(256) void envmap_pack_kernel(int h, int w, const float* __restrict__ src, float4* __restrict__ dst) {
  const int b = blockIdx.y, hw = h * w;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  const int y = i / w, x = i - y * w;
  const float* s = src + (size_t)b * 3 * hw;
  const bool bx = x + 1 < w, by = y + 1 < h;
  auto tap = [&](int k, bool ok) {
    return ok ? make_float4(s[k], s[hw + k], s[2 * (size_t)hw + k], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  float4* d = dst + ((size_t)b * hw + i) * 4;
  d[0] = tap(i, true); d[1] = tap(i + 1, bx); d[2] = tap(i + w, by); d[3] = tap(i + w + 1, bx && by);
}

int check_in(const gol_shade_in* in) {
  GOL_REQUIRE(in != nullptr, "null gol_shade_in");
  GOL_REQUIRE(in->B >= 0 && in->N >= 0, "negative size");
  GOL_REQUIRE(in->B <= 65535, "B > 65535");
  GOL_REQUIRE(in->n_color_coef > 0 && in->n_mono_coef >= 0, "bad SH sizes");
  if (in->B == 0 || in->N == 0) return GOL_OK;
  GOL_REQUIRE(in->f_vnocond && in->f_vcond && in->postex && in->tn && in->albedo && in->light_sh && in->campos,
              "null input");
  GOL_REQUIRE(in->n_mips >= 0 && in->n_mips <= GOL_MAX_MIPS, "n_mips out of range");
  if (in->n_mips > 0) {
    GOL_REQUIRE(in->lightrot != nullptr, "env map needs lightrot");
    GOL_REQUIRE(in->mips_shared == 0 || in->mips_shared == 1, "mips_shared must be 0 or 1");
    for (int i = 0; i < in->n_mips; ++i)
      GOL_REQUIRE(in->mips[i] && in->mip_h[i] > 0 && in->mip_w[i] > 0, "bad mip level");
    for (int i = 0; i < in->n_mips; ++i)
      GOL_REQUIRE((in->mips_packed[i] != nullptr) == (in->mips_packed[0] != nullptr), "mips_packed: all levels or none");
  } else {
    GOL_REQUIRE(in->L >= 0 && in->n_lights != nullptr, "point lights need n_lights");
    GOL_REQUIRE(in->L == 0 || (in->light_intensity && in->light_pos), "null light arrays");
  }
  return GOL_OK;
}

}  // namespace

extern "C" int gol_envmap_pack(int B, int h, int w, const float* src, float* dst, void* stream) {
  GOL_REQUIRE(B >= 0 && h > 0 && w > 0, "bad size");
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(src && dst, "null pointer");
  GOL_REQUIRE(B <= 65535, "B > 65535");
  envmap_pack_kernel<<<dim3(gol_cdiv((long long)h * w, 256), B), 256, 0, (hipStream_t)stream>>>(
      h, w, src, reinterpret_cast<float4*>(dst));
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

// VEC4: the SH planes of a view are 16-byte aligned (N % 4 == 0) -> 16-byte plane loads; otherwise scalar ones
#define GOL_SHADE_FWD_V(V4, P, ...)                                                              \
  do {                                                                                           \
    dim3 grid(gol_cdiv(in->N, 256), in->B);                                                      \
    if (env && rnd) shade_fwd_kernel<true, true, P, V4><<<grid, 256, 0, s>>>(__VA_ARGS__);       \
    else if (env) shade_fwd_kernel<true, false, P, V4><<<grid, 256, 0, s>>>(__VA_ARGS__);        \
    else if (rnd) shade_fwd_kernel<false, true, P, V4><<<grid, 256, 0, s>>>(__VA_ARGS__);        \
    else shade_fwd_kernel<false, false, P, V4><<<grid, 256, 0, s>>>(__VA_ARGS__);                \
  } while (0)

#define GOL_SHADE_FWD_DISPATCH(P, ...)                                                           \
  do {                                                                                           \
    const bool env = in->n_mips > 0, rnd = in->light_sh_rand != nullptr;                         \
    if (in->N % 4 == 0) GOL_SHADE_FWD_V(true, P, __VA_ARGS__);                                   \
    else GOL_SHADE_FWD_V(false, P, __VA_ARGS__);                                                 \
  } while (0)

#define GOL_SHADE_BWD_CASE(E, R, P)                                                              \
  do {                                                                                           \
    if (in->N % 4 == 0) shade_bwd_kernel<E, R, true, P><<<grid, 256, 0, s>>>(*in, *saved, *g, *gin, pj, grad_records, with_depth); \
    else shade_bwd_kernel<E, R, false, P><<<grid, 256, 0, s>>>(*in, *saved, *g, *gin, pj, grad_records, with_depth); \
  } while (0)

namespace {

int check_proj(const gol_shade_in* in, const gol_shade_proj* pj) {
  GOL_REQUIRE(pj != nullptr, "null gol_shade_proj");
  GOL_REQUIRE(pj->img_h > 0 && pj->img_w > 0, "empty image");
  if (in->B == 0 || in->N == 0) return GOL_OK;
  GOL_REQUIRE(pj->viewmats && pj->intrins, "null camera");
  GOL_REQUIRE(pj->xys && pj->depths && pj->radii && pj->conics && pj->comp && pj->opac_eff && pj->records,
              "null projection buffer");
  return GOL_OK;
}

int shade_fwd_launch(const gol_shade_in* in, const gol_shade_out* out, const gol_shade_proj* proj, void* stream) {
  int rc = check_in(in);
  if (rc != GOL_OK) return rc;
  if (proj && (rc = check_proj(in, proj)) != GOL_OK) return rc;
  if (in->B == 0 || in->N == 0) return GOL_OK;
  GOL_REQUIRE(out != nullptr, "null gol_shade_out");
  GOL_REQUIRE(out->color && out->opacity && out->primpos && out->primqvec && out->primscale &&
                  out->primscale_preclip && out->sigma && out->spec_vis && out->spec_nml && out->spec_dnml &&
                  out->diff_color && out->spec_color && out->primnmlbase && out->diff_sum,
              "null output");
  GOL_REQUIRE((in->light_sh_rand == nullptr) || out->color_rand, "color_rand output missing");
  hipStream_t s = (hipStream_t)stream;
  gol_shade_proj pj;
  memset(&pj, 0, sizeof(pj));
  if (proj) {
    pj = *proj;
    GOL_SHADE_FWD_DISPATCH(true, *in, *out, pj);
  } else {
    GOL_SHADE_FWD_DISPATCH(false, *in, *out, pj);
  }
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

int shade_bwd_launch(const gol_shade_in* in, const gol_shade_out* saved, const gol_shade_out_grad* g,
                     const gol_shade_proj* proj, const float* grad_records, int with_depth, const gol_shade_in_grad* gin,
                     void* stream) {
  int rc = check_in(in);
  if (rc != GOL_OK) return rc;
  if (proj && (rc = check_proj(in, proj)) != GOL_OK) return rc;
  if (in->B == 0 || in->N == 0) return GOL_OK;
  GOL_REQUIRE(saved && g && gin, "null struct");
  GOL_REQUIRE(saved->diff_sum != nullptr, "saved diff_sum missing");
  GOL_REQUIRE((in->light_sh_rand == nullptr) || saved->color_rand, "saved color_rand missing");
  GOL_REQUIRE(gin->f_vnocond && gin->f_vcond && gin->postex && gin->tn && gin->albedo_per_view, "null gradient output");
  GOL_REQUIRE(!proj || grad_records, "null gradient records");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(gol_cdiv(in->N, 256), in->B);
  const bool env = in->n_mips > 0, rnd = in->light_sh_rand != nullptr;
  gol_shade_proj pj;
  memset(&pj, 0, sizeof(pj));
  if (proj) {
    pj = *proj;
    if (env && rnd) GOL_SHADE_BWD_CASE(true, true, true);
    else if (env) GOL_SHADE_BWD_CASE(true, false, true);
    else if (rnd) GOL_SHADE_BWD_CASE(false, true, true);
    else GOL_SHADE_BWD_CASE(false, false, true);
  } else {
    if (env && rnd) GOL_SHADE_BWD_CASE(true, true, false);
    else if (env) GOL_SHADE_BWD_CASE(true, false, false);
    else if (rnd) GOL_SHADE_BWD_CASE(false, true, false);
    else GOL_SHADE_BWD_CASE(false, false, false);
  }
  GOL_CHECK_LAUNCH();
  if (gin->albedo) {  // gradient of the albedo shared by the views = sum over B of the per-view gradients
    const size_t M = (size_t)in->N * 3;
    sum_views_kernel<<<gol_cdiv((long long)M, 256), 256, 0, s>>>(in->B, M, gin->albedo_per_view, gin->albedo);
    GOL_CHECK_LAUNCH();
  }
  return GOL_OK;
}

}  // namespace

extern "C" int gol_shade_fwd(const gol_shade_in* in, const gol_shade_out* out, void* stream) {
  return shade_fwd_launch(in, out, nullptr, stream);
}

extern "C" int gol_shade_project_fwd(const gol_shade_in* in, const gol_shade_out* out, const gol_shade_proj* proj,
                                     void* stream) {
  GOL_REQUIRE(proj != nullptr, "null gol_shade_proj");
  return shade_fwd_launch(in, out, proj, stream);
}

extern "C" int gol_shade_bwd(const gol_shade_in* in, const gol_shade_out* saved, const gol_shade_out_grad* g,
                             const gol_shade_in_grad* gin, void* stream) {
  return shade_bwd_launch(in, saved, g, nullptr, nullptr, 0, gin, stream);
}

extern "C" int gol_shade_project_bwd(const gol_shade_in* in, const gol_shade_out* saved, const gol_shade_out_grad* g,
                                     const gol_shade_proj* proj, const float* grad_records, int with_depth,
                                     const gol_shade_in_grad* gin, void* stream) {
  GOL_REQUIRE(proj != nullptr, "null gol_shade_proj");
  return shade_bwd_launch(in, saved, g, proj, grad_records, with_depth, gin, stream);
}
