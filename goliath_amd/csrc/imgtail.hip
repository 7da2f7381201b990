// imgtail.hip -- the image tail of AutoEncoder.forward, between the rasterizer and the losses, as ONE pass forward
// and ONE pass backward over the [B,3,H,W] image, gfx950.
//
// Replaces (/root/reference/ca_code):
//   nn/color_cal.py:211-241   CalV5.forward         per-view colour gain / bias (a per-view host loop with a tensor
//                                                   comparison -- a sync -- per view); here a per-view 3x3 matrix + bias
//                                                   read from device memory (grey cameras = three identical rows)
//   models/rgca.py:226-230    rgb + (1 - alpha) * bg training background composite (bg gated per view)
//   nn/dof_cal.py:44-56       LearnableBlur.forward  w0 * img + w1 * gaussian_blur(img, 3) + w2 * gaussian_blur(img, 7),
//                                                   torchvision semantics: reflect padding, separable kernel with
//                                                   sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8 (two pad + depthwise-conv
//                                                   round trips over the image in the reference)
// One 256-thread workgroup per 32x32 tile of one view: the calibrated + composited tile (+3 halo, reflect-indexed) of
// all three channels is staged in LDS once, the two separable filters run out of LDS.  Backward: the blur is linear,
// so its adjoint is the same 7-tap gather with border-folded weights (reflect padding folds the taps that fall outside
// back inside); parameter gradients (blur weights, gain matrix, bias: 15 sums per view) are block-reduced into
// per-workgroup partial sums -- no float atomics.
#include "gol_common.h"

namespace {

constexpr int kT = 32;            // tile edge
constexpr int kR = 3;             // halo = radius of the 7-tap filter
constexpr int kS = kT + 2 * kR;   // staged edge (38)
constexpr int kP = kS + 2;        // padded LDS row (40 floats)

struct TailArgs {
  int B, H, W;
  const float* rgb;       // [B,3,H,W]
  const float* alpha;     // [B,H,W] or null
  const float* bg;        // [B,3,H,W] or null (with alpha)
  const float* bg_scale;  // [B] or null: per-view gate of the background (is_fully_lit_frame)
  const float* M;         // [B,3,3] or null (identity)
  const float* bvec;      // [B,3] or null
  const float* blur_w;    // [B,3] softmaxed weights or null (no blur)
  float k3[2];            // centre, +-1
  float k7[4];            // centre, +-1, +-2, +-3
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
  // torch 'reflect' padding (no edge repeat); safe for any i once n >= 2, clamps for n == 1
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  i = i % period;
  if (i < 0) i += period;
  return i < n ? i : period - i;
}

struct View {
  float m[3][3], b[3], bgs;
};

__device__ __forceinline__ View load_view(const TailArgs& a, int v) {
  View w;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int j = 0; j < 3; ++j) w.m[c][j] = a.M ? a.M[(size_t)v * 9 + c * 3 + j] : (c == j ? 1.f : 0.f);
    w.b[c] = a.bvec ? a.bvec[(size_t)v * 3 + c] : 0.f;
  }
  w.bgs = a.bg ? (a.bg_scale ? a.bg_scale[v] : 1.f) : 0.f;
  return w;
}

// calibrated + composited pixel (all three channels) at image position (y, x) of view v
__device__ __forceinline__ void pixel_x(const TailArgs& a, const View& w, int v, int y, int x, float (&o)[3]) {
  const size_t hw = (size_t)a.H * a.W, p = (size_t)y * a.W + x;
  const float* q = a.rgb + (size_t)v * 3 * hw + p;
  const float r = q[0], g = q[hw], bl = q[2 * hw];
  float add[3] = {0.f, 0.f, 0.f};
  if (a.bg) {
    const float om = (1.f - a.alpha[(size_t)v * hw + p]) * w.bgs;
    const float* bq = a.bg + (size_t)v * 3 * hw + p;
    add[0] = om * bq[0]; add[1] = om * bq[hw]; add[2] = om * bq[2 * hw];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = w.m[c][0] * r + w.m[c][1] * g + w.m[c][2] * bl + w.b[c] + add[c];
}

__global__ __launch_bounds__(256) void imgtail_fwd_kernel(TailArgs a, float* __restrict__ out) {
  __shared__ float s_x[3][kS][kP];
  __shared__ float s_h3[kS][kT + 1];
  __shared__ float s_h7[kS][kT + 1];
  const int v = blockIdx.z, ty0 = blockIdx.y * kT, tx0 = blockIdx.x * kT, tid = threadIdx.x;
  const View w = load_view(a, v);
  const size_t hw = (size_t)a.H * a.W;
  if (!a.blur_w) {  // calibration / composite only: no neighbourhood
    for (int i = tid; i < kT * kT; i += 256) {
      const int y = ty0 + i / kT, x = tx0 + i % kT;
      if (y < a.H && x < a.W) {
        float o[3];
        pixel_x(a, w, v, y, x, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((size_t)v * 3 + c) * hw + (size_t)y * a.W + x] = o[c];
      }
    }
    return;
  }
  for (int i = tid; i < kS * kS; i += 256) {
    const int r = i / kS, c0 = i % kS;
    float o[3];
    pixel_x(a, w, v, reflect_idx(ty0 - kR + r, a.H), reflect_idx(tx0 - kR + c0, a.W), o);
    s_x[0][r][c0] = o[0]; s_x[1][r][c0] = o[1]; s_x[2][r][c0] = o[2];
  }
  const float w0 = a.blur_w[v * 3], w1 = a.blur_w[v * 3 + 1], w2 = a.blur_w[v * 3 + 2];
  for (int c = 0; c < 3; ++c) {
    __syncthreads();  // s_x staged (c == 0) / previous channel's s_h consumed
    for (int i = tid; i < kS * kT; i += 256) {
      const int r = i / kT, x = i % kT;
      const float* row = &s_x[c][r][x];  // row[kR] is the centre
      s_h3[r][x] = a.k3[0] * row[3] + a.k3[1] * (row[2] + row[4]);
      s_h7[r][x] = a.k7[0] * row[3] + a.k7[1] * (row[2] + row[4]) + a.k7[2] * (row[1] + row[5]) + a.k7[3] * (row[0] + row[6]);
    }
    __syncthreads();
    for (int i = tid; i < kT * kT; i += 256) {
      const int r = i / kT, x = i % kT, y = ty0 + r, gx = tx0 + x;
      if (y >= a.H || gx >= a.W) continue;
      const float b3 = a.k3[0] * s_h3[r + 3][x] + a.k3[1] * (s_h3[r + 2][x] + s_h3[r + 4][x]);
      const float b7 = a.k7[0] * s_h7[r + 3][x] + a.k7[1] * (s_h7[r + 2][x] + s_h7[r + 4][x]) +
                       a.k7[2] * (s_h7[r + 1][x] + s_h7[r + 5][x]) + a.k7[3] * (s_h7[r][x] + s_h7[r + 6][x]);
      out[((size_t)v * 3 + c) * hw + (size_t)y * a.W + gx] = w0 * s_x[c][r + 3][x + 3] + w1 * b3 + w2 * b7;
    }
  }
}

// weight of input position q in the ADJOINT of the reflect-padded filter at output position p (1-D, n samples):
// the plain tap k[|p - q|] plus the taps that the forward pass folded back inside at the two borders
template <int R>
__device__ __forceinline__ float adj_weight(const float* k, int p, int q, int n) {
  const int d = p > q ? p - q : q - p;
  float wgt = d <= R ? k[d] : 0.f;
  const int dl = p + q;                 // forward read x[reflect(-p)] = x[p] from output q with tap -(p + q)
  if (p >= 1 && dl <= R) wgt += k[dl];
  const int dr = 2 * (n - 1) - p - q;   // ... and x[reflect(2(n-1) - p)] at the far border
  if (p <= n - 2 && dr >= 0 && dr <= R) wgt += k[dr];
  return wgt;
}

__global__ __launch_bounds__(256) void imgtail_bwd_kernel(TailArgs a, const float* __restrict__ g_out,
                                                          float* __restrict__ g_rgb, float* __restrict__ partials) {
  __shared__ float s_g[3][kS][kP];
  __shared__ float s_h3[kS][kT + 1];
  __shared__ float s_h7[kS][kT + 1];
  __shared__ float s_red[4][16];
  const int v = blockIdx.z, ty0 = blockIdx.y * kT, tx0 = blockIdx.x * kT, tid = threadIdx.x;
  const View w = load_view(a, v);
  const size_t hw = (size_t)a.H * a.W;
  const bool blur = a.blur_w != nullptr;
  const float w0 = blur ? a.blur_w[v * 3] : 1.f, w1 = blur ? a.blur_w[v * 3 + 1] : 0.f, w2 = blur ? a.blur_w[v * 3 + 2] : 0.f;
  if (blur) {
    for (int i = tid; i < kS * kS; i += 256) {
      const int r = i / kS, c0 = i % kS, y = ty0 - kR + r, x = tx0 - kR + c0;
      const bool in = y >= 0 && y < a.H && x >= 0 && x < a.W;
#pragma unroll
      for (int c = 0; c < 3; ++c) s_g[c][r][c0] = in ? g_out[((size_t)v * 3 + c) * hw + (size_t)y * a.W + x] : 0.f;
    }
  }
  // per-thread pixels: rows tid / 32 + 8 i, column tid % 32
  const int px = tid & 31, gx = tx0 + px;
  float gxv[4][3];        // d loss / d x per pixel and channel
  float sums[16];         // blur weights 0..2 | bias 3..5 | M 6..14
#pragma unroll
  for (int k = 0; k < 16; ++k) sums[k] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) gxv[i][c] = 0.f;
  float xs[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = ty0 + (tid >> 5) + 8 * i;
    if (y < a.H && gx < a.W) pixel_x(a, w, v, y, gx, xs[i]);
    else xs[i][0] = xs[i][1] = xs[i][2] = 0.f;
  }
  for (int c = 0; c < 3; ++c) {
    if (blur) {
      __syncthreads();
      // adjoint along x for every staged row (rows outside the image hold zeros)
      for (int i = tid; i < kS * kT; i += 256) {
        const int r = i / kT, x = i % kT, p = tx0 + x;
        float h3 = 0.f, h7 = 0.f;
        if (p < a.W) {
#pragma unroll
          for (int d = -kR; d <= kR; ++d) {
            const int q = p + d;
            if (q < 0 || q >= a.W) continue;
            const float gv = s_g[c][r][x + kR + d];
            h7 += adj_weight<3>(a.k7, p, q, a.W) * gv;
            if (d >= -1 && d <= 1) h3 += adj_weight<1>(a.k3, p, q, a.W) * gv;
          }
        }
        s_h3[r][x] = h3; s_h7[r][x] = h7;
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 5) + 8 * i, y = ty0 + r;
      if (y >= a.H || gx >= a.W) continue;
      float g0, a3 = 0.f, a7 = 0.f;
      if (blur) {
        g0 = s_g[c][r + kR][px + kR];
#pragma unroll
        for (int d = -kR; d <= kR; ++d) {
          const int q = y + d;
          if (q < 0 || q >= a.H) continue;
          a7 += adj_weight<3>(a.k7, y, q, a.H) * s_h7[r + kR + d][px];
          if (d >= -1 && d <= 1) a3 += adj_weight<1>(a.k3, y, q, a.H) * s_h3[r + kR + d][px];
        }
      } else {
        g0 = g_out[((size_t)v * 3 + c) * hw + (size_t)y * a.W + gx];
      }
      const float x_c = xs[i][c];
      sums[0] += g0 * x_c; sums[1] += a3 * x_c; sums[2] += a7 * x_c;   // <g, F_k x> = <F_k^T g, x>
      gxv[i][c] = w0 * g0 + w1 * a3 + w2 * a7;
    }
  }
  // chain through the calibration: x_c = sum_j M[c][j] rgb_j + b_c (+ background, no gradient)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = ty0 + (tid >> 5) + 8 * i;
    if (y >= a.H || gx >= a.W) continue;
    const size_t p = (size_t)y * a.W + gx;
    const float* q = a.rgb + (size_t)v * 3 * hw + p;
    const float rgbv[3] = {q[0], q[hw], q[2 * hw]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      sums[3 + c] += gxv[i][c];
#pragma unroll
      for (int j = 0; j < 3; ++j) sums[6 + c * 3 + j] += gxv[i][c] * rgbv[j];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
      g_rgb[((size_t)v * 3 + j) * hw + p] = w.m[0][j] * gxv[i][0] + w.m[1][j] * gxv[i][1] + w.m[2][j] * gxv[i][2];
  }
  // 16 block sums: four 4-way wave reductions (results in lanes 15/31/47/63), then across the 4 waves through LDS
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const float r = gol_wave_sum4(sums[4 * k4], sums[4 * k4 + 1], sums[4 * k4 + 2], sums[4 * k4 + 3]);
    if ((lane & 15) == 15) s_red[wave][4 * k4 + (lane >> 4)] = r;
  }
  __syncthreads();
  if (tid < 16) {
    const size_t blk = ((size_t)v * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partials[blk * 16 + tid] = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
  }
}

void fill_kernels(TailArgs& a) {
  // torchvision _get_gaussian_kernel1d: linspace(-(k-1)/2, (k-1)/2, k), exp(-0.5 (x / sigma)^2), normalised, in fp32
  const float s3 = 0.3f * ((3 - 1) * 0.5f - 1.f) + 0.8f, s7 = 0.3f * ((7 - 1) * 0.5f - 1.f) + 0.8f;
  float p3[2], p7[4], n3 = 0.f, n7 = 0.f;
  for (int i = 0; i < 2; ++i) { const float x = (float)i / s3; p3[i] = expf(-0.5f * x * x); n3 += (i ? 2.f : 1.f) * p3[i]; }
  for (int i = 0; i < 4; ++i) { const float x = (float)i / s7; p7[i] = expf(-0.5f * x * x); n7 += (i ? 2.f : 1.f) * p7[i]; }
  for (int i = 0; i < 2; ++i) a.k3[i] = p3[i] / n3;
  for (int i = 0; i < 4; ++i) a.k7[i] = p7[i] / n7;
}

int check(const char* fn, int B, int H, int W, const float* rgb, const float* alpha, const float* bg, const float* blur_w) {
  if (B < 0 || H <= 0 || W <= 0) { gol_set_error("%s: bad size", fn); return GOL_ERR_INVALID_ARG; }
  if (B > 65535) { gol_set_error("%s: B > 65535", fn); return GOL_ERR_INVALID_ARG; }
  if (B && !rgb) { gol_set_error("%s: null image", fn); return GOL_ERR_INVALID_ARG; }
  if ((alpha == nullptr) != (bg == nullptr)) { gol_set_error("%s: alpha and bg go together", fn); return GOL_ERR_INVALID_ARG; }
  if (blur_w && (H < 4 || W < 4)) {  // torch reflect padding needs pad < size
    gol_set_error("%s: the 7x7 blur needs an image of at least 4x4 pixels", fn);
    return GOL_ERR_INVALID_ARG;
  }
  return GOL_OK;
}

}  // namespace

extern "C" int64_t gol_imgtail_partial_floats(int B, int H, int W) {
  return (int64_t)B * gol_cdiv(H, kT) * gol_cdiv(W, kT) * 16;
}

extern "C" int gol_imgtail_fwd(int B, int H, int W, const float* rgb, const float* alpha, const float* bg,
                               const float* bg_scale, const float* cal_M, const float* cal_b, const float* blur_w,
                               float* out, void* stream) {
  const int rc = check("gol_imgtail_fwd", B, H, W, rgb, alpha, bg, blur_w);
  if (rc != GOL_OK) return rc;
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(out != nullptr, "null output");
  TailArgs a{B, H, W, rgb, alpha, bg, bg_scale, cal_M, cal_b, blur_w, {0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  fill_kernels(a);
  imgtail_fwd_kernel<<<dim3(gol_cdiv(W, kT), gol_cdiv(H, kT), B), 256, 0, (hipStream_t)stream>>>(a, out);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_imgtail_bwd(int B, int H, int W, const float* rgb, const float* alpha, const float* bg,
                               const float* bg_scale, const float* cal_M, const float* cal_b, const float* blur_w,
                               const float* g_out, float* g_rgb, float* partials, void* stream) {
  const int rc = check("gol_imgtail_bwd", B, H, W, rgb, alpha, bg, blur_w);
  if (rc != GOL_OK) return rc;
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(g_out && g_rgb && partials, "null pointer");
  TailArgs a{B, H, W, rgb, alpha, bg, bg_scale, cal_M, cal_b, blur_w, {0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  fill_kernels(a);
  imgtail_bwd_kernel<<<dim3(gol_cdiv(W, kT), gol_cdiv(H, kT), B), 256, 0, (hipStream_t)stream>>>(a, g_out, g_rgb, partials);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
