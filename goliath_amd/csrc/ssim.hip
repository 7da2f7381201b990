// ssim.hip -- fused SSIM image loss (forward + backward), gfx950.  SURVEY 8f #3.
//
// Replaces the ATen chain of `rgb_ssim` (/root/reference/ca_code/loss/__init__.py:478-494) ->
// `ssim` / `_ssim` (/root/reference/ca_code/utils/ssim.py:25-65): five depthwise 11x11 Gaussian convolutions
// (sigma 1.5, zero padding 5) of img1, img2, img1^2, img2^2, img1*img2, the SSIM map, a masked mean -- and the
// autograd graph behind them (five more convolutions backward).  Here:
//   forward   a workgroup walks 32x32 tiles of one image plane (the next tile's halo is prefetched into registers
//             while the current one is convolved): the 42x42 halo of both images is staged in LDS,
//             the window is applied separably (11 horizontal taps, then 11 vertical taps) to the five moment maps,
//             the SSIM value is formed, masked and block-reduced (per-workgroup partial sums, caller adds them:
//             deterministic), and the three maps the backward needs are written:
//                 M0 = mask * df/dmu2,  M1 = mask * df/dE[y^2],  M2 = mask * df/dE[xy]       (f = SSIM value)
//   backward  d loss/d img2[q] = g * ( conv(M0)[q] + 2 img2[q] conv(M1)[q] + img1[q] conv(M2)[q] )  (symmetric window):
//             three separable convolutions of the saved maps, same tiling.
// HBM per pixel and channel: fwd 8 B in (+ mask) + 12 B out, bwd 12 B + 8 B in, 4 B out; everything else is LDS.
#include "gol_common.h"

namespace {

constexpr int kWin = 11, kHalo = 5;
constexpr int kTile = 32, kIn = kTile + 2 * kHalo;  // 42
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;
constexpr int kPer = (kIn * kIn + 255) / 256;  // halo values per thread and image (7)
constexpr int kStrips = 6;                    // workgroups per tile row (each walks tiles_x / kStrips tiles)

// gaussian(11, 1.5) / sum  (ssim.py:15-17), float32 like the reference's th.Tensor
struct Window {
  float w[kWin];
};
Window make_window() {
  Window g;
  float e[kWin], s = 0.f;
  for (int i = 0; i < kWin; ++i) {
    const float d = (float)(i - kWin / 2);
    e[i] = (float)exp(-(double)(d * d) / (2.0 * 1.5 * 1.5));
    s += e[i];
  }
  for (int i = 0; i < kWin; ++i) g.w[i] = e[i] / s;
  return g;
}

// Workgroup -> (tile row, contiguous run of tiles).  blockIdx.x is linear over (tile row, strip); workgroup b runs on XCD
// b % 8, so die x is given a contiguous band of tile rows and walks it strip by strip, row by row: the 10 halo rows shared
// by vertically adjacent tiles and the 10 halo columns shared by consecutive tiles of a strip are then L2 hits instead
// of fabric reads (PMC: the forward fetched 3.4x its algorithmic bytes with a scattered order).
struct Strip { int ty, tx_begin, tx_end; bool ok; };
__host__ __device__ inline int ssim_strips(int W) { const int t = (W + kTile - 1) / kTile; return t < kStrips ? t : kStrips; }
__device__ __forceinline__ Strip strip_of_block(int H, int W) {
  const int tiles_y = (H + kTile - 1) / kTile, tiles_x = (W + kTile - 1) / kTile, strips = ssim_strips(W);
  const int rows_per_xcd = (tiles_y + 7) / 8;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int per = (tiles_x + strips - 1) / strips;
  Strip s;
  s.ty = xcd * rows_per_xcd + j / strips;
  const int st = j % strips;
  s.tx_begin = st * per;
  s.tx_end = min(tiles_x, s.tx_begin + per);
  s.ok = (j / strips) < rows_per_xcd && s.ty < tiles_y;
  return s;
}

__device__ __forceinline__ float block_sum(float v, float* s_part) {
  v = gol_wave_sum_to_lane63(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) s_part[wave] = v;
  __syncthreads();
  return s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(int C, int H, int W, int mask_c, const Window win,
                                                       const float* __restrict__ img1, const float* __restrict__ img2,
                                                       const float* __restrict__ mask, float* __restrict__ partial,
                                                       float* __restrict__ dmap, size_t plane_total) {
  __shared__ float s_x[kIn][kIn + 1], s_y[kIn][kIn + 1];
  __shared__ float s_h[5][kIn][kTile];
  __shared__ float s_part[4];
  const int tid = threadIdx.x;
  const Strip sp = strip_of_block(H, W);
  if (!sp.ok) {  // padding workgroup of the XCD-banded grid: its slot of the partial sums still has to be defined
    if (tid == 0) partial[(size_t)blockIdx.z * gridDim.x + blockIdx.x] = 0.f;
    return;
  }
  const int bc = blockIdx.z, ty0 = sp.ty * kTile;
  const size_t HW = (size_t)H * W;
  const float* p1 = img1 + (size_t)bc * HW;
  const float* p2 = img2 + (size_t)bc * HW;
  const int c = tid & 31, r0 = (tid >> 5) * 4;
  float acc = 0.f;
  // A workgroup walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... of its tile row; the halo of the NEXT tile is
  // fetched into registers while the current one is convolved (one exposed memory latency per workgroup, not per tile)
  float ra[kPer], rb[kPer];
  auto issue = [&](int tx0) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int idx = min(tid + 256 * u, kIn * kIn - 1);
      const int r = idx / kIn, cc = idx - r * kIn;
      const int gy = ty0 - kHalo + r, gx = tx0 - kHalo + cc;
      const size_t o = (size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
      ra[u] = p1[o]; rb[u] = p2[o];
    }
  };
  auto stage = [&](int tx0) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int idx = tid + 256 * u;
      if (idx < kIn * kIn) {
        const int r = idx / kIn, cc = idx - r * kIn;
        const int gy = ty0 - kHalo + r, gx = tx0 - kHalo + cc;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        s_x[r][cc] = in ? ra[u] : 0.f;
        s_y[r][cc] = in ? rb[u] : 0.f;
      }
    }
  };
  if (sp.tx_begin < sp.tx_end) issue(sp.tx_begin * kTile);
  for (int tx = sp.tx_begin; tx < sp.tx_end; ++tx) {
    const int tx0 = tx * kTile;
    __syncthreads();  // the previous tile's vertical pass is done with the LDS buffers
    stage(tx0);
    __syncthreads();
    if (tx + 1 < sp.tx_end) issue((tx + 1) * kTile);
    // horizontal taps: one item = 4 adjacent outputs of a row, sharing a 14-value register window (3x fewer LDS reads)
    for (int item = tid; item < kIn * (kTile / 4); item += 256) {
      const int r = item / (kTile / 4), c0 = (item - r * (kTile / 4)) * 4;
      float xv[kWin + 3], yv[kWin + 3];
  #pragma unroll
      for (int k = 0; k < kWin + 3; ++k) { xv[k] = s_x[r][c0 + k]; yv[k] = s_y[r][c0 + k]; }
  #pragma unroll
      for (int o = 0; o < 4; ++o) {
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
  #pragma unroll
        for (int k = 0; k < kWin; ++k) {
          const float x = xv[o + k], y = yv[o + k], wk = win.w[k];
          m1 += wk * x; m2 += wk * y; e11 += wk * (x * x); e22 += wk * (y * y); e12 += wk * (x * y);
        }
        s_h[0][r][c0 + o] = m1; s_h[1][r][c0 + o] = m2; s_h[2][r][c0 + o] = e11; s_h[3][r][c0 + o] = e22;
        s_h[4][r][c0 + o] = e12;
      }
    }
    __syncthreads();
    // vertical taps: a thread owns 4 consecutive rows of one column (14-row register window per moment map)
    float qq[4][5];
  #pragma unroll
    for (int i = 0; i < 4; ++i)
  #pragma unroll
      for (int v = 0; v < 5; ++v) qq[i][v] = 0.f;
  #pragma unroll
    for (int v = 0; v < 5; ++v) {
      float col[kWin + 3];
  #pragma unroll
      for (int k = 0; k < kWin + 3; ++k) col[k] = s_h[v][r0 + k][c];
  #pragma unroll
      for (int i = 0; i < 4; ++i)
  #pragma unroll
        for (int k = 0; k < kWin; ++k) qq[i][v] += win.w[k] * col[i + k];
    }
  #pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + i;
      const float* q = qq[i];
      const int gy = ty0 + r, gx = tx0 + c;
      if (gy < H && gx < W) {
        const float m1 = q[0], m2 = q[1];
        const float s11 = q[2] - m1 * m1, s22 = q[3] - m2 * m2, s12 = q[4] - m1 * m2;
        const float a1 = 2.f * m1 * m2 + kC1, a2 = 2.f * s12 + kC2;
        const float b1 = m1 * m1 + m2 * m2 + kC1, b2 = s11 + s22 + kC2;
        const float ib1 = __builtin_amdgcn_rcpf(b1), ib2 = __builtin_amdgcn_rcpf(b2), ib = ib1 * ib2;  // v_rcp_f32, 1 ulp
        const float f = a1 * a2 * ib;
        const size_t o = (size_t)gy * W + gx;
        float mk = 1.f;
        if (mask) mk = mask[((size_t)(mask_c == 1 ? bc / C : bc)) * HW + o];
        acc += f * mk;
        if (dmap) {
          // derivatives with the raw moments (mu2, E[y^2], E[xy]) as independent variables
          const float d_e12 = 2.f * a1 * ib;
          const float d_e22 = -f * ib2;
          const float d_m2 = 2.f * m1 * a2 * ib - m1 * d_e12 - 2.f * m2 * f * ib1 - 2.f * m2 * d_e22;
          float* d = dmap + (size_t)bc * HW + o;
          d[0] = mk * d_m2; d[plane_total] = mk * d_e22; d[2 * plane_total] = mk * d_e12;
        }
      }
    }
  }
  const float tsum = block_sum(acc, s_part);
  if (tid == 0) partial[(size_t)bc * gridDim.x + blockIdx.x] = tsum;
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(int H, int W, const Window win, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, const float* __restrict__ dmap,
                                                       const float* __restrict__ g_scale, float* __restrict__ g_img2,
                                                       size_t plane_total) {
  __shared__ float s_m[3][kIn][kIn + 1];
  __shared__ float s_h[3][kIn][kTile];
  const int tid = threadIdx.x;
  const Strip sp = strip_of_block(H, W);
  if (!sp.ok) return;
  const int bc = blockIdx.z, ty0 = sp.ty * kTile;
  const size_t HW = (size_t)H * W;
  const float* d = dmap + (size_t)bc * HW;
  const int c = tid & 31, r0 = (tid >> 5) * 4;
  const float gs = g_scale[0];
  float rm[3][kPer];  // next tile's halo of the three maps, in flight during the current tile's convolutions
  auto issue = [&](int tx0) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int idx = min(tid + 256 * u, kIn * kIn - 1);
      const int r = idx / kIn, cc = idx - r * kIn;
      const int gy = ty0 - kHalo + r, gx = tx0 - kHalo + cc;
      const size_t o = (size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
      rm[0][u] = d[o]; rm[1][u] = d[plane_total + o]; rm[2][u] = d[2 * plane_total + o];
    }
  };
  auto stage = [&](int tx0) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int idx = tid + 256 * u;
      if (idx < kIn * kIn) {
        const int r = idx / kIn, cc = idx - r * kIn;
        const int gy = ty0 - kHalo + r, gx = tx0 - kHalo + cc;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        s_m[0][r][cc] = in ? rm[0][u] : 0.f; s_m[1][r][cc] = in ? rm[1][u] : 0.f; s_m[2][r][cc] = in ? rm[2][u] : 0.f;
      }
    }
  };
  if (sp.tx_begin < sp.tx_end) issue(sp.tx_begin * kTile);
  for (int tx = sp.tx_begin; tx < sp.tx_end; ++tx) {
    const int tx0 = tx * kTile;
    __syncthreads();
    stage(tx0);
    __syncthreads();
    if (tx + 1 < sp.tx_end) issue((tx + 1) * kTile);
    for (int item = tid; item < kIn * (kTile / 4) * 3; item += 256) {  // (map, row, group of 4 outputs)
      const int v = item / (kIn * (kTile / 4)), rem = item - v * (kIn * (kTile / 4));
      const int r = rem / (kTile / 4), c0 = (rem - r * (kTile / 4)) * 4;
      float mv[kWin + 3];
#pragma unroll
      for (int k = 0; k < kWin + 3; ++k) mv[k] = s_m[v][r][c0 + k];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < kWin; ++k) q += win.w[k] * mv[o + k];
        s_h[v][r][c0 + o] = q;
      }
    }
    __syncthreads();
    float qq[4][3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      float col[kWin + 3];
#pragma unroll
      for (int k = 0; k < kWin + 3; ++k) col[k] = s_h[v][r0 + k][c];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < kWin; ++k) q += win.w[k] * col[i + k];
        qq[i][v] = q;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = ty0 + r0 + i, gx = tx0 + c;
      if (gy < H && gx < W) {
        const size_t o = (size_t)bc * HW + (size_t)gy * W + gx;
        g_img2[o] = gs * (qq[i][0] + 2.f * img2[o] * qq[i][1] + img1[o] * qq[i][2]);
      }
    }
  }
}

int check(int B, int C, int H, int W) {
  GOL_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, "bad sizes");
  GOL_REQUIRE((long long)B * C <= 65535 && gol_cdiv(H, kTile) <= 65535, "grid limit");
  return GOL_OK;
}

}  // namespace

static int grid_x(int H, int W) { return 8 * gol_cdiv(gol_cdiv(H, kTile), 8) * ssim_strips(W); }
extern "C" int gol_ssim_blocks(int H, int W) { return grid_x(H, W); }

extern "C" int gol_ssim_fwd(int B, int C, int H, int W, int mask_c, const float* img1, const float* img2,
                            const float* mask, float* partial, float* dmap, void* stream) {
  const int rc = check(B, C, H, W);
  if (rc != GOL_OK) return rc;
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(img1 && img2 && partial, "null pointer");
  GOL_REQUIRE(!mask || mask_c == 1 || mask_c == C, "mask must have 1 or C channels");
  static const Window win = make_window();
  const dim3 grid(grid_x(H, W), 1, B * C);
  ssim_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(C, H, W, mask_c, win, img1, img2, mask, partial, dmap,
                                                        (size_t)B * C * H * W);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_ssim_bwd(int B, int C, int H, int W, const float* img1, const float* img2, const float* dmap,
                            const float* g_scale, float* g_img2, void* stream) {
  const int rc = check(B, C, H, W);
  if (rc != GOL_OK) return rc;
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(img1 && img2 && dmap && g_scale && g_img2, "null pointer");
  static const Window win = make_window();
  const dim3 grid(grid_x(H, W), 1, B * C);
  ssim_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(H, W, win, img1, img2, dmap, g_scale, g_img2,
                                                        (size_t)B * C * H * W);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
