// tail.hip -- the light-contracted last decoder layer, gfx950.
//
// The reference ends both decoders with ConvTranspose2d(16 -> C_out, k4 s2 p1) + an untied bias
// (/root/reference/ca_code/models/rgca.py:427,455; ca_code/nn/layers.py:331-397) and contracts 113 of the
// 125 channels with the light's SH coefficients right away (rgca.py:506-514,528-530).  The host folds the
// light into the weights (goliath_amd/tail.py), so what runs here is a transposed conv with PER-VIEW weights
// and few output channels (CH = 3|6 contracted sums + 12 Gaussian channels, or 4 for the view-conditioned
// decoder) plus the matching contraction of the untied bias:
//     out[b,ch,oy,ox] = sum_{ci,ky,kx} x[b,ci,iy,ix] * weff[b,ci,ch,ky,kx]        (oy = 2 iy - 1 + ky)
//                       + (ch < E ? sum_k lc[k,b,ch] * bias[k,oy,ox] : bias[nd + ch - E, oy, ox])
// Kernels:
//   fwd     pass 1 contracts the bias planes (views are the inner loop, so the 113 planes are read ONCE per batch:
//           452 B per Gaussian per batch instead of per view); pass 2 is the conv as four per-parity GEMMs on
//           v_mfma_f32_16x16x4_f32 with the weights resident in VGPRs;
//   bwd_x   gather form (no atomics): one lane = one input pixel, 16 input channels, 4x4 output window;
//   bwd_w   the weight gradient is a GEMM over pixels: for each of the 16 taps  gW[ci,ch] = sum_pix x[ci,pix]
//           g[ch,pix'] -> v_mfma_f32_16x16x4_f32 (exact fp32), K = 4 input pixels per instruction,
//           16 accumulator tiles (one per tap) per wave, LDS reduction over the 4 waves, then atomics;
//   bwd_b   bias gradient: contracted planes  gbias[k] = sum_{b,e} lc[k,b,e] g[b,e], direct planes sum_b g.
#include "gol_common.h"

namespace {

constexpr int kCi = 16;    // input channels of the last decoder layers (rgca.py:427,455)
constexpr int kMaxB = 8;   // views per pass of the bias contraction (register budget)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// sizes travel in a struct, pointers as __restrict__ kernel parameters: only then can the compiler prove that the
// stores to `out` / the gradients do not clobber the weights and coefficients, and fetch those wave-uniform
// values with scalar loads (s_load) into SGPR operands instead of 64-lane vector loads
struct TailDims {
  int B, h, w, CH, E, nd, wB;
};
struct TailPtr {
  const float *x, *weff, *lc, *bias;
  float* out;
};
#define TAIL_FWD_ARGS const TailDims p, const float* __restrict__ px, const float* __restrict__ pweff, \
                      const float* __restrict__ plc, const float* __restrict__ pbias, float* __restrict__ pout

// pass 1 of the forward: out[b,e,n] = sum_k lc[k,b,e] * bias[k,n] for the E contracted channels.  Views are the
// inner loop, so the nd bias planes are streamed once per batch.
template <int E>
__global__ __launch_bounds__(256) void tail_bias_fwd_kernel(TAIL_FWD_ARGS) {
  const size_t N = (size_t)4 * p.h * p.w;
  const size_t n = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (n >= N) return;
  typedef float f2 __attribute__((ext_vector_type(2)));
  for (int b0 = 0; b0 < p.B; b0 += kMaxB) {
    float sh[kMaxB][E][2];
#pragma unroll
    for (int bb = 0; bb < kMaxB; ++bb)
#pragma unroll
      for (int e = 0; e < E; ++e) sh[bb][e][0] = sh[bb][e][1] = 0.f;
#pragma unroll 2
    for (int kk = 0; kk < p.nd; ++kk) {
      const f2 bv = __builtin_nontemporal_load(reinterpret_cast<const f2*>(pbias + (size_t)kk * N + n));
      const float* L = plc + ((size_t)kk * p.B + b0) * E;  // lc is [nd,B,E]: one contiguous run per plane
#pragma unroll
      for (int bb = 0; bb < kMaxB; ++bb) {
        if (b0 + bb < p.B) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const float l = L[bb * E + e];
            sh[bb][e][0] += l * bv.x;
            sh[bb][e][1] += l * bv.y;
          }
        }
      }
    }
#pragma unroll
    for (int bb = 0; bb < kMaxB; ++bb) {
      if (b0 + bb < p.B) {
#pragma unroll
        for (int e = 0; e < E; ++e)
          *reinterpret_cast<float2*>(pout + ((size_t)(b0 + bb) * p.CH + e) * N + n) = make_float2(sh[bb][e][0], sh[bb][e][1]);
      }
    }
  }
}

// pass 2: the transposed conv on the matrix cores (exact fp32: v_mfma_f32_16x16x4_f32).
// An output 2x2 quad {2m-1, 2m} x {2k-1, 2k} reads exactly the 2x2 input block {m-1, m} x {k-1, k} and uses each of
// the 16 taps once: out(a,q)[ch] = sum_{ci,ty,tx} x[ci][m-ty][k-tx] * W[ci][ch][a+2ty][q+2tx].  Per output parity
// class (a,q) that is a GEMM  D[ch][quad] = A[ch][(ci,tap)] * B[(ci,tap)][quad]  with K = 16 ci x 4 taps:
//   * B (the 2x2 input blocks of 16 consecutive quads) is the SAME for the four classes -> 16 loads feed 64 MFMAs;
//   * A (the weights, 4 classes x 16 steps = 64 VGPRs) is loaded once per workgroup and reused for every tile
//     -- no scalar-cache traffic in the loop, which is what throttled the VALU/SGPR formulation;
//   * the next tile's B operands and this tile's bias terms are in flight while the 64 MFMAs run.
// One wave = one tile of 16 quads (32 output columns x 2 rows x 16 channels) per iteration.
__global__ __launch_bounds__(256) void tail_conv_fwd_kernel(TAIL_FWD_ARGS) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, mt = blockIdx.y;
  const int h = p.h, w = p.w, CH = p.CH, E = p.E;
  const size_t HWi = (size_t)h * w, HWo = 4 * HWi;
  const int ty = g >> 1, tx = g & 1;
  // A[cls = 2a+q][ci]: row i = channel mt*16 + j, k = tap g
  float A[4][kCi];
  {
    const int ch = mt * 16 + j;
    const float* wp = pweff + ((size_t)(p.wB == 1 ? 0 : b) * kCi * CH + min(ch, CH - 1)) * 16;
#pragma unroll
    for (int ci = 0; ci < kCi; ++ci)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float v = wp[(size_t)ci * CH * 16 + (a + 2 * ty) * 4 + q + 2 * tx];
          A[2 * a + q][ci] = ch < CH ? v : 0.f;
        }
  }
  const float* xb = px + (size_t)b * kCi * HWi;
  const int tpr = (w + 1 + 15) >> 4, ntiles = (h + 1) * tpr;
  const int stride = gridDim.x * 4;
  float Bn[kCi];
  auto issue = [&](int tile) {  // B[ci]: k = tap g, column = quad j
    const int m = tile / tpr, k0 = (tile - m * tpr) * 16;
    const int r = m - ty, c = k0 + j - tx;
    const bool v = r >= 0 && r < h && c >= 0 && c < w;
    const size_t off = (size_t)min(max(r, 0), h - 1) * w + min(max(c, 0), w - 1);
#pragma unroll
    for (int ci = 0; ci < kCi; ++ci) {
      const float t = xb[(size_t)ci * HWi + off];
      Bn[ci] = v ? t : 0.f;
    }
  };
  int tile = blockIdx.x * 4 + wave;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += stride) {
    const int m = tile / tpr, k0 = (tile - m * tpr) * 16;
    float Bc[kCi];
#pragma unroll
    for (int ci = 0; ci < kCi; ++ci) Bc[ci] = Bn[ci];
    // this tile's additive terms: D row i = channel mt*16 + 4g + r, column = quad j
    float init[4][4];
    size_t ooff[4];
    bool ook[4];
    const int kq = k0 + j;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int oy = 2 * m - 1 + a, ox = 2 * kq - 1 + q;
        ook[2 * a + q] = oy >= 0 && oy < 2 * h && ox >= 0 && ox < 2 * w;
        ooff[2 * a + q] = (size_t)min(max(oy, 0), 2 * h - 1) * (2 * w) + min(max(ox, 0), 2 * w - 1);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = mt * 16 + 4 * g + r, chc = min(ch, CH - 1);
      // contracted channels: pass 1 left their bias in `out`; direct channels: the bias plane itself
      const float* src = chc < E ? pout + ((size_t)b * CH + chc) * HWo : pbias + (size_t)(p.nd + chc - E) * HWo;
#pragma unroll
      for (int cls = 0; cls < 4; ++cls) init[cls][r] = src[ooff[cls]];
    }
    if (tile + stride < ntiles) issue(tile + stride);
    f32x4 acc[4];
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) acc[cls] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < kCi; ++ci)
#pragma unroll
      for (int cls = 0; cls < 4; ++cls)
        acc[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cls][ci], Bc[ci], acc[cls], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = mt * 16 + 4 * g + r;
      if (ch < CH) {
        float* dst = pout + ((size_t)b * CH + ch) * HWo;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls)
          if (ook[cls]) dst[ooff[cls]] = acc[cls][r] + init[cls][r];
      }
    }
  }
}

#define TAIL_BWD_ARGS const TailDims p, const float* __restrict__ px, const float* __restrict__ pwt, \
                      const float* __restrict__ plc, const float* __restrict__ pg, float* __restrict__ pgx, \
                      float* __restrict__ pgw, float* __restrict__ pgbias, float* __restrict__ pscratch

// gx[b,ci,iy,ix] = sum_{ch,dy,dx} g[b,ch,2iy-1+dy,2ix-1+dx] * w[b,ci,ch,dy,dx];  wt = w as [wB,CH,4,4,16]
__global__ __launch_bounds__(256) void tail_conv_bwd_x_kernel(TAIL_BWD_ARGS) {
  const int ix = blockIdx.x * 256 + threadIdx.x, iy = blockIdx.y, b = blockIdx.z;
  const int h = p.h, w = p.w, CH = p.CH;
  if (ix >= w) return;
  const size_t HWi = (size_t)h * w, HWo = 4 * HWi;
  const float* wt = pwt + (size_t)(p.wB == 1 ? 0 : b) * CH * 256;
  const float* gb = pg + (size_t)b * CH * HWo;
  float acc[kCi];
#pragma unroll
  for (int ci = 0; ci < kCi; ++ci) acc[ci] = 0.f;
#pragma unroll
  for (int dy = 0; dy < 4; ++dy) {
    const int oy = 2 * iy - 1 + dy;
    if (oy < 0 || oy >= 2 * h) continue;  // block-uniform
    const int ox0 = 2 * ix - 1;
    const bool ok0 = ox0 >= 0, ok3 = ox0 + 3 < 2 * w;
    for (int ch = 0; ch < CH; ++ch) {
      const float* gr = gb + (size_t)ch * HWo + (size_t)oy * (2 * w) + ox0;
      float gv[4];  // unconditional loads (edge lanes re-read a neighbour), then masked: no divergent branches
      gv[0] = gr[ok0 ? 0 : 1]; gv[1] = gr[1]; gv[2] = gr[2]; gv[3] = gr[ok3 ? 3 : 2];
      gv[0] = ok0 ? gv[0] : 0.f; gv[3] = ok3 ? gv[3] : 0.f;
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const float* wq = wt + ((size_t)(ch * 4 + dy) * 4 + dx) * 16;  // wave-uniform
#pragma unroll
        for (int ci = 0; ci < kCi; ++ci) acc[ci] += gv[dx] * wq[ci];
      }
    }
  }
#pragma unroll
  for (int ci = 0; ci < kCi; ++ci) pgx[((size_t)b * kCi + ci) * HWi + (size_t)iy * w + ix] = acc[ci];
}

// gw[b,ci,ch,dy,dx] += sum_{iy,ix} x[b,ci,iy,ix] * g[b,ch,2iy-1+dy,2ix-1+dx]
// A workgroup walks tiles of one input row x 64 input pixels: x[16][64] and the matching g window
// [16 ch][4 rows][130 cols] are staged in LDS with coalesced loads; each of the 4 waves then feeds its 16-pixel
// chunk to 64 MFMAs (16 taps x K = 16 pixels / 4 per instruction).  A[i = ci][k], B[k][j = ch] operands come from
// LDS in the (lane&15, lane>>4) layout the instruction wants.
constexpr int kWTile = 64;
constexpr int kGRow = 2 * kWTile + 4;     // 130 used
constexpr int kGPlane = 4 * kGRow + 2;    // == 2 (mod 32): the 16 planes of a lane group spread over the banks
constexpr int kXRow = kWTile + 4;         // == 4 (mod 32)

__global__ __launch_bounds__(256) void tail_conv_bwd_w_kernel(TAIL_BWD_ARGS) {
  __shared__ __attribute__((aligned(16))) float s_mem[16 * kGPlane + 16 * kXRow];
  float* s_g = s_mem;
  float* s_x = s_mem + 16 * kGPlane;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, nt = blockIdx.y;
  const int h = p.h, w = p.w, CH = p.CH;
  const size_t HWi = (size_t)h * w, HWo = 4 * HWi;
  const int l15 = lane & 15, kq = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int dy = 0; dy < 4; ++dy)
#pragma unroll
    for (int dx = 0; dx < 4; ++dx) acc[dy][dx] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tpr = (w + kWTile - 1) / kWTile, ntiles = h * tpr;
  // Staging is software-pipelined: the 37 global loads of the NEXT tile are issued (into registers) before the
  // MFMA phase of the current one, all of them independent -- one memory latency per tile, overlapped with 64 MFMAs.
  //   g: wave -> 16 of the 64 (plane, row) segments, columns [0,128) as two 64-lane loads; the 2 leftover
  //      columns of the 64 segments go to threads 0..127;  x: thread -> (ci, 4 pixels)
  float rg[32], rl, rx[4];
  const int xci = threadIdx.x >> 4, xj4 = (threadIdx.x & 15) * 4;
  const int lseg = threadIdx.x >> 1, lcl = 2 * kWTile + (threadIdx.x & 1);
  auto issue = [&](int tile) {
    const int iy = tile / tpr, ix0 = (tile - iy * tpr) * kWTile;
    const float* xr = px + ((size_t)b * kCi + xci) * HWi + (size_t)iy * w;
#pragma unroll
    for (int u = 0; u < 4; ++u) rx[u] = xr[min(ix0 + xj4 + u, w - 1)];
#pragma unroll
    for (int sgm = 0; sgm < 16; ++sgm) {
      const int seg = wave * 16 + sgm, ch = nt * 16 + (seg >> 2), oy = 2 * iy - 1 + (seg & 3);
      const float* gr = pg + ((size_t)b * CH + min(ch, CH - 1)) * HWo + (size_t)min(max(oy, 0), 2 * h - 1) * (2 * w);
#pragma unroll
      for (int u = 0; u < 2; ++u) rg[2 * sgm + u] = gr[min(max(2 * ix0 - 1 + lane + 64 * u, 0), 2 * w - 1)];
    }
    {
      const int ch = nt * 16 + (lseg >> 2), oy = 2 * iy - 1 + (lseg & 3);
      const float* gr = pg + ((size_t)b * CH + min(ch, CH - 1)) * HWo + (size_t)min(max(oy, 0), 2 * h - 1) * (2 * w);
      rl = gr[min(max(2 * ix0 - 1 + lcl, 0), 2 * w - 1)];
    }
  };
  auto stage = [&](int tile) {  // registers -> LDS, out-of-range elements as zeros
    const int iy = tile / tpr, ix0 = (tile - iy * tpr) * kWTile;
#pragma unroll
    for (int u = 0; u < 4; ++u) s_x[xci * kXRow + xj4 + u] = (ix0 + xj4 + u < w) ? rx[u] : 0.f;
#pragma unroll
    for (int sgm = 0; sgm < 16; ++sgm) {
      const int seg = wave * 16 + sgm, pl = seg >> 2, dy = seg & 3, oy = 2 * iy - 1 + dy;
      const bool rv = nt * 16 + pl < CH && oy >= 0 && oy < 2 * h;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int cl = lane + 64 * u, col = 2 * ix0 - 1 + cl;
        s_g[pl * kGPlane + dy * kGRow + cl] = (rv && col >= 0 && col < 2 * w) ? rg[2 * sgm + u] : 0.f;
      }
    }
    if (threadIdx.x < 128) {
      const int pl = lseg >> 2, dy = lseg & 3, oy = 2 * iy - 1 + dy, col = 2 * ix0 - 1 + lcl;
      const bool rv = nt * 16 + pl < CH && oy >= 0 && oy < 2 * h && col < 2 * w;
      s_g[pl * kGPlane + dy * kGRow + lcl] = rv ? rl : 0.f;
    }
  };
  if ((int)blockIdx.x < ntiles) issue(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();  // the previous tile's operands have been consumed
    stage(tile);
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
    const int xl = 16 * wave + 4 * kq;  // this lane's 4 input pixels (K slice) inside the tile
    const float4 xa4 = *reinterpret_cast<const float4*>(&s_x[l15 * kXRow + xl]);
    const float xa[4] = {xa4.x, xa4.y, xa4.z, xa4.w};
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      const float* gp = &s_g[l15 * kGPlane + dy * kGRow + 2 * xl];  // column 0 of s_g is global column 2*ix0-1
      float gv[10];
#pragma unroll
      for (int q = 0; q < 10; q += 2) {
        const float2 t = *reinterpret_cast<const float2*>(gp + q);
        gv[q] = t.x; gv[q + 1] = t.y;
      }
#pragma unroll
      for (int dx = 0; dx < 4; ++dx)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[dy][dx] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[t], gv[2 * t + dx], acc[dy][dx], 0, 0, 0);
    }
  }
  // D[i = ci][j = ch]: lane holds column j = lane&15, rows (lane>>4)*4 + r.  Sum the 4 waves through LDS.
  __syncthreads();
  float (*s_red)[256] = reinterpret_cast<float (*)[256]>(s_mem);
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* sp = &s_red[dy * 4 + dx][(kq * 4 + r) * 16 + l15];
            *sp = (wv == 0 ? 0.f : *sp) + acc[dy][dx][r];
          }
    }
    __syncthreads();
  }
  if (pscratch) {
    // per-workgroup partials [B][NT][NBLK][16 taps][256 = ci*16 + ch_local], summed by tail_conv_bwd_w_reduce_kernel:
    // hundreds of workgroups adding to the same few thousand addresses with memory-side float atomics cost up to
    // half of this kernel
    float* dst = pscratch + ((((size_t)b * gridDim.y + nt) * gridDim.x + blockIdx.x) * 16) * 256 + threadIdx.x;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) dst[(size_t)tap * 256] = s_red[tap][threadIdx.x];
    return;
  }
  const int ci = threadIdx.x >> 4, cj = nt * 16 + (threadIdx.x & 15);
  if (cj < CH) {
    float* dst = pgw + (((size_t)(p.wB == 1 ? 0 : b) * kCi + ci) * CH + cj) * 16;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) atomicAdd(dst + tap, s_red[tap][threadIdx.x]);
  }
}

// gw[wb,ci,ch,tap] += sum over workgroups (and over views when the weights are shared) of the partials
__global__ __launch_bounds__(256) void tail_conv_bwd_w_reduce_kernel(const TailDims p, int NT, int NBLK,
                                                                     const float* __restrict__ scratch,
                                                                     float* __restrict__ gw) {
  const int tap = blockIdx.x, nt = blockIdx.y, b = blockIdx.z;
  const int ci = threadIdx.x >> 4, cj = nt * 16 + (threadIdx.x & 15);
  if (cj >= p.CH) return;
  const float* src = scratch + ((((size_t)b * NT + nt) * NBLK) * 16 + tap) * 256 + threadIdx.x;
  float acc = 0.f;
#pragma unroll 8
  for (int k = 0; k < NBLK; ++k) acc += src[(size_t)k * 16 * 256];
  if (p.wB == 1) atomicAdd(gw + (((size_t)ci) * p.CH + cj) * 16 + tap, acc);  // shared weights: B adds per address
  else gw[(((size_t)b * kCi + ci) * p.CH + cj) * 16 + tap] += acc;
}

// gbias[k,n] = sum_{b,e} lc[k,b,e] g[b,e,n]  (k < nd);  gbias[nd+j,n] = sum_b g[b,E+j,n]
template <int E>
__global__ __launch_bounds__(256) void tail_bias_bwd_kernel(TAIL_BWD_ARGS) {
  const size_t N = (size_t)4 * p.h * p.w;
  const size_t n = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (n >= N) return;
  const int CH = p.CH, nd = p.nd;
  for (int b0 = 0; b0 < p.B; b0 += kMaxB) {
    const bool accum = b0 > 0;
    if constexpr (E > 0) {
      float G[kMaxB][E][2];
#pragma unroll
      for (int bb = 0; bb < kMaxB; ++bb)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          G[bb][e][0] = G[bb][e][1] = 0.f;
          if (b0 + bb < p.B) {
            const float2 t = *reinterpret_cast<const float2*>(pg + ((size_t)(b0 + bb) * CH + e) * N + n);
            G[bb][e][0] = t.x; G[bb][e][1] = t.y;
          }
        }
#pragma unroll 1
      for (int kk = 0; kk < nd; ++kk) {
        float s0 = 0.f, s1 = 0.f;
        const float* L = plc + ((size_t)kk * p.B + b0) * E;
#pragma unroll
        for (int bb = 0; bb < kMaxB; ++bb) {
          if (b0 + bb < p.B) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const float l = L[bb * E + e];
              s0 += l * G[bb][e][0];
              s1 += l * G[bb][e][1];
            }
          }
        }
        float2* dst = reinterpret_cast<float2*>(pgbias + (size_t)kk * N + n);
        if (accum) { const float2 o = *dst; s0 += o.x; s1 += o.y; }
        *dst = make_float2(s0, s1);
      }
    }
    for (int j = 0; j < CH - E; ++j) {
      float s0 = 0.f, s1 = 0.f;
      for (int bb = 0; bb < kMaxB && b0 + bb < p.B; ++bb) {
        const float2 t = *reinterpret_cast<const float2*>(pg + ((size_t)(b0 + bb) * CH + E + j) * N + n);
        s0 += t.x; s1 += t.y;
      }
      float2* dst = reinterpret_cast<float2*>(pgbias + (size_t)(nd + j) * N + n);
      if (accum) { const float2 o = *dst; s0 += o.x; s1 += o.y; }
      *dst = make_float2(s0, s1);
    }
  }
}

int check_common(int B, int Ci, int h, int w, int CH, int E, int nd, int wB) {
  GOL_REQUIRE(B >= 0 && h > 0 && w > 0, "bad sizes");
  GOL_REQUIRE(Ci == kCi, "the last decoder layers have 16 input channels");
  GOL_REQUIRE(CH >= 1 && CH <= 32, "1 <= CH <= 32");
  GOL_REQUIRE(E == 0 || E == 3 || E == 6, "E (light-contracted channels) must be 0, 3 or 6");
  GOL_REQUIRE(E <= CH && nd >= 0 && (E > 0) == (nd > 0), "E and nd go together");
  GOL_REQUIRE(wB == 1 || wB == B, "weights are shared (wB = 1) or per view (wB = B)");
  GOL_REQUIRE(2 * h + 1 <= 65535 && B <= 65535, "grid limit");
  return GOL_OK;
}

}  // namespace

extern "C" int gol_tail_conv_fwd(int B, int Ci, int h, int w, int CH, int E, int nd, int wB, const float* x,
                                 const float* weff, const float* lc, const float* bias, float* out, void* stream) {
  const int rc = check_common(B, Ci, h, w, CH, E, nd, wB);
  if (rc != GOL_OK) return rc;
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(x && weff && bias && out && (E == 0 || lc), "null pointer");
  const TailDims p{B, h, w, CH, E, nd, wB};
  hipStream_t s = (hipStream_t)stream;
  if (E > 0) {  // pass 1: contracted bias -> out[:, :E]
    const dim3 gb(gol_cdiv((long long)2 * h * w, 256));
    if (E == 3) tail_bias_fwd_kernel<3><<<gb, 256, 0, s>>>(p, x, weff, lc, bias, out);
    else tail_bias_fwd_kernel<6><<<gb, 256, 0, s>>>(p, x, weff, lc, bias, out);
    GOL_CHECK_LAUNCH();
  }
  {  // pass 2: conv (+ direct bias planes), accumulating onto pass 1
    const int ntiles = (h + 1) * gol_cdiv(w + 1, 16), mtiles = gol_cdiv(CH, 16);
    int nblk = gol_cdiv(768, (long long)mtiles * B);  // ~3 workgroups per CU in total
    const int cap = gol_cdiv(ntiles, 4);
    nblk = nblk < 1 ? 1 : (nblk > cap ? cap : nblk);
    tail_conv_fwd_kernel<<<dim3(nblk, mtiles, B), 256, 0, s>>>(p, x, weff, lc, bias, out);
  }
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

static int bwd_w_blocks(int B, int h, int w, int CH) {
  const int ntiles = h * gol_cdiv(w, kWTile), ntile_ch = gol_cdiv(CH, 16);
  int nblk = gol_cdiv(768, (long long)ntile_ch * (B > 0 ? B : 1));  // ~3 workgroups per CU
  return nblk < 1 ? 1 : (nblk > ntiles ? ntiles : nblk);
}

extern "C" long long gol_tail_conv_bwd_scratch_floats(int B, int h, int w, int CH) {
  if (B <= 0 || h <= 0 || w <= 0 || CH <= 0) return 0;
  return (long long)B * gol_cdiv(CH, 16) * bwd_w_blocks(B, h, w, CH) * 16 * 256;
}

extern "C" int gol_tail_conv_bwd(int B, int Ci, int h, int w, int CH, int E, int nd, int wB, const float* x,
                                 const float* weff_t, const float* lc, const float* g_out, float* g_x, float* g_weff,
                                 float* g_bias, float* w_scratch, void* stream) {
  const int rc = check_common(B, Ci, h, w, CH, E, nd, wB);
  if (rc != GOL_OK) return rc;
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(g_out, "null pointer");
  GOL_REQUIRE(!g_x || weff_t, "g_x needs weff_t");
  GOL_REQUIRE(!g_weff || x, "g_weff needs x");
  GOL_REQUIRE(!g_bias || E == 0 || lc, "g_bias needs lc");
  const TailDims p{B, h, w, CH, E, nd, wB};
  hipStream_t s = (hipStream_t)stream;
  if (g_x) {
    tail_conv_bwd_x_kernel<<<dim3(gol_cdiv(w, 256), h, B), 256, 0, s>>>(p, x, weff_t, lc, g_out, g_x, g_weff, g_bias, nullptr);
    GOL_CHECK_LAUNCH();
  }
  if (g_weff) {
    const int ntile_ch = gol_cdiv(CH, 16), nblk = bwd_w_blocks(B, h, w, CH);
    tail_conv_bwd_w_kernel<<<dim3(nblk, ntile_ch, B), 256, 0, s>>>(p, x, weff_t, lc, g_out, g_x, g_weff, g_bias, w_scratch);
    GOL_CHECK_LAUNCH();
    if (w_scratch) {
      tail_conv_bwd_w_reduce_kernel<<<dim3(16, ntile_ch, B), 256, 0, s>>>(p, ntile_ch, nblk, w_scratch, g_weff);
      GOL_CHECK_LAUNCH();
    }
  }
  if (g_bias) {
    const dim3 grid(gol_cdiv((long long)2 * h * w, 256));
    if (E == 0) tail_bias_bwd_kernel<0><<<grid, 256, 0, s>>>(p, x, weff_t, lc, g_out, g_x, g_weff, g_bias, nullptr);
    else if (E == 3) tail_bias_bwd_kernel<3><<<grid, 256, 0, s>>>(p, x, weff_t, lc, g_out, g_x, g_weff, g_bias, nullptr);
    else tail_bias_bwd_kernel<6><<<grid, 256, 0, s>>>(p, x, weff_t, lc, g_out, g_x, g_weff, g_bias, nullptr);
    GOL_CHECK_LAUNCH();
  }
  return GOL_OK;
}
