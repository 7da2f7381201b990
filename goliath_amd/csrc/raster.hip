// raster.hip -- tile rasterizer of depth-sorted 2-D Gaussians, forward + backward, gfx950.
//
// Replaces gsplat 0.1.11 rasterize_forward / rasterize_backward_kernel (3-channel specialisation;
// not in the reference tree; call sites /root/reference/ca_code/utils/render_gsplat.py:65-78 and
// :91-104; semantics SURVEY.md A.3 / A.4).  CDNA4 design:
//   * one 128-thread workgroup per 16x16 tile = 2 wave64s, each wave owns a 16x8 half and each lane two
//     vertically adjacent pixels: the per-pixel recurrences run on 2-vectors, i.e. packed fp32 VALU ops
//     (v_pk_fma/mul/add_f32), and the cross-lane gradient reduction of the backward is paid once per 128 pixels;
//   * the tile's sorted list is staged through LDS in batches as three records per Gaussian; the per-Gaussian
//     reads in the pixel loop are wave-uniform ds_read_b128 broadcasts;
//   * at staging time every Gaussian gets a 2-bit half mask from the exact ellipse-vs-rectangle test of its
//     alpha >= 1/255 region, so a wave skips (scalar loop over a 64-bit ballot) Gaussians that cannot touch its
//     half -- gsplat's 3-sigma tile bbox is far looser than the 1/255 cut;
//   * colour and the optional 4th "extra" channel (depth) are composited in ONE pass instead of
//     the reference's two rasterize calls (render_gsplat.py:65-104);
//   * backward: per-Gaussian gradients are reduced over the 64 lanes four-at-a-time with
//     v_permlane32_swap / v_permlane16_swap + DPP row adds (no LDS, no atomics), parked in per-wave LDS
//     slots, merged across the waves once per batch and leave the workgroup as one float atomic per Gaussian per
//     tile per component (gsplat: one per 32-lane warp => 8x more); with 64-byte gradient records
//     (GOL_GRAD_RECORD) the 10 atomics of a Gaussian share a cache line and are issued by 16 adjacent lanes --
//     the memory-side atomic units, which cost 25 % of the kernel with dense [N,k] arrays, drop out of the profile;
//   * tile -> workgroup mapping interleaves tile rows over the 8 XCDs (tile_of_block).
#include <cstdlib>

#include "gol_common.h"

namespace {


constexpr int kBatch = 256;
// conics are staged in LDS pre-multiplied by log2(e) -- alpha = opacity * 2^(-sigma') is one v_exp_f32 with a negated
// operand instead of a multiply + exp per pixel -- and the diagonal terms by the 1/2 of sigma = (a dx^2 + c dy^2) / 2 +
// b dx dy as well (GOL_SC_A / GOL_SC_B, applied where the records are written); kUnA / kUnB bring the true conic back
// where the backward needs it
constexpr float kUnA = GOL_UN_A, kUnB = GOL_UN_B;
#ifdef GOL_EXACT_MATH
// TEST-ONLY exact-math twin (goliath_amd/build.py, variant "exact"): the conic is staged unscaled, sigma is evaluated in
// the order the CPU oracle (and gsplat) writes it -- 0.5 (a dx^2 + c dy^2) + b dx dy, every product and sum rounded
// separately -- exp goes through double precision (correctly rounded to fp32) and the transmittance recurrence is
// T (1 - alpha) instead of T - alpha T.  With bit-identical inputs the alpha >= 1/255 and T <= 1e-4 decisions then
// coincide with the oracle's: what remains between the two is rounding noise, no threshold flips.
__device__ __forceinline__ float exact_sigma(float a, float b, float c, float dx, float dy) {
#pragma clang fp contract(off)
  const float t1 = (a * dx) * dx, t2 = (c * dy) * dy, t3 = (b * dx) * dy;
  const float h = 0.5f * (t1 + t2);
  return h + t3;
}
__device__ __forceinline__ float exact_exp_neg(float s) { return (float)exp(-(double)s); }
__device__ __forceinline__ float exact_next_T(float T, float alpha) {
#pragma clang fp contract(off)
  const float om = 1.f - alpha;
  return T * om;
}
#endif

struct TileCoord { int tile, tx, ty; bool ok; };

// XCD-aware remap.  Consecutive workgroups land on different XCDs (observed: block b -> XCD b % 8), so
// XCD x is given tile rows x, x+8, x+16, ...: inside a die consecutive workgroups walk along a tile
// row (neighbouring tiles share most of their Gaussians -> L2 hits), while the rows of every die are
// spread over the whole image so the dies stay balanced (contiguous image eighths per die left most of
// the chip idle: the head covers only the middle rows).  Speed only -- any mapping is correct.
__device__ __forceinline__ TileCoord tile_of_block(int bid, int T, int tiles_x) {
  const int tiles_y = T / tiles_x;
  const int xcd = bid & 7, j = bid >> 3;
  TileCoord tc;
  const int row_local = j / tiles_x;
  tc.tx = j - row_local * tiles_x;
  tc.ty = row_local * 8 + xcd;
  tc.ok = tc.ty < tiles_y;
  tc.tile = tc.ty * tiles_x + tc.tx;
  return tc;
}

typedef float f1 __attribute__((ext_vector_type(1)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i1 __attribute__((ext_vector_type(1)));
typedef int i2 __attribute__((ext_vector_type(2)));

// Wave footprints.  PPL = pixels per lane.
//   PPL = 2 (the default): 2 waves per 16x16 tile, wave w owns the 16x8 half (rows 8w..8w+7), lane = (x = lane & 15, row
//           pair lane >> 4) owns two vertically adjacent pixels -- the per-pixel recurrences run on 2-vectors (packed fp32
//           VALU ops) and the per-Gaussian work of a visit is shared by 128 pixels: the fewest instructions per pixel.
//   PPL = 1 (round 4, launches of one or two views): 4 waves per tile, wave w owns the 8x8 quadrant (x half w & 1, y half
//           w >> 1), one pixel per lane.  ~9 % more instructions in total, but a wave's chain through the tile's list -- which
//           IS the duration of a single-view launch: 2942 non-empty tiles of 670 entries on average and up to 1280 fit on
//           the chip at once, the launch ends when the longest list does (measured: raster_bwd 227 us for one view, 131 us
//           per view in an 8-view launch; with every list clipped to 512 entries 130 us) -- is ~0.55x as long: the 8x8
//           footprint is culled against 21 % more entries and a visit costs ~0.7x the instructions.
template <int PPL> struct Pix;
template <> struct Pix<2> { typedef f2 fv; typedef i2 iv; static constexpr int kWaves = 2; };
template <> struct Pix<1> { typedef f1 fv; typedef i1 iv; static constexpr int kWaves = 4; };

template <int PPL>
__device__ __forceinline__ void wave_rect(int wave, float tile_x0, float tile_y0, float& x0, float& x1, float& y0, float& y1) {
  if (PPL == 2) {
    x0 = tile_x0 + 0.5f; x1 = x0 + 15.f; y0 = tile_y0 + (float)(wave * 8) + 0.5f; y1 = y0 + 7.f;
  } else {
    x0 = tile_x0 + (float)((wave & 1) * 8) + 0.5f; x1 = x0 + 7.f; y0 = tile_y0 + (float)((wave >> 1) * 8) + 0.5f; y1 = y0 + 7.f;
  }
}

// bit w of the mask = the alpha >= 1/255 region of a Gaussian can reach the footprint of wave w:
// exact ellipse-vs-rectangle test (minimum of sigma over the footprint's pixel centres against ln(255*opacity)),
// conservative only by a rounding margin; degenerate conics -> all.  tau, 1/a, 1/c and the validity flag come from the
// record (gol_common.h: computed once per Gaussian by the projection).
template <int PPL>
__device__ __forceinline__ int wave_mask(float gx, float gy, float ca, float cb, float cc, float tau, float ia, float ic,
                                         float exact, float tile_x0, float tile_y0) {
  constexpr int NW = Pix<PPL>::kWaves;
  if (!(tau >= 0.f)) return 0;  // alpha < 1/255 everywhere (also NaN opacity: skipped by gsplat too)
  if (exact == 0.f) return (1 << NW) - 1;
  int m = 0;
#pragma unroll
  for (int q = 0; q < NW; ++q) {
    float x0, x1, y0, y1;
    wave_rect<PPL>(q, tile_x0, tile_y0, x0, x1, y0, y1);
    const float ms = gol_min_sigma_rect(gx, gy, ca, cb, cc, ia, ic, x0, x1, y0, y1);
    m |= (ms <= tau) ? (1 << q) : 0;
  }
  return m;
}

// pixel coordinates of a lane: column j, first row i0 (the lane's PPL pixels are rows i0 .. i0 + PPL - 1)
template <int PPL>
__device__ __forceinline__ void lane_pixel(int tx, int ty, int wave, int lane, int& j, int& i0) {
  if (PPL == 2) { j = tx * 16 + (lane & 15); i0 = ty * 16 + wave * 8 + (lane >> 4) * 2; }
  else { j = tx * 16 + (wave & 1) * 8 + (lane & 7); i0 = ty * 16 + (wave >> 1) * 8 + (lane >> 3); }
}

// stage one list entry: four 16-byte loads from the Gaussian's 64-byte record
// LDS row of a staged list entry: a = (x, y, conic a', conic b'), b = (conic c', opacity, r, g), c = (b, extra)
struct __attribute__((aligned(16))) StagedRow { float4 a, b; float2 c; float2 pad; };

struct Staged { float4 a, b; float2 c; int mask; };
template <int PPL>
__device__ __forceinline__ Staged stage_entry(const float* __restrict__ records, size_t g, float tile_x0, float tile_y0) {
  const float4* R = reinterpret_cast<const float4*>(records + g * GOL_SPLAT_RECORD);
  const float4 q0 = R[0], q1 = R[1], q2 = R[2], q3 = R[3];
  Staged s;
  s.a = q0;                              // x, y, a', b'
  s.b = q1;                              // c', opacity, r, g
  s.c = make_float2(q2.x, q2.y);         // b, extra
  s.mask = wave_mask<PPL>(q0.x, q0.y, q0.z * kUnA, q0.w * kUnB, q1.x * kUnA, q2.z, q2.w, q3.x, q3.y, tile_x0, tile_y0);
  return s;
}

template <typename V, int P>
__device__ __forceinline__ bool any_live(const V& live) {
  unsigned u = 0u;
#pragma unroll
  for (int q = 0; q < P; ++q) u |= __float_as_uint(live[q]);
  return u != 0u;
}

// Forward.  One workgroup per 16x16 tile (PPL = 2: 128 threads, PPL = 1: 256; see Pix).
// LAZY (the fused path, planar images): everything that only changes when a pixel STOPS -- its liveness, the index the
// backward may start from, the "is this half finished" test -- moves out of the per-visit instruction stream into a
// wave-uniform branch taken only on visits where some pixel of the half stops (a pixel stops once; a visit costs 7 vector
// instructions less).  final_idx then holds, per pixel, an UPPER BOUND of the index of its last contributor that excludes
// the entry it stopped at: stop index - 1, or the end of the list for a pixel that never stopped.  The backward only needs
// that (entries between the true last contributor and the bound fail its alpha >= 1/255 test again).  !LAZY (the
// gsplat-compatible operator): gsplat's exact final_idx.
template <bool EXTRA, bool LAZY, int PPL>
__global__ __launch_bounds__(64 * Pix<PPL>::kWaves) void raster_fwd_kernel(
    int N, int img_h, int img_w, int planar, int tiles_x, int tiles_y, const int2* __restrict__ tile_bins,
    const int32_t* __restrict__ sorted_ids, int64_t capacity, const float* __restrict__ records,
    const float* __restrict__ background, float* __restrict__ out_img,
    float* __restrict__ out_extra, float* __restrict__ final_Ts, int32_t* __restrict__ final_idx,
    float* __restrict__ out_alpha, float* __restrict__ out_extra_norm, float norm_lo,
    const float* __restrict__ l1_target, const float* __restrict__ l1_mask, int l1_mask_c,
    uint8_t* __restrict__ l1_sign, float* __restrict__ l1_partial, int n_views) {
  typedef typename Pix<PPL>::fv fv;
  typedef typename Pix<PPL>::iv iv;
  constexpr int NW = Pix<PPL>::kWaves, NT = 64 * NW;
  __shared__ float s_l1[NW];
  // (separate arrays: the 48-byte rows of the backward kernel, which save it an address register move per visit, cost the
  // forward 3-5 % -- measured side by side on one box, profiles/r03g_raster_ab.txt)
  __shared__ float4 s_a[kBatch];  // x, y, conic.a, conic.b
  __shared__ float4 s_b[kBatch];  // conic.c, opacity, r, g
  __shared__ float2 s_c[kBatch];  // b, extra  (an 8-byte stride on purpose: with a 16-byte one the compiler issues all
                                  // three reads of a visit at the loop top from one address register, and the forward
                                  // runs 3-4 % slower -- profiles/r03g_raster_ab.txt)
  __shared__ int32_t s_mask[kBatch];  // wave mask
  const int T = tiles_x * tiles_y;
  // workgroup g = slot * B + view (the view is the fastest index: the workgroups of all views of a launch are handed out
  // together -- the last view's long lists do not start when the others are already done; with B = 8 a view also stays
  // on one die, with B = 4 on two)
  const int view = blockIdx.x % n_views, slot = blockIdx.x / n_views;
  const TileCoord tc = tile_of_block(slot, T, tiles_x);
  if (!tc.ok) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int j, i0;
  lane_pixel<PPL>(tc.tx, tc.ty, wave, lane, j, i0);
  bool in[PPL];
  fv py, live;
#pragma unroll
  for (int q = 0; q < PPL; ++q) {
    in[q] = (i0 + q < img_h) && (j < img_w);
    py[q] = (float)(i0 + q) + 0.5f;
    // 1 while the pixel still composites, 0 once it stopped (or lies outside the image).  Carried as a FLOAT in a VGPR
    // and multiplied into alpha: a dead pixel then fails the alpha >= 1/255 test by itself.  As loop-carried booleans the
    // same state lived in SGPR lane masks and cost ~30 scalar instructions per visit to maintain (PMC: the scalar pipes
    // of this kernel were as busy as the vector pipes, 76 % / 78 %).
    live[q] = in[q] ? 1.f : 0.f;
  }
  const float px = (float)j + 0.5f;

  const int2 range = tile_bins[(size_t)view * T + tc.tile];
  const int32_t* ids = sorted_ids + (size_t)view * capacity;
  const size_t goff = (size_t)view * N;

  fv T_cur = 1.f;
  iv cur_idx = 0;
  fv acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;

  const int n_batches = (range.y - range.x + kBatch - 1) / kBatch;
  for (int bb = 0; bb < n_batches; ++bb) {
    if (__syncthreads_and(!any_live<fv, PPL>(live))) break;  // also protects the LDS batch from being overwritten early
    const int batch_start = range.x + bb * kBatch;
    for (int k = tid; k < kBatch; k += NT) {
      const int idx = batch_start + k;
      if (idx < range.y) {
        const Staged st = stage_entry<PPL>(records, goff + (size_t)ids[idx], (float)(tc.tx * 16), (float)(tc.ty * 16));
        s_a[k] = st.a; s_b[k] = st.b; s_c[k] = st.c; s_mask[k] = st.mask;
      } else {
        s_mask[k] = 0;
      }
    }
    __syncthreads();
    const int batch_size = min(kBatch, range.y - batch_start);
    // Each wave walks ONLY the entries whose alpha >= 1/255 region touches its footprint: one ballot per
    // 64 entries, then a scalar loop over the set bits (s_ff1) -- culled entries cost nothing.
    for (int chunk = 0; chunk < batch_size; chunk += 64) {
      unsigned long long bits = gol_ballot((s_mask[chunk + lane] >> wave) & 1);
      if (LAZY && gol_ballot(any_live<fv, PPL>(live)) == 0ull) break;  // footprint finished
      while (bits) {
        if (!LAZY && gol_ballot(any_live<fv, PPL>(live)) == 0ull) { chunk = batch_size; break; }  // this wave's footprint is finished
        const int t = chunk + __builtin_ctzll(bits);
        bits &= bits - 1;
        const float4 a4 = s_a[t];
        const float4 b4 = s_b[t];
        const float2 c2 = s_c[t];
        // branch-free pixel update: selects instead of exec-mask regions (the loop is issue-bound)
        const float dx = a4.x - px;
        const fv dy = a4.y - py;
        fv sigma, alpha;
#ifndef GOL_EXACT_MATH
        sigma = (a4.z * dx * dx + b4.x * dy * dy) + (a4.w * dx) * dy;  // log2e * gsplat's sigma
#pragma unroll
        for (int q = 0; q < PPL; ++q) alpha[q] = fminf(GOL_ALPHA_CAP_FWD, b4.y * __builtin_amdgcn_exp2f(-sigma[q]));
#else
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
          sigma[q] = exact_sigma(a4.z, a4.w, b4.x, dx, dy[q]);
          alpha[q] = fminf(GOL_ALPHA_CAP_FWD, b4.y * exact_exp_neg(sigma[q]));
        }
#endif
        alpha *= live;
        fv vis = alpha * T_cur;
#ifndef GOL_EXACT_MATH
        const fv next_T = T_cur - vis;  // = T (1 - alpha)
#else
        fv next_T;
#pragma unroll
        for (int q = 0; q < PPL; ++q) next_T[q] = exact_next_T(T_cur[q], alpha[q]);
#endif
        // contributes: !(sigma < 0 || alpha < 1/255); stop / take: one compare per pixel -- all as scalar lane masks
        // (ballots of the plain compares; written with bools the compiler issues a second, NaN-aware compare for the
        // negation).  (An early-out for visits without a taker, as the backward has it, does not pay here: 602-612 vs 613 us)
        unsigned long long mc[PPL], ms[PPL], any_stop = 0ull;
        bool take[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
          mc[q] = gol_ballot(!(sigma[q] < 0.f)) & gol_ballot(!(alpha[q] < GOL_ALPHA_FLOOR));
          ms[q] = gol_ballot(next_T[q] <= GOL_T_STOP);
          take[q] = __builtin_amdgcn_inverse_ballot_w64(mc[q] & ~ms[q]);
          any_stop |= mc[q] & ms[q];
        }
        bool half_done = false;
        if (!LAZY || any_stop != 0ull) {   // LAZY: wave-uniform and rare -- some pixel stops here
#pragma unroll
          for (int q = 0; q < PPL; ++q) {
            const bool stop = __builtin_amdgcn_inverse_ballot_w64(mc[q] & ms[q]);
            live[q] = stop ? 0.f : live[q];
            if (LAZY) cur_idx[q] = stop ? (batch_start + t - 1) : cur_idx[q];
          }
          if (LAZY) half_done = gol_ballot(any_live<fv, PPL>(live)) == 0ull;
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) vis[q] = take[q] ? vis[q] : 0.f;
        acc0 += b4.z * vis; acc1 += b4.w * vis; acc2 += c2.x * vis;
        if (EXTRA) acc3 += c2.y * vis;
#ifndef GOL_EXACT_MATH
        T_cur -= vis;                   // unchanged where the entry is not taken
#else
#pragma unroll
        for (int q = 0; q < PPL; ++q) T_cur[q] = take[q] ? next_T[q] : T_cur[q];
#endif
        if (!LAZY) {
#pragma unroll
          for (int q = 0; q < PPL; ++q) cur_idx[q] = take[q] ? (batch_start + t) : cur_idx[q];
        }
        if (LAZY && half_done) { chunk = batch_size; break; }
      }
    }
  }
  if (LAZY) {   // a pixel that never stopped may have taken entries up to the end of the list
#pragma unroll
    for (int q = 0; q < PPL; ++q) cur_idx[q] = live[q] != 0.f ? range.y - 1 : cur_idx[q];
  }

  // planar: [B,3,H,W] (what the model consumes, rgca.py:139); else gsplat's [B,H,W,3].
  // Addressing: wave-uniform per-view base pointers (scalar registers) + one 32-bit pixel offset per lane, so every
  // access is the SGPR-base + VGPR-offset form -- the epilogue runs for ALL tiles (71 % of them empty at the benchmarked
  // views) and 64-bit per-lane address arithmetic was a visible share of the kernel's vector instructions.
  const unsigned hw = (unsigned)img_h * (unsigned)img_w;
  const size_t vplane = (size_t)view * hw;
  float* __restrict__ o_T = final_Ts + vplane;
  int32_t* __restrict__ o_idx = final_idx + vplane;
  float* __restrict__ o_img = out_img + 3 * vplane;
  float* __restrict__ o_ex = (EXTRA && out_extra) ? out_extra + vplane : nullptr;
  float* __restrict__ o_alpha = out_alpha ? out_alpha + vplane : nullptr;
  float* __restrict__ o_norm = (EXTRA && out_extra_norm) ? out_extra_norm + vplane : nullptr;
  const float* __restrict__ i_tgt = l1_target ? l1_target + 3 * vplane : nullptr;
  const float* __restrict__ i_mask = l1_mask ? l1_mask + (size_t)l1_mask_c * vplane : nullptr;
  uint8_t* __restrict__ o_sign = l1_target ? l1_sign + vplane : nullptr;
  const float bg0 = background[0], bg1 = background[1], bg2 = background[2];
  // channel planes as separate uniform bases (non-planar: element 3 * pix + c)
  float* __restrict__ o_img1 = o_img + (planar ? hw : 1u);
  float* __restrict__ o_img2 = o_img + (planar ? 2u * hw : 2u);
  // a tile without entries: its final_idx / sign bytes are never read (the backward skips the tile: its range is
  // empty) -- 5 of the 41 bytes per pixel that the epilogue moves for the ~70 % empty tiles of a head-and-shoulders view
  // (final_T = 1 is still written: render_gsplat.render returns it)
  const bool bwd_state = range.y > range.x;
  float l1_acc = 0.f;
#pragma unroll
  for (int q = 0; q < PPL; ++q) {
    if (!in[q]) continue;
    const unsigned pix = (unsigned)(i0 + q) * (unsigned)img_w + (unsigned)j;
    const unsigned b4 = pix * 4u;                  // byte offset inside a one-channel plane (checked < 4 GiB on the host)
    const unsigned bimg = planar ? b4 : 3u * b4;
    const float Tq = T_cur[q];
    const float c0 = acc0[q] + Tq * bg0, c1 = acc1[q] + Tq * bg1, c2 = acc2[q] + Tq * bg2, ex = acc3[q];
    *gol_at(o_T, b4) = Tq;
    if (bwd_state || !LAZY) *gol_at(o_idx, b4) = cur_idx[q];
    *gol_at(o_img, bimg) = c0; *gol_at(o_img1, bimg) = c1; *gol_at(o_img2, bimg) = c2;
    if (o_ex) *gol_at(o_ex, b4) = ex;
    // optional fused epilogue of AutoEncoder.render (rgca.py:137,144-145): alpha = 1 - T, depth / clamp(alpha, lo, 1)
    if (o_alpha) *gol_at(o_alpha, b4) = 1.f - Tq;
    if (o_norm) *gol_at(o_norm, b4) = ex / fminf(fmaxf(1.f - Tq, norm_lo), 1.f);
    // optional fused masked L1 against a target image (rgb_l1, ca_code/loss/__init__.py:391-411; planar layout): the
    // |difference| goes into a per-tile partial sum and its sign -- the loss gradient up to mask x scalar -- into ONE
    // byte per pixel (2 bits per channel: sign + 1), which the backward decodes as its upstream image gradient: the two
    // separate passes over the image of the loss disappear, and the epilogue writes 1 instead of 12 bytes per pixel for
    // it (the epilogue's traffic is not hidden: 71 % of the tiles of the benchmarked views are empty and do nothing else)
    if (i_tgt) {
      const float m0 = i_mask ? *gol_at(i_mask, b4) : 1.f;
      const float cs[3] = {c0, c1, c2};
      unsigned code = 0u;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float m = (i_mask && l1_mask_c == 3) ? *gol_at(i_mask + (size_t)c * hw, b4) : m0;
        const float d = (cs[c] - *gol_at(i_tgt + (size_t)c * hw, b4)) * m;
        l1_acc += fabsf(d);
        code |= (d > 0.f ? 2u : (d < 0.f ? 0u : 1u)) << (2 * c);
      }
      if (bwd_state) *gol_at(o_sign, pix) = (uint8_t)code;
    }
  }
  if (l1_target) {  // (kernel-uniform) per-tile sum of |difference|: the caller adds the tiles up (deterministic)
    const float ws = gol_wave_sum_to_lane63(l1_acc);
    if (lane == 63) s_l1[wave] = ws;
    __syncthreads();
    if (tid == 0) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += s_l1[w];
      l1_partial[(size_t)view * T + tc.tile] = tot;
    }
  }
}

// loss value of the fused L1: l1_out[0] = scale * sum(l1_partial[0 .. n)), one workgroup, fixed summation order.  (Folding
// this into the raster launch itself -- the last workgroup to finish adds the partial sums up -- was measured in round 4:
// the device-scope fence every workgroup then needs before its ticket writes back / invalidates L2 and took the 8-view step
// from 2.9 to 6.7 ms; a 3 us kernel of its own replaces the ATen reduction + multiply of rounds 1-3.)
__global__ __launch_bounds__(1024) void l1_sum_kernel(int n, const float* __restrict__ l1_partial, float scale,
                                                      float* __restrict__ l1_out) {
  __shared__ float s_w[16];
  // 16-byte loads, four of them in flight per lane (a plain strided loop waits for every load: 33 us for 86 k floats)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int n4 = n >> 2;
  const float4* p4 = reinterpret_cast<const float4*>(l1_partial);
  int i = threadIdx.x;
  for (; i + 3 * 1024 < n4; i += 4 * 1024) {
    const float4 v0 = p4[i], v1 = p4[i + 1024], v2 = p4[i + 2048], v3 = p4[i + 3072];
    a0 += (v0.x + v0.y) + (v0.z + v0.w); a1 += (v1.x + v1.y) + (v1.z + v1.w);
    a2 += (v2.x + v2.y) + (v2.z + v2.w); a3 += (v3.x + v3.y) + (v3.z + v3.w);
  }
  for (; i < n4; i += 1024) { const float4 v = p4[i]; a0 += (v.x + v.y) + (v.z + v.w); }
  if ((int)threadIdx.x < (n & 3)) a1 += l1_partial[4 * n4 + threadIdx.x];
  const float ws = gol_wave_sum_to_lane63((a0 + a1) + (a2 + a3));
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = ws;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += s_w[w];
    l1_out[0] = tot * scale;
  }
}

constexpr int kBatchB = 64;   // backward batch (smaller: per-wave gradient slots live in LDS)
constexpr int kAcc = 12;      // r g b v_opacity | Mx My Mxx Mxy | Myy extra - -   (M = moments of gop, see the loop)

// ---- backward ------------------------------------------------------------------------------------
// Same workgroup / wave / lane layout as the forward (Pix).  With two pixels per lane the per-pixel recurrences of the
// two pixels are independent, so the body is written on 2-vectors and maps onto packed fp32 VALU ops (v_pk_fma/mul/add_f32:
// two pixels per instruction), and the cross-lane reduction of the 10 per-Gaussian sums -- the largest single cost
// of the one-pixel-per-lane kernel -- is paid once per 128 pixels instead of once per 64.
static_assert(kBatchB == 64, "raster_bwd_kernel ballots one 64-entry chunk per batch");

template <bool EXTRA, bool PACKED, int PPL>
__global__ __launch_bounds__(64 * Pix<PPL>::kWaves) void raster_bwd_kernel(
    int N, int img_h, int img_w, int planar, int tiles_x, int tiles_y, const int2* __restrict__ tile_bins,
    const int32_t* __restrict__ sorted_ids, int64_t capacity, const float* __restrict__ records,
    const float* __restrict__ background,
    const float* __restrict__ final_Ts, const int32_t* __restrict__ final_idx,
    const float* __restrict__ v_out_img, const float* __restrict__ v_out_extra,
    const float* __restrict__ v_out_alpha, float* __restrict__ v_xy, float* __restrict__ v_conic,
    float* __restrict__ v_colors, float* __restrict__ v_extra, float* __restrict__ v_opacity,
    const uint8_t* __restrict__ v_sign, const float* __restrict__ v_sign_mask, int v_sign_mask_c,
    const float* __restrict__ v_img_scale, float v_img_scale_mul, int n_views) {
  typedef typename Pix<PPL>::fv fv;
  typedef typename Pix<PPL>::iv iv;
  constexpr int NW = Pix<PPL>::kWaves, NT = 64 * NW;
  __shared__ StagedRow s_e[kBatchB];  // one 48-byte row per entry: the reads of a visit share their address register
  __shared__ int32_t s_mask[kBatchB];
  __shared__ int32_t s_id[kBatchB];
  __shared__ __attribute__((aligned(16))) float s_acc[NW][kBatchB][kAcc];
  __shared__ int32_t s_touched[NW][kBatchB];
  __shared__ int32_t s_wmax[NW];
  const int T = tiles_x * tiles_y;
  const int view = blockIdx.x % n_views, slot = blockIdx.x / n_views;   // (see raster_fwd_kernel)
  const TileCoord tc = tile_of_block(slot, T, tiles_x);
  if (!tc.ok) return;
  const int2 range = tile_bins[(size_t)view * T + tc.tile];
  if (range.y <= range.x) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int j, i0;
  lane_pixel<PPL>(tc.tx, tc.ty, wave, lane, j, i0);
  const float px = (float)j + 0.5f;
  const size_t hw = (size_t)img_h * img_w;
  const int32_t* ids = sorted_ids + (size_t)view * capacity;
  const size_t goff = (size_t)view * N;
  bool in[PPL];
  size_t pp[PPL];
  fv py, T_final;
  iv bin_final;
#pragma unroll
  for (int q = 0; q < PPL; ++q) {
    in[q] = (i0 + q < img_h) && (j < img_w);
    py[q] = (float)(i0 + q) + 0.5f;
    pp[q] = in[q] ? ((size_t)view * img_h + i0 + q) * img_w + j : 0;
    T_final[q] = in[q] ? final_Ts[pp[q]] : 1.f;
    bin_final[q] = in[q] ? final_idx[pp[q]] : (range.x - 1);
  }
  fv T_cur = T_final;
  fv vo0 = 0.f, vo1 = 0.f, vo2 = 0.f, vo3 = 0.f, voa = 0.f;
  {
    // upstream image gradient = v_out_img (optional) + the fused L1's term: (sign code - 1) x mask x v_img_scale, with the
    // sign codes the forward epilogue left (one byte per pixel) and the scalar g / n as a device value (no sync)
    const float vsc = (v_img_scale ? v_img_scale[0] : 1.f) * v_img_scale_mul;
    const size_t os = planar ? hw : 1;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
      if (!in[q]) continue;
      const size_t p = pp[q];
      const int i = i0 + q;
      const size_t o = planar ? (size_t)view * 3 * hw + (size_t)i * img_w + j : 3 * p;
      float g0 = 0.f, g1 = 0.f, g2 = 0.f;
      if (v_out_img) { g0 = v_out_img[o]; g1 = v_out_img[o + os]; g2 = v_out_img[o + 2 * os]; }
      if (v_sign) {
        const unsigned code = v_sign[p];
        float m0 = vsc, m1 = vsc, m2 = vsc;
        if (v_sign_mask) {
          const float* mk = v_sign_mask + (size_t)view * v_sign_mask_c * hw + (size_t)i * img_w + j;
          m0 *= mk[0]; m1 *= (v_sign_mask_c == 3) ? mk[hw] : mk[0]; m2 *= (v_sign_mask_c == 3) ? mk[2 * hw] : mk[0];
        }
        g0 += (float)((int)(code & 3u) - 1) * m0;
        g1 += (float)((int)((code >> 2) & 3u) - 1) * m1;
        g2 += (float)((int)((code >> 4) & 3u) - 1) * m2;
      }
      vo0[q] = g0; vo1[q] = g1; vo2[q] = g2;
      if (EXTRA && v_out_extra) vo3[q] = v_out_extra[p];
      if (v_out_alpha) voa[q] = v_out_alpha[p];
    }
  }
  const fv tail = T_final * (voa - (background[0] * vo0 + background[1] * vo1 + background[2] * vo2));
  fv qsum = 0.f;  // running sum over the Gaussians behind of fac * <colour, v_out>

  int wmax = bin_final[0];
#pragma unroll
  for (int q = 1; q < PPL; ++q) wmax = max(wmax, bin_final[q]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off, 64));
  wmax = __builtin_amdgcn_readfirstlane(wmax);
  if (lane == 0) s_wmax[wave] = wmax;
  __syncthreads();
  int bmax = s_wmax[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) bmax = max(bmax, s_wmax[w]);
  bmax = min(bmax, range.y - 1);
  if (bmax < range.x) return;

  float* acc_lane = &s_acc[wave][0][lane >> 4];  // this lane's column (lanes 15, 31, 47, 63 hold sums 0..3 / 4..7)
  const int n_batches = (bmax - range.x + kBatchB) / kBatchB;
  // A batch is staged through two dependent loads (list id -> the Gaussian's record).  The id of the NEXT batch's entry is
  // fetched while this batch is being worked on, so that only the record gather is left behind the barrier (nothing hides it
  // in the tail of a launch, when a SIMD is down to one or two waves).
  auto entry_id = [&](int bb2) {
    const int e = bmax - bb2 * kBatchB - tid;
    return (tid < kBatchB && e >= range.x) ? ids[e] : 0;
  };
  int gid_next = entry_id(0);
  for (int bb = 0; bb < n_batches; ++bb) {
    __syncthreads();
    const int batch_end = bmax - bb * kBatchB;
    const int batch_size = min(kBatchB, batch_end + 1 - range.x);
    const int gid = gid_next;
    if (tid < kBatchB) {
      if (tid < batch_size) {
        const Staged st = stage_entry<PPL>(records, goff + (size_t)gid, (float)(tc.tx * 16), (float)(tc.ty * 16));
        s_e[tid].a = st.a; s_e[tid].b = st.b; s_e[tid].c = st.c; s_mask[tid] = st.mask;
        s_id[tid] = gid;
      } else {
        s_mask[tid] = 0;
      }
    }
    if (bb + 1 < n_batches) gid_next = entry_id(bb + 1);
    (&s_touched[0][0])[tid] = 0;  // NW * kBatchB == blockDim
    __syncthreads();

    const int t0 = max(0, batch_end - wmax);
    unsigned long long bits = gol_ballot((s_mask[lane] >> wave) & 1);  // kBatchB == 64: one chunk
    if (t0 > 0) bits &= ~0ull << t0;
    while (bits) {
      const int t = __builtin_ctzll(bits);
      bits &= bits - 1;
      const float4 a4 = s_e[t].a;
      const float4 b4 = s_e[t].b;
      const float2 c2 = s_e[t].c;
      const int li = batch_end - t;
      const float dx = a4.x - px;
      const fv dy = a4.y - py;
      fv sigma, vis;
#ifndef GOL_EXACT_MATH
      sigma = (a4.z * dx * dx + b4.x * dy * dy) + (a4.w * dx) * dy;
#pragma unroll
      for (int q = 0; q < PPL; ++q) vis[q] = __builtin_amdgcn_exp2f(-sigma[q]);  // sigma = log2e * gsplat's
#else
#pragma unroll
      for (int q = 0; q < PPL; ++q) { sigma[q] = exact_sigma(a4.z, a4.w, b4.x, dx, dy[q]); vis[q] = exact_exp_neg(sigma[q]); }
#endif
      fv alpha = b4.y * vis;
      // taken by the pixel: within its list && !(sigma < 0 || alpha < 1/255) -- as scalar lane masks (ballots of the plain
      // compares; the ballot of a combined bool costs a v_cndmask + v_cmp)
      unsigned long long mv[PPL], many = 0ull;
#pragma unroll
      for (int q = 0; q < PPL; ++q) {
        alpha[q] = fminf(GOL_ALPHA_CAP_BWD, alpha[q]);
        mv[q] = gol_ballot(li <= bin_final[q]) & gol_ballot(!(sigma[q] < 0.f)) & gol_ballot(!(alpha[q] < GOL_ALPHA_FLOOR));
        many |= mv[q];
      }
      if (many == 0ull) continue;
      bool v[PPL];
      fv ra;
#pragma unroll
      for (int q = 0; q < PPL; ++q) {
        v[q] = __builtin_amdgcn_inverse_ballot_w64(mv[q]);
        // an entry the pixel did not take enters with alpha = 0: 1 / (1 - 0) = 1 exactly, so T and the running sums pass
        // through unchanged without further selects
        alpha[q] = v[q] ? alpha[q] : 0.f;
      }
      const fv one_m = 1.f - alpha;
#pragma unroll
      for (int q = 0; q < PPL; ++q) ra[q] = __builtin_amdgcn_rcpf(one_m[q]);
      const fv T_new = T_cur * ra;
      const fv fac = alpha * T_new;
      T_cur = T_new;
      // gsplat: v_alpha = sum_c (rgb_c T - buffer_c ra) v_out_c + T_final ra (v_out_alpha - <bg, v_out>) with
      // buffer_c = sum over the Gaussians behind of rgb_c alpha T.  All channels enter through ONE dot product with the
      // upstream gradient, w = <colour, v_out>, so the three running colour buffers collapse into the running scalar
      // q = sum_behind fac w:  v_alpha = T w + ra (tail - q)
      fv w = b4.z * vo0 + b4.w * vo1 + c2.x * vo2;
      if (EXTRA) w += c2.y * vo3;
      const fv v_alpha = T_new * w + ra * (tail - qsum);
      qsum += fac * w;
      // d loss / d sigma per pixel is -opacity * gop; the (wave-uniform) factor -opacity is applied once per Gaussian in
      // the merge step: the lanes reduce the moments of gop itself, whose zeroth moment IS v_opacity
      fv gop = vis * v_alpha;
#pragma unroll
      for (int q = 0; q < PPL; ++q) gop[q] = v[q] ? gop[q] : 0.f;
      const fv gy = gop * dy;
      const fv gyy = gy * dy;
      // the lane's pixels, summed (PPL = 2: one multiply + one scalar FMA per colour sum -- written on scalars: from the
      // 2-vector form the compiler builds v_mul + v_pk_fma and throws the packed op's upper half away)
      float g0s, g1s, g2s, g3 = 0.f, m0, my, myy;
      if (PPL == 2) {
        g0s = __builtin_fmaf(fac[0], vo0[0], fac[PPL - 1] * vo0[PPL - 1]);
        g1s = __builtin_fmaf(fac[0], vo1[0], fac[PPL - 1] * vo1[PPL - 1]);
        g2s = __builtin_fmaf(fac[0], vo2[0], fac[PPL - 1] * vo2[PPL - 1]);
        if (EXTRA) g3 = __builtin_fmaf(fac[0], vo3[0], fac[PPL - 1] * vo3[PPL - 1]);
        m0 = gop[0] + gop[PPL - 1];       // sum_pix gop
        my = gy[0] + gy[PPL - 1];         // sum gop dy
        myy = gyy[0] + gyy[PPL - 1];
      } else {
        g0s = fac[0] * vo0[0]; g1s = fac[0] * vo1[0]; g2s = fac[0] * vo2[0];
        if (EXTRA) g3 = fac[0] * vo3[0];
        m0 = gop[0]; my = gy[0]; myy = gyy[0];
      }
      const float mx = m0 * dx, mxx = mx * dx, mxy = my * dx;
      const float r0 = gol_wave_sum4(g0s, g1s, g2s, m0);
      const float r1 = gol_wave_sum4(mx, my, mxx, mxy);
      // the 9th (and 10th) sum: a 6-instruction DPP ladder each (total in lane 63) instead of a third 4-way reduction
      const float r2 = gol_wave_sum_to_lane63(myy);
      const float r3 = EXTRA ? gol_wave_sum_to_lane63(g3) : 0.f;
      if ((lane & 15) == 15) {
        // slot row t of this wave + a 32-bit byte offset (plain pointer arithmetic on the int t becomes a v_mad_u64_u32)
        float* a = gol_at(acc_lane, (unsigned)t * (unsigned)(kAcc * sizeof(float)));
        a[0] = r0; a[4] = r1;
        if (lane == 63) { a[8 - 3] = r2; a[9 - 3] = r3; }  // (lane 63's base points at slot 3)
        s_touched[wave][t] = 1;
      }
    }
    __syncthreads();
    if (PACKED) {
      // gradients live in 64-byte records [r g b | opacity | x y | conic a b c | extra | pad]: 16 consecutive
      // lanes own one Gaussian's record, so an atomic instruction touches 4 cache lines instead of 64
      float* rec = v_colors;
      for (int idx = tid; idx < batch_size * 16; idx += NT) {
        const int t = idx >> 4, c = idx & 15;
        bool any = false;
#pragma unroll
        for (int w = 0; w < NW; ++w) any = any || (s_touched[w][t] != 0);
        if (!any || c > (EXTRA ? 9 : 8)) continue;
        // component c = w1 * S[k1] + w2 * S[k2] of the wave-summed slots S; slots 4..8 hold moments of gop:
        // v_sigma-sums = -opacity * moment (conic back from its log2e scaling with ln 2)
        const float4 a4 = s_e[t].a;
        const float4 b4 = s_e[t].b;
        const float nop = -b4.y, cc = b4.x;
        const int k1 = (c == 5) ? 4 : c, k2 = 5;
        const float w1 = (c == 4) ? nop * a4.z * kUnA : (c == 5) ? nop * a4.w * kUnB : (c == 6 || c == 8) ? 0.5f * nop
                       : (c == 7) ? nop : 1.f;
        const float w2 = (c == 4) ? nop * a4.w * kUnB : (c == 5) ? nop * cc * kUnA : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const bool tw = s_touched[w][t] != 0;
          s1 += tw ? s_acc[w][t][k1] : 0.f;
          s2 += tw ? s_acc[w][t][k2] : 0.f;
        }
        atomicAdd(rec + (goff + (size_t)s_id[t]) * 16 + c, w1 * s1 + w2 * s2);
      }
    } else if (tid < batch_size) {
      float a[kAcc];
#pragma unroll
      for (int k = 0; k < kAcc; ++k) a[k] = 0.f;
      bool any = false;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        if (s_touched[w][tid]) {
          any = true;
          const float4* q = reinterpret_cast<const float4*>(&s_acc[w][tid][0]);
          const float4 q0 = q[0], q1 = q[1], q2 = q[2];
          a[0] += q0.x; a[1] += q0.y; a[2] += q0.z; a[3] += q0.w;
          a[4] += q1.x; a[5] += q1.y; a[6] += q1.z; a[7] += q1.w;
          a[8] += q2.x; a[9] += q2.y;
        }
      }
      if (any) {
        const size_t g = goff + (size_t)s_id[tid];
        const float4 a4 = s_e[tid].a;
        const float4 b4 = s_e[tid].b;
        const float nop = -b4.y;  // slots 4..8 are moments of gop: v_sigma-sums = -opacity * moment
        const float ca = a4.z * kUnA, cb = a4.w * kUnB, cc = b4.x * kUnA;
        atomicAdd(v_colors + 3 * g, a[0]); atomicAdd(v_colors + 3 * g + 1, a[1]); atomicAdd(v_colors + 3 * g + 2, a[2]);
        atomicAdd(v_opacity + g, a[3]);
        atomicAdd(v_xy + 2 * g, nop * (ca * a[4] + cb * a[5]));
        atomicAdd(v_xy + 2 * g + 1, nop * (cb * a[4] + cc * a[5]));
        atomicAdd(v_conic + 3 * g, 0.5f * nop * a[6]); atomicAdd(v_conic + 3 * g + 1, nop * a[7]);
        atomicAdd(v_conic + 3 * g + 2, 0.5f * nop * a[8]);
        if (EXTRA && v_extra) atomicAdd(v_extra + g, a[9]);
      }
    }
  }
}

// DIAGNOSTIC (bench.py's algorithmic roofline): per view, the number of (pixel, list entry) pairs up to the pixel's
// final_idx ("tested": what any per-pixel compositor has to look at) and of those with alpha >= 1/255 ("taken": what is
// composited and differentiated).  One thread per pixel, no staging: slow and simple.
__global__ __launch_bounds__(256) void raster_count_pairs_kernel(int N, int img_h, int img_w, int tiles_x, int tiles_y,
                                                                 const int2* __restrict__ tile_bins,
                                                                 const int32_t* __restrict__ sorted_ids, int64_t capacity,
                                                                 const float* __restrict__ records,
                                                                 const int32_t* __restrict__ final_idx,
                                                                 unsigned long long* __restrict__ counts) {
  __shared__ unsigned long long s_cnt[2];
  const int view = blockIdx.y, tile = blockIdx.x, T = tiles_x * tiles_y;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int j = tx * 16 + (threadIdx.x & 15), i = ty * 16 + (threadIdx.x >> 4);
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0ull;
  __syncthreads();
  const int2 range = tile_bins[(size_t)view * T + tile];
  unsigned long long tested = 0, taken = 0;
  if (i < img_h && j < img_w && range.y > range.x) {
    const int last = min(range.y - 1, final_idx[((size_t)view * img_h + i) * img_w + j]);
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    for (int li = range.x; li <= last; ++li) {
      const float4* R = reinterpret_cast<const float4*>(records + ((size_t)view * N + sorted_ids[(size_t)view * capacity + li]) * GOL_SPLAT_RECORD);
      const float4 q0 = R[0], q1 = R[1];
      const float dx = q0.x - px, dy = q0.y - py;
      const float sg = (q0.z * dx * dx + q1.x * dy * dy) + (q0.w * dx) * dy;
#ifndef GOL_EXACT_MATH
      const float alpha = fminf(GOL_ALPHA_CAP_FWD, q1.y * __builtin_amdgcn_exp2f(-sg));
#else
      const float alpha = fminf(GOL_ALPHA_CAP_FWD, q1.y * expf(-sg));
#endif
      ++tested;
      taken += (!(sg < 0.f) && !(alpha < GOL_ALPHA_FLOOR)) ? 1 : 0;
    }
  }
  atomicAdd(&s_cnt[0], tested);
  atomicAdd(&s_cnt[1], taken);
  __syncthreads();
  if (threadIdx.x < 2 && s_cnt[threadIdx.x]) atomicAdd(counts + 2 * view + threadIdx.x, s_cnt[threadIdx.x]);
}

// gsplat-compatible operators hand over separate attribute arrays: pack them into records (the fused path's projection
// writes the records itself)
__global__ __launch_bounds__(256) void splat_pack_kernel(size_t n, const float* __restrict__ xys,
                                                         const float* __restrict__ conics,
                                                         const float* __restrict__ colors,
                                                         const float* __restrict__ extra,
                                                         const float* __restrict__ opacities,
                                                         float* __restrict__ records) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  gol_record_write(records + e * GOL_SPLAT_RECORD, xys[2 * e], xys[2 * e + 1], conics[3 * e], conics[3 * e + 1],
                   conics[3 * e + 2], opacities[e], colors[3 * e], colors[3 * e + 1], colors[3 * e + 2],
                   extra ? extra[e] : 0.f);
}

}  // namespace

// The forward's wave footprint for a launch of B views (GOL_RASTER_PPL = 1 | 2 overrides it for experiments).  One or two
// views do not fill the chip for the whole launch: all their workgroups are resident at once, every SIMD keeps the waves it
// was dealt, and the launch lasts until the most loaded SIMD is done (per-workgroup timeline, profiles/r04_raster_tail.txt:
// 2942 workgroups of a view start within 5 us, half of them are done after 120 us, the last one after 257 us; an 8-view
// launch takes 131-148 us per view).  Four waves per tile spread a tile's work over more SIMDs: forward 121 -> 106 us for
// one view; in the backward the 17 % extra instructions of the finer footprint cancel the gain (236 -> 240 us): it keeps
// two pixels per lane.  From three views on the instruction count decides.
extern "C" int gol_raster_plan(int B, int* fwd_pixels_per_lane) {
  static const int forced = getenv("GOL_RASTER_PPL") ? atoi(getenv("GOL_RASTER_PPL")) : 0;
  if (fwd_pixels_per_lane) *fwd_pixels_per_lane = (forced == 1 || forced == 2) ? forced : (B <= 2 ? 1 : 2);
  return GOL_OK;
}

extern "C" int gol_raster_count_pairs(int B, int N, int img_h, int img_w, const int32_t* tile_bins, const int32_t* sorted_ids,
                                      int64_t capacity, const float* records, const int32_t* final_idx, uint64_t* counts,
                                      void* stream) {
  GOL_REQUIRE(B >= 0 && N >= 0 && img_h > 0 && img_w > 0, "bad size");
  if (B == 0) return GOL_OK;
  GOL_REQUIRE(tile_bins && final_idx && counts && (capacity == 0 || sorted_ids) && (N == 0 || records), "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(counts, 0, sizeof(uint64_t) * 2 * (size_t)B, s) != hipSuccess) {
    gol_set_error("gol_raster_count_pairs: hipMemsetAsync failed");
    return GOL_ERR_LAUNCH;
  }
  const int tiles_x = (img_w + 15) / 16, tiles_y = (img_h + 15) / 16;
  raster_count_pairs_kernel<<<dim3(tiles_x * tiles_y, B), 256, 0, s>>>(
      N, img_h, img_w, tiles_x, tiles_y, reinterpret_cast<const int2*>(tile_bins), sorted_ids, capacity, records, final_idx,
      reinterpret_cast<unsigned long long*>(counts));
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_splat_pack(int B, int N, const float* xys, const float* conics, const float* colors,
                              const float* extra, const float* opacities, float* records, void* stream) {
  GOL_REQUIRE(B >= 0 && N >= 0, "negative size");
  if (B == 0 || N == 0) return GOL_OK;
  GOL_REQUIRE(xys && conics && colors && opacities && records, "null pointer");
  const size_t n = (size_t)B * N;
  splat_pack_kernel<<<gol_cdiv((long long)n, 256), 256, 0, (hipStream_t)stream>>>(n, xys, conics, colors, extra, opacities,
                                                                                 records);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_rasterize_fwd(int B, int N, int img_h, int img_w, int block, int planar, const int32_t* tile_bins,
                                 const int32_t* sorted_ids, int64_t capacity, const float* records, int with_extra,
                                 const float* background, float* out_img,
                                 float* out_extra, float* final_Ts, int32_t* final_idx, float* out_alpha,
                                 float* out_extra_norm, float norm_lo, const float* l1_target, const float* l1_mask,
                                 int l1_mask_c, uint8_t* l1_sign, float* l1_partial, float* l1_out, float l1_scale,
                                 int pixels_per_lane, void* stream) {
  GOL_REQUIRE(B >= 0 && N >= 0, "negative size");
  GOL_REQUIRE(block == 16, "only block_width == 16 is implemented (the reference's value, render_gsplat.py:28)");
  GOL_REQUIRE(img_h > 0 && img_w > 0, "empty image");
  if (B == 0) return GOL_OK;
  GOL_REQUIRE((int64_t)B * 8 * (((img_h + 15) / 16 + 7) / 8) * ((img_w + 15) / 16) < (1ll << 31), "too many tiles");
  GOL_REQUIRE(tile_bins && background && out_img && final_Ts && final_idx, "null pointer");
  GOL_REQUIRE((uint64_t)img_h * (uint64_t)img_w * 12ull < (1ull << 32), "image too large (32-bit byte offsets inside a view)");
  GOL_REQUIRE(capacity == 0 || sorted_ids, "null sorted_ids");
  GOL_REQUIRE(N == 0 || records, "null Gaussian records");
  GOL_REQUIRE((!out_extra && !out_extra_norm) || with_extra || N == 0, "out_extra / out_extra_norm need the extra channel");
  GOL_REQUIRE(!with_extra || out_extra || out_extra_norm, "extra without an output for it");
  GOL_REQUIRE(!l1_target || (planar && l1_sign && l1_partial), "the fused L1 needs planar images, l1_sign and l1_partial");
  GOL_REQUIRE(!l1_mask || (l1_target && (l1_mask_c == 1 || l1_mask_c == 3)), "l1_mask: 1 or 3 channels, with l1_target");
  GOL_REQUIRE(!l1_out || l1_target, "l1_out needs l1_target");
  const int tiles_x = (img_w + 15) / 16, tiles_y = (img_h + 15) / 16;
  dim3 grid(8 * ((tiles_y + 7) / 8) * tiles_x * B);
  const int2* bins = reinterpret_cast<const int2*>(tile_bins);
  hipStream_t s = (hipStream_t)stream;
  GOL_REQUIRE(pixels_per_lane >= 0 && pixels_per_lane <= 2, "pixels_per_lane: 0 (choose by B), 1 or 2");
  // one pixel per lane (4 waves per tile) for launches of one or two views: their duration is the longest list's chain
  // through one wave, which the finer footprint shortens to ~0.55x for ~9 % more instructions in total (see Pix)
  int ppl = pixels_per_lane;
  if (ppl == 0) gol_raster_plan(B, &ppl);
#define GOL_LAUNCH_FWD(EX, LZ)                                                                                          \
  do {                                                                                                                  \
    if (ppl == 2)                                                                                                       \
      raster_fwd_kernel<EX, LZ, 2><<<grid, 128, 0, s>>>(N, img_h, img_w, planar, tiles_x, tiles_y, bins, sorted_ids,   \
                                                        capacity, records, background, out_img, out_extra, final_Ts,   \
                                                        final_idx, out_alpha, EX ? out_extra_norm : nullptr, norm_lo,   \
                                                        l1_target, l1_mask, l1_mask_c, l1_sign, l1_partial, B);         \
    else                                                                                                                \
      raster_fwd_kernel<EX, LZ, 1><<<grid, 256, 0, s>>>(N, img_h, img_w, planar, tiles_x, tiles_y, bins, sorted_ids,   \
                                                        capacity, records, background, out_img, out_extra, final_Ts,   \
                                                        final_idx, out_alpha, EX ? out_extra_norm : nullptr, norm_lo,   \
                                                        l1_target, l1_mask, l1_mask_c, l1_sign, l1_partial, B);         \
  } while (0)
  const bool ex = out_extra || out_extra_norm;
  // planar = the fused path: final_idx is the backward's start bound (see raster_fwd_kernel); gsplat's layout: exact
  if (ex && planar) GOL_LAUNCH_FWD(true, true);
  else if (ex) GOL_LAUNCH_FWD(true, false);
  else if (planar) GOL_LAUNCH_FWD(false, true);
  else GOL_LAUNCH_FWD(false, false);
#undef GOL_LAUNCH_FWD
  if (l1_out) l1_sum_kernel<<<1, 1024, 0, s>>>(B * tiles_x * tiles_y, l1_partial, l1_scale, l1_out);
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}

extern "C" int gol_rasterize_bwd(int B, int N, int img_h, int img_w, int block, int planar, const int32_t* tile_bins,
                                 const int32_t* sorted_ids, int64_t capacity, const float* records, int with_extra,
                                 const float* background, const float* final_Ts,
                                 const int32_t* final_idx, const float* v_out_img, const float* v_out_extra,
                                 const float* v_out_alpha, float* v_xy, float* v_conic, float* v_colors,
                                 float* v_extra, float* v_opacity, int grad_stride, const uint8_t* v_sign,
                                 const float* v_sign_mask, int v_sign_mask_c, const float* v_img_scale,
                                 float v_img_scale_mul, int pixels_per_lane, void* stream) {
  GOL_REQUIRE(B >= 0 && N >= 0, "negative size");
  GOL_REQUIRE(block == 16, "only block_width == 16 is implemented (the reference's value, render_gsplat.py:28)");
  GOL_REQUIRE(img_h > 0 && img_w > 0, "empty image");
  if (B == 0 || N == 0 || capacity == 0) return GOL_OK;
  GOL_REQUIRE((int64_t)B * 8 * (((img_h + 15) / 16 + 7) / 8) * ((img_w + 15) / 16) < (1ll << 31), "too many tiles");
  GOL_REQUIRE(tile_bins && sorted_ids && background && final_Ts && final_idx, "null pointer");
  GOL_REQUIRE(v_out_img || v_sign, "no upstream image gradient (v_out_img or v_sign)");
  GOL_REQUIRE(!v_sign || planar, "the sign image of the fused L1 goes with planar images");
  GOL_REQUIRE(!v_sign_mask || (v_sign && (v_sign_mask_c == 1 || v_sign_mask_c == 3)), "v_sign_mask: 1 or 3 channels, with v_sign");
  GOL_REQUIRE(records, "null Gaussian records");
  GOL_REQUIRE(v_xy && v_conic && v_colors && v_opacity, "null gradient output");
  GOL_REQUIRE(!(v_out_extra || v_extra) || with_extra, "extra-channel gradients need the extra channel");
  GOL_REQUIRE(grad_stride == 0 || grad_stride == GOL_GRAD_RECORD, "grad_stride must be 0 (dense arrays) or 16 (records)");
  const bool packed = grad_stride == GOL_GRAD_RECORD;
  if (packed)
    GOL_REQUIRE(v_opacity == v_colors + 3 && v_xy == v_colors + 4 && v_conic == v_colors + 6 &&
                    (!v_extra || v_extra == v_colors + 9),
                "record layout is [rgb | opacity | xy | conic | extra | pad] (GOL_GRAD_RECORD floats)");
  const int tiles_x = (img_w + 15) / 16, tiles_y = (img_h + 15) / 16;
  dim3 grid(8 * ((tiles_y + 7) / 8) * tiles_x * B);
  const int2* bins = reinterpret_cast<const int2*>(tile_bins);
  hipStream_t s = (hipStream_t)stream;
  GOL_REQUIRE(pixels_per_lane >= 0 && pixels_per_lane <= 2, "pixels_per_lane: 0 (2), 1 or 2");
  const int ppl = pixels_per_lane ? pixels_per_lane : 2;
  const bool ex = with_extra && (v_out_extra || v_extra);
#define GOL_LAUNCH_BWD(EX, PK)                                                                                          \
  do {                                                                                                                  \
    if (ppl == 2)                                                                                                       \
      raster_bwd_kernel<EX, PK, 2><<<grid, 128, 0, s>>>(N, img_h, img_w, planar, tiles_x, tiles_y, bins, sorted_ids,   \
                                                        capacity, records, background, final_Ts, final_idx, v_out_img, \
                                                        EX ? v_out_extra : nullptr, v_out_alpha, v_xy, v_conic,        \
                                                        v_colors, EX ? v_extra : nullptr, v_opacity, v_sign,           \
                                                        v_sign_mask, v_sign_mask_c, v_img_scale, v_img_scale_mul, B);  \
    else                                                                                                                \
      raster_bwd_kernel<EX, PK, 1><<<grid, 256, 0, s>>>(N, img_h, img_w, planar, tiles_x, tiles_y, bins, sorted_ids,   \
                                                        capacity, records, background, final_Ts, final_idx, v_out_img, \
                                                        EX ? v_out_extra : nullptr, v_out_alpha, v_xy, v_conic,        \
                                                        v_colors, EX ? v_extra : nullptr, v_opacity, v_sign,           \
                                                        v_sign_mask, v_sign_mask_c, v_img_scale, v_img_scale_mul, B);  \
  } while (0)
  if (ex && packed) GOL_LAUNCH_BWD(true, true);
  else if (ex) GOL_LAUNCH_BWD(true, false);
  else if (packed) GOL_LAUNCH_BWD(false, true);
  else GOL_LAUNCH_BWD(false, false);
#undef GOL_LAUNCH_BWD
  GOL_CHECK_LAUNCH();
  return GOL_OK;
}
