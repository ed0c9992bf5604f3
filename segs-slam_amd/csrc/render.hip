// render.hip -- per-tile alpha compositing, forward and backward (gfx950 / MI355X).
//
//   render_fwd_kernel  K10  reference: renderCUDA forward   cuda_rasterizer/forward.cu:339-452
//   render_bwd_kernel  K11  reference: renderCUDA backward  cuda_rasterizer/backward.cu:399-557
//
// Neither kernel is HBM-bound: per (Gaussian, 8x8 quadrant) the forward issues ~22 vector instructions, the backward ~39,
// against 64 B of record; the forward is VALU-issue bound, the backward VALU- and LDS-bound at once.
// MI355X mapping:
//  * a 16x16 tile (the binning unit, fixed by the reference's key format) is rendered by 4 wave64; each wave owns an 8x8
//    pixel quadrant and walks the tile's list on its own: no workgroup barrier anywhere.  The forward launches them as one
//    256-thread workgroup, the backward as four single-wave workgroups on the same XCD (its lists end at different depths
//    per quadrant, and a finished wave should give its LDS and registers back at once).
//  * the list is read 64 entries at a time with one coalesced vector load; each entry carries a 4-bit "quadrants this
//    instance can reach" mask (gs_layout.h); the entries of this wave's quadrant are compacted to the low lanes with
//    ds_permute, their 64-byte records gathered one per lane (two chunks in flight ahead of the one being evaluated).
//  * the per-Gaussian operands are wave-uniform: the batch's records are staged in LDS and read as broadcasts
//    (scalar loads of the records, the first design, were issue- and latency-bound; operands by DPP row broadcast cost as
//    much VALU issue as the LDS time they save -- both measured).
//  * early termination is per wave (8x8 block) instead of per 16x16 tile; a terminated pixel keeps T = 0 and a negative
//    threshold, which makes later Gaussians arithmetic no-ops for it without per-lane branches.
//  * exp(power) = exp2(log2e * power) with log2e folded into the stored conic (v_exp_f32).
//  * backward: two roles alternate inside each wave, 16 list entries at a time (see render_bwd_kernel): pixel lanes
//    emit w = dL/dG * G and alpha*T into LDS tiles, Gaussian lanes read them transposed and accumulate nine moment sums
//    with FMAs (dL/dpixel arrives by DPP row broadcast fused into those FMAs); 4 row partials are folded with
//    v_permlane32/16_swap and the sums of 4 Gaussians leave as ONE float-atomic wave instruction (36 contiguous bytes per
//    Gaussian) instead of the reference's 9 atomics per (pixel, Gaussian) pair.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gs_layout.h"
#include "kernels.h"

namespace segs {

// XCD-aware workgroup -> tile mapping.  Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so with
// the identity mapping horizontally adjacent tiles -- which share most of their Gaussians -- land on 8 different L2s and
// every record is fetched up to 8 times.  Here each group of 64 consecutive tiles is cut into 8 runs of 8 adjacent tiles,
// one run per XCD: neighbours share an L2, while the work stays interleaved finely enough for load balance.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t wg, uint32_t num_tiles) {
  const uint32_t full = num_tiles & ~63u;
  if (wg >= full) return wg;                       // ragged tail: identity
  const uint32_t xcd = wg & 7u, local = wg >> 3;   // local = index among this XCD's workgroups
  return (local >> 3) * 64u + xcd * 8u + (local & 7u);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}


__device__ __forceinline__ float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Compiler-only fence: LDS operations of one wave execute in order in hardware; this keeps the compiler from
// reordering this lane's LDS stores and the (other lanes' data) loads that follow.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }

// Wave-level compaction of the list entries this wave (quadrant) must evaluate: lanes 0..n-1 receive, in list
// order, the entry value and its index inside the 64-entry chunk.  ds_permute moves data between lanes through
// the LDS crossbar without touching LDS memory.
struct Compacted { uint32_t val; uint32_t pos; int n; };
template <bool REVERSE>
__device__ __forceinline__ Compacted compact_chunk(uint32_t v, uint32_t qbit, int lane) {
  const bool rel = (v & qbit) != 0u;
  const uint64_t m = __ballot(rel);
  const int n = __popcll(m);
  const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  // forward: k-th relevant entry from the front goes to lane k; reverse: k-th from the back goes to lane k
  const int rank = REVERSE ? (n - 1 - below) : below;
  const int dest = rel ? rank : n + (lane - below);  // a permutation of 0..63
  Compacted c;
  c.val = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)v);
  c.pos = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, lane);
  c.n = n;
  return c;
}

constexpr int FWD_REC = 12;  // floats per staged record: x y a2 b2 | c2 o r g | b pos - -

__global__ void __launch_bounds__(256) render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
    const float* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color) {
  __shared__ float4 s_rec[4][64][FWD_REC / 4];
  const uint32_t tiles_x = (W + TILE_X - 1) / TILE_X;
  const uint32_t tile = xcd_tile(blockIdx.x, gridDim.x);
  const uint32_t tile_x = tile % tiles_x, tile_y = tile / tiles_x;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t px = tile_x * TILE_X + (wv & 1) * 8 + (lane & 7);
  const uint32_t py = tile_y * TILE_Y + (wv >> 1) * 8 + (lane >> 3);
  const bool inside = px < (uint32_t)W && py < (uint32_t)H;
  const float pxf = (float)px, pyf = (float)py;
  const uint32_t qbit = 1u << (ID_BITS + wv);

  const uint2 range = ranges[tile];
  // Per-pixel state.  A pixel that has terminated ("dead") keeps T = 0 and thr = -1, which makes every later
  // Gaussian a no-op for it without any per-lane branch: w = alpha*T = 0 and the stop test T' < thr is false.
  float T = inside ? 1.0f : 0.f, thr = inside ? 0.0001f : -1.f;
  float Tfin = 0.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t last = 0u, lastfin = 0u;

  // Software pipeline over 64-entry chunks of the tile list: while chunk c is evaluated, the records of chunk
  // c+1 (gathered one per lane, compacted order) and the list entries of chunk c+2 are in flight.
  uint32_t v_nxt, v_nn = 0u;
  Compacted cc;
  float4 r0, r1; float rb;
  {
    const uint32_t i0 = range.x + lane, i1 = range.x + 64 + lane;
    const uint32_t v0 = i0 < range.y ? point_list[i0] : 0u;
    v_nxt = i1 < range.y ? point_list[i1] : 0u;
    cc = compact_chunk<false>(v0, qbit, lane);
    r0 = make_float4(0.f, 0.f, 0.f, 0.f); r1 = r0; rb = 0.f;
    if (lane < cc.n) {
      const float4* p = reinterpret_cast<const float4*>(rec + (size_t)(cc.val & ID_MASK) * REC_DWORDS);
      r0 = p[0]; r1 = p[1]; rb = reinterpret_cast<const float*>(p)[8];
    }
  }
  bool alive = true;  // wave-uniform: some pixel of this 8x8 block still accumulates
  for (uint32_t base = range.x; base < range.y && alive; base += 64) {
    // stage chunk c's records for broadcast reads
    const int n = __builtin_amdgcn_readfirstlane(cc.n);
    // The entries are evaluated four at a time (records at immediate LDS offsets, the loop counter scalar); the last group is
    // padded with null records: opacity 0 gives alpha 0, which is not a contribution and changes nothing.
    const int n4 = (n + 3) & ~3;
    if (lane < n4) {
      const bool real = lane < n;
      s_rec[wv][lane][0] = real ? r0 : make_float4(0.f, 0.f, 0.f, 0.f);
      s_rec[wv][lane][1] = real ? r1 : make_float4(0.f, 0.f, 0.f, 0.f);
      s_rec[wv][lane][2] = make_float4(real ? rb : 0.f, __uint_as_float(base - range.x + cc.pos + 1u), 0.f, 0.f);
    }
    wave_lds_fence();
    // put chunk c+1's records and chunk c+2's list entries in flight
    {
      const uint32_t i2 = base + 128 + lane;
      v_nn = i2 < range.y ? point_list[i2] : 0u;
      cc = compact_chunk<false>(v_nxt, qbit, lane);
      if (lane < cc.n) {
        const float4* p = reinterpret_cast<const float4*>(rec + (size_t)(cc.val & ID_MASK) * REC_DWORDS);
        r0 = p[0]; r1 = p[1]; rb = reinterpret_cast<const float*>(p)[8];
      }
      v_nxt = v_nn;
    }
    for (int k0 = 0; k0 < n4 && alive; k0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + u;
      const float4 q0 = s_rec[wv][k][0], q1 = s_rec[wv][k][1];
      const float2 q2 = *reinterpret_cast<const float2*>(&s_rec[wv][k][2]);
      const float dx = q0.x - pxf, dy = q0.y - pyf;
      const float power2 = dx * (q0.z * dx + q0.w * dy) + (q1.x * dy) * dy;  // log2e * power
      const float alpha = fminf(0.99f, q1.y * fast_exp2(power2));
      const bool ok = power2 <= 0.0f && alpha >= 1.0f / 255.0f;
      const float ae = ok ? alpha : 0.f;
      float w = ae * T;
      float test_T = T - w;
      const bool stop = test_T < thr;  // forward.cu:420-425; never true for dead pixels (thr = -1)
      if (__ballot(stop) != 0ull) {    // rare: some pixel saturates at this Gaussian
        Tfin = stop ? T : Tfin;
        lastfin = stop ? last : lastfin;   // `last` before this Gaussian: a dead pixel reports lastfin, so the
        thr = stop ? -1.f : thr;           // unconditional update of `last` below is harmless for it
        w = stop ? 0.f : w;
        test_T = stop ? 0.f : test_T;
        alive = __ballot(thr > 0.f) != 0ull;
      }
      T = test_T;
      C0 += q1.z * w; C1 += q1.w * w; C2 += q2.x * w;
      last = ok ? __float_as_uint(q2.y) : last;
      if (!alive) break;
    }
    }
  }
  if (inside) {
    const bool dead = thr < 0.f;
    const size_t pix_id = (size_t)W * py + px;
    const size_t HW = (size_t)H * W;
    const float Tout = dead ? Tfin : T;
    final_T[pix_id] = Tout;
    n_contrib[pix_id] = dead ? lastfin : last;
    out_color[pix_id] = C0 + Tout * bg[0];
    out_color[HW + pix_id] = C1 + Tout * bg[1];
    out_color[2 * HW + pix_id] = C2 + Tout * bg[2];
  }
}

// ---- cross-lane helpers -----------------------------------------------------------------------
// lanes 0-31 <- a[l] + a[l+32], lanes 32-63 <- b[l-32] + b[l]
__device__ __forceinline__ float fold32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows (16 lanes) [x0,x1,x2,x3],[y0..y3] -> [x0+x1, y0+y1, x2+x3, y2+y3]
__device__ __forceinline__ float fold16(float x, float y) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// a,b,c,d each hold, in lane (row, s), a partial for slot s.  Result rows: row0 = sum_rows(a), row1 = sum_rows(c),
// row2 = sum_rows(b), row3 = sum_rows(d), per slot s.
__device__ __forceinline__ float fold_rows4(float a, float b, float c, float d) {
  return fold16(fold32(a, b), fold32(c, d));
}

// Backward tile kernel.  Two roles alternate inside each wave, 16 list entries ("slots") at a time:
//  (1) pixel role (lane = pixel of the 8x8 quadrant): walk the 16 Gaussians back to front, update the pixel's
//      transmittance / behind-colour state and produce just TWO numbers per (pixel, Gaussian):
//         w  = dL/dG * G          (everything geometric is linear in it)
//         aT = alpha * T          (weight of dL/dcolour)
//      which go to LDS as two 16x64 tiles;
//  (2) Gaussian role (lane = (slot, 16-pixel part)): read the tiles transposed and accumulate the nine sums
//      S0=Sum w, S1x=Sum w px, S1y=Sum w py, Sxx, Sxy, Syy (px,py = pixel coords local to the quadrant) and
//      Sum aT*dL/dpixel[rgb] with plain FMAs -- no cross-lane reduction per pair.  Four row partials per slot are
//      folded with permlane swaps, shifted from quadrant-local to Gaussian-relative moments, and leave as one
//      float-atomic wave instruction per 4 Gaussians (36 contiguous bytes per Gaussian).
// The accumulated row holds raw moments (Mx, My, Mxx, Mxy, Myy, S0 / o, Sr, Sg, Sb); preprocess_bwd_kernel turns
// them into the reference's dL/dmean2D and dL/dconic (backward.cu:541-554) with the per-Gaussian conic; S0 / o IS dL/dopacity.
// acc += a * (value of `v` in lane I of this lane's 16-lane row): the DPP row broadcast rides on the FMA itself, so the
// Gaussian role gets dL/dpixel of pixel 16*part + I straight from the pixel lanes' registers -- no LDS read.
template <int I>
__device__ __forceinline__ void fmac_row_bcast(float& acc, float v, float a) {
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(v), "v"(a), "n"(I));
}
constexpr int BW_SLOTS = 16;
constexpr int BW_STRIDE = 65;  // padded row of the transposed tiles: conflict-free in both roles
struct BwdLds {                  // 9088 B per wave -> 16 single-wave workgroups per CU
  float4 rec[BW_SLOTS][3];      // staged records of the current batch: [0] x y a2 b2  [1] c2 o r g  [2] b pos id -
  float2 wa[BW_SLOTS][BW_STRIDE];   // (w, alpha T) per (slot, pixel): ONE 8-byte LDS write per pair in the pixel role, one 8-byte read
                                    // per pair in the Gaussian role (two 4-byte tiles before: 7 -> 5 LDS instructions per pair);
                                    // its first 16x12 floats are reused as the moment exchange area `mom`
};


// Pixel role of the backward kernel over one batch of `nb` staged records (see render_bwd_kernel).  USE_BG = false
// drops the background term of dL/dalpha (bg . dL/dpixel == 0 for every pixel of the wave: SLAM renders on black).
template <bool USE_BG>
__device__ __forceinline__ void pixel_role(BwdLds& L, int nb, int lane, float pxf, float pyf, uint32_t last_contributor,
                                           float dp0, float dp1, float dp2, float T_final, float bg_dot_dpixel, float& T,
                                           float& accd) {
#ifndef PIX_UNROLL
#define PIX_UNROLL 4
#endif
#pragma unroll PIX_UNROLL
  for (int sl = 0; sl < nb; sl++) {
    const float4 q0 = L.rec[sl][0], q1 = L.rec[sl][1];
    const float2 q2 = *reinterpret_cast<const float2*>(&L.rec[sl][2]);
    const float dx = q0.x - pxf, dy = q0.y - pyf;
    const float power2 = dx * (q0.z * dx + q0.w * dy) + (q1.x * dy) * dy;
#if defined(SEGS_MEASURE) && defined(ABLATE_NO_TRANS)
    const float ar = q1.y * (power2 * 0.001f + 1.0f);
#else
    const float ar = q1.y * fast_exp2(power2);  // o * G
#endif
    // alpha = min(0.99, ar) >= 1/255  <=>  ar >= 1/255
    const bool ok = __float_as_uint(q2.y) < last_contributor && power2 <= 0.0f && ar >= 1.0f / 255.0f;
    const float aw = ok ? ar : 0.f;      // o * G (the clamp at 0.99 is not differentiated, backward.cu:497); 0 on skipped pairs
    const float ae = fminf(0.99f, aw);   // alpha; 0 makes every update below a no-op
#if defined(SEGS_MEASURE) && defined(ABLATE_NO_TRANS)
    const float rinv = 1.f + ae;
#else
    const float rinv = fast_rcp(1.f - ae);
#endif
    T = T * rinv;                        // T / (1 - alpha)
    // The colour accumulated behind this Gaussian enters only through its dot product with dL/dpixel, so ONE running
    // scalar accd = sum_ch accum_ch * g_ch replaces the three accumulators of backward.cu:513-516:
    //   sum_ch (c_ch - accum_ch) g_ch = c.g - accd;   accum' = accum + alpha (c - accum)  =>  accd' = accd + alpha (c.g - accd)
    const float diff = (q1.z * dp0 + q1.w * dp1 + q2.x * dp2) - accd;
    float dL_dalpha = diff * T;
    if (USE_BG) dL_dalpha += (-T_final * rinv) * bg_dot_dpixel;
    accd += ae * diff;
    L.wa[sl][lane] = make_float2(aw * dL_dalpha /* = dL_dG * G */, ae * T /* = dchannel_dcolor */);
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool MFMA>
__device__ __forceinline__ void render_bwd_body(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
    const float* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix, float* __restrict__ gacc, uint32_t num_tiles) {
  // One wave (one 8x8 quadrant) per workgroup: the four quadrants of a tile share nothing but their inputs, and as one
  // 256-thread workgroup the three shorter ones held their LDS and wave slots until the longest list ended (the longest
  // quadrant of a tile is 1.15-1.17 x the mean, tools/tile_stats.py).  Workgroup b = ((tl * 4 + q) << 3) | xcd: the four
  // quadrants of a tile stay on one XCD, tiles are dealt to the XCDs as before (xcd_tile).
  __shared__ BwdLds lds_one;
  const uint32_t tiles_x = (W + TILE_X - 1) / TILE_X;
  const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
  const uint32_t v = ((local >> 2) << 3) | xcd;   // the tile-level workgroup id of the 256-thread formulation
  if (v >= num_tiles) return;
  const uint32_t tile = xcd_tile(v, num_tiles);
  const uint32_t tile_x = tile % tiles_x, tile_y = tile / tiles_x;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(local & 3u));
  BwdLds& L = lds_one;
  const uint32_t qx0 = tile_x * TILE_X + (wv & 1) * 8, qy0 = tile_y * TILE_Y + (wv >> 1) * 8;
  const uint32_t px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
  const bool inside = px < (uint32_t)W && py < (uint32_t)H;
  const float pxf = (float)px, pyf = (float)py;
  const size_t pix_id = (size_t)W * py + px;
  const size_t HW = (size_t)H * W;
  const uint32_t qbit = 1u << (ID_BITS + wv);

  const uint2 range = ranges[tile];
  const float T_final = inside ? final_T[pix_id] : 0.f;
  const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0u;
  float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f;
  if (inside) { dp0 = dL_dpix[pix_id]; dp1 = dL_dpix[HW + pix_id]; dp2 = dL_dpix[2 * HW + pix_id]; }
  const float bg_dot_dpixel = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;
  const bool use_bg = __ballot(bg_dot_dpixel != 0.f) != 0ull;   // wave-uniform

  // Start at the deepest contributor of this 8x8 block (backward.cu:487-488).
  uint32_t wave_last = last_contributor;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, off, 64));
  wave_last = __builtin_amdgcn_readfirstlane(wave_last);
  if (wave_last == 0u) return;

  float T = T_final, accd = 0.f;  // accd = (colour accumulated BEHIND the current Gaussian) . dL/dpixel

  // Gaussian-role constants
  const int gs = lane & 15, part = lane >> 4;
  const float pyl0 = (float)(part * 2), pyl1 = pyl0 + 1.0f;
  const float qx0f = (float)qx0, qy0f = (float)qy0;

  // Matrix-pipe form of the Gaussian role (MFMA = true): the batch's nine sums per slot are ONE product
  //   D[16 slots x 16] = [W | A](16 x 128) . [[F1, 0], [0, F2]](128 x 16),   W, A = the (w, alpha T) tiles in LDS,
  // F1[pixel] = (1, x, y, x^2, x y, y^2) about the quadrant centre (3.5, 3.5), F2[pixel] = dL/dpixel rgb -- both the same for
  // every batch of the wave, so the 2 x 16 B operands of v_mfma_f32_16x16x4_f32 are built once.  Step t contracts the four
  // pixels 16 g + t (g = lane / 16 = the instruction's k index): lane (j, g) holds column j of F at that pixel.
  float bmono[16], bcol[16];
  if (MFMA) {
    float* dps = reinterpret_cast<float*>(&L.wa[0][0]);   // [3][64] staging of dL/dpixel, free before the first batch
    dps[lane] = dp0; dps[64 + lane] = dp1; dps[128 + lane] = dp2;
    wave_lds_fence();
    const float j1 = gs == 1 ? 1.f : 0.f, j3 = gs == 3 ? 1.f : 0.f, j4 = gs == 4 ? 1.f : 0.f;
    const bool colour = gs >= 6 && gs < 9;
    const int crow = colour ? (gs - 6) * 64 + part * 16 : 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const float y = (float)(2 * part + h) - 3.5f;
      // column j at (x, y): c0 + x (c1 + x c2);  j: 0 -> 1, 1 -> x, 2 -> y, 3 -> x^2, 4 -> x y, 5 -> y^2, others 0
      const float c0 = gs == 0 ? 1.f : gs == 2 ? y : gs == 5 ? y * y : 0.f;
      const float c1 = j1 + j4 * y;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const float x = (float)c - 3.5f;
        bmono[8 * h + c] = c0 + x * (c1 + x * j3);
      }
    }
#pragma unroll
    for (int t = 0; t < 16; t++) { const float v = dps[crow + t]; bcol[t] = colour ? v : 0.f; }
    wave_lds_fence();
  }

  const int32_t cfirst = (int32_t)((wave_last - 1u) & ~63u);
  uint32_t v_nxt = 0u;
  Compacted cc;
  float4 r0, r1; float rb;
  {
    const uint32_t p0 = (uint32_t)cfirst + lane;
    const uint32_t v0 = p0 < wave_last ? point_list[range.x + p0] : 0u;
    if (cfirst >= 64) v_nxt = point_list[range.x + (uint32_t)(cfirst - 64) + lane];
    cc = compact_chunk<true>(v0, qbit, lane);
    r0 = make_float4(0.f, 0.f, 0.f, 0.f); r1 = r0; rb = 0.f;
    if (lane < cc.n) {
      const float4* p = reinterpret_cast<const float4*>(rec + (size_t)(cc.val & ID_MASK) * REC_DWORDS);
      r0 = p[0]; r1 = p[1]; rb = reinterpret_cast<const float*>(p)[8];
    }
  }
  // Batches of BW_SLOTS entries are cut from the STREAM of compacted entries, not from each 64-entry chunk: what a chunk
  // leaves over (fewer than BW_SLOTS records) stays staged in L.rec and the next chunk tops it up, so both roles always
  // run on full batches (at 3 M Gaussians a chunk holds ~18 entries of this quadrant: per-chunk batches were 58 % full).
  // One more pass of the loop after the last chunk (tail) flushes the final partial batch.
  float (*mom)[12] = reinterpret_cast<float (*)[12]>(&L.wa[0][0]);
  int fill = 0;   // records staged in L.rec, wave-uniform
  for (int32_t cbase = cfirst;; cbase -= 64) {
    const bool tail = cbase < 0;
    const int n = tail ? 0 : __builtin_amdgcn_readfirstlane(cc.n);
    // current chunk's records stay in registers (one per lane, compacted order); the next chunk's go in flight
    const float4 c0 = r0, c1 = r1;
    const float4 c2 = make_float4(rb, __uint_as_float((uint32_t)cbase + cc.pos), __uint_as_float(cc.val & ID_MASK), 0.f);
    if (cbase >= 64) {
      uint32_t v_nn = 0u;
      if (cbase >= 128) v_nn = point_list[range.x + (uint32_t)(cbase - 128) + lane];
      cc = compact_chunk<true>(v_nxt, qbit, lane);
      if (lane < cc.n) {
        const float4* p = reinterpret_cast<const float4*>(rec + (size_t)(cc.val & ID_MASK) * REC_DWORDS);
        r0 = p[0]; r1 = p[1]; rb = reinterpret_cast<const float*>(p)[8];
      }
      v_nxt = v_nn;
    }
    int taken = 0;
    do {
      const int take = min(BW_SLOTS - fill, n - taken);
      if (lane >= taken && lane < taken + take) {
        const int s = fill + lane - taken;
        L.rec[s][0] = c0; L.rec[s][1] = c1; L.rec[s][2] = c2;
      }
      fill += take; taken += take;
      if (fill < BW_SLOTS && !(tail && fill > 0)) continue;   // batch not full yet: the next chunk tops it up
      const int nb = fill;
      fill = 0;
      wave_lds_fence();
      // ---------------- (1) pixel role
      if (use_bg) pixel_role<true>(L, nb, lane, pxf, pyf, last_contributor, dp0, dp1, dp2, T_final, bg_dot_dpixel, T, accd);
      else pixel_role<false>(L, nb, lane, pxf, pyf, last_contributor, dp0, dp1, dp2, T_final, bg_dot_dpixel, T, accd);
      wave_lds_fence();
#if defined(SEGS_MEASURE) && defined(ABLATE_NO_GAUSS_ROLE)
      continue;   // (the do-while's condition is evaluated)
#endif
      // ---------------- (2) Gaussian role: lane = (part, gs): slot gs, pixels part*16 .. part*16+15
      const float2 gxy = *reinterpret_cast<const float2*>(&L.rec[gs][0]);
      const float gxr = gxy.x - qx0f, gyr = gxy.y - qy0f;
      float cx, cy;
      if (MFMA) {
        // One dependent chain per operand tile (the instruction's dependent latency is 40 cycles against 32 of issue: two
        // independent accumulators keep the pipe full).  A operand: lane (gs, part) reads slot gs, pixel 16 part + t.
        cx = 3.5f; cy = 3.5f;
        f32x4 dw = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
        const float2* warow = &L.wa[gs][part * 16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
          const float2 wa_t = warow[t];
#ifdef SEGS_MFMA_ONE_CHAIN
          dw = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_t.x, bmono[t], dw, 0, 0, 0);
          dw = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_t.y, bcol[t], dw, 0, 0, 0);
#else
          dw = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_t.x, bmono[t], dw, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_t.y, bcol[t], da, 0, 0, 0);
#endif
        }
        // D: lane (j = gs, g = part) holds column j of slots 4 g .. 4 g + 3; columns 0-5 come from dw, 6-8 from da (the other
        // tile's columns are exact zeros there)
        if (gs < 9) {
#pragma unroll
          for (int r = 0; r < 4; r++) mom[4 * part + r][gs] = dw[r] + da[r];
        }
      } else {
      // Moments are taken about the quadrant pixel NEAREST to the Gaussian's centre (cx, cy in 0..7), not about the
      // quadrant corner: for a centre inside the quadrant |gxr - cx| <= 0.5, so the later shift to centre-relative
      // moments (Mxx = g'^2 S0 - 2 g' S1x + Sxx) subtracts nothing large even for sub-pixel Gaussians.
      cx = fminf(7.f, fmaxf(0.f, rintf(gxr))); cy = fminf(7.f, fmaxf(0.f, rintf(gyr)));
      // The six w-moments are separable over the lane's two pixel rows (8 pixels each): per pixel only the row sums
      // R0 = sum w, R1 = sum w x, R2 = sum w x^2 (3 ops), per row six more; the colour sums need dL/dpixel per pixel.
      float px8[8], px8q[8];
#pragma unroll
      for (int c = 0; c < 8; c++) { px8[c] = (float)c - cx; px8q[c] = px8[c] * px8[c]; }
      const float pyc0 = pyl0 - cy, pyc1 = pyl1 - cy;
      float S0, S1x, S1y, Sxx, Sxy, Syy, Sr = 0.f, Sg = 0.f, Sb = 0.f;
      {
        float R0[2] = {0.f, 0.f}, R1[2] = {0.f, 0.f}, R2[2] = {0.f, 0.f};
        const float2* warow = &L.wa[gs][part * 16];
#define GAUSS_STEP(i)                                                                         \
        {                                                                                     \
          const float2 wa_i = warow[i];                                                       \
          const float w = wa_i.x, a = wa_i.y;                                                 \
          R0[(i) >> 3] += w; R1[(i) >> 3] += w * px8[(i) & 7]; R2[(i) >> 3] += w * px8q[(i) & 7]; \
          fmac_row_bcast<(i)>(Sr, dp0, a); fmac_row_bcast<(i)>(Sg, dp1, a); fmac_row_bcast<(i)>(Sb, dp2, a); \
        }
        GAUSS_STEP(0) GAUSS_STEP(1) GAUSS_STEP(2) GAUSS_STEP(3) GAUSS_STEP(4) GAUSS_STEP(5) GAUSS_STEP(6) GAUSS_STEP(7)
        GAUSS_STEP(8) GAUSS_STEP(9) GAUSS_STEP(10) GAUSS_STEP(11) GAUSS_STEP(12) GAUSS_STEP(13) GAUSS_STEP(14) GAUSS_STEP(15)
#undef GAUSS_STEP
        S0 = R0[0] + R0[1]; S1x = R1[0] + R1[1]; Sxx = R2[0] + R2[1];
        S1y = pyc0 * R0[0] + pyc1 * R0[1];
        Sxy = pyc0 * R1[0] + pyc1 * R1[1];
        Syy = (pyc0 * pyc0) * R0[0] + (pyc1 * pyc1) * R0[1];
      }
      const float g1 = fold_rows4(S0, S1y, S1x, Sxx);   // rows: S0, S1x, S1y, Sxx
      const float g2 = fold_rows4(Sxy, Sr, Syy, Sg);    // rows: Sxy, Syy, Sr, Sg
      const float g3 = fold_rows4(Sb, Sb, Sb, Sb);      // every row: Sb total
      mom[gs][part] = g1;
      mom[gs][4 + part] = g2;
      if (part == 0) mom[gs][8] = g3;
      }
      wave_lds_fence();
      {
        // every lane of column gs shifts slot gs from quadrant-local moments to Gaussian-relative ones
        const float4 m0 = *reinterpret_cast<const float4*>(&mom[gs][0]);  // S0 S1x S1y Sxx
        const float4 m1 = *reinterpret_cast<const float4*>(&mom[gs][4]);  // Sxy Syy Sr Sg
        const float m8 = mom[gs][8];
        const float hx = gxr - cx, hy = gyr - cy;   // dx = hx - pxl', dy = hy - pyl'  (pxl', pyl' relative to (cx, cy))
        const float Mx = hx * m0.x - m0.y, My = hy * m0.x - m0.z;
        const float Mxx = hx * (hx * m0.x - 2.f * m0.y) + m0.w;
        const float Mxy = hx * (hy * m0.x - m0.z) - hy * m0.y + m1.x;
        const float Myy = hy * (hy * m0.x - 2.f * m0.z) + m1.y;
        wave_lds_fence();
        // S0 = sum of w = o * sum of G dL/dalpha leaves as dL/dopacity = S0 / o (backward.cu:554): the per-Gaussian backward
        // then needs no opacity -- and with its conic recomputed there, no record gather at all
        const float inv_o = fast_rcp(reinterpret_cast<const float*>(&L.rec[gs][1])[1]);
        wave_lds_fence();
        if (part == 0) {
          *reinterpret_cast<float4*>(&mom[gs][0]) = make_float4(Mx, My, Mxx, Mxy);
          *reinterpret_cast<float4*>(&mom[gs][4]) = make_float4(Myy, m0.x * inv_o, m1.z, m1.w);
          mom[gs][8] = m8;
        }
      }
      wave_lds_fence();
      // one atomic wave instruction per 4 slots: lane (row, col<9) adds element col of slot 4j + row
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int slot = 4 * j + part;
        if (slot < nb && gs < 9) {
          const uint32_t id = __float_as_uint(reinterpret_cast<const float*>(&L.rec[slot][2])[2]);
#if !(defined(SEGS_MEASURE) && defined(ABLATE_NO_ATOMICS))
          atomicAdd(gacc + (size_t)id * GACC_DWORDS + gs, mom[slot][gs]);
#endif
        }
      }
      wave_lds_fence();
    } while (taken < n);
    if (tail) break;
  }
}

__global__ void __launch_bounds__(64) render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
    const float* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix, float* __restrict__ gacc, uint32_t num_tiles) {
  render_bwd_body<false>(ranges, point_list, W, H, rec, bg, final_T, n_contrib, dL_dpix, gacc, num_tiles);
}
// The Gaussian role's sums on the matrix pipe: the measured A/B partner (SEGS_RENDER_BWD_MFMA=1, capi.hip).
__global__ void __launch_bounds__(64) render_bwd_mfma_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
    const float* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix, float* __restrict__ gacc, uint32_t num_tiles) {
  render_bwd_body<true>(ranges, point_list, W, H, rec, bg, final_T, n_contrib, dL_dpix, gacc, num_tiles);
}

}  // namespace segs
