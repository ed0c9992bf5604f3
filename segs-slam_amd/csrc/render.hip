// render.hip -- per-tile alpha compositing, forward and backward (gfx950 / MI355X).
//
//   render_fwd_kernel  K10  reference: renderCUDA forward   cuda_rasterizer/forward.cu:339-452
//   render_bwd_kernel  K11  reference: renderCUDA backward  cuda_rasterizer/backward.cu:399-557
//
// These two kernels are VALU-bound, not HBM-bound (~25 / ~80 vector ops per (pixel, Gaussian) pair
// against 64 B of record per 64 pairs).  MI355X mapping:
//  * workgroup = one 16x16 tile (the binning unit, fixed by the reference's key format) = 4 wave64;
//    each wave owns an 8x8 pixel quadrant and walks the tile's list on its own: no LDS staging, no
//    workgroup barrier.
//  * the list is read 64 entries at a time with one coalesced vector load; each entry carries a 4-bit
//    "quadrants this instance can touch" mask (gs_layout.h), so one ballot yields the 64-bit set of entries
//    this wave must evaluate and everything else is skipped with scalar bit-scans (s_ff1 / s_flbit).
//  * the per-Gaussian operands are wave-uniform: the 64-byte record is fetched with scalar (SMEM) loads
//    straight into SGPRs and used as the scalar operand of the VALU instructions -- no LDS broadcast
//    reads (which would be LDS-issue bound at 4 waves per CU-cycle).
//  * early termination is per wave (8x8 block) instead of per 16x16 tile.
//  * exp(power) = exp2(log2e * power) with log2e folded into the stored conic (v_exp_f32).
//  * backward: the 9 partial sums of FOUR Gaussians are reduced together: v_permlane32_swap and
//    v_permlane16_swap fold 4x(64 lanes) into 4 rows of 16 lanes in 5 adds, four DPP row steps finish all
//    four at once (10 cross-lane ops per quantity per 4 Gaussians instead of 24), and the 4x9 sums leave as
//    ONE float-atomic wave instruction (one 36-byte segment of a 64-byte gradient row per Gaussian) instead
//    of the reference's 9 atomics per (pixel, Gaussian) pair.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gs_layout.h"
#include "kernels.h"

namespace segs {

struct RecS { float x, y, a2, b2, c2, o, r, g, b, ca, cb, cc; };

__device__ __forceinline__ RecS load_rec(const float* __restrict__ rec, uint32_t id) {
  const float4* p = reinterpret_cast<const float4*>(rec + (size_t)id * REC_DWORDS);
  const float4 q0 = p[0], q1 = p[1], q2 = p[2];
  RecS r;
  r.x = q0.x; r.y = q0.y; r.a2 = q0.z; r.b2 = q0.w;
  r.c2 = q1.x; r.o = q1.y; r.r = q1.z; r.g = q1.w;
  r.b = q2.x; r.ca = q2.y; r.cb = q2.z; r.cc = q2.w;
  return r;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}


// Per-lane copy of one record (the lane's own list entry); q2.x = blue, q2.yzw = conic (backward only).
struct LaneRec { float4 q0, q1, q2; };
__device__ __forceinline__ LaneRec load_lane_rec_fwd(const float* __restrict__ rec, uint32_t v, uint32_t qbit) {
  LaneRec r;
  r.q0 = make_float4(0.f, 0.f, 0.f, 0.f); r.q1 = r.q0; r.q2 = r.q0;
  if ((v & qbit) != 0u) {
    const float4* p = reinterpret_cast<const float4*>(rec + (size_t)(v & ID_MASK) * REC_DWORDS);
    r.q0 = p[0]; r.q1 = p[1]; r.q2.x = reinterpret_cast<const float*>(p)[8];
  }
  return r;
}
__device__ __forceinline__ LaneRec load_lane_rec_bwd(const float* __restrict__ rec, uint32_t v, uint32_t qbit) {
  LaneRec r;
  r.q0 = make_float4(0.f, 0.f, 0.f, 0.f); r.q1 = r.q0; r.q2 = r.q0;
  if ((v & qbit) != 0u) {
    const float4* p = reinterpret_cast<const float4*>(rec + (size_t)(v & ID_MASK) * REC_DWORDS);
    r.q0 = p[0]; r.q1 = p[1]; r.q2 = p[2];
  }
  return r;
}
__device__ __forceinline__ float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__global__ void __launch_bounds__(256) render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
    const float* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color) {
  const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t px = blockIdx.x * TILE_X + (wv & 1) * 8 + (lane & 7);
  const uint32_t py = blockIdx.y * TILE_Y + (wv >> 1) * 8 + (lane >> 3);
  const bool inside = px < (uint32_t)W && py < (uint32_t)H;
  const float pxf = (float)px, pyf = (float)py;
  const uint32_t qbit = 1u << (ID_BITS + wv);

  const uint2 range = ranges[tile];
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t last_contributor = 0;
  bool done = !inside;

  // Software pipeline over 64-entry chunks of the tile list: while chunk c is evaluated, the records of
  // chunk c+1 and the list entries of chunk c+2 are in flight (vector loads, counted vmcnt waits).
  // Lane l fetches the record of list entry (base + l) only if this wave's quadrant bit is set in it;
  // the evaluation loop then broadcasts one lane's record at a time into SGPRs (v_readlane).
  LaneRec cur, nxt;
  uint32_t v_cur = 0u, v_nxt = 0u, v_nn = 0u;
  {
    const uint32_t i0 = range.x + lane, i1 = range.x + 64 + lane;
    v_cur = i0 < range.y ? point_list[i0] : 0u;
    v_nxt = i1 < range.y ? point_list[i1] : 0u;
    cur = load_lane_rec_fwd(rec, v_cur, qbit);
  }
  for (uint32_t base = range.x; base < range.y; base += 64) {
    if (__ballot(!done) == 0ull) break;  // whole 8x8 block finished (forward.cu:386-389, per wave)
    nxt = load_lane_rec_fwd(rec, v_nxt, qbit);
    {
      const uint32_t i2 = base + 128 + lane;
      v_nn = i2 < range.y ? point_list[i2] : 0u;
    }
    uint64_t m = __ballot((v_cur & qbit) != 0u);
    while (m != 0ull) {
      const int j = __builtin_ctzll(m);
      m &= m - 1ull;
      const float gx = rl(cur.q0.x, j), gy = rl(cur.q0.y, j), a2 = rl(cur.q0.z, j), b2 = rl(cur.q0.w, j);
      const float c2 = rl(cur.q1.x, j), go = rl(cur.q1.y, j), cr = rl(cur.q1.z, j), cg = rl(cur.q1.w, j), cb = rl(cur.q2.x, j);
      const float dx = gx - pxf, dy = gy - pyf;
      const float power2 = dx * (a2 * dx + b2 * dy) + (c2 * dy) * dy;  // log2e * power
      const float alpha = fminf(0.99f, go * fast_exp2(power2));
      const bool ok = !done && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
      if (__ballot(ok) != 0ull) {
        const float test_T = T * (1.f - alpha);
        const bool stop = ok && test_T < 0.0001f;
        const bool upd = ok && !stop;
        const float w = upd ? alpha * T : 0.f;
        C0 += cr * w; C1 += cg * w; C2 += cb * w;
        T = upd ? test_T : T;
        last_contributor = upd ? (base - range.x + (uint32_t)j + 1u) : last_contributor;
        done = done || stop;
        if (__ballot(!done) == 0ull) break;
      }
    }
    cur = nxt; v_cur = v_nxt; v_nxt = v_nn;
  }
  if (inside) {
    const size_t pix_id = (size_t)W * py + px;
    const size_t HW = (size_t)H * W;
    final_T[pix_id] = T;
    n_contrib[pix_id] = last_contributor;
    out_color[pix_id] = C0 + T * bg[0];
    out_color[HW + pix_id] = C1 + T * bg[1];
    out_color[2 * HW + pix_id] = C2 + T * bg[2];
  }
}

// ---- cross-lane helpers -----------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
  return v + __int_as_float(moved);
}
// every lane of each 16-lane row ends with the sum over its row
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return v;
}
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
// Sum each of a,b,c,d over the 64 lanes; result rows hold: row0 = sum(a), row1 = sum(c), row2 = sum(b), row3 = sum(d).
__device__ __forceinline__ float reduce4(float a, float b, float c, float d) {
  return row16_sum(fold16(fold32(a, b), fold32(c, d)));
}

struct PixelState {
  float T, acc0, acc1, acc2, lc0, lc1, lc2, last_alpha;
};

// One (pixel, Gaussian) backward step (backward.cu:489-555).  Writes the 9 per-lane partials (zeros for lanes
// that skip) into v[0..8].
__device__ __forceinline__ void bwd_pair(const RecS& g, uint32_t pos, uint32_t last_contributor, float pxf, float pyf,
                                         float dp0, float dp1, float dp2, float T_final, float bg_dot_dpixel,
                                         float ddelx_dx, float ddely_dy, PixelState& s, float* v) {
  const float dx = g.x - pxf, dy = g.y - pyf;
  const float power2 = dx * (g.a2 * dx + g.b2 * dy) + (g.c2 * dy) * dy;
  const float Graw = fast_exp2(power2);
  const float alpha = fminf(0.99f, g.o * Graw);
  const bool ok = pos < last_contributor && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
  const float G = ok ? Graw : 0.f;  // lanes that skip contribute exact zeros (Graw may be inf)
  const float one_m_alpha_inv = fast_rcp(1.f - alpha);
  const float Tn = s.T * one_m_alpha_inv;
  s.T = ok ? Tn : s.T;
  const float dchannel_dcolor = ok ? alpha * s.T : 0.f;
  // accum_rec / last_color recursion (backward.cu:513-523)
  const float a0 = s.last_alpha * s.lc0 + (1.f - s.last_alpha) * s.acc0;
  const float a1 = s.last_alpha * s.lc1 + (1.f - s.last_alpha) * s.acc1;
  const float a2 = s.last_alpha * s.lc2 + (1.f - s.last_alpha) * s.acc2;
  s.acc0 = ok ? a0 : s.acc0; s.acc1 = ok ? a1 : s.acc1; s.acc2 = ok ? a2 : s.acc2;
  s.lc0 = ok ? g.r : s.lc0; s.lc1 = ok ? g.g : s.lc1; s.lc2 = ok ? g.b : s.lc2;
  float dL_dalpha = (g.r - a0) * dp0 + (g.g - a1) * dp1 + (g.b - a2) * dp2;
  dL_dalpha *= s.T;
  s.last_alpha = ok ? alpha : s.last_alpha;
  dL_dalpha += (-T_final * one_m_alpha_inv) * bg_dot_dpixel;
  dL_dalpha = ok ? dL_dalpha : 0.f;
  const float dL_dG = g.o * dL_dalpha;
  const float gdx = G * dx, gdy = G * dy;
  const float dG_ddelx = -gdx * g.ca - gdy * g.cb;
  const float dG_ddely = -gdy * g.cc - gdx * g.cb;
  v[0] = dL_dG * dG_ddelx * ddelx_dx;
  v[1] = dL_dG * dG_ddely * ddely_dy;
  v[2] = -0.5f * gdx * dx * dL_dG;
  v[3] = -0.5f * gdx * dy * dL_dG;
  v[4] = -0.5f * gdy * dy * dL_dG;
  v[5] = G * dL_dalpha;
  v[6] = dchannel_dcolor * dp0;
  v[7] = dchannel_dcolor * dp1;
  v[8] = dchannel_dcolor * dp2;
}

__global__ void __launch_bounds__(256) render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
    const float* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix, float* __restrict__ gacc) {
  const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t px = blockIdx.x * TILE_X + (wv & 1) * 8 + (lane & 7);
  const uint32_t py = blockIdx.y * TILE_Y + (wv >> 1) * 8 + (lane >> 3);
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
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
  PixelState s{T_final, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // Start at the deepest contributor of this 8x8 block: everything behind it is skipped by every pixel
  // (backward.cu:487-488).
  uint32_t wave_last = last_contributor;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, off, 64));
  wave_last = __builtin_amdgcn_readfirstlane(wave_last);
  if (wave_last == 0u) return;

  const int row = lane >> 4, col = lane & 15;
  const int32_t cfirst = (int32_t)((wave_last - 1u) & ~63u);
  uint32_t v_cur, v_nxt = 0u;
  {
    const uint32_t p0 = (uint32_t)cfirst + lane;
    v_cur = p0 < wave_last ? point_list[range.x + p0] : 0u;
    if (cfirst >= 64) v_nxt = point_list[range.x + (uint32_t)(cfirst - 64) + lane];
  }
  LaneRec cur = load_lane_rec_bwd(rec, v_cur, qbit);
  for (int32_t cbase = cfirst; cbase >= 0; cbase -= 64) {
    // records of the next (shallower) chunk and the list entries of the one after it go in flight now
    const LaneRec nxt = load_lane_rec_bwd(rec, v_nxt, qbit);
    uint32_t v_nn = 0u;
    if (cbase >= 128) v_nn = point_list[range.x + (uint32_t)(cbase - 128) + lane];
    const uint32_t v = v_cur;
    uint64_t m = __ballot((v & qbit) != 0u);
    while (m != 0ull) {
      // up to four list entries, back to front
      uint32_t ids[4]; uint32_t pos[4]; bool have[4]; int jj[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        have[k] = m != 0ull;
        const int j = have[k] ? 63 - __builtin_clzll(m) : 0;
        if (have[k]) m &= ~(1ull << j);
        jj[k] = j;
        ids[k] = readlane_u32(v, j) & ID_MASK;
        pos[k] = (uint32_t)cbase + (uint32_t)j;
      }
      float vq[4][9];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (have[k]) {  // wave-uniform
          RecS g;
          g.x = rl(cur.q0.x, jj[k]); g.y = rl(cur.q0.y, jj[k]); g.a2 = rl(cur.q0.z, jj[k]); g.b2 = rl(cur.q0.w, jj[k]);
          g.c2 = rl(cur.q1.x, jj[k]); g.o = rl(cur.q1.y, jj[k]); g.r = rl(cur.q1.z, jj[k]); g.g = rl(cur.q1.w, jj[k]);
          g.b = rl(cur.q2.x, jj[k]); g.ca = rl(cur.q2.y, jj[k]); g.cb = rl(cur.q2.z, jj[k]); g.cc = rl(cur.q2.w, jj[k]);
          bwd_pair(g, pos[k], last_contributor, pxf, pyf, dp0, dp1, dp2, T_final, bg_dot_dpixel, ddelx_dx, ddely_dy, s, vq[k]);
        } else {
#pragma unroll
          for (int q = 0; q < 9; q++) vq[k][q] = 0.f;
        }
      }
      // rows after reduce4: row0 <- k=0, row1 <- k=2, row2 <- k=1, row3 <- k=3
      float mine = 0.f;
#pragma unroll
      for (int q = 0; q < 9; q++) {
        const float z = reduce4(vq[0][q], vq[1][q], vq[2][q], vq[3][q]);
        mine = (col == q) ? z : mine;
      }
      const uint32_t my_id = row == 0 ? ids[0] : row == 1 ? ids[2] : row == 2 ? ids[1] : ids[3];
      const bool my_have = row == 0 ? have[0] : row == 1 ? have[2] : row == 2 ? have[1] : have[3];
      if (my_have && col < 9) atomicAdd(gacc + (size_t)my_id * GACC_DWORDS + col, mine);
    }
    cur = nxt; v_cur = v_nxt; v_nxt = v_nn;
  }
}

}  // namespace segs
