// render.hip -- per-tile alpha compositing, forward and backward (gfx950 / MI355X).
//
//   render_fwd_kernel  K10  reference: renderCUDA forward   cuda_rasterizer/forward.cu:339-452
//   render_bwd_kernel  K11  reference: renderCUDA backward  cuda_rasterizer/backward.cu:399-557
//
// These two kernels are VALU-bound, not HBM-bound (~25 / ~110 vector ops per (pixel, Gaussian) pair
// against 64 B of record per 64 pairs).  MI355X mapping:
//  * workgroup = one 16x16 tile (the binning unit, fixed by the reference's key format) = 4 wave64;
//    each wave owns an 8x8 pixel quadrant and walks the tile's list on its own: no LDS staging and no
//    workgroup barrier.  The per-Gaussian operands are wave-uniform, so they are fetched with scalar
//    (SMEM) loads of the 64-byte record straight into SGPRs and used as the scalar operand of the VALU
//    instructions -- the LDS broadcast reads of a warp-style port (2-3 ds_read per pair-iteration per
//    wave, LDS-issue bound at 4 waves/CU-cycle) disappear.
//  * early termination is per wave (ballot), i.e. per 8x8 block instead of per 16x16 tile; a wave also
//    skips a Gaussian outright when none of its 64 pixels passes the alpha test.
//  * exp(power) = exp2(log2e * power) with log2e folded into the stored conic (v_exp_f32).
//  * backward: the 9 per-Gaussian partial sums are reduced across the wave with DPP row operations
//    (no LDS), then issued as ONE packed float-atomic wave instruction (lanes 0..8 -> one 64-byte
//    gradient row), instead of the reference's 9 atomics per (pixel, Gaussian) pair.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gs_layout.h"
#include "kernels.h"

namespace segs {

constexpr int BATCH = 4;  // records fetched ahead per loop trip (scalar loads in flight)

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

  const uint2 range = ranges[tile];
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t last_contributor = 0;
  bool done = !inside;

  for (uint32_t s = range.x; s < range.y; s += BATCH) {
    if (__ballot(!done) == 0ull) break;  // whole 8x8 block finished (forward.cu:386-389, per wave)
    RecS g[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; k++) {
      const uint32_t sk = min(s + k, range.y - 1);  // clamped: always a valid, wave-uniform address
      g[k] = load_rec(rec, point_list[sk]);
    }
#pragma unroll
    for (int k = 0; k < BATCH; k++) {
      if (s + k < range.y) {  // wave-uniform
        const float dx = g[k].x - pxf, dy = g[k].y - pyf;
        const float power2 = dx * (g[k].a2 * dx + g[k].b2 * dy) + (g[k].c2 * dy) * dy;  // log2e * power
        const float alpha = fminf(0.99f, g[k].o * fast_exp2(power2));
        const bool ok = !done && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
        if (__ballot(ok) != 0ull) {
          const float test_T = T * (1.f - alpha);
          const bool stop = ok && test_T < 0.0001f;
          const bool upd = ok && !stop;
          const float w = upd ? alpha * T : 0.f;
          C0 += g[k].r * w; C1 += g[k].g * w; C2 += g[k].b * w;
          T = upd ? test_T : T;
          last_contributor = upd ? (s + k - range.x + 1) : last_contributor;
          done = done || stop;
        }
      }
    }
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

// ---- wave64 sum with DPP row operations; the total ends in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);  // row_mirror            -> every lane holds its 16-lane row sum
  v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave sum
  return v;
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

  const uint2 range = ranges[tile];
  const float T_final = inside ? final_T[pix_id] : 0.f;
  float T = T_final;
  const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0u;
  float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f;
  if (inside) { dp0 = dL_dpix[pix_id]; dp1 = dL_dpix[HW + pix_id]; dp2 = dL_dpix[2 * HW + pix_id]; }
  const float bg_dot_dpixel = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;  // accum_rec
  float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;     // last_color
  float last_alpha = 0.f;

  // Start at the deepest contributor of this 8x8 block: everything behind it is skipped by every pixel
  // (backward.cu:487-488).
  uint32_t wave_last = last_contributor;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, off, 64));
  wave_last = __builtin_amdgcn_readfirstlane(wave_last);

  for (int32_t top = (int32_t)wave_last - 1; top >= 0; top -= BATCH) {
    RecS g[BATCH];
    uint32_t ids[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; k++) {
      const int32_t pos = max(top - k, 0);
      ids[k] = point_list[range.x + pos];
      g[k] = load_rec(rec, ids[k]);
    }
#pragma unroll
    for (int k = 0; k < BATCH; k++) {
      const int32_t pos = top - k;  // 0-based position in the tile list
      if (pos >= 0) {               // wave-uniform
        const float dx = g[k].x - pxf, dy = g[k].y - pyf;
        const float power2 = dx * (g[k].a2 * dx + g[k].b2 * dy) + (g[k].c2 * dy) * dy;
        const float Graw = fast_exp2(power2);
        const float alpha = fminf(0.99f, g[k].o * Graw);
        const bool ok = (uint32_t)pos < last_contributor && power2 <= 0.0f && alpha >= 1.0f / 255.0f;
        if (__ballot(ok) != 0ull) {
          const float G = ok ? Graw : 0.f;  // lanes that skip contribute exact zeros (Graw may be inf)
          const float one_m_alpha_inv = fast_rcp(1.f - alpha);
          const float Tn = T * one_m_alpha_inv;
          T = ok ? Tn : T;
          const float dchannel_dcolor = ok ? alpha * T : 0.f;
          // accum_rec / last_color recursion (backward.cu:513-523)
          const float a0 = last_alpha * lc0 + (1.f - last_alpha) * acc0;
          const float a1 = last_alpha * lc1 + (1.f - last_alpha) * acc1;
          const float a2 = last_alpha * lc2 + (1.f - last_alpha) * acc2;
          acc0 = ok ? a0 : acc0; acc1 = ok ? a1 : acc1; acc2 = ok ? a2 : acc2;
          lc0 = ok ? g[k].r : lc0; lc1 = ok ? g[k].g : lc1; lc2 = ok ? g[k].b : lc2;
          float dL_dalpha = (g[k].r - a0) * dp0 + (g[k].g - a1) * dp1 + (g[k].b - a2) * dp2;
          dL_dalpha *= T;
          last_alpha = ok ? alpha : last_alpha;
          dL_dalpha += (-T_final * one_m_alpha_inv) * bg_dot_dpixel;
          dL_dalpha = ok ? dL_dalpha : 0.f;
          const float dL_dG = g[k].o * dL_dalpha;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * g[k].ca - gdy * g[k].cb;
          const float dG_ddely = -gdy * g[k].cc - gdx * g[k].cb;
          float v[9];
          v[0] = dL_dG * dG_ddelx * ddelx_dx;
          v[1] = dL_dG * dG_ddely * ddely_dy;
          v[2] = -0.5f * gdx * dx * dL_dG;
          v[3] = -0.5f * gdx * dy * dL_dG;
          v[4] = -0.5f * gdy * dy * dL_dG;
          v[5] = G * dL_dalpha;
          v[6] = dchannel_dcolor * dp0;
          v[7] = dchannel_dcolor * dp1;
          v[8] = dchannel_dcolor * dp2;
          float mine = 0.f;
#pragma unroll
          for (int q = 0; q < 9; q++) {
            const float tot = wave_sum_to_lane63(v[q]);
            const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 63));
            mine = (lane == q) ? sc : mine;
          }
          if (lane < 9) atomicAdd(gacc + (size_t)ids[k] * GACC_DWORDS + lane, mine);
        }
      }
    }
  }
}

}  // namespace segs
