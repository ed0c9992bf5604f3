// binning.hip -- tile binning: prefix sum, instance emission, device radix sort, tile ranges.
//
//   scan_block_sums_kernel        K5  reference: cub::DeviceScan::InclusiveSum   rasterizer_impl.cu:276-277
//   duplicate_with_keys_kernel    K7  reference: duplicateWithKeys               rasterizer_impl.cu:70-111
//   radix_{count,scan,scatter}    K8  reference: cub::DeviceRadixSort::SortPairs rasterizer_impl.cu:303-308
//   identify_tile_ranges_kernel   K9  reference: identifyTileRanges              rasterizer_impl.cu:116-138
//
// Integer/byte work, HBM-bound.  The sort is a stable LSD radix sort, 8 bits per pass over the key bits
// [0, 32+bit) exactly as the reference asks of CUB, written for wave64:
//  * a wave ranks 64 keys at a time with 8 ballots (one per digit bit) -> per-lane peer mask, rank =
//    mbcnt(peers), so equal digits cost no LDS-atomic serialisation (the depth exponent byte and the
//    tile bytes are extremely skewed);
//  * each workgroup owns 4096 consecutive keys (16 per lane), reorders them by digit in LDS and writes
//    every digit run with consecutive lanes on consecutive addresses;
//  * per-pass global offsets come from a digit-major [256][nblocks] count matrix scanned by 256
//    independent workgroups (no inter-workgroup hand-off inside a launch, so no cross-XCD visibility
//    protocol is needed).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gs_layout.h"
#include "kernels.h"

namespace segs {

// ---------------------------------------------------------------------------------------------
// K5: exclusive scan of the per-workgroup tiles_touched sums written by preprocess_fwd_kernel.
// One 1024-thread workgroup; in-place; total (= num_rendered R) to *total.
__global__ void __launch_bounds__(1024) scan_block_sums_kernel(uint32_t* __restrict__ block_sums, int nblocks,
                                                               uint32_t* __restrict__ total) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + tid;
    const uint32_t v = i < nblocks ? block_sums[i] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wv; w++) wbase += wave_tot[w];
    const uint32_t carry = carry_s;
    if (i < nblocks) block_sums[i] = carry + wbase + x - v;  // exclusive
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + x;
    __syncthreads();
  }
  if (tid == 0) *total = carry_s;
}

// ---------------------------------------------------------------------------------------------
// K7: one lane per Gaussian.  Finishes the inclusive scan inside the workgroup (point_offsets is the
// reference's GeometryState::point_offsets, bit-exact) and emits one (tile|depth, idx) pair per tile of
// the rect, row-major (y outer, x inner) like the reference.
__global__ void __launch_bounds__(256) duplicate_with_keys_kernel(
    int P, const BinInfo* __restrict__ bin, const uint32_t* __restrict__ block_offsets,
    uint32_t* __restrict__ point_offsets, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t gx) {
  __shared__ uint32_t wave_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int idx = blockIdx.x * 256 + tid;
  uint4 b = make_uint4(0, 0, 0, 0);
  float4 e = make_float4(0.f, 0.f, -1.f, -1.f);
  if (idx < P) {
    b = reinterpret_cast<const uint4*>(bin)[2 * (size_t)idx];
    e = reinterpret_cast<const float4*>(bin)[2 * (size_t)idx + 1];
  }
  uint32_t x = b.w;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wave_tot[wv] = x;
  __syncthreads();
  uint32_t base = block_offsets[blockIdx.x];
  for (int w = 0; w < wv; w++) base += wave_tot[w];
  const uint32_t incl = base + x;
  if (idx >= P) return;
  point_offsets[idx] = incl;
  if (b.w == 0) return;
  uint32_t off = incl - b.w;
  const uint32_t minx = b.y & 0xFFFFu, miny = b.y >> 16, maxx = b.z & 0xFFFFu, maxy = b.z >> 16;
  const float bx0 = e.x - e.z, bx1 = e.x + e.z, by0 = e.y - e.w, by1 = e.y + e.w;  // alpha-support box
  for (uint32_t y = miny; y < maxy; y++) {
    // rows of 8x8 quadrants of this tile row: pixel centres y*16 .. y*16+7 and y*16+8 .. y*16+15
    const float ty = (float)(y * TILE_Y);
    const uint32_t rowm = ((by0 <= ty + 7.f && by1 >= ty) ? 0x3u : 0u) | ((by0 <= ty + 15.f && by1 >= ty + 8.f) ? 0xCu : 0u);
    for (uint32_t xx = minx; xx < maxx; xx++) {
      const float tx = (float)(xx * TILE_X);
      const uint32_t colm = ((bx0 <= tx + 7.f && bx1 >= tx) ? 0x5u : 0u) | ((bx0 <= tx + 15.f && bx1 >= tx + 8.f) ? 0xAu : 0u);
      uint64_t key = (uint64_t)(y * gx + xx);
      key <<= 32;
      key |= (uint64_t)b.x;
      keys[off] = key;
      vals[off] = (uint32_t)idx | ((rowm & colm) << ID_BITS);
      off++;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K8 helpers.  digit of a key for this pass.
__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift) { return (uint32_t)(key >> shift) & 0xFFu; }

// Per-lane mask of the lanes (among `valid`) holding the same 8-bit digit: 8 ballots.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, uint64_t valid) {
  uint64_t peers = valid;
#pragma unroll
  for (int bit = 0; bit < 8; bit++) {
    const bool set = (d >> bit) & 1u;
    const uint64_t m = __ballot(set);
    peers &= set ? m : ~m;
  }
  return peers;
}
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) {  // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Count matrix: block_hist[d * nblocks + b] = number of keys of workgroup b's 4096-key tile with digit d.
__global__ void __launch_bounds__(SORT_THREADS) radix_count_kernel(const uint64_t* __restrict__ keys, int n, int shift,
                                                                    uint32_t* __restrict__ block_hist, int nblocks) {
  __shared__ uint32_t cnt[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 4 * 256; i += SORT_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const size_t wave_base = (size_t)blockIdx.x * SORT_TILE + (size_t)wv * (SORT_TILE / 4);
  volatile uint32_t* my = cnt[wv];
#pragma unroll 4
  for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {
    const size_t i = wave_base + (size_t)r * 64 + lane;
    const bool valid = i < (size_t)n;
    const uint32_t d = valid ? digit_of(keys[i], shift) : 0u;
    const uint64_t vmask = __ballot(valid);
    const uint64_t peers = match_digit(d, vmask);
    if (valid && mbcnt(peers) == 0) my[d] = my[d] + (uint32_t)__popcll(peers);  // one leader lane per digit
  }
  __syncthreads();
  block_hist[(size_t)tid * nblocks + blockIdx.x] = cnt[0][tid] + cnt[1][tid] + cnt[2][tid] + cnt[3][tid];
}

// Row d of the count matrix -> exclusive scan in place; row total -> digit_totals[d].  Grid = 256 workgroups.
__global__ void __launch_bounds__(256) radix_scan_kernel(uint32_t* __restrict__ block_hist, int nblocks,
                                                         uint32_t* __restrict__ digit_totals) {
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t* row = block_hist + (size_t)blockIdx.x * nblocks;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 256) {
    const int i = base + tid;
    const uint32_t v = i < nblocks ? row[i] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wv; w++) wbase += wave_tot[w];
    const uint32_t carry = carry_s;
    if (i < nblocks) row[i] = carry + wbase + x - v;
    __syncthreads();
    if (tid == 255) carry_s = carry + wbase + x;
    __syncthreads();
  }
  if (tid == 0) digit_totals[blockIdx.x] = carry_s;
}

// Stable scatter of one 4096-key tile.
__global__ void __launch_bounds__(SORT_THREADS) radix_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int n, int shift, const uint32_t* __restrict__ block_hist,
    const uint32_t* __restrict__ digit_totals, int nblocks) {
  __shared__ uint64_t s_keys[SORT_TILE];
  __shared__ uint32_t s_vals[SORT_TILE];
  __shared__ uint32_t cnt[4][256];       // per-wave running digit counters, then per-wave bases
  __shared__ uint32_t local_start[256];  // start of digit run inside the tile
  __shared__ int32_t gdelta[256];        // global position - local position, per digit
  __shared__ uint32_t scan_tmp[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 4 * 256; i += SORT_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();

  const size_t tile_base = (size_t)blockIdx.x * SORT_TILE;
  const size_t wave_base = tile_base + (size_t)wv * (SORT_TILE / 4);
  const int nvalid = (int)min((size_t)SORT_TILE, (size_t)n - tile_base);

  uint64_t key[SORT_ITEMS_PER_THREAD];
  uint32_t val[SORT_ITEMS_PER_THREAD];
  uint32_t drank[SORT_ITEMS_PER_THREAD];  // digit | wave-local rank << 8 ; 0xFFFFFFFF = invalid
  volatile uint32_t* my = cnt[wv];
#pragma unroll
  for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {
    const size_t i = wave_base + (size_t)r * 64 + lane;
    const bool valid = i < (size_t)n;
    key[r] = valid ? keys_in[i] : 0ull;
    val[r] = valid ? vals_in[i] : 0u;
  }
#pragma unroll
  for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {
    const size_t i = wave_base + (size_t)r * 64 + lane;
    const bool valid = i < (size_t)n;
    const uint32_t d = valid ? digit_of(key[r], shift) : 0u;
    const uint64_t vmask = __ballot(valid);
    const uint64_t peers = match_digit(d, vmask);
    const uint32_t below = mbcnt(peers);
    uint32_t old = 0;
    if (valid) old = my[d];
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) my[d] = old + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    drank[r] = valid ? (d | ((old + below) << 8)) : 0xFFFFFFFFu;
  }
  __syncthreads();

  // digit `tid`: per-wave exclusive bases, tile count, then exclusive scan over digits.
  uint32_t run = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = run; run += c; }
  uint32_t x = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) scan_tmp[wv] = x;
  // global base of digit `tid` = (sum of totals of smaller digits) + this tile's offset inside the digit row
  uint32_t tot = digit_totals[tid];
  uint32_t tx = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t y = __shfl_up(tx, off, 64);
    if (lane >= off) tx += y;
  }
  __shared__ uint32_t tot_tmp[4];
  if (lane == 63) tot_tmp[wv] = tx;
  __syncthreads();
  uint32_t lbase = 0, gbase = 0;
  for (int w = 0; w < wv; w++) { lbase += scan_tmp[w]; gbase += tot_tmp[w]; }
  const uint32_t lstart = lbase + x - run;
  const uint32_t gstart = gbase + tx - tot + block_hist[(size_t)tid * nblocks + blockIdx.x];
  local_start[tid] = lstart;
  gdelta[tid] = (int32_t)(gstart - lstart);
  __syncthreads();

#pragma unroll
  for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {
    if (drank[r] != 0xFFFFFFFFu) {
      const uint32_t d = drank[r] & 0xFFu;
      const uint32_t pos = local_start[d] + cnt[wv][d] + (drank[r] >> 8);
      s_keys[pos] = key[r];
      s_vals[pos] = val[r];
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {
    const int lp = r * SORT_THREADS + tid;
    if (lp < nvalid) {
      const uint64_t k = s_keys[lp];
      const size_t gp = (size_t)((int64_t)lp + (int64_t)gdelta[digit_of(k, shift)]);
      keys_out[gp] = k;
      vals_out[gp] = s_vals[lp];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K9 (ranges are zeroed by the caller with hipMemsetAsync, as rasterizer_impl.cu:310 does).
__global__ void __launch_bounds__(256) identify_tile_ranges_kernel(int L, const uint64_t* __restrict__ keys,
                                                                   uint2* __restrict__ ranges) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= L) return;
  const uint32_t cur = (uint32_t)(keys[idx] >> 32);
  if (idx == 0) ranges[cur].x = 0;
  else {
    const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
    if (cur != prev) { ranges[prev].y = idx; ranges[cur].x = idx; }
  }
  if (idx == L - 1) ranges[cur].y = L;
}

// ---------------------------------------------------------------------------------------------
// Test support: expand the packed per-Gaussian state into the reference's GeometryState arrays.
__global__ void __launch_bounds__(256) unpack_geometry_kernel(
    int P, const float* __restrict__ rec, const BinInfo* __restrict__ bin, const int* __restrict__ radii,
    float* __restrict__ means2D, float* __restrict__ conic_opacity, float* __restrict__ depths,
    uint32_t* __restrict__ tiles_touched, float* __restrict__ rgb) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const bool vis = radii[idx] > 0;
  const float* r = rec + (size_t)idx * REC_DWORDS;
  means2D[2 * (size_t)idx + 0] = vis ? r[REC_X] : 0.f;
  means2D[2 * (size_t)idx + 1] = vis ? r[REC_Y] : 0.f;
  conic_opacity[4 * (size_t)idx + 0] = vis ? r[REC_CA] : 0.f;
  conic_opacity[4 * (size_t)idx + 1] = vis ? r[REC_CB] : 0.f;
  conic_opacity[4 * (size_t)idx + 2] = vis ? r[REC_CC] : 0.f;
  conic_opacity[4 * (size_t)idx + 3] = vis ? r[REC_O] : 0.f;
  depths[idx] = vis ? r[REC_DEPTH] : 0.f;
  tiles_touched[idx] = bin[idx].tiles_touched;
  if (rgb) {
    rgb[3 * (size_t)idx + 0] = vis ? r[REC_R] : 0.f;
    rgb[3 * (size_t)idx + 1] = vis ? r[REC_G] : 0.f;
    rgb[3 * (size_t)idx + 2] = vis ? r[REC_B] : 0.f;
  }
}

// point_list as the reference defines it: strip the quadrant mask from the sorted values.
__global__ void __launch_bounds__(256) strip_mask_kernel(int n, const uint32_t* __restrict__ vals, uint32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = vals[i] & ID_MASK;
}

}  // namespace segs
