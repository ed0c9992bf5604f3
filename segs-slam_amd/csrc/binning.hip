// binning.hip -- tile binning: prefix sum, instance emission, device radix sort, tile ranges.
//
//   scan_block_sums_kernel        K5  reference: cub::DeviceScan::InclusiveSum   rasterizer_impl.cu:276-277
//   duplicate_with_keys_kernel    K7  reference: duplicateWithKeys               rasterizer_impl.cu:70-111
//   radix_{count,scan,scatter}    K8  reference: cub::DeviceRadixSort::SortPairs rasterizer_impl.cu:303-308
//   identify_tile_ranges_kernel   K9  reference: identifyTileRanges              rasterizer_impl.cu:116-138
//
// Integer/byte work.  The reference sorts R 64-bit (tile | depth) keys in ceil((32+bit)/8) byte passes; here the order is
// produced by a TWO-LEVEL sort with the same result (capi.hip run_binning): the P Gaussians are sorted by their depth
// bits once (the passes over P keys are bound by their launch count, so this sort uses 9-bit digits: 3 passes for the
// usual 26-27 significant bits), instances are emitted in that order by a slot-parallel emitter, and the R instances are
// then sorted by their tile id alone (2 byte passes at <= 65 536 tiles; in resident mode the first of them also drops
// the instances the emitter marked dead).  Both use this stable LSD radix sort, 32-bit keys (segs_sort_pairs exposes the
// 64-bit instantiation), written for wave64:
//  * a wave ranks 64 keys at a time with one ballot per digit bit -> per-lane peer mask, rank = mbcnt(peers), so
//    equal digits cost no LDS-atomic serialisation (the depth exponent byte and the tile bytes are extremely skewed);
//  * each workgroup owns 2048 consecutive keys (8 per lane), reorders them by digit in LDS and writes every digit run with
//    consecutive lanes on consecutive addresses;
//  * per-pass global offsets come from a digit-major [digits][nblocks] count matrix scanned by one independent workgroup
//    per digit (no inter-workgroup hand-off inside a launch, so no cross-XCD visibility protocol is needed).
// On this part rocPRIM's radix_sort_pairs takes 124 us / 83 us for the two sorts of the headline workload, these kernels
// 62 us / 68 us (profiles/r01_rocprim_sort_reference.txt).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gs_layout.h"
#include "kernels.h"

namespace segs {

constexpr int PREFIX_ROWS = PREFIX_ROWS_PER_WG;  // gs_layout.h

// ---------------------------------------------------------------------------------------------
// K5 (reference: cub::DeviceScan::InclusiveSum + the D2H copy of its last element, rasterizer_impl.cu:276-281): what the
// host needs before it can size the binning scratch -- the TOTAL of the per-workgroup tiles_touched sums written by
// preprocess_fwd_kernel (= num_rendered R) and the depth range of the binned Gaussians.  The offsets themselves are formed
// later, in depth order (ordered_offsets_kernel), so this is a reduction, not a scan: one 1024-thread workgroup, every
// load of a thread in flight at once.  (Until round 4 it also wrote the exclusive scan of the sums back in place -- twelve
// dependent rounds of load / wave scan / barrier at 3 M Gaussians, 46 us for a result nothing read any more.)
__global__ void __launch_bounds__(1024) scan_block_sums_kernel(uint32_t* __restrict__ block_sums, int nblocks,
                                                               const uint32_t* __restrict__ depth_range,
                                                               uint32_t* __restrict__ total) {
  __shared__ uint32_t w_sum[16], w_dmax[16], w_dnmin[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t sum = 0, dmax = 0, dnmin = 0;
  for (int i = tid; i < nblocks; i += 1024) {
    sum += block_sums[i];
    dmax = max(dmax, depth_range[i]);
    dnmin = max(dnmin, depth_range[nblocks + i]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, 64);
    dmax = max(dmax, (uint32_t)__shfl_down((int)dmax, off, 64));
    dnmin = max(dnmin, (uint32_t)__shfl_down((int)dnmin, off, 64));
  }
  if (lane == 0) { w_sum[wv] = sum; w_dmax[wv] = dmax; w_dnmin[wv] = dnmin; }
  __syncthreads();
  if (tid == 0) {
    uint32_t s = 0, a = 0, b = 0;
    for (int w = 0; w < 16; w++) { s += w_sum[w]; a = max(a, w_dmax[w]); b = max(b, w_dnmin[w]); }
    total[0] = s; total[1] = b; total[2] = a;  // R, max(~depth), max(depth)
  }
}

// ---------------------------------------------------------------------------------------------
// K7.  Finishes the inclusive scan inside the workgroup (point_offsets is the reference's
// GeometryState::point_offsets, bit-exact) and emits one (tile|depth, idx) pair per tile of each rect,
// row-major (y outer, x inner) like the reference.  Emission is load-balanced per wave: the wave's 64 rects
// are expanded slot by slot (lane j handles output slot j, owner found by binary search in the wave's
// prefix), so the 12-byte pairs leave as coalesced stores whatever the rect sizes are.
//
// Each emitted VALUE also carries the 4-bit mask of 8x8 quadrants of the tile in which this Gaussian can pass
// the alpha >= 1/255 test at all: exact minimum of the quadratic form over the quadrant's pixel rectangle
// against 2 ln(255 o) (conservatively inflated).  See gs_layout.h.
// The test for one pixel rectangle (spans relative to the centre): the minimum of A dx^2 + 2 B dx dy + C dy^2 over the
// rectangle against k; along a line (say dx fixed) the form is a quadratic in dy, clamped to the span at its vertex
// -B dx / C.  quadrant_mask does this for the four 8x8 quadrants of the tile at (fx, fy) at once; evaluated as
// a dx^2 + dy (2 b dx + c dy); the threshold k is inflated by the caller, so last-bit differences stay on the conservative side.
__device__ __forceinline__ uint32_t quadrant_mask(float cx, float cy, float A, float B, float C, float k, float nb_c /* -B/C */,
                                                  float nb_a /* -B/A */, float fx, float fy) {
  // q(dx, dy) = A dx^2 + 2 B dx dy + C dy^2 is convex with its minimum at the centre, so over a pixel rectangle that does not
  // contain the centre the minimum sits on an edge FACING the centre (walking from any other boundary point towards the
  // centre lowers q and leaves the rectangle through such an edge).  With xn / yn = the centre's coordinates clamped into
  // the rectangle's spans, two clamped 1-D minima decide a quadrant: along x = xn (vertex at dy = -B/C xn) and along
  // y = yn (vertex at dx = -B/A yn); a centre inside the rectangle gives xn = yn = 0 and q = 0.  (The first version took
  // all four edges of each quadrant, 16 edge minima per tile at ~150 operations; this is ~85, and the emitter is VALU-bound.)
  const float x_lo[2] = {fx - cx, fx + 8.f - cx}, x_hi[2] = {fx + 7.f - cx, fx + 15.f - cx};
  const float y_lo[2] = {fy - cy, fy + 8.f - cy}, y_hi[2] = {fy + 7.f - cy, fy + 15.f - cy};
  const float B2 = 2.f * B;
  float xn[2], yn[2], axx[2], bx2[2], tv[2], cyy[2], by2[2], th[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    xn[i] = fminf(x_hi[i], fmaxf(x_lo[i], 0.f));
    yn[i] = fminf(y_hi[i], fmaxf(y_lo[i], 0.f));
    axx[i] = A * xn[i] * xn[i]; bx2[i] = B2 * xn[i]; tv[i] = nb_c * xn[i];
    cyy[i] = C * yn[i] * yn[i]; by2[i] = B2 * yn[i]; th[i] = nb_a * yn[i];
  }
  uint32_t mask = 0u;
#pragma unroll
  for (int qd = 0; qd < 4; qd++) {
    const int qx = qd & 1, qy = qd >> 1;
    const float yy = fminf(y_hi[qy], fmaxf(y_lo[qy], tv[qx]));
    const float v = axx[qx] + yy * (bx2[qx] + C * yy);
    const float xx = fminf(x_hi[qx], fmaxf(x_lo[qx], th[qy]));
    const float h = cyy[qy] + xx * (by2[qy] + A * xx);
    if (fminf(v, h) <= k) mask |= 1u << qd;
  }
  return mask;
}

// Inclusive offsets of tiles_touched in DEPTH order (one per sorted slot): block prefix from scan_block_sums_kernel
// plus an in-workgroup scan.
__global__ void __launch_bounds__(256) ordered_offsets_kernel(int P, const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ incl /* in: tiles_touched in depth order (ordered_block_sums_kernel); out: inclusive offsets */,
                                                              uint32_t* __restrict__ total_out,
                                                              const uint32_t* __restrict__ ng_dev /* slots that hold a binned Gaussian, or null = P */,
                                                              uint32_t* __restrict__ first_owner, uint32_t owner_entries) {
  // A workgroup owns PREFIX_ROWS rows of 256 consecutive slots (the same cut as ordered_block_sums_kernel).  Workgroups wholly
  // behind the binned Gaussians have nothing to scan (the emitter never looks there); the last one still reports the total.
  const uint32_t slot0 = (uint32_t)blockIdx.x * 256u * PREFIX_ROWS;
  if (ng_dev && slot0 >= *ng_dev && blockIdx.x != gridDim.x - 1) return;
  // Every workgroup sums the (unscanned) sums of the workgroups before it on its own -- P/2048 values, a few loads per
  // thread -- instead of a separate single-workgroup scan kernel between the two passes (one launch less).  (With one sum per
  // 256 slots these reads were quadratic in earnest: 270 MB of L2 traffic at 3 M Gaussians.)
  __shared__ uint32_t wave_tot[4], red[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t before = 0;
  for (int b = tid; b < (int)blockIdx.x; b += 256) before += block_sums[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off, 64);
  if (lane == 0) red[wv] = before;
  __syncthreads();
  uint32_t carry = red[0] + red[1] + red[2] + red[3];
  uint32_t vals[PREFIX_ROWS];
#pragma unroll
  for (int r = 0; r < PREFIX_ROWS; r++) {   // all rows in flight before the first scan
    const uint32_t slot = slot0 + (uint32_t)r * 256u + tid;
    vals[r] = slot < (uint32_t)P ? incl[slot] : 0u;
  }
#pragma unroll
  for (int r = 0; r < PREFIX_ROWS; r++) {
    const uint32_t slot = slot0 + (uint32_t)r * 256u + tid;
    uint32_t x = vals[r];
    const uint32_t t0 = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    uint32_t base = carry;
    for (int w = 0; w < wv; w++) base += wave_tot[w];
    const uint32_t v = base + x;
    if (slot < (uint32_t)P) incl[slot] = v;
    // The emitter's workgroup k starts at instance slot k * EMIT_SLOTS: the Gaussian whose range [excl, incl) holds that slot
    // says so here, which saves the emitter two 64-ary searches (eight dependent loads) per workgroup.
    if (slot < (uint32_t)P && first_owner && t0 != 0u) {
      const uint32_t excl = v - t0;
      for (uint32_t k = (excl + EMIT_SLOTS_PER_WG - 1) / EMIT_SLOTS_PER_WG; k * EMIT_SLOTS_PER_WG < v && k < owner_entries; k++)
        first_owner[k] = slot;
    }
    carry += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  }
  if (total_out && blockIdx.x == gridDim.x - 1 && tid == 0) total_out[0] = carry;   // R = inclusive total
}

// The emitter is parallel over OUTPUT slots, not over Gaussians: a sub-batch is 512 consecutive instance slots of the unsorted
// (depth-ordered) list; the Gaussians (in depth order) that own them come from the first_owner table the prefix kernel left
// (a 64-ary search in the inclusive offsets before: eight dependent loads per workgroup), are staged once in LDS, and every
// thread resolves the owners of its TWO CONSECUTIVE slots and writes the two 8-byte pairs as one coalesced store each.
// A Gaussian covering thousands of tiles is thereby spread over many workgroups instead of serialising one wave.
//
// What bounds it (round 3 measurements, 3 M Gaussians: 7.9 M slots, 2.1 M owners, 66 us): the 2.1 M gathers of emit records
// in depth order.  tools/ubench_gather.hip: 2.1 M random gathers from a freshly written table take 43-47 us whether a record
// is 16, 32, 64 or 128 bytes -- a gather pulls one whole 128-byte line, ~47 G lines/s chip-wide -- so 45 us is the floor
// of ANY kernel that visits the records in depth order, and carrying them through the depth sort instead costs more.
// Tried against the other 20 us and not kept: (i) the whole load chain of a sub-batch (first_owner -> order / offsets ->
// record) software-pipelined across the 8-16 sub-batches of a chunk-walking workgroup that also leaves the first tile-id
// pass's count tables (emit_count_kernel, removed again): 92 us against 66 + 15 for this kernel plus radix_count_kernel --
// 965-1930 fat workgroups keep fewer gathers in flight than 15 400 small ones, whatever the chunk size (PMC: 2.4 waves per
// SIMD resident on average against 5.2); (ii) fewer instructions: the owner of a slot used to be found by an LDS binary
// search (8 dependent rounds) and the tile row by an IEEE division; now every owner drops its index at the position of its
// first slot into a 512-entry LDS array and a running maximum over the slots (max of the thread's two entries, a wave64 scan,
// four wave totals) gives every slot its owner, and the row comes from one v_rcp_f32 plus the two corrections that were
// there anyway: 12 % fewer VALU instructions, 66.6 -> 65.6 us.  Kept because it is the simpler code.
// 512 slots: 24 KB of LDS per workgroup -> 6 workgroups per CU.
constexpr int EMIT_SLOTS = EMIT_SLOTS_PER_WG;   // gs_layout.h
static_assert(EMIT_SLOTS == 2 * 256, "two consecutive slots per thread");
struct EmitLds {
  uint32_t excl[EMIT_SLOTS + 2];   // first slot of owner k (= inclusive offset of the Gaussian before it); n_own + 1 entries
  uint32_t idx[EMIT_SLOTS + 1];    // Gaussian index of owner k
  uint2 bin[EMIT_SLOTS + 1];       // rect min (x | y << 16), rect width in tiles
  float4 geo0[EMIT_SLOTS + 1];     // x, y, A', B' (the conic scaled for exp2, sign flipped: see emit_stage_owner)
  float2 geo1[EMIT_SLOTS + 1];     // C', k = inflated log2(255 o) (or -1: no pixel can reach alpha >= 1/255)
  uint32_t mark[EMIT_SLOTS];       // 1 << 10 | owner, at the owner's first slot (0 = no owner starts here)
  uint32_t wave_max[4];
};
constexpr uint32_t EMIT_OWNER_MASK = 1023u;

__device__ __forceinline__ void emit_stage_owner(EmitLds& L, int k, uint32_t idx, const float4& e0, const float4& e1) {
  // e0, e1: x y A2 B2 | C2 opacity rect_min rect_max, picked from the Gaussian's record.  The quadrant test runs on
  // q(d) = -(A2 dx^2 + B2 dx dy + C2 dy^2) = -log2 of the Gaussian's falloff: alpha >= 1/255  <=>  q <= log2(255 o).
  const uint32_t rmin = __float_as_uint(e1.z), rmax = __float_as_uint(e1.w);
  L.idx[k] = idx;
  L.bin[k] = make_uint2(rmin, (rmax & 0xFFFFu) - (rmin & 0xFFFFu));
  const float op = e1.y;
  L.geo0[k] = make_float4(e0.x, e0.y, -e0.z, -0.5f * e0.w);    // x, y, A' = -A2, B' = -B2 / 2  (q = A' dx^2 + 2 B' dx dy + C' dy^2)
  L.geo1[k] = make_float2(-e1.x, (op * 255.0f > 1.0f) ? __builtin_amdgcn_logf(255.0f * op) * 1.0001f + 1e-3f : -1.0f);
}
// owner k's first slot is `start`; its mark goes to the slot's position inside the sub-batch [s0, s0 + 512) (owner 0 may
// start before s0: position 0; an owner that starts behind the sub-batch -- the next sub-batch's first -- leaves no mark).
// Two owners write the same mark only when one of them owns no instance, which happens only in a launch flagged through
// status[2] (a binned depth beyond the key range): which of them wins is then not deterministic, and that launch's output is dropped.
__device__ __forceinline__ void emit_mark_owner(uint32_t* mark, int k, uint32_t start, uint32_t s0, uint32_t tag) {
  const uint32_t pos = start > s0 ? start - s0 : 0u;
  if (pos < (uint32_t)EMIT_SLOTS) mark[pos] = (tag << 10) | (uint32_t)k;
}
// Owners of slots 2 tid and 2 tid + 1 of the sub-batch from the marks (first half: up to the wave totals; a barrier between
// the halves is the caller's, which has other LDS writes to publish with it).
__device__ __forceinline__ void emit_scan_begin(EmitLds& L, const uint32_t* mark, int tid, uint32_t& m0, uint32_t& incl) {
  const uint2 m = *reinterpret_cast<const uint2*>(mark + 2 * tid);
  m0 = m.x;
  uint32_t x = max(m.x, m.y);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off, 64);
    if ((tid & 63) >= off) x = max(x, y);
  }
  incl = x;
  if ((tid & 63) == 63) L.wave_max[tid >> 6] = x;
}
__device__ __forceinline__ void emit_scan_end(const EmitLds& L, int tid, uint32_t m0, uint32_t incl, int& g0, int& g1) {
  const int wv = tid >> 6;
  uint32_t before = __shfl_up(incl, 1, 64);
  if ((tid & 63) == 0) before = 0u;
  for (int w = 0; w < wv; w++) before = max(before, L.wave_max[w]);
  g0 = (int)(max(before, m0) & EMIT_OWNER_MASK);
  g1 = (int)(max(before, incl) & EMIT_OWNER_MASK);
}
// One (tile | quadrant mask, Gaussian) instance: slot j of owner g.
__device__ __forceinline__ void emit_instance(const EmitLds& L, int g, uint32_t j, uint32_t gx, int mark_dead, uint32_t& key, uint32_t& val) {
  const uint2 bb = L.bin[g];
  const uint32_t t = j - L.excl[g];
  const uint32_t minx = bb.x & 0xFFFFu, miny = bb.x >> 16, w = bb.y;
  // row = t / w: the reciprocal is good to an ulp and t < 2^24, so the estimate is off by at most one -- the corrections
  uint32_t q = (uint32_t)((float)t * __builtin_amdgcn_rcpf((float)w));
  if (q * w > t) q--;
  if ((q + 1) * w <= t) q++;
  const uint32_t ty = miny + q, tx = minx + (t - q * w);
  const float4 ge0 = L.geo0[g];
  const float2 ge1 = L.geo1[g];
  const float k = ge1.y;
  // vertex positions -B/C x and -B/A y of the two line minima: a reciprocal is exact enough -- at a minimum the form is flat
  // in the position, and k is inflated (quadrant_mask)
  const uint32_t mask = k > 0.f ? quadrant_mask(ge0.x, ge0.y, ge0.z, ge0.w, ge1.x, k, -ge0.w * __builtin_amdgcn_rcpf(ge1.x),
                                                -ge0.w * __builtin_amdgcn_rcpf(ge0.z), (float)(tx * TILE_X), (float)(ty * TILE_Y)) : 0u;
  key = (mark_dead && mask == 0u) ? DEAD_KEY : ty * gx + tx;   // tile id only: depth order is already the emission order
  val = L.idx[g] | (mask << ID_BITS);
}

__global__ void __launch_bounds__(256) duplicate_with_keys_kernel(
    int P, int R, const float* __restrict__ rec /* the 64-byte per-Gaussian records, gs_layout.h */, const uint32_t* __restrict__ order,
    const uint32_t* __restrict__ incl, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t gx,
    const uint32_t* __restrict__ n_dev /* resident mode: R lives on the device, `R` is the capacity */,
    int mark_dead /* instances that reach no quadrant get DEAD_KEY: the first tile-id pass drops them */,
    const uint32_t* __restrict__ ng_dev /* resident mode: Gaussians left in `order` after the depth sort dropped the culled ones */,
    const uint32_t* __restrict__ first_owner /* ordered_offsets_kernel: owner of slot k * EMIT_SLOTS */) {
  __shared__ EmitLds L;
  if (n_dev) R = (int)min(*n_dev, (uint32_t)R);
  if (ng_dev) P = (int)min(*ng_dev, (uint32_t)P);
  if (P <= 0) return;
  const int tid = threadIdx.x;
  const uint32_t s0 = blockIdx.x * EMIT_SLOTS, s1 = min((uint32_t)R, s0 + EMIT_SLOTS);
  if (s0 >= (uint32_t)R) return;
  // first owner: first g with incl[g] > s0 -- the owner of this workgroup's first slot, left in the table by
  // ordered_offsets_kernel; last owner: the owner of the NEXT workgroup's first slot can only be the same Gaussian or the one
  // after the last owner here (every entry ahead of the zero-instance tail owns at least one slot), so it bounds the staging;
  // the last workgroup stages up to the end.
  const int g_lo = (int)first_owner[blockIdx.x];
  const int lo = s1 < (uint32_t)R ? (int)first_owner[blockIdx.x + 1] : P - 1;
  // <= EMIT_SLOTS when every owner has at least one instance in range.  In resident mode a binned Gaussian whose depth key
  // reached the culled key (the step is then flagged through status[2] and redone) can leave zero-instance owners between
  // binned ones: the clamp keeps that flagged launch inside the LDS staging arrays.
  const int n_own = min(min(lo, P - 1) - g_lo + 1, EMIT_SLOTS + 1);
  *reinterpret_cast<uint2*>(&L.mark[2 * tid]) = make_uint2(0u, 0u);
  __syncthreads();
  for (int k = tid; k <= n_own; k += 256) {
    const uint32_t start = (g_lo + k == 0) ? 0u : incl[g_lo + k - 1];
    L.excl[k] = start;
    if (k < n_own) {
      emit_mark_owner(L.mark, k, start, s0, 1u);
      const uint32_t idx = order[g_lo + k];
      const float* r = rec + (size_t)idx * REC_DWORDS;   // ONE line per owner, 32 bytes of it in three loads (gs_layout.h)
      const float4 q0 = *reinterpret_cast<const float4*>(r);                   // x y A2 B2
      const float2 q1 = *reinterpret_cast<const float2*>(r + REC_C2);          // C2 opacity
      const float2 q2 = *reinterpret_cast<const float2*>(r + REC_RECT_MIN);    // rect_min rect_max
      emit_stage_owner(L, k, idx, q0, make_float4(q1.x, q1.y, q2.x, q2.y));
    }
  }
  __syncthreads();
  uint32_t m0, inc;
  emit_scan_begin(L, L.mark, tid, m0, inc);
  __syncthreads();
  int g0, g1;
  emit_scan_end(L, tid, m0, inc, g0, g1);
  const uint32_t j = s0 + 2 * tid;
  uint32_t k0 = 0u, v0 = 0u, k1 = 0u, v1 = 0u;
  if (j < s1) emit_instance(L, g0, j, gx, mark_dead, k0, v0);
  if (j + 1 < s1) emit_instance(L, g1, j + 1, gx, mark_dead, k1, v1);
  if (j + 1 < s1) {
    *reinterpret_cast<uint2*>(keys + j) = make_uint2(k0, k1);
    *reinterpret_cast<uint2*>(vals + j) = make_uint2(v0, v1);
  } else if (j < s1) {
    keys[j] = k0; vals[j] = v0;
  }
}

// ---------------------------------------------------------------------------------------------
// K8 helpers.  64-bit keys (segs_sort_pairs): the digit is taken from k' = (hi32 << dbits) | (lo32 - dmin), or from hi32
// alone when dbits < 0.  32-bit keys (the pipeline): digit of (key - dmin).  Depth bits of positive floats are monotone
// in the depth, so sorting on them (ties included) gives exactly the reference's order.
struct KeyMap { uint32_t dmin; int dbits; };
#define DEAD_KEY_OF(K) (~(K)0)   // an all-ones key marks an entry the first pass of a sort may drop (see run_binning)
template <int BITS>
__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift, KeyMap km) {
  const uint64_t kc = km.dbits < 0 ? (key >> 32) : (((key >> 32) << km.dbits) | (uint64_t)((uint32_t)key - km.dmin));
  return (uint32_t)(kc >> shift) & ((1u << BITS) - 1u);
}
// 32-bit keys (the pipeline's own two sorts: depth bits of the P Gaussians, tile ids of the R instances): 20 instead of
// 32 bytes moved per key and pass.
template <int BITS>
__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift, KeyMap km) { return ((key - km.dmin) >> shift) & ((1u << BITS) - 1u); }

// Per-lane mask of the lanes (among `valid`) holding the same digit: one ballot per digit bit that can be set (`nbits` <=
// BITS, wave-uniform: the last pass of a sort usually has fewer significant bits than a full digit).
template <int BITS>
__device__ __forceinline__ uint64_t match_digit(uint32_t d, uint64_t valid, int nbits) {
  uint64_t peers = valid;
#pragma unroll
  for (int bit = 0; bit < BITS; bit++) {
    if (bit < nbits) {
      const bool set = (d >> bit) & 1u;
      const uint64_t m = __ballot(set);
      peers &= set ? m : ~m;
    }
  }
  return peers;
}
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) {  // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Digit counts of one pass, in two levels.  A workgroup walks a CHUNK of chunk_tiles (1, 2 or 4: sort_pairs) consecutive 2048-key tiles and keeps a
// running per-digit count: for every tile it writes the count of each digit in the chunk's EARLIER tiles,
//     tile_prefix[b * NDIG + d]            (tile-major: one coalesced 1-2 KB row per tile, read back coalesced by the scatter)
// and at the end the chunk's totals, chunk_hist[d * nchunks + c] (digit-major rows for radix_scan_kernel).  Only that small
// matrix -- chunk_tiles times fewer columns than one column per tile -- goes through the row scan.  (One column per tile,
// digit-major, cost 14 x write amplification -- one dword per 64-B line -- and a 10 us scan launch per pass at 3 M Gaussians.)
// BITS = 8 everywhere except the depth sort, whose 26-27 significant key bits take three 9-bit passes instead of four
// 8-bit ones (the passes over P keys are bound by their launch count, not by bytes).
template <typename K, int BITS>
__global__ void __launch_bounds__(SORT_THREADS) radix_count_kernel(const K* __restrict__ keys, int n, int shift,
                                                                    uint32_t dmin, int dbits,
                                                                    uint32_t* __restrict__ tile_prefix, uint32_t* __restrict__ chunk_hist,
                                                                    int nblocks, int nchunks,
                                                                    const uint32_t* __restrict__ n_dev, int drop_dead, int chunk_tiles,
                                                                    int nbits /* significant bits of this pass's digit: key bits at or above the sort's end_bit are ignored */) {
  const KeyMap km{dmin, dbits};
  if (n_dev) n = (int)min(*n_dev, (uint32_t)n);
  constexpr int NDIG = 1 << BITS;
  constexpr int DPT = NDIG / SORT_THREADS;
  __shared__ uint32_t cnt[4][NDIG];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t dmask = (1u << nbits) - 1u;
  uint32_t* my = cnt[wv];
  uint32_t running[DPT];
#pragma unroll
  for (int j = 0; j < DPT; j++) running[j] = 0u;
  K kreg[SORT_ITEMS_PER_THREAD], knext[SORT_ITEMS_PER_THREAD];
  // chunk of this workgroup: each XCD takes a contiguous range of the chunks in use, like the scatter's tiles -- the chunk
  // totals are written one dword per digit row, and consecutive chunks' dwords share their lines
  const int chunks_in_use = (int)(((size_t)n + (size_t)SORT_TILE * chunk_tiles - 1) / ((size_t)SORT_TILE * chunk_tiles));
  const int chunks_per_xcd = (chunks_in_use + 7) >> 3;
  const int chunk_id = (int)(blockIdx.x & 7u) * chunks_per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= chunks_per_xcd) {
    // no keys for this workgroup: it zeroes the totals of one of the columns beyond the chunks in use (the row scan runs over
    // all nchunks columns of the capacity)
    const int col = 8 * chunks_per_xcd + ((int)(blockIdx.x >> 3) - chunks_per_xcd) * 8 + (int)(blockIdx.x & 7u);
    if (col < nchunks) {
      constexpr int ND = 1 << BITS;
      for (int d = threadIdx.x; d < ND; d += SORT_THREADS) chunk_hist[(size_t)d * nchunks + col] = 0u;
    }
    return;
  }
  if (chunk_id >= nchunks) return;
  const int b0 = chunk_id * chunk_tiles;
  auto load_tile = [&](int b, K* dst) {
    const size_t wave_base = (size_t)b * SORT_TILE + (size_t)wv * (SORT_TILE / 4);
#pragma unroll
    for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {  // all loads in flight before the first use
      const size_t i = wave_base + (size_t)r * 64 + lane;
      dst[r] = (b < nblocks && i < (size_t)n) ? keys[i] : (K)DEAD_KEY_OF(K);
    }
  };
  load_tile(b0, kreg);
  for (int t = 0; t < chunk_tiles; t++) {
    const int b = b0 + t;
    if (b >= nblocks) break;
    for (int i = tid; i < 4 * NDIG; i += SORT_THREADS) (&cnt[0][0])[i] = 0;
    if (t + 1 < chunk_tiles) load_tile(b + 1, knext);      // next tile's keys in flight while this one is counted
    __syncthreads();
    // Counting needs no ranks: one LDS atomic per key into the wave's private histogram.  Lanes with equal digits
    // serialise inside the LDS (worst case, one digit for the whole wave, about the cost of the 8-ballot peer match the
    // scatter needs for its stable ranks), spread digits cost a few cycles.
    const size_t wave_base = (size_t)b * SORT_TILE + (size_t)wv * (SORT_TILE / 4);
#pragma unroll
    for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {
      const size_t i = wave_base + (size_t)r * 64 + lane;
      if (i < (size_t)n && !(drop_dead && kreg[r] == (K)DEAD_KEY_OF(K))) atomicAdd(&my[digit_of<BITS>(kreg[r], shift, km) & dmask], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DPT; j++) {
      const int d = tid * DPT + j;
      tile_prefix[(size_t)b * NDIG + d] = running[j];
      running[j] += cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) kreg[r] = knext[r];
  }
#pragma unroll
  for (int j = 0; j < DPT; j++) chunk_hist[(size_t)(tid * DPT + j) * nchunks + chunk_id] = running[j];
}

// Row d of the count matrix -> exclusive scan in place; row total -> digit_totals[d].  Grid = one workgroup per digit.
// Eight consecutive entries per thread and round (2048 per round: one round up to R = 4 M instances), one barrier pair
// per round -- the previous one-entry-per-thread loop spent 6 us per launch on barriers.
__global__ void __launch_bounds__(256) radix_scan_kernel(uint32_t* __restrict__ block_hist, int nblocks,
                                                         uint32_t* __restrict__ digit_totals) {
  constexpr int ITEMS = 8;
  __shared__ uint32_t wave_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t* row = block_hist + (size_t)blockIdx.x * nblocks;
  uint32_t carry = 0;
  for (int base = 0; base < nblocks; base += 256 * ITEMS) {
    const int i0 = base + tid * ITEMS;
    uint32_t v[ITEMS], sum = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) { v[q] = (i0 + q < nblocks) ? row[i0 + q] : 0u; sum += v[q]; }
    uint32_t x = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t t = wave_tot[w]; if (w < wv) wbase += t; total += t; }
    uint32_t run = carry + wbase + x - sum;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) { if (i0 + q < nblocks) row[i0 + q] = run; run += v[q]; }
    carry += total;
    __syncthreads();
  }
  if (tid == 0) digit_totals[blockIdx.x] = carry;
}

// Stable scatter of one 2048-key tile (SORT_TILE).
// vals_in == nullptr: the value of entry i is i itself (first pass of an index sort: no iota array is written or read).
// AUX (last pass of the depth sort): aux_out[position of entry i] = aux_in[value of entry i], i.e. tiles_touched arrives in
// depth order with the final scatter, whose other work covers the gather's latency; the depth-ordered prefix then reads it
// coalesced.  (A separate gather kernel took 29 us at 3 M; carrying the payload through all three passes cost 37 us.)
template <typename K, int BITS, bool AUX>
__global__ void __launch_bounds__(SORT_THREADS) radix_scatter_kernel(
    const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, K* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int n, int shift, uint32_t dmin, int dbits, const uint32_t* __restrict__ tile_prefix,
    const uint32_t* __restrict__ chunk_prefix /* chunk_hist after radix_scan_kernel */, const uint32_t* __restrict__ digit_totals,
    int nblocks, int nchunks, const uint32_t* __restrict__ n_dev, int drop_dead, uint32_t* __restrict__ n_live_out,
    const uint32_t* __restrict__ aux_in, uint32_t* __restrict__ aux_out, int nbits /* significant bits of this pass's digit */,
    int chunk_tiles /* tiles per chunk of the count kernel that produced the tables */,
    int pack_shift /* > 0 with vals_in == nullptr and !AUX: the value of entry i is i | min(aux_in[i], tmax) << pack_shift */,
    uint2* __restrict__ ranges_out /* LAST pass of the tile-id sort: the range table (K9) is filled here, see below */,
    uint32_t* __restrict__ status, uint32_t* __restrict__ status_mirror /* resident mode, with ranges_out: the status words */,
    int write_keys /* 0: nobody reads the sorted keys of this (last) pass */) {
  const KeyMap km{dmin, dbits};
  const uint32_t dmask = (1u << nbits) - 1u;   // key bits at or above the sort's end_bit are not part of the order
  if (ranges_out != nullptr && status != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    // resident mode (what identify_tile_ranges_kernel does on the unfused path): overflow word and the host-mapped mirror.
    // status[0] = instances emitted (ordered_offsets_kernel), `n` = capacity, *n_dev = entries this pass sorts
    const uint32_t ntot = status[0];
    const uint32_t over = (ntot > (uint32_t)n || status[2] != 0u) ? 1u : 0u;   // status[2]: depth outside the key range (K1)
    status[2] = 0u;
    status[3] = over;
    const uint32_t live = min(n_dev ? *n_dev : ntot, min(ntot, (uint32_t)n));
    status[1] = live;
    if (status_mirror) { status_mirror[0] = ntot; status_mirror[1] = live; status_mirror[3] = over; }
  }
  if (n_dev) n = (int)min(*n_dev, (uint32_t)n);
  constexpr int NDIG = 1 << BITS;
  constexpr int DPT = NDIG / SORT_THREADS;   // consecutive digits per thread in the prefix step (1 or 2)
  static_assert(NDIG % SORT_THREADS == 0 && BITS <= 11, "digit | rank << BITS must fit the rank of 2048 keys");
  __shared__ K s_keys[SORT_TILE];
  __shared__ uint32_t s_vals[SORT_TILE];
  __shared__ uint32_t s_aux[AUX ? SORT_TILE : 1];
  __shared__ uint32_t cnt[4][NDIG];       // per-wave running digit counters, then per-wave bases
  __shared__ uint32_t local_start[NDIG];  // start of digit run inside the tile
  __shared__ int32_t gdelta[NDIG];        // global position - local position, per digit
  __shared__ uint32_t scan_tmp[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 4 * NDIG; i += SORT_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();

  // Which tile this workgroup sorts.  Workgroup ids are dealt round-robin to the 8 XCDs, and the runs that consecutive tiles
  // write for one digit are ADJACENT in the output (4-8 keys each): with tile = workgroup id the eight L2s each hold an
  // eighth of every output line and write it back as masked partial sectors.  Giving each XCD a contiguous range of tiles
  // lets its L2 complete the lines before they leave (ubench_sort_passes: 29 -> 21 us per 9-bit pass over 2.1 M keys,
  // 67 -> 46 us for the first tile-id pass over 7.9 M).  The grid is rounded up to a multiple of 8; surplus workgroups have
  // no tile.
  // The ranges are cut from the tiles that HOLD keys (n may come from the device and be well below the launch's capacity):
  // cut from the capacity they would leave the last XCDs idle.
#if defined(SEGS_MEASURE) && defined(SORT_TILE_IDENTITY)
  const int tile_id = (int)blockIdx.x;
#else
  const int tiles_in_use = (int)(((size_t)n + SORT_TILE - 1) / SORT_TILE);
  const int tiles_per_xcd = (tiles_in_use + 7) >> 3;
  const int tile_id = (int)(blockIdx.x & 7u) * tiles_per_xcd + (int)(blockIdx.x >> 3);
  if (tiles_in_use == 0 && n_live_out != nullptr && blockIdx.x == 0 && tid == 0) *n_live_out = 0u;   // nobody has tile 0 then
  if ((int)(blockIdx.x >> 3) >= tiles_per_xcd) return;
#endif
  if (tile_id >= nblocks) return;
  const size_t tile_base = (size_t)tile_id * SORT_TILE;
  const size_t wave_base = tile_base + (size_t)wv * (SORT_TILE / 4);
  const int nvalid = (size_t)n > tile_base ? (int)min((size_t)SORT_TILE, (size_t)n - tile_base) : 0;

  K key[SORT_ITEMS_PER_THREAD];
  uint32_t val[SORT_ITEMS_PER_THREAD];
  uint32_t aux[AUX ? SORT_ITEMS_PER_THREAD : 1];
  uint32_t drank[SORT_ITEMS_PER_THREAD];  // digit | wave-local rank << BITS ; 0xFFFFFFFF = invalid
  volatile uint32_t* my = cnt[wv];
#pragma unroll
  for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {
    const size_t i = wave_base + (size_t)r * 64 + lane;
    const bool valid = i < (size_t)n;
    key[r] = valid ? keys_in[i] : (K)0;
    val[r] = vals_in ? (valid ? vals_in[i] : 0u) : (uint32_t)i;
    if (AUX) aux[r] = valid ? aux_in[val[r]] : 0u;
    // first pass of an index sort with a small payload riding in the value's spare high bits: read here, where entry i IS
    // Gaussian i, it is a coalesced load; fetched after the sort it is a random 4-byte gather per entry
    if (!AUX && pack_shift > 0 && valid) val[r] |= min(aux_in[i], (0xFFFFFFFFu >> pack_shift)) << pack_shift;
  }
#pragma unroll
  for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {
    const size_t i = wave_base + (size_t)r * 64 + lane;
    const bool valid = i < (size_t)n && !(drop_dead && key[r] == (K)DEAD_KEY_OF(K));   // dead keys take no rank: dropped here
    const uint32_t d = valid ? (digit_of<BITS>(key[r], shift, km) & dmask) : 0u;
    const uint64_t vmask = __ballot(valid);
    const uint64_t peers = match_digit<BITS>(d, vmask, nbits);
    const uint32_t below = mbcnt(peers);
    uint32_t old = 0;
    if (valid) old = my[d];
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) my[d] = old + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    drank[r] = valid ? (d | ((old + below) << BITS)) : 0xFFFFFFFFu;
  }
  __syncthreads();

  // digits tid*DPT .. tid*DPT+DPT-1: per-wave exclusive bases, tile count, then exclusive scan over digits.
  uint32_t run[DPT], tot[DPT], thread_run = 0, thread_tot = 0;
#pragma unroll
  for (int j = 0; j < DPT; j++) {
    const int d = tid * DPT + j;
    uint32_t r = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t c = cnt[w][d]; cnt[w][d] = r; r += c; }
    run[j] = r; thread_run += r;
    tot[j] = digit_totals[d]; thread_tot += tot[j];
  }
  uint32_t x = thread_run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) scan_tmp[wv] = x;
  // global base of a digit = (sum of totals of smaller digits) + this tile's offset inside the digit row
  uint32_t tx = thread_tot;
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
  uint32_t lstart = lbase + x - thread_run, gstart = gbase + tx - thread_tot;
#pragma unroll
  for (int j = 0; j < DPT; j++) {
    const int d = tid * DPT + j;
    local_start[d] = lstart;
    // keys with digit d: in smaller digits' runs (gstart), in earlier chunks, in earlier tiles of this chunk
    gdelta[d] = (int32_t)(gstart + chunk_prefix[(size_t)d * nchunks + tile_id / chunk_tiles] +
                          tile_prefix[(size_t)tile_id * NDIG + d] - lstart);
    lstart += run[j]; gstart += tot[j];
  }
  // with dead keys dropped the tile holds fewer than nvalid entries, and the pass leaves sum(digit_totals) of them in all
  const int nout = drop_dead ? (int)(scan_tmp[0] + scan_tmp[1] + scan_tmp[2] + scan_tmp[3]) : nvalid;
  if (n_live_out != nullptr && tile_id == 0 && tid == 0) *n_live_out = tot_tmp[0] + tot_tmp[1] + tot_tmp[2] + tot_tmp[3];
  __syncthreads();

#pragma unroll
  for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {
    if (drank[r] != 0xFFFFFFFFu) {
      const uint32_t d = drank[r] & (uint32_t)(NDIG - 1);
      const uint32_t pos = local_start[d] + cnt[wv][d] + (drank[r] >> BITS);
      s_keys[pos] = key[r];
      s_vals[pos] = val[r];
      if (AUX) s_aux[pos] = aux[r];
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int r = 0; r < SORT_ITEMS_PER_THREAD; r++) {
    const int lp = r * SORT_THREADS + tid;
    if (lp < nout) {
      const K k = s_keys[lp];
#if defined(SEGS_MEASURE) && defined(ABLATE_SCATTER_LINEAR)   // measurement only (tools/ubench_sort_passes.hip): full-line stores, wrong order
      const size_t gp = tile_base + (size_t)lp + (size_t)(gdelta[digit_of<BITS>(k, shift, km) & dmask] & 0);
#else
      const size_t gp = (size_t)((int64_t)lp + (int64_t)gdelta[digit_of<BITS>(k, shift, km) & dmask]);
#endif
#if defined(SEGS_MEASURE) && defined(ABLATE_SCATTER_NO_STORE)
      if (k == (K)0x12345677 && gp == 77) keys_out[gp] = k;
      continue;
#endif
      if (write_keys) keys_out[gp] = k;
      vals_out[gp] = s_vals[lp];
      if (AUX) aux_out[gp] = s_aux[lp];
      if constexpr (sizeof(K) == 4) {
        // K9 fused into the last pass (rasterizer_impl.cu:116-138).  The input of an LSD pass is sorted on all lower bits, so
        // after the reorder by this pass's digit the tile's keys stand in LDS in full-key order: the neighbours in LDS tell
        // where a tile id starts and ends INSIDE this sort tile, and since a stable sort keeps the sort tiles' order per key,
        // the list of tile id t is [min over sort tiles of its first position, max of its last + 1).  Two integer atomics per
        // (sort tile, tile id) pair -- a sort tile of the last pass holds a few dozen distinct ids -- instead of a pass over
        // all sorted keys.  The table starts out as {0xFFFFFFFF, 0} per tile (K1 / make_depth_keys_kernel); an id without
        // instances keeps that, which every reader takes as empty (start >= end).
        if (ranges_out != nullptr) {
          const K prev = lp > 0 ? s_keys[lp - 1] : ~k, next = lp + 1 < nout ? s_keys[lp + 1] : ~k;
          if (k != prev) atomicMin(&ranges_out[k].x, (uint32_t)gp);
          if (k != next) atomicMax(&ranges_out[k].y, (uint32_t)gp + 1u);
        }
      }
    }
  }
}

#define SEGS_INSTANTIATE_RADIX(K, BITS)                                                                                          \
  template __global__ void radix_count_kernel<K, BITS>(const K*, int, int, uint32_t, int, uint32_t*, uint32_t*, int, int,       \
                                                       const uint32_t*, int, int, int);                                                  \
  template __global__ void radix_scatter_kernel<K, BITS, false>(const K*, const uint32_t*, K*, uint32_t*, int, int, uint32_t, int, \
                                                                const uint32_t*, const uint32_t*, const uint32_t*, int, int,        \
                                                                const uint32_t*, int, uint32_t*, const uint32_t*, uint32_t*, int, int, int, \
                                                                uint2*, uint32_t*, uint32_t*, int);
SEGS_INSTANTIATE_RADIX(uint64_t, 8)
SEGS_INSTANTIATE_RADIX(uint32_t, 8)
SEGS_INSTANTIATE_RADIX(uint32_t, 9)
SEGS_INSTANTIATE_RADIX(uint32_t, 11)
#undef SEGS_INSTANTIATE_RADIX
#define SEGS_INSTANTIATE_AUX(BITS)                                                                                                     \
  template __global__ void radix_scatter_kernel<uint32_t, BITS, true>(const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, int, int, \
                                                                      uint32_t, int, const uint32_t*, const uint32_t*, const uint32_t*, \
                                                                      int, int, const uint32_t*, int, uint32_t*, const uint32_t*, uint32_t*, int, int, int, \
                                                                      uint2*, uint32_t*, uint32_t*, int);
SEGS_INSTANTIATE_AUX(8)
SEGS_INSTANTIATE_AUX(9)
#undef SEGS_INSTANTIATE_AUX

// ---------------------------------------------------------------------------------------------
// K9 (rasterizer_impl.cu:116-138).  The range table is zeroed by make_depth_keys_kernel earlier in the same stream
// (rasterizer_impl.cu:310 uses a memset).  A per-tile binary search over the sorted ids instead of this R-sized pass was
// tried and is slower (44 dependent loads per tile: 17 us vs 9 us at R = 3.7 M).
__global__ void __launch_bounds__(256) identify_tile_ranges_kernel(int L, const uint32_t* __restrict__ keys,
                                                                   uint2* __restrict__ ranges, const uint32_t* __restrict__ n_dev,
                                                                   uint32_t* __restrict__ status, uint32_t* __restrict__ status_mirror,
                                                                   const uint32_t* __restrict__ n_live /* entries left after dead keys were dropped, or null */) {
  static_assert(RANGE_KEYS_PER_THREAD == 4, "one uint4 per thread");
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (n_dev) {
    const uint32_t n = *n_dev;
    if (idx == 0 && status) {   // resident mode: overflow word, and the status words mirrored into host-mapped memory
      // status[2]: preprocess_fwd_kernel saw a depth outside the resident sort's key range (see DEPTH_KEY_BITS)
      const uint32_t over = (n > (uint32_t)L || status[2] != 0u) ? 1u : 0u;
      status[2] = 0u;
      status[3] = over;
      // status[1]: instances left after the first tile-id pass dropped the dead ones (what the tile kernels walk)
      const uint32_t live = n_live ? min(*n_live, min(n, (uint32_t)L)) : min(n, (uint32_t)L);
      status[1] = live;
      if (status_mirror) { status_mirror[0] = n; status_mirror[1] = live; status_mirror[3] = over; }
    }
    L = (int)min(n, (uint32_t)L);
  }
  if (n_live) L = (int)min(*n_live, (uint32_t)L);
  // four consecutive keys per thread (one 16-byte load; the kernel is a pure stream over the sorted keys: one key per thread
  // took 14 us for 6.8 M keys)
  const int i0 = idx * RANGE_KEYS_PER_THREAD;
  if (i0 >= L) return;
  uint32_t k[RANGE_KEYS_PER_THREAD];
  if (i0 + RANGE_KEYS_PER_THREAD <= L) {
    const uint4 v = *reinterpret_cast<const uint4*>(keys + i0);
    k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
  } else {
#pragma unroll
    for (int q = 0; q < RANGE_KEYS_PER_THREAD; q++) k[q] = i0 + q < L ? keys[i0 + q] : 0u;
  }
  uint32_t prev = i0 > 0 ? keys[i0 - 1] : 0u;
#pragma unroll
  for (int q = 0; q < RANGE_KEYS_PER_THREAD; q++) {
    const int i = i0 + q;
    if (i < L) {
      const uint32_t cur = k[q];
      if (i == 0) ranges[cur].x = 0;
      else if (cur != prev) { ranges[prev].y = i; ranges[cur].x = i; }
      if (i == L - 1) ranges[cur].y = L;
      prev = cur;
    }
  }
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

// Depth-sort input: one (depth bits, idx) pair per Gaussian.  Culled Gaussians (no instances) get a 1 in the tile
// field, i.e. compacted key 1 << dbits, strictly behind every visible one when sorting dbits + 1 bits.
__global__ void __launch_bounds__(256) make_depth_keys_kernel(int P, const BinInfo* __restrict__ bin, uint32_t dcull,
                                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                              uint2* __restrict__ ranges, int num_tiles) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  for (int t = i; t < num_tiles; t += gridDim.x * 256) ranges[t] = make_uint2(RANGE_EMPTY_START, 0u);   // rasterizer_impl.cu:310 ({0,0} there)
  if (i >= P) return;
  const uint4 b = reinterpret_cast<const uint4*>(bin)[i];
  keys[i] = b.w ? b.x : dcull;   // culled Gaussians sort strictly after every visible one
  (void)vals;                    // the sort's first pass takes the index itself as the value
}
// Per-workgroup sums of tiles_touched taken in depth order (feeds ordered_offsets_kernel for the emitter).  `touched` is
// either already in depth order (`order` == nullptr: it rode along with the depth sort as the scatter's aux payload) or is
// gathered into depth order here ONCE and left in `sorted_touched` for ordered_offsets_kernel.
// pack_shift > 0: `order_rw` holds index | min(tiles_touched, tmax) << pack_shift (the payload rode through the depth sort in the
// values' spare bits, see radix_scatter_kernel); it is taken apart here -- the plain index goes back into order_rw for the
// emitter, a saturated count (rare: a Gaussian touching more tiles than the spare bits hold) is fetched from `touched`.
// (`touched` and `sorted_touched` are the SAME array on the gather-in-the-last-pass path, read and rewritten slot by slot: no
// __restrict__ on those two.)
__global__ void __launch_bounds__(256) ordered_block_sums_kernel(int P, const uint32_t* touched, const uint32_t* __restrict__ order,
                                                                 uint32_t* __restrict__ block_sums, uint32_t* sorted_touched,
                                                                 const uint32_t* __restrict__ ng_dev /* entries of `order` that are valid, or null = P */,
                                                                 uint32_t* __restrict__ order_rw, int pack_shift) {
  __shared__ uint32_t wave_sums[4];
  const int ng = ng_dev ? (int)min(*ng_dev, (uint32_t)P) : P;
  uint32_t s = 0;
#pragma unroll
  for (int r = 0; r < PREFIX_ROWS; r++) {   // PREFIX_ROWS rows of 256 slots per workgroup (ordered_offsets_kernel's cut)
    const int slot = (blockIdx.x * PREFIX_ROWS + r) * 256 + threadIdx.x;
    uint32_t v = 0u;
    if (slot < ng) {
      if (pack_shift > 0) {
        const uint32_t packed = order_rw[slot], idx = packed & ((1u << pack_shift) - 1u);
        v = packed >> pack_shift;
        if (v == (0xFFFFFFFFu >> pack_shift)) v = touched[idx];
        order_rw[slot] = idx;
      } else {
        v = order ? touched[order[slot]] : touched[slot];
      }
    }
    if (slot < P) sorted_touched[slot] = v;
    s += v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
}
// GeometryState::point_offsets (inclusive scan of tiles_touched in index order, rasterizer_impl.cu:276-277).  The
// pipeline itself no longer needs it (instances are emitted in depth order); produced on request for parity checks.
__global__ void __launch_bounds__(1024) point_offsets_kernel(int P, const BinInfo* __restrict__ bin, uint32_t* __restrict__ offsets) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < P; base += 1024) {
    const int i = base + tid;
    uint32_t x = i < P ? bin[i].tiles_touched : 0u;
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
    if (i < P) offsets[i] = carry + wbase + x;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + x;
    __syncthreads();
  }
}

// Range table as the reference defines it: tiles without instances read {0, 0} (kernels.h RANGE_EMPTY_START).
__global__ void __launch_bounds__(256) normalize_ranges_kernel(int num_tiles, uint2* __restrict__ ranges) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < num_tiles && ranges[t].y == 0u) ranges[t].x = 0u;
}

// point_list as the reference defines it: strip the quadrant mask from the sorted values.
__global__ void __launch_bounds__(256) strip_mask_kernel(int n, const uint32_t* __restrict__ vals, uint32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = vals[i] & ID_MASK;
}

}  // namespace segs

namespace segs {
// Test support: the reference's sorted 64-bit keys (tile << 32 | depth bits, rasterizer_impl.cu:100-104) rebuilt from the
// sorted tile ids and the depth of each instance's Gaussian.
__global__ void __launch_bounds__(256) rebuild_keys_kernel(int R, const uint32_t* __restrict__ tile_keys, const uint32_t* __restrict__ vals,
                                                           const BinInfo* __restrict__ bin, uint64_t* __restrict__ keys64) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  keys64[i] = ((uint64_t)tile_keys[i] << 32) | (uint64_t)bin[vals[i] & ID_MASK].depth_bits;
}
}  // namespace segs
