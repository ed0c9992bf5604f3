// densify.hip -- anchor statistics and one level of anchor growing on the device (include/segs_densify.h).
// Reference: GaussianModel::training_statis / anchor_growing, src/gaussian_model.cpp:1459-1503, 1559-1699.
//
//   stats_kernel            : thread = anchor; the four accumulators of training_statis in the candidate domain
//   select_candidates_kernel: thread = candidate slot; candidate test, xyz = anchor + offset * exp(scaling[:3]),
//                             voxel = round(xyz / cur_size), packed 63-bit key, wave-aggregated append
//   anchor_keys_kernel      : key of round(anchor / cur_size) for every existing anchor
//   (segs_sort_pairs)       : stable LSD radix sort of both key sets -> at::unique_dim's lexicographic order
//   survive_flags_kernel    : first of each run of equal candidate keys that is not found (binary search) among the
//                             sorted anchor keys  == selected_grid_coords_unique[~remove_duplicates]
//   scan (3 small kernels)  : exclusive scan of the flags -> row of every new anchor
//   emit_new_anchors_kernel : 32 lanes per new anchor: coords * cur_size, per-feature max over the voxel's run of parents
// Built with -ffp-contract=off: xyz and the quantisation follow the reference's separate multiply / add / divide.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/segs_densify.h"
#include "../../include/segs_raster.h"
#include "kernels.h"

namespace {

constexpr int KEY_BIAS = 1 << 20;   // voxel coordinates in [-2^20, 2^20): |p| < 1 km at the finest voxel of 1 mm

__device__ __forceinline__ uint64_t pack_key(int x, int y, int z) {
  const int lo = -KEY_BIAS, hi = KEY_BIAS - 1;
  x = min(max(x, lo), hi); y = min(max(y, lo), hi); z = min(max(z, lo), hi);
  return ((uint64_t)(x + KEY_BIAS) << 42) | ((uint64_t)(y + KEY_BIAS) << 21) | (uint64_t)(z + KEY_BIAS);
}

// thread = candidate slot (coalesced accesses: with a thread per anchor every load instruction of a wave touched 64-120
// separate sectors and the kernel was bound by that, 15 us at 50 k anchors), 256 / no anchors per workgroup; the per-anchor sum of
// the clamped opacities goes through LDS.  ONE memory round trip deep: the guard word, the visibility and everything a candidate
// may need are requested together and unconditionally, the conditions only gate the stores.
__global__ void __launch_bounds__(256) stats_kernel(int A, int no, const float* __restrict__ nop, const int* __restrict__ vis,
                                                    const int* __restrict__ radii, const float* __restrict__ g2d,
                                                    float* __restrict__ opacity_accum, float* __restrict__ anchor_demon,
                                                    float* __restrict__ grad_accum, float* __restrict__ denom,
                                                    const uint32_t* __restrict__ skip_flag) {
  __shared__ float clamped[256];
  const int apw = 256 / no;                             // anchors per workgroup
  const int tid = threadIdx.x;
  const int la = tid / no;                              // local anchor of this thread's candidate
  const int a = blockIdx.x * apw + la;
  const bool in = la < apw && a < A;
  const size_t c = (size_t)blockIdx.x * apw * no + tid; // = a * no + k
  // guarded form: a pass the resident rasterizer flagged as overflowed must not enter the statistics (its radii and
  // gradients are meaningless); dropped here on the device, like the optimizer step (segs_adam_step_guarded)
  const uint32_t skip = skip_flag != nullptr ? *skip_flag : 0u;
  const int visible = in ? (vis ? vis[a] : 1) : 0;
  const float op = in ? nop[c] : 0.f;
  const int rad = in ? radii[c] : 0;
  const float gx = in ? g2d[c * 3] : 0.f, gy = in ? g2d[c * 3 + 1] : 0.f;
  const float ga = in ? grad_accum[c] : 0.f, dn = in ? denom[c] : 0.f;
  const int a2 = blockIdx.x * apw + tid;                // the anchor whose sums this thread writes (tid < apw)
  const bool owner = tid < apw && a2 < A;
  const float oa = owner ? opacity_accum[a2] : 0.f, ad = owner ? anchor_demon[a2] : 0.f;
  const int vis2 = owner ? (vis ? vis[a2] : 1) : 0;
  clamped[tid] = op < 0.f ? 0.f : op;                   // :1467
  __syncthreads();
  if (skip != 0u) return;
  if (in && visible > 0 && op > 0.f && rad > 0) {       // anchor_visible_mask (:1471-1478), offset_selection_mask && update_filter (:1481-1484)
    grad_accum[c] = ga + sqrtf(gx * gx + gy * gy);      // :1494-1499
    denom[c] = dn + 1.f;
  }
  if (owner && vis2 > 0) {
    float s = 0.f;
    for (int k = 0; k < no; k++) s += clamped[tid * no + k];   // in candidate order, like the reference's sum over dim 1
    opacity_accum[a2] = oa + s;
    anchor_demon[a2] = ad + 1.f;
  }
}

__global__ void __launch_bounds__(256) select_candidates_kernel(int n_slots, int no, const float* __restrict__ anchor,
                                                                const float* __restrict__ offset, const float* __restrict__ scaling_log,
                                                                const float* __restrict__ grads, const uint8_t* __restrict__ offset_mask,
                                                                const float* __restrict__ rnd, float threshold, float rand_threshold,
                                                                float cur_size, uint32_t* __restrict__ count, uint64_t* __restrict__ keys,
                                                                uint32_t* __restrict__ parents) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  bool cand = false;
  uint64_t key = 0;
  uint32_t a = 0;
  if (c < n_slots) {
    cand = grads[c] >= threshold && offset_mask[c] != 0 && rnd[c] > rand_threshold;   // :1565-1570
    if (cand) {
      a = (uint32_t)(c / no);
      int g[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const float xyz = anchor[(size_t)a * 3 + d] + offset[(size_t)c * 3 + d] * expf(scaling_log[(size_t)a * 6 + d]);   // :1583-1584
        g[d] = (int)rintf(xyz / cur_size);                                                                              // :1592
      }
      key = pack_key(g[0], g[1], g[2]);
    }
  }
  const uint64_t m = __ballot(cand);
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == 0 && m) base = atomicAdd(count, (uint32_t)__popcll(m));
  base = __shfl(base, 0, 64);
  if (cand) {
    const uint32_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
    keys[pos] = key;
    parents[pos] = a;
  }
}

__global__ void __launch_bounds__(256) anchor_keys_kernel(int A, const float* __restrict__ anchor, float cur_size,
                                                          uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= A) return;
  const int x = (int)rintf(anchor[(size_t)a * 3] / cur_size), y = (int)rintf(anchor[(size_t)a * 3 + 1] / cur_size),
            z = (int)rintf(anchor[(size_t)a * 3 + 2] / cur_size);                                                    // :1589
  keys[a] = pack_key(x, y, z);
  vals[a] = (uint32_t)a;
}

__global__ void __launch_bounds__(256) survive_flags_kernel(int n, const uint64_t* __restrict__ ckeys, int A,
                                                            const uint64_t* __restrict__ akeys, uint32_t* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = ckeys[i];
  uint32_t f = (i == 0 || ckeys[i - 1] != k) ? 1u : 0u;     // head of its run: one row of unique_dim
  if (f) {
    int lo = 0, hi = A;                                      // any anchor in this voxel? (:1601-1620)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (akeys[mid] < k) lo = mid + 1; else hi = mid;
    }
    if (lo < A && akeys[lo] == k) f = 0u;
  }
  flags[i] = f;
}

// ---- exclusive scan of n u32 values (n up to a few million; rare call, simplicity over speed)
constexpr int SCAN_TILE = 1024;
__global__ void __launch_bounds__(256) scan_block_sums(int n, const uint32_t* __restrict__ in, uint32_t* __restrict__ sums) {
  __shared__ uint32_t w[4];
  const int base = blockIdx.x * SCAN_TILE;
  uint32_t s = 0;
  for (int i = threadIdx.x; i < SCAN_TILE; i += 256) s += (base + i < n) ? in[base + i] : 0u;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}
// exclusive scan of the per-tile sums by ONE workgroup: 1024 entries per round, wave scans + a carry (n candidates / 1024
// tiles: 2 900 entries for 3 M candidate slots; a single thread walked them one dependent load at a time)
__global__ void __launch_bounds__(1024) scan_sums_kernel(int nblocks, uint32_t* __restrict__ sums, int* __restrict__ total) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + tid;
    const uint32_t v = i < nblocks ? sums[i] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wv; w++) wbase += wave_tot[w];
    const uint32_t carry = carry_s;
    if (i < nblocks) sums[i] = carry + wbase + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + x;
    __syncthreads();
  }
  if (tid == 0) *total = (int)carry_s;
}
__global__ void __launch_bounds__(256) scan_apply(int n, const uint32_t* __restrict__ in, const uint32_t* __restrict__ sums,
                                                  uint32_t* __restrict__ out) {
  __shared__ uint32_t part[256];
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  uint32_t v[4], s = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) { v[q] = (base + q < n) ? in[base + q] : 0u; s += v[q]; }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {    // Hillis-Steele over the 256 per-thread sums
    const uint32_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = sums[blockIdx.x] + part[threadIdx.x] - s;
#pragma unroll
  for (int q = 0; q < 4; q++) { if (base + q < n) out[base + q] = run; run += v[q]; }
}

__global__ void __launch_bounds__(256) compact_heads_kernel(int n, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ excl,
                                                            uint32_t* __restrict__ head_pos) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && flags[i]) head_pos[excl[i]] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) emit_new_anchors_kernel(int n, const int* __restrict__ n_new_dev, int max_new,
                                                               const uint32_t* __restrict__ head_pos, const uint64_t* __restrict__ ckeys,
                                                               const uint32_t* __restrict__ parents, const float* __restrict__ anchor_feat,
                                                               float cur_size, float* __restrict__ new_anchor, float* __restrict__ new_feat) {
  const int v = blockIdx.x * 8 + (threadIdx.x >> 5);       // 32 lanes (features) per new anchor
  const int f = threadIdx.x & 31;
  const int n_new = min(*n_new_dev, max_new);
  if (v >= n_new) return;
  const uint32_t i0 = head_pos[v];
  const uint64_t k = ckeys[i0];
  if (f < 3) {
    const int g = (int)((k >> (42 - 21 * f)) & 0x1FFFFFu) - KEY_BIAS;
    new_anchor[(size_t)v * 3 + f] = (float)g * cur_size;   // selected_grid_coords_unique * cur_size (:1621)
  }
  float mx = -INFINITY;                                     // scatter_max over the voxel's candidates (:1632-1637)
  for (uint32_t i = i0; i < (uint32_t)n && ckeys[i] == k; i++) mx = fmaxf(mx, anchor_feat[(size_t)parents[i] * 32 + f]);
  new_feat[(size_t)v * 32 + f] = mx;
}

struct GrowTemp {
  uint32_t* count;      // [0] candidates, [1] n_new (int)
  uint64_t *ckeys_in, *ckeys, *akeys_in, *akeys;
  uint32_t *cvals_in, *cvals, *avals_in, *avals, *flags, *excl, *sums, *head_pos;
  char* sort_temp;
};
size_t grow_carve(int A, int nc, char* base, GrowTemp* t) {
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
  const size_t nA = (size_t)(A > 0 ? A : 1), nC = (size_t)(nc > 0 ? nc : 1);
  const size_t o_count = take(256), o_ck0 = take(nC * 8), o_ck1 = take(nC * 8), o_ak0 = take(nA * 8), o_ak1 = take(nA * 8),
               o_cv0 = take(nC * 4), o_cv1 = take(nC * 4), o_av0 = take(nA * 4), o_av1 = take(nA * 4), o_fl = take(nC * 4),
               o_ex = take(nC * 4), o_su = take((nC / SCAN_TILE + 2) * 4), o_hp = take(nC * 4),
               o_sort = take(segs_binning_bytes((int)(nA > nC ? nA : nC)) + 256);
  if (t) {
    t->count = (uint32_t*)(base + o_count);
    t->ckeys_in = (uint64_t*)(base + o_ck0); t->ckeys = (uint64_t*)(base + o_ck1);
    t->akeys_in = (uint64_t*)(base + o_ak0); t->akeys = (uint64_t*)(base + o_ak1);
    t->cvals_in = (uint32_t*)(base + o_cv0); t->cvals = (uint32_t*)(base + o_cv1);
    t->avals_in = (uint32_t*)(base + o_av0); t->avals = (uint32_t*)(base + o_av1);
    t->flags = (uint32_t*)(base + o_fl); t->excl = (uint32_t*)(base + o_ex); t->sums = (uint32_t*)(base + o_su);
    t->head_pos = (uint32_t*)(base + o_hp); t->sort_temp = base + o_sort;
  }
  return off;
}

}  // namespace

extern "C" {

int segs_training_statis_guarded(int A, int n_offsets, const float* neural_opacity, const int* visible_radii, const int* radii,
                                 const float* dL_dmean2D, float* opacity_accum, float* anchor_demon, float* offset_gradient_accum,
                                 float* offset_denom, const uint32_t* skip_flag, void* stream) {
  if (A < 0 || n_offsets <= 0 || n_offsets > 16) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size; n_offsets <= 16)");
  if (A == 0) return SEGS_OK;
  if (!neural_opacity || !radii || !dL_dmean2D || !opacity_accum || !anchor_demon || !offset_gradient_accum || !offset_denom)
    return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  const int apw = 256 / n_offsets;
  stats_kernel<<<(A + apw - 1) / apw, 256, 0, (hipStream_t)stream>>>(A, n_offsets, neural_opacity, visible_radii, radii, dL_dmean2D,
                                                                 opacity_accum, anchor_demon, offset_gradient_accum, offset_denom,
                                                                 skip_flag);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

int segs_training_statis(int A, int n_offsets, const float* neural_opacity, const int* visible_radii, const int* radii,
                         const float* dL_dmean2D, float* opacity_accum, float* anchor_demon, float* offset_gradient_accum,
                         float* offset_denom, void* stream) {
  return segs_training_statis_guarded(A, n_offsets, neural_opacity, visible_radii, radii, dL_dmean2D, opacity_accum, anchor_demon,
                                      offset_gradient_accum, offset_denom, nullptr, stream);
}

size_t segs_anchor_growing_temp_bytes(int A, int n_candidates) {
  if (A < 0 || n_candidates < 0) return 0;
  return grow_carve(A, n_candidates, nullptr, nullptr);
}

int segs_anchor_growing_level(int A, int A_init, int n_offsets, int feat_dim, const float* anchor, const float* offset,
                              const float* scaling_log, const float* anchor_feat, const float* grads, const uint8_t* offset_mask,
                              const float* rnd, float threshold, float rand_threshold, float cur_size, int max_new,
                              float* new_anchor, float* new_feat, int* n_new, char* temp, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (feat_dim != 32) return segs::set_error(SEGS_ERR_UNSUPPORTED, "unsupported configuration (feat_dim must be 32, n_offsets 10, appearance_dim <= 64)");
  if (A <= 0 || A_init <= 0 || A_init > A || n_offsets <= 0 || max_new < 0 || !(cur_size > 0.f)) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  if (!anchor || !offset || !scaling_log || !anchor_feat || !grads || !offset_mask || !rnd || !new_anchor || !new_feat || !n_new || !temp)
    return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  const int n_slots = A_init * n_offsets;
  GrowTemp T;
  grow_carve(A, n_slots, temp, &T);
  hipError_t e = hipMemsetAsync(T.count, 0, 8, st);
  if (e != hipSuccess) return segs::set_hip_error(e, __func__);
  e = hipMemsetAsync(n_new, 0, sizeof(int), st);
  if (e != hipSuccess) return segs::set_hip_error(e, __func__);
  select_candidates_kernel<<<(n_slots + 255) / 256, 256, 0, st>>>(n_slots, n_offsets, anchor, offset, scaling_log, grads, offset_mask,
                                                                   rnd, threshold, rand_threshold, cur_size, T.count, T.ckeys_in,
                                                                   T.cvals_in);
  anchor_keys_kernel<<<(A + 255) / 256, 256, 0, st>>>(A, anchor, cur_size, T.akeys_in, T.avals_in);
  uint32_t n_cand = 0;   // the one host synchronisation of this (rare) call: the sort and the launches are sized by it
  e = hipMemcpyAsync(&n_cand, T.count, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return segs::set_hip_error(e, __func__);
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return segs::set_hip_error(e, __func__);
  if (n_cand == 0) return SEGS_OK;
  const int n = (int)n_cand;
  int rc = segs_sort_pairs(T.ckeys_in, T.cvals_in, T.ckeys, T.cvals, n, 63, T.sort_temp, stream);
  if (rc) return rc;
  rc = segs_sort_pairs(T.akeys_in, T.avals_in, T.akeys, T.avals, A, 63, T.sort_temp, stream);
  if (rc) return rc;
  survive_flags_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, T.ckeys, A, T.akeys, T.flags);
  const int nblk = (n + SCAN_TILE - 1) / SCAN_TILE;
  scan_block_sums<<<nblk, 256, 0, st>>>(n, T.flags, T.sums);
  scan_sums_kernel<<<1, 1024, 0, st>>>(nblk, T.sums, n_new);
  scan_apply<<<nblk, 256, 0, st>>>(n, T.flags, T.sums, T.excl);
  compact_heads_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, T.flags, T.excl, T.head_pos);
  const int cap = n < max_new ? n : max_new;
  if (cap > 0)
    emit_new_anchors_kernel<<<(cap + 7) / 8, 256, 0, st>>>(n, n_new, max_new, T.head_pos, T.ckeys, T.cvals, anchor_feat, cur_size,
                                                           new_anchor, new_feat);
  e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

}  // extern "C"
