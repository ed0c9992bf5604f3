// preprocess.hip -- per-Gaussian kernels of the rasterizer (gfx950 / MI355X).
//
//   preprocess_fwd_kernel   K1  reference: preprocessCUDA          cuda_rasterizer/forward.cu:155-256
//   visible_filter_kernel   K2  reference: filter_preprocessCUDA   cuda_rasterizer/forward.cu:259-334
//   mark_visible_kernel     K4  reference: checkFrustum            cuda_rasterizer/rasterizer_impl.cu:54-66
//   preprocess_bwd_kernel   K12+K13 fused; reference: computeCov2DCUDA backward.cu:144-274,
//                           preprocessCUDA backward.cu:346-396, computeCov3D backward.cu:278-341
//
// All are HBM-bound streaming kernels (one Gaussian per lane, 256-thread workgroups):
//  * the (P,3) AoS inputs are fetched as three fully coalesced dword sweeps per workgroup and
//    transposed through LDS (stride-3 LDS reads are conflict free), rotations as 16 B/lane;
//  * everything the tile kernels need is emitted as ONE 64-byte record per Gaussian (gs_layout.h);
//  * the per-workgroup sum of tiles_touched is produced here so the prefix sum needs no extra pass
//    over the per-Gaussian data;
//  * the backward fuses the reference's two kernels and its 108 B/Gaussian of zero-fills (K14): every
//    output row is written exactly once (zeros for culled Gaussians), cov3D is recomputed instead of
//    being stored and re-read.
//
// Arithmetic: this translation unit is built with -ffp-contract=off and evaluates every expression
// in the reference's order with one binary32 rounding per operation, so that radii, rects, depths
// (hence tile/sort keys) are bit-identical to the CPU oracle.  glm semantics: mat3 is column-major,
// m[c][r] = column c, row r.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gs_layout.h"
#include "kernels.h"
#include "project_gaussian.h"
#include "sh_color.h"

#pragma clang fp contract(off)

namespace segs {

// Workgroup-coalesced fetch of row `tid` of a (P,3) float array: three dword sweeps + LDS transpose.
__device__ __forceinline__ float3 load_row3(const float* __restrict__ a, int P, float* lds /*768 floats*/) {
  const int tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 768;
  const size_t lim = (size_t)P * 3;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const size_t i = base + k * 256 + tid;
    lds[k * 256 + tid] = i < lim ? a[i] : 0.f;
  }
  __syncthreads();
  float3 v = make_float3(lds[3 * tid], lds[3 * tid + 1], lds[3 * tid + 2]);
  __syncthreads();
  return v;
}

// Up to three (P,3) arrays at once: all nine dword sweeps are in flight before the first wait and ONE barrier pair covers
// them.  Three load_row3 calls in a row were three dependent memory round trips per wave, and the per-Gaussian kernels run
// only a few waves per SIMD, so that latency was not hidden: 131 -> 108 us (forward) and 153 -> 122 us (backward) at 3 M
// Gaussians.  (Hoisting the remaining per-lane loads -- opacity, radii, rotation -- the same way changed nothing.)
// lds: 768 floats per array; null arrays are skipped.
__device__ __forceinline__ void load_rows3(const float* __restrict__ a0, const float* __restrict__ a1, const float* __restrict__ a2,
                                           int P, float* lds /*3 x 768 floats*/, float3& v0, float3& v1, float3& v2) {
  const int tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 768;
  const size_t lim = (size_t)P * 3;
  const float* arr[3] = {a0, a1, a2};
  float t[3][3];
#pragma unroll
  for (int m = 0; m < 3; m++)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const size_t i = base + k * 256 + tid;
      t[m][k] = (arr[m] && i < lim) ? arr[m][i] : 0.f;
    }
#pragma unroll
  for (int m = 0; m < 3; m++)
#pragma unroll
    for (int k = 0; k < 3; k++) lds[m * 768 + k * 256 + tid] = t[m][k];
  __syncthreads();
  v0 = make_float3(lds[3 * tid], lds[3 * tid + 1], lds[3 * tid + 2]);
  v1 = make_float3(lds[768 + 3 * tid], lds[768 + 3 * tid + 1], lds[768 + 3 * tid + 2]);
  v2 = make_float3(lds[1536 + 3 * tid], lds[1536 + 3 * tid + 1], lds[1536 + 3 * tid + 2]);
  __syncthreads();
}

__global__ void __launch_bounds__(256) preprocess_fwd_kernel(
    int P, const float* __restrict__ means3D, const float* __restrict__ scales, float mod,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ colors,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y, uint32_t gx, uint32_t gy,
    int* __restrict__ radii, float* __restrict__ rec, BinInfo* __restrict__ bin, uint32_t* __restrict__ block_sums,
    uint32_t* __restrict__ depth_range /* per workgroup: [b] = max(depth_bits), [nblocks + b] = max(~depth_bits), visible only */,
    const float* __restrict__ shs, int D, int M, const float* __restrict__ cam_pos, uint32_t* __restrict__ clamped,
    uint32_t flags, uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ depth_vals, uint2* __restrict__ ranges,
    int num_tiles, uint32_t* __restrict__ depth_overflow /* resident: set when a binned depth leaves the 27-bit key range */,
    uint32_t* __restrict__ touched_dense) {
  __shared__ float lds[3 * 768];
  __shared__ uint32_t wave_sums[4], wave_dmax[4], wave_dnmin[4];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  float4 rot = make_float4(0, 0, 0, 0);
  if (scales && idx < P) rot = reinterpret_cast<const float4*>(rotations)[idx];   // in flight together with the sweeps below
  float3 p, sc, col;
  load_rows3(means3D, scales, colors, P, lds, p, sc, col);

  uint32_t touched = 0, dbits_mine = 0;
  float4 rq0 = make_float4(0, 0, 0, 0), rq1 = rq0, rq2 = rq0, rq3 = rq0;   // this Gaussian's record, stored below
  bool has_record = false;
  if (idx < P) {
    Projected g = project_gaussian(p, sc, mod, rot, cov3D_precomp ? cov3D_precomp + (size_t)6 * idx : nullptr,
                                   viewmatrix, projmatrix, W, H, tan_fovx, tan_fovy, focal_x, focal_y, gx, gy);
    // SEGS_RASTER_SKIP_NONPOSITIVE_OPACITY: such a Gaussian can never pass alpha >= 1/255; dropping it here equals
    // the reference's compaction of masked-out neural Gaussians before the rasterizer (gaussian_renderer.cpp:320)
    if ((flags & 1u) && !(opacities[idx] > 0.f)) g.radius = 0;
    BinInfo b{0u, 0u, 0u, 0u};
    if (g.radius > 0) {
      const float op = opacities[idx];
      if (!colors) {  // forward.cu:241-247
        uint32_t cb;
        col = sh::to_rgb(idx, D, M, p, cam_pos, shs, &cb);
        clamped[idx] = cb;
      }
      const BinnedRecord br = make_record(g, op, col, flags);
      touched = br.touched;
      b.depth_bits = br.depth_bits;
      dbits_mine = b.depth_bits;
      b.rect_min = br.rect_min;
      b.rect_max = br.rect_max;
      b.tiles_touched = touched;
      rq0 = br.q0; rq1 = br.q1; rq2 = br.q2; rq3 = br.q3;
      has_record = true;
    }
    radii[idx] = g.radius;
    // BinInfo feeds make_depth_keys_kernel and the parity-test unpackers of the reference-shaped path; the resident path
    // (depth keys written right here) reads it nowhere: 16 B x P saved per step
    if (!depth_keys) reinterpret_cast<uint4*>(bin)[idx] = make_uint4(b.depth_bits, b.rect_min, b.rect_max, b.tiles_touched);
    touched_dense[idx] = b.tiles_touched;
    if (depth_keys) {   // resident mode: what make_depth_keys_kernel would write (32-bit depth keys, culled = 0xFFFFFFFF)
      // The resident depth sort looks at DEPTH_KEY_BITS bits above the near plane's bit pattern (three 9-bit passes).
      // Gaussians without instances carry the all-ones key: the FIRST pass gives them no histogram count and no rank, so the
      // later passes, the depth-ordered prefix and the emitter's searches only see the binned ones (30 % fewer at 3 M).  A
      // binned depth beyond the key range -- 0.2 * 2^16 = 13 107 -- flags the step; the host redoes it through the exact path.
      depth_keys[idx] = b.tiles_touched ? b.depth_bits : 0xFFFFFFFFu;
      (void)depth_vals;   // the sort's first pass takes the index itself as the value (no iota array)
      if (depth_overflow && b.tiles_touched && (b.depth_bits - DEPTH_KEY_MIN) >= ((1u << DEPTH_KEY_BITS) - 1u)) *depth_overflow = 1u;
    }
  }
  // The records leave through a per-wave LDS transpose: a lane storing its own 64 bytes as four float4 makes every store
  // instruction touch 64 different lines, a quarter of each; transposed, an instruction writes sixteen whole records (1 KB
  // contiguous), 32 records of the wave at a time in the (by now free) staging buffer of the input sweeps.
  {
    const int lane = threadIdx.x & 63;
    float* const wl = lds + (threadIdx.x >> 6) * 576;   // 32 records x 16 dwords of this wave (576 = 2304 / 4)
    const uint64_t recs = __ballot(has_record);
    const size_t wave_first = (size_t)blockIdx.x * 256 + (threadIdx.x & ~63);
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if ((lane >> 5) == half) {
        float4* mine = reinterpret_cast<float4*>(wl + (lane & 31) * REC_DWORDS);
        mine[0] = rq0; mine[1] = rq1; mine[2] = rq2; mine[3] = rq3;
      }
      asm volatile("" ::: "memory");   // (a wave's LDS instructions complete in order)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int r = 16 * j + (lane >> 2);        // record of this half the lane carries a quarter of
        const int owner = 32 * half + r;
        if ((recs >> owner) & 1ull)
          reinterpret_cast<float4*>(rec + (wave_first + owner) * REC_DWORDS)[lane & 3] = reinterpret_cast<const float4*>(wl + r * REC_DWORDS)[lane & 3];
      }
      asm volatile("" ::: "memory");
    }
  }
  if (ranges)
    for (int t = idx; t < num_tiles; t += gridDim.x * 256) ranges[t] = make_uint2(RANGE_EMPTY_START, 0u);   // rasterizer_impl.cu:310 ({0,0} there; see kernels.h)
  // workgroup sum of tiles_touched -> block_sums[blockIdx] (feeds the prefix sum, K5); depth range for the sort
  uint32_t s = touched;
  uint32_t dmax = touched ? dbits_mine : 0u, dnmin = touched ? ~dbits_mine : 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    dmax = max(dmax, (uint32_t)__shfl_down((int)dmax, off, 64));
    dnmin = max(dnmin, (uint32_t)__shfl_down((int)dnmin, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    wave_sums[threadIdx.x >> 6] = s; wave_dmax[threadIdx.x >> 6] = dmax; wave_dnmin[threadIdx.x >> 6] = dnmin;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    block_sums[blockIdx.x] = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
    depth_range[blockIdx.x] = max(max(wave_dmax[0], wave_dmax[1]), max(wave_dmax[2], wave_dmax[3]));
    depth_range[gridDim.x + blockIdx.x] = max(max(wave_dnmin[0], wave_dnmin[1]), max(wave_dnmin[2], wave_dnmin[3]));
  }
}

__global__ void __launch_bounds__(256) visible_filter_kernel(
    int P, const float* __restrict__ means3D, const float* __restrict__ scales, float mod,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, int W, int H,
    float tan_fovx, float tan_fovy, float focal_x, float focal_y, uint32_t gx, uint32_t gy, int* __restrict__ radii,
    int log_scale_stride /* > 0: `scales` holds LOG-scales in rows of this many floats and the first three are used through
    exp(): prefilter_voxel's get_scaling()[:, :3] (src/gaussian_renderer.cpp:150-152) without the intermediate tensor */) {
  __shared__ float lds[768];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const float3 p = load_row3(means3D, P, lds);
  float3 sc = make_float3(0, 0, 0);
  float4 rot = make_float4(0, 0, 0, 0);
  if (scales) {
    if (log_scale_stride > 0) {
      if (idx < P) {
        const float* r = scales + (size_t)idx * log_scale_stride;
        sc = make_float3(expf(r[0]), expf(r[1]), expf(r[2]));
      }
    } else {
      sc = load_row3(scales, P, lds);
    }
    if (idx < P) rot = reinterpret_cast<const float4*>(rotations)[idx];
  }
  if (idx >= P) return;
  Projected g = project_gaussian(p, sc, mod, rot, cov3D_precomp ? cov3D_precomp + (size_t)6 * idx : nullptr,
                                 viewmatrix, projmatrix, W, H, tan_fovx, tan_fovy, focal_x, focal_y, gx, gy);
  radii[idx] = g.radius;
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ viewmatrix, uint8_t* __restrict__ present) {
  __shared__ float lds[768];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const float3 p = load_row3(means3D, P, lds);
  if (idx >= P) return;
  float3 p_view = transformPoint4x3(p, viewmatrix);
  present[idx] = (p_view.z <= 0.2f) ? 0 : 1;
}

// Workgroup-coalesced store of row `tid` of a (P,3) float array: LDS transpose + three dword sweeps (the mirror of
// load_row3; per-lane 12-byte rows written as three strided dwords filled only a third of every store instruction).
__device__ __forceinline__ void store_row3(float* __restrict__ a, int P, float* lds /*768 floats*/, float x, float y, float z) {
  const int tid = threadIdx.x;
  lds[3 * tid] = x; lds[3 * tid + 1] = y; lds[3 * tid + 2] = z;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * 768;
  const size_t lim = (size_t)P * 3;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const size_t i = base + k * 256 + tid;
    if (i < lim) a[i] = lds[k * 256 + tid];
  }
  __syncthreads();
}

// ---- Backward of the per-Gaussian stage: the reference's computeCov2DCUDA + preprocessCUDA + computeCov3D backward
// (backward.cu:144-274, 346-396, 278-341) as ONE pass, derived here from the forward map rather than transcribed:
//
//   t  = Wv p + tv                        view space; t.xy/t.z clamped to +-1.3 tan(fov/2)      (forward.cu:74-113)
//   J  = [[fx/tz, 0, -fx tx/tz^2], [0, fy/tz, -fy ty/tz^2]]                   M2 = J Wv   (2x3)
//   L  = R(q) diag(s),  s = mod * scale,  Sigma = L L^T                       U  = M2 L   (2x3)
//   C  = U U^T + 0.3 I  = [[a, b], [b, c]]                                    conic Q = adj(C) / det
//
// Given the tile kernel's  G = [[gA, gB], [gB, gC]]  (dL/dconic with the off-diagonal stored once, backward.cu:545-549):
//   Dc    = dL/dC = -k adj(C) G adj(C),   k = 1 / (det^2 + 1e-7)             (the reference's guarded 1/det^2, :205)
//   dL/dSigma = M2^T Dc M2                 -> dL_dcov3D (off-diagonals doubled: six unique entries)
//   dL/dM2    = 2 Dc (U L^T)               -> dL/dJ = dL/dM2 Wv^T -> dL/dt  (a clamped coordinate passes nothing, :175-176)
//   dL/dL     = 2 M2^T (Dc U)              -> dL/ds_k = sum_i R_ik dL/dL_ik  (w.r.t. the MODIFIED scale s: the reference
//                                             applies no chain factor for mod, :316-320),  dL/dR_ik = s_k dL/dL_ik -> dL/dq
//                                             through R(q) with q taken as given (no normalisation Jacobian, :340)
//   dL/dp     = Wv^T dL/dt  +  d(ndc.xy)/dp^T (g2x, g2y),   ndc = (PV p).xy / ((PV p).w + 1e-7)                 (:370-387)
// The factorisation through U = M2 L never forms Sigma when scales and rotations are given.
struct Sym2 { float xx, xy, yy; };

__device__ __forceinline__ void quat_rows(float4 q, float R[3][3]) {   // standard rotation matrix of (r, x, y, z), rows
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z);       R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z);       R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y);       R[2][1] = 2.f * (y * z + r * x);       R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// gacc rows hold the tile kernel's per-Gaussian sums (gs_layout.h); if gacc is null the caller supplied dL_dmean2D /
// dL_dconic directly (segs_debug_preprocess_backward: this stage alone against the oracle).
__global__ void __launch_bounds__(256) preprocess_bwd_kernel(
    int P, const float* __restrict__ means3D, const int* __restrict__ radii, const float* __restrict__ scales,
    const float* __restrict__ rotations, float mod, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ view, const float* __restrict__ proj, float h_x, float h_y, float tan_fovx, float tan_fovy,
    float* __restrict__ gacc, float img_w, float img_h,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D,
    float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
    int clean_gacc /* write zeros back over the consumed accumulator row (resident backward) */) {
  __shared__ float lds[3 * 768];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < P;
  const bool binned = live && radii[idx] > 0;
  float3 mean, scale, unused;
  load_rows3(means3D, scales, nullptr, P, lds, mean, scale, unused);

  // ---- what arrives from the tile kernel
  float g2x = 0.f, g2y = 0.f;            // dL/dmean2D, NDC-scaled (backward.cu:541-542)
  Sym2 G{0.f, 0.f, 0.f};                 // dL/dconic
  float dcol0 = 0.f, dcol1 = 0.f, dcol2 = 0.f, dop = 0.f;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;   // the tile kernel's row: Mx My Mxx Mxy | Myy dL/dopacity Sr Sg | Sb
  if (gacc) {
    // Rows hold raw moments (render.hip): sums over the (pixel, Gaussian) pairs of w = dL/dG * G times 1, dx, dy, dx^2, ...;
    // the reference's per-pair terms (backward.cu:541-554) are linear in them:
    //   dL/dmean2D.x = -(A Mx + B My) W/2,  .y = -(C My + B Mx) H/2,  dL/dconic = -(Mxx, Mxy, Myy)/2
    // with (A, B, C) the conic, formed below from this kernel's own 2D covariance (round 2 gathered it -- and the opacity, which
    // the tile kernel now divides out itself -- from the 32-byte emit record: 9 % of this kernel's traffic).
    // The rows arrive (and are cleared) through a per-wave LDS transpose, like K1's records: a lane reading its own row as
    // two float4 and a dword makes each load touch 64 lines; transposed, an instruction moves sixteen whole rows.
    {
      __shared__ __attribute__((aligned(16))) float gl[4][32 * GACC_DWORDS];
      const int lane = threadIdx.x & 63;
      float* const wl = gl[threadIdx.x >> 6];
      const uint64_t rows = __ballot(binned);   // only binned Gaussians' rows can have been touched by the tile kernel
      const size_t wave_first = (size_t)blockIdx.x * 256 + (threadIdx.x & ~63);
#pragma unroll
      for (int half = 0; half < 2; half++) {
        float4 q[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int r = 16 * j + (lane >> 2), owner = 32 * half + r;
          q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (((rows >> owner) & 1ull) && (lane & 3) < 3) {
            float4* src = reinterpret_cast<float4*>(gacc + (wave_first + owner) * GACC_DWORDS) + (lane & 3);
            q[j] = *src;
            if (clean_gacc) *src = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int j = 0; j < 2; j++) reinterpret_cast<float4*>(wl + (16 * j + (lane >> 2)) * GACC_DWORDS)[lane & 3] = q[j];
        asm volatile("" ::: "memory");   // (a wave's LDS instructions complete in order)
        if ((lane >> 5) == half) {
          const float4* mine = reinterpret_cast<const float4*>(wl + (lane & 31) * GACC_DWORDS);
          a0 = mine[0]; a1 = mine[1]; dcol2 = wl[(lane & 31) * GACC_DWORDS + 8];
        }
        asm volatile("" ::: "memory");
      }
    }
    if (binned) {
      G.xx = -0.5f * a0.z; G.xy = -0.5f * a0.w; G.yy = -0.5f * a1.x;
      dop = a1.y;
      dcol0 = a1.z; dcol1 = a1.w;
    }
    store_row3(dL_dcolor, P, lds, dcol0, dcol1, dcol2);
    if (live) {
      if (dL_dconic) reinterpret_cast<float4*>(dL_dconic)[idx] = make_float4(G.xx, G.xy, 0.f, G.yy);   // internal product: optional
      dL_dopacity[idx] = dop;
    }
  } else if (live) {
    g2x = dL_dmean2D[3 * (size_t)idx + 0]; g2y = dL_dmean2D[3 * (size_t)idx + 1];
    G.xx = dL_dconic[4 * (size_t)idx + 0]; G.xy = dL_dconic[4 * (size_t)idx + 1]; G.yy = dL_dconic[4 * (size_t)idx + 3];
  }

  float out_mean[3] = {0, 0, 0}, out_cov[6] = {0, 0, 0, 0, 0, 0}, out_scale[3] = {0, 0, 0}, out_rot[4] = {0, 0, 0, 0};
  if (binned) {
    // ---- forward quantities again (cheaper than storing them: 36 B of inputs against ~30 floats of intermediates)
    const float tx0 = view[0] * mean.x + view[4] * mean.y + view[8] * mean.z + view[12];
    const float ty0 = view[1] * mean.x + view[5] * mean.y + view[9] * mean.z + view[13];
    const float tz = view[2] * mean.x + view[6] * mean.y + view[10] * mean.z + view[14];
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float rx = tx0 / tz, ry = ty0 / tz;
    const bool free_x = !(rx < -limx || rx > limx), free_y = !(ry < -limy || ry > limy);
    const float tx = fminf(limx, fmaxf(-limx, rx)) * tz, ty = fminf(limy, fmaxf(-limy, ry)) * tz;
    const float iz = 1.f / tz, iz2 = iz * iz;
    const float J00 = h_x * iz, J02 = -h_x * tx * iz2, J11 = h_y * iz, J12 = -h_y * ty * iz2;
    float M2[2][3];   // M2 = J Wv,  Wv[i][j] = view[4 j + i]
#pragma unroll
    for (int j = 0; j < 3; j++) {
      M2[0][j] = J00 * view[4 * j] + J02 * view[4 * j + 2];
      M2[1][j] = J11 * view[4 * j + 1] + J12 * view[4 * j + 2];
    }
    float R[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, s3[3] = {0, 0, 0};
    float4 q = make_float4(0, 0, 0, 0);
    float E[2][3];    // E = M2 Sigma (= U L^T)
    float U[2][3] = {{0, 0, 0}, {0, 0, 0}};
    float a, b, c;
    if (cov3D_precomp) {
      const float* cv = cov3D_precomp + (size_t)6 * idx;
      const float S[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) E[r][j] = M2[r][0] * S[0][j] + M2[r][1] * S[1][j] + M2[r][2] * S[2][j];
      a = E[0][0] * M2[0][0] + E[0][1] * M2[0][1] + E[0][2] * M2[0][2] + 0.3f;
      b = E[0][0] * M2[1][0] + E[0][1] * M2[1][1] + E[0][2] * M2[1][2];
      c = E[1][0] * M2[1][0] + E[1][1] * M2[1][1] + E[1][2] * M2[1][2] + 0.3f;
    } else {
      q = reinterpret_cast<const float4*>(rotations)[idx];
      quat_rows(q, R);
      s3[0] = mod * scale.x; s3[1] = mod * scale.y; s3[2] = mod * scale.z;
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) U[r][k] = (M2[r][0] * R[0][k] + M2[r][1] * R[1][k] + M2[r][2] * R[2][k]) * s3[k];
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) E[r][j] = U[r][0] * R[j][0] * s3[0] + U[r][1] * R[j][1] * s3[1] + U[r][2] * R[j][2] * s3[2];
      a = U[0][0] * U[0][0] + U[0][1] * U[0][1] + U[0][2] * U[0][2] + 0.3f;
      b = U[0][0] * U[1][0] + U[0][1] * U[1][1] + U[0][2] * U[1][2];
      c = U[1][0] * U[1][0] + U[1][1] * U[1][1] + U[1][2] * U[1][2] + 0.3f;
    }
    // ---- Dc = -k adj(C) G adj(C)
    const float det = a * c - b * b;
    const float k = 1.0f / (det * det + 0.0000001f);
    if (gacc) {   // dL/dmean2D from the moments and the conic (c, -b, a) / det (forward.cu:217-218)
      const float di = 1.0f / det;
      const float cA = c * di, cB = -b * di, cC = a * di;
      g2x = -(cA * a0.x + cB * a0.y) * (0.5f * img_w);
      g2y = -(cC * a0.y + cB * a0.x) * (0.5f * img_h);
    }
    const float h0 = c * G.xx - b * G.xy, h1 = c * G.xy - b * G.yy;     // rows of adj(C) G
    const float h2 = a * G.xy - b * G.xx, h3 = a * G.yy - b * G.xy;
    Sym2 Dc;
    Dc.xx = -k * (h0 * c - h1 * b);
    Dc.xy = -k * (h1 * a - h0 * b);
    Dc.yy = -k * (h3 * a - h2 * b);
    float DM[2][3];   // Dc M2
#pragma unroll
    for (int j = 0; j < 3; j++) { DM[0][j] = Dc.xx * M2[0][j] + Dc.xy * M2[1][j]; DM[1][j] = Dc.xy * M2[0][j] + Dc.yy * M2[1][j]; }
    if (dL_dcov3D || cov3D_precomp) {   // dL/dSigma = M2^T Dc M2, six unique entries
      const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
      for (int e = 0; e < 6; e++) {
        const float v = M2[0][ii[e]] * DM[0][jj[e]] + M2[1][ii[e]] * DM[1][jj[e]];
        out_cov[e] = ii[e] == jj[e] ? v : 2.f * v;
      }
    }
    // ---- through M2 = J Wv to t, then to the mean
    float dM2[2][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      dM2[0][j] = 2.f * (Dc.xx * E[0][j] + Dc.xy * E[1][j]);
      dM2[1][j] = 2.f * (Dc.xy * E[0][j] + Dc.yy * E[1][j]);
    }
    // dL/dJ[r][i] = sum_j dM2[r][j] Wv[i][j]; only the four structural entries of J
    const float dJ00 = dM2[0][0] * view[0] + dM2[0][1] * view[4] + dM2[0][2] * view[8];
    const float dJ02 = dM2[0][0] * view[2] + dM2[0][1] * view[6] + dM2[0][2] * view[10];
    const float dJ11 = dM2[1][0] * view[1] + dM2[1][1] * view[5] + dM2[1][2] * view[9];
    const float dJ12 = dM2[1][0] * view[2] + dM2[1][1] * view[6] + dM2[1][2] * view[10];
    const float iz3 = iz2 * iz;
    const float dtx = free_x ? -h_x * iz2 * dJ02 : 0.f;
    const float dty = free_y ? -h_y * iz2 * dJ12 : 0.f;
    const float dtz = -h_x * iz2 * dJ00 - h_y * iz2 * dJ11 + 2.f * h_x * tx * iz3 * dJ02 + 2.f * h_y * ty * iz3 * dJ12;
    // projection term: ndc = hom.xy * w,  w = 1 / (hom.w + 1e-7)
    const float hx = proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12];
    const float hy = proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13];
    const float hw = proj[3] * mean.x + proj[7] * mean.y + proj[11] * mean.z + proj[15];
    const float w = 1.0f / (hw + 0.0000001f);
    const float ux = hx * w * w, uy = hy * w * w;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float from_cov = view[4 * j] * dtx + view[4 * j + 1] * dty + view[4 * j + 2] * dtz;
      const float from_proj = (proj[4 * j] * w - proj[4 * j + 3] * ux) * g2x + (proj[4 * j + 1] * w - proj[4 * j + 3] * uy) * g2y;
      out_mean[j] = from_cov + from_proj;
    }
    // ---- scales and rotation: dL/dL = 2 M2^T (Dc U)
    if (scales) {
      float DU[2][3];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) { DU[0][kk] = Dc.xx * U[0][kk] + Dc.xy * U[1][kk]; DU[1][kk] = Dc.xy * U[0][kk] + Dc.yy * U[1][kk]; }
      float GR[3][3];   // dL/dR
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        float ds = 0.f;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const float dLik = 2.f * (M2[0][i] * DU[0][kk] + M2[1][i] * DU[1][kk]);
          ds += R[i][kk] * dLik;
          GR[i][kk] = dLik * s3[kk];
        }
        out_scale[kk] = ds;
      }
      const float r = q.x, x = q.y, y = q.z, z = q.w;
      out_rot[0] = 2.f * (z * (GR[1][0] - GR[0][1]) + y * (GR[0][2] - GR[2][0]) + x * (GR[2][1] - GR[1][2]));
      out_rot[1] = 2.f * (y * (GR[0][1] + GR[1][0]) + z * (GR[0][2] + GR[2][0]) + r * (GR[2][1] - GR[1][2])) - 4.f * x * (GR[1][1] + GR[2][2]);
      out_rot[2] = 2.f * (x * (GR[0][1] + GR[1][0]) + r * (GR[0][2] - GR[2][0]) + z * (GR[1][2] + GR[2][1])) - 4.f * y * (GR[0][0] + GR[2][2]);
      out_rot[3] = 2.f * (r * (GR[1][0] - GR[0][1]) + x * (GR[0][2] + GR[2][0]) + y * (GR[1][2] + GR[2][1])) - 4.f * z * (GR[0][0] + GR[1][1]);
    }
  }
  if (gacc) store_row3(dL_dmean2D, P, lds, g2x, g2y, 0.f);
  store_row3(dL_dmean3D, P, lds, out_mean[0], out_mean[1], out_mean[2]);
  if (dL_dscale) store_row3(dL_dscale, P, lds, out_scale[0], out_scale[1], out_scale[2]);
  if (!live) return;
  if (dL_dcov3D) {   // only a caller that passed cov3D_precomp has a use for it
#pragma unroll
    for (int e = 0; e < 6; e++) dL_dcov3D[6 * (size_t)idx + e] = out_cov[e];
  }
  if (dL_drot) reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(out_rot[0], out_rot[1], out_rot[2], out_rot[3]);
}

// SH colour branch only (callers that pass `shs`; off the live SEGS-SLAM path): dL/dsh and the view-direction term of
// dL/dmean3D from the summed dL/dcolor (backward.cu:390-391), as a pass of its own so that its tables stay out of the
// per-Gaussian backward every training step runs.
__global__ void __launch_bounds__(256) sh_backward_kernel(int P, const float* __restrict__ means3D, const int* __restrict__ radii,
                                                          const float* __restrict__ shs, int D, int M, const float* __restrict__ cam_pos,
                                                          const uint32_t* __restrict__ clamped, const float* __restrict__ dL_dcolor,
                                                          float* __restrict__ dL_dmean3D, float* __restrict__ dL_dsh) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P || radii[idx] <= 0) return;
  const float3 mean = make_float3(means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]);
  const float3 dcol = make_float3(dL_dcolor[3 * (size_t)idx], dL_dcolor[3 * (size_t)idx + 1], dL_dcolor[3 * (size_t)idx + 2]);
  const float3 g = sh::backward(idx, D, M, mean, cam_pos, shs, clamped[idx], dcol, dL_dsh);
  dL_dmean3D[3 * (size_t)idx] += g.x; dL_dmean3D[3 * (size_t)idx + 1] += g.y; dL_dmean3D[3 * (size_t)idx + 2] += g.z;
}

}  // namespace segs
