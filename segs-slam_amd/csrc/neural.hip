// neural.hip -- fused neural-Gaussian generation (Scaffold-GS anchors -> Gaussians), forward and backward
// (include/segs_neural.h).  Reference: GaussianRenderer::generate_neural_gaussians, src/gaussian_renderer.cpp:214-334,
// MLP stacks src/gaussian_model.cpp:61-98.
//
//   compact_visible_kernel : radii>0 -> visible-anchor list + device-side count (no host sync); clears the opacity of
//                            the slots of invisible anchors so the rasterizer skips them.  Projecting forward: works out the
//                            anchors' visibility itself (prefilter_voxel) and leaves K1's "culled" outputs for those slots
//   neural_fwd_kernel      : wave = 32 visible anchors; view direction, feature bank, three 35->32->{10,70,30} MLPs
//                            chained on fp32 MFMA (activations stay in registers, weights as LDS operand images),
//                            mask, xyz/scale/rot assembly, written straight into the rasterizer's input arrays.
//                            <true> (segs_neural_forward_projected): also runs the rasterizer's per-Gaussian
//                            stage K1 (project_gaussian.h) on the live candidates and writes its 64-byte records, radii,
//                            tile counts and depth keys into the resident rasterizer buffers
//   neural_bwd_kernel      : same mapping; recomputes the forward (cheaper than saving ~100 floats/anchor),
//                            back-propagates the candidate-domain gradients to anchor/offset/feature/scaling and leaves
//                            the per-anchor (activation, pre-activation gradient) rows in scratch
//   wgrad_mfma_kernel      : dW[j][i] = sum_anchors dpre[a][j] * act[a][i] for all eight Linear layers with
//                            v_mfma_f32_32x32x2_f32 (exact fp32), operands loaded straight from the scratch rows in
//                            the MFMA lane order (lane = column, two anchors per instruction); fixed wave count,
//                            per-wave partial tiles
//   wgrad_reduce_kernel    : deterministic sum of the partial tiles, += into the flat gradient block
//   appearance_finish_kernel: the appearance embedding Linear(7->app) feeds every anchor the same vector, so it is
//                            folded into the colour MLP's first bias; its gradients follow from that bias gradient.
// A thread-per-anchor VALU version of the two MLP kernels (weights as LDS broadcasts) was LDS-issue bound: 119 / 289 us
// at 35 k visible anchors; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <type_traits>
#include <cstdlib>
#include "../../include/segs_neural.h"
#include "../../include/segs_raster.h"
#include "kernels.h"
#include "project_gaussian.h"
#pragma clang fp contract(fast)   // (project_gaussian.h switches contraction off for its own functions; the MLPs may contract)

namespace {

constexpr int FD = 32;     // feat_dim
constexpr int NO = 10;     // n_offsets
constexpr int XD = 36;     // MLP input: feat(32) | ob_view(3) | ob_dist(1)
constexpr int ROW = 512;   // scratch floats per visible anchor
// Every field is laid out so that each lane half of the backward kernel writes ONE contiguous run (wide stores): the
// kernel is bound by the number of scattered VMEM wave-instructions (64 cache lines each), not by bytes.
constexpr int R_X = 0;     // x[36] in lane order: position 18 h + s holds input 2 s + h
constexpr int R_H = 64;    // + 32*m : hidden activations, m = 0 opacity, 1 cov, 2 colour, 3 feature bank
constexpr int R_DH = 192;  // + 32*m : dL/d(hidden pre-activation)
constexpr int R_DF = 440;  // dL/d(feature-bank logits) [3]
// Compact row of the pair backward's feature-bank path (round 4): only what the bank's two Linears need, 512 B per anchor,
// contiguous -- the weight-gradient kernel streams it (the 2-KB rows above cost it a separate DRAM burst per 128-B field).
constexpr int BROW = 128;
constexpr int RB_H = 0;     // bank hidden activations [32]
constexpr int RB_DH = 32;   // dL/d(bank hidden pre-activation) [32]
constexpr int RB_X = 64;    // x[36] in lane order (the bank's Linear(4 -> 32) reads view, dist: inputs 32..35)
constexpr int RB_DF = 100;  // dL/d(bank logits) [3]
constexpr int WG_WAVES = 256;  // waves per weight-gradient job
constexpr int WG_UNROLL = 8;   // row pairs whose operand loads are in flight together
constexpr int WG_JOBS = 8;
constexpr int WG_TILE = 2 * 3 * 1024;  // floats of partial sums per (job, wave): up to 2x3 tiles of 32x32
constexpr int MAX_APP = 64;
constexpr int MAX_ANCHORS = 8000000;   // 32-bit element offsets: A * ROW and A * 40 stay below 2^32

struct Layout {           // float offsets into the flat parameter block
  int w1[3], b1[3], w2[3], b2[3];
  int in[3];              // row length of w1[m]
  int dist[3];            // 1: the MLP sees ob_dist (column 35 of its input)
  int kapp;               // first appearance column of the colour w1
  int app, aw, ab;        // appearance Linear(7 -> app)
  int bank, fw1, fb1, fw2, fb2;
  int total;
};

int make_layout(const segs_neural_dims* d, Layout* L, int64_t* offsets, int64_t* counts, int* ntensors) {
  if (!d || d->feat_dim != FD || d->n_offsets != NO || d->appearance_dim < 0 || d->appearance_dim > MAX_APP)
    return segs::set_error(SEGS_ERR_UNSUPPORTED, "unsupported configuration (feat_dim must be 32, n_offsets 10, appearance_dim <= 64)");
  int pos = 0, n = 0;
  auto add = [&](int count) {
    if (offsets) offsets[n] = pos;
    if (counts) counts[n] = count;
    n++;
    const int at = pos;
    pos += count;
    return at;
  };
  const int dist[3] = {d->add_opacity_dist ? 1 : 0, d->add_cov_dist ? 1 : 0, d->add_color_dist ? 1 : 0};
  const int nout[3] = {NO, 7 * NO, 3 * NO};
  for (int m = 0; m < 3; m++) {
    L->dist[m] = dist[m];
    L->in[m] = FD + 3 + dist[m] + (m == 2 ? d->appearance_dim : 0);
    L->w1[m] = add(FD * L->in[m]);
    L->b1[m] = add(FD);
    L->w2[m] = add(nout[m] * FD);
    L->b2[m] = add(nout[m]);
  }
  L->kapp = FD + 3 + dist[2];
  L->app = d->appearance_dim;
  L->aw = L->ab = 0;
  if (L->app > 0) { L->aw = add(L->app * 7); L->ab = add(L->app); }
  L->bank = d->use_feat_bank ? 1 : 0;
  L->fw1 = L->fb1 = L->fw2 = L->fb2 = 0;
  if (L->bank) { L->fw1 = add(FD * 4); L->fb1 = add(FD); L->fw2 = add(3 * FD); L->fb2 = add(3); }
  L->total = pos;
  if (ntensors) *ntensors = n;
  return SEGS_OK;
}

struct Temp {             // carve-up of the caller's scratch
  uint32_t* count;        // [0] visible anchors
  uint32_t* vis;          // [A]
  float* rows;            // [A][ROW]
  float* partial;         // [WG_JOBS][WG_WAVES][WG_TILE]
  float* gsum;            // [total + pad] this call's parameter-gradient sums
  float* images;          // [N_IMG_BWD][64] MFMA A-operand images of the MLP weights (built by the forward call)
  void* small;            // Small tables
};
size_t temp_carve(int A, int total, int bank, char* base, Temp* t) {
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
  // scratch rows: only the feature bank's two small Linears still take their weight gradients from them (2 KB per anchor)
  const size_t o_count = take(256), o_vis = take((size_t)A * 4), o_rows = take(bank ? (size_t)A * ROW * 4 : 256),
               o_part = take((size_t)WG_JOBS * WG_WAVES * WG_TILE * 4), o_gsum = take((size_t)(total + 64) * 4),
               o_img = take((size_t)262 * 64 * 4), o_small = take(8192);
  if (t) {
    t->count = (uint32_t*)(base + o_count); t->vis = (uint32_t*)(base + o_vis); t->rows = (float*)(base + o_rows);
    t->partial = (float*)(base + o_part); t->gsum = (float*)(base + o_gsum);
    t->images = (float*)(base + o_img); t->small = (void*)(base + o_small);
  }
  return off;
}

// ---- MLP chain on the matrix cores ------------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 (exact fp32): D[i][j] += sum_{k<2} A[i][k] B[k][j]; lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; D register r of lane l is D[i = rho(r, l>>5)][j = l&31], rho(r,h) = (r&3) + 8 (r>>2) + 4 h.
// Here j (the lane) is the ANCHOR: a wave pushes 32 anchors at a time through  H = relu(W1 X + b1),  OUT = W2 H + b2
// as transposed products, the weight matrices on the A side.  Because D's row index sits in the registers, a layer's
// result is directly the next layer's B operand (register s <-> k index rho(s,h)); activations never leave registers
// and there is no LDS traffic besides the A operands.  The A operands are precomputed once per workgroup as 64-float
// "images" (one per MFMA issue, already in lane order, with the k / row permutations folded in) and read with one
// conflict-free ds_read_b32 each.
// Output rows are assigned so that lane half h of anchor n receives everything about candidates 5h .. 5h+4 of that
// anchor: opacity tile rows rho(r,h), r<5; colour tile r<15 (candidate-major, rgb); three covariance tiles holding
// candidates {0,1}, {2,3}, {4} of the half, 7 values each in registers 0..13.
// Both kernels walk the five tiles in ONE rolled loop (the tile kind selects the element-wise code with wave-uniform
// branches): fully unrolled, the straight-line code (20 / 40 KB, executed once per wave) made the kernels
// instruction-fetch bound (SQ_WAIT_INST_ANY ~30 % of wave time, 32 / 116 us); see DESIGN.md.
constexpr int L1_STEPS = XD / 2;        // 18
constexpr int N_TILES = 5;              // 0 opacity | 1 colour | 2,3,4 covariance
constexpr int I_L1 = 0;                 // [m][18]
constexpr int I_L2 = 3 * L1_STEPS;      // [tile][16]
constexpr int I_DH = I_L2 + N_TILES * 16;   // [tile][16]   backward: dH = W2^T dOUT
constexpr int I_DX = I_DH + N_TILES * 16;   // [m][16]      backward: dX = W1^T dHpre (inputs 0..31)
constexpr int N_IMG_FWD = I_DH, N_IMG_BWD = I_DX + 3 * 16;

struct Small {                          // small per-workgroup tables next to the operand images
  float b1[3][FD];                      // colour: includes W1k[:, appearance columns] . appearance_feat
  float b2t[N_TILES][2][16];            // second-layer bias of the unit held in register r of lane half h, per tile
  float w1tail[3][FD][4];               // W1_m[j][32..35]: view xyz, dist (0 when the MLP does not see it)
  float fw1[FD][4], fb1[FD], fw2[FD][4], fb2[4];   // feature bank; fw2 transposed [j][c]
  float app[MAX_APP];
};

__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ int tile_mlp(int tile) { return tile == 0 ? 0 : (tile == 1 ? 2 : 1); }
// output unit (row of the MLP's second weight matrix) held by lane half h in register r of `tile`, or -1
__device__ __forceinline__ int out_row(int tile, int h, int r) {
  if (tile == 0) return r < 5 ? 5 * h + r : -1;
  if (tile == 1) return r < 15 ? 15 * h + r : -1;
  const int cand = 2 * (tile - 2) + r / 7;       // this half's candidate 0..4: two per covariance tile, one in the last
  return (r < 14 && cand < 5) ? 7 * (5 * h + cand) + r % 7 : -1;
}

// Built once per forward call by one small kernel into the caller's scratch (global), then copied into LDS by every
// workgroup with coalesced float4 loads (gathering the images per workgroup cost 25-45 us of dependent loads).
__global__ void __launch_bounds__(256) pack_tables_kernel(Layout L, const float* __restrict__ P, const float* __restrict__ pose7,
                                                          float* __restrict__ img, Small* __restrict__ Sg,
                                                          uint32_t* __restrict__ count, float* __restrict__ reg_sum) {
  // first kernel of the forward: also clears the two counters and the regulariser sum (a hipMemsetAsync of 8 bytes is a
  // 5 us kernel of its own)
  // The LAST workgroup builds the small tables (two dependent levels: appearance vector, then the colour bias that folds it in);
  // the others build the operand images, one element per thread and one level deep.  (Sixteen workgroups that all formed the
  // appearance vector first, the first of them also the small tables: 8.9 us, all of it latency.)
  const int tid = threadIdx.x, nt = blockDim.x;
  const int img_blocks = (int)gridDim.x - 1;
  if ((int)blockIdx.x < img_blocks)
  for (int e = blockIdx.x * nt + tid; e < N_IMG_BWD * 64; e += img_blocks * nt) {
    const int im = e >> 6, l = e & 63, i = l & 31, hA = l >> 5;
    float v = 0.f;
    if (im < I_L2) {                     // layer 1: A[i = hidden][k = 2s + hA]
      const int m = im / L1_STEPS, s = im - m * L1_STEPS, k = 2 * s + hA;
      if (k < FD + 3 + L.dist[m]) v = P[L.w1[m] + i * L.in[m] + k];
    } else if (im < I_DH) {              // layer 2: A[i = tile row][k <-> hidden rho(s, hA)]
      const int tile = (im - I_L2) >> 4, s = (im - I_L2) & 15, m = tile_mlp(tile);
      const int o = out_row(tile, (i >> 2) & 1, (i & 3) + 4 * (i >> 3));
      if (o >= 0) v = P[L.w2[m] + o * FD + rho(s, hA)];
    } else if (im < I_DX) {              // dH: A[i = hidden][k <-> tile row rho(s, hA)]
      const int tile = (im - I_DH) >> 4, s = (im - I_DH) & 15, m = tile_mlp(tile);
      const int o = out_row(tile, hA, s);
      if (o >= 0) v = P[L.w2[m] + o * FD + i];
    } else {                             // dX: A[i = input 0..31][k <-> hidden rho(s, hA)]
      const int m = (im - I_DX) >> 4, s = (im - I_DX) & 15;
      v = P[L.w1[m] + rho(s, hA) * L.in[m] + i];
    }
    img[e] = v;
  }
  if ((int)blockIdx.x != img_blocks) return;
  if (tid == 0) { count[0] = 0u; count[1] = 0u; *reg_sum = 0.f; }
  __shared__ float app[MAX_APP];
  for (int a = tid; a < MAX_APP; a += nt) {
    float s = 0.f;
    if (a < L.app) {
      s = P[L.ab + a];
      for (int q = 0; q < 7; q++) s += P[L.aw + a * 7 + q] * pose7[q];
    }
    app[a] = s;
  }
  __syncthreads();
  Small& S = *Sg;
  for (int e = tid; e < 3 * FD; e += nt) {
    const int m = e / FD, j = e - m * FD;
    float b = P[L.b1[m] + j];
    if (m == 2)
      for (int a = 0; a < L.app; a++) b += P[L.w1[2] + j * L.in[2] + L.kapp + a] * app[a];
    S.b1[m][j] = b;
    for (int c = 0; c < 4; c++) S.w1tail[m][j][c] = (c < 3 + L.dist[m]) ? P[L.w1[m] + j * L.in[m] + FD + c] : 0.f;
  }
  for (int e = tid; e < N_TILES * 32; e += nt) {
    const int tile = e >> 5, hh = (e >> 4) & 1, r = e & 15;
    const int o = out_row(tile, hh, r);
    S.b2t[tile][hh][r] = o >= 0 ? P[L.b2[tile_mlp(tile)] + o] : 0.f;
  }
  for (int e = tid; e < FD * 4; e += nt) {
    const int j = e >> 2, c = e & 3;
    (&S.fw1[0][0])[e] = L.bank ? P[L.fw1 + e] : 0.f;
    S.fw2[j][c] = (L.bank && c < 3) ? P[L.fw2 + c * FD + j] : 0.f;
  }
  for (int e = tid; e < FD; e += nt) S.fb1[e] = L.bank ? P[L.fb1 + e] : 0.f;
  for (int e = tid; e < 4; e += nt) S.fb2[e] = (L.bank && e < 3) ? P[L.fb2 + e] : 0.f;
  for (int e = tid; e < MAX_APP; e += nt) S.app[e] = app[e];
}

static_assert(sizeof(Small) % 16 == 0, "Small is copied as float4");
__device__ __forceinline__ void stage_tables(float* __restrict__ img, Small& S, int n_img, const float* __restrict__ g_img,
                                             const Small* __restrict__ g_small) {
  const float4* src = reinterpret_cast<const float4*>(g_img);
  float4* dst = reinterpret_cast<float4*>(img);
#pragma unroll 4
  for (int e = threadIdx.x; e < n_img * 16; e += blockDim.x) dst[e] = src[e];
  const float4* s2 = reinterpret_cast<const float4*>(g_small);
  float4* d2 = reinterpret_cast<float4*>(&S);
  for (int e = threadIdx.x; e < (int)(sizeof(Small) / 16); e += blockDim.x) d2[e] = s2[e];
  __syncthreads();
}

// v_exp_f32 / v_rcp_f32 (1 ulp) instead of the ocml routines: the MLP outputs are compared at 2e-5 absolute
__device__ __forceinline__ float fast_exp(float v) { return __builtin_amdgcn_exp2f(v * 1.4426950408889634f); }
__device__ __forceinline__ float sigmoidf(float v) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-v)); }
__device__ __forceinline__ float fast_tanh(float v) {
  const float e = fast_exp(-2.0f * fabsf(v));            // in (0, 1]: no overflow
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, v);
}
// Runs of N floats with only 4-byte alignment as dwordx4/x3/x2 accesses (gfx950 global memory needs dword alignment only).
struct __attribute__((packed, aligned(4))) U4 { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) U3 { float x, y, z; };
struct __attribute__((packed, aligned(4))) U2 { float x, y; };
template <int N>
__device__ __forceinline__ void ldn(const float* __restrict__ p, float* v) {
  constexpr int Q = N / 4, R = N % 4;
#pragma unroll
  for (int i = 0; i < Q; i++) { const U4 t = *reinterpret_cast<const U4*>(p + 4 * i); v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
  if (R == 3) { const U3 t = *reinterpret_cast<const U3*>(p + 4 * Q); v[4 * Q] = t.x; v[4 * Q + 1] = t.y; v[4 * Q + 2] = t.z; }
  if (R == 2) { const U2 t = *reinterpret_cast<const U2*>(p + 4 * Q); v[4 * Q] = t.x; v[4 * Q + 1] = t.y; }
  if (R == 1) v[4 * Q] = p[4 * Q];
}
template <int N>
__device__ __forceinline__ void stn(float* __restrict__ p, const float* v) {
  constexpr int Q = N / 4, R = N % 4;
#pragma unroll
  for (int i = 0; i < Q; i++) *reinterpret_cast<U4*>(p + 4 * i) = U4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
  if (R == 3) *reinterpret_cast<U3*>(p + 4 * Q) = U3{v[4 * Q], v[4 * Q + 1], v[4 * Q + 2]};
  if (R == 2) *reinterpret_cast<U2*>(p + 4 * Q) = U2{v[4 * Q], v[4 * Q + 1]};
  if (R == 1) p[4 * Q] = v[4 * Q];
}
__device__ __forceinline__ float other_half(float v) { return __shfl_xor(v, 32, 64); }

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Per-lane view of one anchor (both lane halves hold the same anchor).
struct AnchorLane {
  float xo[L1_STEPS];     // this half's inputs: xo[s] = x[2s + h], x = feat'(32) | view(3) | dist
  float anc[3], gs[6];    // anchor, exp(scaling_log)
  float view[3], dist, inv_dist;
  float bw[3];            // feature-bank softmax weights
};

__device__ __forceinline__ void load_feat(const float* __restrict__ anchor_feat, uint32_t a, float* f) {
  const float4* p = reinterpret_cast<const float4*>(anchor_feat + a * FD);
#pragma unroll
  for (int q = 0; q < FD / 4; q++) {
    const float4 v = p[q];
    f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
  }
}

// feature-bank hidden units are split between the lane halves (16 each); returns this half's partial logits
__device__ __forceinline__ void bank_logits_partial(const Small& S, int h, const float* cat4, float* lg) {
  // (opaque to the optimiser: the 48 LDS addresses of this half's weights are otherwise formed once per kernel, kept across the
  // slab loop and spilled; formed here they are one base register and immediate offsets)
  asm volatile("" : "+v"(h));
  lg[0] = lg[1] = lg[2] = 0.f;
#pragma unroll 4
  for (int jj = 0; jj < FD / 2; jj++) {
    const int j = 16 * h + jj;
    const float4 w = *reinterpret_cast<const float4*>(S.fw1[j]);
    const float s = S.fb1[j] + w.x * cat4[0] + w.y * cat4[1] + w.z * cat4[2] + w.w * cat4[3];
    const float hj = fmaxf(s, 0.f);
    const float4 u = *reinterpret_cast<const float4*>(S.fw2[j]);
    lg[0] += u.x * hj; lg[1] += u.y * hj; lg[2] += u.z * hj;
  }
}

// What a lane reads about its anchor, as it comes from memory.  Kept apart from the arithmetic so that the forward kernel can
// request the NEXT slab's anchors while it works on the current one.
struct RawAnchor { float feat[FD]; float anc[3]; float sl[6]; };
__device__ __forceinline__ void load_raw(uint32_t a, const float* __restrict__ anchor, const float* __restrict__ anchor_feat,
                                         const float* __restrict__ scaling_log, RawAnchor& r) {
  load_feat(anchor_feat, a, r.feat);
  ldn<3>(anchor + a * 3, r.anc);
  ldn<6>(scaling_log + a * 6, r.sl);
}

template <bool BANK>
__device__ __forceinline__ void anchor_lane_raw(const Small& S, int h, const RawAnchor& raw, const float* __restrict__ campos,
                                                AnchorLane& st) {
  const float* feat = raw.feat;
#pragma unroll
  for (int c = 0; c < 3; c++) st.anc[c] = raw.anc[c];
#pragma unroll
  for (int c = 0; c < 6; c++) st.gs[c] = expf(raw.sl[c]);
  const float ox = st.anc[0] - campos[0], oy = st.anc[1] - campos[1], oz = st.anc[2] - campos[2];
  st.dist = sqrtf(ox * ox + oy * oy + oz * oz);
  st.inv_dist = 1.0f / st.dist;
  st.view[0] = ox / st.dist; st.view[1] = oy / st.dist; st.view[2] = oz / st.dist;
  if (BANK) {
    const float cat4[4] = {st.view[0], st.view[1], st.view[2], st.dist};
    float lg[3];
    bank_logits_partial(S, h, cat4, lg);
#pragma unroll
    for (int c = 0; c < 3; c++) lg[c] = lg[c] + other_half(lg[c]) + S.fb2[c];
    const float mx = fmaxf(lg[0], fmaxf(lg[1], lg[2]));
    const float e0 = expf(lg[0] - mx), e1 = expf(lg[1] - mx), e2 = expf(lg[2] - mx);
    const float inv = 1.0f / (e0 + e1 + e2);
    st.bw[0] = e0 * inv; st.bw[1] = e1 * inv; st.bw[2] = e2 * inv;
    // feat'[k] = feat[4 (k%8)] bw0 + feat[2 (k%16)] bw1 + feat[k] bw2   (gaussian_renderer.cpp:242-247), k = 2s + h
#pragma unroll
    for (int s = 0; s < FD / 2; s++) {
      const int k0 = 2 * s, k1 = 2 * s + 1;
      const float v0 = feat[4 * (k0 % 8)] * st.bw[0] + feat[2 * (k0 % 16)] * st.bw[1] + feat[k0] * st.bw[2];
      const float v1 = feat[4 * (k1 % 8)] * st.bw[0] + feat[2 * (k1 % 16)] * st.bw[1] + feat[k1] * st.bw[2];
      st.xo[s] = h ? v1 : v0;
    }
  } else {
    st.bw[0] = st.bw[1] = st.bw[2] = 0.f;
#pragma unroll
    for (int s = 0; s < FD / 2; s++) st.xo[s] = h ? feat[2 * s + 1] : feat[2 * s];
  }
  st.xo[16] = h ? st.view[1] : st.view[0];
  st.xo[17] = h ? st.dist : st.view[2];
}
template <bool BANK>
__device__ __forceinline__ void anchor_lane(const Small& S, const Layout& L, uint32_t a, int h, const float* __restrict__ anchor,
                                            const float* __restrict__ anchor_feat, const float* __restrict__ scaling_log,
                                            const float* __restrict__ campos, AnchorLane& st) {
  (void)L;
  RawAnchor raw;
  load_raw(a, anchor, anchor_feat, scaling_log, raw);
  anchor_lane_raw<BANK>(S, h, raw, campos, st);
}

// H_pre = W1_m X + b1_m  (rows rho(r,h) in the registers); m is a runtime value
__device__ __forceinline__ f32x16 layer1(const float* __restrict__ img, const Small& S, int m, int lane, int h, const float* xo) {
  f32x16 d;
#pragma unroll
  for (int r = 0; r < 16; r++) d[r] = S.b1[m][rho(r, h)];
  const float* im = img + (I_L1 + m * L1_STEPS) * 64 + lane;
#pragma unroll
  for (int s = 0; s < L1_STEPS; s++) d = __builtin_amdgcn_mfma_f32_32x32x2f32(im[s * 64], xo[s], d, 0, 0, 0);
  return d;
}
// OUT tile = W2 H + b2 for this half's candidates
__device__ __forceinline__ f32x16 layer2(const float* __restrict__ img, const Small& S, int tile, int lane, int h, const f32x16& hpre) {
  f32x16 d;
#pragma unroll
  for (int r = 0; r < 16; r++) d[r] = S.b2t[tile][h][r];
  const float* im = img + (I_L2 + tile * 16) * 64 + lane;
#pragma unroll
  for (int s = 0; s < 16; s++) d = __builtin_amdgcn_mfma_f32_32x32x2f32(im[s * 64], fmaxf(hpre[s], 0.f), d, 0, 0, 0);
  return d;
}

// Persistent forward workgroups: one of eight waves per CU (two waves per SIMD; four per SIMD was slower: 112 -> 141 us at 300 k
// anchors), each wave loops over 32-anchor slabs.  Eight waves share one copy of the tables and leave room for a 12.6-KB output
// staging area per wave (below).
constexpr int NEURAL_GRID = 256;
constexpr int FWD_WAVES = 8;
// Output staging of one wave: the outputs of a slab -- 32 anchors x (rotations 40 | scales 30 | means 30) floats, or x 10
// opacities, or x 30 colour values -- go to LDS in the layout of the output arrays and leave as contiguous runs: lanes write
// consecutive 8- or 16-byte elements of consecutive anchors' runs, so a store instruction covers whole lines.  Stored straight from
// the registers (12-16 bytes per lane, 60-120 bytes apart) every store instruction touched 64 lines, and the 24 of them per slab
// were what the kernel waited for: 97 us at 300 k anchors, 60 us with the stores compiled out.
constexpr int STG_ROW = 100;                       // floats per anchor of the covariance phase: rotations 0..39 | scales 40..69 | means 70..99
constexpr int STG_WAVE = 32 * STG_ROW + 32;        // + the slab's anchor indices
constexpr size_t FWD_LDS = (N_IMG_FWD * 64 + FWD_WAVES * STG_WAVE) * sizeof(float) + sizeof(Small);
// The projecting forward adds 32 records x 16 dwords per wave behind the anchor indices: the 64-byte records leave
// through it half a wave at a time (as in preprocess_fwd_kernel), the slab's neural opacities before that.
constexpr int STG_SPARE = 32 * segs::REC_DWORDS;
static_assert(32 * NO + 32 * NO / 2 <= STG_SPARE, "the slab's opacities + the dense list of its live candidates (u16) share that space");
constexpr int STG_WAVE_PROJ = STG_WAVE + STG_SPARE;
constexpr size_t FWD_LDS_PROJ = (N_IMG_FWD * 64 + FWD_WAVES * STG_WAVE_PROJ) * sizeof(float) + sizeof(Small);
static_assert(FWD_LDS <= 160 * 1024 && FWD_LDS_PROJ <= 160 * 1024, "one forward workgroup per CU");

// What the projecting forward needs of the rasterizer's resident state (segs_projection_targets) and of the camera.
struct Proj {
  float* rec; int* radii; uint32_t* touched; uint32_t* keys; uint32_t* overflow;
  const float* view; const float* proj;
  int W, H;
  float tanx, tany, fx, fy, mod;
  uint32_t gx, gy, flags;
};

// 2048 anchors per workgroup (two rounds of 1024 threads; eight rounds of 256 until round 4) and ONE returning atomic on the list's count per workgroup: the count is a
// single word, which takes about 12 ns per returning atomic whatever the parallelism -- with one per 256 anchors this kernel
// spent 14 of its 17 us at 300 k anchors queueing on it.
// Shape by size: <1024 threads, 2 rounds> = 2048 anchors per workgroup for large maps (the atomic), <256, 1> below 131 072
// anchors, where 2048 per workgroup would leave most CUs without one (50 k anchors: 25 workgroups; 12.7 us against ~7).
// proj_radii != null (projecting forward): the candidates of invisible anchors get what K1 would have left for them -- radius 0,
// no tiles, the culled depth key -- instead of a zero opacity for K1 to find, and the tile range table is reset here.
// pf.rot != null: the anchors' visibility (prefilter_voxel, src/gaussian_renderer.cpp:131-199: the anchors drawn as Gaussians with
// exp(scaling[:, :3]) and their stored rotation -- filter_preprocessCUDA, cuda_rasterizer/forward.cu:259-334) is worked out right
// here with the arithmetic of visible_filter_kernel and written to radii_out; `radii` is then not read.
struct Prefilter {
  const float* anchor; const float* scaling_log; const float* rot; const float* view; const float* proj;
  int W, H;
  float tanx, tany, fx, fy;
  uint32_t gx, gy;
};
template <int CV_THREADS, int CV_ROUNDS>
__global__ void __launch_bounds__(CV_THREADS) compact_visible_kernel(int A, const int* __restrict__ radii, uint32_t* __restrict__ count,
                                                                     uint32_t* __restrict__ vis, float* __restrict__ opacity,
                                                                     float* __restrict__ neural_opacity, int* __restrict__ proj_radii,
                                                                     uint32_t* __restrict__ proj_touched, uint32_t* __restrict__ proj_keys,
                                                                     uint2* __restrict__ ranges, int num_tiles, Prefilter pf,
                                                                     int* __restrict__ radii_out) {
  constexpr int CV_WAVES = CV_THREADS / 64;
  __shared__ uint32_t wave_n[CV_ROUNDS][CV_WAVES], block_base;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int a0 = blockIdx.x * (CV_THREADS * CV_ROUNDS) + threadIdx.x;
  uint32_t mine = 0u, below[CV_ROUNDS];
#pragma unroll
  for (int r = 0; r < CV_ROUNDS; r++) {
    const int a = a0 + r * CV_THREADS;
    bool v;
    if (pf.rot) {
      int rad = 0;
      if (a < A) {
        const float* const sl = pf.scaling_log + (size_t)a * 6;
        const float3 p = make_float3(pf.anchor[(size_t)a * 3], pf.anchor[(size_t)a * 3 + 1], pf.anchor[(size_t)a * 3 + 2]);
        const float3 sc = make_float3(expf(sl[0]), expf(sl[1]), expf(sl[2]));
        const float4 q = reinterpret_cast<const float4*>(pf.rot)[a];
        rad = segs::project_gaussian(p, sc, 1.0f, q, nullptr, pf.view, pf.proj, pf.W, pf.H, pf.tanx, pf.tany, pf.fx, pf.fy, pf.gx, pf.gy).radius;
        radii_out[a] = rad;
      }
      v = rad > 0;
    } else {
      v = a < A && (radii == nullptr || radii[a] > 0);
    }
    const uint64_t m = __ballot(v);
    if (lane == 0) wave_n[r][wv] = (uint32_t)__popcll(m);
    below[r] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    mine |= (v ? 1u : 0u) << r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
#pragma unroll
    for (int r = 0; r < CV_ROUNDS; r++)
#pragma unroll
      for (int w = 0; w < CV_WAVES; w++) tot += wave_n[r][w];
    block_base = tot ? atomicAdd(count, tot) : 0u;   // ONE returning atomic on the list's count per 2048 anchors (about 12 ns each, whatever the parallelism)
  }
  __syncthreads();
  uint32_t base = block_base;
#pragma unroll
  for (int r = 0; r < CV_ROUNDS; r++) {
    const int a = a0 + r * CV_THREADS;
    uint32_t mybase = base, round_total = 0u;
#pragma unroll
    for (int w = 0; w < CV_WAVES; w++) {
      const uint32_t c = wave_n[r][w];
      mybase += w < wv ? c : 0u;
      round_total += c;
    }
    const bool v = (mine >> r) & 1u;
    if (v) vis[mybase + below[r]] = (uint32_t)a;
    if (a < A && !v) {
      if (proj_radii) {
#pragma unroll
        for (int k = 0; k < NO; k++) {
          neural_opacity[(size_t)a * NO + k] = 0.f;
          proj_radii[(size_t)a * NO + k] = 0; proj_touched[(size_t)a * NO + k] = 0u; proj_keys[(size_t)a * NO + k] = 0xFFFFFFFFu;
        }
      } else {
#pragma unroll
        for (int k = 0; k < NO; k++) { opacity[(size_t)a * NO + k] = 0.f; neural_opacity[(size_t)a * NO + k] = 0.f; }
      }
    }
    base += round_total;
  }
  if (ranges)
    for (int t = blockIdx.x * CV_THREADS + threadIdx.x; t < num_tiles; t += gridDim.x * CV_THREADS) ranges[t] = make_uint2(segs::RANGE_EMPTY_START, 0u);
}

// PROJECT (src/gaussian_renderer.cpp:299-333 feeding cuda_rasterizer/forward.cu:155-256): the wave also runs K1 on the
// candidates it generates -- project_gaussian with contraction off, bit for bit what preprocess_fwd_kernel computes from the arrays
// this kernel would have written -- and leaves the 64-byte record, radius, tile count and depth key in the rasterizer's resident
// buffers.  Colours and opacities then live only in the records; means / scales / rotations are still written (the rasterizer's
// backward re-reads them), and neural_opacity (densification statistics).  The tiles run covariance first, so that a candidate's
// opacity and colour are in registers when its geometry is read back from the staging rows.
template <bool PROJECT>
__global__ void __launch_bounds__(FWD_WAVES * 64, 1) neural_fwd_kernel(
    Layout L, const uint32_t* __restrict__ count, const uint32_t* __restrict__ vis, const float* __restrict__ anchor,
    const float* __restrict__ offset, const float* __restrict__ anchor_feat, const float* __restrict__ scaling_log,
    const float* __restrict__ g_img, const Small* __restrict__ g_small, const float* __restrict__ campos,
    float* __restrict__ means3D, float* __restrict__ colors, float* __restrict__ opacity, float* __restrict__ scales,
    float* __restrict__ rotations, float* __restrict__ neural_opacity, uint32_t* __restrict__ n_kept, Proj pj) {
  extern __shared__ __align__(16) float lds_dyn[];
  float* img = lds_dyn;
  Small& S = *reinterpret_cast<Small*>(lds_dyn + N_IMG_FWD * 64);
  __shared__ uint32_t blk_kept;
  const uint32_t n = *count;
  if (blockIdx.x * (FWD_WAVES * 32u) >= n) return;
  if (threadIdx.x == 0) blk_kept = 0u;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = lane & 31, h = lane >> 5;
  uint32_t kept_total = 0;
  // Software pipeline over the wave's slabs: the anchors (index, features, position, scaling, the five offsets of this lane
  // half) of slab k + 1 are requested before slab k is computed, so no slab waits for its own loads (each used to pay about
  // seven dependent round trips: index -> anchor data, then one per candidate for the offsets).
  // The anchor INDEX runs two slabs ahead: the data request of slab k + 1 then starts from an index that arrived a slab ago, not
  // from one it has to wait for (index -> data was one exposed round trip per slab, behind the previous slab's stores).
  const uint32_t stride = gridDim.x * (FWD_WAVES * 32u);
  float* const stg = lds_dyn + N_IMG_FWD * 64 + sizeof(Small) / 4 + wv * (PROJECT ? STG_WAVE_PROJ : STG_WAVE);
  uint32_t* const stg_a = reinterpret_cast<uint32_t*>(stg + 32 * STG_ROW);
  float* const spare = stg + STG_WAVE;   // PROJECT only
  // the camera, read once (uniform: scalar registers) -- left behind pj.view / pj.proj every candidate would re-read the 32 floats
  // after each store, which the compiler cannot tell apart from them
  float cam_view[16], cam_proj[16];
  if (PROJECT) {
#pragma unroll
    for (int i = 0; i < 16; i++) { cam_view[i] = pj.view[i]; cam_proj[i] = pj.proj[i]; }
  }
  auto index_of = [&](uint32_t g) { const uint32_t t = g + col; return vis[t < n ? t : n - 1]; };
  RawAnchor raw_nx;
  float off_nx[15];
  auto request = [&](uint32_t a_of) {
    load_raw(a_of, anchor, anchor_feat, scaling_log, raw_nx);
    ldn<15>(offset + (a_of * NO + 5 * h) * 3, off_nx);
  };
  const uint32_t g_first = (blockIdx.x * (uint32_t)FWD_WAVES + wv) * 32u;
  stage_tables(img, S, N_IMG_FWD, g_img, g_small);   // (requesting the first slab in front of this copy changed nothing measurable)
  uint32_t a_cur = index_of(g_first), a_next = index_of(g_first + stride);
  request(a_cur);
  for (uint32_t g0 = g_first; g0 < n; g0 += stride) {
    const uint32_t t = g0 + col;
    const bool valid = t < n;
    const uint32_t a = a_cur;
    const RawAnchor raw = raw_nx;
    float off_q[15];   // offsets of the candidates still to come, the next covariance tile's two in front
#pragma unroll
    for (int q = 0; q < 15; q++) off_q[q] = off_nx[q];
    // (PROJECT: the request goes out after the covariance tiles, which run first there and are where the registers are scarcest --
    // the 56 registers it lands in are then not held through them; the opacity / projection / colour phases still cover its latency)
    const bool more = g0 + stride < n;
    if (!PROJECT && more) request(a_next);
    a_cur = a_next;
    a_next = index_of(g0 + 2u * stride);
    AnchorLane st;
    if (L.bank) anchor_lane_raw<true>(S, h, raw, campos, st);
    else anchor_lane_raw<false>(S, h, raw, campos, st);
    f32x16 hp;
    uint32_t kept = 0;   // candidates of this lane with neural opacity > 0: P of the reference's compacted tensors
    const uint32_t nv = min(32u, n - g0);   // anchors of this slab (wave-uniform): slots nv .. 31 are copies of the last anchor and store nothing
    if (h == 0) stg_a[col] = a;
    // runs of RUN elements (float2, or float4 for the rotations) per anchor, staged at stg + slot * ROW + OFF, to dst + anchor * RUN
    // (all of a field's LDS reads are issued before its first store: a rolled loop pays an LDS round trip per iteration)
    // PROJECT: the lane number the output staging indexes with is made opaque once per slab -- left visible, the dozens of LDS
    // addresses derived from it are hoisted out of the slab loop, and with the projection's registers on top the kernel spills
    // (46 registers; a reload of one waits for every output store issued before it: vmcnt counts in order).
    uint32_t fl = (uint32_t)lane;
    if (PROJECT) asm volatile("" : "+v"(fl));
    const int pc = (int)(fl & 31u), ph = (int)(fl >> 5);   // col, h for the projecting branches
    auto flush2 = [&](float* __restrict__ dst, int row, int off, auto run_c /* float2 per anchor */) {
      constexpr uint32_t run = decltype(run_c)::value, ITER = (32u * run + 63u) / 64u;
      float2 v[ITER];
      uint32_t at[ITER];
#pragma unroll
      for (uint32_t i = 0; i < ITER; i++) {
        const uint32_t e = fl + 64u * i, s = min(e / run, 31u), k = e - (e / run) * run;
        v[i] = *reinterpret_cast<const float2*>(stg + s * row + off + 2 * k);
        at[i] = stg_a[s] * run + k;
      }
#pragma unroll
      for (uint32_t i = 0; i < ITER; i++)
        if (fl + 64u * i < nv * run) reinterpret_cast<float2*>(dst)[at[i]] = v[i];
    };
    using R5 = std::integral_constant<uint32_t, 5u>;
    using R15 = std::integral_constant<uint32_t, 15u>;
    auto do_tile = [&](const int tile) __attribute__((always_inline)) {
      if (tile <= 2) hp = layer1(img, S, tile_mlp(tile), lane, h, st.xo);
      const f32x16 o = layer2(img, S, tile, lane, h, hp);
      if (tile == 0) {
        float* const ostg = PROJECT ? spare : stg;   // (PROJECT: the rows still hold the slab's geometry)
        bool live[5];
#pragma unroll
        for (int r = 0; r < 5; r++) {
          const float op = fast_tanh(o[r]);
          live[r] = valid && op > 0.f;
          if (valid) { kept += op > 0.f ? 1u : 0u; ostg[col * 10 + 5 * h + r] = op; }
        }
        asm volatile("" ::: "memory");   // (a wave's LDS instructions complete in order)
        flush2(neural_opacity, 10, PROJECT ? STG_WAVE : 0, R5());
        if (!PROJECT) flush2(opacity, 10, 0, R5());
        asm volatile("" ::: "memory");
        if (PROJECT) {
          // K1 on the slab's LIVE candidates (neural opacity > 0: the ones the reference's mask keeps, src/gaussian_renderer.cpp:279,320
          // -- SEGS_RASTER_SKIP_NONPOSITIVE_OPACITY in K1), packed: typically a third of the 320, so two passes of 64 lanes where a
          // pass per own candidate takes five, and the ~750 instructions of the projection are what a pass costs.  Dense order:
          // candidate cc of any lane before candidate cc' > cc; `list` maps a dense index to the candidate slot * 10 + k, whose
          // opacity sits at spare[slot * 10 + k].  A pass rewrites its candidates' row slots IN PLACE with what the record needs:
          //   rotation slot (row + 4k): radius | rect_min | rect_max | opacity     scale slot (row + 40 + 3k): x | y | depth
          //   mean slot (row + 70 + 3k): conic A | B | C
          // (tile count and depth key follow from rect and depth); the colour tile below assembles the 64-byte records from them.
          float* const row = stg + pc * STG_ROW;
          uint16_t* const list = reinterpret_cast<uint16_t*>(spare + 32 * NO);
          uint32_t n_live = 0u;
          const uint64_t lt_mask = (1ull << fl) - 1ull;
#pragma unroll
          for (int cc = 0; cc < 5; cc++) {
            const uint64_t m = __ballot(live[cc]);
            const int k = 5 * ph + cc;
            if (live[cc]) list[n_live + (uint32_t)__popcll(m & lt_mask)] = (uint16_t)(pc * NO + k);
            else if (valid) *reinterpret_cast<float4*>(row + 4 * k) = make_float4(0.f, 0.f, 0.f, 0.f);   // radius 0, empty rectangle
            n_live += (uint32_t)__popcll(m);
          }
          asm volatile("" ::: "memory");
#pragma unroll 1
          for (uint32_t j0 = 0; j0 < n_live; j0 += 64u) {
            if (j0 + fl < n_live) {
              const uint32_t cand = list[j0 + fl];
              const uint32_t cs = cand / 10u, ck = cand - cs * 10u;
              float* const crow = stg + cs * STG_ROW;
              const float op = spare[cand];
              const float4 quat = *reinterpret_cast<const float4*>(crow + 4 * ck);
              const float3 sc = make_float3(crow[40 + 3 * ck], crow[41 + 3 * ck], crow[42 + 3 * ck]);
              const float3 mean = make_float3(crow[70 + 3 * ck], crow[71 + 3 * ck], crow[72 + 3 * ck]);
              segs::Projected g = segs::project_gaussian(mean, sc, pj.mod, quat, nullptr, cam_view, cam_proj, pj.W, pj.H, pj.tanx, pj.tany,
                                                         pj.fx, pj.fy, pj.gx, pj.gy);
              float4 head = make_float4(0.f, 0.f, 0.f, op);
              if (g.radius > 0) {
                const segs::BinnedRecord br = segs::make_record(g, op, make_float3(0.f, 0.f, 0.f), pj.flags);
                head = make_float4(__int_as_float(g.radius), __uint_as_float(br.rect_min), __uint_as_float(br.rect_max), op);
                crow[40 + 3 * ck] = br.q0.x; crow[41 + 3 * ck] = br.q0.y; crow[42 + 3 * ck] = br.q2.y;
                crow[70 + 3 * ck] = br.q3.x; crow[71 + 3 * ck] = br.q3.y; crow[72 + 3 * ck] = br.q3.z;
                if (br.touched && (br.depth_bits - segs::DEPTH_KEY_MIN) >= ((1u << segs::DEPTH_KEY_BITS) - 1u)) *pj.overflow = 1u;
              }
              *reinterpret_cast<float4*>(crow + 4 * ck) = head;
            }
            asm volatile("" ::: "memory");
          }
        }
      } else if (tile == 1) {
        if (PROJECT) {
          // the colours arrive: every lane assembles the records of its own candidates (radius > 0) from the row slots the passes
          // above left, 32 records at a time through the staging buffer, four lanes storing a record (as preprocess_fwd_kernel)
          const float* const row = stg + pc * STG_ROW;
#pragma unroll
          for (int cc = 0; cc < 5; cc++) {
            const int k = 5 * ph + cc;
            const float4 head = *reinterpret_cast<const float4*>(row + 4 * k);
            const bool has_record = valid && __float_as_int(head.x) > 0;
            const float x = row[40 + 3 * k], y = row[41 + 3 * k], depth = row[42 + 3 * k];
            const float cA = row[70 + 3 * k], cB = row[71 + 3 * k], cC = row[72 + 3 * k];
            const float4 sq = segs::scaled_conic(cA, cB, cC);
            const uint64_t recs = __ballot(has_record);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int half = 0; half < 2; half++) {
              if (ph == half) {
                float4* mine = reinterpret_cast<float4*>(spare + pc * segs::REC_DWORDS);
                mine[0] = make_float4(x, y, sq.x, sq.y);
                mine[1] = make_float4(sq.z, head.w, sigmoidf(o[3 * cc]), sigmoidf(o[3 * cc + 1]));
                mine[2] = make_float4(sigmoidf(o[3 * cc + 2]), depth, head.y, head.z);
                mine[3] = make_float4(cA, cB, cC, 0.f);
              }
              asm volatile("" ::: "memory");
#pragma unroll
              for (int j = 0; j < 2; j++) {
                const uint32_t r = 16u * j + (fl >> 2);        // anchor slot whose record (candidate 5 * half + cc) this lane carries a quarter of
                if ((recs >> (32 * half + r)) & 1ull) {
                  const size_t gidx = (size_t)stg_a[r] * NO + 5 * half + cc;
                  reinterpret_cast<float4*>(pj.rec + gidx * segs::REC_DWORDS)[fl & 3u] = reinterpret_cast<const float4*>(spare + r * segs::REC_DWORDS)[fl & 3u];
                }
              }
              asm volatile("" ::: "memory");
            }
          }
          {   // ten radii / tile counts / depth keys per anchor, as five 8-byte pairs each
            uint2 v[3][3];
            uint32_t at[3];
#pragma unroll
            for (uint32_t i = 0; i < 3; i++) {
              const uint32_t e = fl + 64u * i, s = min(e / 5u, 31u), k2 = e - (e / 5u) * 5u;
              const float* const r0 = stg + s * STG_ROW;
              uint32_t w[3][2];
#pragma unroll
              for (int c = 0; c < 2; c++) {
                const uint32_t k = 2u * k2 + c;
                const float4 head = *reinterpret_cast<const float4*>(r0 + 4 * k);
                const uint32_t rmin = __float_as_uint(head.y), rmax = __float_as_uint(head.z);
                const uint32_t touched = ((rmax >> 16) - (rmin >> 16)) * ((rmax & 0xFFFFu) - (rmin & 0xFFFFu));
                w[0][c] = __float_as_uint(head.x);
                w[1][c] = touched;
                w[2][c] = touched ? __float_as_uint(r0[42 + 3 * k]) : 0xFFFFFFFFu;
              }
#pragma unroll
              for (int q = 0; q < 3; q++) v[q][i] = make_uint2(w[q][0], w[q][1]);
              at[i] = stg_a[s] * 5u + k2;
            }
#pragma unroll
            for (uint32_t i = 0; i < 3; i++)
              if (fl + 64u * i < nv * 5u) {
                reinterpret_cast<uint2*>(pj.radii)[at[i]] = v[0][i];
                reinterpret_cast<uint2*>(pj.touched)[at[i]] = v[1][i];
                reinterpret_cast<uint2*>(pj.keys)[at[i]] = v[2][i];
              }
            asm volatile("" ::: "memory");
          }
        } else {
          if (valid) {
#pragma unroll
            for (int r = 0; r < 15; r++) stg[col * 30 + 15 * h + r] = sigmoidf(o[r]);
          }
          asm volatile("" ::: "memory");
          flush2(colors, 30, 0, R15());
          asm volatile("" ::: "memory");
        }
      } else {
        if (valid) {
#pragma unroll
          for (int cl = 0; cl < 2; cl++) {
            const int cc = 2 * (tile - 2) + cl;
            if (cc < 5) {
              const float q0 = o[7 * cl + 3], q1 = o[7 * cl + 4], q2 = o[7 * cl + 5], q3 = o[7 * cl + 6];
              float* const row = stg + col * STG_ROW;
              const int k = 5 * h + cc;   // candidate
#pragma unroll
              for (int c = 0; c < 3; c++) {
                row[40 + 3 * k + c] = st.gs[3 + c] * sigmoidf(o[7 * cl + c]);   // :327-328
                row[70 + 3 * k + c] = st.anc[c] + off_q[3 * cl + c] * st.gs[c];  // :331-332
              }
              const float nrm = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);  // F::normalize
              *reinterpret_cast<float4*>(row + 4 * k) = make_float4(q0 / nrm, q1 / nrm, q2 / nrm, q3 / nrm);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 9; q++) off_q[q] = off_q[q + 6];   // the tile loop is rolled: register arrays take no runtime index
        if (tile == N_TILES - 1) {
          asm volatile("" ::: "memory");
          {   // rotations: ten float4 per anchor
            float4 v[5];
            uint32_t at[5];
#pragma unroll
            for (uint32_t i = 0; i < 5; i++) {
              const uint32_t e = fl + 64u * i, s = e / 10u, k = e - s * 10u;
              v[i] = *reinterpret_cast<const float4*>(stg + s * STG_ROW + 4 * k);
              at[i] = stg_a[s] * 10u + k;
            }
#pragma unroll
            for (uint32_t i = 0; i < 5; i++)
              if (fl + 64u * i < nv * 10u) reinterpret_cast<float4*>(rotations)[at[i]] = v[i];
          }
          flush2(scales, STG_ROW, 40, R15());
          flush2(means3D, STG_ROW, 70, R15());
          asm volatile("" ::: "memory");
        }
      }
    };
    if (PROJECT) {
      // covariance first (the projection needs the geometry in the rows when the opacities arrive), then the request for the next
      // slab, then opacity + projection, then colour + records
#pragma unroll 1
      for (int tile = 2; tile < N_TILES; tile++) do_tile(tile);
      if (more) request(a_cur);
      do_tile(0);
      do_tile(1);
    } else {
#pragma unroll 1
      for (int tile = 0; tile < N_TILES; tile++) do_tile(tile);
    }
    kept_total += valid ? kept : 0u;
  }
  // one global atomic per workgroup (one per wave on a single word serialised ~1500 atomics: +10 us)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) kept_total += (uint32_t)__shfl_xor((int)kept_total, off, 64);
  if (lane == 0) atomicAdd(&blk_kept, kept_total);
  __syncthreads();
  if (threadIdx.x == 0 && blk_kept) atomicAdd(n_kept, blk_kept);
}

// ---- weight gradients inside the backward kernel ---------------------------------------------------------------
// dW[j][i] = sum_anchors dpre[a][j] act[a][i] contracts over the anchors, which the MLP chain keeps on the LANES; the matrix
// cores contract over k.  So every operand of a weight-gradient product goes once through a per-wave LDS tile
// T[anchor][unit] (row stride TS floats: conflict-free 16-byte writes from the chain's register layout, conflict-free
// dword reads in the transposed one) and comes back as lane = unit, k = anchor: 16 v_mfma_f32_32x32x2_f32 per 32x32 tile
// of dW and slab of 32 anchors, accumulated in registers over all the slabs of the wave (8 tiles = 128 accumulator
// registers: the kernel runs one wave per SIMD with the 512-register budget).  The tile pointers are deliberately not
// __restrict__: the compiler has to keep the order of a wave's LDS stores and the (other lanes') loads that follow.  The previous design wrote every
// activation and pre-activation gradient to a 2-KB scratch row per anchor for a separate kernel: 0.6 GB written and 0.55 GB
// read back per step at 300 k anchors, more than everything else the backward touches.
constexpr int TS = 36;                       // floats per anchor row of a transposition tile
constexpr int T_TILE = 32 * TS;
constexpr int WAVE_LDS = 3 * T_TILE + 32 * 4;   // X | H | D | tail(view xyz, dist), floats per wave
constexpr int BWD_GRID = 256;                // one workgroup per CU; also the number of partial tiles per job
static_assert(BWD_GRID == WG_WAVES, "the fused backward's partial tiles reuse the per-wave slots of the reduce kernel");
constexpr int N_SMALL = 20;                  // per-lane scalar accumulators: 5 + 3 bias sums, 3 x 4 tail columns

__device__ __forceinline__ void get_tile(const float* buf, int col, int h, f32x16& v) {   // what put_tile of the same lane wrote
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const float4 q = *reinterpret_cast<const float4*>(buf + col * TS + 8 * g + 4 * h);
    v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
  }
}
__device__ __forceinline__ void put_tile(float* buf, int col, int h, const f32x16& v) {
#pragma unroll
  for (int g = 0; g < 4; g++)    // registers 4g..4g+3 are units 8g + 4h .. + 3
    *reinterpret_cast<float4*>(buf + col * TS + 8 * g + 4 * h) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
// acc[i][j] += sum_a A[a][i] B[a][j] over the slab's 32 anchors; asum += this lane's column of A (bias gradient of unit i,
// half of the anchors per lane half)
template <int GROUP = 16>   // steps whose operand loads may be in flight together (the gradient wave of the pair kernel has no registers for 16)
__device__ __forceinline__ void wgrad_chain(const float* bufA, const float* bufB, int lane, f32x16& acc,
                                            float& asum) {
  const float* pa = bufA + (lane >> 5) * TS + (lane & 31);
  const float* pb = bufB + (lane >> 5) * TS + (lane & 31);
#pragma unroll
  for (int s = 0; s < 16; s++) {
    const float av = pa[2 * s * TS], bv = pb[2 * s * TS];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    asum += av;
    if (GROUP < 16 && (s + 1) % GROUP == 0) __builtin_amdgcn_sched_barrier(0);
  }
}
// the same for a first layer: B = the 32 features; the four tail inputs (view xyz, dist) are too few for a tile of their
// own and are accumulated on the VALU from the A values that are in registers anyway
template <int TSTRIDE = 4, int GROUP = 16>   // floats between the tail rows of consecutive anchors; see wgrad_chain
__device__ __forceinline__ void wgrad_chain_tail(const float* bufA, const float* bufB,
                                                 const float* bufT, int lane, f32x16& acc, float& asum, float* tl) {
  const int half = lane >> 5;
  const float* pa = bufA + half * TS + (lane & 31);
  const float* pb = bufB + half * TS + (lane & 31);
#pragma unroll
  for (int s = 0; s < 16; s++) {
    const float av = pa[2 * s * TS], bv = pb[2 * s * TS];
    const float4 t = *reinterpret_cast<const float4*>(bufT + (2 * s + half) * TSTRIDE);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    asum += av;
    tl[0] += av * t.x; tl[1] += av * t.y; tl[2] += av * t.z; tl[3] += av * t.w;
    if (GROUP < 16 && (s + 1) % GROUP == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

// End of one MLP in the backward pass: relu mask on dH = W2^T dOUT (accumulated tile by tile), dX (inputs 0..31) on the
// matrix cores and the view/dist tail on the VALU; dHpre goes to the wave's D tile for the first layer's weight gradient.
__device__ __forceinline__ void finish_mlp(const float* __restrict__ img, const Small& S, int m, int lane, int h, const f32x16& hpre,
                                           f32x16& dh, f32x16& dx, float* dtail, float* bufD) {
#pragma unroll
  for (int r = 0; r < 16; r++) dh[r] = hpre[r] > 0.f ? dh[r] : 0.f;
  put_tile(bufD, lane & 31, h, dh);
  const float* im = img + (I_DX + m * 16) * 64 + lane;
#pragma unroll
  for (int s = 0; s < 16; s++) dx = __builtin_amdgcn_mfma_f32_32x32x2f32(im[s * 64], dh[s], dx, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const float4 w = *reinterpret_cast<const float4*>(S.w1tail[m][rho(r, h)]);
    dtail[0] += w.x * dh[r]; dtail[1] += w.y * dh[r]; dtail[2] += w.z * dh[r]; dtail[3] += w.w * dh[r];
  }
}

template <bool BANK>
__global__ void __launch_bounds__(256, 1) neural_bwd_kernel(
    Layout L, const uint32_t* __restrict__ count, const uint32_t* __restrict__ vis, const float* __restrict__ anchor,
    const float* __restrict__ offset, const float* __restrict__ anchor_feat, const float* __restrict__ scaling_log,
    const float* __restrict__ g_img, const Small* __restrict__ g_small, const float* __restrict__ campos,
    const float* __restrict__ g_means, const float* __restrict__ g_colors, const float* __restrict__ g_opacity,
    const float* __restrict__ g_scales, const float* __restrict__ g_rot, float* __restrict__ d_anchor,
    float* __restrict__ d_offset, float* __restrict__ d_feat, float* __restrict__ d_scaling_log, float* __restrict__ rows,
    float* __restrict__ partial, float reg_weight, float* __restrict__ reg_sum) {
  extern __shared__ __align__(16) float lds_dyn[];
  float* img = lds_dyn;
  Small& S = *reinterpret_cast<Small*>(lds_dyn + N_IMG_BWD * 64);
  const uint32_t n = *count;
  if (blockIdx.x * 128u >= n) return;
  stage_tables(img, S, N_IMG_BWD, g_img, g_small);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = lane & 31, h = lane >> 5;
  float* const bufX = lds_dyn + N_IMG_BWD * 64 + sizeof(Small) / 4 + wv * WAVE_LDS;
  float* const bufH = bufX + T_TILE;
  float* const bufD = bufH + T_TILE;
  float* const bufT = bufD + T_TILE;
  // weight-gradient accumulators of this wave: second layers per output tile, first layers per MLP (feature columns)
  f32x16 aW2_0, aW2_1, aW2_2, aW2_3, aW2_4, aW1_0, aW1_1, aW1_2;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    aW2_0[r] = 0.f; aW2_1[r] = 0.f; aW2_2[r] = 0.f; aW2_3[r] = 0.f; aW2_4[r] = 0.f; aW1_0[r] = 0.f; aW1_1[r] = 0.f; aW1_2[r] = 0.f;
  }
  float sm[N_SMALL];    // [0..4] bias sums of the output tiles, [5..7] of the hidden layers, [8 + 4m + c] tail column c of MLP m
#pragma unroll
  for (int q = 0; q < N_SMALL; q++) sm[q] = 0.f;

  for (uint32_t g0 = (blockIdx.x * 4u + wv) * 32u; g0 < n; g0 += gridDim.x * 128u) {
    const uint32_t t = g0 + col;
    const bool valid = t < n;
    const uint32_t a = vis[valid ? t : n - 1];
    float* row = rows + (valid ? t : 0u) * ROW;
    AnchorLane st;
    anchor_lane<BANK>(S, L, a, h, anchor, anchor_feat, scaling_log, campos, st);
    if (BANK && valid) stn<L1_STEPS>(row + R_X + L1_STEPS * h, st.xo);   // the feature bank's Linear(4 -> 32) reads view, dist
    // X tile: unit 16 h + s holds input 2 s + h (this half's 16 features in one contiguous run); tail as one float4 per anchor
#pragma unroll
    for (int g = 0; g < 4; g++)
      *reinterpret_cast<float4*>(bufX + col * TS + 16 * h + 4 * g) = make_float4(st.xo[4 * g], st.xo[4 * g + 1], st.xo[4 * g + 2], st.xo[4 * g + 3]);
    if (h == 0) *reinterpret_cast<float4*>(bufT + col * 4) = make_float4(st.view[0], st.view[1], st.view[2], st.dist);
    const uint32_t c0 = a * NO + 5 * h;   // 32-bit element offsets (the entry points bound A): one SGPR base + one VGPR offset per access
    // With one wave per SIMD every dependent round trip to memory is exposed, and the loop made about twenty per slab (one
    // per field and candidate).  So the inputs of a tile are requested together and EARLY: opacity and colour gradients
    // here, next to the anchor's own data; a covariance tile's at the top of its iteration, in front of the MFMA chains;
    // the rows the slab's results are added to with the last tile.
    // (The feature-bank variant has no registers to spare for this: it requests each tile's inputs where they are used.)
    constexpr bool EARLY = !BANK;
    float in_op[5], in_col[15];
    if (EARLY) { ldn<5>(g_opacity + c0, in_op); ldn<15>(g_colors + c0 * 3, in_col); }
    float in_sc[6], in_mu[6], in_off[6], in_rot[8], acc_off[6];   // the (up to) two candidates of the current covariance tile
    float4* const dfo = reinterpret_cast<float4*>(d_feat + a * FD);
    float4 acc_feat[4];
    float acc_scl[6], acc_anc[3];
    f32x16 dx;
#pragma unroll
    for (int r = 0; r < 16; r++) dx[r] = 0.f;
    float dtail[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t keep = 0;   // bit r: candidate 5h + r has neural opacity > 0 (the reference's mask, :279)
    // scaling regulariser of the mapper loss, reg_weight * mean_P(prod(scaling)) (src/gaussian_mapper.cpp:926-928):
    // every kept candidate adds reg_weight / P * prod / s_c to dL/dscaling_c
    const float reg_w = reg_weight != 0.f ? reg_weight / (float)max(count[1], 1u) : 0.f;
    float reg_acc = 0.f;
    float dgs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, danc[3] = {0.f, 0.f, 0.f};
    f32x16 hp, dh;
#pragma unroll 1
    for (int tile = 0; tile < N_TILES; tile++) {
      const int m = tile_mlp(tile);
      auto request_cov_inputs = [&]() {
#pragma unroll
        for (int cl = 0; cl < 2; cl++) {
          const int cc = 2 * (tile - 2) + cl;
          if (cc < 5 && ((keep >> cc) & 1u)) {   // masked-out candidates are never read
            ldn<3>(g_scales + (c0 + cc) * 3, in_sc + 3 * cl);
            ldn<3>(g_means + (c0 + cc) * 3, in_mu + 3 * cl);
            ldn<3>(offset + (c0 + cc) * 3, in_off + 3 * cl);
            ldn<4>(g_rot + (c0 + cc) * 4, in_rot + 4 * cl);
            ldn<3>(d_offset + (c0 + cc) * 3, acc_off + 3 * cl);
          }
        }
      };
      auto request_accumulated_rows = [&]() {
#pragma unroll
        for (int g = 0; g < 4; g++) acc_feat[g] = dfo[2 * g + h];
        ldn<6>(d_scaling_log + a * 6, acc_scl);
        ldn<3>(d_anchor + a * 3, acc_anc);
      };
      if (EARLY && tile >= 2) {
        request_cov_inputs();
        if (tile == N_TILES - 1) request_accumulated_rows();
      }
      if (!EARLY) {   // nothing is carried from tile to tile: keeps the register allocator from holding these across the loop
#pragma unroll
        for (int q = 0; q < 6; q++) { in_sc[q] = 0.f; in_mu[q] = 0.f; in_off[q] = 0.f; acc_off[q] = 0.f; }
#pragma unroll
        for (int q = 0; q < 8; q++) in_rot[q] = 0.f;
      }
      if (tile <= 2) {
        hp = layer1(img, S, m, lane, h, st.xo);
#pragma unroll
        for (int r = 0; r < 16; r++) dh[r] = 0.f;
        f32x16 hr;
#pragma unroll
        for (int r = 0; r < 16; r++) hr[r] = fmaxf(hp[r], 0.f);
        put_tile(bufH, col, h, hr);
      }
      f32x16 o = layer2(img, S, tile, lane, h, hp);
      // ---- element-wise: outputs -> dL/d(output pre-activation), in place (zero for the padding lanes of the last slab:
      // they carry a copy of the last anchor and must not reach the weight gradients)
      if (tile == 0) {
        float d[5];
        if (!EARLY) ldn<5>(g_opacity + c0, in_op);
#pragma unroll
        for (int r = 0; r < 5; r++) {
          const float op = fast_tanh(o[r]);
          d[r] = 0.f;
          if (op > 0.f) { keep |= 1u << r; d[r] = in_op[r] * (1.f - op * op); }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = (r < 5 && valid) ? d[r < 5 ? r : 0] : 0.f;
      } else if (tile == 1) {
        float d[15];
        if (!EARLY) ldn<15>(g_colors + c0 * 3, in_col);
#pragma unroll
        for (int r = 0; r < 15; r++) {
          const float colv = sigmoidf(o[r]);
          d[r] = ((keep >> (r / 3)) & 1u) ? in_col[r] * colv * (1.f - colv) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = (r < 15 && valid) ? d[r < 15 ? r : 0] : 0.f;
      } else {
        if (!EARLY) request_cov_inputs();
#pragma unroll
        for (int cl = 0; cl < 2; cl++) {
          const int cc = 2 * (tile - 2) + cl;
          const bool on = cc < 5 && ((keep >> cc) & 1u);
          float dsr[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          float doff[3] = {0.f, 0.f, 0.f};
          if (on) {
            const float sr3 = o[7 * cl + 3], sr4 = o[7 * cl + 4], sr5 = o[7 * cl + 5], sr6 = o[7 * cl + 6];
            float gsc[3] = {in_sc[3 * cl], in_sc[3 * cl + 1], in_sc[3 * cl + 2]};
            const float gm[3] = {in_mu[3 * cl], in_mu[3 * cl + 1], in_mu[3 * cl + 2]};
            const float off[3] = {in_off[3 * cl], in_off[3 * cl + 1], in_off[3 * cl + 2]};
            if (reg_w != 0.f) {
              const float s0 = st.gs[3] * sigmoidf(o[7 * cl]), s1 = st.gs[4] * sigmoidf(o[7 * cl + 1]),
                          s2 = st.gs[5] * sigmoidf(o[7 * cl + 2]);
              gsc[0] += reg_w * (s1 * s2); gsc[1] += reg_w * (s0 * s2); gsc[2] += reg_w * (s0 * s1);
              reg_acc += s0 * s1 * s2;
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
              const float sg = sigmoidf(o[7 * cl + c]);
              dgs[3 + c] += gsc[c] * sg;
              dsr[c] = gsc[c] * st.gs[3 + c] * sg * (1.f - sg);
              danc[c] += gm[c];
              doff[c] = gm[c] * st.gs[c];
              dgs[c] += gm[c] * off[c];
            }
            const float4 gr = make_float4(in_rot[4 * cl], in_rot[4 * cl + 1], in_rot[4 * cl + 2], in_rot[4 * cl + 3]);
            const float nr = sqrtf(sr3 * sr3 + sr4 * sr4 + sr5 * sr5 + sr6 * sr6);
            if (nr >= 1e-12f) {   // r = v / |v|:  dv = (g - r (r.g)) / |v|
              const float inv = 1.0f / nr;
              const float r0 = sr3 * inv, r1 = sr4 * inv, r2 = sr5 * inv, r3 = sr6 * inv;
              const float dot = r0 * gr.x + r1 * gr.y + r2 * gr.z + r3 * gr.w;
              dsr[3] = (gr.x - r0 * dot) * inv; dsr[4] = (gr.y - r1 * dot) * inv;
              dsr[5] = (gr.z - r2 * dot) * inv; dsr[6] = (gr.w - r3 * dot) * inv;
            } else {              // clamp_min(|v|, eps) active: r = v / eps
              dsr[3] = gr.x * 1e12f; dsr[4] = gr.y * 1e12f; dsr[5] = gr.z * 1e12f; dsr[6] = gr.w * 1e12f;
            }
          }
          if (valid && on) {   // masked-out candidates add nothing
            const float cur[3] = {acc_off[3 * cl] + doff[0], acc_off[3 * cl + 1] + doff[1], acc_off[3 * cl + 2] + doff[2]};
            stn<3>(d_offset + (c0 + cc) * 3, cur);
          }
#pragma unroll
          for (int q = 0; q < 7; q++) o[7 * cl + q] = valid ? dsr[q] : 0.f;
        }
        o[14] = 0.f; o[15] = 0.f;
      }
      // ---- second-layer weight gradient of this tile: dOUT tile x H of the MLP
      put_tile(bufD, col, h, o);
      switch (tile) {    // wave-uniform; keeps every accumulator in statically named registers
        case 0: wgrad_chain(bufD, bufH, lane, aW2_0, sm[0]); break;
        case 1: wgrad_chain(bufD, bufH, lane, aW2_1, sm[1]); break;
        case 2: wgrad_chain(bufD, bufH, lane, aW2_2, sm[2]); break;
        case 3: wgrad_chain(bufD, bufH, lane, aW2_3, sm[3]); break;
        default: wgrad_chain(bufD, bufH, lane, aW2_4, sm[4]); break;
      }
      // ---- dH += W2^T dOUT for this tile
      {
        const float* im = img + (I_DH + tile * 16) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 16; s++) dh = __builtin_amdgcn_mfma_f32_32x32x2f32(im[s * 64], o[s], dh, 0, 0, 0);
      }
      if (tile != 2 && tile != 3) {
        finish_mlp(img, S, m, lane, h, hp, dh, dx, dtail, bufD);
          switch (m) {   // first-layer weight gradient: dHpre x (features | tail)
          case 0: wgrad_chain_tail(bufD, bufX, bufT, lane, aW1_0, sm[5], &sm[8]); break;
          case 1: wgrad_chain_tail(bufD, bufX, bufT, lane, aW1_1, sm[6], &sm[12]); break;
          default: wgrad_chain_tail(bufD, bufX, bufT, lane, aW1_2, sm[7], &sm[16]); break;
        }
        }
    }
    if (reg_sum != nullptr && reg_w != 0.f) {   // sum of prod(scaling) over the kept candidates, for the loss value
      reg_acc = valid ? reg_acc : 0.f;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) reg_acc += __shfl_xor(reg_acc, off, 64);
      if (lane == 0 && reg_acc != 0.f) atomicAdd(reg_sum, reg_acc);
    }
    // ---- combine the lane halves
#pragma unroll
    for (int c = 0; c < 6; c++) dgs[c] += other_half(dgs[c]);
#pragma unroll
    for (int c = 0; c < 3; c++) danc[c] += other_half(danc[c]);
#pragma unroll
    for (int c = 0; c < 4; c++) dtail[c] += other_half(dtail[c]);
    if (valid && h == 0) {
      float v[6];
      if (!EARLY) ldn<6>(d_scaling_log + a * 6, acc_scl);
#pragma unroll
      for (int c = 0; c < 6; c++) v[c] = acc_scl[c] + dgs[c] * st.gs[c];   // through exp()
      stn<6>(d_scaling_log + a * 6, v);
    }
    // dL/dfeat' for all 32 inputs: mine = rows rho(r,h), the other half's = rows rho(r,1-h)
    float dxf[FD];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float mine = dx[r], oth = other_half(dx[r]);
      dxf[rho(r, 0)] = h ? oth : mine;
      dxf[rho(r, 1)] = h ? mine : oth;
    }
    float dview[4] = {dtail[0], dtail[1], dtail[2], dtail[3]};   // view xyz, dist (weights are zero where unused)
    float df[FD];
    if (BANK) {
      float feat[FD];
      load_feat(anchor_feat, a, feat);
      float dbw[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < FD; j++) {
        dbw[0] += dxf[j] * feat[4 * (j % 8)];
        dbw[1] += dxf[j] * feat[2 * (j % 16)];
        dbw[2] += dxf[j] * feat[j];
        df[j] = st.bw[2] * dxf[j];
      }
#pragma unroll
      for (int j = 0; j < FD; j++) {
        df[4 * (j % 8)] += st.bw[0] * dxf[j];
        df[2 * (j % 16)] += st.bw[1] * dxf[j];
      }
      const float sdot = st.bw[0] * dbw[0] + st.bw[1] * dbw[1] + st.bw[2] * dbw[2];
      float dlg[3];
#pragma unroll
      for (int c = 0; c < 3; c++) dlg[c] = st.bw[c] * (dbw[c] - sdot);
      if (valid && h == 0) stn<3>(row + R_DF, dlg);
      // feature-bank hidden layer: this half's 16 units
      const float cat4[4] = {st.view[0], st.view[1], st.view[2], st.dist};
      float dv[4] = {0.f, 0.f, 0.f, 0.f};
      float hfv[FD / 2], dfv[FD / 2];
#pragma unroll
      for (int jj = 0; jj < FD / 2; jj++) {
        const int j = 16 * h + jj;
        const float4 w = *reinterpret_cast<const float4*>(S.fw1[j]);
        const float s = S.fb1[j] + w.x * cat4[0] + w.y * cat4[1] + w.z * cat4[2] + w.w * cat4[3];
        const float4 u = *reinterpret_cast<const float4*>(S.fw2[j]);
        float d = u.x * dlg[0] + u.y * dlg[1] + u.z * dlg[2];
        d = s > 0.f ? d : 0.f;
        hfv[jj] = fmaxf(s, 0.f); dfv[jj] = d;
        dv[0] += w.x * d; dv[1] += w.y * d; dv[2] += w.z * d; dv[3] += w.w * d;
      }
      if (valid) { stn<FD / 2>(row + R_H + 3 * FD + 16 * h, hfv); stn<FD / 2>(row + R_DH + 3 * FD + 16 * h, dfv); }
#pragma unroll
      for (int c = 0; c < 4; c++) dview[c] += dv[c] + other_half(dv[c]);
    } else {
#pragma unroll
      for (int j = 0; j < FD; j++) df[j] = dxf[j];
    }
    if (valid) {   // each half adds its own 16 features: float4 groups 8g + 4h
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float4 v = EARLY ? acc_feat[g] : dfo[2 * g + h];
        v.x += h ? df[8 * g + 4] : df[8 * g]; v.y += h ? df[8 * g + 5] : df[8 * g + 1];
        v.z += h ? df[8 * g + 6] : df[8 * g + 2]; v.w += h ? df[8 * g + 7] : df[8 * g + 3];
        dfo[2 * g + h] = v;
      }
    }
    // view = ob / |ob|, dist = |ob|:  d ob = (dview - view (view . dview)) / dist + ddist * view
    if (valid && h == 0) {
      const float vx = st.view[0], vy = st.view[1], vz = st.view[2];
      const float dot = vx * dview[0] + vy * dview[1] + vz * dview[2];
      if (!EARLY) ldn<3>(d_anchor + a * 3, acc_anc);
      const float v[3] = {acc_anc[0] + danc[0] + (dview[0] - vx * dot) * st.inv_dist + dview[3] * vx,
                          acc_anc[1] + danc[1] + (dview[1] - vy * dot) * st.inv_dist + dview[3] * vy,
                          acc_anc[2] + danc[2] + (dview[2] - vz * dot) * st.inv_dist + dview[3] * vz};
      stn<3>(d_anchor + a * 3, v);
    }
  }

  // ---- the workgroup's weight-gradient partials: the four waves' accumulators summed in a fixed order through LDS (the
  // operand images are no longer needed), written in wgrad_reduce_kernel's partial-tile layout, slot = workgroup
  __syncthreads();
  float* const stage = lds_dyn;                       // [4 waves][max(1024, N_SMALL * 64)] floats
  constexpr int STAGE = N_SMALL * 64 > 1024 ? N_SMALL * 64 : 1024;
  const int tid = threadIdx.x;
  auto slot = [&](int job) { return partial + ((size_t)job * WG_WAVES + blockIdx.x) * WG_TILE; };
#pragma unroll 1
  for (int k = 0; k < 8; k++) {
    f32x16 v;
    switch (k) {
      case 0: v = aW2_0; break; case 1: v = aW2_1; break; case 2: v = aW2_2; break; case 3: v = aW2_3; break;
      case 4: v = aW2_4; break; case 5: v = aW1_0; break; case 6: v = aW1_1; break; default: v = aW1_2; break;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) stage[wv * STAGE + r * 64 + lane] = v[r];
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
      const float sum = (stage[e] + stage[STAGE + e]) + (stage[2 * STAGE + e] + stage[3 * STAGE + e]);
      const int r = e >> 6, l = e & 63, i = rho(r, l >> 5), j = l & 31;   // D[i][j]
      if (k < 5) {            // second layer, tile k: i = tile row -> output unit, j = hidden unit (the Linear's input)
        const int o = out_row(k, (i >> 2) & 1, (i & 3) + 4 * (i >> 3));
        if (o >= 0) slot(3 + tile_mlp(k))[(((o >> 5)) * 32 + j) * 32 + (o & 31)] = sum;
      } else {                // first layer of MLP k - 5: i = hidden unit, j = feature position 16 (input & 1) + (input >> 1)
        const int input = 2 * (j & 15) + (j >> 4);
        slot(k - 5)[(size_t)input * 32 + i] = sum;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < N_SMALL; q++) stage[wv * STAGE + q * 64 + lane] = sm[q];
  __syncthreads();
  for (int e = tid; e < N_SMALL * 32; e += 256) {
    const int q = e >> 5, u = e & 31;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; w++) sum += stage[w * STAGE + q * 64 + u] + stage[w * STAGE + q * 64 + 32 + u];   // both lane halves
    if (q < 5) {              // bias of output tile q: u = tile row
      const int o = out_row(q, (u >> 2) & 1, (u & 3) + 4 * (u >> 3));
      if (o >= 0) slot(3 + tile_mlp(q))[((1 * 3 + (o >> 5)) * 32 + 31) * 32 + (o & 31)] = sum;
    } else if (q < 8) {       // bias of the hidden layer of MLP q - 5: u = hidden unit
      slot(q - 5)[((1 * 3 + 0) * 32 + 31) * 32 + u] = sum;
    } else {                  // tail input 32 + c of MLP (q - 8) / 4
      const int mm = (q - 8) >> 2, c = (q - 8) & 3;
      slot(mm)[((1 * 3 + 0) * 32 + c) * 32 + u] = sum;
    }
  }
}

// ---- the plain model's backward: chain waves and weight-gradient waves ------------------------------------------------
// The one-kernel form above keeps 128 accumulator registers next to the chain's own state in ONE wave: nothing else fits on its
// SIMD, the matrix pipe idles while that wave does element-wise work or waits for memory (34 % busy at 300 k anchors), and a
// third of its vector instructions are register copies between the two halves of the 512-register file.  Here a workgroup is
// four PAIRS of waves, a pair per SIMD:
//   chain wave  (waves 0-3): forward recompute, element-wise backward, dH and dX chains -- 262 MFMAs per slab, no accumulators;
//   wgrad wave  (waves 4-7): the eight weight-gradient products of the slab -- 128 MFMAs, the 128 accumulators, nothing else.
// The chain wave hands every operand over as a transposition tile in LDS (the tiles existed already); a workgroup barrier per
// product ("job") orders the hand-over: the chain wave writes tile D[j & 1] of job j and passes barrier j, the wgrad wave runs
// job j after barrier j and arrives at barrier j + 1 when it is done -- so job j's MFMAs run under the chain wave's work on job
// j + 1.  Single-buffered tiles are written where the partner cannot be reading them: H of an MLP after the barrier of the
// previous MLP's first-layer job (which reads D, X, T), X and T of a slab after the slab's first barrier (the previous slab's
// last job is over then).  All eight waves of a workgroup run the same number of rounds, so the barriers match.
constexpr int PAIR_LDS = 4 * T_TILE + 32 * 4;   // X | H | D0 | D1 | tail(view xyz, dist), floats per pair of waves
constexpr int PAIR_STAGE = 8 * 1024 + N_SMALL * 64;   // end of kernel: a wgrad wave's eight tiles and its scalar sums

// BANK (round 4): the feature-bank model (the Replica configurations) takes the same pairs.  Its chain wave mixes the features
// at the top of the slab (anchor_lane<true>) and, in its epilogue, takes dL/d(mixed features) back through the mixing and the
// softmax to the bank's two small Linears 4 -> 32 -> 3; only THEIR weight gradients still go through per-anchor scratch rows
// (103 of the row's 512 floats) to wgrad_mfma_kernel.  Until round 4 this model ran the one-kernel form: 240 + 91 us at 200 k
// anchors against 153 us for the plain model's pairs.
template <bool BANK>
__global__ void __launch_bounds__(512, 1) neural_bwd_pair_kernel(
    Layout L, const uint32_t* __restrict__ count, const uint32_t* __restrict__ vis, const float* __restrict__ anchor,
    const float* __restrict__ offset, const float* __restrict__ anchor_feat, const float* __restrict__ scaling_log,
    const float* __restrict__ g_img, const Small* __restrict__ g_small, const float* __restrict__ campos,
    const float* __restrict__ g_means, const float* __restrict__ g_colors, const float* __restrict__ g_opacity,
    const float* __restrict__ g_scales, const float* __restrict__ g_rot, float* __restrict__ d_anchor,
    float* __restrict__ d_offset, float* __restrict__ d_feat, float* __restrict__ d_scaling_log, float* __restrict__ rows,
    float* __restrict__ partial, float reg_weight, float* __restrict__ reg_sum) {
  extern __shared__ __align__(16) float lds_dyn[];
  float* img = lds_dyn;
  Small& S = *reinterpret_cast<Small*>(lds_dyn + N_IMG_BWD * 64);
  const uint32_t n = *count;
  if (blockIdx.x * 128u >= n) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int pair = wv & 3;          // waves w and w + 4 of a workgroup share SIMD w (measured: the even/odd pairing is 9 % slower)
  const bool chain_role = wv < 4;   // wave-uniform
  const int col = lane & 31, h = lane >> 5;
  const uint32_t a_first = vis[min((blockIdx.x * 4u + (uint32_t)pair) * 32u + (uint32_t)col, n - 1u)];   // in flight under the copy of the tables
  stage_tables(img, S, N_IMG_BWD, g_img, g_small);
  float* const bufX = lds_dyn + N_IMG_BWD * 64 + sizeof(Small) / 4 + pair * PAIR_LDS;
  float* const bufH = bufX + T_TILE;
  float* const bufD0 = bufH + T_TILE;
  float* const bufD1 = bufD0 + T_TILE;
  float* const bufT = bufD1 + T_TILE;
  const uint32_t per_round = gridDim.x * 128u;
  const uint32_t rounds = (n - blockIdx.x * 128u + per_round - 1u) / per_round;   // of this workgroup: the same for its eight waves
  float* const stage = lds_dyn;   // [4 wgrad waves][PAIR_STAGE], after the last round

  if (chain_role) {
    // Memory round trips are what a chain wave waits for.  The next slab's anchor index is requested during tile 1 (one register);
    // opacity and colour gradients at the top of the slab next to the anchor's own data; a covariance tile's five fields at the top
    // of its iteration, in front of the MFMA chains; the rows the slab's results are added to with the last tile.  (Requesting the
    // next slab's anchor data or a covariance tile's inputs a tile ahead needs 40 / 32 registers more than there are: the spills
    // cost more than the round trips -- 0.24 -> 0.30-0.37 ms, profiles/r03_neural_bwd_pair_notes.txt.)
    auto slab_lane = [&](uint32_t rd) { return (rd * gridDim.x + blockIdx.x) * 128u + (uint32_t)pair * 32u + (uint32_t)col; };
    uint32_t a_nx = a_first;
    for (uint32_t rd = 0; rd < rounds; rd++) {
      const uint32_t t = slab_lane(rd);
      const bool valid = t < n;   // a pair's last round may be empty: it runs on a copy of the last anchor and stores nothing
      const uint32_t a = a_nx;
      AnchorLane st;
      if (BANK) asm volatile("" ::: "memory");   // the bank's weights are re-read from LDS per slab: hoisted out of the loop they cost 45 spilled registers
      anchor_lane<BANK>(S, L, a, h, anchor, anchor_feat, scaling_log, campos, st);
      float* const row = rows + (valid ? t : 0u) * BROW;
      if (BANK && valid) stn<L1_STEPS>(row + RB_X + L1_STEPS * h, st.xo);   // the feature bank's Linear(4 -> 32) reads view, dist
      const uint32_t c0 = a * NO + 5 * h;   // 32-bit element offsets (the entry points bound A): one SGPR base + one VGPR offset per access
      float in_op[5], in_col[15];
      ldn<5>(g_opacity + c0, in_op);
      ldn<15>(g_colors + c0 * 3, in_col);
      float in_sc[6], in_mu[6], in_off[6], in_rot[8], acc_off[6];   // the (up to) two candidates of the current covariance tile
      float4* const dfo = reinterpret_cast<float4*>(d_feat + a * FD);
      float4 acc_feat[4];
      float acc_scl[6], acc_anc[3];
      f32x16 dx;
#pragma unroll
      for (int r = 0; r < 16; r++) dx[r] = 0.f;
      float dtail[4] = {0.f, 0.f, 0.f, 0.f};
      uint32_t keep = 0;   // bit r: candidate 5h + r has neural opacity > 0 (the reference's mask, :279)
      // scaling regulariser of the mapper loss, reg_weight * mean_P(prod(scaling)) (src/gaussian_mapper.cpp:926-928):
      // every kept candidate adds reg_weight / P * prod / s_c to dL/dscaling_c
      const float reg_w = reg_weight != 0.f ? reg_weight / (float)max(count[1], 1u) : 0.f;
      float reg_acc = 0.f;
      float dgs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, danc[3] = {0.f, 0.f, 0.f};
      f32x16 hp, dh;
      int job = 0;   // jobs handed to the wgrad wave in this slab: t0 f0 t1 f2 t2 t3 t4 f1
#pragma unroll 1
      for (int tile = 0; tile < N_TILES; tile++) {
        const int m = tile_mlp(tile);
        if (tile >= 2) {
#pragma unroll
          for (int cl = 0; cl < 2; cl++) {
            const int cc = 2 * (tile - 2) + cl;
            if (cc < 5 && ((keep >> cc) & 1u)) {   // masked-out candidates are never read
              ldn<3>(g_scales + (c0 + cc) * 3, in_sc + 3 * cl);
              ldn<3>(g_means + (c0 + cc) * 3, in_mu + 3 * cl);
              ldn<3>(offset + (c0 + cc) * 3, in_off + 3 * cl);
              ldn<4>(g_rot + (c0 + cc) * 4, in_rot + 4 * cl);
              ldn<3>(d_offset + (c0 + cc) * 3, acc_off + 3 * cl);
            }
          }
        }
        if (tile == 1) a_nx = vis[min(slab_lane(rd + 1), n - 1u)];
        if (tile == N_TILES - 1) {
#pragma unroll
          for (int g = 0; g < 4; g++) acc_feat[g] = dfo[2 * g + h];
          ldn<6>(d_scaling_log + a * 6, acc_scl);
          ldn<3>(d_anchor + a * 3, acc_anc);
        }
        if (tile <= 2) {
          hp = layer1(img, S, m, lane, h, st.xo);
#pragma unroll
          for (int r = 0; r < 16; r++) dh[r] = 0.f;
          f32x16 hr;
#pragma unroll
          for (int r = 0; r < 16; r++) hr[r] = fmaxf(hp[r], 0.f);
          put_tile(bufH, col, h, hr);   // (the partner is in a first-layer job or idle: it reads D, X, T)
        }
        f32x16 o = layer2(img, S, tile, lane, h, hp);
        // ---- element-wise: outputs -> dL/d(output pre-activation), in place (zero for the padding lanes of the last slab:
        // they carry a copy of the last anchor and must not reach the weight gradients)
        if (tile == 0) {
          float d[5];
#pragma unroll
          for (int r = 0; r < 5; r++) {
            const float op = fast_tanh(o[r]);
            d[r] = 0.f;
            if (op > 0.f) { keep |= 1u << r; d[r] = in_op[r] * (1.f - op * op); }
          }
#pragma unroll
          for (int r = 0; r < 16; r++) o[r] = (r < 5 && valid) ? d[r < 5 ? r : 0] : 0.f;
        } else if (tile == 1) {
          float d[15];
#pragma unroll
          for (int r = 0; r < 15; r++) {
            const float colv = sigmoidf(o[r]);
            d[r] = ((keep >> (r / 3)) & 1u) ? in_col[r] * colv * (1.f - colv) : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; r++) o[r] = (r < 15 && valid) ? d[r < 15 ? r : 0] : 0.f;
        } else {
#pragma unroll
          for (int cl = 0; cl < 2; cl++) {
            const int cc = 2 * (tile - 2) + cl;
            const bool on = cc < 5 && ((keep >> cc) & 1u);
            float dsr[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float doff[3] = {0.f, 0.f, 0.f};
            if (on) {
              const float sr3 = o[7 * cl + 3], sr4 = o[7 * cl + 4], sr5 = o[7 * cl + 5], sr6 = o[7 * cl + 6];
              float gsc[3] = {in_sc[3 * cl], in_sc[3 * cl + 1], in_sc[3 * cl + 2]};
              const float gm[3] = {in_mu[3 * cl], in_mu[3 * cl + 1], in_mu[3 * cl + 2]};
              const float off[3] = {in_off[3 * cl], in_off[3 * cl + 1], in_off[3 * cl + 2]};
              float sg[3];
#pragma unroll
              for (int c = 0; c < 3; c++) sg[c] = sigmoidf(o[7 * cl + c]);
              if (reg_w != 0.f) {
                const float s0 = st.gs[3] * sg[0], s1 = st.gs[4] * sg[1], s2 = st.gs[5] * sg[2];
                gsc[0] += reg_w * (s1 * s2); gsc[1] += reg_w * (s0 * s2); gsc[2] += reg_w * (s0 * s1);
                reg_acc += s0 * s1 * s2;
              }
#pragma unroll
              for (int c = 0; c < 3; c++) {
                dgs[3 + c] += gsc[c] * sg[c];
                dsr[c] = gsc[c] * st.gs[3 + c] * sg[c] * (1.f - sg[c]);
                danc[c] += gm[c];
                doff[c] = gm[c] * st.gs[c];
                dgs[c] += gm[c] * off[c];
              }
              const float4 gr = make_float4(in_rot[4 * cl], in_rot[4 * cl + 1], in_rot[4 * cl + 2], in_rot[4 * cl + 3]);
              const float nr = sqrtf(sr3 * sr3 + sr4 * sr4 + sr5 * sr5 + sr6 * sr6);
              if (nr >= 1e-12f) {   // r = v / |v|:  dv = (g - r (r.g)) / |v|
                const float inv = 1.0f / nr;
                const float r0 = sr3 * inv, r1 = sr4 * inv, r2 = sr5 * inv, r3 = sr6 * inv;
                const float dot = r0 * gr.x + r1 * gr.y + r2 * gr.z + r3 * gr.w;
                dsr[3] = (gr.x - r0 * dot) * inv; dsr[4] = (gr.y - r1 * dot) * inv;
                dsr[5] = (gr.z - r2 * dot) * inv; dsr[6] = (gr.w - r3 * dot) * inv;
              } else {              // clamp_min(|v|, eps) active: r = v / eps
                dsr[3] = gr.x * 1e12f; dsr[4] = gr.y * 1e12f; dsr[5] = gr.z * 1e12f; dsr[6] = gr.w * 1e12f;
              }
            }
            if (valid && on) {   // masked-out candidates add nothing
              const float cur[3] = {acc_off[3 * cl] + doff[0], acc_off[3 * cl + 1] + doff[1], acc_off[3 * cl + 2] + doff[2]};
              stn<3>(d_offset + (c0 + cc) * 3, cur);
            }
#pragma unroll
            for (int q = 0; q < 7; q++) o[7 * cl + q] = valid ? dsr[q] : 0.f;
          }
          o[14] = 0.f; o[15] = 0.f;
        }
        // ---- job "t": dOUT tile x H of the MLP, the partner's
        put_tile((job & 1) ? bufD1 : bufD0, col, h, o);
        __syncthreads();
        job++;
        if (tile == 0) {
          // X tile: unit 16 h + s holds input 2 s + h (this half's 16 features in one contiguous run); tail as one float4 per
          // anchor.  Written here: the partner has finished the previous slab's last job, which read them.
#pragma unroll
          for (int g = 0; g < 4; g++)
            *reinterpret_cast<float4*>(bufX + col * TS + 16 * h + 4 * g) = make_float4(st.xo[4 * g], st.xo[4 * g + 1], st.xo[4 * g + 2], st.xo[4 * g + 3]);
          if (h == 0) *reinterpret_cast<float4*>(bufT + col * 4) = make_float4(st.view[0], st.view[1], st.view[2], st.dist);
        }
        // ---- dH += W2^T dOUT for this tile: step s contracts the tile rows held in register s of the two lane halves, and only
        // registers 0..4 | 0..14 | 0..13 | 0..13 | 0..6 of the five tiles hold output units (out_row): 55 MFMAs instead of 80
        {
          const float* im = img + (I_DH + tile * 16) * 64 + lane;
          const int steps = tile == 0 ? 5 : (tile == 1 ? 15 : (tile == 4 ? 7 : 14));
#pragma unroll
          for (int s = 0; s < 16; s++)
            if (s < 5 || s < steps) dh = __builtin_amdgcn_mfma_f32_32x32x2f32(im[s * 64], o[s], dh, 0, 0, 0);
        }
        if (tile != 2 && tile != 3) {
          // end of MLP m: relu mask, job "f" (dHpre x (features | tail)) for the partner, then dX on the matrix cores and the
          // view/dist tail on the VALU
#pragma unroll
          for (int r = 0; r < 16; r++) dh[r] = hp[r] > 0.f ? dh[r] : 0.f;
          put_tile((job & 1) ? bufD1 : bufD0, col, h, dh);
          __syncthreads();
          job++;
          const float* im = img + (I_DX + m * 16) * 64 + lane;
#pragma unroll
          for (int s = 0; s < 16; s++) dx = __builtin_amdgcn_mfma_f32_32x32x2f32(im[s * 64], dh[s], dx, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const float4 w = *reinterpret_cast<const float4*>(S.w1tail[m][rho(r, h)]);
            dtail[0] += w.x * dh[r]; dtail[1] += w.y * dh[r]; dtail[2] += w.z * dh[r]; dtail[3] += w.w * dh[r];
          }
        }
      }
      if (reg_sum != nullptr && reg_w != 0.f) {   // sum of prod(scaling) over the kept candidates, for the loss value
        reg_acc = valid ? reg_acc : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) reg_acc += __shfl_xor(reg_acc, off, 64);
        if (lane == 0 && reg_acc != 0.f) atomicAdd(reg_sum, reg_acc);
      }
      // ---- combine the lane halves
#pragma unroll
      for (int c = 0; c < 6; c++) dgs[c] += other_half(dgs[c]);
#pragma unroll
      for (int c = 0; c < 3; c++) danc[c] += other_half(danc[c]);
#pragma unroll
      for (int c = 0; c < 4; c++) dtail[c] += other_half(dtail[c]);
      if (valid && h == 0) {
        float v[6];
#pragma unroll
        for (int c = 0; c < 6; c++) v[c] = acc_scl[c] + dgs[c] * st.gs[c];   // through exp()
        stn<6>(d_scaling_log + a * 6, v);
      }
      if (!BANK) {
        if (valid) {   // dL/dfeat: each half adds its own 16 features: float4 group 2g + h is rows rho(4g + q, h), this half's registers 4g + q
#pragma unroll
          for (int g = 0; g < 4; g++) {
            float4 v = acc_feat[g];
            v.x += dx[4 * g]; v.y += dx[4 * g + 1]; v.z += dx[4 * g + 2]; v.w += dx[4 * g + 3];
            dfo[2 * g + h] = v;
          }
        }
      } else {
        // dL/dfeat' for all 32 mixed features (mine = rows rho(r,h), the other half's = rows rho(r,1-h)) back through the mixing
        // feat'[k] = feat[4 (k%8)] bw0 + feat[2 (k%16)] bw1 + feat[k] bw2 and the softmax to the bank's Linears.  Written to keep
        // few values alive at once (the chain wave has 256 registers; the straightforward form spilled 96 of them): the features
        // come back one float4 at a time, the transposed mixing is formed from partial sums of dxf, and every lane only forms the
        // 16 entries of dL/dfeat it stores.
        float dxf[FD];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float mine = dx[r], oth = other_half(dx[r]);
          dxf[rho(r, 0)] = h ? oth : mine;
          dxf[rho(r, 1)] = h ? mine : oth;
        }
        float s2[16], s4[8];   // s2[m] = dxf[m] + dxf[m + 16]  (what feat[2m] sees through bw1);  s4[k] = sum_{j % 8 == k} dxf[j]  (feat[4k], bw0)
#pragma unroll
        for (int m2 = 0; m2 < 16; m2++) s2[m2] = dxf[m2] + dxf[m2 + 16];
#pragma unroll
        for (int k = 0; k < 8; k++) s4[k] = s2[k] + s2[k + 8];
        float dbw[3] = {0.f, 0.f, 0.f};
        const float4* const fp4 = reinterpret_cast<const float4*>(anchor_feat + a * FD);
#pragma unroll
        for (int q = 0; q < FD / 4; q++) {
          const float4 f = fp4[q];
          dbw[2] += dxf[4 * q] * f.x + dxf[4 * q + 1] * f.y + dxf[4 * q + 2] * f.z + dxf[4 * q + 3] * f.w;
          dbw[1] += s2[2 * q] * f.x + s2[2 * q + 1] * f.z;
          dbw[0] += s4[q] * f.x;
        }
        float dlg[3];
        {
          const float sdot = st.bw[0] * dbw[0] + st.bw[1] * dbw[1] + st.bw[2] * dbw[2];
#pragma unroll
          for (int c = 0; c < 3; c++) dlg[c] = st.bw[c] * (dbw[c] - sdot);
        }
        if (valid && h == 0) stn<3>(row + RB_DF, dlg);
        if (valid) {   // each half adds its own 16 features: float4 groups 2g + h = features 8g + 4h .. + 3
#pragma unroll
          for (int g = 0; g < 4; g++) {
            float4 v = acc_feat[g];
            float add[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              // feature index i = 8g + 4h + q: df[i] = bw2 dxf[i] + (i % 2 == 0) bw1 s2[i / 2] + (i % 4 == 0) bw0 s4[i / 4]
              const int i0 = 8 * g + q, i1 = 8 * g + 4 + q;
              float a0 = st.bw[2] * dxf[i0], a1 = st.bw[2] * dxf[i1];
              if (q % 2 == 0) { a0 += st.bw[1] * s2[i0 / 2]; a1 += st.bw[1] * s2[i1 / 2]; }
              if (q == 0) { a0 += st.bw[0] * s4[i0 / 4]; a1 += st.bw[0] * s4[i1 / 4]; }
              add[q] = h ? a1 : a0;
            }
            v.x += add[0]; v.y += add[1]; v.z += add[2]; v.w += add[3];
            dfo[2 * g + h] = v;
          }
        }
        // feature-bank hidden layer: this half's 16 units, four at a time
        const float cat4[4] = {st.view[0], st.view[1], st.view[2], st.dist};
        float dv[4] = {0.f, 0.f, 0.f, 0.f};
        int hb = h;
        asm volatile("" : "+v"(hb));   // see bank_logits_partial
#pragma unroll
        for (int g4 = 0; g4 < FD / 8; g4++) {
          float hq[4], dq[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int j = 16 * hb + 4 * g4 + q;
            const float4 w = *reinterpret_cast<const float4*>(S.fw1[j]);
            const float sj = S.fb1[j] + w.x * cat4[0] + w.y * cat4[1] + w.z * cat4[2] + w.w * cat4[3];
            const float4 u = *reinterpret_cast<const float4*>(S.fw2[j]);
            float d = u.x * dlg[0] + u.y * dlg[1] + u.z * dlg[2];
            d = sj > 0.f ? d : 0.f;
            hq[q] = fmaxf(sj, 0.f); dq[q] = d;
            dv[0] += w.x * d; dv[1] += w.y * d; dv[2] += w.z * d; dv[3] += w.w * d;
          }
          if (valid) {
            *reinterpret_cast<float4*>(row + RB_H + 16 * h + 4 * g4) = make_float4(hq[0], hq[1], hq[2], hq[3]);
            *reinterpret_cast<float4*>(row + RB_DH + 16 * h + 4 * g4) = make_float4(dq[0], dq[1], dq[2], dq[3]);
          }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) dtail[c] += dv[c] + other_half(dv[c]);
      }
      // view = ob / |ob|, dist = |ob|:  d ob = (dview - view (view . dview)) / dist + ddist * view
      if (valid && h == 0) {
        const float vx = st.view[0], vy = st.view[1], vz = st.view[2];
        const float dot = vx * dtail[0] + vy * dtail[1] + vz * dtail[2];
        const float v[3] = {acc_anc[0] + danc[0] + (dtail[0] - vx * dot) * st.inv_dist + dtail[3] * vx,
                            acc_anc[1] + danc[1] + (dtail[1] - vy * dot) * st.inv_dist + dtail[3] * vy,
                            acc_anc[2] + danc[2] + (dtail[2] - vz * dot) * st.inv_dist + dtail[3] * vz};
        stn<3>(d_anchor + a * 3, v);
      }
    }
    __syncthreads();   // every wave is done with the operand images and the tiles
  } else {
    // weight-gradient accumulators: second layers per output tile, first layers per MLP (feature columns)
    f32x16 aW2_0, aW2_1, aW2_2, aW2_3, aW2_4, aW1_0, aW1_1, aW1_2;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      aW2_0[r] = 0.f; aW2_1[r] = 0.f; aW2_2[r] = 0.f; aW2_3[r] = 0.f; aW2_4[r] = 0.f; aW1_0[r] = 0.f; aW1_1[r] = 0.f; aW1_2[r] = 0.f;
    }
    float sm[N_SMALL];    // [0..4] bias sums of the output tiles, [5..7] of the hidden layers, [8 + 4m + c] tail column c of MLP m
#pragma unroll
    for (int q = 0; q < N_SMALL; q++) sm[q] = 0.f;
    for (uint32_t rd = 0; rd < rounds; rd++) {
      __syncthreads(); wgrad_chain(bufD0, bufH, lane, aW2_0, sm[0]);                          // t0
      __syncthreads(); wgrad_chain_tail(bufD1, bufX, bufT, lane, aW1_0, sm[5], &sm[8]);       // f0
      __syncthreads(); wgrad_chain(bufD0, bufH, lane, aW2_1, sm[1]);                          // t1
      __syncthreads(); wgrad_chain_tail(bufD1, bufX, bufT, lane, aW1_2, sm[7], &sm[16]);      // f2
      __syncthreads(); wgrad_chain(bufD0, bufH, lane, aW2_2, sm[2]);                          // t2
      __syncthreads(); wgrad_chain(bufD1, bufH, lane, aW2_3, sm[3]);                          // t3
      __syncthreads(); wgrad_chain(bufD0, bufH, lane, aW2_4, sm[4]);                          // t4
      __syncthreads(); wgrad_chain_tail(bufD1, bufX, bufT, lane, aW1_1, sm[6], &sm[12]);      // f1
    }
    __syncthreads();   // every wave is done with the operand images and the tiles
    float* const mine = stage + pair * PAIR_STAGE;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      mine[0 * 1024 + r * 64 + lane] = aW2_0[r]; mine[1 * 1024 + r * 64 + lane] = aW2_1[r]; mine[2 * 1024 + r * 64 + lane] = aW2_2[r];
      mine[3 * 1024 + r * 64 + lane] = aW2_3[r]; mine[4 * 1024 + r * 64 + lane] = aW2_4[r]; mine[5 * 1024 + r * 64 + lane] = aW1_0[r];
      mine[6 * 1024 + r * 64 + lane] = aW1_1[r]; mine[7 * 1024 + r * 64 + lane] = aW1_2[r];
    }
#pragma unroll
    for (int q = 0; q < N_SMALL; q++) mine[8 * 1024 + q * 64 + lane] = sm[q];
  }
  __syncthreads();
  // ---- the workgroup's weight-gradient partials: the four wgrad waves' accumulators summed in a fixed order, written in
  // wgrad_reduce_kernel's partial-tile layout, slot = workgroup
  const int tid = threadIdx.x;
  auto slot = [&](int job) { return partial + ((size_t)job * WG_WAVES + blockIdx.x) * WG_TILE; };
  for (int e8 = tid; e8 < 8 * 1024; e8 += 512) {
    const int k = e8 >> 10, e = e8 & 1023;
    const float sum = (stage[e8] + stage[PAIR_STAGE + e8]) + (stage[2 * PAIR_STAGE + e8] + stage[3 * PAIR_STAGE + e8]);
    const int r = e >> 6, l = e & 63, i = rho(r, l >> 5), j = l & 31;   // D[i][j]
    if (k < 5) {            // second layer, tile k: i = tile row -> output unit, j = hidden unit (the Linear's input)
      const int o = out_row(k, (i >> 2) & 1, (i & 3) + 4 * (i >> 3));
      if (o >= 0) slot(3 + tile_mlp(k))[(((o >> 5)) * 32 + j) * 32 + (o & 31)] = sum;
    } else {                // first layer of MLP k - 5: i = hidden unit, j = feature position 16 (input & 1) + (input >> 1)
      const int input = 2 * (j & 15) + (j >> 4);
      slot(k - 5)[(size_t)input * 32 + i] = sum;
    }
  }
  for (int e = tid; e < N_SMALL * 32; e += 512) {
    const int q = e >> 5, u = e & 31;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; w++) sum += stage[w * PAIR_STAGE + 8 * 1024 + q * 64 + u] + stage[w * PAIR_STAGE + 8 * 1024 + q * 64 + 32 + u];   // both lane halves
    if (q < 5) {              // bias of output tile q: u = tile row
      const int o = out_row(q, (u >> 2) & 1, (u & 3) + 4 * (u >> 3));
      if (o >= 0) slot(3 + tile_mlp(q))[((1 * 3 + (o >> 5)) * 32 + 31) * 32 + (o & 31)] = sum;
    } else if (q < 8) {       // bias of the hidden layer of MLP q - 5: u = hidden unit
      slot(q - 5)[((1 * 3 + 0) * 32 + 31) * 32 + u] = sum;
    } else {                  // tail input 32 + c of MLP (q - 8) / 4
      const int mm = (q - 8) >> 2, c = (q - 8) & 3;
      slot(mm)[((1 * 3 + 0) * 32 + c) * 32 + u] = sum;
    }
  }
}

// ---- weight gradients ---------------------------------------------------------------------------------------------
struct WJob {
  int a_off, M;       // activation field of the scratch row (the Linear's input), M columns
  int a_x0;           // >= 0: the field is x in lane order and column i is input a_x0 + i at position 18 (k&1) + (k>>1)
  int b_off, N;       // pre-activation gradient field (the Linear's output), N stored columns
  int b_half;         // > 0: stored column j = b_half * h + r is output b_real * h + r for r < b_real (else padding)
  int b_real;
  int w_off, ldw;     // dW[j][i] -> gsum[w_off + j*ldw + i]
  int bias_off;       // db[j]   -> gsum[bias_off + j]
  int active;
  int via_rows;       // 1: partial tiles come from wgrad_mfma_kernel (one per wave, WG_WAVES of them) over the scratch rows;
                      // 0: from neural_bwd_kernel itself (one per workgroup that had anchors to process)
};
struct WJobs { WJob j[WG_JOBS]; };

// Weight gradients of the jobs that still go through the scratch rows: the feature bank's two small Linears (one 32 x 32 tile
// each: M, N <= 32).  One workgroup of WGM_WAVES waves per (job, slot); MFMA operands are read straight from the scratch rows in
// lane order: lane l supplies act[row k0 + (l>>5)][column l&31] as A[i][k] and dpre[row][column l&31] as B[k][j]; both are 128-B
// contiguous per row.  C[i][j] accumulates dW^T; the waves' tiles are summed through LDS in a fixed order into the slot's partial
// tile.  (Until round 4: one wave per slot, 256 waves per job -- at 200 k anchors each walked 780 rows in 49 dependent rounds of
// loads, 91 us for 100 MB; eight waves per slot: 6 rounds.)
constexpr int WGM_WAVES = 8;
__global__ void __launch_bounds__(WGM_WAVES * 64) wgrad_mfma_kernel(WJobs jobs, const uint32_t* __restrict__ count,
                                                                   const float* __restrict__ rows, float* __restrict__ partial,
                                                                   int row_stride /* floats per anchor row: ROW or BROW */) {
  __shared__ float stage[WGM_WAVES][1024 + 64];
  const WJob job = jobs.j[blockIdx.y];
  if (!job.active || !job.via_rows) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  const uint32_t n = *count;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  float bsum = 0.f;
  const int kx = job.a_x0 + c;
  const int apos = job.a_x0 >= 0 ? 18 * (kx & 1) + (kx >> 1) : c;
  // rows are dealt to the waves of this job in interleaved pairs; WG_UNROLL pairs are loaded before their MFMAs are issued
  for (uint32_t k0 = 2u * (blockIdx.x * WGM_WAVES + wv); k0 < n; k0 += 2u * WG_WAVES * WGM_WAVES * WG_UNROLL) {
    float av[WG_UNROLL], bv[WG_UNROLL];
#pragma unroll
    for (int u = 0; u < WG_UNROLL; u++) {
      const uint32_t k = k0 + 2u * WG_WAVES * WGM_WAVES * u + (uint32_t)half;
      const bool live = k < n;
      const float* row = rows + (size_t)k * row_stride;
      av[u] = (live && c < job.M) ? row[job.a_off + apos] : 0.f;
      bv[u] = (live && c < job.N) ? row[job.b_off + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < WG_UNROLL; u++) {
      bsum += bv[u];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; r++) stage[wv][r * 64 + lane] = acc[r];
  stage[wv][1024 + lane] = bsum;
  __syncthreads();
  float* out = partial + ((size_t)blockIdx.y * WG_WAVES + blockIdx.x) * WG_TILE;
  // partial tile layout: [it][jt][i_local 0..31][j_local 0..31] with it = jt = 0 here; C/D map: col = lane&31,
  // row = (r&3) + 8 (r>>2) + 4 (lane>>5).  Only rows i < M are stored; the bias partial goes to row 31 of tile (1, 0), which
  // no job's weight rows reach.
  for (int e = threadIdx.x; e < 1024; e += WGM_WAVES * 64) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < WGM_WAVES; w++) sum += stage[w][e];
    const int r = e >> 6, l = e & 63;
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    if (i < job.M) out[i * 32 + (l & 31)] = sum;
  }
  if (threadIdx.x < 32) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < WGM_WAVES; w++) sum += stage[w][1024 + threadIdx.x] + stage[w][1024 + 32 + threadIdx.x];   // both lane halves
    out[((1 * 3 + 0) * 32 + 31) * 32 + threadIdx.x] = sum;
  }
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(WJobs jobs, const uint32_t* __restrict__ count, const float* __restrict__ partial,
                                                           float* __restrict__ gsum, float* __restrict__ dparams) {
  // 16 elements x 16 slices of the partial waves per workgroup: every thread has its 16 loads in flight at once (the sum over
  // 256 per-wave partials was a latency-bound chain of load batches: 20 us; the data is only ~10 MB)
  constexpr int SLICES = 16, PER = WG_WAVES / SLICES;
  __shared__ float part[SLICES][16];
  const WJob job = jobs.j[blockIdx.y];
  if (!job.active) return;
  const int nelem = (job.M + 1) * job.N;   // weight entries + one bias row
  if (blockIdx.x * 16 >= nelem) return;
  const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s = 0.f;
  int dst = 0;
  bool live = false;
  if (e < nelem) {
    const int i = e / job.N, j = e - i * job.N;
    int slot;
    if (i < job.M) slot = (((i >> 5) * 3 + (j >> 5)) * 32 + (i & 31)) * 32 + (j & 31);
    else slot = ((1 * 3 + (j >> 5)) * 32 + 31) * 32 + (j & 31);
    int o = j;   // output unit of stored column j
    if (job.b_half > 0) { const int hh = j / job.b_half, r = j - hh * job.b_half; o = r < job.b_real ? job.b_real * hh + r : -1; }
    live = o >= 0;
    dst = i < job.M ? job.w_off + o * job.ldw + i : job.bias_off + o;
    const float* p = partial + ((size_t)blockIdx.y * WG_WAVES + q * PER) * WG_TILE + slot;
    // partial tiles that exist: every wave's for the row-based jobs, one per workgroup of the backward kernel that had a slab
    const int nparts = job.via_rows ? WG_WAVES : (int)min((uint32_t)BWD_GRID, (count[0] + 127u) / 128u);
    float acc[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) acc[u] = (q * PER + u < nparts) ? p[(size_t)u * WG_TILE] : 0.f;
#pragma unroll
    for (int st = PER / 2; st > 0; st >>= 1)
#pragma unroll
      for (int u = 0; u < st; u++) acc[u] += acc[u + st];
    s = acc[0];
  }
  part[q][el] = s;
  __syncthreads();
  if (q == 0 && live) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < SLICES; k++) t += part[k][el];
    gsum[dst] = t;
    dparams[dst] += t;
  }
}

// Appearance embedding: appearance_feat = Wa pose + ba is the same for every anchor and enters the colour MLP's first
// layer linearly, so with g = dL/d(colour hidden pre-activation bias) (already reduced into gsum):
//   dW1k[j][kapp + a] = g[j] app[a];   dapp[a] = sum_j W1k[j][kapp + a] g[j];   dWa[a][q] = dapp[a] pose[q];  dba = dapp.
// One 1024-thread workgroup, one (hidden j, embedding a) product per thread and round: the first version walked the 32
// read-modify-writes of a column in one thread (11.6 us of dependent global round trips).  The regulariser's final
// division rides along when the caller asked for it (reg_out != nullptr), saving reg_finish_kernel's launch.
__global__ void __launch_bounds__(1024) appearance_finish_kernel(Layout L, const float* __restrict__ params,
                                                                 const float* __restrict__ pose7, const float* __restrict__ gsum,
                                                                 float* __restrict__ dparams, const uint32_t* __restrict__ count,
                                                                 float* __restrict__ reg_sum, float reg_w, float* __restrict__ reg_out) {
  __shared__ float app[MAX_APP], g[FD], part[FD][MAX_APP + 1];
  const int tid = threadIdx.x;
  if (tid == 0 && reg_out != nullptr) {
    *reg_out = reg_w * (*reg_sum) / (float)max(count[1], 1u);   // reg_weight * mean(prod(scaling))
    *reg_sum = 0.f;
  }
  if (tid < FD) g[tid] = gsum[L.b1[2] + tid];
  if (tid < L.app) {
    float s = params[L.ab + tid];
    for (int q = 0; q < 7; q++) s += params[L.aw + tid * 7 + q] * pose7[q];
    app[tid] = s;
  }
  __syncthreads();
  for (int idx = tid; idx < FD * L.app; idx += 1024) {
    const int j = idx / L.app, a = idx - j * L.app;
    const int w = L.w1[2] + j * L.in[2] + L.kapp + a;
    part[j][a] = params[w] * g[j];
    dparams[w] += g[j] * app[a];
  }
  __syncthreads();
  if (tid < L.app) {
    float dapp = 0.f;
    for (int j = 0; j < FD; j++) dapp += part[j][tid];      // fixed order: deterministic
    for (int q = 0; q < 7; q++) dparams[L.aw + tid * 7 + q] += dapp * pose7[q];
    dparams[L.ab + tid] += dapp;
  }
}

__global__ void reg_finish_kernel(const uint32_t* count, float* reg_sum, float w, float* out) {
  *out = w * (*reg_sum) / (float)max(count[1], 1u);   // reg_weight * mean(prod(scaling))
  *reg_sum = 0.f;                                      // a second backward on the same forward state starts from zero again
}

WJobs make_jobs(const Layout& L, bool compact_rows /* the pair backward's 512-B rows (BROW) instead of the one-kernel form's */) {
  WJobs J;
  const int nout[3] = {NO, 7 * NO, 3 * NO};
  for (int m = 0; m < 3; m++) {   // accumulated inside neural_bwd_kernel: only the shapes and destinations matter here
    J.j[m] = WJob{0, FD + 3 + L.dist[m], 0, 0, FD, 0, 0, L.w1[m], L.in[m], L.b1[m], 1, 0};
    J.j[3 + m] = WJob{0, FD, -1, 0, nout[m], 0, 0, L.w2[m], FD, L.b2[m], 1, 0};
  }
  if (compact_rows) {
    J.j[6] = WJob{RB_X, 4, FD, RB_DH, FD, 0, 0, L.fw1, 4, L.fb1, L.bank, 1};
    J.j[7] = WJob{RB_H, FD, -1, RB_DF, 3, 0, 0, L.fw2, FD, L.fb2, L.bank, 1};
  } else {
    J.j[6] = WJob{R_X, 4, FD, R_DH + FD * 3, FD, 0, 0, L.fw1, 4, L.fb1, L.bank, 1};
    J.j[7] = WJob{R_H + FD * 3, FD, -1, R_DF, 3, 0, 0, L.fw2, FD, L.fb2, L.bank, 1};
  }
  return J;
}

}  // namespace

namespace { thread_local uint32_t g_neural_flags = 0u; }

extern "C" {

uint32_t segs_neural_set_flags(uint32_t flags) {
  const uint32_t old = g_neural_flags;
  g_neural_flags = flags;
  return old;
}


int segs_neural_param_layout(const segs_neural_dims* dims, int64_t* offsets, int64_t* counts, int* ntensors, int64_t* total) {
  Layout L;
  const int rc = make_layout(dims, &L, offsets, counts, ntensors);
  if (rc == SEGS_OK && total) *total = L.total;
  return rc;
}

size_t segs_neural_temp_bytes(const segs_neural_dims* dims, int A) {
  Layout L;
  if (make_layout(dims, &L, nullptr, nullptr, nullptr) != SEGS_OK || A < 0) return 0;
  return temp_carve(A, L.total, L.bank, nullptr, nullptr);
}

static int neural_forward_impl(const segs_neural_dims* dims, int A, const float* anchor, const float* offset, const float* anchor_feat,
                               const float* scaling_log, const int* visible_radii, const float* mlp_params, const float* camera_center,
                               const float* pose7, float* means3D, float* colors, float* opacity, float* scales, float* rotations,
                               float* neural_opacity, char* temp, void* stream, const segs_projection_targets* tg, const Proj* pj,
                               const float* anchor_rotations = nullptr, int* visible_radii_out = nullptr) {
  hipStream_t st = (hipStream_t)stream;
  Layout L;
  int rc = make_layout(dims, &L, nullptr, nullptr, nullptr);
  if (rc != SEGS_OK) return rc;
  if (A < 0) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  if (A > MAX_ANCHORS) return segs::set_error(SEGS_ERR_UNSUPPORTED, "more than 8 M anchors (the kernels index with 32-bit element offsets)");
  if (A == 0) return SEGS_OK;
  if (!anchor || !offset || !anchor_feat || !scaling_log || !mlp_params || !camera_center || !means3D || (!pj && (!colors || !opacity)) ||
      !scales || !rotations || !neural_opacity || !temp || (L.app > 0 && !pose7))
    return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  Temp T;
  temp_carve(A, L.total, L.bank, temp, &T);
  static_assert(N_IMG_BWD == 262 && sizeof(Small) <= 8192, "temp_carve sizes");
  // T.count: [0] visible anchors, [1] kept candidates; cleared (with the regulariser sum) by pack_tables_kernel
  pack_tables_kernel<<<(N_IMG_BWD * 64 + 255) / 256 + 1, 256, 0, st>>>(L, mlp_params, pose7, T.images, (Small*)T.small, T.count, T.gsum + L.total + 8);
  const bool small_map = A < 131072;
  const int per_wg = small_map ? 256 : 2048;
  const int nb = (A + per_wg - 1) / per_wg;
  Prefilter pf{};
  if (pj && anchor_rotations) {
    pf.anchor = anchor; pf.scaling_log = scaling_log; pf.rot = anchor_rotations; pf.view = pj->view; pf.proj = pj->proj;
    pf.W = pj->W; pf.H = pj->H; pf.tanx = pj->tanx; pf.tany = pj->tany; pf.fx = pj->fx; pf.fy = pj->fy; pf.gx = pj->gx; pf.gy = pj->gy;
  }
  auto launch_compact = [&](auto kernel, int threads) {
    if (pj)
      kernel<<<nb, threads, 0, st>>>(A, visible_radii, T.count, T.vis, nullptr, neural_opacity, tg->radii, tg->tiles_touched, tg->depth_keys,
                                     reinterpret_cast<uint2*>(tg->tile_ranges), tg->num_tiles, pf, visible_radii_out);
    else
      kernel<<<nb, threads, 0, st>>>(A, visible_radii, T.count, T.vis, opacity, neural_opacity, nullptr, nullptr, nullptr, nullptr, 0, pf, nullptr);
  };
  if (small_map) launch_compact(compact_visible_kernel<256, 1>, 256);
  else launch_compact(compact_visible_kernel<1024, 2>, 1024);
  // (a per-device attribute: set on every call -- it is cheap -- so that a process driving several GPUs gets it on each)
  const hipError_t fwd_attr = pj ? hipFuncSetAttribute(reinterpret_cast<const void*>(neural_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWD_LDS_PROJ)
                                 : hipFuncSetAttribute(reinterpret_cast<const void*>(neural_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWD_LDS);
  if (fwd_attr != hipSuccess) return segs::set_hip_error(fwd_attr, __func__);
  if (pj)
    neural_fwd_kernel<true><<<NEURAL_GRID, FWD_WAVES * 64, FWD_LDS_PROJ, st>>>(L, T.count, T.vis, anchor, offset, anchor_feat, scaling_log, T.images, (const Small*)T.small,
                                          camera_center, means3D, nullptr, nullptr, scales, rotations, neural_opacity, T.count + 1, *pj);
  else
    neural_fwd_kernel<false><<<NEURAL_GRID, FWD_WAVES * 64, FWD_LDS, st>>>(L, T.count, T.vis, anchor, offset, anchor_feat, scaling_log, T.images, (const Small*)T.small,
                                          camera_center, means3D, colors, opacity, scales, rotations, neural_opacity, T.count + 1, Proj{});
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

int segs_neural_forward(const segs_neural_dims* dims, int A, const float* anchor, const float* offset, const float* anchor_feat,
                        const float* scaling_log, const int* visible_radii, const float* mlp_params, const float* camera_center,
                        const float* pose7, float* means3D, float* colors, float* opacity, float* scales, float* rotations,
                        float* neural_opacity, char* temp, void* stream) {
  return neural_forward_impl(dims, A, anchor, offset, anchor_feat, scaling_log, visible_radii, mlp_params, camera_center, pose7, means3D,
                             colors, opacity, scales, rotations, neural_opacity, temp, stream, nullptr, nullptr);
}

int segs_neural_forward_projected(const segs_neural_dims* dims, int A, const float* anchor, const float* offset, const float* anchor_feat,
                                  const float* scaling_log, int* visible_radii, const float* anchor_rotations, const float* mlp_params, const float* camera_center,
                                  const float* pose7, float* means3D, float* scales, float* rotations, float* neural_opacity,
                                  const segs_projection_targets* tg, const float* viewmatrix, const float* projmatrix, int width,
                                  int height, float tan_fovx, float tan_fovy, float scale_modifier, char* temp, void* stream) {
  if (!tg || !tg->records || !tg->radii || !tg->tiles_touched || !tg->depth_keys || !tg->tile_ranges || !tg->depth_overflow ||
      !viewmatrix || !projmatrix || width <= 0 || height <= 0 || (anchor_rotations && !visible_radii))
    return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  Proj pj;
  pj.rec = tg->records; pj.radii = tg->radii; pj.touched = tg->tiles_touched; pj.keys = tg->depth_keys; pj.overflow = tg->depth_overflow;
  pj.view = viewmatrix; pj.proj = projmatrix;
  pj.W = width; pj.H = height;
  pj.tanx = tan_fovx; pj.tany = tan_fovy;
  pj.fy = height / (2.0f * tan_fovy);   // rasterizer_impl.cu:221-222
  pj.fx = width / (2.0f * tan_fovx);
  pj.mod = scale_modifier;
  pj.gx = (uint32_t)((width + segs::TILE_X - 1) / segs::TILE_X); pj.gy = (uint32_t)((height + segs::TILE_Y - 1) / segs::TILE_Y);
  pj.flags = tg->flags;
  if ((int)(pj.gx * pj.gy) != tg->num_tiles) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "projection targets were made for another image size");
  return neural_forward_impl(dims, A, anchor, offset, anchor_feat, scaling_log, visible_radii, mlp_params, camera_center, pose7, means3D,
                             nullptr, nullptr, scales, rotations, neural_opacity, temp, stream, tg, &pj, anchor_rotations, visible_radii);
}

int segs_neural_backward(const segs_neural_dims* dims, int A, const float* anchor, const float* offset, const float* anchor_feat,
                         const float* scaling_log, const float* mlp_params, const float* camera_center, const float* pose7,
                         const float* dL_dmeans3D, const float* dL_dcolors, const float* dL_dopacity, const float* dL_dscales,
                         const float* dL_drotations, float* dL_danchor, float* dL_doffset, float* dL_dfeat,
                         float* dL_dscaling_log, float* dL_dmlp_params, float scaling_reg_weight, float* scaling_reg_out, char* temp,
                         void* stream) {
  hipStream_t st = (hipStream_t)stream;
  Layout L;
  int rc = make_layout(dims, &L, nullptr, nullptr, nullptr);
  if (rc != SEGS_OK) return rc;
  if (A < 0) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  if (A > MAX_ANCHORS) return segs::set_error(SEGS_ERR_UNSUPPORTED, "more than 8 M anchors (the kernels index with 32-bit element offsets)");
  if (A == 0) return SEGS_OK;
  if (!anchor || !offset || !anchor_feat || !scaling_log || !mlp_params || !camera_center || !dL_dmeans3D || !dL_dcolors ||
      !dL_dopacity || !dL_dscales || !dL_drotations || !dL_danchor || !dL_doffset || !dL_dfeat || !dL_dscaling_log ||
      !dL_dmlp_params || !temp || (L.app > 0 && !pose7))
    return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  Temp T;
  temp_carve(A, L.total, L.bank, temp, &T);
  constexpr size_t bwd_lds = (N_IMG_BWD * 64 + 4 * WAVE_LDS) * sizeof(float) + sizeof(Small);   // > 64 KB: needs the opt-in below
  constexpr size_t pair_run = (N_IMG_BWD * 64 + 4 * PAIR_LDS) * sizeof(float) + sizeof(Small), pair_end = (size_t)4 * PAIR_STAGE * sizeof(float);
  constexpr size_t pair_lds = pair_run > pair_end ? pair_run : pair_end;
  static_assert(bwd_lds <= 160 * 1024 && pair_lds <= 160 * 1024, "one workgroup per CU");
  auto allow_lds = [](const void* kernel, size_t bytes) { return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); };
  const hipError_t attr_rc[4] = {allow_lds(reinterpret_cast<const void*>(neural_bwd_kernel<false>), bwd_lds),
                                 allow_lds(reinterpret_cast<const void*>(neural_bwd_kernel<true>), bwd_lds),
                                 allow_lds(reinterpret_cast<const void*>(neural_bwd_pair_kernel<false>), pair_lds),
                                 allow_lds(reinterpret_cast<const void*>(neural_bwd_pair_kernel<true>), pair_lds)};
  for (const hipError_t rc_attr : attr_rc)
    if (rc_attr != hipSuccess) return segs::set_hip_error(rc_attr, __func__);
  // the regulariser sum was cleared by the forward (pack_tables_kernel) and is cleared again by reg_finish_kernel; it is
  // only accumulated when somebody reads it
  float* reg_sum = scaling_reg_out ? T.gsum + L.total + 8 : nullptr;
  // Chain waves + weight-gradient waves (neural_bwd_pair_kernel; the feature-bank model too since round 4, its epilogue on
  // compact rows).  SEGS_NEURAL_ONE_KERNEL_BACKWARD (segs_neural_set_flags, segs_neural.h) selects the one-kernel form: the
  // A/B of profiles/ and tests/test_neural_gpu.py.
  const bool one_role = (g_neural_flags & SEGS_NEURAL_ONE_KERNEL_BACKWARD) != 0u;
  if (!one_role)
    (L.bank ? neural_bwd_pair_kernel<true> : neural_bwd_pair_kernel<false>)<<<BWD_GRID, 512, pair_lds, st>>>(
        L, T.count, T.vis, anchor, offset, anchor_feat, scaling_log, T.images, (const Small*)T.small, camera_center, dL_dmeans3D,
        dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, dL_danchor, dL_doffset, dL_dfeat, dL_dscaling_log, T.rows, T.partial,
        scaling_reg_weight, reg_sum);
  else
    (L.bank ? neural_bwd_kernel<true> : neural_bwd_kernel<false>)<<<BWD_GRID, 256, bwd_lds, st>>>(
        L, T.count, T.vis, anchor, offset, anchor_feat, scaling_log, T.images, (const Small*)T.small, camera_center, dL_dmeans3D,
        dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, dL_danchor, dL_doffset, dL_dfeat, dL_dscaling_log, T.rows, T.partial,
        scaling_reg_weight, reg_sum);
  if (scaling_reg_out && L.app == 0) reg_finish_kernel<<<1, 1, 0, st>>>(T.count, reg_sum, scaling_reg_weight, scaling_reg_out);
  const WJobs J = make_jobs(L, !one_role);
  if (L.bank) wgrad_mfma_kernel<<<dim3(WG_WAVES, WG_JOBS), WGM_WAVES * 64, 0, st>>>(J, T.count, T.rows, T.partial, one_role ? ROW : BROW);   // the feature bank's two small Linears
  wgrad_reduce_kernel<<<dim3((37 * 72 + 15) / 16, WG_JOBS), 256, 0, st>>>(J, T.count, T.partial, T.gsum, dL_dmlp_params);
  if (L.app > 0) appearance_finish_kernel<<<1, 1024, 0, st>>>(L, mlp_params, pose7, T.gsum, dL_dmlp_params, T.count, reg_sum,
                                                              scaling_reg_weight, scaling_reg_out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

}  // extern "C"
