// neural.hip -- fused neural-Gaussian generation (Scaffold-GS anchors -> Gaussians), forward and backward
// (include/segs_neural.h).  Reference: GaussianRenderer::generate_neural_gaussians, src/gaussian_renderer.cpp:214-334,
// MLP stacks src/gaussian_model.cpp:61-98.
//
//   compact_visible_kernel : radii>0 -> visible-anchor list + device-side count (no host sync); clears the opacity of
//                            the slots of invisible anchors so the rasterizer skips them
//   neural_fwd_kernel      : thread = visible anchor; view direction, feature bank, three 35->32->{10,70,30} MLPs,
//                            mask, xyz/scale/rot assembly, written straight into the rasterizer's input arrays
//   neural_bwd_kernel      : thread = visible anchor; recomputes the forward (cheaper than saving ~100 floats/anchor),
//                            back-propagates the candidate-domain gradients to anchor/offset/feature/scaling and leaves
//                            the per-anchor (activation, pre-activation gradient) rows in scratch
//   wgrad_mfma_kernel      : dW[j][i] = sum_anchors dpre[a][j] * act[a][i] for all eight Linear layers with
//                            v_mfma_f32_32x32x2_f32 (exact fp32), operands loaded straight from the scratch rows in
//                            the MFMA lane order (lane = column, two anchors per instruction); fixed wave count,
//                            per-wave partial tiles
//   wgrad_reduce_kernel    : deterministic sum of the partial tiles, += into the flat gradient block
//   appearance_finish_kernel: the appearance embedding Linear(7->app) feeds every anchor the same vector, so it is
//                            folded into the colour MLP's first bias; its gradients follow from that bias gradient.
// MLP weights (30 KB) sit in LDS and are read as wave-uniform broadcasts; everything per anchor lives in registers.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include "../../include/segs_neural.h"
#include "../../include/segs_raster.h"

namespace {

constexpr int FD = 32;     // feat_dim
constexpr int NO = 10;     // n_offsets
constexpr int XD = 36;     // MLP input: feat(32) | ob_view(3) | ob_dist(1)
constexpr int ROW = 512;   // scratch floats per visible anchor
constexpr int R_X = 0;     // x[36]
constexpr int R_H = 64;    // + 32*m : hidden activations, m = 0 opacity, 1 cov, 2 colour, 3 feature bank
constexpr int R_DH = 192;  // + 32*m : dL/d(hidden pre-activation)
constexpr int R_DO = 320;  // dL/d(opacity MLP output pre-tanh) [10]
constexpr int R_DC = 352;  // dL/d(cov MLP output) [70]
constexpr int R_DK = 448;  // dL/d(colour MLP output pre-sigmoid) [30]
constexpr int R_DF = 480;  // dL/d(feature-bank logits) [3]
constexpr int WG_WAVES = 256;  // waves per weight-gradient job
constexpr int WG_UNROLL = 8;   // row pairs whose operand loads are in flight together
constexpr int WG_JOBS = 8;
constexpr int WG_TILE = 2 * 3 * 1024;  // floats of partial sums per (job, wave): up to 2x3 tiles of 32x32
constexpr int MAX_APP = 64;

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
    return SEGS_ERR_UNSUPPORTED;
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
};
size_t temp_carve(int A, int total, char* base, Temp* t) {
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
  const size_t o_count = take(256), o_vis = take((size_t)A * 4), o_rows = take((size_t)A * ROW * 4),
               o_part = take((size_t)WG_JOBS * WG_WAVES * WG_TILE * 4), o_gsum = take((size_t)(total + 64) * 4);
  if (t) {
    t->count = (uint32_t*)(base + o_count); t->vis = (uint32_t*)(base + o_vis); t->rows = (float*)(base + o_rows);
    t->partial = (float*)(base + o_part); t->gsum = (float*)(base + o_gsum);
  }
  return off;
}

constexpr int PAD_O = 12, PAD_C = 72, PAD_K = 32;   // second-layer widths (10, 70, 30) padded to float4
struct Lds {
  float w1[3][FD][XD];      // [m][j][i]; column 35 is zero when the MLP does not see ob_dist
  float b1[3][FD];          // colour: includes W1k[:, appearance columns] . appearance_feat
  float w2o[FD][PAD_O], w2c[FD][PAD_C], w2k[FD][PAD_K];   // second layers TRANSPOSED: [hidden j][output o]
  float b2o[PAD_O], b2c[PAD_C], b2k[PAD_K];
  float fw1[FD][4], fb1[FD], fw2[FD][4], fb2[4];          // feature bank; fw2 transposed [j][c]
  float app[MAX_APP];
};

__device__ void stage_weights(Lds& S, const Layout& L, const float* __restrict__ P, const float* __restrict__ pose7) {
  const int tid = threadIdx.x, nt = blockDim.x;
  if (L.app > 0) {
    for (int a = tid; a < L.app; a += nt) {
      float s = P[L.ab + a];
      for (int q = 0; q < 7; q++) s += P[L.aw + a * 7 + q] * pose7[q];
      S.app[a] = s;
    }
    __syncthreads();
  }
  for (int e = tid; e < 3 * FD * XD; e += nt) {
    const int m = e / (FD * XD), r = e - m * FD * XD, j = r / XD, i = r - j * XD;
    const int ncol = FD + 3 + L.dist[m];
    (&S.w1[0][0][0])[e] = i < ncol ? P[L.w1[m] + j * L.in[m] + i] : 0.f;
  }
  for (int e = tid; e < 3 * FD; e += nt) {
    const int m = e / FD, j = e - m * FD;
    float b = P[L.b1[m] + j];
    if (m == 2)
      for (int a = 0; a < L.app; a++) b += P[L.w1[2] + j * L.in[2] + L.kapp + a] * S.app[a];
    S.b1[m][j] = b;
  }
  for (int e = tid; e < FD * PAD_O; e += nt) { const int j = e / PAD_O, o = e - j * PAD_O; S.w2o[j][o] = o < NO ? P[L.w2[0] + o * FD + j] : 0.f; }
  for (int e = tid; e < FD * PAD_C; e += nt) { const int j = e / PAD_C, o = e - j * PAD_C; S.w2c[j][o] = o < 7 * NO ? P[L.w2[1] + o * FD + j] : 0.f; }
  for (int e = tid; e < FD * PAD_K; e += nt) { const int j = e / PAD_K, o = e - j * PAD_K; S.w2k[j][o] = o < 3 * NO ? P[L.w2[2] + o * FD + j] : 0.f; }
  for (int e = tid; e < PAD_O; e += nt) S.b2o[e] = e < NO ? P[L.b2[0] + e] : 0.f;
  for (int e = tid; e < PAD_C; e += nt) S.b2c[e] = e < 7 * NO ? P[L.b2[1] + e] : 0.f;
  for (int e = tid; e < PAD_K; e += nt) S.b2k[e] = e < 3 * NO ? P[L.b2[2] + e] : 0.f;
  if (L.bank) {
    for (int e = tid; e < FD * 4; e += nt) (&S.fw1[0][0])[e] = P[L.fw1 + e];
    for (int e = tid; e < FD; e += nt) S.fb1[e] = P[L.fb1 + e];
    for (int e = tid; e < FD * 4; e += nt) { const int j = e >> 2, c = e & 3; S.fw2[j][c] = c < 3 ? P[L.fw2 + c * FD + j] : 0.f; }
    for (int e = tid; e < 4; e += nt) S.fb2[e] = e < 3 ? P[L.fb2 + e] : 0.f;
  }
  __syncthreads();
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

// Two-layer MLP, hidden loop kept rolled: per hidden unit j one LDS row of each layer is read (wave-uniform broadcast);
// x / out / dout / dx are register arrays with static indices only.
template <int NOUT, int PAD>
__device__ __forceinline__ void mlp_forward(const float (*w1)[XD], const float* b1, const float (*w2t)[PAD], const float* b2,
                                            const float* x, float* out) {
#pragma unroll
  for (int o = 0; o < NOUT; o++) out[o] = b2[o];
#pragma unroll 1
  for (int j = 0; j < FD; j++) {
    float s = b1[j];
#pragma unroll
    for (int i = 0; i < XD; i++) s += w1[j][i] * x[i];
    const float hj = fmaxf(s, 0.f);
#pragma unroll
    for (int o = 0; o < NOUT; o++) out[o] += w2t[j][o] * hj;
  }
}
// Backward through the same MLP given dL/d(output pre-activation): accumulates dL/dx, stores the hidden activations
// and the hidden pre-activation gradients (operands of the weight-gradient MFMAs) into the anchor's scratch row.
template <int NOUT, int PAD>
__device__ __forceinline__ void mlp_backward(const float (*w1)[XD], const float* b1, const float (*w2t)[PAD], const float* x,
                                             const float* dout, float* dx, float* __restrict__ row_h, float* __restrict__ row_dh) {
#pragma unroll 1
  for (int j = 0; j < FD; j++) {
    float s = b1[j];
#pragma unroll
    for (int i = 0; i < XD; i++) s += w1[j][i] * x[i];
    float dh = 0.f;
#pragma unroll
    for (int o = 0; o < NOUT; o++) dh += w2t[j][o] * dout[o];
    const float d = s > 0.f ? dh : 0.f;
    row_h[j] = fmaxf(s, 0.f);
    row_dh[j] = d;
#pragma unroll
    for (int i = 0; i < XD; i++) dx[i] += w1[j][i] * d;
  }
}

// Per-anchor inputs shared by forward and backward: view direction / distance, feature bank, MLP input vector x.
struct AnchorState {
  float x[XD];            // feat' (bank-blended) | view | dist
  float anc[3], gs[6];    // anchor, exp(scaling_log)
  float bw[3];            // feature-bank softmax weights
  float inv_dist;
};

__device__ __forceinline__ void load_feat(const float* __restrict__ anchor_feat, uint32_t a, float* f) {
  const float4* p = reinterpret_cast<const float4*>(anchor_feat + (size_t)a * FD);
#pragma unroll
  for (int q = 0; q < FD / 4; q++) {
    const float4 v = p[q];
    f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
  }
}

__device__ __forceinline__ void anchor_state(const Lds& S, const Layout& L, uint32_t a, const float* __restrict__ anchor,
                                             const float* __restrict__ anchor_feat, const float* __restrict__ scaling_log,
                                             const float* __restrict__ campos, AnchorState& st) {
  float feat[FD];
  load_feat(anchor_feat, a, feat);
#pragma unroll
  for (int c = 0; c < 3; c++) st.anc[c] = anchor[(size_t)a * 3 + c];
#pragma unroll
  for (int c = 0; c < 6; c++) st.gs[c] = expf(scaling_log[(size_t)a * 6 + c]);
  const float ox = st.anc[0] - campos[0], oy = st.anc[1] - campos[1], oz = st.anc[2] - campos[2];
  const float dist = sqrtf(ox * ox + oy * oy + oz * oz);
  st.inv_dist = 1.0f / dist;
  st.x[FD] = ox / dist; st.x[FD + 1] = oy / dist; st.x[FD + 2] = oz / dist; st.x[FD + 3] = dist;
  if (L.bank) {
    float lg[3] = {S.fb2[0], S.fb2[1], S.fb2[2]};
#pragma unroll 1
    for (int j = 0; j < FD; j++) {
      float s = S.fb1[j];
#pragma unroll
      for (int i = 0; i < 4; i++) s += S.fw1[j][i] * st.x[FD + i];
      const float hj = fmaxf(s, 0.f);
#pragma unroll
      for (int c = 0; c < 3; c++) lg[c] += S.fw2[j][c] * hj;
    }
    const float mx = fmaxf(lg[0], fmaxf(lg[1], lg[2]));
    const float e0 = expf(lg[0] - mx), e1 = expf(lg[1] - mx), e2 = expf(lg[2] - mx);
    const float inv = 1.0f / (e0 + e1 + e2);
    st.bw[0] = e0 * inv; st.bw[1] = e1 * inv; st.bw[2] = e2 * inv;
    // feat'[j] = feat[4 (j%8)] bw0 + feat[2 (j%16)] bw1 + feat[j] bw2   (gaussian_renderer.cpp:242-247)
#pragma unroll
    for (int j = 0; j < FD; j++) st.x[j] = feat[4 * (j % 8)] * st.bw[0] + feat[2 * (j % 16)] * st.bw[1] + feat[j] * st.bw[2];
  } else {
    st.bw[0] = st.bw[1] = st.bw[2] = 0.f;
#pragma unroll
    for (int j = 0; j < FD; j++) st.x[j] = feat[j];
  }
}

__global__ void __launch_bounds__(256) compact_visible_kernel(int A, const int* __restrict__ radii, uint32_t* __restrict__ count,
                                                              uint32_t* __restrict__ vis, float* __restrict__ opacity,
                                                              float* __restrict__ neural_opacity) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  const bool v = a < A && (radii == nullptr || radii[a] > 0);
  const uint64_t m = __ballot(v);
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == 0 && m) base = atomicAdd(count, (uint32_t)__popcll(m));
  base = __shfl(base, 0, 64);
  if (v) vis[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)a;
  if (a < A && !v) {
#pragma unroll
    for (int k = 0; k < NO; k++) { opacity[(size_t)a * NO + k] = 0.f; neural_opacity[(size_t)a * NO + k] = 0.f; }
  }
}

__global__ void __launch_bounds__(256) neural_fwd_kernel(
    Layout L, const uint32_t* __restrict__ count, const uint32_t* __restrict__ vis, const float* __restrict__ anchor,
    const float* __restrict__ offset, const float* __restrict__ anchor_feat, const float* __restrict__ scaling_log,
    const float* __restrict__ params, const float* __restrict__ campos, const float* __restrict__ pose7,
    float* __restrict__ means3D, float* __restrict__ colors, float* __restrict__ opacity, float* __restrict__ scales,
    float* __restrict__ rotations, float* __restrict__ neural_opacity) {
  __shared__ Lds S;
  const uint32_t n = *count;
  if (blockIdx.x * 256u >= n) return;
  stage_weights(S, L, params, pose7);
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= n) return;
  const uint32_t a = vis[t];
  AnchorState st;
  anchor_state(S, L, a, anchor, anchor_feat, scaling_log, campos, st);
  const size_t c0 = (size_t)a * NO;
  {
    float out[NO];
    mlp_forward<NO, PAD_O>(S.w1[0], S.b1[0], S.w2o, S.b2o, st.x, out);
#pragma unroll
    for (int k = 0; k < NO; k++) {
      const float op = tanhf(out[k]);
      neural_opacity[c0 + k] = op;
      opacity[c0 + k] = op;
    }
  }
  {
    float out[3 * NO];
    mlp_forward<3 * NO, PAD_K>(S.w1[2], S.b1[2], S.w2k, S.b2k, st.x, out);
#pragma unroll
    for (int e = 0; e < 3 * NO; e++) colors[c0 * 3 + e] = sigmoidf(out[e]);
  }
  {
    float out[7 * NO];
    mlp_forward<7 * NO, PAD_C>(S.w1[1], S.b1[1], S.w2c, S.b2c, st.x, out);
#pragma unroll
    for (int k = 0; k < NO; k++) {
      const float* sr = out + 7 * k;
#pragma unroll
      for (int c = 0; c < 3; c++) scales[(c0 + k) * 3 + c] = st.gs[3 + c] * sigmoidf(sr[c]);   // :327-328
      const float nrm = fmaxf(sqrtf(sr[3] * sr[3] + sr[4] * sr[4] + sr[5] * sr[5] + sr[6] * sr[6]), 1e-12f);  // F::normalize
      reinterpret_cast<float4*>(rotations)[c0 + k] = make_float4(sr[3] / nrm, sr[4] / nrm, sr[5] / nrm, sr[6] / nrm);
#pragma unroll
      for (int c = 0; c < 3; c++) means3D[(c0 + k) * 3 + c] = st.anc[c] + offset[(c0 + k) * 3 + c] * st.gs[c];  // :331-332
    }
  }
}

__global__ void __launch_bounds__(256) neural_bwd_kernel(
    Layout L, const uint32_t* __restrict__ count, const uint32_t* __restrict__ vis, const float* __restrict__ anchor,
    const float* __restrict__ offset, const float* __restrict__ anchor_feat, const float* __restrict__ scaling_log,
    const float* __restrict__ params, const float* __restrict__ campos, const float* __restrict__ pose7,
    const float* __restrict__ g_means, const float* __restrict__ g_colors, const float* __restrict__ g_opacity,
    const float* __restrict__ g_scales, const float* __restrict__ g_rot, float* __restrict__ d_anchor,
    float* __restrict__ d_offset, float* __restrict__ d_feat, float* __restrict__ d_scaling_log, float* __restrict__ rows) {
  __shared__ Lds S;
  const uint32_t n = *count;
  if (blockIdx.x * 256u >= n) return;
  stage_weights(S, L, params, pose7);
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= n) return;
  const uint32_t a = vis[t];
  float* row = rows + (size_t)t * ROW;
  AnchorState st;
  anchor_state(S, L, a, anchor, anchor_feat, scaling_log, campos, st);
#pragma unroll
  for (int i = 0; i < XD; i++) row[R_X + i] = st.x[i];
  const size_t c0 = (size_t)a * NO;
  float dx[XD];
#pragma unroll
  for (int i = 0; i < XD; i++) dx[i] = 0.f;
  uint32_t keep = 0;  // bit k: candidate k has neural opacity > 0 (the reference's mask, :279)

  // ---- opacity MLP
  {
    float out[NO];
    mlp_forward<NO, PAD_O>(S.w1[0], S.b1[0], S.w2o, S.b2o, st.x, out);
#pragma unroll
    for (int k = 0; k < NO; k++) {
      const float op = tanhf(out[k]);
      float dpre = 0.f;
      if (op > 0.f) { keep |= 1u << k; dpre = g_opacity[c0 + k] * (1.f - op * op); }
      out[k] = dpre;
      row[R_DO + k] = dpre;
    }
    mlp_backward<NO, PAD_O>(S.w1[0], S.b1[0], S.w2o, st.x, out, dx, row + R_H, row + R_DH);
  }
  // ---- colour MLP
  {
    float out[3 * NO];
    mlp_forward<3 * NO, PAD_K>(S.w1[2], S.b1[2], S.w2k, S.b2k, st.x, out);
#pragma unroll
    for (int e = 0; e < 3 * NO; e++) {
      const float col = sigmoidf(out[e]);
      const float dpre = ((keep >> (e / 3)) & 1u) ? g_colors[c0 * 3 + e] * col * (1.f - col) : 0.f;
      out[e] = dpre;
      row[R_DK + e] = dpre;
    }
    mlp_backward<3 * NO, PAD_K>(S.w1[2], S.b1[2], S.w2k, st.x, out, dx, row + R_H + 2 * FD, row + R_DH + 2 * FD);
  }
  // ---- covariance MLP + per-candidate assembly
  float dgs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, danc[3] = {0.f, 0.f, 0.f};
  {
    float out[7 * NO];
    mlp_forward<7 * NO, PAD_C>(S.w1[1], S.b1[1], S.w2c, S.b2c, st.x, out);
#pragma unroll
    for (int k = 0; k < NO; k++) {
      const bool on = (keep >> k) & 1u;
      float* sr = out + 7 * k;
      float dsr[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float doff[3] = {0.f, 0.f, 0.f};
      if (on) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float sg = sigmoidf(sr[c]);
          const float gsc = g_scales[(c0 + k) * 3 + c];
          dgs[3 + c] += gsc * sg;
          dsr[c] = gsc * st.gs[3 + c] * sg * (1.f - sg);
          const float gm = g_means[(c0 + k) * 3 + c];
          danc[c] += gm;
          doff[c] = gm * st.gs[c];
          dgs[c] += gm * offset[(c0 + k) * 3 + c];
        }
        const float4 gr = reinterpret_cast<const float4*>(g_rot)[c0 + k];
        const float nr = sqrtf(sr[3] * sr[3] + sr[4] * sr[4] + sr[5] * sr[5] + sr[6] * sr[6]);
        if (nr >= 1e-12f) {   // r = v / |v|:  dv = (g - r (r.g)) / |v|
          const float inv = 1.0f / nr;
          const float r0 = sr[3] * inv, r1 = sr[4] * inv, r2 = sr[5] * inv, r3 = sr[6] * inv;
          const float dot = r0 * gr.x + r1 * gr.y + r2 * gr.z + r3 * gr.w;
          dsr[3] = (gr.x - r0 * dot) * inv; dsr[4] = (gr.y - r1 * dot) * inv;
          dsr[5] = (gr.z - r2 * dot) * inv; dsr[6] = (gr.w - r3 * dot) * inv;
        } else {              // clamp_min(|v|, eps) active: r = v / eps
          dsr[3] = gr.x * 1e12f; dsr[4] = gr.y * 1e12f; dsr[5] = gr.z * 1e12f; dsr[6] = gr.w * 1e12f;
        }
      }
#pragma unroll
      for (int c = 0; c < 3; c++) d_offset[(c0 + k) * 3 + c] += doff[c];
#pragma unroll
      for (int q = 0; q < 7; q++) { sr[q] = dsr[q]; row[R_DC + 7 * k + q] = dsr[q]; }
    }
    mlp_backward<7 * NO, PAD_C>(S.w1[1], S.b1[1], S.w2c, st.x, out, dx, row + R_H + FD, row + R_DH + FD);
  }
#pragma unroll
  for (int c = 0; c < 6; c++) d_scaling_log[(size_t)a * 6 + c] += dgs[c] * st.gs[c];   // through exp()

  // ---- input vector: feature bank, view direction, distance
  float dview[4] = {dx[FD], dx[FD + 1], dx[FD + 2], dx[FD + 3]};   // the dist column of w1 is zero when unused
  float4* dfo = reinterpret_cast<float4*>(d_feat + (size_t)a * FD);
  if (L.bank) {
    float feat[FD];
    load_feat(anchor_feat, a, feat);
    float dbw[3] = {0.f, 0.f, 0.f};
    float df[FD];
#pragma unroll
    for (int j = 0; j < FD; j++) {
      dbw[0] += dx[j] * feat[4 * (j % 8)];
      dbw[1] += dx[j] * feat[2 * (j % 16)];
      dbw[2] += dx[j] * feat[j];
      df[j] = st.bw[2] * dx[j];
    }
#pragma unroll
    for (int j = 0; j < FD; j++) {
      df[4 * (j % 8)] += st.bw[0] * dx[j];
      df[2 * (j % 16)] += st.bw[1] * dx[j];
    }
#pragma unroll
    for (int q = 0; q < FD / 4; q++) {
      float4 v = dfo[q];
      v.x += df[4 * q]; v.y += df[4 * q + 1]; v.z += df[4 * q + 2]; v.w += df[4 * q + 3];
      dfo[q] = v;
    }
    const float sdot = st.bw[0] * dbw[0] + st.bw[1] * dbw[1] + st.bw[2] * dbw[2];
    float dlg[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { dlg[c] = st.bw[c] * (dbw[c] - sdot); row[R_DF + c] = dlg[c]; }
#pragma unroll 1
    for (int j = 0; j < FD; j++) {
      float s = S.fb1[j];
#pragma unroll
      for (int i = 0; i < 4; i++) s += S.fw1[j][i] * st.x[FD + i];
      float d = S.fw2[j][0] * dlg[0] + S.fw2[j][1] * dlg[1] + S.fw2[j][2] * dlg[2];
      d = s > 0.f ? d : 0.f;
      row[R_H + 3 * FD + j] = fmaxf(s, 0.f);
      row[R_DH + 3 * FD + j] = d;
#pragma unroll
      for (int i = 0; i < 4; i++) dview[i] += S.fw1[j][i] * d;
    }
  } else {
#pragma unroll
    for (int q = 0; q < FD / 4; q++) {
      float4 v = dfo[q];
      v.x += dx[4 * q]; v.y += dx[4 * q + 1]; v.z += dx[4 * q + 2]; v.w += dx[4 * q + 3];
      dfo[q] = v;
    }
  }
  // view = ob / |ob|, dist = |ob|:  d ob = (dview - view (view . dview)) / dist + ddist * view
  {
    const float vx = st.x[FD], vy = st.x[FD + 1], vz = st.x[FD + 2];
    const float dot = vx * dview[0] + vy * dview[1] + vz * dview[2];
    danc[0] += (dview[0] - vx * dot) * st.inv_dist + dview[3] * vx;
    danc[1] += (dview[1] - vy * dot) * st.inv_dist + dview[3] * vy;
    danc[2] += (dview[2] - vz * dot) * st.inv_dist + dview[3] * vz;
  }
#pragma unroll
  for (int c = 0; c < 3; c++) d_anchor[(size_t)a * 3 + c] += danc[c];
}

// ---- weight gradients ---------------------------------------------------------------------------------------------
struct WJob {
  int a_off, M;       // activation columns [a_off, a_off+M) of the scratch row (the Linear's input)
  int b_off, N;       // pre-activation gradient columns (the Linear's output)
  int w_off, ldw;     // dW[j][i] -> gsum[w_off + j*ldw + i]
  int bias_off;       // db[j]   -> gsum[bias_off + j]
  int active;
};
struct WJobs { WJob j[WG_JOBS]; };

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One wave per (job, slice).  MFMA operands are read straight from the scratch rows in lane order: lane l supplies
// act[row k0 + (l>>5)][column l&31] as A[i][k] and dpre[row][column l&31] as B[k][j]; both are 128-B contiguous per
// row.  C[i][j] accumulates dW^T.
__global__ void __launch_bounds__(64) wgrad_mfma_kernel(WJobs jobs, const uint32_t* __restrict__ count,
                                                        const float* __restrict__ rows, float* __restrict__ partial) {
  const WJob job = jobs.j[blockIdx.y];
  if (!job.active) return;
  const int lane = threadIdx.x;
  const int c = lane & 31, half = lane >> 5;
  const uint32_t n = *count;
  const int nit = (job.M + 31) / 32, njt = (job.N + 31) / 32;   // <= 2, <= 3
  f32x16 acc[2][3];
#pragma unroll
  for (int it = 0; it < 2; it++)
#pragma unroll
    for (int jt = 0; jt < 3; jt++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[it][jt][r] = 0.f;
  float bsum[3] = {0.f, 0.f, 0.f};
  // rows are dealt to the WG_WAVES waves of this job in interleaved pairs; WG_UNROLL pairs are loaded before their
  // MFMAs are issued (the loop is load-latency bound otherwise: one dependent L2 round trip per row pair)
  for (uint32_t k0 = 2u * blockIdx.x; k0 < n; k0 += 2u * WG_WAVES * WG_UNROLL) {
    float av[WG_UNROLL][2], bv[WG_UNROLL][3];
#pragma unroll
    for (int u = 0; u < WG_UNROLL; u++) {
      const uint32_t k = k0 + 2u * WG_WAVES * u + (uint32_t)half;
      const bool live = k < n;
      const float* row = rows + (size_t)k * ROW;
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const int i = it * 32 + c;
        av[u][it] = (live && i < job.M) ? row[job.a_off + i] : 0.f;
      }
#pragma unroll
      for (int jt = 0; jt < 3; jt++) {
        const int j = jt * 32 + c;
        bv[u][jt] = (live && j < job.N) ? row[job.b_off + j] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < WG_UNROLL; u++) {
#pragma unroll
      for (int jt = 0; jt < 3; jt++) bsum[jt] += bv[u][jt];
#pragma unroll
      for (int it = 0; it < 2; it++)
#pragma unroll
        for (int jt = 0; jt < 3; jt++)
          if (it < nit && jt < njt)
            acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][it], bv[u][jt], acc[it][jt], 0, 0, 0);
    }
  }
  float* out = partial + ((size_t)blockIdx.y * WG_WAVES + blockIdx.x) * WG_TILE;
  // partial tile layout: [it][jt][i_local 0..31][j_local 0..31]; C/D map: col = lane&31, row = (r&3) + 8 (r>>2) + 4 (lane>>5).
  // Only rows i < M are stored; the bias partial of column tile jt goes to row 31 of tile (1, jt), which no job's
  // weight rows reach (two-i-tile jobs have M <= 36, i.e. local rows 0..3 of tile (1, *)).
#pragma unroll
  for (int it = 0; it < 2; it++)
#pragma unroll
    for (int jt = 0; jt < 3; jt++)
      if (it < nit && jt < njt) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
          if (it * 32 + i < job.M) out[((it * 3 + jt) * 32 + i) * 32 + c] = acc[it][jt][r];
        }
      }
#pragma unroll
  for (int jt = 0; jt < 3; jt++) {
    const float other = __shfl_xor(bsum[jt], 32, 64);   // the two lane halves summed different rows of one column
    if (half == 0 && jt < njt) out[((1 * 3 + jt) * 32 + 31) * 32 + c] = bsum[jt] + other;
  }
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(WJobs jobs, const float* __restrict__ partial, float* __restrict__ gsum,
                                                           float* __restrict__ dparams) {
  const WJob job = jobs.j[blockIdx.y];
  if (!job.active) return;
  const int nelem = (job.M + 1) * job.N;   // weight entries + one bias row
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nelem) return;
  const int i = e / job.N, j = e - i * job.N;
  int slot;
  if (i < job.M) slot = (((i >> 5) * 3 + (j >> 5)) * 32 + (i & 31)) * 32 + (j & 31);
  else slot = ((1 * 3 + (j >> 5)) * 32 + 31) * 32 + (j & 31);
  const float* p = partial + (size_t)blockIdx.y * WG_WAVES * WG_TILE + slot;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int w = 0; w < WG_WAVES; w += 4) {
    s0 += p[(size_t)w * WG_TILE]; s1 += p[(size_t)(w + 1) * WG_TILE];
    s2 += p[(size_t)(w + 2) * WG_TILE]; s3 += p[(size_t)(w + 3) * WG_TILE];
  }
  const float s = (s0 + s1) + (s2 + s3);
  const int dst = i < job.M ? job.w_off + j * job.ldw + i : job.bias_off + j;
  gsum[dst] = s;
  dparams[dst] += s;
}

// Appearance embedding: appearance_feat = Wa pose + ba is the same for every anchor and enters the colour MLP's first
// layer linearly, so with g = dL/d(colour hidden pre-activation bias) (already reduced into gsum):
//   dW1k[j][kapp + a] = g[j] app[a];   dapp[a] = sum_j W1k[j][kapp + a] g[j];   dWa[a][q] = dapp[a] pose[q];  dba = dapp.
__global__ void __launch_bounds__(64) appearance_finish_kernel(Layout L, const float* __restrict__ params,
                                                               const float* __restrict__ pose7, const float* __restrict__ gsum,
                                                               float* __restrict__ dparams) {
  __shared__ float app[MAX_APP], g[FD];
  const int a = threadIdx.x;
  if (a < FD) g[a] = gsum[L.b1[2] + a];
  if (a < L.app) {
    float s = params[L.ab + a];
    for (int q = 0; q < 7; q++) s += params[L.aw + a * 7 + q] * pose7[q];
    app[a] = s;
  }
  __syncthreads();
  if (a < L.app) {
    float dapp = 0.f;
    for (int j = 0; j < FD; j++) {
      dapp += params[L.w1[2] + j * L.in[2] + L.kapp + a] * g[j];
      dparams[L.w1[2] + j * L.in[2] + L.kapp + a] += g[j] * app[a];
    }
    for (int q = 0; q < 7; q++) dparams[L.aw + a * 7 + q] += dapp * pose7[q];
    dparams[L.ab + a] += dapp;
  }
}

WJobs make_jobs(const Layout& L) {
  WJobs J;
  const int nout[3] = {NO, 7 * NO, 3 * NO};
  const int dout_off[3] = {R_DO, R_DC, R_DK};
  for (int m = 0; m < 3; m++) {
    J.j[m] = WJob{R_X, FD + 3 + L.dist[m], R_DH + FD * m, FD, L.w1[m], L.in[m], L.b1[m], 1};
    J.j[3 + m] = WJob{R_H + FD * m, FD, dout_off[m], nout[m], L.w2[m], FD, L.b2[m], 1};
  }
  J.j[6] = WJob{R_X + FD, 4, R_DH + FD * 3, FD, L.fw1, 4, L.fb1, L.bank};
  J.j[7] = WJob{R_H + FD * 3, FD, R_DF, 3, L.fw2, FD, L.fb2, L.bank};
  return J;
}

}  // namespace

extern "C" {

int segs_neural_param_layout(const segs_neural_dims* dims, int64_t* offsets, int64_t* counts, int* ntensors, int64_t* total) {
  Layout L;
  const int rc = make_layout(dims, &L, offsets, counts, ntensors);
  if (rc == SEGS_OK && total) *total = L.total;
  return rc;
}

size_t segs_neural_temp_bytes(const segs_neural_dims* dims, int A) {
  Layout L;
  if (make_layout(dims, &L, nullptr, nullptr, nullptr) != SEGS_OK || A < 0) return 0;
  return temp_carve(A, L.total, nullptr, nullptr);
}

int segs_neural_forward(const segs_neural_dims* dims, int A, const float* anchor, const float* offset, const float* anchor_feat,
                        const float* scaling_log, const int* visible_radii, const float* mlp_params, const float* camera_center,
                        const float* pose7, float* means3D, float* colors, float* opacity, float* scales, float* rotations,
                        float* neural_opacity, char* temp, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  Layout L;
  int rc = make_layout(dims, &L, nullptr, nullptr, nullptr);
  if (rc != SEGS_OK) return rc;
  if (A < 0) return SEGS_ERR_INVALID_ARGUMENT;
  if (A == 0) return SEGS_OK;
  if (!anchor || !offset || !anchor_feat || !scaling_log || !mlp_params || !camera_center || !means3D || !colors || !opacity ||
      !scales || !rotations || !neural_opacity || !temp || (L.app > 0 && !pose7))
    return SEGS_ERR_INVALID_ARGUMENT;
  Temp T;
  temp_carve(A, L.total, temp, &T);
  hipError_t e = hipMemsetAsync(T.count, 0, sizeof(uint32_t), st);
  if (e != hipSuccess) return (int)e;
  const int nb = (A + 255) / 256;
  compact_visible_kernel<<<nb, 256, 0, st>>>(A, visible_radii, T.count, T.vis, opacity, neural_opacity);
  neural_fwd_kernel<<<nb, 256, 0, st>>>(L, T.count, T.vis, anchor, offset, anchor_feat, scaling_log, mlp_params, camera_center,
                                        pose7, means3D, colors, opacity, scales, rotations, neural_opacity);
  e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : (int)e;
}

int segs_neural_backward(const segs_neural_dims* dims, int A, const float* anchor, const float* offset, const float* anchor_feat,
                         const float* scaling_log, const float* mlp_params, const float* camera_center, const float* pose7,
                         const float* dL_dmeans3D, const float* dL_dcolors, const float* dL_dopacity, const float* dL_dscales,
                         const float* dL_drotations, float* dL_danchor, float* dL_doffset, float* dL_dfeat,
                         float* dL_dscaling_log, float* dL_dmlp_params, char* temp, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  Layout L;
  int rc = make_layout(dims, &L, nullptr, nullptr, nullptr);
  if (rc != SEGS_OK) return rc;
  if (A < 0) return SEGS_ERR_INVALID_ARGUMENT;
  if (A == 0) return SEGS_OK;
  if (!anchor || !offset || !anchor_feat || !scaling_log || !mlp_params || !camera_center || !dL_dmeans3D || !dL_dcolors ||
      !dL_dopacity || !dL_dscales || !dL_drotations || !dL_danchor || !dL_doffset || !dL_dfeat || !dL_dscaling_log ||
      !dL_dmlp_params || !temp || (L.app > 0 && !pose7))
    return SEGS_ERR_INVALID_ARGUMENT;
  Temp T;
  temp_carve(A, L.total, temp, &T);
  const int nb = (A + 255) / 256;
  neural_bwd_kernel<<<nb, 256, 0, st>>>(L, T.count, T.vis, anchor, offset, anchor_feat, scaling_log, mlp_params, camera_center,
                                        pose7, dL_dmeans3D, dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, dL_danchor,
                                        dL_doffset, dL_dfeat, dL_dscaling_log, T.rows);
  const WJobs J = make_jobs(L);
  wgrad_mfma_kernel<<<dim3(WG_WAVES, WG_JOBS), 64, 0, st>>>(J, T.count, T.rows, T.partial);
  wgrad_reduce_kernel<<<dim3((37 * 70 + 255) / 256, WG_JOBS), 256, 0, st>>>(J, T.partial, T.gsum, dL_dmlp_params);
  if (L.app > 0) appearance_finish_kernel<<<1, 64, 0, st>>>(L, mlp_params, pose7, T.gsum, dL_dmlp_params);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : (int)e;
}

}  // extern "C"
