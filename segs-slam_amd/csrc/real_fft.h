// real_fft.h -- the two real 2-D transforms of the frequency regulariser, written for its image sizes (freq_loss.hip).
// Reference: torch::fft::fft2 of a real (3,H,W) image inside loss_utils::high_frequency_loss / multi_scale_loss
// (include/loss_utils.h:147-165, 216-237) and the autograd backward of it (an unnormalised inverse transform of a Hermitian
// spectrum, freq_loss.hip's header).  The reference -- and rounds 3-4 of this build -- leave both to the vendor FFT library,
// which runs each as FOUR launches: row pass on the even/odd-packed real data, post-process + transpose, column pass,
// transpose back (35 + 39 us at 1200x680, two thirds of the regulariser).  Here each direction is TWO launches and no
// transpose:
//   rows     a workgroup of 512 threads takes four image rows: the W reals are W/2 complex numbers, a mixed-radix Stockham FFT
//            of length W/2 runs in LDS (ping-pong buffers, radices 8 / 4 / 2 / 3 / 5 / 17 -- 600 = 8.3.5.5, 320 = 8.8.5,
//            960 = 8.8.3.5 -- one barrier per stage, the butterflies of all four rows dealt over the threads), then the split into
//            the W/2 + 1 non-redundant coefficients; the inverse direction runs the same steps backwards and ADDS its result into
//            dL/dimage (the separate add launch is gone);
//   columns  a workgroup of 512 threads takes 8 adjacent half-spectrum columns of one channel -- ONE contiguous block of the
//            tile-major storage (tiled_index below) -- transposes it into LDS, runs the eight length-H transforms (680 = 8.5.17,
//            480 = 8.4.3.5, 1080 = 8.3.3.3.5) stage by stage and writes the block back; the inverse pass also folds the loss
//            partials of the kernel in front of it (the separate finish launch is gone).
// Twiddle factors come from tables of N-th roots of unity made once per plan in double precision.
// Sizes whose factors are not all in {2, 3, 5, 17}, or whose tiles do not fit the LDS, keep the library transforms.
#pragma once
#include <hip/hip_runtime.h>

// freq_loss.hip is built with contraction off (its target tables and its spectra must come out of the same float32 operations in
// two template instantiations of one kernel).  The transforms are ONE set of kernels for both, so they may contract: a third fewer
// vector instructions in the butterflies and one rounding less per multiply-add.
#pragma clang fp contract(fast)

namespace rfft {

constexpr int MAX_STAGES = 10;
constexpr int TILE_COLS = 8;
struct Stages { int n; int radix[MAX_STAGES]; };

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 muli(float2 a, float sgn) { return make_float2(-sgn * a.y, sgn * a.x); }   // a * (sgn i)

// Storage of a half spectrum (C, H, Wc): tiles of TILE_COLS adjacent columns, each one contiguous (H x TILE_COLS) block.  The
// column pass then streams whole tiles (as 32- or 64-byte pieces of rows 4.8 KB apart it ran at the speed of its loads and stores:
// 17 of its 21 us at 1200x680 with the transform compiled out); the row passes write / read 64-byte pieces instead.
__host__ __device__ __forceinline__ int tiles_of(int Wc) { return (Wc + TILE_COLS - 1) / TILE_COLS; }
__host__ __device__ __forceinline__ size_t tiled_index(int c, int ky, int kx, int H, int Wc) {
  return (size_t)(((uint32_t)c * tiles_of(Wc) + (uint32_t)(kx / TILE_COLS)) * H + ky) * TILE_COLS + (uint32_t)(kx % TILE_COLS);
}

template <int R> struct Roots;
template <> struct Roots<3> {
  static constexpr float c[3] = {1.000000000e+00f, -5.000000000e-01f, -5.000000000e-01f};
  static constexpr float s[3] = {0.000000000e+00f, 8.660254038e-01f, -8.660254038e-01f};
};
template <> struct Roots<5> {
  static constexpr float c[5] = {1.000000000e+00f, 3.090169944e-01f, -8.090169944e-01f, -8.090169944e-01f, 3.090169944e-01f};
  static constexpr float s[5] = {0.000000000e+00f, 9.510565163e-01f, 5.877852523e-01f, -5.877852523e-01f, -9.510565163e-01f};
};
template <> struct Roots<17> {
  static constexpr float c[17] = {1.000000000e+00f, 9.324722294e-01f, 7.390089172e-01f, 4.457383558e-01f, 9.226835946e-02f, -2.736629901e-01f, -6.026346364e-01f, -8.502171357e-01f, -9.829730997e-01f, -9.829730997e-01f, -8.502171357e-01f, -6.026346364e-01f, -2.736629901e-01f, 9.226835946e-02f, 4.457383558e-01f, 7.390089172e-01f, 9.324722294e-01f};
  static constexpr float s[17] = {0.000000000e+00f, 3.612416662e-01f, 6.736956436e-01f, 8.951632914e-01f, 9.957341763e-01f, 9.618256432e-01f, 7.980172273e-01f, 5.264321629e-01f, 1.837495178e-01f, -1.837495178e-01f, -5.264321629e-01f, -7.980172273e-01f, -9.618256432e-01f, -9.957341763e-01f, -8.951632914e-01f, -6.736956436e-01f, -3.612416662e-01f};
};

// r-point DFT in place, X_p = sum_q v_q e^{sgn 2 pi i p q / r}
template <int R> __device__ __forceinline__ void dft(float2 (&v)[R], float sgn);
template <> __device__ __forceinline__ void dft<2>(float2 (&v)[2], float) {
  const float2 a = v[0], b = v[1];
  v[0] = cadd(a, b); v[1] = csub(a, b);
}
template <> __device__ __forceinline__ void dft<4>(float2 (&v)[4], float sgn) {
  const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = muli(csub(v[1], v[3]), sgn);
  v[0] = cadd(t0, t2); v[2] = csub(t0, t2); v[1] = cadd(t1, t3); v[3] = csub(t1, t3);
}
template <> __device__ __forceinline__ void dft<8>(float2 (&v)[8], float sgn) {
  float2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
  dft<4>(e, sgn); dft<4>(o, sgn);
  const float h = 0.70710678118654752f;
  // w8^p = e^{sgn 2 pi i p / 8}: 1, (1 + sgn i) h, sgn i, (-1 + sgn i) h
  const float2 o1 = make_float2(h * (o[1].x - sgn * o[1].y), h * (o[1].y + sgn * o[1].x));
  const float2 o2 = muli(o[2], sgn);
  const float2 o3 = make_float2(h * (-o[3].x - sgn * o[3].y), h * (-o[3].y + sgn * o[3].x));
  v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
  v[1] = cadd(e[1], o1);   v[5] = csub(e[1], o1);
  v[2] = cadd(e[2], o2);   v[6] = csub(e[2], o2);
  v[3] = cadd(e[3], o3);   v[7] = csub(e[3], o3);
}
// odd prime r: pairs a_q = v_q + v_{r-q}, b_q = v_q - v_{r-q};  X_p = v_0 + sum_q c_{pq} a_q + sgn i sum_q s_{pq} b_q, X_{r-p} its mirror
template <int R> __device__ __forceinline__ void dft_odd(float2 (&v)[R], float sgn) {
  constexpr int Hh = (R - 1) / 2;
  float2 a[Hh], b[Hh];
#pragma unroll
  for (int q = 1; q <= Hh; q++) { a[q - 1] = cadd(v[q], v[R - q]); b[q - 1] = csub(v[q], v[R - q]); }
  const float2 x0 = v[0];
  float2 s0 = x0;
#pragma unroll
  for (int q = 0; q < Hh; q++) s0 = cadd(s0, a[q]);
  v[0] = s0;
#pragma unroll
  for (int p = 1; p <= Hh; p++) {
    float2 A = x0, B = make_float2(0.f, 0.f);
#pragma unroll
    for (int q = 1; q <= Hh; q++) {
      const float c = Roots<R>::c[(p * q) % R], s = Roots<R>::s[(p * q) % R];
      A.x += c * a[q - 1].x; A.y += c * a[q - 1].y;
      B.x += s * b[q - 1].x; B.y += s * b[q - 1].y;
    }
    const float2 iB = muli(B, sgn);
    v[p] = cadd(A, iB); v[R - p] = csub(A, iB);
  }
}
template <> __device__ __forceinline__ void dft<3>(float2 (&v)[3], float sgn) { dft_odd<3>(v, sgn); }
template <> __device__ __forceinline__ void dft<5>(float2 (&v)[5], float sgn) { dft_odd<5>(v, sgn); }
template <> __device__ __forceinline__ void dft<17>(float2 (&v)[17], float sgn) { dft_odd<17>(v, sgn); }

// i / d for 0 <= i < 2^21, d >= 1 (inv = 1.0f / d): (i + 0.5) / d is at least 0.5 / d away from an integer and the float product
// is off by less than 2^-22 of its value, so the truncation is exact -- four instructions instead of the ~30 of an integer division.
__device__ __forceinline__ int fdiv(int i, float inv) { return (int)(((float)i + 0.5f) * inv); }

// One Stockham stage of `nseq` length-N transforms that stand in LDS `pitch` elements apart, executed by the whole workgroup:
// butterfly i = (sequence i / m, index j = i mod m), m = N / R.  root[n] = e^{-2 pi i n / N}.
//   in[j + q m] * root^{k q N / (Ns R)}  ->  DFT_R  ->  out[(j / Ns) Ns R + k + p Ns],   k = j mod Ns
// (Ns = product of the radices already done: outputs stand in natural order after the last stage; the first stage's twiddles
// are all 1 and are skipped).
template <int R>
__device__ __forceinline__ void stage(const float2* __restrict__ in, float2* __restrict__ out, int N, int Ns, int nseq, int pitch,
                                      const float2* __restrict__ root, float sgn) {
  const int m = N / R, tw = N / (Ns * R), total = nseq * m;
  const float inv_m = 1.0f / (float)m, inv_ns = 1.0f / (float)Ns;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int sq = fdiv(i, inv_m), j = i - sq * m;
    const int blk = fdiv(j, inv_ns), k = j - blk * Ns;
    const float2* src = in + (size_t)sq * pitch;
    float2 v[R];
#pragma unroll
    for (int q = 0; q < R; q++) v[q] = src[j + q * m];
    if (Ns > 1) {
#pragma unroll
      for (int q = 1; q < R; q++) {
        float2 w = root[k * q * tw];          // k q tw < N
        w.y = sgn < 0.f ? w.y : -w.y;          // the table holds the forward roots
        v[q] = cmul(v[q], w);
      }
    }
    dft<R>(v, sgn);
    float2* dst = out + (size_t)sq * pitch + blk * Ns * R + k;
#pragma unroll
    for (int p = 0; p < R; p++) dst[p * Ns] = v[p];
  }
}

// The whole transform of the workgroup's sequences (every thread of the workgroup calls it: one barrier per stage); returns
// the buffer that holds the results (a or b).
__device__ __forceinline__ float2* transform(float2* a, float2* b, int N, const Stages& st, int nseq, int pitch, const float2* root, float sgn) {
  int Ns = 1;
  for (int s = 0; s < st.n; s++) {
    const int r = st.radix[s];
    switch (r) {
      case 2: stage<2>(a, b, N, Ns, nseq, pitch, root, sgn); break;
      case 3: stage<3>(a, b, N, Ns, nseq, pitch, root, sgn); break;
      case 4: stage<4>(a, b, N, Ns, nseq, pitch, root, sgn); break;
      case 5: stage<5>(a, b, N, Ns, nseq, pitch, root, sgn); break;
      case 8: stage<8>(a, b, N, Ns, nseq, pitch, root, sgn); break;
      default: stage<17>(a, b, N, Ns, nseq, pitch, root, sgn); break;
    }
    __syncthreads();
    Ns *= r;
    float2* t = a; a = b; b = t;
  }
  return a;
}

// ---- rows, forward: (rows, W) real -> (rows, W/2 + 1) complex ------------------------------------------------------------
// A workgroup of ROW_THREADS threads takes ROWS_PER_WG consecutive rows.  LDS (float2): rootM[M] | a[ROWS][M + 1] | b[ROWS][M + 1],
// M = W/2;  rootW[n] = e^{-2 pi i n / W} (global), rootM[n] = rootW[2 n].
constexpr int GU = 5;   // global loads of a thread in flight together (row and column passes)
constexpr int ROWS_PER_WG = 4, ROW_THREADS = 512;   // (ROWS_PER_WG and TILE_COLS are powers of two: index masks below)
static_assert((ROWS_PER_WG & (ROWS_PER_WG - 1)) == 0 && (TILE_COLS & (TILE_COLS - 1)) == 0, "index masks");
__global__ void __launch_bounds__(ROW_THREADS) rows_r2c_kernel(const float* __restrict__ img, float2* __restrict__ X, int rows, int H, int W, Stages st,
                                                               const float2* __restrict__ rootW) {
  extern __shared__ __align__(16) float2 lds2[];
  const int M = W / 2, pitch = M + 1;
  float2* rootM = lds2;
  float2* a = lds2 + M;
  float2* b = a + (size_t)ROWS_PER_WG * pitch;
  const int row0 = blockIdx.x * ROWS_PER_WG, nrows = min(ROWS_PER_WG, rows - row0);
  for (int n = threadIdx.x; n < M; n += ROW_THREADS) rootM[n] = rootW[2 * n];
  const float inv_M = 1.0f / (float)M, inv_M1 = 1.0f / (float)(M + 1);
  // z[n] = x[2n] + i x[2n+1].  Global loads in groups of GU, all of a group in flight before the first is used: as a plain loop
  // every iteration waited for its own load (five dependent round trips per pass at 1200 columns).
  for (int base = 0; base < nrows * M; base += GU * ROW_THREADS) {
    float2 v[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = base + u * ROW_THREADS + (int)threadIdx.x;
      const int r = fdiv(i, inv_M), n = i - r * M;
      v[u] = i < nrows * M ? reinterpret_cast<const float2*>(img + (size_t)(row0 + r) * W)[n] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = base + u * ROW_THREADS + (int)threadIdx.x;
      const int r = fdiv(i, inv_M), n = i - r * M;
      if (i < nrows * M) a[(size_t)r * pitch + n] = v[u];
    }
  }
  __syncthreads();
#if defined(SEGS_MEASURE) && defined(RFFT_ABLATE_NO_ROW_TRANSFORM)   // measurement only
  const float2* Z = a;
#else
  const float2* Z = transform(a, b, M, st, nrows, pitch, rootM, -1.f);
#endif
  // X[k] = E + e^{-2 pi i k / W} O,  E = (Z[k] + conj Z[M-k]) / 2,  O = (Z[k] - conj Z[M-k]) / (2 i),  k = 0 .. M  (Z[M] = Z[0])
  // Store order: the workgroup's four rows of one column tile are ONE contiguous 256-byte run of the tile-major spectrum, so
  // consecutive lanes take (row, column within the tile) of one tile rather than consecutive columns of one row.
  const int c = row0 / H, y0 = row0 - c * H, ntl = tiles_of(M + 1);
  (void)inv_M1;
  for (int i = threadIdx.x; i < ntl * ROWS_PER_WG * TILE_COLS; i += ROW_THREADS) {
    const int kk = i & (TILE_COLS - 1), r = (i / TILE_COLS) & (ROWS_PER_WG - 1), tl = i / (TILE_COLS * ROWS_PER_WG);
    const int k = tl * TILE_COLS + kk;
    if (r >= nrows || k > M) continue;
    const float2* z = Z + (size_t)r * pitch;
    const float2 zk = z[k == M ? 0 : k], zm = z[k == 0 ? 0 : M - k];
    const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    const float2 d = make_float2(zk.x - zm.x, zk.y + zm.y);                    // Z[k] - conj Z[M-k]
    const float2 o = make_float2(0.5f * d.y, -0.5f * d.x);                     // / (2 i)
    X[tiled_index(c, y0 + r, k, H, M + 1)] = cadd(e, cmul(rootW[k], o));
  }
}

// ---- rows, inverse: (rows, W/2 + 1) Hermitian half -> (rows, W) real, unnormalised, ADDED into dst --------------------------
__global__ void __launch_bounds__(ROW_THREADS) rows_c2r_add_kernel(const float2* __restrict__ D, float* __restrict__ dst_img, int rows, int H, int W, Stages st,
                                                                   const float2* __restrict__ rootW) {
  extern __shared__ __align__(16) float2 lds2[];
  const int M = W / 2, pitch = M + 1;
  float2* rootM = lds2;
  float2* a = lds2 + M;
  float2* b = a + (size_t)ROWS_PER_WG * pitch;
  const int row0 = blockIdx.x * ROWS_PER_WG, nrows = min(ROWS_PER_WG, rows - row0);
  for (int n = threadIdx.x; n < M; n += ROW_THREADS) rootM[n] = rootW[2 * n];
  const float inv_M = 1.0f / (float)M;
  {
    const int c = row0 / H, y0 = row0 - c * H, ntl = tiles_of(M + 1);     // (load order: as the forward pass stores, 256-byte runs)
    const int total = ntl * ROWS_PER_WG * TILE_COLS;
    for (int base = 0; base < total; base += GU * ROW_THREADS) {
      float2 v[GU];
#pragma unroll
      for (int u = 0; u < GU; u++) {
        const int i = base + u * ROW_THREADS + (int)threadIdx.x;
        const int kk = i & (TILE_COLS - 1), r = (i / TILE_COLS) & (ROWS_PER_WG - 1), tl = i / (TILE_COLS * ROWS_PER_WG);
        const int k = tl * TILE_COLS + kk;
        v[u] = (i < total && r < nrows && k <= M) ? D[tiled_index(c, y0 + r, k, H, M + 1)] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < GU; u++) {
        const int i = base + u * ROW_THREADS + (int)threadIdx.x;
        const int kk = i & (TILE_COLS - 1), r = (i / TILE_COLS) & (ROWS_PER_WG - 1), tl = i / (TILE_COLS * ROWS_PER_WG);
        const int k = tl * TILE_COLS + kk;
        if (i < total && r < nrows && k <= M) {
          float2 w = v[u];
          if (k == 0 || k == M) w.y = 0.f;      // a real signal's DC and Nyquist coefficients are real (what a C2R transform assumes)
          b[(size_t)r * pitch + k] = w;
        }
      }
    }
  }
  __syncthreads();
  // Zin[k] = E + i O,  E = D[k] + conj D[M-k],  O = (D[k] - conj D[M-k]) e^{+2 pi i k / W},  k = 0 .. M-1
  for (int i = threadIdx.x; i < nrows * M; i += ROW_THREADS) {
    const int r = fdiv(i, inv_M), k = i - r * M;
    const float2* d0 = b + (size_t)r * pitch;
    const float2 dk = d0[k], dm = d0[M - k];
    const float2 e = make_float2(dk.x + dm.x, dk.y - dm.y);
    const float2 d = make_float2(dk.x - dm.x, dk.y + dm.y);
    float2 w = rootW[k];
    w.y = -w.y;
    const float2 o = cmul(d, w);
    a[(size_t)r * pitch + k] = make_float2(e.x - o.y, e.y + o.x);
  }
  __syncthreads();
#if defined(SEGS_MEASURE) && defined(RFFT_ABLATE_NO_ROW_TRANSFORM)
  const float2* z = a;
#else
  const float2* z = transform(a, b, M, st, nrows, pitch, rootM, +1.f);
#endif
  for (int base = 0; base < nrows * M; base += GU * ROW_THREADS) {
    float2 v[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = base + u * ROW_THREADS + (int)threadIdx.x;
      const int r = fdiv(i, inv_M), n = i - r * M;
      v[u] = i < nrows * M ? reinterpret_cast<const float2*>(dst_img + (size_t)(row0 + r) * W)[n] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = base + u * ROW_THREADS + (int)threadIdx.x;
      const int r = fdiv(i, inv_M), n = i - r * M;
      if (i < nrows * M) {
        const float2 zz = z[(size_t)r * pitch + n];
        reinterpret_cast<float2*>(dst_img + (size_t)(row0 + r) * W)[n] = make_float2(v[u].x + zz.x, v[u].y + zz.y);
      }
    }
  }
}

// ---- columns, either direction, in place: X (C, H, Wc) complex, transform along H --------------------------------------------
// A workgroup takes TILE_COLS adjacent columns of one channel.  LDS: rootH[H] | a[TILE_COLS][H] | b[TILE_COLS][H]
constexpr int COL_THREADS = 512;
// partial != null (the inverse pass of the regulariser): workgroup (0, 0) also folds the loss partials the kernel in front of this one
// left -- in double, a few hundred terms of very different size -- into *sum_out and adds the sum to *sum_inout (the 4.6-us launch
// that did only this is gone).
__global__ void __launch_bounds__(COL_THREADS) cols_kernel(float2* __restrict__ X, int H, int Wc, Stages st, const float2* __restrict__ rootH_g, float sgn,
                                                           const float* __restrict__ partial, int npartial, float* __restrict__ sum_out,
                                                           float* __restrict__ sum_inout) {
  extern __shared__ __align__(16) float2 lds2[];
  float2* rootH = lds2;
  float2* A = lds2 + H;
  float2* B = A + (size_t)TILE_COLS * H;
  const int kx0 = blockIdx.x * TILE_COLS, c = blockIdx.y;
  const int ncols = min(TILE_COLS, Wc - kx0);
  float2* tile = X + ((size_t)c * tiles_of(Wc) + blockIdx.x) * H * TILE_COLS;     // [y][t], contiguous
  for (int n = threadIdx.x; n < H; n += COL_THREADS) rootH[n] = rootH_g[n];
  for (int base = 0; base < H * TILE_COLS; base += GU * COL_THREADS) {     // (GU loads in flight per thread, as in the row passes)
    float2 v[GU];
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = base + u * COL_THREADS + (int)threadIdx.x;
      v[u] = i < H * TILE_COLS ? tile[i] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < GU; u++) {
      const int i = base + u * COL_THREADS + (int)threadIdx.x;
      const int y = i / TILE_COLS, t = i - y * TILE_COLS;
      if (i < H * TILE_COLS) A[(size_t)t * H + y] = v[u];     // (columns beyond Wc hold whatever the buffer holds: transformed nowhere)
    }
  }
  __syncthreads();
#if defined(SEGS_MEASURE) && defined(RFFT_ABLATE_NO_TRANSFORM)   // measurement only (tools/ab_fft_variant.sh): tile in, tile out
  const float2* R = A;
#else
  const float2* R = transform(A, B, H, st, ncols, H, rootH, sgn);
#endif
  for (int i = threadIdx.x; i < H * TILE_COLS; i += COL_THREADS) {
    const int y = i / TILE_COLS, t = i - y * TILE_COLS;
    if (t < ncols) tile[i] = R[(size_t)t * H + y];
  }
  if (partial != nullptr && blockIdx.x == 0 && blockIdx.y == 0) {
    __syncthreads();
    double* red = reinterpret_cast<double*>(lds2);
    double acc = 0.0;
    for (int i = threadIdx.x; i < npartial; i += COL_THREADS) acc += (double)partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double sum = 0.0;
      for (int w = 0; w < COL_THREADS / 64; w++) sum += red[w];
      *sum_out = (float)sum;
      if (sum_inout) *sum_inout += (float)sum;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// radices of N from {8, 4, 2, 3, 5, 17}; false when another prime is left
inline bool factorize(int N, Stages& st) {
  st.n = 0;
  if (N < 2) return false;
  const int cand[6] = {8, 4, 2, 3, 5, 17};
  for (int c = 0; c < 6; c++)
    while (N % cand[c] == 0) {
      if (st.n == MAX_STAGES) return false;
      st.radix[st.n++] = cand[c];
      N /= cand[c];
    }
  return N == 1;
}
inline size_t rows_lds_bytes(int W) { return (size_t)(W / 2 + ROWS_PER_WG * 2 * (W / 2 + 1)) * sizeof(float2); }
inline size_t cols_lds_bytes(int H) { return (size_t)(H + 2 * TILE_COLS * H) * sizeof(float2); }

}  // namespace rfft

#pragma clang fp contract(off)
