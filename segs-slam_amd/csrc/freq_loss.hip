// freq_loss.hip -- the mapper's frequency regulariser around the library FFTs (include/segs_train.h, segs_freq_*).
// Reference: loss_utils::high_frequency_loss / multi_scale_loss (include/loss_utils.h:126-165, 216-237) as used at
// src/gaussian_mapper.cpp:930-945:
//     loss += lambda_high * sum_s  s * mean( | |fft2(resize_s(image))| - |fft2(resize_s(gt))| | ),   s in {1, 1/2, 1/4}
// (the "high-pass" mask of :139-140 is indexed on (channel, row) and therefore all ones for any real image size, and
// fftshift does not change a mean -- tests/golden/loss_reference.npz, generated from the reference's compiled header,
// pins both statements).  The reference evaluates this with ~10 ATen kernels per scale and direction plus autograd.
// Here the two library transforms per scale (real-to-complex forward, complex-to-real inverse; hipFFT through the
// caller's tensor library, as in the reference) are the only image-sized passes besides three small kernels:
//   freq_pyramid_fwd_kernel   image -> the down-scaled copies (bilinear, align_corners = false), all scales in one launch
//   freq_spectrum_kernel      G = rfft2(level):  loss partial  w * m(kx) * | |G| - |T| |,   G <- w * sign(|G| - |T|) * G / |G|
//                             in place, all scales in one launch; |T| of the target is cached by the host per keyframe
//   freq_pyramid_bwd_kernel   dL/dimage += sum_s  resize_s^T( irfft2_unnormalised(G_s) )       (gather form, no atomics)
// When H and W are multiples of 4 and the scales are 1, 1/2, 1/4 (every shipped configuration: 1200x680, 640x480), the
// half- and quarter-size copies are exact 2-tap decimations (src = s d + (s - 1)/2: taps s d + s/2 - 1 and s d + s/2, weights
// 1/2), and the spectrum of a decimated signal is an alias fold of the full-size one:
//     Y_s[k] = 1/s^2 sum_{a,b < s}  X[ky + a H/s, kx + b W/s] * C_s(ky + a H/s; H) * C_s(kx + b W/s; W),
//     C_2(k; N) = (1 + e^{2 pi i k/N}) / 2,      C_4(k; N) = (e^{2 pi i k/N} + e^{4 pi i k/N}) / 2.
// freq_fold_kernel therefore needs ONE forward and ONE inverse transform per step, both at full size: a thread owns a
// quarter-size frequency (ky2, kx2), i.e. its 16 full-size aliases and the 4 half-size frequencies they fold into, evaluates
// all three scales' loss terms and writes  D[k'] = Q_0[k'] + Q_1[k' mod] conj(C_2 C_2)/4 + Q_2[k' mod] conj(C_4 C_4)/16  with
// Q_l = w_l sign(|Y_l| - |T_l|) Y_l/|Y_l|; the inverse transform of D is the gradient through all three scales, resizes included.
// Why one inverse transform is the whole backward: L = w sum_k | |G_k| - |T_k| | with G_k = sum_n g_n e^{-i th_kn} gives
//   dL/dg_n = w sum_k sign_k Re( conj(G_k)/|G_k| e^{-i th_kn} ) = Re sum_k (w sign_k G_k/|G_k|) e^{+i th_kn},
// the unnormalised inverse DFT of a Hermitian spectrum, i.e. exactly what a C2R transform of its non-redundant half
// returns.  In the half spectrum a column kx stands for itself and its mirror image W - kx, hence m(kx) = 2 except for
// kx = 0 and kx = W/2 (W even).  d|z|/dz at z = 0 and sign(0) are 0, as in LibTorch's abs backward.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <initializer_list>
#include <vector>
#include "../../include/segs_raster.h"
#include "kernels.h"
#include "../../include/segs_train.h"
#include "real_fft.h"

namespace {
constexpr int MAXL = SEGS_FREQ_MAX_LEVELS;

struct PyrLevels {
  int n;
  int h[MAXL], w[MAXL];
  float* out[MAXL];          // forward: (C, h, w) destination;  backward: (C, h, w) gradient (read)
  unsigned first[MAXL + 1];  // forward: first flat element of the level in the launch
};

// upsample_bilinear2d with align_corners = false and no explicit scale (recompute_scale_factor = true hands the kernel
// sizes only): scale = in / out in float, src = scale * (dst + 0.5) - 0.5 clamped at 0, taps (i0, min(i0 + 1, in - 1)).
__device__ __forceinline__ void bilinear_tap(int dst, int in, float scale, int& i0, int& i1, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 > in - 1 ? in - 1 : i0;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

__global__ void __launch_bounds__(256) freq_pyramid_fwd_kernel(const float* __restrict__ img, int C, int H, int W, PyrLevels lv) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= lv.first[lv.n]) return;
  int l = 0;
#pragma unroll
  for (int i = 1; i < MAXL; i++) l += (i < lv.n && t >= lv.first[i]) ? 1 : 0;
  const int h = lv.h[l], w = lv.w[l];
  const unsigned e = t - lv.first[l];
  const int x = e % w, y = (e / w) % h, c = e / ((unsigned)w * h);
  int y0, y1, x0, x1;
  float ly, lx;
  bilinear_tap(y, H, (float)H / (float)h, y0, y1, ly);
  bilinear_tap(x, W, (float)W / (float)w, x0, x1, lx);
  const float* p = img + (size_t)c * H * W;
  const float a00 = p[(size_t)y0 * W + x0], a01 = p[(size_t)y0 * W + x1];
  const float a10 = p[(size_t)y1 * W + x0], a11 = p[(size_t)y1 * W + x1];
  lv.out[l][e] = (1.f - ly) * ((1.f - lx) * a00 + lx * a01) + ly * ((1.f - lx) * a10 + lx * a11);
}

struct SpecLevels {
  int n;
  int w[MAXL];               // full width of the level (the half spectrum has w/2 + 1 columns)
  float2* spec[MAXL];        // (C, h, w/2+1) complex, in/out
  const float* tmag[MAXL];   // |T|, same shape, real
  float weight[MAXL];        // lambda * scale / (C h w)
  unsigned first[MAXL + 1];
};

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) freq_spectrum_kernel(SpecLevels lv, float* __restrict__ partial) {
  __shared__ float red[4];
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  float term = 0.f;
  if (t < lv.first[lv.n]) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < MAXL; i++) l += (i < lv.n && t >= lv.first[i]) ? 1 : 0;
    const unsigned e = t - lv.first[l];
    const int w = lv.w[l], wc = w / 2 + 1;
    const int kx = e % wc;
    const float2 g = lv.spec[l][e];
    const float m = sqrtf(g.x * g.x + g.y * g.y);
    const float d = m - lv.tmag[l][e];
    const float mult = (kx == 0 || (2 * kx == w)) ? 1.f : 2.f;
    const float wl = lv.weight[l];
    term = wl * mult * fabsf(d);
    const float s = d > 0.f ? wl : (d < 0.f ? -wl : 0.f);
    const float k = m > 0.f ? s / m : 0.f;
    lv.spec[l][e] = make_float2(k * g.x, k * g.y);
  }
  const float sum = block_sum_256(term, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = sum;
}

__global__ void __launch_bounds__(256) spectrum_magnitude_kernel(const float2* __restrict__ spec, size_t n, float* __restrict__ mag) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n) { const float2 g = spec[i]; mag[i] = sqrtf(g.x * g.x + g.y * g.y); }
}

// Sum of the partials in double (a few thousand terms of very different size: DC of the largest scale against the tail).
__global__ void __launch_bounds__(1024) freq_finish_kernel(const float* __restrict__ partial, int n, float* __restrict__ freq_out,
                                                            float* __restrict__ loss_inout) {
  __shared__ double r[16];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) a += (double)partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < 16; w++) s += r[w];
    *freq_out = (float)s;
    if (loss_inout) *loss_inout += (float)s;
  }
}

// Transposed bilinear resize, gather form.  Source index y is a tap of destination d iff i0(d) == y or i1(d) == y; the
// destinations that can qualify have src(d) in (y - 1, y + 1), i.e. d in ((y - 0.5) / scale - 0.5, (y + 1.5) / scale - 0.5):
// an interval of length 2 / scale <= 2 (levels are never larger than the image), covered with room for rounding by the NT
// candidates from floor(lower bound - 0.001) on.  Returns the first candidate and the candidates' weights (zero where y is
// not a tap of that destination; every candidate is tested with the forward's own arithmetic).
constexpr int NT = 4;
__device__ __forceinline__ int transposed_taps(int y, int in, int out, float scale, float (&wt)[NT]) {
  int d0 = (int)floorf(((float)y - 0.5f) / scale - 0.501f);
  d0 = d0 < 0 ? 0 : d0;
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int d = d0 + j;
    float w = 0.f;
    if (d < out) {
      int i0, i1;
      float l1;
      bilinear_tap(d, in, scale, i0, i1, l1);
      w = (i0 == y ? 1.f - l1 : 0.f) + (i1 == y ? l1 : 0.f);
    }
    wt[j] = w;
  }
  return d0;
}

__global__ void __launch_bounds__(256) freq_pyramid_bwd_kernel(float* __restrict__ dL, int C, int H, int W, PyrLevels lv) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = blockIdx.z;
  if (x >= W || y >= H) return;
  const size_t o = ((size_t)c * H + y) * W + x;
  float acc = dL[o];
  for (int l = 0; l < lv.n; l++) {
    const int h = lv.h[l], w = lv.w[l];
    const float* g = lv.out[l] + (size_t)c * h * w;
    if (h == H && w == W) { acc += g[(size_t)y * W + x]; continue; }
    float wy[NT], wx[NT];
    const int dy0 = transposed_taps(y, H, h, (float)H / (float)h, wy);
    const int dx0 = transposed_taps(x, W, w, (float)W / (float)w, wx);
#pragma unroll
    for (int j = 0; j < NT; j++) {
      if (wy[j] == 0.f) continue;
      float row = 0.f;
#pragma unroll
      for (int i = 0; i < NT; i++)
        if (wx[i] != 0.f) row += wx[i] * g[(size_t)(dy0 + j) * w + dx0 + i];
      acc += wy[j] * row;
    }
  }
  dL[o] = acc;
}

// ---- alias-folded path (H, W multiples of 4; scales 1, 1/2, 1/4) -------------------------------------------------------
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a * conj(b)
__device__ __forceinline__ float2 rot90(float2 e, int q) {   // e * i^q
  return q == 0 ? e : (q == 1 ? make_float2(-e.y, e.x) : (q == 2 ? make_float2(-e.x, -e.y) : make_float2(e.y, -e.x)));
}
// index of full-spectrum entry (ky, kx) of an h x w transform in its stored half (columns 0 .. w/2); `mirrored` tells the
// caller to conjugate (|.| tables ignore it)
// `tiled` (the plan's own transforms, real_fft.h): the half spectrum is stored in tiles of rfft::TILE_COLS adjacent columns, each
// tile one contiguous (h x TILE_COLS) block, so that the column pass streams whole tiles: [c][kx / T][ky][kx mod T]
// (32-bit element offsets: one address register per access instead of two -- the fold kernel has sixteen of them in flight; a
// plan's spectra stay far below 2^32 elements)
__device__ __forceinline__ uint32_t half_index(int c, int ky, int kx, int h, int w, bool& mirrored, bool tiled) {
  mirrored = kx > w / 2;
  if (mirrored) { ky = ky ? h - ky : 0; kx = w - kx; }
  if (tiled) return (uint32_t)rfft::tiled_index(c, ky, kx, h, w / 2 + 1);
  return ((uint32_t)c * h + ky) * (w / 2 + 1) + kx;
}
// Q = wl * sign(|y| - t) * y / |y|, and the loss term wl * | |y| - t |
__device__ __forceinline__ float2 coeff(float2 y, float t, float wl, float& term) {
  const float m = sqrtf(y.x * y.x + y.y * y.y);
  const float d = m - t;
  term += wl * fabsf(d);
  const float s = d > 0.f ? wl : (d < 0.f ? -wl : 0.f);
  const float k = m > 0.f ? s / m : 0.f;
  return make_float2(k * y.x, k * y.y);
}

// TARGET: write |Y_l| of the target's spectrum X into T0 (half spectrum, like X) and T1 / T2 (FULL spectra: the magnitude of
// a folded coefficient and of its Hermitian mirror are equal in exact arithmetic only -- they come out of different twiddles
// -- so each thread keeps the table entry of its own frequency and identical images give |Y| - |T| = 0 exactly, as they do in
// the reference).  Otherwise: loss partials + D (half spectrum).
template <bool TARGET>
__global__ void __launch_bounds__(256) freq_fold_kernel(int H, int W, const float2* __restrict__ X, float2* __restrict__ D,
                                                         float* __restrict__ T0, float* __restrict__ T1, float* __restrict__ T2,
                                                         float w0, float w1, float w2, float* __restrict__ partial, bool tiled) {
  __shared__ float red[4];
  const int H4 = H / 4, W4 = W / 4;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  float term = 0.f;
  // Which quarter-size frequency this thread owns.  Row-major spectra: consecutive threads take consecutive kx2.  Tiled spectra
  // (real_fft.h): a wave takes an 8 x 8 patch (kx2 low bits = lane & 7, ky2 low bits = lane >> 3), so that each of its accesses to
  // X, T0 and D covers eight adjacent rows of one or two column tiles -- 512 contiguous bytes -- instead of eight 64-byte pieces
  // 43 KB apart (21 us against 15 at 1200x680).
  int kx2, ky2, c;
  bool valid;
  if (tiled) {
    const unsigned nkx = ((unsigned)W4 + 7u) / 8u, nky = ((unsigned)H4 + 7u) / 8u;
    const unsigned patch = idx >> 6, within = idx & 63u;
    kx2 = (int)((patch % nkx) * 8u + (within & 7u));
    ky2 = (int)(((patch / nkx) % nky) * 8u + (within >> 3));
    c = (int)(patch / (nkx * nky));
    valid = c < 3 && kx2 < W4 && ky2 < H4;
  } else {
    kx2 = idx % W4; ky2 = (idx / W4) % H4; c = idx / ((unsigned)W4 * H4);
    valid = idx < 3u * H4 * W4;
  }
  if (valid) {
    // e^{2 pi i k/N} of the first alias per axis; the others differ by powers of i
    float2 ey, ex;
    sincospif(2.f * (float)ky2 / (float)H, &ey.y, &ey.x);
    sincospif(2.f * (float)kx2 / (float)W, &ex.y, &ex.x);
    // The fold factors C_2, C_4 of alias q along an axis are functions of e i^q (e = the axis' first-alias phase): formed where
    // they are used instead of kept as four tables of four -- with the sixteen aliases re-read in the second loop (below) this
    // kernel needs 60 registers less, i.e. four waves per SIMD instead of two (its 2 500 waves fit the chip in one round).
    auto c2 = [](float2 e0, int q) { const float2 e = rot90(e0, q); return make_float2(0.5f * (1.f + e.x), 0.5f * e.y); };
    auto c4 = [](float2 e0, int q) { const float2 e = rot90(e0, q), e2 = cmul(e, e); return make_float2(0.5f * (e.x + e2.x), 0.5f * (e.y + e2.y)); };
    float2 y1[2][2] = {}, y2 = make_float2(0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) {
        bool mir;
        const uint32_t o = half_index(c, ky2 + a * H4, kx2 + b * W4, H, W, mir, tiled);
        const float2 v = X[o];
        const float2 xab = mir ? make_float2(v.x, -v.y) : v;
        const float2 z1 = cmul(xab, cmul(c2(ey, a), c2(ex, b)));
        const float2 z2 = cmul(xab, cmul(c4(ey, a), c4(ex, b)));
        y1[a & 1][b & 1].x += z1.x; y1[a & 1][b & 1].y += z1.y;
        y2.x += z2.x; y2.y += z2.y;
      }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) { y1[a][b].x *= 0.25f; y1[a][b].y *= 0.25f; }
    y2.x *= 0.0625f; y2.y *= 0.0625f;
    const int h1 = H / 2, wd1 = W / 2;
    bool mir;
    if (TARGET) {
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
          if (b < 2 || kx2 == 0) {
            const uint32_t o = half_index(c, ky2 + a * H4, kx2 + b * W4, H, W, mir, tiled);
            const float2 xv = X[o];                 // (|.| of the stored entry; a mirrored one has the same magnitude)
            T0[o] = sqrtf(xv.x * xv.x + xv.y * xv.y);
          }
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
          T1[((size_t)c * h1 + ky2 + a * H4) * wd1 + kx2 + b * W4] = sqrtf(y1[a][b].x * y1[a][b].x + y1[a][b].y * y1[a][b].y);
      T2[((size_t)c * H4 + ky2) * W4 + kx2] = sqrtf(y2.x * y2.x + y2.y * y2.y);
    } else {
      float2 q1[2][2];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
          q1[a][b] = coeff(y1[a][b], T1[((size_t)c * h1 + ky2 + a * H4) * wd1 + kx2 + b * W4], w1, term);
      const float2 q2 = coeff(y2, T2[((size_t)c * H4 + ky2) * W4 + kx2], w2, term);
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          uint32_t o = half_index(c, ky2 + a * H4, kx2 + b * W4, H, W, mir, tiled);
          asm volatile("" : "+v"(o));              // (keeps the compiler from holding the first loop's sixteen loads instead)
          const float2 xv = X[o];
          const float2 xab = mir ? make_float2(xv.x, -xv.y) : xv;
          float2 d = coeff(xab, T0[o], w0, term);
          if (b < 2 || (b == 2 && kx2 == 0)) {      // the stored half: columns 0 .. W/2
            const float2 t1 = cmulc(q1[a & 1][b & 1], cmul(c2(ey, a), c2(ex, b)));
            const float2 t2 = cmulc(q2, cmul(c4(ey, a), c4(ex, b)));
            d.x += 0.25f * t1.x + 0.0625f * t2.x;
            d.y += 0.25f * t1.y + 0.0625f * t2.y;
            D[o] = d;
          }
        }
    }
  }
  if (!TARGET) {
    const float sum = block_sum_256(term, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = sum;
  }
}

__global__ void __launch_bounds__(256) add_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

int bad(const char* what) { return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, what); }

// ---- hipFFT, bound at run time --------------------------------------------------------------------------------------
// The process that loads this library has, as a rule, a tensor library in it that already carries a hipFFT (PyTorch-ROCm
// bundles libhipfft.so.0 next to its HIP runtime).  Linking a second copy in through DT_NEEDED could bind another HIP runtime;
// dlopen by soname picks the copy that is already loaded (RTLD_NOLOAD first) and only then the system one.
struct HipFft {
  hipfftResult (*PlanMany)(hipfftHandle*, int, int*, int*, int, int, int*, int, int, hipfftType, int) = nullptr;
  hipfftResult (*SetStream)(hipfftHandle, hipStream_t) = nullptr;
  hipfftResult (*ExecR2C)(hipfftHandle, hipfftReal*, hipfftComplex*) = nullptr;
  hipfftResult (*ExecC2R)(hipfftHandle, hipfftComplex*, hipfftReal*) = nullptr;
  hipfftResult (*Destroy)(hipfftHandle) = nullptr;
  bool ok = false;
};
const HipFft& hipfft() {
  static const HipFft api = [] {
    HipFft a;
    void* h = nullptr;
    for (const char* name : {"libhipfft.so.0", "libhipfft.so"}) {
      if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
    }
    if (!h) for (const char* name : {"libhipfft.so.0", "libhipfft.so"}) {
      if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!h) return a;
    a.PlanMany = reinterpret_cast<decltype(a.PlanMany)>(dlsym(h, "hipfftPlanMany"));
    a.SetStream = reinterpret_cast<decltype(a.SetStream)>(dlsym(h, "hipfftSetStream"));
    a.ExecR2C = reinterpret_cast<decltype(a.ExecR2C)>(dlsym(h, "hipfftExecR2C"));
    a.ExecC2R = reinterpret_cast<decltype(a.ExecC2R)>(dlsym(h, "hipfftExecC2R"));
    a.Destroy = reinterpret_cast<decltype(a.Destroy)>(dlsym(h, "hipfftDestroy"));
    a.ok = a.PlanMany && a.SetStream && a.ExecR2C && a.ExecC2R && a.Destroy;
    return a;
  }();
  return api;
}
}  // namespace

// The plan: transforms, scratch and the level table of one image size.
struct segs_freq_plan {
  int H = 0, W = 0, n = 0;
  int h[MAXL] = {}, w[MAXL] = {};
  float weight[MAXL] = {};
  size_t toff[MAXL + 1] = {};          // float offset of level l inside a target block
  bool folded = false;
  hipfftHandle r2c[MAXL] = {}, c2r[MAXL] = {};
  bool have[MAXL] = {};
  float* level[MAXL] = {};             // generic path: down-scaled copies (NULL for a full-size level)
  float* spec[MAXL] = {};              // half spectra (folded: [0] = X, [1] = D)
  float* grad[MAXL] = {};              // inverse transforms
  float* partial = nullptr;
  char* arena = nullptr;
  // folded plans whose sizes factor into {2, 3, 5, 17}: both full-size transforms by this library's own kernels (real_fft.h)
  bool own_fft = false;
  rfft::Stages st_rows{}, st_cols{};
  float2* root_w = nullptr;            // e^{-2 pi i n / W}, n < W
  float2* root_h = nullptr;            // e^{-2 pi i n / H}, n < H
};

extern "C" {

int segs_freq_pyramid(const float* image, int C, int H, int W, int nlevels, const int* level_h, const int* level_w,
                      float* const* level_out, void* stream) {
  if (!image || C <= 0 || H <= 0 || W <= 0 || nlevels < 0 || nlevels > MAXL || (nlevels && (!level_h || !level_w || !level_out)))
    return bad("segs_freq_pyramid: invalid argument");
  PyrLevels lv{};
  unsigned total = 0;
  for (int l = 0; l < nlevels; l++) {
    if (level_h[l] <= 0 || level_w[l] <= 0 || level_h[l] > H || level_w[l] > W) return bad("segs_freq_pyramid: a level must be no larger than the image");
    if (level_h[l] == H && level_w[l] == W) continue;        // scale 1: bilinear resize to the same size is the identity
    if (!level_out[l]) return bad("segs_freq_pyramid: null level buffer");
    const int k = lv.n++;
    lv.h[k] = level_h[l]; lv.w[k] = level_w[l]; lv.out[k] = level_out[l]; lv.first[k] = total;
    total += (unsigned)C * level_h[l] * level_w[l];
  }
  lv.first[lv.n] = total;
  if (!total) return SEGS_OK;
  freq_pyramid_fwd_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(image, C, H, W, lv);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

int segs_spectrum_magnitude(const float* spectrum, size_t n_complex, float* magnitude, void* stream) {
  if ((!spectrum || !magnitude) && n_complex) return bad("segs_spectrum_magnitude: null pointer");
  if (!n_complex) return SEGS_OK;
  spectrum_magnitude_kernel<<<(unsigned)((n_complex + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      reinterpret_cast<const float2*>(spectrum), n_complex, magnitude);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

size_t segs_freq_temp_bytes(int C, int nlevels, const int* level_h, const int* level_w) {
  size_t total = 0;
  for (int l = 0; l < nlevels && l < MAXL; l++) total += (size_t)C * level_h[l] * (level_w[l] / 2 + 1);
  return ((total + 255) / 256 + 1) * sizeof(float);
}

int segs_freq_spectrum_loss(int C, int nlevels, const int* level_h, const int* level_w, float* const* spectrum,
                            const float* const* target_magnitude, const float* level_weight, float* freq_loss_out,
                            float* loss_inout, char* temp, void* stream) {
  if (C <= 0 || nlevels <= 0 || nlevels > MAXL || !level_h || !level_w || !spectrum || !target_magnitude || !level_weight ||
      !freq_loss_out || !temp)
    return bad("segs_freq_spectrum_loss: invalid argument");
  SpecLevels lv{};
  unsigned total = 0;
  lv.n = nlevels;
  for (int l = 0; l < nlevels; l++) {
    if (level_h[l] <= 0 || level_w[l] <= 0 || !spectrum[l] || !target_magnitude[l]) return bad("segs_freq_spectrum_loss: bad level");
    lv.w[l] = level_w[l];
    lv.spec[l] = reinterpret_cast<float2*>(spectrum[l]);
    lv.tmag[l] = target_magnitude[l];
    lv.weight[l] = level_weight[l];
    lv.first[l] = total;
    total += (unsigned)C * level_h[l] * (level_w[l] / 2 + 1);
  }
  lv.first[nlevels] = total;
  const unsigned nblk = (total + 255) / 256;
  float* partial = reinterpret_cast<float*>(temp);
  hipStream_t st = (hipStream_t)stream;
  freq_spectrum_kernel<<<nblk, 256, 0, st>>>(lv, partial);
  freq_finish_kernel<<<1, 1024, 0, st>>>(partial, (int)nblk, freq_loss_out, loss_inout);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

int segs_freq_pyramid_backward_add(float* dL_dimage, int C, int H, int W, int nlevels, const int* level_h, const int* level_w,
                                   const float* const* level_grad, void* stream) {
  if (!dL_dimage || C <= 0 || H <= 0 || W <= 0 || nlevels <= 0 || nlevels > MAXL || !level_h || !level_w || !level_grad)
    return bad("segs_freq_pyramid_backward_add: invalid argument");
  PyrLevels lv{};
  lv.n = nlevels;
  for (int l = 0; l < nlevels; l++) {
    if (level_h[l] <= 0 || level_w[l] <= 0 || level_h[l] > H || level_w[l] > W || !level_grad[l])
      return bad("segs_freq_pyramid_backward_add: bad level");
    lv.h[l] = level_h[l]; lv.w[l] = level_w[l]; lv.out[l] = const_cast<float*>(level_grad[l]);
  }
  const dim3 grid((W + 63) / 64, (H + 3) / 4, C);
  freq_pyramid_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dL_dimage, C, H, W, lv);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

}  // extern "C"

// ---- plan-level entry points: the whole regulariser in one call, transforms included ------------------------------------
extern "C" {

void segs_freq_plan_destroy(segs_freq_plan* p) {
  if (!p) return;
  const HipFft& f = hipfft();
  for (int l = 0; l < MAXL; l++)
    if (p->have[l] && f.ok) { f.Destroy(p->r2c[l]); f.Destroy(p->c2r[l]); }
  if (p->arena) (void)hipFree(p->arena);
  delete p;
}

int segs_freq_plan_create(int H, int W, int nscales, const float* scales, float lambda_high, segs_freq_plan** out) {
  if (!out || H <= 0 || W <= 0 || nscales <= 0 || nscales > MAXL || !scales) return bad("segs_freq_plan_create: invalid argument");
  *out = nullptr;
  const HipFft& f = hipfft();
  segs_freq_plan* p = new segs_freq_plan();
  p->H = H; p->W = W; p->n = nscales;
  size_t toff = 0;
  for (int l = 0; l < nscales; l++) {
    // F.interpolate(scale_factor = s, recompute_scale_factor = true): output size floor(size * s) in double
    p->h[l] = (int)std::floor((double)H * (double)scales[l]);
    p->w[l] = (int)std::floor((double)W * (double)scales[l]);
    if (p->h[l] <= 0 || p->w[l] <= 0 || p->h[l] > H || p->w[l] > W) { delete p; return bad("segs_freq_plan_create: scale outside (0, 1]"); }
    p->weight[l] = lambda_high * scales[l] / (3.f * (float)p->h[l] * (float)p->w[l]);   // loss_utils.h:235: scale * mean(...)
  }
  p->folded = nscales == 3 && scales[0] == 1.f && scales[1] == 0.5f && scales[2] == 0.25f && H % 4 == 0 && W % 4 == 0;
  // SEGS_FREQ_HIPFFT=1: keep the library transforms (A/B measurements, and the fallback's tests)
  static const bool force_library = [] { const char* e = getenv("SEGS_FREQ_HIPFFT"); return e && e[0] == '1'; }();
  p->own_fft = p->folded && !force_library && rfft::factorize(W / 2, p->st_rows) && rfft::factorize(H, p->st_cols) &&
               rfft::rows_lds_bytes(W) <= 160 * 1024 && rfft::cols_lds_bytes(H) <= 160 * 1024;
  if (!p->own_fft && !f.ok) {
    delete p;
    return segs::set_error(SEGS_ERR_UNSUPPORTED, "segs_freq_plan_create: libhipfft.so.0 could not be loaded (no FFT library in this process)");
  }
  // columns STORED per row of the full-size half spectrum: W/2 + 1, rounded up to whole tiles under the plan's own transforms
  const size_t wc0 = p->own_fft ? (size_t)rfft::tiles_of(W / 2 + 1) * rfft::TILE_COLS : (size_t)(W / 2 + 1);
  for (int l = 0; l < nscales; l++) {     // target tables: half spectra; a folded plan keeps levels 1, 2 as full spectra
    p->toff[l] = toff;
    toff += (size_t)3 * p->h[l] * ((p->folded && l > 0) ? (size_t)p->w[l] : (l == 0 ? wc0 : (size_t)(p->w[l] / 2 + 1)));
  }
  p->toff[nscales] = toff;
  // scratch: one arena (256-byte aligned pieces)
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t bytes = 0, off_level[MAXL] = {}, off_spec[MAXL] = {}, off_grad[MAXL] = {};
  const int nfft = p->folded ? 1 : nscales;
  for (int l = 0; l < nscales; l++) {
    const size_t real = al((size_t)3 * p->h[l] * p->w[l] * sizeof(float)),
                 cplx = al((size_t)3 * p->h[l] * (l == 0 ? wc0 : (size_t)(p->w[l] / 2 + 1)) * sizeof(float2));
    if (p->folded) {
      if (l == 0) { off_spec[0] = bytes; bytes += cplx; off_spec[1] = bytes; bytes += cplx; off_grad[0] = bytes; bytes += real; }
    } else {
      if (p->h[l] != H || p->w[l] != W) { off_level[l] = bytes; bytes += real; } else off_level[l] = (size_t)-1;
      off_spec[l] = bytes; bytes += cplx;
      off_grad[l] = bytes; bytes += real;
    }
  }
  const size_t off_partial = bytes;
  bytes += al((toff / 256 + 2) * sizeof(float));
  const size_t off_root_w = bytes;
  bytes += al((size_t)W * sizeof(float2));
  const size_t off_root_h = bytes;
  bytes += al((size_t)H * sizeof(float2));
  if (hipMalloc(&p->arena, bytes) != hipSuccess) { delete p; return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "segs_freq_plan_create: out of device memory"); }
  p->partial = reinterpret_cast<float*>(p->arena + off_partial);
  if (p->own_fft) {
    p->root_w = reinterpret_cast<float2*>(p->arena + off_root_w);
    p->root_h = reinterpret_cast<float2*>(p->arena + off_root_h);
    std::vector<float2> tw((size_t)W), th((size_t)H);
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < W; n++) tw[n] = make_float2((float)std::cos(two_pi * n / W), (float)-std::sin(two_pi * n / W));
    for (int n = 0; n < H; n++) th[n] = make_float2((float)std::cos(two_pi * n / H), (float)-std::sin(two_pi * n / H));
    if (hipMemcpy(p->root_w, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->root_h, th.data(), th.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(rfft::rows_r2c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rfft::rows_lds_bytes(W)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(rfft::rows_c2r_add_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rfft::rows_lds_bytes(W)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(rfft::cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rfft::cols_lds_bytes(H)) != hipSuccess) {
      segs_freq_plan_destroy(p);
      return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "segs_freq_plan_create: could not set up the transform kernels");
    }
  }
  for (int l = 0; l < nscales; l++) {
    if (p->folded) {
      if (l < 2) p->spec[l] = reinterpret_cast<float*>(p->arena + off_spec[l]);
      if (l == 0) p->grad[0] = reinterpret_cast<float*>(p->arena + off_grad[0]);
    } else {
      p->level[l] = off_level[l] == (size_t)-1 ? nullptr : reinterpret_cast<float*>(p->arena + off_level[l]);
      p->spec[l] = reinterpret_cast<float*>(p->arena + off_spec[l]);
      p->grad[l] = reinterpret_cast<float*>(p->arena + off_grad[l]);
    }
  }
  for (int l = 0; l < (p->own_fft ? 0 : nfft); l++) {
    int n[2] = {p->h[l], p->w[l]};
    if (f.PlanMany(&p->r2c[l], 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_R2C, 3) != HIPFFT_SUCCESS ||
        f.PlanMany(&p->c2r[l], 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_C2R, 3) != HIPFFT_SUCCESS) {
      segs_freq_plan_destroy(p);
      return segs::set_error(SEGS_ERR_UNSUPPORTED, "segs_freq_plan_create: hipfftPlanMany failed");
    }
    p->have[l] = true;
  }
  *out = p;
  return SEGS_OK;
}

int segs_freq_plan_levels(const segs_freq_plan* p, int* level_h, int* level_w, int* folded) {
  if (!p) return bad("segs_freq_plan_levels: null plan");
  for (int l = 0; l < p->n; l++) { if (level_h) level_h[l] = p->h[l]; if (level_w) level_w[l] = p->w[l]; }
  if (folded) *folded = p->folded ? 1 : 0;
  return p->n;
}

size_t segs_freq_target_floats(const segs_freq_plan* p) { return p ? p->toff[p->n] : 0; }

static int fft_fail(const char* what) { return segs::set_error(SEGS_ERR_UNSUPPORTED, what); }
// threads of freq_fold_kernel: one per quarter-size frequency, in 8 x 8 patches (some of them partly empty) over tiled spectra
static unsigned fold_threads(const segs_freq_plan* p) {
  const unsigned h4 = p->H / 4, w4 = p->W / 4;
  return p->own_fft ? 3u * ((w4 + 7u) / 8u) * ((h4 + 7u) / 8u) * 64u : 3u * h4 * w4;
}

// own transforms of a folded plan (real_fft.h): image (3,H,W) -> half spectrum (3,H,W/2+1), and the way back, added into dL
static void own_forward(const segs_freq_plan* p, const float* image, float2* spec, hipStream_t st) {
  const int rows = 3 * p->H, wc = p->W / 2 + 1;
  rfft::rows_r2c_kernel<<<(rows + rfft::ROWS_PER_WG - 1) / rfft::ROWS_PER_WG, rfft::ROW_THREADS, rfft::rows_lds_bytes(p->W), st>>>(image, spec, rows, p->H, p->W, p->st_rows, p->root_w);
  rfft::cols_kernel<<<dim3(rfft::tiles_of(wc), 3), rfft::COL_THREADS, rfft::cols_lds_bytes(p->H), st>>>(spec, p->H, wc, p->st_cols, p->root_h, -1.f, nullptr, 0, nullptr, nullptr);
}
static void own_inverse_add(const segs_freq_plan* p, float2* spec, float* dL, hipStream_t st, int npartial, float* freq_loss_out, float* loss_inout) {
  const int rows = 3 * p->H, wc = p->W / 2 + 1;
  rfft::cols_kernel<<<dim3(rfft::tiles_of(wc), 3), rfft::COL_THREADS, rfft::cols_lds_bytes(p->H), st>>>(spec, p->H, wc, p->st_cols, p->root_h, +1.f, p->partial, npartial, freq_loss_out, loss_inout);
  rfft::rows_c2r_add_kernel<<<(rows + rfft::ROWS_PER_WG - 1) / rfft::ROWS_PER_WG, rfft::ROW_THREADS, rfft::rows_lds_bytes(p->W), st>>>(spec, dL, rows, p->H, p->W, p->st_rows, p->root_w);
}

// down-scaled copies of `image` for the generic path; returns the per-level source pointers in src[]
static int plan_pyramid(segs_freq_plan* p, const float* image, const float* src[MAXL], hipStream_t st) {
  float* outs[MAXL];
  for (int l = 0; l < p->n; l++) { outs[l] = p->level[l]; src[l] = p->level[l] ? p->level[l] : image; }
  return segs_freq_pyramid(image, 3, p->H, p->W, p->n, p->h, p->w, outs, st);
}

int segs_freq_target(segs_freq_plan* p, const float* gt, float* target_out, void* stream) {
  if (!p || !gt || !target_out) return bad("segs_freq_target: null argument");
  const HipFft& f = hipfft();
  hipStream_t st = (hipStream_t)stream;
  if (p->folded) {
    if (p->own_fft) own_forward(p, gt, reinterpret_cast<float2*>(p->spec[0]), st);
    else if (f.SetStream(p->r2c[0], st) != HIPFFT_SUCCESS ||
        f.ExecR2C(p->r2c[0], const_cast<float*>(gt), reinterpret_cast<hipfftComplex*>(p->spec[0])) != HIPFFT_SUCCESS)
      return fft_fail("segs_freq_target: hipfftExecR2C failed");
    const unsigned nthr = fold_threads(p);
    freq_fold_kernel<true><<<(nthr + 255) / 256, 256, 0, st>>>(p->H, p->W, reinterpret_cast<const float2*>(p->spec[0]), nullptr,
                                                               target_out + p->toff[0], target_out + p->toff[1], target_out + p->toff[2],
                                                               0.f, 0.f, 0.f, nullptr, p->own_fft);
  } else {
    const float* src[MAXL];
    if (int rc = plan_pyramid(p, gt, src, st)) return rc;
    for (int l = 0; l < p->n; l++) {
      if (f.SetStream(p->r2c[l], st) != HIPFFT_SUCCESS ||
          f.ExecR2C(p->r2c[l], const_cast<float*>(src[l]), reinterpret_cast<hipfftComplex*>(p->spec[l])) != HIPFFT_SUCCESS)
        return fft_fail("segs_freq_target: hipfftExecR2C failed");
      if (int rc = segs_spectrum_magnitude(p->spec[l], p->toff[l + 1] - p->toff[l], target_out + p->toff[l], st)) return rc;
    }
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

int segs_freq_loss(segs_freq_plan* p, const float* image, const float* target, float* dL_inout, float* freq_loss_out,
                   float* loss_inout, void* stream) {
  if (!p || !image || !target || !dL_inout || !freq_loss_out) return bad("segs_freq_loss: null argument");
  const HipFft& f = hipfft();
  hipStream_t st = (hipStream_t)stream;
  const size_t npix = (size_t)3 * p->H * p->W;
  if (p->folded) {
    if (p->own_fft) own_forward(p, image, reinterpret_cast<float2*>(p->spec[0]), st);
    else {
    if (f.SetStream(p->r2c[0], st) != HIPFFT_SUCCESS || f.SetStream(p->c2r[0], st) != HIPFFT_SUCCESS) return fft_fail("segs_freq_loss: hipfftSetStream failed");
    if (f.ExecR2C(p->r2c[0], const_cast<float*>(image), reinterpret_cast<hipfftComplex*>(p->spec[0])) != HIPFFT_SUCCESS)
      return fft_fail("segs_freq_loss: hipfftExecR2C failed");
    }
    const unsigned nthr = fold_threads(p), nblk = (nthr + 255) / 256;
    freq_fold_kernel<false><<<nblk, 256, 0, st>>>(p->H, p->W, reinterpret_cast<const float2*>(p->spec[0]), reinterpret_cast<float2*>(p->spec[1]),
                                                  const_cast<float*>(target + p->toff[0]), const_cast<float*>(target + p->toff[1]),
                                                  const_cast<float*>(target + p->toff[2]), p->weight[0], p->weight[1], p->weight[2], p->partial, p->own_fft);
    if (p->own_fft) {
      // (the column pass folds the loss partials on its way, the row pass adds into dL/dimage itself)
      own_inverse_add(p, reinterpret_cast<float2*>(p->spec[1]), dL_inout, st, (int)nblk, freq_loss_out, loss_inout);
    } else {
      freq_finish_kernel<<<1, 1024, 0, st>>>(p->partial, (int)nblk, freq_loss_out, loss_inout);
      if (f.ExecC2R(p->c2r[0], reinterpret_cast<hipfftComplex*>(p->spec[1]), p->grad[0]) != HIPFFT_SUCCESS)
        return fft_fail("segs_freq_loss: hipfftExecC2R failed");
      add_kernel<<<(unsigned)((npix + 255) / 256), 256, 0, st>>>(dL_inout, p->grad[0], npix);
    }
  } else {
    const float* src[MAXL];
    if (int rc = plan_pyramid(p, image, src, st)) return rc;
    float* specs[MAXL];
    const float* tm[MAXL];
    const float* grads[MAXL];
    for (int l = 0; l < p->n; l++) {
      if (f.SetStream(p->r2c[l], st) != HIPFFT_SUCCESS || f.SetStream(p->c2r[l], st) != HIPFFT_SUCCESS) return fft_fail("segs_freq_loss: hipfftSetStream failed");
      if (f.ExecR2C(p->r2c[l], const_cast<float*>(src[l]), reinterpret_cast<hipfftComplex*>(p->spec[l])) != HIPFFT_SUCCESS)
        return fft_fail("segs_freq_loss: hipfftExecR2C failed");
      specs[l] = p->spec[l]; tm[l] = target + p->toff[l]; grads[l] = p->grad[l];
    }
    if (int rc = segs_freq_spectrum_loss(3, p->n, p->h, p->w, specs, tm, p->weight, freq_loss_out, loss_inout,
                                         reinterpret_cast<char*>(p->partial), st)) return rc;
    for (int l = 0; l < p->n; l++)
      if (f.ExecC2R(p->c2r[l], reinterpret_cast<hipfftComplex*>(p->spec[l]), p->grad[l]) != HIPFFT_SUCCESS)
        return fft_fail("segs_freq_loss: hipfftExecC2R failed");
    if (int rc = segs_freq_pyramid_backward_add(dL_inout, 3, p->H, p->W, p->n, p->h, p->w, grads, st)) return rc;
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

}  // extern "C"
