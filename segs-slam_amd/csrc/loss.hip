// loss.hip -- fused L1 + SSIM loss, forward and backward, for the trainer step (include/segs_train.h).
// Reference: loss_utils::l1_loss / ssim (include/loss_utils.h:29-32,51-124) as combined at
// src/gaussian_trainer.cpp:89-90 and src/gaussian_mapper.cpp:924-928:  loss = (1-l) * mean|x1-x2| + l * (1 - mean(SSIM)).
// The reference runs 5 depthwise 11x11 conv2d forward plus their backward through LibTorch (>= 10 image-sized
// round trips; 11 ms at 1080p through MIOpen on this part).  Here: two HBM-streaming kernels, the 11x11 Gaussian
// window applied separably (11+11 taps) out of LDS tiles:
//   ssim_fwd_kernel : x1,x2 -> mu, E[x^2], E[x1 x2] -> SSIM map; emits the three partial-derivative maps
//                     Dm = dS/dmu1 (total), D11 = dS/dE[x1^2], D12 = dS/dE[x1 x2] and the two loss sums;
//   ssim_bwd_kernel : dL/dx1 = (1-l)/N sign(x1-x2) - l/N * ( G*Dm + 2 x1 G*D11 + x2 G*D12 ).
// Zero padding (5) at the image border, window built from integer offsets i-5 like the reference (:56-64).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include "../../include/segs_raster.h"
#include "kernels.h"
#include "../../include/segs_train.h"

namespace {
// Tiling (both kernels): a 256-thread workgroup produces a 32 x 32 output tile from a 42 x 42 input tile (halo 5).
//  * horizontal pass: a thread owns one input row and a PAIR of adjacent output columns, reads its 12 inputs as six
//    ds_read_b64 (16 lanes cover one contiguous 128-B row segment -> no bank conflicts) and forms the products once per
//    input, not once per tap;
//  * vertical pass: a thread owns one column and FOUR adjacent output rows, so the 14 rows it needs are read once for
//    4 x 11 taps (half-waves read contiguous 128-B rows -> no bank conflicts).
// The first version (16 x 16 tiles, one output per thread, odd LDS pitches) spent 43 % of its LDS cycles in bank
// conflicts and 2.3x the VALU instructions on index arithmetic and per-tap products: 45 + 35 us at 1200x680.
// Small images (640x480: 900 tiles of 32 x 32 on 256 CUs) use 32 x 16 tiles instead (template parameter TY).
constexpr int TX = 32;              // output tile width; height TY = 32 or 16
constexpr int HALO = 5;
constexpr int IW = TX + 2 * HALO;   // 42 input columns
constexpr int SP = 44;              // input pitch in floats: even (b64 reads), 44 mod 32 = 12 keeps row pairs apart
struct Win { float g[11]; };

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int tid = threadIdx.x;
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Stage the IH x IW windows of N planes at (y0 - HALO, x0 - HALO) into dst[n][IH][SP], zero outside the image.  All
// N * 12 loads of a thread are issued before the first LDS store: a rolled loop here costs one global round trip per
// iteration (18 in the backward), which at 3 workgroups per CU was most of the kernel time.
template <int N, int IH>
__device__ __forceinline__ void load_tiles(float (*dst)[IH][SP], const float* const (&src)[N], int H, int W, int x0, int y0) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  constexpr int RI = (IH + 7) / 8;   // 6 row rounds
  float v[N][RI][2];
#pragma unroll
  for (int i = 0; i < RI; i++) {
    const int r = ty + 8 * i;
    const int gy = y0 + r - HALO;
    const bool row_in = r < IH && gy >= 0 && gy < H;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int c = tx + 32 * j;
      const int gx = x0 + c - HALO;
      const bool in = row_in && c < IW && gx >= 0 && gx < W;
      const size_t o = in ? (size_t)gy * W + gx : 0;
#pragma unroll
      for (int n = 0; n < N; n++) { const float t = src[n][o]; v[n][i][j] = in ? t : 0.f; }
    }
  }
#pragma unroll
  for (int i = 0; i < RI; i++) {
    const int r = ty + 8 * i;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int c = tx + 32 * j;
      if (r < IH && c < IW) {
#pragma unroll
        for (int n = 0; n < N; n++) dst[n][r][c] = v[n][i][j];
      }
    }
  }
}

template <int TY>
__global__ void __launch_bounds__(256) ssim_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                                                       Win win, float* __restrict__ Dm, float* __restrict__ D11,
                                                       float* __restrict__ D12, float2* __restrict__ partial /* per workgroup: (sum|d|, sum S) */) {
  constexpr int IH = TY + 2 * HALO;   // input rows
  constexpr int RPT = TY / 8;         // output rows per thread in the vertical pass (8 thread rows of 32 columns)
  __shared__ __attribute__((aligned(16))) float sin[2][IH][SP];
  float (*s1)[SP] = sin[0];
  float (*s2)[SP] = sin[1];
  __shared__ __attribute__((aligned(16))) float h[5][IH][TX];
  __shared__ float red[4];
  const int ch = blockIdx.z;
  const size_t plane = (size_t)ch * H * W;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const int tid = threadIdx.x;
  {
    const float* const src[2] = {img1 + plane, img2 + plane};
    load_tiles<2, IH>(sin, src, H, W, x0, y0);
  }
  __syncthreads();
  {  // horizontal pass
    const int p = tid & 15;
    for (int r = tid >> 4; r < IH; r += 16) {
      float u[12], v[12];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const float2 a = *reinterpret_cast<const float2*>(&s1[r][2 * p + 2 * k]);
        const float2 b = *reinterpret_cast<const float2*>(&s2[r][2 * p + 2 * k]);
        u[2 * k] = a.x; u[2 * k + 1] = a.y; v[2 * k] = b.x; v[2 * k + 1] = b.y;
      }
      float uu[12], vv[12], uv[12];
#pragma unroll
      for (int k = 0; k < 12; k++) { uu[k] = u[k] * u[k]; vv[k] = v[k] * v[k]; uv[k] = u[k] * v[k]; }
      float o[5][2];
#pragma unroll
      for (int e = 0; e < 2; e++) {
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
          const float g = win.g[k];
          a += g * u[e + k]; b += g * v[e + k]; aa += g * uu[e + k]; bb += g * vv[e + k]; ab += g * uv[e + k];
        }
        o[0][e] = a; o[1][e] = b; o[2][e] = aa; o[3][e] = bb; o[4][e] = ab;
      }
#pragma unroll
      for (int m = 0; m < 5; m++) *reinterpret_cast<float2*>(&h[m][r][2 * p]) = make_float2(o[m][0], o[m][1]);
    }
  }
  __syncthreads();
  const int lx = tid & 31, ly = (tid >> 5) * RPT;
  float acc[5][RPT];
#pragma unroll
  for (int m = 0; m < 5; m++) {
    float col[RPT + 10];
#pragma unroll
    for (int j = 0; j < RPT + 10; j++) col[j] = h[m][ly + j][lx];
#pragma unroll
    for (int e = 0; e < RPT; e++) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) a += win.g[k] * col[e + k];
      acc[m][e] = a;
    }
  }
  const int gx = x0 + lx;
  float l1 = 0.f, Ssum = 0.f;
#pragma unroll
  for (int e = 0; e < RPT; e++) {
    const int gy = y0 + ly + e;
    if (gx < W && gy < H) {
      const float mu1 = acc[0][e], mu2 = acc[1][e], e11 = acc[2][e], e22 = acc[3][e], e12 = acc[4][e];
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float s11 = e11 - mu1_sq, s22 = e22 - mu2_sq, s12 = e12 - mu12;
      const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s11 + s22 + C2;
      const float inv = 1.f / (B1 * B2);
      const float S = A1 * A2 * inv;
      const float d11 = -S / B2;                 // dS/ds11 = -(A1 A2)/(B1 B2^2)
      const float d12 = 2.f * A1 * inv;          // dS/ds12
      const float dmu = 2.f * mu2 * A2 * inv - 2.f * mu1 * S / B1 - 2.f * mu1 * d11 - mu2 * d12;
      const size_t o = plane + (size_t)gy * W + gx;
      Dm[o] = dmu; D11[o] = d11; D12[o] = d12;
      Ssum += S;
      l1 += fabsf(s1[ly + e + HALO][lx + HALO] - s2[ly + e + HALO][lx + HALO]);
    }
  }
  const float t1 = block_sum(l1, red);
  __syncthreads();
  const float t2 = block_sum(Ssum, red);
  // one slot per workgroup (a same-address float atomic from ~10^4 workgroups serialises: 0.25 ms at 1200x680)
  if (tid == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = make_float2(t1, t2);
}

template <int TY>
__global__ void __launch_bounds__(256) ssim_bwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                                                       Win win, const float* __restrict__ Dm, const float* __restrict__ D11,
                                                       const float* __restrict__ D12, float w_l1, float w_ssim,
                                                       float* __restrict__ dL, const float2* __restrict__ partial, int n_partial,
                                                       float inv_n, float lambda_dssim, float* __restrict__ loss_out) {
  constexpr int IH = TY + 2 * HALO;
  constexpr int RPT = TY / 8;
  __shared__ __attribute__((aligned(16))) float s[3][IH][SP];
  __shared__ __attribute__((aligned(16))) float h[3][IH][TX];
  const int ch = blockIdx.z;
  const size_t plane = (size_t)ch * H * W;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const int tid = threadIdx.x;
  {
    const float* const src[3] = {Dm + plane, D11 + plane, D12 + plane};
    load_tiles<3, IH>(s, src, H, W, x0, y0);
  }
  __syncthreads();
  {
    const int p = tid & 15;
    for (int r = tid >> 4; r < IH; r += 16) {
#pragma unroll
      for (int m = 0; m < 3; m++) {
        float u[12];
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const float2 a = *reinterpret_cast<const float2*>(&s[m][r][2 * p + 2 * k]);
          u[2 * k] = a.x; u[2 * k + 1] = a.y;
        }
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) { o0 += win.g[k] * u[k]; o1 += win.g[k] * u[k + 1]; }
        *reinterpret_cast<float2*>(&h[m][r][2 * p]) = make_float2(o0, o1);
      }
    }
  }
  __syncthreads();
  const int lx = tid & 31, ly = (tid >> 5) * RPT;
  float acc[3][RPT];
#pragma unroll
  for (int m = 0; m < 3; m++) {
    float col[RPT + 10];
#pragma unroll
    for (int j = 0; j < RPT + 10; j++) col[j] = h[m][ly + j][lx];
#pragma unroll
    for (int e = 0; e < RPT; e++) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) a += win.g[k] * col[e + k];
      acc[m][e] = a;
    }
  }
  const int gx = x0 + lx;
#pragma unroll
  for (int e = 0; e < RPT; e++) {
    const int gy = y0 + ly + e;
    if (gx < W && gy < H) {
      const size_t o = plane + (size_t)gy * W + gx;
      const float u = img1[o], v = img2[o];
      const float d = u - v;
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);  // torch: d|x|/dx = sign(x), 0 at 0
      dL[o] = w_l1 * sgn - w_ssim * (acc[0][e] + 2.f * u * acc[1][e] + v * acc[2][e]);
    }
  }
  // The loss value itself: the forward kernel's per-tile sums, folded by ONE workgroup of this kernel in a fixed order (a
  // launch of its own, finish_loss_kernel, cost 5 us of latency per step for 20 KB of reads; nothing on the device waits for
  // the value, and here it rides under thousands of other workgroups).
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    __shared__ float r1[4], r2[4];
    float a = 0.f, b = 0.f;
    for (int i = tid; i < n_partial; i += 256) { const float2 v = partial[i]; a += v.x; b += v.y; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    __syncthreads();
    if ((tid & 63) == 0) { r1[tid >> 6] = a; r2[tid >> 6] = b; }
    __syncthreads();
    if (tid == 0) {
      const float l1 = ((r1[0] + r1[1]) + (r1[2] + r1[3])) * inv_n, ssim = ((r2[0] + r2[1]) + (r2[2] + r2[3])) * inv_n;
      loss_out[0] = (1.f - lambda_dssim) * l1 + lambda_dssim * (1.f - ssim);
      loss_out[1] = l1;
      loss_out[2] = ssim;
    }
  }
}

}  // namespace

extern "C" {

static int tile_rows(int H, int W) {   // 32-row tiles once they fill the chip about twice over
  return (size_t)3 * ((W + TX - 1) / TX) * ((H + 31) / 32) >= 1536 ? 32 : 16;
}
static size_t partial_bytes(int H, int W) {
  const size_t nblk = (size_t)3 * ((W + TX - 1) / TX) * ((H + 15) / 16);   // sized for the smaller tile
  return (nblk * sizeof(float2) + 255) & ~(size_t)255;
}
size_t segs_l1_ssim_temp_bytes(int H, int W) { return (size_t)3 * 3 * H * W * sizeof(float) + partial_bytes(H, W); }

int segs_l1_ssim_loss(const float* img1, const float* img2, int H, int W, float lambda_dssim, float* loss_out, float* dL_dimg1,
                      char* temp, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!img1 || !img2 || !loss_out || !dL_dimg1 || !temp || H <= 0 || W <= 0) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  Win win;
  float sum = 0.f;
  for (int x = 0; x < 11; ++x) {  // loss_utils.h:51-64: integer offsets, float arithmetic
    const int t = x - 11 / 2;
    win.g[x] = std::exp(-(float)(t * t) / (2.0f * 1.5f * 1.5f));
    sum += win.g[x];
  }
  for (int x = 0; x < 11; ++x) win.g[x] /= sum;
  const size_t plane3 = (size_t)3 * H * W;
  float2* partial = reinterpret_cast<float2*>(temp);
  float* Dm = reinterpret_cast<float*>(temp + partial_bytes(H, W));
  float* D11 = Dm + plane3;
  float* D12 = D11 + plane3;
  const int ty = tile_rows(H, W);
  const dim3 grid((W + TX - 1) / TX, (H + ty - 1) / ty, 3), block(256);
  if (ty == 32) ssim_fwd_kernel<32><<<grid, block, 0, st>>>(img1, img2, H, W, win, Dm, D11, D12, partial);
  else ssim_fwd_kernel<16><<<grid, block, 0, st>>>(img1, img2, H, W, win, Dm, D11, D12, partial);
  const float inv_n = 1.0f / (float)plane3;
  const int n_partial = (int)(grid.x * grid.y * grid.z);
  if (ty == 32) ssim_bwd_kernel<32><<<grid, block, 0, st>>>(img1, img2, H, W, win, Dm, D11, D12, (1.f - lambda_dssim) * inv_n, lambda_dssim * inv_n, dL_dimg1,
                                                            partial, n_partial, inv_n, lambda_dssim, loss_out);
  else ssim_bwd_kernel<16><<<grid, block, 0, st>>>(img1, img2, H, W, win, Dm, D11, D12, (1.f - lambda_dssim) * inv_n, lambda_dssim * inv_n, dL_dimg1,
                                                   partial, n_partial, inv_n, lambda_dssim, loss_out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

}  // extern "C"
