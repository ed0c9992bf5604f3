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
constexpr int TS = 16;          // output tile
constexpr int HALO = 5;
constexpr int TW = TS + 2 * HALO;  // 26
struct Win { float g[11]; };

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int tid = threadIdx.y * TS + threadIdx.x;
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) ssim_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                                                       Win win, float* __restrict__ Dm, float* __restrict__ D11,
                                                       float* __restrict__ D12, float2* __restrict__ partial /* per workgroup: (sum|d|, sum S) */) {
  __shared__ float s1[TW][TW + 1], s2[TW][TW + 1];
  __shared__ float h[5][TW][TS + 1];
  __shared__ float red[4];
  const int ch = blockIdx.z;
  const size_t plane = (size_t)ch * H * W;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const int tid = threadIdx.y * TS + threadIdx.x;
  for (int i = tid; i < TW * TW; i += 256) {
    const int ly = i / TW, lx = i - ly * TW;
    const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    s1[ly][lx] = in ? img1[plane + (size_t)gy * W + gx] : 0.f;
    s2[ly][lx] = in ? img2[plane + (size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < TW * TS; i += 256) {  // horizontal pass: 26 rows x 16 columns
    const int ly = i / TS, lx = i - ly * TS;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float u = s1[ly][lx + k], v = s2[ly][lx + k], g = win.g[k];
      a += g * u; b += g * v; aa += g * (u * u); bb += g * (v * v); ab += g * (u * v);
    }
    h[0][ly][lx] = a; h[1][ly][lx] = b; h[2][ly][lx] = aa; h[3][ly][lx] = bb; h[4][ly][lx] = ab;
  }
  __syncthreads();
  const int lx = threadIdx.x, ly = threadIdx.y;
  float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
  for (int k = 0; k < 11; k++) {
    const float g = win.g[k];
    mu1 += g * h[0][ly + k][lx]; mu2 += g * h[1][ly + k][lx];
    e11 += g * h[2][ly + k][lx]; e22 += g * h[3][ly + k][lx]; e12 += g * h[4][ly + k][lx];
  }
  const int gx = x0 + lx, gy = y0 + ly;
  float l1 = 0.f, S = 0.f;
  if (gx < W && gy < H) {
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s11 = e11 - mu1_sq, s22 = e22 - mu2_sq, s12 = e12 - mu12;
    const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s11 + s22 + C2;
    const float inv = 1.f / (B1 * B2);
    S = A1 * A2 * inv;
    const float d11 = -S / B2;                 // dS/ds11 = -(A1 A2)/(B1 B2^2)
    const float d12 = 2.f * A1 * inv;          // dS/ds12
    const float dmu = 2.f * mu2 * A2 * inv - 2.f * mu1 * S / B1 - 2.f * mu1 * d11 - mu2 * d12;
    const size_t o = plane + (size_t)gy * W + gx;
    Dm[o] = dmu; D11[o] = d11; D12[o] = d12;
    l1 = fabsf(s1[ly + HALO][lx + HALO] - s2[ly + HALO][lx + HALO]);
  }
  const float t1 = block_sum(l1, red);
  __syncthreads();
  const float t2 = block_sum(S, red);
  // one slot per workgroup (a same-address float atomic from ~10^4 workgroups serialises: 0.25 ms at 1200x680)
  if (tid == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = make_float2(t1, t2);
}

__global__ void __launch_bounds__(256) ssim_bwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                                                       Win win, const float* __restrict__ Dm, const float* __restrict__ D11,
                                                       const float* __restrict__ D12, float w_l1, float w_ssim,
                                                       float* __restrict__ dL) {
  __shared__ float s[3][TW][TW + 1];
  __shared__ float h[3][TW][TS + 1];
  const int ch = blockIdx.z;
  const size_t plane = (size_t)ch * H * W;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const int tid = threadIdx.y * TS + threadIdx.x;
  for (int i = tid; i < TW * TW; i += 256) {
    const int ly = i / TW, lx = i - ly * TW;
    const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t o = plane + (size_t)gy * W + gx;
    s[0][ly][lx] = in ? Dm[o] : 0.f; s[1][ly][lx] = in ? D11[o] : 0.f; s[2][ly][lx] = in ? D12[o] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < TW * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float g = win.g[k];
      a += g * s[0][ly][lx + k]; b += g * s[1][ly][lx + k]; c += g * s[2][ly][lx + k];
    }
    h[0][ly][lx] = a; h[1][ly][lx] = b; h[2][ly][lx] = c;
  }
  __syncthreads();
  const int lx = threadIdx.x, ly = threadIdx.y;
  float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
  for (int k = 0; k < 11; k++) {
    const float g = win.g[k];
    a += g * h[0][ly + k][lx]; b += g * h[1][ly + k][lx]; c += g * h[2][ly + k][lx];
  }
  const int gx = x0 + lx, gy = y0 + ly;
  if (gx < W && gy < H) {
    const size_t o = plane + (size_t)gy * W + gx;
    const float u = img1[o], v = img2[o];
    const float d = u - v;
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);  // torch: d|x|/dx = sign(x), 0 at 0
    dL[o] = w_l1 * sgn - w_ssim * (a + 2.f * u * b + v * c);
  }
}

__global__ void __launch_bounds__(1024) finish_loss_kernel(const float2* __restrict__ partial, int n, float inv_n,
                                                            float lambda_dssim, float* __restrict__ out) {
  __shared__ float r1[16], r2[16];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) { const float2 v = partial[i]; a += v.x; b += v.y; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
  if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = a; r2[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s1 = 0.f, s2 = 0.f;
    for (int w = 0; w < 16; w++) { s1 += r1[w]; s2 += r2[w]; }
    const float l1 = s1 * inv_n, ssim = s2 * inv_n;
    out[0] = (1.f - lambda_dssim) * l1 + lambda_dssim * (1.f - ssim);
    out[1] = l1;
    out[2] = ssim;
  }
}
}  // namespace

extern "C" {

static size_t partial_bytes(int H, int W) {
  const size_t nblk = (size_t)3 * ((W + TS - 1) / TS) * ((H + TS - 1) / TS);
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
  const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, 3), block(TS, TS);
  ssim_fwd_kernel<<<grid, block, 0, st>>>(img1, img2, H, W, win, Dm, D11, D12, partial);
  const float inv_n = 1.0f / (float)plane3;
  finish_loss_kernel<<<1, 1024, 0, st>>>(partial, (int)(grid.x * grid.y * grid.z), inv_n, lambda_dssim, loss_out);
  ssim_bwd_kernel<<<grid, block, 0, st>>>(img1, img2, H, W, win, Dm, D11, D12, (1.f - lambda_dssim) * inv_n, lambda_dssim * inv_n, dL_dimg1);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

}  // extern "C"
