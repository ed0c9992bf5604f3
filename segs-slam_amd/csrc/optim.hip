// optim.hip -- fused multi-segment Adam over the flat Gaussian-parameter bucket (include/segs_train.h).
// Reference: torch::optim::Adam as configured in src/gaussian_model.cpp:620-872 and stepped at
// src/gaussian_mapper.cpp:1027-1030 / src/gaussian_trainer.cpp:115-116 (LibTorch 2.0.1 arithmetic).
// HBM-bound streaming kernel: 28 B per parameter, 16 B/lane vector accesses, grid-stride.
// Built with -ffp-contract=off so the update is bit-reproducible against the float32 restatement in the tests.
#include <hip/hip_runtime.h>
#include <cmath>
#include <string>
#include "../../include/segs_raster.h"
#include "kernels.h"
#include "../../include/segs_train.h"

#pragma clang fp contract(off)

namespace {
constexpr int MAX_SEG = 16;
constexpr int VEC_PER_THREAD = 4;                       // float4 accesses in flight per thread and tensor
constexpr int CHUNK_VEC = 256 * VEC_PER_THREAD;         // float4 vectors per workgroup (4096 parameters)
struct SegTable {
  long long offset[MAX_SEG];
  long long count[MAX_SEG];
  float step_size[MAX_SEG];  // lr / (1 - b1^t)
  double lr[MAX_SEG];        // device-side step count (segs_adam_step_device): step_size is formed in the kernel
  int block_start[MAX_SEG + 1];  // first workgroup of each segment: segments run side by side, not one after the other
  int nseg;
};

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float b1, float b2, float omb1, float omb2,
                                         float sqrt_bc2, float eps, float step_size, float gscale) {
  const float gr = g * gscale;
  m = m * b1 + gr * omb1;
  v = v * b2 + gr * gr * omb2;
  const float denom = sqrtf(v) / sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
}

// One workgroup = one chunk of 1024 aligned float4 vectors of ONE segment (plus, for the segment's first workgroup, its
// <= 3 unaligned head and <= 3 tail elements).  The first version looped over the segments inside every thread: the ten
// small MLP segments then cost one dependent global round trip each (27 us for 3.6 M parameters, 47 % of the copy rate).
// DEVICE_STEP: the step count t lives on the device (two int64 words used in turn: this launch reads steps[parity] and its
// first workgroup leaves the count after this call in steps[parity ^ 1], so no workgroup can read a count another one has
// already advanced).  A guarded step that is skipped does not advance it -- torch::optim::Adam only counts the steps it
// takes -- and the host never has to learn whether it was.  The bias corrections are formed like LibTorch's, in double
// (1 - pow(beta, t)), by one lane per workgroup.
template <bool DEVICE_STEP>
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m,
                                                   float* __restrict__ v, SegTable tab, float b1, float b2, float omb1, float omb2,
                                                   float sqrt_bc2, float eps, float gscale, int zero_grad,
                                                   const uint32_t* __restrict__ skip_flag, long long* __restrict__ steps, int parity,
                                                   double beta1, double beta2, const double* __restrict__ device_lr) {
  // guarded step: the gradients of an iteration the resident rasterizer flagged as overflowed are discarded on the
  // device (parameters and moments untouched, gradient bucket cleared) without the host having to look first
  const bool skip = skip_flag != nullptr && *skip_flag != 0u;
  int s = 0;
  while (s + 1 < tab.nseg && (int)blockIdx.x >= tab.block_start[s + 1]) s++;   // wave-uniform, <= 15 steps
  const int b = (int)blockIdx.x - tab.block_start[s];
  const long long off = tab.offset[s], cnt = tab.count[s];
  float ss = tab.step_size[s];
  if (DEVICE_STEP) {
    __shared__ float s_ss, s_sqrt_bc2;
    if (threadIdx.x == 0) {
      // parity < 0 (segs_adam_step_graph): which word of the pair holds the count is itself a device word, steps[2], flipped by
      // adam_flip_kernel behind this launch -- so that the same launch can be replayed from a hipGraph step after step
      if (parity < 0) parity = (int)(steps[2] & 1);
      const long long done = steps[parity];
      const double t = (double)(done + 1);
      const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
      s_ss = (float)((device_lr ? device_lr[s] : tab.lr[s]) / bc1);
      s_sqrt_bc2 = (float)sqrt(bc2);
      if (blockIdx.x == 0) steps[parity ^ 1] = skip ? done : done + 1;
    }
    __syncthreads();
    ss = s_ss;
    sqrt_bc2 = s_sqrt_bc2;
  }
  const long long head = min(cnt, (long long)((4 - (off & 3)) & 3));
  const long long nvec = (cnt - head) / 4;
  const long long a0 = off + head;                       // first 16-byte aligned element of the segment
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  long long e[VEC_PER_THREAD];
  bool ok[VEC_PER_THREAD];
  float4 p4[VEC_PER_THREAD], g4[VEC_PER_THREAD], m4[VEC_PER_THREAD], v4[VEC_PER_THREAD];
#pragma unroll
  for (int u = 0; u < VEC_PER_THREAD; u++) {
    const long long i = (long long)b * CHUNK_VEC + u * 256 + threadIdx.x;
    ok[u] = i < nvec;
    e[u] = a0 + 4 * (ok[u] ? i : 0);
  }
  if (!skip) {
#pragma unroll
    for (int u = 0; u < VEC_PER_THREAD; u++) {
      if (!ok[u]) continue;
      p4[u] = *reinterpret_cast<float4*>(param + e[u]); g4[u] = *reinterpret_cast<float4*>(grad + e[u]);
      m4[u] = *reinterpret_cast<float4*>(m + e[u]); v4[u] = *reinterpret_cast<float4*>(v + e[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < VEC_PER_THREAD; u++) {
    if (!ok[u]) continue;
    if (!skip) {
      adam_one(p4[u].x, g4[u].x, m4[u].x, v4[u].x, b1, b2, omb1, omb2, sqrt_bc2, eps, ss, gscale);
      adam_one(p4[u].y, g4[u].y, m4[u].y, v4[u].y, b1, b2, omb1, omb2, sqrt_bc2, eps, ss, gscale);
      adam_one(p4[u].z, g4[u].z, m4[u].z, v4[u].z, b1, b2, omb1, omb2, sqrt_bc2, eps, ss, gscale);
      adam_one(p4[u].w, g4[u].w, m4[u].w, v4[u].w, b1, b2, omb1, omb2, sqrt_bc2, eps, ss, gscale);
      *reinterpret_cast<float4*>(param + e[u]) = p4[u];
      *reinterpret_cast<float4*>(m + e[u]) = m4[u];
      *reinterpret_cast<float4*>(v + e[u]) = v4[u];
    }
    if (zero_grad) *reinterpret_cast<float4*>(grad + e[u]) = zero4;
  }
  if (b == 0) {   // the segment's unaligned head and tail, scalar
    const long long tail0 = head + 4 * nvec, nscal = head + (cnt - tail0);
    if ((long long)threadIdx.x < nscal) {
      const long long i = threadIdx.x;
      const long long es = off + (i < head ? i : tail0 + (i - head));
      if (!skip) {
        float p1 = param[es], g1 = grad[es], m1 = m[es], v1 = v[es];
        adam_one(p1, g1, m1, v1, b1, b2, omb1, omb2, sqrt_bc2, eps, ss, gscale);
        param[es] = p1; m[es] = m1; v[es] = v1;
      }
      if (zero_grad) grad[es] = 0.f;
    }
  }
}

__global__ void adam_flip_kernel(long long* steps) { steps[2] += 1; }
template <int N>
struct Doubles { double v[N]; };
__global__ void set_doubles_kernel(double* dst, Doubles<16> vals, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}
}  // namespace

namespace {
int adam_launch(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const segs_adam_segment* segments, int nseg,
                double beta1, double beta2, double eps, int64_t step, float grad_scale, int zero_grad, const uint32_t* skip_flag,
                long long* device_steps, int parity, void* stream, const double* device_lr = nullptr) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !segments || nseg <= 0 || nseg > MAX_SEG || (!device_steps && step <= 0))
    return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  SegTable tab{};
  tab.nseg = nseg;
  long long total = 0, blocks = 0;
  // bias corrections in double on the host, like LibTorch (1 - std::pow(beta, step)); with a device-side step count the
  // kernel forms them itself
  const double bc1 = device_steps ? 1.0 : 1.0 - std::pow(beta1, (double)step);
  const double bc2 = device_steps ? 1.0 : 1.0 - std::pow(beta2, (double)step);
  for (int i = 0; i < nseg; i++) {
    tab.lr[i] = segments[i].lr;
    if (segments[i].offset < 0 || segments[i].count < 0) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
    tab.offset[i] = segments[i].offset;
    tab.count[i] = segments[i].count;
    tab.step_size[i] = (float)(segments[i].lr / bc1);
    total += segments[i].count;
    tab.block_start[i] = (int)blocks;
    // >= 1 workgroup per non-empty segment (its first one also takes the unaligned head/tail); an empty segment gets none
    if (segments[i].count > 0) blocks += (segments[i].count / 4 + CHUNK_VEC - 1) / CHUNK_VEC + (segments[i].count < 4 ? 1 : 0);
    if (blocks > 0x7FFFFFFF) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "Adam bucket too large for one launch");
  }
  if (device_steps && blocks == 0) blocks = 1;   // nothing to update (an empty shard): one workgroup still advances the count
  tab.block_start[nseg] = (int)blocks;
  if (total == 0 && !device_steps) return SEGS_OK;
  const float sqrt_bc2 = (float)std::sqrt(bc2);
  // Scalars cross into the float32 tensor arithmetic the way LibTorch's do: computed in double, rounded once
  if (device_steps)
    adam_kernel<true><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, tab, (float)beta1, (float)beta2,
                                                                    (float)(1.0 - beta1), (float)(1.0 - beta2), sqrt_bc2, (float)eps,
                                                                    grad_scale, zero_grad, skip_flag, device_steps, parity < 0 ? -1 : (parity & 1), beta1, beta2,
                                                                    device_lr);
  else
    adam_kernel<false><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, tab, (float)beta1, (float)beta2,
                                                                     (float)(1.0 - beta1), (float)(1.0 - beta2), sqrt_bc2, (float)eps,
                                                                     grad_scale, zero_grad, skip_flag, nullptr, 0, beta1, beta2, nullptr);
  if (device_steps && parity < 0) adam_flip_kernel<<<1, 1, 0, (hipStream_t)stream>>>(device_steps);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}
}  // namespace

extern "C" int segs_adam_step_guarded(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const segs_adam_segment* segments,
                                      int nseg, double beta1, double beta2, double eps, int64_t step, float grad_scale, int zero_grad,
                                      const uint32_t* skip_flag, void* stream) {
  return adam_launch(param, grad, exp_avg, exp_avg_sq, segments, nseg, beta1, beta2, eps, step, grad_scale, zero_grad, skip_flag,
                     nullptr, 0, stream);
}

extern "C" int segs_adam_step_device(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const segs_adam_segment* segments,
                                     int nseg, double beta1, double beta2, double eps, int64_t* device_steps, int call_index,
                                     float grad_scale, int zero_grad, const uint32_t* skip_flag, void* stream) {
  if (!device_steps) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "null device step state");
  return adam_launch(param, grad, exp_avg, exp_avg_sq, segments, nseg, beta1, beta2, eps, 0, grad_scale, zero_grad, skip_flag,
                     (long long*)device_steps, call_index, stream);
}

extern "C" int segs_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const segs_adam_segment* segments,
                              int nseg, double beta1, double beta2, double eps, int64_t step, float grad_scale, int zero_grad,
                              void* stream) {
  return segs_adam_step_guarded(param, grad, exp_avg, exp_avg_sq, segments, nseg, beta1, beta2, eps, step, grad_scale, zero_grad,
                                nullptr, stream);
}

extern "C" int segs_adam_step_graph(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const segs_adam_segment* segments,
                                    int nseg, const double* device_lr, double beta1, double beta2, double eps, int64_t* device_steps3,
                                    float grad_scale, int zero_grad, const uint32_t* skip_flag, void* stream) {
  if (!device_steps3 || !device_lr) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "null device step state / learning-rate table");
  return adam_launch(param, grad, exp_avg, exp_avg_sq, segments, nseg, beta1, beta2, eps, 0, grad_scale, zero_grad, skip_flag,
                     (long long*)device_steps3, -1, stream, device_lr);
}

extern "C" int segs_set_doubles(double* device_dst, const double* host_values, int n, void* stream) {
  if (!device_dst || !host_values || n < 0 || n > 16) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "segs_set_doubles: null pointer or n > 16");
  if (n == 0) return SEGS_OK;
  Doubles<16> vals{};
  for (int i = 0; i < n; i++) vals.v[i] = host_values[i];
  set_doubles_kernel<<<1, 64, 0, (hipStream_t)stream>>>(device_dst, vals, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}
