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
struct SegTable {
  long long offset[MAX_SEG];
  long long count[MAX_SEG];
  float step_size[MAX_SEG];  // lr / (1 - b1^t)
  int nseg;
};

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float b1, float b2, float omb1, float omb2,
                                         float inv_sqrt_bc2_is_div, float sqrt_bc2, float eps, float step_size, float gscale) {
  (void)inv_sqrt_bc2_is_div;
  const float gr = g * gscale;
  m = m * b1 + gr * omb1;
  v = v * b2 + gr * gr * omb2;
  const float denom = sqrtf(v) / sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m,
                                                   float* __restrict__ v, SegTable tab, float b1, float b2, float omb1, float omb2,
                                                   float sqrt_bc2, float eps, float gscale, int zero_grad,
                                                   const uint32_t* __restrict__ skip_flag) {
  // guarded step: the gradients of an iteration the resident rasterizer flagged as overflowed are discarded on the
  // device (parameters and moments untouched, gradient bucket cleared) without the host having to look first
  const bool skip = skip_flag != nullptr && *skip_flag != 0u;
  for (int s = 0; s < tab.nseg; s++) {
    const long long off = tab.offset[s], cnt = tab.count[s];
    const float ss = tab.step_size[s];
    // vector body on the 16-byte aligned middle, scalar head/tail
    const long long head = min(cnt, (long long)((4 - (off & 3)) & 3));
    const long long nvec = (cnt - head) / 4;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long gsz = (long long)gridDim.x * blockDim.x;
    for (long long i = gtid; i < nvec; i += gsz) {
      const long long e = off + head + 4 * i;
      if (skip) { if (zero_grad) *reinterpret_cast<float4*>(grad + e) = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
      float4 p4 = *reinterpret_cast<float4*>(param + e), g4 = *reinterpret_cast<float4*>(grad + e);
      float4 m4 = *reinterpret_cast<float4*>(m + e), v4 = *reinterpret_cast<float4*>(v + e);
      adam_one(p4.x, g4.x, m4.x, v4.x, b1, b2, omb1, omb2, 0.f, sqrt_bc2, eps, ss, gscale);
      adam_one(p4.y, g4.y, m4.y, v4.y, b1, b2, omb1, omb2, 0.f, sqrt_bc2, eps, ss, gscale);
      adam_one(p4.z, g4.z, m4.z, v4.z, b1, b2, omb1, omb2, 0.f, sqrt_bc2, eps, ss, gscale);
      adam_one(p4.w, g4.w, m4.w, v4.w, b1, b2, omb1, omb2, 0.f, sqrt_bc2, eps, ss, gscale);
      *reinterpret_cast<float4*>(param + e) = p4;
      *reinterpret_cast<float4*>(m + e) = m4;
      *reinterpret_cast<float4*>(v + e) = v4;
      if (zero_grad) *reinterpret_cast<float4*>(grad + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long long tail0 = head + 4 * nvec;
    for (long long i = gtid; i < head + (cnt - tail0); i += gsz) {
      const long long e = off + (i < head ? i : tail0 + (i - head));
      if (skip) { if (zero_grad) grad[e] = 0.f; continue; }
      float p1 = param[e], g1 = grad[e], m1 = m[e], v1 = v[e];
      adam_one(p1, g1, m1, v1, b1, b2, omb1, omb2, 0.f, sqrt_bc2, eps, ss, gscale);
      param[e] = p1; m[e] = m1; v[e] = v1;
      if (zero_grad) grad[e] = 0.f;
    }
  }
}
}  // namespace

extern "C" int segs_adam_step_guarded(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const segs_adam_segment* segments,
                                      int nseg, double beta1, double beta2, double eps, int64_t step, float grad_scale, int zero_grad,
                                      const uint32_t* skip_flag, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !segments || nseg <= 0 || nseg > MAX_SEG || step <= 0) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
  SegTable tab{};
  tab.nseg = nseg;
  long long total = 0;
  // bias corrections in double on the host, like LibTorch (1 - std::pow(beta, step))
  const double bc1 = 1.0 - std::pow(beta1, (double)step);
  const double bc2 = 1.0 - std::pow(beta2, (double)step);
  for (int i = 0; i < nseg; i++) {
    if (segments[i].offset < 0 || segments[i].count < 0) return segs::set_error(SEGS_ERR_INVALID_ARGUMENT, "invalid argument (null pointer or bad size)");
    tab.offset[i] = segments[i].offset;
    tab.count[i] = segments[i].count;
    tab.step_size[i] = (float)(segments[i].lr / bc1);
    total += segments[i].count;
  }
  if (total == 0) return SEGS_OK;
  const float sqrt_bc2 = (float)std::sqrt(bc2);
  long long blocks = (total / 4 + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;  // 8 workgroups per CU, grid-stride the rest
  if (blocks < 1) blocks = 1;
  // Scalars cross into the float32 tensor arithmetic the way LibTorch's do: computed in double, rounded once
  adam_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, tab, (float)beta1, (float)beta2,
                                                            (float)(1.0 - beta1), (float)(1.0 - beta2), sqrt_bc2, (float)eps,
                                                            grad_scale, zero_grad, skip_flag);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEGS_OK : segs::set_hip_error(e, __func__);
}

extern "C" int segs_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const segs_adam_segment* segments,
                              int nseg, double beta1, double beta2, double eps, int64_t step, float grad_scale, int zero_grad,
                              void* stream) {
  return segs_adam_step_guarded(param, grad, exp_avg, exp_avg_sq, segments, nseg, beta1, beta2, eps, step, grad_scale, zero_grad,
                                nullptr, stream);
}
