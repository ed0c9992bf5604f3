// tools/ubench_mfma.hip -- what a dependent chain of v_mfma_f32_32x32x2_f32 costs on this part: ns per MFMA with one and two
// waves per SIMD (the neural kernels' chains are dependent accumulations of exactly this instruction), and the same with
// independent accumulators.  hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o /tmp/ubench_mfma && /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) chain_kernel(float* out, int iters, float a, float b) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; c++)
    for (int r = 0; r < 16; r++) acc[c][r] = (float)(threadIdx.x + c);
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; c++)
    for (int r = 0; r < 16; r++) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(const char* what, int waves_per_simd) {
  float* out;
  const int blocks = 256 * waves_per_simd;   // 256 CUs x 4 waves per workgroup = one wave per SIMD per workgroup
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  chain_kernel<CHAINS><<<blocks, 256>>>(out, 10, 1.0f, 0.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  chain_kernel<CHAINS><<<blocks, 256>>>(out, iters, 1.0f, 0.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * 16 * CHAINS;            // MFMAs issued by one wave
  const double ns = ms * 1e6 / (per_wave * waves_per_simd);       // SIMD time per MFMA
  printf("%-40s waves/SIMD %d: %.3f ms, %.1f ns of SIMD time per MFMA (64 cycles at 2.4 GHz = 26.7 ns), %.1f TFLOP/s\n", what,
         waves_per_simd, ms, ns, 4096.0 * per_wave * waves_per_simd * 1024 / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<1>("one dependent chain per wave", 1);
  run<1>("one dependent chain per wave", 2);
  run<2>("two independent chains per wave", 1);
  run<4>("four independent chains per wave", 1);
  run<4>("four independent chains per wave", 2);
  return 0;
}
