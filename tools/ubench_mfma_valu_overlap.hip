// tools/ubench_mfma_valu_overlap.hip -- does v_mfma_f32_16x16x4_f32 run BESIDE another wave's vector instructions, or in their
// place?  Two waves per SIMD: wave A issues a chain of f32 MFMAs, wave B a chain of independent v_fma_f32.  Timed alone and
// together: "together = max" means two pipes, "together = sum" means the f32 MFMA occupies the vector ALU (round 5: what decides
// whether render_bwd_kernel's moment sums can move to the matrix pipe).  The same with the bf16 form for comparison.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_valu_overlap.hip -o /tmp/ubench_overlap && /tmp/ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// ROLE_BY = 0: waves 0-3 issue MFMAs, 4-7 FMAs; 1: even waves MFMAs, odd waves FMAs.  Whatever the wave -> SIMD placement
// (w % 4 or w / 2), one of the two puts an MFMA wave and an FMA wave on every SIMD and the other puts two waves of one kind
// there (which always costs the sum): overlap exists iff ONE of the two assignments runs in max(alone, alone).
template <int KIND, int ROLE_BY>   // KIND 0: f32 16x16x4, two independent accumulators; 1: bf16 16x16x32, two; 2: f32 16x16x4, ONE dependent chain; 3: as 0 with the FMA waves at s_setprio 3; 4: as 0 with the MFMA waves the YOUNGER ones;
// 5 / 6 / 7: ONE chain with one / two / three s_nop 15 behind every MFMA (the wave presents its next MFMA to the vector issue port only
// about when the pipe is free again)
__global__ void __launch_bounds__(512) overlap_kernel(float* out, int mfma_iters, int valu_iters, float a, float b) {
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  if (KIND == 4 ? (ROLE_BY == 0 ? wave >= 4 : (wave & 1) == 1) : (ROLE_BY == 0 ? wave < 4 : (wave & 1) == 0)) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; i++) { ha[i] = (short)(threadIdx.x + i); hb[i] = (short)(threadIdx.x * 3 + i); }
    for (int i = 0; i < mfma_iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (KIND == 0 || KIND == 3 || KIND == 4) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
        } else if (KIND >= 5) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
          asm volatile("s_nop 15" : "+v"(acc0)); if (KIND >= 6) asm volatile("s_nop 15" : "+v"(acc0)); if (KIND >= 7) asm volatile("s_nop 15" : "+v"(acc0));
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc0, 0, 0, 0);
          asm volatile("s_nop 15" : "+v"(acc0)); if (KIND >= 6) asm volatile("s_nop 15" : "+v"(acc0)); if (KIND >= 7) asm volatile("s_nop 15" : "+v"(acc0));
        } else if (KIND == 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc0, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb, ha, acc1, 0, 0, 0);
        }
      }
    }
    s = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  } else {
    if (KIND == 3) __builtin_amdgcn_s_setprio(3);
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = (float)(threadIdx.x + i);
    for (int i = 0; i < valu_iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = __builtin_fmaf(v[u], a, b);   // 8 independent chains: issue-bound
    }
    for (int i = 0; i < 8; i++) s += v[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int ROLE_BY>
float run(float* out, int mi, int vi) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  overlap_kernel<KIND, ROLE_BY><<<256, 512>>>(out, mi ? 10 : 0, vi ? 10 : 0, 1.0f, 0.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  overlap_kernel<KIND, ROLE_BY><<<256, 512>>>(out, mi, vi, 1.0f, 0.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int KIND, int ROLE_BY>
void report(float* out, const char* name) {
  const int MI = 20000;                         // 16 MFMAs per iteration
  const float m = run<KIND, ROLE_BY>(out, MI, 0);
  const float v1 = run<KIND, ROLE_BY>(out, 0, 100000);   // size the FMA waves to about the same time alone: 8 FMAs per iteration
  const int VI = (int)(100000.0 * m / v1);
  const float v = run<KIND, ROLE_BY>(out, 0, VI);
  const float both = run<KIND, ROLE_BY>(out, MI, VI);
  printf("%-26s roles by %-9s: MFMA waves alone %.3f ms (%.1f cycles per MFMA at 2.4 GHz) | FMA waves alone %.3f ms (%.2f cycles per "
         "v_fma_f32) | together %.3f ms = %.2f x max, %.2f x sum\n", name, ROLE_BY == 0 ? "wave / 4" : "wave % 2", m,
         m * 1e-3 * 2.4e9 / (MI * 16.0), v, v * 1e-3 * 2.4e9 / (VI * 8.0), both, both / fmaxf(m, v), both / (m + v));
}

int main() {
  float* out;
  if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
  report<0, 0>(out, "v_mfma_f32_16x16x4_f32");
  report<0, 1>(out, "v_mfma_f32_16x16x4_f32");
  report<2, 0>(out, "f32_16x16x4, one chain");
  report<2, 1>(out, "f32_16x16x4, one chain");
  report<3, 0>(out, "f32, FMA waves s_setprio 3");
  report<3, 1>(out, "f32, FMA waves s_setprio 3");
  report<4, 0>(out, "f32, MFMA waves younger");
  report<4, 1>(out, "f32, MFMA waves younger");
  report<5, 0>(out, "f32 + 1 s_nop 15 each");
  report<6, 0>(out, "f32 + 2 s_nop 15 each");
  report<7, 0>(out, "f32 + 3 s_nop 15 each");
  report<1, 0>(out, "v_mfma_f32_16x16x32_bf16");
  report<1, 1>(out, "v_mfma_f32_16x16x32_bf16");
  (void)hipFree(out);
  return 0;
}
