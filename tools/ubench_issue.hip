// Microbenchmark: per-SIMD issue rate of VALU / packed VALU / SALU / v_readlane / v_exp on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_issue ubench_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int n) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f, c = 0.5f;
  unsigned s0 = n, s1 = n + 1, s2 = n + 2, s3 = n + 3;
  for (int i = 0; i < N_ITER; i++) {
    if (MODE == 0) {  // 8 independent v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (MODE == 1) {  // 8 s_add_u32 (SALU)
      asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n"
                   "s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0"
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
    } else if (MODE == 2) {  // 4 v_fma + 4 s_add interleaved
      asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 %4, %4, %5\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 %5, %5, %6\n"
                   "v_fma_f32 %2, %2, %8, %9\n s_add_u32 %6, %6, %7\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 %7, %7, %4"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b), "v"(c));
    } else if (MODE == 3) {  // 8 v_readlane
      asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %6, 7\n v_readlane_b32 %3, %7, 9\n"
                   "v_readlane_b32 %0, %5, 3\n v_readlane_b32 %1, %6, 5\n v_readlane_b32 %2, %7, 7\n v_readlane_b32 %3, %4, 9"
                   : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    } else if (MODE == 4) {  // 8 v_exp_f32
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                   "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 5) {  // 4 v_pk_fma_f32 (8 fma)
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
      asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else if (MODE == 6) {  // 8 v_cndmask with vcc
      asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %3, %3, %0, vcc\n"
                   "v_cmp_gt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f32 vcc, %3, %0\n v_cndmask_b32 %1, %1, %2, vcc"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");
    } else if (MODE == 7) {  // 8 v_add_f32 dpp
      asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 8) {  // 8 v_permlane32_swap
      asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                   "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3);
}
template <int MODE>
void run(const char* name, int blocks_per_cu, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  k<MODE><<<grid, 256>>>(d, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(d, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: each block = 4 waves = 1 per SIMD of its CU; blocks_per_cu waves per SIMD
  double insts_per_simd = (double)blocks_per_cu * N_ITER * 8;
  double ns_per_inst = ms * 1e6 / insts_per_simd;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, blocks_per_cu, ms, ns_per_inst, ns_per_inst * 2.4);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w, d); run<5>("v_pk_fma_f32 (x4 = 8 fma)", w, d); run<1>("s_add_u32", w, d); run<2>("v_fma + s_add interleaved", w, d);
    run<3>("v_readlane_b32", w, d); run<4>("v_exp_f32", w, d); run<6>("v_cmp+v_cndmask", w, d); run<7>("v_add_f32_dpp", w, d); run<8>("v_permlane32_swap", w, d);
  }
  return 0;
}
