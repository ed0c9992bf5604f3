// Reference point for the two sorts of the binning stage: rocPRIM's device radix sort (onesweep) on the same sizes.
// Measurement aid only -- the product path does not link rocPRIM.   hipcc --offload-arch=gfx950 -O3 tools/ubench_rocprim_sort.hip -o tools/ubench_rocprim_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
static void run(size_t n, unsigned bits, const char* what) {
  std::vector<unsigned> hk(n), hv(n);
  std::mt19937 rng(1);
  for (size_t i = 0; i < n; i++) { hk[i] = rng() & ((1u << bits) - 1u); hv[i] = (unsigned)i; }
  unsigned *k0, *k1, *v0, *v1;
  hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
  size_t tb = 0;
  rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, n, 0, bits);
  void* tmp; hipMalloc(&tmp, tb);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 5; w++) rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, bits);
  hipDeviceSynchronize();
  const int reps = 50;
  hipEventRecord(a);
  for (int r = 0; r < reps; r++) rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, bits);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  printf("%-44s n=%zu bits=%u: %.1f us per sort (temp %zu KB)\n", what, n, bits, 1e3 * ms / reps, tb / 1024);
  hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
}
int main() {
  run(500000, 27, "depth sort (P Gaussians)");
  run(500000, 32, "depth sort, all 32 bits");
  run(2600335, 13, "tile-id sort (instances binned)");
  run(3744721, 13, "tile-id sort (reference R)");
  run(10472370, 13, "tile-id sort (1080p_3m R)");
  return 0;
}
