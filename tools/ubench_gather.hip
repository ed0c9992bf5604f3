// What does a random narrow gather cost on MI355X?  (measurement only)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_gather.hip -o tools/ubench_gather
// N random records of REC bytes (16 / 32 / 64 / 128, naturally aligned) are gathered from a table of T records that a fill
// kernel has just written (as preprocess_fwd_kernel writes the emit records right before the emitter gathers them in depth
// order), one record per lane, 8 bytes per lane written back coalesced.  The time per gather as a function of REC tells the
// fetch granularity: if 32-byte gathers cost what 128-byte ones do, a gather pulls a whole 128-byte line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

__global__ void fill_kernel(float4* t, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    t[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
template <int Q>   // Q float4 per record
__global__ void __launch_bounds__(256) gather_kernel(const float4* __restrict__ table, const uint32_t* __restrict__ idx, int n, float2* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4* r = table + (size_t)idx[i] * Q;
  float4 acc = r[0];
#pragma unroll
  for (int q = 1; q < Q; q++) { const float4 v = r[q]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  out[i] = make_float2(acc.x + acc.z, acc.y + acc.w);
}

template <int Q>
void run(int T, int N, const std::vector<uint32_t>& idx_h, bool refill) {
  float4* table; uint32_t* idx; float2* out;
  CK(hipMalloc(&table, (size_t)T * Q * 16)); CK(hipMalloc(&idx, (size_t)N * 4)); CK(hipMalloc(&out, (size_t)N * 8));
  CK(hipMemcpy(idx, idx_h.data(), (size_t)N * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  double tot = 0; const int reps = 10;
  for (int rep = 0; rep < reps + 2; rep++) {
    if (refill || rep == 0) fill_kernel<<<2048, 256>>>(table, (size_t)T * Q);
    CK(hipEventRecord(a));
    gather_kernel<Q><<<(N + 255) / 256, 256>>>(table, idx, N, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep >= 2) tot += ms;
  }
  const double us = tot / reps * 1e3;
  printf("record %3d B  table %6.1f MB  %s: %7.1f us for %d gathers = %5.2f ns/gather  -> %6.2f TB/s if 128 B per gather, %6.2f if 64 B, %6.2f on the record bytes\n",
         Q * 16, (double)T * Q * 16 / 1e6, refill ? "table just written" : "table cold      ", us, N, us * 1e3 / N,
         (double)N * 128 / us / 1e6, (double)N * 64 / us / 1e6, (double)N * Q * 16 / us / 1e6);
  CK(hipFree(table)); CK(hipFree(idx)); CK(hipFree(out));
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 3000000, N = argc > 2 ? atoi(argv[2]) : 2100000;
  std::vector<uint32_t> idx(N);
  uint64_t s = 12345;
  for (int i = 0; i < N; i++) idx[i] = (uint32_t)(sm(s) % (uint64_t)T);
  for (int refill = 1; refill >= 0; refill--) {
    run<1>(T, N, idx, refill); run<2>(T, N, idx, refill); run<4>(T, N, idx, refill); run<8>(T, N, idx, refill);
  }
  return 0;
}
