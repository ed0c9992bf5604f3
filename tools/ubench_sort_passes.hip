// Per-kernel timing of the two device sorts of the binning stage on synthetic keys of the 1080p_3m sizes (measurement
// only; includes csrc/binning.hip directly so that ABLATE_* variants can be compiled):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -DSEGS_MEASURE [-DABLATE_SCATTER_LINEAR] tools/ubench_sort_passes.hip -o /tmp/ubench_sort
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
namespace segs { int set_error(int, const char*) { return 0; } int set_hip_error(hipError_t, const char*) { return 0; } }
#include "../segs-slam_amd/csrc/binning.hip"
using namespace segs;

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <int BITS>
void run(const char* name, const std::vector<uint32_t>& keys, int end_bit, uint32_t dmin, int dbits, bool iota) {
  const int n = (int)keys.size();
  BinningLayout L = binning_layout(n);
  char* bin; CK(hipMalloc(&bin, L.total));
  CK(hipMemset(bin, 0, L.total));
  const int passes = (end_bit + BITS - 1) / BITS;
  int side0 = passes & 1;
  std::vector<uint32_t> vals(n); for (int i = 0; i < n; i++) vals[i] = i;
  hipEvent_t ev[64]; for (auto& evt : ev) CK(hipEventCreate(&evt));
  std::vector<double> tc(passes, 0), ts(passes, 0), tx(passes, 0);
  const int reps = 20;
  const int chunk_tiles = getenv("SEGS_COUNT_CHUNK") ? atoi(getenv("SEGS_COUNT_CHUNK")) : SORT_COUNT_CHUNK_TILES;
  const int nchunks = (L.nblocks + chunk_tiles - 1) / chunk_tiles;
  uint32_t* tile_prefix = (uint32_t*)(bin + L.tile_prefix); uint32_t* chunk_hist = (uint32_t*)(bin + L.chunk_hist);
  uint32_t* digit_totals = (uint32_t*)(bin + L.digit_totals);
  for (int rep = 0; rep < reps + 2; rep++) {
    CK(hipMemcpy(bin + L.keys[side0], keys.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bin + L.vals[side0], vals.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    int side = side0, e = 0;
    for (int p = 0; p < passes; p++) {
      const uint32_t* kin = (const uint32_t*)(bin + L.keys[side]); const uint32_t* vin = (const uint32_t*)(bin + L.vals[side]);
      uint32_t* kout = (uint32_t*)(bin + L.keys[side ^ 1]); uint32_t* vout = (uint32_t*)(bin + L.vals[side ^ 1]);
      const int shift = BITS * p, nbits = std::min(BITS, end_bit - shift);
      CK(hipEventRecord(ev[e++]));
      radix_count_kernel<uint32_t, BITS><<<(nchunks + 7) / 8 * 8, SORT_THREADS>>>(kin, n, shift, dmin, dbits, tile_prefix, chunk_hist, L.nblocks, nchunks, nullptr, 0, chunk_tiles, nbits);
      CK(hipEventRecord(ev[e++]));
      radix_scan_kernel<<<1 << BITS, 256>>>(chunk_hist, nchunks, digit_totals);
      CK(hipEventRecord(ev[e++]));
      radix_scatter_kernel<uint32_t, BITS, false><<<(L.nblocks + 7) / 8 * 8, SORT_THREADS>>>(kin, (iota && p == 0) ? nullptr : vin, kout, vout, n, shift, dmin, dbits, tile_prefix,
                                                                          chunk_hist, digit_totals, L.nblocks, nchunks, nullptr, 0, nullptr, nullptr, nullptr, nbits, chunk_tiles, 0,
                                                                          nullptr, nullptr, nullptr, 1);
      CK(hipEventRecord(ev[e++]));
      side ^= 1;
    }
    CK(hipDeviceSynchronize());
    if (rep >= 2) for (int p = 0; p < passes; p++) {
      float a, b, c; CK(hipEventElapsedTime(&a, ev[4 * p], ev[4 * p + 1])); CK(hipEventElapsedTime(&b, ev[4 * p + 1], ev[4 * p + 2]));
      CK(hipEventElapsedTime(&c, ev[4 * p + 2], ev[4 * p + 3]));
      tc[p] += a; ts[p] += b; tx[p] += c;
    }
  }
  // check the result (skipped for ablated builds: they are wrong by construction)
  std::vector<uint32_t> out(n);
  CK(hipMemcpy(out.data(), bin + L.keys[0], (size_t)n * 4, hipMemcpyDeviceToHost));
  bool sorted = true;
  const uint32_t mask = end_bit >= 32 ? 0xFFFFFFFFu : ((1u << end_bit) - 1u);
  for (int i = 1; i < n && sorted; i++) sorted = ((out[i - 1] - dmin) & mask) <= ((out[i] - dmin) & mask);
  double tot = 0;
  printf("%s n=%d bits=%d digit=%d tile=%d: %s\n", name, n, end_bit, BITS, SORT_TILE, sorted ? "sorted" : "NOT SORTED (expected for ablated builds)");
  for (int p = 0; p < passes; p++) {
    printf("  pass %d: count %6.1f us  scan %5.1f us  scatter %6.1f us\n", p, tc[p] / reps * 1e3, ts[p] / reps * 1e3, tx[p] / reps * 1e3);
    tot += (tc[p] + ts[p] + tx[p]) / reps * 1e3;
  }
  printf("  total %.1f us\n", tot);
  CK(hipFree(bin));
}

int main() {
  uint64_t s = 42;
  // depth keys: bits of floats uniform in [1, 6)
  const int P = 2096178;
  std::vector<uint32_t> dk(P);
  uint32_t lo = 0xFFFFFFFFu, hi = 0;
  for (auto& k : dk) { float z = 1.0f + 5.0f * (float)((sm(s) >> 40) * (1.0 / 16777216.0)); memcpy(&k, &z, 4); lo = std::min(lo, k); hi = std::max(hi, k); }
  int dbits = 1; while (((uint64_t)1 << dbits) <= (uint64_t)(hi - lo)) dbits++;
  run<9>("depth sort", dk, dbits, lo, dbits, true);
  // tile ids: each "Gaussian" emits a 1-3 x 1-3 rectangle of tiles of a 120 x 68 grid, row-major
  std::vector<uint32_t> tk; tk.reserve(8000000);
  while (tk.size() < 7900000) {
    const uint32_t r = (uint32_t)sm(s);
    const uint32_t w = 1 + (r & 3) % 3, h = 1 + ((r >> 2) & 3) % 3, x0 = (r >> 4) % (120 - w + 1), y0 = (r >> 12) % (68 - h + 1);
    for (uint32_t y = y0; y < y0 + h; y++) for (uint32_t x = x0; x < x0 + w; x++) tk.push_back(y * 120 + x);
  }
  run<8>("tile-id sort", tk, 13, 0u, 0, false);
  return 0;
}
