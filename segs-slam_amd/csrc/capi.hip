// capi.hip -- host orchestration + extern "C" entry points declared in include/segs_raster.h.
//
// Mirrors CudaRasterizer::Rasterizer::{forward,backward,visible_filter,markVisible,project2_image}
// (cuda_rasterizer/rasterizer_impl.cu:141-153,198-336,339-393,397-490,494-585) as a C ABI.
// Launch order of forward: K1 preprocess -> K5 scan -> (one host sync for R, as the reference's
// cudaMemcpy at :281) -> K7 duplicate -> K8 radix sort -> K9 ranges -> K10 render.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/segs_raster.h"
#include "gs_layout.h"
#include "kernels.h"

using namespace segs;

namespace {
thread_local std::string g_err;   // segs_last_error(): message of the last failing entry point on this host thread
}
// Shared by every translation unit of the library (kernels.h): record the message, return the status code.
int segs::set_error(int code, const char* what) {
  g_err = what;
  return code;
}
int segs::set_hip_error(hipError_t e, const char* where) {
  g_err = std::string(where) + ": " + hipGetErrorString(e);
  return (int)e > 0 ? (int)e : 1;
}

namespace {

thread_local uint32_t g_flags = 0u;   // segs_raster_set_flags
thread_local uint32_t* g_status_mirror = nullptr;   // segs_raster_set_status_mirror

int fail(int code, const char* what) { return segs::set_error(code, what); }
int hip_fail(hipError_t e, const char* where) { return segs::set_hip_error(e, where); }
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) return hip_fail(_e, #expr);   \
  } while (0)
#define LAUNCH_TRY(name)                                \
  do {                                                  \
    hipError_t _e = hipGetLastError();                  \
    if (_e != hipSuccess) return hip_fail(_e, name);    \
  } while (0)

// ---- measurement support: per-kernel HIP-event timing on the launch stream (segs_profile_*) ----
enum KernelId { K_PREPROCESS_FWD = 0, K_SCAN, K_DUPLICATE, K_RADIX_COUNT, K_RADIX_SCAN, K_RADIX_SCATTER, K_RANGES,
                K_RENDER_FWD, K_RENDER_BWD, K_PREPROCESS_BWD, K_MEMSET, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"preprocess_fwd_kernel", "scan_block_sums_kernel", "duplicate_with_keys_kernel",
                                           "radix_count_kernel", "radix_scan_kernel", "radix_scatter_kernel",
                                           "identify_tile_ranges_kernel", "render_fwd_kernel", "render_bwd_kernel",
                                           "preprocess_bwd_kernel", "memset"};
struct Profiler {
  unsigned mask = 0;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  struct Span { int id; size_t e0, e1; };
  std::vector<Span> spans;
  double total_ms[K_COUNT] = {0};
  long count[K_COUNT] = {0};
  hipEvent_t get() {
    if (used == pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
    return pool[used++];
  }
} g_prof;
struct ProfScope {
  int id; hipStream_t st; size_t e0 = 0; bool on;
  ProfScope(int id_, hipStream_t st_) : id(id_), st(st_), on((g_prof.mask >> id_) & 1u) {
    if (on) { e0 = g_prof.used; hipEvent_t e = g_prof.get(); if (e) (void)hipEventRecord(e, st); else on = false; }
  }
  ~ProfScope() {
    if (on) { size_t e1 = g_prof.used; hipEvent_t e = g_prof.get(); if (e) { (void)hipEventRecord(e, st); g_prof.spans.push_back({id, e0, e1}); } }
  }
};
#define PROF(id) ProfScope _prof_scope_##id(id, st)

// rasterizer_impl.cu:35-50
uint32_t getHigherMsb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4, step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

// K8: stable LSD radix sort, BITS (8 or 9) bits per pass over key bits [0,end_bit).  Input is expected in side
// `passes & 1` of the ping-pong pair so that the result lands in side 0.
// drop_dead: entries whose key is all ones are left out by the FIRST pass (they take no histogram count and no rank), so
// every later pass -- and the caller, through *n_live -- works on the survivors only: a stable partition for free.
// iota_vals: the values of the input are 0..n-1 and are not read (nor need they have been written).
// aux_in / aux_final (32-bit keys only): the LAST pass also writes aux_final[position] = aux_in[value] (a gather by the
// sorted values, fused into the scatter).
// Tiles per chunk of the count kernel (a workgroup walks its chunk's tiles one after the other; only the chunk totals go
// through the row scan).  Few tiles: one tile per workgroup -- the launch is latency-bound and a 4-tile walk quadruples
// that latency for nothing (50 k Gaussians at 640x480: 0.206 -> 0.187 ms per step with one tile per chunk).  Many tiles:
// longer chunks keep the row scan short (3 M Gaussians at 1080p: 1.102 / 1.108 / 1.126 ms with 4 / 2 / 1).
int count_chunk_tiles(int nblocks) {
#ifdef SEGS_MEASURE   // measurement builds only (tools/): anything but 1, 2 or 4 tiles per chunk is ignored
  static const int chunk_override = [] { const char* e = getenv("SEGS_COUNT_CHUNK"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();
  if (chunk_override > 0) return chunk_override;
#endif
  return nblocks <= 256 ? 1 : (nblocks <= 2048 ? 2 : SORT_COUNT_CHUNK_TILES);
}
// What the caller of the tile-id sort fuses into its last pass (run_binning).
struct SortFusion {
  uint2* ranges = nullptr;           // non-null: the LAST pass fills the range table (K9) -- only valid with >= 2 passes or a
                                     // single pass whose digit is the whole key (see radix_scatter_kernel)
  uint32_t* status = nullptr;        // resident mode, with `ranges`: the last pass also writes the status words ...
  uint32_t* status_mirror = nullptr; // ... and their host-mapped mirror
  bool keep_sorted_keys = true;      // false: the last pass does not store the sorted keys (nobody reads them)
};
template <typename K, int BITS = 8>
int sort_pairs(char* bin, const BinningLayout& L, int n, int end_bit, uint32_t dmin, int dbits, hipStream_t st,
               const uint32_t* n_dev = nullptr, bool drop_dead = false, bool iota_vals = false, const uint32_t* aux_in = nullptr,
               uint32_t* aux_final = nullptr, int pack_shift = 0 /* > 0 (with iota_vals, aux_in, no aux_final): the FIRST pass packs
               min(aux_in[i], tmax) into the value's bits from pack_shift up; the caller takes the sorted values apart */,
               const SortFusion& fuse = SortFusion(), const K* first_keys = nullptr /* the FIRST pass reads its keys here instead of
               from its ping-pong side (keys produced before the sort scratch existed) */) {
  if (n <= 0) return SEGS_OK;
  uint32_t* n_live = (uint32_t*)(bin + L.n_live);
  const int passes = (end_bit + BITS - 1) / BITS;
  int side = passes & 1;
  uint32_t* tile_prefix = (uint32_t*)(bin + L.tile_prefix);
  uint32_t* chunk_hist = (uint32_t*)(bin + L.chunk_hist);
  uint32_t* digit_totals = (uint32_t*)(bin + L.digit_totals);
  const int chunk_tiles = count_chunk_tiles(L.nblocks);
  const int nchunks = (L.nblocks + chunk_tiles - 1) / chunk_tiles;
  for (int p = 0; p < passes; p++) {
    const K* kin = (p == 0 && first_keys) ? first_keys : (const K*)(bin + L.keys[side]);
    const uint32_t* vin = (const uint32_t*)(bin + L.vals[side]);
    K* kout = (K*)(bin + L.keys[side ^ 1]);
    uint32_t* vout = (uint32_t*)(bin + L.vals[side ^ 1]);
    const int shift = BITS * p;
    const int drop = drop_dead && p == 0;
    const uint32_t* n_in = (drop_dead && p > 0) ? n_live : n_dev;
    const int nbits = std::min(BITS, end_bit - shift);   // the last pass may have fewer significant bits than a full digit
    { PROF(K_RADIX_COUNT);
    radix_count_kernel<K, BITS><<<(nchunks + 7) / 8 * 8, SORT_THREADS, 0, st>>>(kin, n, shift, dmin, dbits, tile_prefix, chunk_hist, L.nblocks, nchunks, n_in, drop, chunk_tiles, nbits);
    }
    LAUNCH_TRY("radix_count_kernel");
    const bool last = p == passes - 1;
    uint2* fr = last ? fuse.ranges : nullptr;
    uint32_t* fs = last ? fuse.status : nullptr;
    uint32_t* fm = last ? fuse.status_mirror : nullptr;
    const int wk = (last && !fuse.keep_sorted_keys) ? 0 : 1;
    { PROF(K_RADIX_SCAN);
    radix_scan_kernel<<<1 << BITS, 256, 0, st>>>(chunk_hist, nchunks, digit_totals);
    }
    LAUNCH_TRY("radix_scan_kernel");
    if (iota_vals && p == 0) vin = nullptr;
    const int scatter_grid = (L.nblocks + 7) / 8 * 8;   // a multiple of the XCD count: see radix_scatter_kernel's tile mapping
    { PROF(K_RADIX_SCATTER);
    if constexpr (sizeof(K) == 4) {
      if (aux_in && aux_final && p == passes - 1) {
        radix_scatter_kernel<K, BITS, true><<<scatter_grid, SORT_THREADS, 0, st>>>(kin, vin, kout, vout, n, shift, dmin, dbits, tile_prefix, chunk_hist,
                                                                        digit_totals, L.nblocks, nchunks, n_in, drop,
                                                                        drop ? n_live : nullptr, aux_in, aux_final, nbits, chunk_tiles, 0, fr, fs, fm, wk);
      } else {
        radix_scatter_kernel<K, BITS, false><<<scatter_grid, SORT_THREADS, 0, st>>>(kin, vin, kout, vout, n, shift, dmin, dbits, tile_prefix, chunk_hist,
                                                                         digit_totals, L.nblocks, nchunks, n_in, drop,
                                                                         drop ? n_live : nullptr, (pack_shift > 0 && p == 0) ? aux_in : nullptr, nullptr,
                                                                         nbits, chunk_tiles, (p == 0) ? pack_shift : 0, fr, fs, fm, wk);
      }
    } else {
      radix_scatter_kernel<K, BITS, false><<<scatter_grid, SORT_THREADS, 0, st>>>(kin, vin, kout, vout, n, shift, dmin, dbits, tile_prefix, chunk_hist,
                                                                       digit_totals, L.nblocks, nchunks, n_in, drop,
                                                                       drop ? n_live : nullptr, nullptr, nullptr, nbits, chunk_tiles, 0, nullptr, nullptr, nullptr, 1);
    }
    }
    LAUNCH_TRY("radix_scatter_kernel");
    side ^= 1;
  }
  return SEGS_OK;
}

struct Geom {
  GeomLayout L;
  char* base;
  float* rec() const { return (float*)(base + L.rec); }
  BinInfo* bin() const { return (BinInfo*)(base + L.bin); }
  uint32_t* offsets() const { return (uint32_t*)(base + L.offsets); }
  int* radii_internal() const { return (int*)(base + L.radii_internal); }
  uint32_t* block_sums() const { return (uint32_t*)(base + L.block_sums); }
  uint32_t* num_rendered() const { return (uint32_t*)(base + L.num_rendered); }
  uint32_t* clamped() const { return (uint32_t*)(base + L.clamped); }
  float* gacc() const { return (float*)(base + L.gacc); }
  uint32_t* touched() const { return (uint32_t*)(base + L.touched); }
};
Geom geom_at(char* p, int P) { return Geom{geom_layout(P), align_ptr(p)}; }
// Resident buffers are carved up for the `rows` they were ALLOCATED (and zero-filled) for, so that the self-cleaned
// accumulator rows stay where they are when the caller rasterizes fewer rows (a map that shrinks inside pre-sized
// buffers); only the launch extent follows P.
Geom geom_at(char* p, int P, int rows) {
  Geom g{geom_layout(rows), align_ptr(p)};
  g.L.P = P;
  g.L.nblocks = (P + 255) / 256;
  return g;
}

int run_preprocess(const Geom& G, int P, int W, int H, const float* means3D, const float* colors, const float* opac,
                   const float* scales, float mod, const float* rots, const float* cov3D_precomp, const float* view,
                   const float* proj, float tan_fovx, float tan_fovy, int* radii, const float* shs, int D, int M,
                   const float* cam_pos, hipStream_t st, uint32_t* depth_keys = nullptr, uint32_t* depth_vals = nullptr,
                   uint2* ranges = nullptr, uint32_t extra_flags = 0u, uint32_t* depth_overflow = nullptr) {
  const float focal_y = H / (2.0f * tan_fovy);   // rasterizer_impl.cu:221-222
  const float focal_x = W / (2.0f * tan_fovx);
  const uint32_t gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
  { PROF(K_PREPROCESS_FWD);
  preprocess_fwd_kernel<<<G.L.nblocks, 256, 0, st>>>(P, means3D, scales, mod, rots, opac, colors, cov3D_precomp, view,
                                                     proj, W, H, tan_fovx, tan_fovy, focal_x, focal_y, gx, gy, radii,
                                                     G.rec(), G.bin(), G.block_sums(), G.block_sums() + (G.L.nblocks + 1), shs, D, M, cam_pos,
                                                     G.clamped(), g_flags | extra_flags, depth_keys, depth_vals, ranges, (int)(gx * gy),
                                                     depth_overflow, G.touched());
  }
  LAUNCH_TRY("preprocess_fwd_kernel");
  return SEGS_OK;
}

// Tile binning (K7, K8, K9).  The reference sorts all R instances on key bits [0, 32+bit) = (tile | depth)
// (rasterizer_impl.cu:300-308).  Same order, far less traffic: (1) stable-sort the P GAUSSIANS by depth (P << R),
// (2) emit instances in that order, (3) stable-sort the R instances by tile id only.  Ties in (tile, depth) keep
// increasing Gaussian index in both formulations, so keys / point_list / ranges are bit-identical.
// `n_cap` is R (host-known) or, in resident mode, the capacity of the instance arrays with the true R in *n_dev.
// `total_out` (3 device words) receives the instance count produced by the depth-ordered scan.
int run_binning(const Geom& G, char* bin, const BinningLayout& BL, const GaussSortLayout& GS, uint2* ranges, int P, int n_cap,
                const uint32_t* n_dev, uint32_t dmin, int dbits, uint32_t dcull, uint32_t gx, uint32_t gy, uint32_t* total_out,
                hipStream_t st, bool depth_keys_ready = false, bool drop_dead = false, bool nine_bit_depth = false,
                const uint32_t* depth_keys_src = nullptr /* with depth_keys_ready: where K1 left the keys, if not in the sort scratch */) {
  const int bit = (int)getHigherMsb(gx * gy);
  // (1)
  char* gbin = bin + GS.base;
  const BinningLayout& GL = GS.inner;
  // depth keys in [dmin, dcull], dcull - dmin < 2^dbits; culled Gaussians carry dcull
  const int gside = (nine_bit_depth ? (dbits + 8) / 9 : (dbits + 7) / 8) & 1;
  if (!depth_keys_ready) { PROF(K_DUPLICATE);
  make_depth_keys_kernel<<<G.L.nblocks, 256, 0, st>>>(P, G.bin(), dcull, (uint32_t*)(gbin + GL.keys[gside]), (uint32_t*)(gbin + GL.vals[gside]),
                                                      ranges, (int)(gx * gy));
  }
  LAUNCH_TRY("make_depth_keys_kernel");
  // resident mode (K1 wrote the keys): culled Gaussians carry the all-ones key and are dropped by the first depth pass
  const bool drop_culled = depth_keys_ready;
  // The values are the Gaussian indices 0..P-1 (never materialised); tiles_touched must follow them into depth order for the
  // emitter's offsets.  It rides in the values' spare high bits, picked up by the FIRST pass (where entry i is Gaussian i: a
  // coalesced read) and taken apart by ordered_block_sums_kernel; a count that does not fit the spare bits saturates and is
  // fetched there.  (As a gather by the sorted values in the last pass it cost 44 instead of 21 us at 3 M Gaussians.)  With
  // fewer than six spare bits (P > 2^26) the last pass gathers it after all.
  int idx_bits = 1;
  while (idx_bits < 32 && ((uint64_t)1 << idx_bits) < (uint64_t)P) idx_bits++;
  int pack_shift = (32 - idx_bits >= 6 && !(g_flags & SEGS_RASTER_GATHER_TILES_TOUCHED)) ? idx_bits : 0;
  if (pack_shift && (g_flags & SEGS_RASTER_TEST_NARROW_PACK)) pack_shift = 30;
  uint32_t* aux_final = pack_shift ? nullptr : G.offsets();
  int rc = nine_bit_depth ? sort_pairs<uint32_t, 9>(gbin, GL, P, dbits, dmin, dbits, st, nullptr, drop_culled, true, G.touched(), aux_final, pack_shift, SortFusion(), depth_keys_src)
                          : sort_pairs<uint32_t>(gbin, GL, P, dbits, dmin, dbits, st, nullptr, drop_culled, true, G.touched(), aux_final, pack_shift, SortFusion(), depth_keys_src);
  if (rc) return rc;
  uint32_t* order = (uint32_t*)(gbin + GL.vals[0]);
  const uint32_t* ng_dev = drop_culled ? (const uint32_t*)(gbin + GL.n_live) : nullptr;
  // (2)
  uint32_t* sums2 = (uint32_t*)(bin + GS.block_sums);
  uint32_t* first_owner = (uint32_t*)(bin + GS.first_owner);
  const int prefix_wgs = (P + 256 * PREFIX_ROWS_PER_WG - 1) / (256 * PREFIX_ROWS_PER_WG);
  { PROF(K_SCAN);
  ordered_block_sums_kernel<<<prefix_wgs, 256, 0, st>>>(P, pack_shift ? G.touched() : G.offsets(), nullptr, sums2, G.offsets(), ng_dev, order, pack_shift);
  }
  LAUNCH_TRY("ordered_block_sums_kernel");
  // tile-id sort: two 8-bit passes in general; ONE 11-bit pass when the image has at most 2048 tiles and the instances fit
  // SORT_WIDE_MAX_TILES sort tiles (640x480: three launches and a pass over the instances less)
#ifdef SEGS_MEASURE   // measurement builds only (tools/)
  static const bool no_wide = getenv("SEGS_NO_WIDE_DIGIT") != nullptr;
#else
  constexpr bool no_wide = false;
#endif
  const bool wide_digit = !no_wide && bit <= 11 && BL.nblocks <= SORT_WIDE_MAX_TILES;
  const int tpasses = wide_digit ? 1 : (bit + 7) / 8;
  const int side = tpasses & 1;
  { PROF(K_SCAN);
  ordered_offsets_kernel<<<prefix_wgs, 256, 0, st>>>(P, sums2, G.offsets(), total_out, ng_dev, first_owner,
                                                      (uint32_t)(n_cap / EMIT_SLOTS_PER_WG + 2));
  }
  LAUNCH_TRY("ordered_offsets_kernel");
  const bool unfused = (g_flags & SEGS_RASTER_UNFUSED_BINNING) != 0u;
  { PROF(K_DUPLICATE);
  duplicate_with_keys_kernel<<<(n_cap + EMIT_SLOTS_PER_WG - 1) / EMIT_SLOTS_PER_WG, 256, 0, st>>>(P, n_cap, G.rec(), order, G.offsets(),
                                                                    (uint32_t*)(bin + BL.keys[side]), (uint32_t*)(bin + BL.vals[side]), gx, n_dev,
                                                                    drop_dead ? 1 : 0, ng_dev, first_owner);
  }
  LAUNCH_TRY("duplicate_with_keys_kernel");
  // (3)  With two passes the last one fills the range table and the status words itself (radix_scatter_kernel); a single pass
  // over depth-ordered input would need two atomics per instance for that and keeps the range kernel.
  SortFusion fuse;
  const bool fused_ranges = !unfused && tpasses >= 2;
  if (fused_ranges) {
    fuse.ranges = ranges;
    fuse.status = n_dev ? total_out : nullptr;
    fuse.status_mirror = n_dev ? g_status_mirror : nullptr;
    fuse.keep_sorted_keys = n_dev == nullptr;   // resident mode: nothing reads the sorted tile ids after this
  }
  rc = wide_digit ? sort_pairs<uint32_t, 11>(bin, BL, n_cap, bit, 0u, 0, st, n_dev, drop_dead, false, nullptr, nullptr, 0, fuse)
                  : sort_pairs<uint32_t>(bin, BL, n_cap, bit, 0u, 0, st, n_dev, drop_dead, false, nullptr, nullptr, 0, fuse);
  if (rc) return rc;
  if (!fused_ranges) { PROF(K_RANGES);
  identify_tile_ranges_kernel<<<(n_cap + 256 * RANGE_KEYS_PER_THREAD - 1) / (256 * RANGE_KEYS_PER_THREAD), 256, 0, st>>>(n_cap, (const uint32_t*)(bin + BL.keys[0]), ranges, n_dev,
                                                                   n_dev ? total_out : nullptr, n_dev ? g_status_mirror : nullptr,
                                                                   drop_dead ? (const uint32_t*)(bin + BL.n_live) : nullptr);
  }
  LAUNCH_TRY("identify_tile_ranges_kernel");
  return SEGS_OK;
}

}  // namespace

extern "C" {

const char* segs_last_error(void) { return g_err.c_str(); }

uint32_t* segs_raster_set_status_mirror(uint32_t* host_mapped_status) {
  uint32_t* old = g_status_mirror;
  g_status_mirror = host_mapped_status;
  return old;
}

uint32_t segs_raster_set_flags(uint32_t flags) {
  const uint32_t old = g_flags;
  g_flags = flags;
  return old;
}

size_t segs_geometry_bytes(int P) { return geom_layout(P < 0 ? 0 : P).total; }
size_t segs_image_bytes(int width, int height) { return image_layout(width, height).total; }
size_t segs_binning_bytes(int num_rendered) { return binning_layout(num_rendered < 0 ? 0 : num_rendered).total; }

int segs_rasterize_forward(segs_alloc_fn geometry_alloc, void* geometry_ctx, segs_alloc_fn binning_alloc, void* binning_ctx,
                           segs_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                           int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                           const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                           int* radii, void* stream, int* num_rendered) {
  (void)prefiltered;
  hipStream_t st = (hipStream_t)stream;
  if (!geometry_alloc || !binning_alloc || !image_alloc) return fail(SEGS_ERR_INVALID_ARGUMENT, "null allocator callback");
  if (P < 0 || width <= 0 || height <= 0) return fail(SEGS_ERR_INVALID_ARGUMENT, "bad P / image size");
  if (P > MAX_GAUSSIANS) return fail(SEGS_ERR_INVALID_ARGUMENT, "P exceeds 2^28 Gaussians (sort values carry a 4-bit quadrant mask)");
  if (!background || !out_color || !viewmatrix || !projmatrix || !num_rendered) return fail(SEGS_ERR_INVALID_ARGUMENT, "null required pointer");
  if (P > 0 && (!means3D || !opacities)) return fail(SEGS_ERR_INVALID_ARGUMENT, "null means3D/opacities");
  if (P > 0 && !colors_precomp) {
    if (!shs || !cam_pos || M <= 0 || D < 0 || (D + 1) * (D + 1) > M || D > 3)
      return fail(SEGS_ERR_INVALID_ARGUMENT, "need colors_precomp, or shs + cam_pos with (D+1)^2 <= M, D <= 3");
  }
  if (P > 0 && !cov3D_precomp && (!scales || !rotations)) return fail(SEGS_ERR_INVALID_ARGUMENT, "need scales+rotations or cov3D_precomp");
  const uint32_t gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
  if (gx > 0xFFFFu || gy > 0xFFFFu) return fail(SEGS_ERR_INVALID_ARGUMENT, "image too large for 16-bit tile coordinates");

  const GeomLayout GL = geom_layout(P);
  char* geom_raw = geometry_alloc(geometry_ctx, GL.total);
  const ImageLayout IL = image_layout(width, height);
  char* img_raw = image_alloc(image_ctx, IL.total);
  if (!geom_raw || !img_raw) return fail(SEGS_ERR_ALLOC, "allocator callback returned null");
  Geom G = geom_at(geom_raw, P);
  char* img = align_ptr(img_raw);
  if (!radii) radii = G.radii_internal();

  int R = 0;
  uint32_t hdr[3] = {0u, 0u, 0u};  // num_rendered, max(~depth_bits), max(depth_bits)
  const bool tight = (g_flags & SEGS_RASTER_TIGHT_BINNING) != 0u;   // segs_raster.h: shorter lists, same image and gradients
  // Tight mode has no use for the per-Gaussian bin records (they feed make_depth_keys_kernel and the debug unpackers): K1 writes
  // the 32-bit depth keys of the resident forward instead -- into the bin records' place, the sort scratch does not exist before
  // R is known -- and the depth sort's first pass reads them there (16 B x P less to write, one launch and a 12 B x P pass less).
  uint2* const ranges_early = (uint2*)(img + IL.ranges);
  bool fast_keys = tight;
  if (P > 0) {
    int rc = run_preprocess(G, P, width, height, means3D, colors_precomp, opacities, cov3D_precomp ? nullptr : scales,
                            scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, radii, shs, D, M,
                            cam_pos, st, fast_keys ? (uint32_t*)G.bin() : nullptr, nullptr, fast_keys ? ranges_early : nullptr,
                            tight ? PREPROCESS_TIGHT_RECT : 0u);
    if (rc) return rc;
    { PROF(K_SCAN);
    scan_block_sums_kernel<<<1, 1024, 0, st>>>(G.block_sums(), G.L.nblocks, G.block_sums() + (G.L.nblocks + 1), G.num_rendered());
    }
    LAUNCH_TRY("scan_block_sums_kernel");
    HIP_TRY(hipMemcpyAsync(hdr, G.num_rendered(), sizeof(hdr), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    R = (int)hdr[0];
    if (R < 0) return fail(SEGS_ERR_INVALID_ARGUMENT, "num_rendered overflows int32");
    if (fast_keys && R > 0 && (hdr[2] - DEPTH_KEY_MIN) >= ((1u << DEPTH_KEY_BITS) - 1u)) {
      // a binned depth beyond the 27-bit key range of the three 9-bit passes (13 107 m): redo K1 for the exact-range sort
      fast_keys = false;
      rc = run_preprocess(G, P, width, height, means3D, colors_precomp, opacities, cov3D_precomp ? nullptr : scales, scale_modifier,
                          rotations, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, radii, shs, D, M, cam_pos, st, nullptr,
                          nullptr, nullptr, PREPROCESS_TIGHT_RECT);
      if (rc) return rc;
    }
  }
  const BinningLayout BL = binning_layout(R);            // instance-level state (what backward re-parses)
  const GaussSortLayout GS = gauss_sort_layout(R, P);    // + Gaussian-level depth sort scratch behind it
  char* bin_raw = binning_alloc(binning_ctx, GS.total);
  if (!bin_raw) return fail(SEGS_ERR_ALLOC, "binning allocator returned null");
  char* bin = align_ptr(bin_raw);

  uint2* ranges = (uint2*)(img + IL.ranges);
  if (R == 0) {   // rasterizer_impl.cu:310; with instances run_binning zeroes the table itself
    PROF(K_MEMSET);
    HIP_TRY(hipMemsetAsync(ranges, 0, (size_t)gx * gy * sizeof(uint2), st));
  }
  if (R > 0) {
    const uint32_t dmin = ~hdr[1], dspan = hdr[2] - dmin + 1u;   // +1: the key of culled Gaussians, one past the deepest visible
    int dbits = 1;
    while (dbits < 32 && (dspan >> dbits) != 0u) dbits++;
    uint32_t* total_scratch = (uint32_t*)(bin + GS.block_sums) + G.L.nblocks;
    const bool nine = (dbits + 8) / 9 < (dbits + 7) / 8;   // e.g. the usual 26 bits: three 9-bit passes instead of four 8-bit ones
    int rc = fast_keys ? run_binning(G, bin, BL, GS, ranges, P, R, nullptr, DEPTH_KEY_MIN, DEPTH_KEY_BITS, 0xFFFFFFFFu, gx, gy, total_scratch,
                                     st, true, true, true, (const uint32_t*)G.bin())
                       : run_binning(G, bin, BL, GS, ranges, P, R, nullptr, dmin, dbits, dmin + dspan, gx, gy, total_scratch, st, false, tight, nine);
    if (rc) return rc;
  }
  { PROF(K_RENDER_FWD);
  render_fwd_kernel<<<gx * gy, 256, 0, st>>>(ranges, (const uint32_t*)(bin + BL.vals[0]), width, height, G.rec(),
                                                  background, (float*)(img + IL.final_T), (uint32_t*)(img + IL.n_contrib), out_color);
  }
  LAUNCH_TRY("render_fwd_kernel");
  *num_rendered = R;
  return SEGS_OK;
}

// self_clean (resident entry point): the per-Gaussian accumulator rows are not cleared by a memset before the tile kernel;
// preprocess_bwd_kernel writes zeros back over every row it consumes, so a buffer that starts out zero-filled is clean
// again after every backward (one 32 MB fill and its launch less per iteration).
static int rasterize_backward_impl(int P, int D, int M, int R, const float* background, int width, int height,
                            const float* means3D, const float* shs, const float* scales,
                            float scale_modifier, const float* rotations, const float* cov3D_precomp,
                            const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                            float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                            const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                            float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, void* stream,
                            bool self_clean, int geom_rows) {
  hipStream_t st = (hipStream_t)stream;
  if (P < 0 || R < 0 || width <= 0 || height <= 0) return fail(SEGS_ERR_INVALID_ARGUMENT, "bad sizes");
  if (geom_rows < P) return fail(SEGS_ERR_INVALID_ARGUMENT, "geom_rows must be >= P");
  if (P == 0) return SEGS_OK;  // src/rasterize_points.cu:159
  if (shs && (!campos || !dL_dsh || M <= 0)) return fail(SEGS_ERR_INVALID_ARGUMENT, "SH path needs campos, dL_dsh and M > 0");
  if (!geom_buffer || !binning_buffer || !image_buffer || !dL_dpix || !background || !means3D || !viewmatrix || !projmatrix)
    return fail(SEGS_ERR_INVALID_ARGUMENT, "null required pointer");
  // dL_dconic (the tile kernel's internal product) and dL_dcov3D (of use only with cov3D_precomp) may be null: not written
  if (!dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || (cov3D_precomp && !dL_dcov3D))
    return fail(SEGS_ERR_INVALID_ARGUMENT, "null gradient output");
  if (!cov3D_precomp && (!scales || !rotations || !dL_dscale || !dL_drot))
    return fail(SEGS_ERR_INVALID_ARGUMENT, "need scales+rotations (+ their gradient outputs) or cov3D_precomp");
  Geom G = geom_at(geom_buffer, P, geom_rows);
  if (!radii) radii = G.radii_internal();
  const ImageLayout IL = image_layout(width, height);
  const BinningLayout BL = binning_layout(R);
  char* img = align_ptr(image_buffer);
  char* bin = align_ptr(binning_buffer);
  const uint32_t gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
  const float focal_y = height / (2.0f * tan_fovy), focal_x = width / (2.0f * tan_fovx);  // rasterizer_impl.cu:436-437

  if (!self_clean) { PROF(K_MEMSET);
  HIP_TRY(hipMemsetAsync(G.gacc(), 0, (size_t)P * GACC_DWORDS * 4, st));
  }
  if (R > 0) {
    { PROF(K_RENDER_BWD);
    static const bool mfma_env = [] { const char* e = getenv("SEGS_RENDER_BWD_MFMA"); return e && e[0] == '1'; }();   // measurement A/B only
    ((mfma_env || (g_flags & SEGS_RASTER_MFMA_MOMENTS)) ? render_bwd_mfma_kernel : render_bwd_kernel)<<<(gx * gy + 7) / 8 * 32, 64, 0, st>>>((const uint2*)(img + IL.ranges), (const uint32_t*)(bin + BL.vals[0]), width,
                                                    height, G.rec(), background, (const float*)(img + IL.final_T),
                                                    (const uint32_t*)(img + IL.n_contrib), dL_dpix, G.gacc(), gx * gy);
    }
    LAUNCH_TRY("render_bwd_kernel");
  }
  { PROF(K_PREPROCESS_BWD);
  preprocess_bwd_kernel<<<G.L.nblocks, 256, 0, st>>>(P, means3D, radii, cov3D_precomp ? nullptr : scales, rotations,
                                                     scale_modifier, cov3D_precomp, viewmatrix, projmatrix, focal_x, focal_y,
                                                     tan_fovx, tan_fovy, G.gacc(), (float)width, (float)height, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                                                     dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, self_clean ? 1 : 0);
  }
  LAUNCH_TRY("preprocess_bwd_kernel");
  if (shs) {   // SH colour branch (off the live SEGS-SLAM path): its own pass over the summed dL/dcolor
    sh_backward_kernel<<<G.L.nblocks, 256, 0, st>>>(P, means3D, radii, shs, D, M, campos, G.clamped(), dL_dcolor, dL_dmean3D, dL_dsh);
    LAUNCH_TRY("sh_backward_kernel");
  }
  return SEGS_OK;
}

int segs_rasterize_backward(int P, int D, int M, int R, const float* background, int width, int height,
                            const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                            float scale_modifier, const float* rotations, const float* cov3D_precomp,
                            const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                            float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                            const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                            float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, void* stream) {
  (void)colors_precomp;
  return rasterize_backward_impl(P, D, M, R, background, width, height, means3D, shs, scales, scale_modifier, rotations,
                                 cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer,
                                 image_buffer, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                                 dL_dscale, dL_drot, stream, false, P);
}

static int visible_filter_impl(int P, int width, int height, const float* means3D, const float* scales, int log_scale_stride,
                               float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                               const float* projmatrix, float tan_fovx, float tan_fovy, int* radii, void* stream);

int segs_visible_filter(int P, int M, int width, int height, const float* means3D, const float* scales,
                        float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                        const float* projmatrix, float tan_fovx, float tan_fovy, int prefiltered, int* radii, void* stream) {
  (void)M; (void)prefiltered;
  return visible_filter_impl(P, width, height, means3D, scales, 0, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                             tan_fovx, tan_fovy, radii, stream);
}

int segs_visible_filter_log_scales(int P, int width, int height, const float* means3D, const float* scaling_log, int stride,
                                   const float* rotations, const float* viewmatrix, const float* projmatrix, float tan_fovx,
                                   float tan_fovy, int* radii, void* stream) {
  if (stride < 3 || !scaling_log || !rotations) return fail(SEGS_ERR_INVALID_ARGUMENT, "need log-scales (stride >= 3) and rotations");
  return visible_filter_impl(P, width, height, means3D, scaling_log, stride, 1.0f, rotations, nullptr, viewmatrix, projmatrix, tan_fovx,
                             tan_fovy, radii, stream);
}

static int visible_filter_impl(int P, int width, int height, const float* means3D, const float* scales, int log_scale_stride,
                               float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                               const float* projmatrix, float tan_fovx, float tan_fovy, int* radii, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (P < 0 || width <= 0 || height <= 0) return fail(SEGS_ERR_INVALID_ARGUMENT, "bad sizes");
  if (P == 0) return SEGS_OK;
  if (!means3D || !viewmatrix || !projmatrix || !radii) return fail(SEGS_ERR_INVALID_ARGUMENT, "null required pointer");
  if (!cov3D_precomp && (!scales || !rotations)) return fail(SEGS_ERR_INVALID_ARGUMENT, "need scales+rotations or cov3D_precomp");
  const float focal_y = height / (2.0f * tan_fovy), focal_x = width / (2.0f * tan_fovx);  // rasterizer_impl.cu:358-359
  const uint32_t gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
  visible_filter_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, means3D, cov3D_precomp ? nullptr : scales, scale_modifier, rotations,
                                                         cov3D_precomp, viewmatrix, projmatrix, width, height, tan_fovx, tan_fovy,
                                                         focal_x, focal_y, gx, gy, radii, log_scale_stride);
  LAUNCH_TRY("visible_filter_kernel");
  return SEGS_OK;
}

int segs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present, void* stream) {
  (void)projmatrix;
  hipStream_t st = (hipStream_t)stream;
  if (P < 0) return fail(SEGS_ERR_INVALID_ARGUMENT, "bad P");
  if (P == 0) return SEGS_OK;
  if (!means3D || !viewmatrix || !present) return fail(SEGS_ERR_INVALID_ARGUMENT, "null required pointer");
  mark_visible_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, means3D, viewmatrix, present);
  LAUNCH_TRY("mark_visible_kernel");
  return SEGS_OK;
}

int segs_debug_unpack_geometry(const char* geom_buffer, int P, const int* radii, float* means2D, float* conic_opacity,
                               float* depths, uint32_t* tiles_touched, uint32_t* point_offsets, float* rgb, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (P <= 0) return SEGS_OK;
  if (!geom_buffer || !means2D || !conic_opacity || !depths || !tiles_touched) return fail(SEGS_ERR_INVALID_ARGUMENT, "null pointer");
  Geom G = geom_at(const_cast<char*>(geom_buffer), P);
  if (!radii) radii = G.radii_internal();
  unpack_geometry_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, G.rec(), G.bin(), radii, means2D, conic_opacity, depths, tiles_touched, rgb);
  LAUNCH_TRY("unpack_geometry_kernel");
  if (point_offsets) {
    point_offsets_kernel<<<1, 1024, 0, st>>>(P, G.bin(), point_offsets);
    LAUNCH_TRY("point_offsets_kernel");
  }
  return SEGS_OK;
}

int segs_debug_unpack_binning(const char* binning_buffer, const char* geom_buffer, int P, int R, int width, int height,
                              uint64_t* keys_sorted, uint32_t* point_list, void* stream) {
  (void)width; (void)height;
  hipStream_t st = (hipStream_t)stream;
  if (R <= 0) return SEGS_OK;
  if (!binning_buffer || (keys_sorted && (!geom_buffer || P <= 0))) return fail(SEGS_ERR_INVALID_ARGUMENT, "null pointer");
  const BinningLayout BL = binning_layout(R);
  const char* bin = align_ptr(binning_buffer);
  if (keys_sorted) {   // the pipeline sorts 32-bit tile ids; the reference's 64-bit keys are rebuilt for parity checks
    const Geom G = geom_at(const_cast<char*>(geom_buffer), P);
    rebuild_keys_kernel<<<(R + 255) / 256, 256, 0, st>>>(R, (const uint32_t*)(bin + BL.keys[0]), (const uint32_t*)(bin + BL.vals[0]),
                                                        G.bin(), keys_sorted);
    LAUNCH_TRY("rebuild_keys_kernel");
  }
  if (point_list) {
    strip_mask_kernel<<<(R + 255) / 256, 256, 0, st>>>(R, (const uint32_t*)(bin + BL.vals[0]), point_list);
    LAUNCH_TRY("strip_mask_kernel");
  }
  return SEGS_OK;
}

int segs_debug_instance_values(const char* binning_buffer, int R, uint32_t* values, void* stream) {
  if (R <= 0) return SEGS_OK;
  if (!binning_buffer || !values) return fail(SEGS_ERR_INVALID_ARGUMENT, "null pointer");
  const BinningLayout BL = binning_layout(R);
  HIP_TRY(hipMemcpyAsync(values, align_ptr(binning_buffer) + BL.vals[0], (size_t)R * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SEGS_OK;
}

int segs_debug_unpack_image(const char* image_buffer, int width, int height, uint32_t* ranges, float* final_T,
                            uint32_t* n_contrib, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!image_buffer) return fail(SEGS_ERR_INVALID_ARGUMENT, "null pointer");
  const ImageLayout IL = image_layout(width, height);
  const char* img = align_ptr(image_buffer);
  const size_t tiles = (size_t)((width + TILE_X - 1) / TILE_X) * ((height + TILE_Y - 1) / TILE_Y);
  if (ranges) {
    HIP_TRY(hipMemcpyAsync(ranges, img + IL.ranges, tiles * 8, hipMemcpyDeviceToDevice, st));
    normalize_ranges_kernel<<<(int)((tiles + 255) / 256), 256, 0, st>>>((int)tiles, (uint2*)ranges);   // empty tiles: {0, 0} as in the reference
    LAUNCH_TRY("normalize_ranges_kernel");
  }
  if (final_T) HIP_TRY(hipMemcpyAsync(final_T, img + IL.final_T, (size_t)width * height * 4, hipMemcpyDeviceToDevice, st));
  if (n_contrib) HIP_TRY(hipMemcpyAsync(n_contrib, img + IL.n_contrib, (size_t)width * height * 4, hipMemcpyDeviceToDevice, st));
  return SEGS_OK;
}

int segs_debug_preprocess_backward(int P, int width, int height, const float* means3D, const int* radii, const float* scales,
                                   float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                   const float* viewmatrix, const float* projmatrix, float tan_fovx, float tan_fovy,
                                   const float* dL_dmean2D, const float* dL_dconic, float* dL_dmean3D, float* dL_dcov3D,
                                   float* dL_dscale, float* dL_drot, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (P <= 0) return SEGS_OK;
  if (!means3D || !radii || !viewmatrix || !projmatrix || !dL_dmean2D || !dL_dconic || !dL_dmean3D || !dL_dcov3D)
    return fail(SEGS_ERR_INVALID_ARGUMENT, "null pointer");
  const float focal_y = height / (2.0f * tan_fovy), focal_x = width / (2.0f * tan_fovx);
  preprocess_bwd_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, means3D, radii, cov3D_precomp ? nullptr : scales, rotations,
                                                         scale_modifier, cov3D_precomp, viewmatrix, projmatrix, focal_x, focal_y,
                                                         tan_fovx, tan_fovy, nullptr, (float)width, (float)height, const_cast<float*>(dL_dmean2D),
                                                         const_cast<float*>(dL_dconic), nullptr, nullptr, dL_dmean3D, dL_dcov3D,
                                                         dL_dscale, dL_drot, 0);
  LAUNCH_TRY("preprocess_bwd_kernel");
  return SEGS_OK;
}

int segs_sort_pairs(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, int n,
                    int end_bit, char* temp, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n < 0 || end_bit <= 0 || end_bit > 64) return fail(SEGS_ERR_INVALID_ARGUMENT, "bad n / end_bit");
  if (n == 0) return SEGS_OK;
  if (!keys_in || !vals_in || !keys_out || !vals_out || !temp) return fail(SEGS_ERR_INVALID_ARGUMENT, "null pointer");
  const BinningLayout BL = binning_layout(n);
  char* bin = align_ptr(temp);
  const int passes = (end_bit + 7) / 8;
  const int side = passes & 1;
  HIP_TRY(hipMemcpyAsync(bin + BL.keys[side], keys_in, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(bin + BL.vals[side], vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  int rc = sort_pairs<uint64_t>(bin, BL, n, end_bit, 0u, 32, st);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(keys_out, bin + BL.keys[0], (size_t)n * 8, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(vals_out, bin + BL.vals[0], (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  return SEGS_OK;
}

int segs_project2_image(int P, int D, int M, int width, int height, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                        float* points_image, int* radii, void* stream) {
  (void)prefiltered;
  hipStream_t st = (hipStream_t)stream;
  if (P < 0 || width <= 0 || height <= 0) return fail(SEGS_ERR_INVALID_ARGUMENT, "bad sizes");
  if (P == 0) return SEGS_OK;
  if (!colors_precomp && (!shs || !cam_pos || M <= 0)) return fail(SEGS_ERR_INVALID_ARGUMENT, "need colors_precomp or shs + cam_pos");
  if (!means3D || !opacities || !viewmatrix || !projmatrix || !out_color || !points_image || !radii)
    return fail(SEGS_ERR_INVALID_ARGUMENT, "null required pointer");
  // scratch: this entry has no allocator callbacks worth keeping (the reference allocates full geometry and
  // image states it then discards); use a stream-ordered temporary.
  const GeomLayout GL = geom_layout(P);
  char* raw = nullptr;
  HIP_TRY(hipMallocAsync((void**)&raw, GL.total + (size_t)P * 4 * 8, st));
  Geom G = geom_at(raw, P);
  float* tmp = (float*)(align_ptr(raw) + GL.total - ALIGN);  // conic(4P) + depth(P) + tiles(P)
  int rc = run_preprocess(G, P, width, height, means3D, colors_precomp, opacities, cov3D_precomp ? nullptr : scales,
                          scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, radii, shs, D, M,
                          cam_pos, st);
  if (rc == SEGS_OK) {
    unpack_geometry_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, G.rec(), G.bin(), radii, points_image, tmp, tmp + (size_t)4 * P,
                                                            (uint32_t*)(tmp + (size_t)5 * P), out_color);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = hip_fail(e, "unpack_geometry_kernel");
  }
  hipError_t fe = hipFreeAsync(raw, st);
  if (rc == SEGS_OK && fe != hipSuccess) return hip_fail(fe, "hipFreeAsync");
  return rc;
}


// ---- Resident (steady-state) variants: no host synchronisation, fixed launch sequence (hipGraph-capturable). ----
size_t segs_resident_binning_bytes(int P, int capacity) { return gauss_sort_layout(capacity < 0 ? 0 : capacity, P < 0 ? 0 : P).total; }

int segs_rasterize_forward_resident(char* geom_buffer, char* binning_buffer, char* image_buffer, int capacity, int geom_rows, int P, int D, int M,
                                    const float* background, int width, int height, const float* means3D, const float* shs,
                                    const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                                    const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, float* out_color,
                                    int* radii, uint32_t* status, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (P <= 0 || capacity <= 0 || width <= 0 || height <= 0) return fail(SEGS_ERR_INVALID_ARGUMENT, "bad sizes");
  if (geom_rows < P) return fail(SEGS_ERR_INVALID_ARGUMENT, "geom_rows (rows the geometry buffer was sized for) must be >= P");
  if (geom_rows > MAX_GAUSSIANS) return fail(SEGS_ERR_INVALID_ARGUMENT, "P exceeds 2^28 Gaussians");
  if (!geom_buffer || !binning_buffer || !image_buffer || !status || !background || !out_color || !viewmatrix || !projmatrix ||
      !means3D || !opacities)
    return fail(SEGS_ERR_INVALID_ARGUMENT, "null required pointer");
  if (!colors_precomp && (!shs || !cam_pos || M <= 0 || D < 0 || (D + 1) * (D + 1) > M || D > 3))
    return fail(SEGS_ERR_INVALID_ARGUMENT, "need colors_precomp, or shs + cam_pos with (D+1)^2 <= M, D <= 3");
  if (!cov3D_precomp && (!scales || !rotations)) return fail(SEGS_ERR_INVALID_ARGUMENT, "need scales+rotations or cov3D_precomp");
  const uint32_t gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
  if (gx > 0xFFFFu || gy > 0xFFFFu) return fail(SEGS_ERR_INVALID_ARGUMENT, "image too large for 16-bit tile coordinates");
  Geom G = geom_at(geom_buffer, P, geom_rows);
  char* img = align_ptr(image_buffer);
  char* bin = align_ptr(binning_buffer);
  const ImageLayout IL = image_layout(width, height);
  const BinningLayout BL = binning_layout(capacity);
  const GaussSortLayout GS = gauss_sort_layout(capacity, P);
  if (!radii) radii = G.radii_internal();
  uint2* ranges = (uint2*)(img + IL.ranges);
  // The exact depth range is only known on the device, so K1 writes the raw depth bits as keys -- and zeroes the range
  // table -- itself (one launch less) and the sort looks at DEPTH_KEY_BITS = 27 bits above the near plane's pattern in
  // three 9-bit passes (see kernels.h); Gaussians that own no instance carry the last key of that range.
  char* gbin = bin + GS.base;
  const int gside = (3 & 1);   // three passes: the sort starts from side 1 (see sort_pairs)
  int rc = run_preprocess(G, P, width, height, means3D, colors_precomp, opacities, cov3D_precomp ? nullptr : scales, scale_modifier,
                          rotations, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, radii, shs, D, M, cam_pos, st,
                          (uint32_t*)(gbin + GS.inner.keys[gside]), (uint32_t*)(gbin + GS.inner.vals[gside]), ranges,
                          (g_flags & SEGS_RASTER_KEEP_DEAD_INSTANCES) ? 0u : PREPROCESS_TIGHT_RECT, status + 2);
  if (rc) return rc;
  // dead instances (no quadrant of their tile can reach alpha >= 1/255: 43 % of them at 500 k Gaussians / 1080p) are
  // dropped by the first tile-id pass; lists, ranges and n_contrib then count live entries only -- an internal contract
  // between this forward and its backward, like the reference's own scratch layout
  rc = run_binning(G, bin, BL, GS, ranges, P, capacity, status, DEPTH_KEY_MIN, DEPTH_KEY_BITS, 0xFFFFFFFFu, gx, gy, status, st, true,
                   (g_flags & SEGS_RASTER_KEEP_DEAD_INSTANCES) == 0u, true);
  if (rc) return rc;
  // status[3] (overflow) and the host mirror are written by identify_tile_ranges_kernel at the end of run_binning
  { PROF(K_RENDER_FWD);
  render_fwd_kernel<<<gx * gy, 256, 0, st>>>(ranges, (const uint32_t*)(bin + BL.vals[0]), width, height, G.rec(), background,
                                                  (float*)(img + IL.final_T), (uint32_t*)(img + IL.n_contrib), out_color);
  }
  LAUNCH_TRY("render_fwd_kernel");
  return SEGS_OK;
}

// ---- K1 done by the producer of the Gaussians (segs_neural_forward_projected): where its outputs go, and the forward without K1.
int segs_resident_projection_targets(char* geom_buffer, char* binning_buffer, char* image_buffer, int capacity, int geom_rows, int P,
                                     int width, int height, int* radii, uint32_t* status, segs_projection_targets* out) {
  if (P <= 0 || capacity <= 0 || width <= 0 || height <= 0 || !out) return fail(SEGS_ERR_INVALID_ARGUMENT, "bad sizes");
  if (geom_rows < P) return fail(SEGS_ERR_INVALID_ARGUMENT, "geom_rows (rows the geometry buffer was sized for) must be >= P");
  if (geom_rows > MAX_GAUSSIANS) return fail(SEGS_ERR_INVALID_ARGUMENT, "P exceeds 2^28 Gaussians");
  if (!geom_buffer || !binning_buffer || !image_buffer || !status) return fail(SEGS_ERR_INVALID_ARGUMENT, "null required pointer");
  const uint32_t gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
  if (gx > 0xFFFFu || gy > 0xFFFFu) return fail(SEGS_ERR_INVALID_ARGUMENT, "image too large for 16-bit tile coordinates");
  Geom G = geom_at(geom_buffer, P, geom_rows);
  const ImageLayout IL = image_layout(width, height);
  const GaussSortLayout GS = gauss_sort_layout(capacity, P);
  const int gside = (3 & 1);   // as in segs_rasterize_forward_resident: three 9-bit depth passes start from side 1
  out->records = G.rec();
  out->radii = radii ? radii : G.radii_internal();
  out->tiles_touched = G.touched();
  out->depth_keys = (uint32_t*)(align_ptr(binning_buffer) + GS.base + GS.inner.keys[gside]);
  out->tile_ranges = (uint32_t*)(align_ptr(image_buffer) + IL.ranges);
  out->depth_overflow = status + 2;
  out->num_tiles = (int)(gx * gy);
  out->flags = (g_flags & SEGS_RASTER_KEEP_DEAD_INSTANCES) ? 0u : PREPROCESS_TIGHT_RECT;
  return SEGS_OK;
}

int segs_rasterize_forward_resident_projected(char* geom_buffer, char* binning_buffer, char* image_buffer, int capacity, int geom_rows, int P,
                                              const float* background, int width, int height, float* out_color, uint32_t* status,
                                              void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (P <= 0 || capacity <= 0 || width <= 0 || height <= 0) return fail(SEGS_ERR_INVALID_ARGUMENT, "bad sizes");
  if (geom_rows < P) return fail(SEGS_ERR_INVALID_ARGUMENT, "geom_rows (rows the geometry buffer was sized for) must be >= P");
  if (geom_rows > MAX_GAUSSIANS) return fail(SEGS_ERR_INVALID_ARGUMENT, "P exceeds 2^28 Gaussians");
  if (!geom_buffer || !binning_buffer || !image_buffer || !status || !background || !out_color)
    return fail(SEGS_ERR_INVALID_ARGUMENT, "null required pointer");
  const uint32_t gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
  if (gx > 0xFFFFu || gy > 0xFFFFu) return fail(SEGS_ERR_INVALID_ARGUMENT, "image too large for 16-bit tile coordinates");
  Geom G = geom_at(geom_buffer, P, geom_rows);
  char* img = align_ptr(image_buffer);
  char* bin = align_ptr(binning_buffer);
  const ImageLayout IL = image_layout(width, height);
  const BinningLayout BL = binning_layout(capacity);
  const GaussSortLayout GS = gauss_sort_layout(capacity, P);
  uint2* ranges = (uint2*)(img + IL.ranges);
  int rc = run_binning(G, bin, BL, GS, ranges, P, capacity, status, DEPTH_KEY_MIN, DEPTH_KEY_BITS, 0xFFFFFFFFu, gx, gy, status, st, true,
                       (g_flags & SEGS_RASTER_KEEP_DEAD_INSTANCES) == 0u, true);
  if (rc) return rc;
  { PROF(K_RENDER_FWD);
  render_fwd_kernel<<<gx * gy, 256, 0, st>>>(ranges, (const uint32_t*)(bin + BL.vals[0]), width, height, G.rec(), background,
                                                  (float*)(img + IL.final_T), (uint32_t*)(img + IL.n_contrib), out_color);
  }
  LAUNCH_TRY("render_fwd_kernel");
  return SEGS_OK;
}

int segs_rasterize_backward_resident(char* geom_buffer, char* binning_buffer, char* image_buffer, int capacity, int geom_rows, int P, int D, int M,
                                     const float* background, int width, int height, const float* means3D, const float* shs,
                                     const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                     const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                                     float tan_fovy, const int* radii, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                                     float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                                     float* dL_dscale, float* dL_drot, void* stream) {
  // identical to segs_rasterize_backward except that the scratch layout is keyed by the capacity, not by R, and that the
  // accumulator rows of geom_buffer are kept clean by the backward itself (the buffer must start out zero-filled)
  return rasterize_backward_impl(P, D, M, capacity, background, width, height, means3D, shs, scales, scale_modifier, rotations,
                                 cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer,
                                 image_buffer, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                                 dL_dscale, dL_drot, stream, true, geom_rows);
}

// ---- measurement support (bench.py): HIP events recorded on the launch stream around selected kernels.
int segs_profile_begin(unsigned kernel_mask) {
  g_prof.mask = kernel_mask; g_prof.used = 0; g_prof.spans.clear();
  for (int i = 0; i < K_COUNT; i++) { g_prof.total_ms[i] = 0; g_prof.count[i] = 0; }
  return SEGS_OK;
}
int segs_profile_end(void) {
  g_prof.mask = 0;
  for (const auto& sp : g_prof.spans) {
    HIP_TRY(hipEventSynchronize(g_prof.pool[sp.e1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, g_prof.pool[sp.e0], g_prof.pool[sp.e1]));
    g_prof.total_ms[sp.id] += ms; g_prof.count[sp.id]++;
  }
  g_prof.spans.clear(); g_prof.used = 0;
  return SEGS_OK;
}
int segs_profile_kernel_count(void) { return K_COUNT; }
const char* segs_profile_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }
int segs_profile_query(int id, double* total_ms, long* launches) {
  if (id < 0 || id >= K_COUNT || !total_ms || !launches) return fail(SEGS_ERR_INVALID_ARGUMENT, "bad profile query");
  *total_ms = g_prof.total_ms[id]; *launches = g_prof.count[id];
  return SEGS_OK;
}

}  // extern "C"
