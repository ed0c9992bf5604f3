// gs_layout.h -- HBM layout of the rasterizer's scratch state (host + device).
//
// The three scratch buffers are opaque bytes to the caller, exactly as in the reference, where
// their carve-up is "an internal contract between forward and backward only"
// (cuda_rasterizer/rasterizer_impl.h:22-73, rasterizer_impl.cu:155-194).  Ours is laid out for
// MI355X: one 64-byte (= one cache line) record per Gaussian holding everything the tile kernels
// consume, so that a per-tile gather is exactly one line per instance and can be fetched by scalar
// (SMEM) loads into SGPRs; compact 16-byte bin records for the instance emitter; 64-byte gradient
// accumulation rows so that one packed float-atomic wave instruction touches a single line.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>

namespace segs {

constexpr int TILE_X = 16;  // cuda_rasterizer/config.h:16
constexpr int TILE_Y = 16;  // cuda_rasterizer/config.h:17
constexpr int NUM_CHANNELS = 3;  // cuda_rasterizer/config.h:15

constexpr float LOG2E = 1.4426950408889634f;

// Per-Gaussian record, 16 dwords = 64 B, 64-B aligned -- the ONLY per-Gaussian record of the pipeline.
//  [0] x  [1] y            pixel-space mean           (reference: GeometryState::means2D)
//  [2] A2 [3] B2 [4] C2    conic pre-scaled for exp2: A2=-0.5*log2e*A, B2=-log2e*B, C2=-0.5*log2e*C
//  [5] opacity             (reference: conic_opacity.w)
//  [6..8] r g b            colour (colors_precomp, or SH->RGB result)
//  [9] depth               view-space z (GeometryState::depths)
//  [10] rect_min  [11] rect_max   tile rectangle, x | y << 16 each (bit patterns)
//  [12..14] A B C          conic as the reference stores it (conic_opacity.xyz; debug unpackers only)   [15] reserved
// The tile kernels read [0..8].  The instance emitter gathers [0..5] and [10..11] of every owner -- the same 32 bytes a
// separate "emit record" held until round 3, in three loads (16 + 8 + 8 bytes) from this record: a random gather pulls the
// whole 128-byte line whatever it reads (tools/ubench_gather.hip), so the second record bought the emitter nothing and cost
// preprocess_fwd_kernel a sixth of its traffic.  The emitter's quadrant test runs on the scaled conic (threshold log2(255 o)).
constexpr int REC_DWORDS = 16;
enum RecField { REC_X = 0, REC_Y, REC_A2, REC_B2, REC_C2, REC_O, REC_R, REC_G, REC_B, REC_DEPTH, REC_RECT_MIN, REC_RECT_MAX, REC_CA, REC_CB, REC_CC };

// Per-Gaussian bin record (uint4): depth bits, rect_min (x | y<<16), rect_max (x | y<<16), tiles_touched.
struct BinInfo { uint32_t depth_bits, rect_min, rect_max, tiles_touched; };
// The instance emitter (K7) tags each (Gaussian, tile) instance with a 4-bit mask of the 8x8 quadrants of the tile
// that contain at least one point of {d : d^T conic d <= 2 ln(255 o)}, i.e. where alpha >= 1/255 is possible at all.
// The mask travels in the top bits of the sort VALUE; a wave (= one quadrant) skips instances whose bit is clear.
// Skipped instances are exactly those the reference `continue`s on for every pixel of the quadrant, so images,
// n_contrib and gradients are unchanged; point_list for parity checks is value & ID_MASK.
constexpr uint32_t ID_BITS = 28;                    // sort value = gaussian idx | quadrant mask << 28
constexpr uint32_t ID_MASK = (1u << ID_BITS) - 1u;
constexpr int MAX_GAUSSIANS = 1 << ID_BITS;

// Gradient accumulation row, 16 dwords (64 B), raw moments of the tile backward (render.hip): [0,1] Mx My  [2,3,4] Mxx Mxy Myy
// [5] dL/dopacity (= S0 / opacity)  [6,7,8] dL/dcolor.
constexpr int GACC_DWORDS = 16;

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v, size_t a = ALIGN) { return (v + a - 1) / a * a; }

struct GeomLayout {
  size_t rec, bin, offsets, radii_internal, block_sums, clamped, num_rendered, gacc, touched, total;
  int P, nblocks;
};
// Mirrors GeometryState::fromChunk (rasterizer_impl.cu:155-170) in role, not in layout.
inline GeomLayout geom_layout(int P) {
  GeomLayout g{};
  g.P = P;
  g.nblocks = (P + 255) / 256;
  size_t o = 0;
  g.rec = o;            o = align_up(o + (size_t)P * REC_DWORDS * 4);
  g.bin = o;            o = align_up(o + (size_t)P * sizeof(BinInfo));
  g.offsets = o;        o = align_up(o + (size_t)P * 4);
  g.radii_internal = o; o = align_up(o + (size_t)P * 4);
  g.block_sums = o;     o = align_up(o + (size_t)(g.nblocks + 1) * 4 * 3);  // tiles sums | max(depth bits) | max(~depth bits)
  g.clamped = o;        o = align_up(o + (size_t)P * 4);  // SH path: 3 clamp flags packed in one word
  g.num_rendered = o;   o = align_up(o + 64);
  g.gacc = o;           o = align_up(o + (size_t)P * GACC_DWORDS * 4);
  g.touched = o;        o = align_up(o + (size_t)P * 4);  // tiles_touched once more, dense: the depth-ordered prefix gathers 4 B, not a 16-B BinInfo
  g.total = o + ALIGN;  // slack so the base pointer can be aligned up
  return g;
}

struct ImageLayout {
  size_t ranges, final_T, n_contrib, total;
};
// Mirrors ImageState::fromChunk (rasterizer_impl.cu:172-179); ranges sized per TILE, not per pixel.
inline ImageLayout image_layout(int W, int H) {
  ImageLayout l{};
  const size_t tiles = (size_t)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
  size_t o = 0;
  l.ranges = o;    o = align_up(o + tiles * 8);
  l.final_T = o;   o = align_up(o + (size_t)W * H * 4);
  l.n_contrib = o; o = align_up(o + (size_t)W * H * 4);
  l.total = o + ALIGN;
  return l;
}

#ifndef SEGS_SORT_ITEMS_PER_THREAD
#define SEGS_SORT_ITEMS_PER_THREAD 8
#endif
constexpr int SORT_ITEMS_PER_THREAD = SEGS_SORT_ITEMS_PER_THREAD;
constexpr int SORT_THREADS = 256;
constexpr int SORT_TILE = SORT_ITEMS_PER_THREAD * SORT_THREADS;  // 2048 items per workgroup
constexpr int PREFIX_ROWS_PER_WG = 8;                             // 256-slot rows per workgroup of the depth-ordered prefix kernels
constexpr int EMIT_SLOTS_PER_WG = 512;                            // instance slots per workgroup of duplicate_with_keys_kernel
constexpr int SORT_COUNT_CHUNK_TILES = 4;                         // tiles per workgroup of radix_count_kernel (binning.hip)
constexpr int RANGE_KEYS_PER_THREAD = 4;                          // identify_tile_ranges_kernel
constexpr int SORT_WIDE_DIGITS = 2048;                           // 11-bit digits: tile-id sort of images of at most 2048 tiles, small instance counts
constexpr int SORT_WIDE_MAX_TILES = 1024;                        // ... up to this many 2048-key sort tiles (2 M instances)
constexpr int SORT_MAX_DIGITS = 512;                             // count-matrix rows: 8-bit passes use 256 of them, 9-bit passes all

struct BinningLayout {
  size_t keys[2], vals[2], tile_prefix, chunk_hist, digit_totals, n_live, total;
  int nchunks;
  int R, nblocks;
};
// Mirrors BinningState::fromChunk (rasterizer_impl.cu:181-194): ping-pong key/value arrays + sort temp.
// The same layout (with n = P) is used for the Gaussian-level depth sort; `binning_bytes(R, P)` holds both.
inline BinningLayout binning_layout(int R) {
  BinningLayout b{};
  b.R = R;
  b.nblocks = (R + SORT_TILE - 1) / SORT_TILE;
  size_t o = 0;
  for (int i = 0; i < 2; i++) { b.keys[i] = o; o = align_up(o + (size_t)R * 8); }
  for (int i = 0; i < 2; i++) { b.vals[i] = o; o = align_up(o + (size_t)R * 4); }
  // per-tile digit offsets inside their chunk, tile-major [nblocks][digits]; per-chunk digit counts, digit-major
  // [digits][nchunks] (scanned in place by radix_scan_kernel)
  b.nchunks = b.nblocks;   // columns ALLOCATED for the chunk totals: one per tile, the smallest chunk sort_pairs may choose
  // count matrices: SORT_MAX_DIGITS rows in general; a sort of at most SORT_WIDE_MAX_TILES tiles may run with SORT_WIDE_DIGITS
  // (one 11-bit pass over the tile ids of a small image instead of two 8-bit ones)
  const size_t cols = (size_t)(b.nblocks > 0 ? b.nblocks : 1);
  const size_t matrix = std::max((size_t)SORT_MAX_DIGITS * cols, (size_t)SORT_WIDE_DIGITS * std::min(cols, (size_t)SORT_WIDE_MAX_TILES));
  b.tile_prefix = o;  o = align_up(o + matrix * 4);
  b.chunk_hist = o;   o = align_up(o + matrix * 4);
  b.digit_totals = o; o = align_up(o + SORT_WIDE_DIGITS * 4);
  b.n_live = o;       o = align_up(o + 4);   // entries left after the first pass dropped the dead keys (sort_pairs)
  b.total = o + ALIGN;
  return b;
}
struct GaussSortLayout {   // appended after the instance-level BinningLayout inside the binning buffer
  size_t base, block_sums, first_owner, total;   // first_owner: per 512-slot emitter workgroup, the depth-ordered Gaussian owning its first slot
  BinningLayout inner;
};
inline GaussSortLayout gauss_sort_layout(int R, int P) {
  GaussSortLayout g{};
  g.inner = binning_layout(P);
  g.base = align_up(binning_layout(R).total);
  g.block_sums = g.base + align_up(g.inner.total);
  g.first_owner = g.block_sums + align_up((size_t)((P + 255) / 256 + 4) * 4);
  g.total = g.first_owner + align_up((size_t)(R / EMIT_SLOTS_PER_WG + 4) * 4) + ALIGN;
  return g;
}

inline char* align_ptr(char* p) { return (char*)(((uintptr_t)p + ALIGN - 1) / ALIGN * ALIGN); }
inline const char* align_ptr(const char* p) { return (const char*)(((uintptr_t)p + ALIGN - 1) / ALIGN * ALIGN); }

}  // namespace segs
