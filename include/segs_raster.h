/* segs_raster.h -- C ABI of the MI355X-native Gaussian-splatting rasterizer (libsegs_raster.so).
 *
 * Drop-in boundary for the hot path of leaner-forever/SEGS-SLAM.  The reference exposes this layer as
 * C++ static methods over raw pointers plus std::function allocator callbacks
 * (cuda_rasterizer/rasterizer.h:24-125, class CudaRasterizer::Rasterizer); a C-ABI replacement must be
 * bindable without C++ types, so each entry point below is that method with
 *   - std::function<char*(size_t)>  ->  a plain C function pointer + void* context,
 *   - an explicit stream (the reference uses the legacy default stream),
 *   - an int status return (0 = ok, <0 = bad argument, >0 = hipError_t) instead of exceptions.
 * All pointers are DEVICE pointers unless stated otherwise; matrices are the reference's transposed
 * 4x4 float tensors (src/gaussian_keyframe.cpp:151-184).  No torch types appear in any signature.
 *
 * The three scratch buffers (geometry / binning / image) are opaque: obtained through the allocator
 * callbacks in forward, handed back verbatim to backward, exactly as in the reference
 * (src/rasterize_points.cu:28-34,71-78,176-178).  Their internal layout is private (csrc/gs_layout.h).
 */
#ifndef SEGS_RASTER_H_
#define SEGS_RASTER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEGS_OK 0
#define SEGS_ERR_INVALID_ARGUMENT (-1)
#define SEGS_ERR_UNSUPPORTED (-2)
#define SEGS_ERR_ALLOC (-3)

/* Replaces std::function<char*(size_t)> (cuda_rasterizer/rasterizer.h:37-39): must return a device
 * buffer of at least `bytes` bytes that stays valid until the matching backward has run. */
typedef char* (*segs_alloc_fn)(void* ctx, size_t bytes);

/* Human-readable text for the last non-zero status returned on this thread. */
const char* segs_last_error(void);

/* Behaviour switches of the forward entry points, per host thread; returns the previous value.
 * SEGS_RASTER_SKIP_NONPOSITIVE_OPACITY: a Gaussian whose opacity is <= 0 is treated as absent (radius 0, not binned).
 * Such a Gaussian never contributes (alpha >= 1/255 is impossible), so image and gradients are unchanged; only `radii`
 * and R differ.  Used with segs_neural_forward's candidate-domain outputs (segs_neural.h), where it stands for the
 * reference's boolean-mask compaction in front of the rasterizer (src/gaussian_renderer.cpp:320). */
#define SEGS_RASTER_SKIP_NONPOSITIVE_OPACITY 1u
/* SEGS_RASTER_KEEP_DEAD_INSTANCES (resident entry points only): by default the resident forward drops, while sorting, every
 * (Gaussian, tile) instance of the reference's bounding-rectangle duplication (rasterizer_impl.cu:70-111) that cannot pass
 * alpha >= 1/255 at any pixel of its tile -- the pairs the reference skips pixel by pixel (forward.cu:404-412).  Image
 * and gradients are unchanged; the per-tile lists get shorter.  With this flag the full lists are kept, as the
 * reference-shaped segs_rasterize_forward always does (its R, point_list and ranges are the reference's, bit for bit). */
#define SEGS_RASTER_KEEP_DEAD_INSTANCES 2u
/* SEGS_RASTER_GATHER_TILES_TOUCHED: take the per-Gaussian tile counts into depth order with a gather in the last depth-sort pass
 * instead of carrying them in the spare bits of the sort values (what the build does by itself above 2^26 Gaussians, where
 * fewer than six spare bits are left).  Same results; exists so that the tests keep that path alive. */
#define SEGS_RASTER_GATHER_TILES_TOUCHED 4u
/* SEGS_RASTER_TEST_NARROW_PACK (test support): carry the tile counts in only the top two bits of the sort values, so that
 * nearly every count saturates the packed field and takes the fetch-on-saturation path. */
#define SEGS_RASTER_TEST_NARROW_PACK 8u
/* SEGS_RASTER_UNFUSED_BINNING (A/B measurements and tests): fill the range table and the status words with their own kernel
 * (identify_tile_ranges) after the tile-id sort instead of inside its last scatter pass.  Same results. */
#define SEGS_RASTER_UNFUSED_BINNING 16u
/* SEGS_RASTER_TIGHT_BINNING (reference-shaped segs_rasterize_forward only; the resident entry points always do this): bin
 * what the resident forward bins -- the bounding box of the alpha >= 1/255 ellipse intersected with the reference's 3-sigma
 * square instead of the square itself (rasterizer_impl.cu:70-111, auxiliary.h:47-57), and leave out, while sorting, every
 * instance that reaches no pixel of its tile.  Image, radii and gradients are what the default gives (image bit for bit);
 * the returned num_rendered is the number of instances BINNED, not the reference's R, and point_list / ranges / n_contrib
 * inside the scratch describe the shorter lists.  Legal for a drop-in because the three scratch buffers are opaque between
 * forward and backward (src/rasterize_points.cu:28-34; rasterizer_impl.h:22-73) and R is only ever handed back to backward
 * (src/gaussian_rasterizer.cpp:60-88,120-141).  The segs_debug_unpack_* helpers describe default-mode scratch only. */
#define SEGS_RASTER_TIGHT_BINNING 32u
/* SEGS_RASTER_MFMA_MOMENTS (A/B measurements and tests; backward entry points): the tile backward forms its nine per-Gaussian sums
 * with v_mfma_f32_16x16x4_f32 (moments about the quadrant centre) instead of vector FMAs.  Same results inside the gradient
 * tolerance; slower on MI355X, where an f32 MFMA takes its cycles out of the SIMD's vector issue (DESIGN.md section 7.0,
 * tools/ubench_mfma_valu_overlap.hip).  Also switched on by the environment variable SEGS_RENDER_BWD_MFMA=1. */
#define SEGS_RASTER_MFMA_MOMENTS 64u
uint32_t segs_raster_set_flags(uint32_t flags);

/* Resident mode only, per host thread; returns the previous pointer.  When set, segs_rasterize_forward_resident also
 * stores status[0] (R) and status[3] (overflow) into this HOST-MAPPED, device-accessible array of 4 words (pinned memory:
 * hipHostMalloc / a pinned torch tensor) from inside its last binning kernel, so the host can learn them after an event
 * wait without a device-to-host copy being enqueued in the stream every iteration. */
uint32_t* segs_raster_set_status_mirror(uint32_t* host_mapped_status);

/* Scratch sizes (reference: CudaRasterizer::required<GeometryState|ImageState|BinningState>,
 * cuda_rasterizer/rasterizer_impl.h:66-72).  segs_binning_bytes(n) is also the temp size of segs_sort_pairs(n); the
 * forward's own binning request (through the callback) additionally covers a P-sized depth sort. */
size_t segs_geometry_bytes(int P);
size_t segs_image_bytes(int width, int height);
size_t segs_binning_bytes(int num_rendered);

/* CudaRasterizer::Rasterizer::forward (cuda_rasterizer/rasterizer.h:36-59, rasterizer_impl.cu:198-336).
 * Exactly one of (shs | colors_precomp) and one of (scales+rotations | cov3D_precomp) is non-null, as
 * GaussianRasterizer::forward enforces (src/gaussian_rasterizer.cpp:175-181).  `radii` (P ints) may be
 * null.  On success *num_rendered receives R (this call synchronises `stream` once to size the binning
 * buffer, like the reference's cudaMemcpy at rasterizer_impl.cu:281).  P == 0 is legal (R = 0). */
int segs_rasterize_forward(segs_alloc_fn geometry_alloc, void* geometry_ctx,
                           segs_alloc_fn binning_alloc, void* binning_ctx,
                           segs_alloc_fn image_alloc, void* image_ctx,
                           int P, int D, int M,
                           const float* background, int width, int height,
                           const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, float scale_modifier,
                           const float* rotations, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                           float tan_fovx, float tan_fovy, int prefiltered,
                           float* out_color, int* radii, void* stream, int* num_rendered);

/* CudaRasterizer::Rasterizer::backward (cuda_rasterizer/rasterizer.h:80-108, rasterizer_impl.cu:397-490).
 * Every output row is written by this call (the caller does not have to zero them, unlike the reference's
 * torch::zeros at src/rasterize_points.cu:149-157).  dL_dconic is (P,2,2); dL_dmean2D is (P,3).  dL_dconic (an internal
 * product of the reference's backward) and, without cov3D_precomp, dL_dcov3D may be NULL: they are then not written. */
int segs_rasterize_backward(int P, int D, int M, int R,
                            const float* background, int width, int height,
                            const float* means3D, const float* shs, const float* colors_precomp,
                            const float* scales, float scale_modifier, const float* rotations,
                            const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                            const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                            char* geom_buffer, char* binning_buffer, char* image_buffer,
                            const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                            float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                            float* dL_dscale, float* dL_drot, void* stream);

/* CudaRasterizer::Rasterizer::visible_filter (cuda_rasterizer/rasterizer.h:62-77, rasterizer_impl.cu:339-393):
 * radii only.  The reference also allocates geometry and image scratch it never reads; not needed here. */
int segs_visible_filter(int P, int M, int width, int height, const float* means3D, const float* scales,
                        float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, float tan_fovx, float tan_fovy,
                        int prefiltered, int* radii, void* stream);

/* The same with the scales given as the model stores them: `scaling_log` holds rows of `stride` floats whose first three are
 * log-scales (GaussianModel::_scaling, exp applied as get_scaling does, src/gaussian_renderer.cpp:150-152), rotations already
 * normalised, scale_modifier 1 -- prefilter_voxel's call without the intermediate exp(...)[:, :3] tensor and its kernels. */
int segs_visible_filter_log_scales(int P, int width, int height, const float* means3D, const float* scaling_log, int stride,
                                   const float* rotations, const float* viewmatrix, const float* projmatrix, float tan_fovx,
                                   float tan_fovy, int* radii, void* stream);

/* CudaRasterizer::Rasterizer::markVisible (cuda_rasterizer/rasterizer.h:29-34, rasterizer_impl.cu:141-153).
 * `present` is P bytes (bool). */
int segs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present, void* stream);

/* CudaRasterizer::Rasterizer::project2_image (cuda_rasterizer/rasterizer.h:110-133, rasterizer_impl.cu:494-585):
 * preprocess only; copies the per-Gaussian pixel centres (P,2) and colours (P,3) out. */
int segs_project2_image(int P, int D, int M, int width, int height, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales,
                        float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                        float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* points_image,
                        int* radii, void* stream);

/* ---- Parity-test support: expand the private scratch into the reference's state arrays
 * (GeometryState / BinningState / ImageState, cuda_rasterizer/rasterizer_impl.h:30-66). Any output may be null.
 * For the scratch of the reference-shaped entry points (segs_rasterize_forward); the resident entry points do not keep
 * tiles_touched / the 64-bit keys' depth bits in the geometry scratch. */
int segs_debug_unpack_geometry(const char* geom_buffer, int P, const int* radii, float* means2D /*P,2*/,
                               float* conic_opacity /*P,4*/, float* depths /*P*/, uint32_t* tiles_touched /*P*/,
                               uint32_t* point_offsets /*P*/, float* rgb /*P,3*/, void* stream);
int segs_debug_unpack_binning(const char* binning_buffer, const char* geom_buffer, int P, int R, int width, int height,
                              uint64_t* keys_sorted /*R*/, uint32_t* point_list /*R*/, void* stream);
/* The sorted instance values as the tile kernels read them: Gaussian index | (quadrants of the 16x16 tile this instance can
 * reach) << 28 (no reference counterpart; used by tools/tile_stats.py to measure what the tile kernels iterate over). */
int segs_debug_instance_values(const char* binning_buffer, int R, uint32_t* values /*R*/, void* stream);
int segs_debug_unpack_image(const char* image_buffer, int width, int height, uint32_t* ranges /*tiles,2*/,
                            float* final_T /*H*W*/, uint32_t* n_contrib /*H*W*/, void* stream);

/* Backward of the per-Gaussian stage alone (K12+K13) from caller-supplied dL_dmean2D (P,3) and dL_dconic
 * (P,2,2): lets a test check it bit-exactly against the oracle. */
int segs_debug_preprocess_backward(int P, int width, int height, const float* means3D, const int* radii,
                                   const float* scales, float scale_modifier, const float* rotations,
                                   const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                   float tan_fovx, float tan_fovy, const float* dL_dmean2D, const float* dL_dconic,
                                   float* dL_dmean3D, float* dL_dcov3D, float* dL_dscale, float* dL_drot, void* stream);

/* Stand-alone stable radix sort of (u64 key, u32 value) pairs on key bits [0, end_bit) -- the semantics of
 * cub::DeviceRadixSort::SortPairs as called at rasterizer_impl.cu:303-308.  `temp` must hold
 * segs_binning_bytes(n) bytes; results land in keys_out / vals_out. */
int segs_sort_pairs(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                    int n, int end_bit, char* temp, void* stream);

/* ---- Resident (steady-state) variants for training loops: NO host synchronisation and a fixed launch sequence
 * (hipGraph-capturable).  The caller owns all scratch: geom_buffer >= segs_geometry_bytes(geom_rows), image_buffer >=
 * segs_image_bytes(W,H), binning_buffer >= segs_resident_binning_bytes(geom_rows, capacity).  `geom_rows` >= P is the row
 * count the geometry buffer was sized (and zero-filled) for: its carve-up is keyed by it, NOT by P, so a caller may
 * rasterize any P <= geom_rows rows from call to call (a map that grows and shrinks inside pre-sized buffers) without the
 * self-cleaned accumulator rows moving under it.  `capacity` bounds the number of
 * (Gaussian, tile) instances.  `status` is 4 device words: [0] = num_rendered R (instances binned), [1] = instances live after the dead ones were dropped, [3] = 1 if R exceeded the capacity
 * (outputs of that call are then meaningless; re-run with a larger capacity).  geom_buffer must be ZERO-FILLED before its
 * first use: the resident backward does not clear the per-Gaussian gradient accumulators inside it with a fill per call,
 * it writes zeros back over each row it consumes (a buffer that is clean stays clean).  The reference has no counterpart: its
 * forward always blocks on a device-to-host copy of R (rasterizer_impl.cu:281). */
size_t segs_resident_binning_bytes(int P, int capacity);
int segs_rasterize_forward_resident(char* geom_buffer, char* binning_buffer, char* image_buffer, int capacity,
                                    int geom_rows, int P, int D, int M, const float* background, int width, int height,
                                    const float* means3D, const float* shs, const float* colors_precomp,
                                    const float* opacities, const float* scales, float scale_modifier,
                                    const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                    float* out_color, int* radii, uint32_t* status, void* stream);
int segs_rasterize_backward_resident(char* geom_buffer, char* binning_buffer, char* image_buffer, int capacity,
                                     int geom_rows, int P, int D, int M, const float* background, int width, int height,
                                     const float* means3D, const float* shs, const float* scales, float scale_modifier,
                                     const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                     const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                                     const int* radii, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                                     float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                                     float* dL_dsh, float* dL_dscale, float* dL_drot, void* stream);

/* ---- A producer that projects its own Gaussians (segs_neural_forward_projected, segs_neural.h).
 * The resident forward is K1 (per-Gaussian projection, cuda_rasterizer/forward.cu:155-256) + binning + tile kernel.  A
 * kernel that GENERATES the Gaussians can run K1's arithmetic on them while they are in registers and leave K1's outputs
 * where the rest of the forward expects them; segs_resident_projection_targets says where that is for one set of resident
 * buffers (same arguments as segs_rasterize_forward_resident; `radii` NULL = the buffer's internal array), and
 * segs_rasterize_forward_resident_projected is the forward without K1: binning and tile kernel over what the producer left.
 * Per Gaussian row i < P the producer writes  radii[i] (0 = culled),  tiles_touched[i],  depth_keys[i] (bits of the view
 * depth, 0xFFFFFFFF when no tile is touched)  and, for radii[i] > 0, the 64-byte record records[16 i ..] -- all exactly as
 * preprocess_fwd_kernel would; it resets tile_ranges[t] = {0xFFFFFFFF, 0} for t < num_tiles and sets *depth_overflow = 1
 * when a binned depth leaves the sort's key range.  `flags` carries the tight-rectangle switch for the record builder. */
typedef struct segs_projection_targets {
  float* records;
  int* radii;
  uint32_t* tiles_touched;
  uint32_t* depth_keys;
  uint32_t* tile_ranges;      /* num_tiles x {start, end} */
  uint32_t* depth_overflow;
  int num_tiles;
  uint32_t flags;
} segs_projection_targets;
int segs_resident_projection_targets(char* geom_buffer, char* binning_buffer, char* image_buffer, int capacity, int geom_rows,
                                     int P, int width, int height, int* radii, uint32_t* status, segs_projection_targets* out);
int segs_rasterize_forward_resident_projected(char* geom_buffer, char* binning_buffer, char* image_buffer, int capacity,
                                              int geom_rows, int P, const float* background, int width, int height,
                                              float* out_color, uint32_t* status, void* stream);

/* ---- Measurement support (bench.py): per-kernel timing with HIP events recorded on the launch stream.
 * kernel_mask bit i selects kernel id i (ids 0..segs_profile_kernel_count()-1, names via
 * segs_profile_kernel_name).  segs_profile_end() synchronises the recorded events and accumulates;
 * segs_profile_query() then returns total milliseconds and launch count per kernel id. */
int segs_profile_begin(unsigned kernel_mask);
int segs_profile_end(void);
int segs_profile_kernel_count(void);
const char* segs_profile_kernel_name(int id);
int segs_profile_query(int id, double* total_ms, long* launches);

#ifdef __cplusplus
}
#endif
#endif /* SEGS_RASTER_H_ */
