// kernels.h -- declarations of the device kernels (defined in preprocess.hip, binning.hip, render.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gs_layout.h"

namespace segs {

// ---- capi.hip: error text behind segs_last_error(), shared by all translation units
int set_error(int code, const char* what);
int set_hip_error(hipError_t e, const char* where);


// ---- preprocess.hip
__global__ void preprocess_fwd_kernel(
    int P, const float* __restrict__ means3D, const float* __restrict__ scales, float mod,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ colors,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y, uint32_t gx, uint32_t gy,
    int* __restrict__ radii, float* __restrict__ rec, BinInfo* __restrict__ bin, uint32_t* __restrict__ block_sums,
    uint32_t* __restrict__ depth_range, const float* __restrict__ shs, int D, int M, const float* __restrict__ cam_pos,
    uint32_t* __restrict__ clamped, uint32_t flags, uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ depth_vals,
    uint2* __restrict__ ranges, int num_tiles, uint32_t* __restrict__ depth_overflow, uint32_t* __restrict__ touched_dense);
// Resident depth sort: every binned Gaussian has view depth > 0.2 (auxiliary.h:155), so its float bits exceed those of
// 0.2f; 27 bits above that (16 binades: depths below 13 107.2) are sorted in three 9-bit passes.
constexpr uint32_t DEPTH_KEY_MIN = 0x3E4CCCCDu;   // bits of 0.2f
constexpr int DEPTH_KEY_BITS = 27;

__global__ void visible_filter_kernel(
    int P, const float* __restrict__ means3D, const float* __restrict__ scales, float mod,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, int W, int H,
    float tan_fovx, float tan_fovy, float focal_x, float focal_y, uint32_t gx, uint32_t gy, int* __restrict__ radii,
    int log_scale_stride);

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ viewmatrix,
                                    uint8_t* __restrict__ present);

__global__ void preprocess_bwd_kernel(
    int P, const float* __restrict__ means3D, const int* __restrict__ radii, const float* __restrict__ scales,
    const float* __restrict__ rotations, float mod, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ view, const float* __restrict__ proj, float h_x, float h_y, float tan_fovx, float tan_fovy,
    float* __restrict__ gacc, float img_w, float img_h,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D,
    float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscale, float* __restrict__ dL_drot, int clean_gacc);
__global__ void sh_backward_kernel(int P, const float* __restrict__ means3D, const int* __restrict__ radii,
                                   const float* __restrict__ shs, int D, int M, const float* __restrict__ cam_pos,
                                   const uint32_t* __restrict__ clamped, const float* __restrict__ dL_dcolor,
                                   float* __restrict__ dL_dmean3D, float* __restrict__ dL_dsh);

// ---- binning.hip
__global__ void scan_block_sums_kernel(uint32_t* __restrict__ block_sums, int nblocks, const uint32_t* __restrict__ depth_range,
                                       uint32_t* __restrict__ total);
__global__ void ordered_offsets_kernel(int P, const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ incl,
                                       uint32_t* __restrict__ total_out, const uint32_t* __restrict__ ng_dev,
                                       uint32_t* __restrict__ first_owner, uint32_t owner_entries);
__global__ void duplicate_with_keys_kernel(int P, int R, const float* __restrict__ rec,
                                           const uint32_t* __restrict__ order, const uint32_t* __restrict__ incl,
                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t gx,
                                           const uint32_t* __restrict__ n_dev, int mark_dead, const uint32_t* __restrict__ ng_dev,
                                           const uint32_t* __restrict__ first_owner);
// Range table entry of a tile without instances: {RANGE_EMPTY_START, 0}.  The last tile-id pass fills the table with
// atomicMin / atomicMax (radix_scatter_kernel), which needs this start value; every reader takes start >= end as empty, and
// segs_debug_unpack_image hands out the reference's {0, 0} (rasterizer_impl.cu:310).
constexpr uint32_t RANGE_EMPTY_START = 0xFFFFFFFFu;
__global__ void normalize_ranges_kernel(int num_tiles, uint2* __restrict__ ranges);
constexpr uint32_t PREPROCESS_TIGHT_RECT = 0x80000000u;   // internal flag bit of preprocess_fwd_kernel (resident forward)
constexpr uint32_t DEAD_KEY = 0xFFFFFFFFu;   // tile-id key of an instance that reaches no quadrant of its tile
template <typename K, int BITS>   // K = uint32_t (the pipeline's own sorts) or uint64_t (segs_sort_pairs); BITS per digit: 8 or 9
__global__ void radix_count_kernel(const K* __restrict__ keys, int n, int shift, uint32_t dmin, int dbits,
                                   uint32_t* __restrict__ tile_prefix, uint32_t* __restrict__ chunk_hist, int nblocks, int nchunks,
                                   const uint32_t* __restrict__ n_dev, int drop_dead, int chunk_tiles, int nbits);
__global__ void radix_scan_kernel(uint32_t* __restrict__ block_hist, int nblocks, uint32_t* __restrict__ digit_totals);
template <typename K, int BITS, bool AUX>
__global__ void radix_scatter_kernel(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                     K* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int n, int shift,
                                     uint32_t dmin, int dbits, const uint32_t* __restrict__ tile_prefix, const uint32_t* __restrict__ chunk_prefix,
                                     const uint32_t* __restrict__ digit_totals, int nblocks, int nchunks,
                                     const uint32_t* __restrict__ n_dev, int drop_dead, uint32_t* __restrict__ n_live_out,
                                     const uint32_t* __restrict__ aux_in, uint32_t* __restrict__ aux_out, int nbits, int chunk_tiles, int pack_shift,
                                     uint2* __restrict__ ranges_out, uint32_t* __restrict__ status, uint32_t* __restrict__ status_mirror,
                                     int write_keys);
__global__ void identify_tile_ranges_kernel(int L, const uint32_t* __restrict__ keys, uint2* __restrict__ ranges,
                                            const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ status,
                                            uint32_t* __restrict__ status_mirror, const uint32_t* __restrict__ n_live);

// ---- render.hip
__global__ void render_fwd_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
                                  const float* __restrict__ rec, const float* __restrict__ bg,
                                  float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color);
__global__ void render_bwd_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
                                  const float* __restrict__ rec, const float* __restrict__ bg,
                                  const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                                  const float* __restrict__ dL_dpix, float* __restrict__ gacc, uint32_t num_tiles);
// A/B partner of render_bwd_kernel: the Gaussian role's sums as v_mfma_f32_16x16x4_f32 instead of VALU FMAs (SEGS_RENDER_BWD_MFMA=1)
__global__ void render_bwd_mfma_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
                                       const float* __restrict__ rec, const float* __restrict__ bg,
                                       const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                                       const float* __restrict__ dL_dpix, float* __restrict__ gacc, uint32_t num_tiles);

// ---- debug / test support (binning.hip)
__global__ void unpack_geometry_kernel(int P, const float* __restrict__ rec, const BinInfo* __restrict__ bin,
                                       const int* __restrict__ radii, float* __restrict__ means2D,
                                       float* __restrict__ conic_opacity, float* __restrict__ depths,
                                       uint32_t* __restrict__ tiles_touched, float* __restrict__ rgb);

__global__ void rebuild_keys_kernel(int R, const uint32_t* __restrict__ tile_keys, const uint32_t* __restrict__ vals,
                                    const BinInfo* __restrict__ bin, uint64_t* __restrict__ keys64);
__global__ void make_depth_keys_kernel(int P, const BinInfo* __restrict__ bin, uint32_t dcull, uint32_t* __restrict__ keys,
                                       uint32_t* __restrict__ vals, uint2* __restrict__ ranges, int num_tiles);
__global__ void ordered_block_sums_kernel(int P, const uint32_t* touched, const uint32_t* __restrict__ order,
                                          uint32_t* __restrict__ block_sums, uint32_t* sorted_touched,
                                          const uint32_t* __restrict__ ng_dev, uint32_t* __restrict__ order_rw, int pack_shift);
__global__ void point_offsets_kernel(int P, const BinInfo* __restrict__ bin, uint32_t* __restrict__ offsets);
__global__ void strip_mask_kernel(int n, const uint32_t* __restrict__ vals, uint32_t* __restrict__ out);

}  // namespace segs
