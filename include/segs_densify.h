/* segs_densify.h -- C ABI of the anchor statistics and anchor growing (part of libsegs_raster.so).
 *
 * Replaces, for the candidate-domain layout of segs_neural.h (every (anchor, offset) pair keeps slot a*n_offsets + k):
 *   GaussianModel::training_statis  src/gaussian_model.cpp:1459-1503  -> segs_training_statis (one fused kernel per
 *       iteration instead of ~15 boolean-mask index_put_ round trips, each a host synchronisation);
 *   one level of GaussianModel::anchor_growing  :1559-1699            -> segs_anchor_growing_level: candidate selection,
 *       voxel quantisation, sorted unique (the reference's at::unique_dim), removal of voxels that already hold an anchor
 *       (the reference compares every unique voxel with every anchor in chunks of 4096, O(N*A); here both sides are
 *       packed into 63-bit keys, radix-sorted and merged by binary search), per-voxel feature maximum (torch_scatter's
 *       scatter_max, taken over the sorted run of each voxel) and the new anchor rows.
 * The tensor bookkeeping around it (concatenating the new rows, resetting counters, pruning by boolean mask, extending
 * the Adam state) is done by the caller on device tensors: segs-slam_amd/densify.py mirrors adjust_anchor :1701-1762.
 * The reference's random keep mask (torch::rand_like, :1568) is an input (`rand`), so that results are reproducible.
 */
#ifndef SEGS_DENSIFY_H_
#define SEGS_DENSIFY_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Per-iteration statistics.  neural_opacity (A*no) and visible_radii (A, may be NULL = all visible) as in segs_neural.h;
 * radii (A*no) and dL_dmean2D (A*no,3) from the rasterizer (radii > 0 is the reference's update_filter; only
 * dL_dmean2D[:, :2] is used).  Accumulators: opacity_accum (A), anchor_demon (A), offset_gradient_accum (A*no),
 * offset_denom (A*no), all fp32 like the reference's. */
int segs_training_statis(int A, int n_offsets, const float* neural_opacity, const int* visible_radii, const int* radii,
                         const float* dL_dmean2D, float* opacity_accum, float* anchor_demon, float* offset_gradient_accum,
                         float* offset_denom, void* stream);
/* Same, dropped on the device when *skip_flag != 0 (the overflow word status[3] of segs_rasterize_forward_resident): the
 * training loop then needs no host synchronisation per iteration to keep an invalid pass out of the statistics. */
int segs_training_statis_guarded(int A, int n_offsets, const float* neural_opacity, const int* visible_radii, const int* radii,
                                 const float* dL_dmean2D, float* opacity_accum, float* anchor_demon, float* offset_gradient_accum,
                                 float* offset_denom, const uint32_t* skip_flag, void* stream);

/* Scratch for one growing level over A anchors with at most n_candidates = A_init * n_offsets candidate slots. */
size_t segs_anchor_growing_temp_bytes(int A, int n_candidates);

/* One level of anchor_growing.  Candidates are the first A_init*n_offsets slots (anchors appended by earlier levels of
 * the same adjust_anchor call never spawn, :1572-1581); all A anchors block their voxel.  grads (A_init*no) = the
 * per-offset mean gradient norm, offset_mask (A_init*no bytes, 0/1), rand (A_init*no).  A slot is a candidate iff
 * grads >= threshold && offset_mask && rand > rand_threshold.  cur_size = voxel_size * size_factor.
 * Outputs: new_anchor (max_new,3), new_feat (max_new,feat_dim) in the reference's order (lexicographically sorted voxel
 * coordinates), n_new = 1 device int (number of new anchors; if it exceeds max_new only max_new rows are written).
 * feat_dim must be 32. */
int segs_anchor_growing_level(int A, int A_init, int n_offsets, int feat_dim, const float* anchor, const float* offset,
                              const float* scaling_log, const float* anchor_feat, const float* grads,
                              const uint8_t* offset_mask, const float* rand, float threshold, float rand_threshold,
                              float cur_size, int max_new, float* new_anchor, float* new_feat, int* n_new, char* temp,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGS_DENSIFY_H_ */
