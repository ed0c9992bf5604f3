/* segs_neural.h -- C ABI of the neural-Gaussian generation step (part of libsegs_raster.so).
 *
 * Replaces GaussianRenderer::generate_neural_gaussians (reference src/gaussian_renderer.cpp:214-334) and its autograd
 * backward: Scaffold-GS anchors -> the Gaussians the rasterizer consumes.  The reference runs it as ~30 ATen kernels
 * with host synchronisations at every boolean-mask index (:228-231, :282, :320); here it is two fused kernels
 * (thread per visible anchor, MLP weights staged in LDS) plus an fp32-MFMA weight-gradient reduction, with no host
 * synchronisation: the visible-anchor list and its length stay on the device.
 *
 * Output convention ("candidate domain"): the reference compacts twice (visible anchors, then opacity > 0) and hands
 * P compacted Gaussians to the rasterizer.  Here every (anchor, offset) pair keeps its slot a*n_offsets + k in arrays
 * of A*n_offsets rows; a slot whose anchor is not visible or whose neural opacity is <= 0 carries opacity <= 0 and is
 * skipped by the rasterizer (SEGS_RASTER_SKIP_NONPOSITIVE_OPACITY, segs_raster.h).  Relative order of the surviving
 * Gaussians equals the reference's compacted order, so per-tile lists, image and gradients are the same; the compacted
 * tensors of the reference are `array[mask]`.
 *
 * MLP parameters live in ONE flat fp32 block in the order of the reference's Adam groups
 * (src/gaussian_model.cpp:654-690) with torch::nn::Linear's [out][in] row-major weights:
 *   mlp_opacity {0.weight [32][35+od], 0.bias [32], 2.weight [10][32], 2.bias [10]},
 *   mlp_cov     {0.weight [32][35+cd], 0.bias, 2.weight [70][32], 2.bias [70]},
 *   mlp_color   {0.weight [32][35+kd+app], 0.bias, 2.weight [30][32], 2.bias [30]},
 *   mlp_apperance {0.weight [app][7], 0.bias [app]}                      (appearance_dim > 0),
 *   mlp_feature_bank {0.weight [32][4], 0.bias, 2.weight [3][32], 2.bias [3]}   (use_feat_bank)
 * (constructor src/gaussian_model.cpp:61-98); segs_neural_param_layout reports the offsets.
 */
#ifndef SEGS_NEURAL_H_
#define SEGS_NEURAL_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct segs_neural_dims {
  int feat_dim;         /* 32 (the only supported width, Model.feat_dim of the cfg/gaussian_mapper yaml files) */
  int n_offsets;        /* 10 */
  int appearance_dim;   /* 0 .. 64 */
  int use_feat_bank;    /* 0 / 1 */
  int add_opacity_dist; /* 0 / 1: the MLP also sees the anchor-camera distance */
  int add_cov_dist;
  int add_color_dist;
} segs_neural_dims;

#define SEGS_NEURAL_MAX_TENSORS 18

/* Behaviour switches of segs_neural_backward, per host thread; returns the previous value.
 * SEGS_NEURAL_ONE_KERNEL_BACKWARD (A/B measurements and tests): run a model WITHOUT feature bank through the one-kernel
 * backward the feature-bank model uses (chain and weight gradients in one wave) instead of the chain-wave / gradient-wave
 * pairs.  Same gradients up to the summation order of the weight-gradient partials. */
#define SEGS_NEURAL_ONE_KERNEL_BACKWARD 1u
uint32_t segs_neural_set_flags(uint32_t flags);

/* Offsets and element counts of the parameter tensors inside the flat block, in the order listed above.
 * Any of offsets/counts/ntensors/total may be NULL. */
int segs_neural_param_layout(const segs_neural_dims* dims, int64_t* offsets, int64_t* counts, int* ntensors, int64_t* total);

/* Bytes of device scratch shared by forward and backward for A anchors (visible list, operand images of the MLP weights,
 * per-workgroup partial sums of the weight gradients; per-anchor scratch rows only for the feature bank's two Linears).
 * A <= 8 000 000 (the kernels index with 32-bit element offsets; SEGS_ERR_UNSUPPORTED beyond). */
size_t segs_neural_temp_bytes(const segs_neural_dims* dims, int A);

/* Forward.  anchor (A,3), offset (A,n_offsets,3), anchor_feat (A,32), scaling_log (A,6) [the stored _scaling; exp is
 * applied here as get_scaling does], visible_radii (A) int or NULL: anchor a is used iff visible_radii[a] > 0 (the
 * output of segs_visible_filter = prefilter_voxel, src/gaussian_renderer.cpp:131-199); camera_center (3) and pose7
 * = (t_xyz, q_wxyz) (:258-261) are DEVICE arrays.  Outputs, all A*n_offsets rows: means3D (.,3), colors (.,3),
 * opacity (.), scales (.,3), rotations (.,4), neural_opacity (.) [tanh output; 0 for slots of invisible anchors]. */
int segs_neural_forward(const segs_neural_dims* dims, int A, const float* anchor, const float* offset,
                        const float* anchor_feat, const float* scaling_log, const int* visible_radii,
                        const float* mlp_params, const float* camera_center, const float* pose7, float* means3D,
                        float* colors, float* opacity, float* scales, float* rotations, float* neural_opacity,
                        char* temp, void* stream);

/* Forward that also runs the rasterizer's per-Gaussian stage (K1) on the candidates it generates and writes K1's outputs
 * into the resident rasterizer buffers `targets` describes (segs_raster.h: segs_resident_projection_targets); follow it with
 * segs_rasterize_forward_resident_projected.  The candidate's colour and opacity then exist only inside its 64-byte record:
 * there are no `colors` / `opacity` outputs.  means3D, scales, rotations (what the rasterizer backward re-reads) and
 * neural_opacity are written as by segs_neural_forward; rows of candidates with neural opacity <= 0 and of invisible anchors
 * get radius 0 (src/gaussian_renderer.cpp:279,320: the reference compacts them away).  viewmatrix / projmatrix (4x4,
 * column-major as the rasterizer takes them) are DEVICE arrays.  Image, radii and every gradient equal those of
 * segs_neural_forward + segs_rasterize_forward_resident bit for bit.
 * anchor_rotations (A,4; normalised) non-NULL: prefilter_voxel (src/gaussian_renderer.cpp:131-199) is folded in as well --
 * visible_radii (A) is then an OUTPUT, filled with what segs_visible_filter_log_scales(anchor, scaling_log, 6,
 * anchor_rotations, ...) gives for this camera, and anchor a is used iff that radius is > 0.  NULL: visible_radii is the
 * input segs_neural_forward takes (or NULL for "every anchor"). */
struct segs_projection_targets;
int segs_neural_forward_projected(const segs_neural_dims* dims, int A, const float* anchor, const float* offset,
                                  const float* anchor_feat, const float* scaling_log, int* visible_radii,
                                  const float* anchor_rotations, const float* mlp_params, const float* camera_center,
                                  const float* pose7, float* means3D, float* scales, float* rotations, float* neural_opacity,
                                  const struct segs_projection_targets* targets, const float* viewmatrix,
                                  const float* projmatrix, int width, int height, float tan_fovx, float tan_fovy,
                                  float scale_modifier, char* temp, void* stream);

/* Backward of the above for the same inputs and the same `temp` (the visible list of the forward call is reused).
 * dL_d{means3D,colors,opacity,scales,rotations}: candidate-domain gradients as produced by the rasterizer backward;
 * slots with neural opacity <= 0 are ignored (the reference's mask index passes no gradient to them).
 * Gradients are ACCUMULATED (+=) into dL_danchor (A,3), dL_doffset (A,n_offsets,3), dL_dfeat (A,32),
 * dL_dscaling_log (A,6) and dL_dmlp_params (flat block): the caller zeroes them (segs_adam_step's zero_grad does).
 * scaling_reg_weight != 0 adds the mapper's scaling regulariser to the loss being differentiated,
 *     scaling_reg_weight * mean over the kept Gaussians of prod(scaling)      (src/gaussian_mapper.cpp:926-928, weight 0.01),
 * i.e. scaling_reg_weight / P * prod / scaling_c on top of dL_dscales; scaling_reg_out (1 device float or NULL) receives
 * the value of that term. */
int segs_neural_backward(const segs_neural_dims* dims, int A, const float* anchor, const float* offset,
                         const float* anchor_feat, const float* scaling_log, const float* mlp_params,
                         const float* camera_center, const float* pose7, const float* dL_dmeans3D,
                         const float* dL_dcolors, const float* dL_dopacity, const float* dL_dscales,
                         const float* dL_drotations, float* dL_danchor, float* dL_doffset, float* dL_dfeat,
                         float* dL_dscaling_log, float* dL_dmlp_params, float scaling_reg_weight, float* scaling_reg_out,
                         char* temp, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGS_NEURAL_H_ */
