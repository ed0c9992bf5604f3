// gaussian_trainer.h -- C++/LibTorch-ROCm host of one mapper / trainer iteration over the C ABI.
//
// What the reference does per iteration with ~200 ATen kernels and LibTorch autograd
// (GaussianTrainer::trainingOnce src/gaussian_trainer.cpp:47-117, GaussianMapper::trainForOneIteration
// src/gaussian_mapper.cpp:823-1032: updateLearningRate -> prefilter_voxel -> render (generate_neural_gaussians + rasterizer)
// -> L1 / SSIM (+ 0.01 scaling regulariser in the mapper) -> backward -> optimizer step, zero_grad), expressed as eight
// launches-worth of entry points of libsegs_raster.so:
//   segs_visible_filter -> segs_neural_forward -> segs_rasterize_forward[_resident] -> segs_l1_ssim_loss
//   -> segs_rasterize_backward[_resident] -> segs_neural_backward -> segs_adam_step_device.
// LibTorch is plumbing only: device memory (torch::Tensor), the current HIP stream and two elementwise ops of the
// prefilter.  This is the C++ twin of segs-slam_amd/neural_gaussians.py::ScaffoldTrainerStep (single rank, no
// densification); tests/test_cpp_trainer.py runs both on the same model and compares losses and parameters.
#pragma once
#include <torch/torch.h>

#include <map>
#include <string>
#include <vector>

#include "../../../include/segs_neural.h"

namespace segs_host {

// Model.* keys of cfg/gaussian_mapper/RGB-D/Replica/office0.yaml:13-26 that shape the MLPs
struct ScaffoldDims {
  int feat_dim = 32, n_offsets = 10, appearance_dim = 32;
  bool use_feat_bank = true, add_opacity_dist = false, add_cov_dist = false, add_color_dist = false;
  segs_neural_dims c() const {
    return segs_neural_dims{feat_dim, n_offsets, appearance_dim, use_feat_bank, add_opacity_dist, add_cov_dist, add_color_dist};
  }
};

// Optimization.* of cfg/gaussian_mapper/RGB-D/Replica/office0.yaml:76-137 used by the step
struct ScaffoldOptimization {
  double lambda_dssim = 0.2;
  double position_lr_init = 0.0, position_lr_final = 0.0; int position_lr_max_steps = 30000;
  double offset_lr_init = 0.08, offset_lr_final = 0.0001; int offset_lr_max_steps = 30000;
  double feature_lr = 0.0010, scaling_lr = 0.005;
  double mlp_opacity_lr_init = 0.002, mlp_opacity_lr_final = 0.00002; int mlp_opacity_lr_max_steps = 30000;
  double mlp_cov_lr_init = 0.004, mlp_cov_lr_final = 0.004; int mlp_cov_lr_max_steps = 30000;
  double mlp_color_lr_init = 0.008, mlp_color_lr_final = 0.00005; int mlp_color_lr_max_steps = 30000;
  double mlp_featurebank_lr_init = 0.01, mlp_featurebank_lr_final = 0.00001; int mlp_featurebank_lr_max_steps = 30000;
  double appearance_lr_init = 0.05, appearance_lr_final = 0.0005; int appearance_lr_max_steps = 30000;
  double beta1 = 0.9, beta2 = 0.999, eps = 1e-15;   // src/gaussian_model.cpp:632-661
};

// what the step needs of a GaussianKeyframe (src/gaussian_keyframe.cpp:151-184): device tensors + scalars
struct KeyframeView {
  torch::Tensor view, proj, campos, pose7;   // (4,4) transposed layouts, (3), (t_xyz, q_wxyz)
  float tanfovx = 0.f, tanfovy = 0.f;
};

class GaussianTrainerStep {
 public:
  GaussianTrainerStep(int64_t num_anchors, const ScaffoldDims& dims, int width, int height, torch::Device device,
                      const ScaffoldOptimization& opt = ScaffoldOptimization(), float scaling_reg_weight = 0.f,
                      double spatial_lr_scale = 1.0);

  // views into the flat parameter bucket (the Adam operand): "anchor" (A,3), "offset" (A,n_offsets,3),
  // "anchor_feat" (A,32), "scaling" (A,6); the MLP block in the order of segs_neural_param_layout
  torch::Tensor param(const std::string& name);
  torch::Tensor mlp_params() { return params_.slice(0, mlp_offset_, n_params_); }
  torch::Tensor params_flat() { return params_; }
  torch::Tensor grads_flat() { return grads_; }

  // One iteration on one keyframe; returns the L1/SSIM part of the loss as a 1-element device tensor (the scaling
  // regulariser's value is in scaling_reg()).  Nothing in it waits for the device except the first, calibrating pass.
  torch::Tensor trainingOnce(const KeyframeView& kf, const torch::Tensor& gt_image);

  torch::Tensor image() { return out_color_; }
  torch::Tensor scaling_reg() { return scaling_reg_; }
  int64_t iteration() const { return iteration_; }
  int64_t steps_taken();        // optimizer steps really taken (device-side count; synchronises)
  bool last_pass_resident() const { return last_resident_; }

 private:
  void learning_rates(int64_t it, std::vector<double>& lr_of_group) const;
  void prefilter(const KeyframeView& kf);
  void render(const KeyframeView& kf);
  void resolve_status();

  ScaffoldDims dims_;
  segs_neural_dims cdims_;
  ScaffoldOptimization opt_;
  int64_t A_, P_;
  int W_, H_;
  torch::Device dev_;
  float reg_weight_;
  double spatial_lr_scale_;
  int64_t iteration_ = 0;

  // flat buckets
  int64_t n_params_ = 0, mlp_offset_ = 0, mlp_total_ = 0;
  std::map<std::string, std::pair<int64_t, int64_t>> seg_;          // name -> (offset, count)
  std::vector<std::pair<int64_t, int64_t>> mlp_group_;              // Adam groups inside the MLP block (offset, count)
  std::vector<int> mlp_group_kind_;                                 // 0 opacity, 1 cov, 2 color, 3 appearance, 4 feature bank
  torch::Tensor params_, grads_, exp_avg_, exp_avg_sq_, rotation_, rot_normalized_;
  torch::Tensor step_words_;                                        // 2 x int64, segs_adam_step_device
  int adam_calls_ = 0;

  // candidate-domain buffers of the neural Gaussians and their gradients
  torch::Tensor means3D_, colors_, opacity_, scales_, rotations_, neural_opacity_, neural_temp_, visible_radii_;
  torch::Tensor g_means3D_, g_colors_, g_opacity_, g_scales_, g_rotations_, dL_dmean2D_;
  // rasterizer state
  torch::Tensor out_color_, radii_, bg_, geom_, binning_, img_, geom_r_, binning_r_, img_r_, status_, status_host_;
  int num_rendered_ = 0, capacity_ = 0;
  bool last_resident_ = false, status_pending_ = false;
  void* status_event_ = nullptr;
  // loss
  torch::Tensor loss_temp_, loss_out_, dL_dimage_, scaling_reg_;
};

}  // namespace segs_host
