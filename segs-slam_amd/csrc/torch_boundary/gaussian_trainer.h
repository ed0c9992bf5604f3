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
// prefilter and the tensor bookkeeping of adjust_anchor.  This is the C++ twin of
// segs-slam_amd/neural_gaussians.py::ScaffoldTrainerStep: the iteration itself, the densification schedule of
// trainForOneIteration (training_statis / adjust_anchor, src/gaussian_mapper.cpp:957-968 -> anchor_densifier.h) and the
// keyframe-parallel exchange (keyframe_exchange.h); tests/test_cpp_trainer.py runs both twins on the same model and compares
// losses, parameters and the map after an adjust_anchor iteration, and two C++ ranks against each other.
#pragma once
#include <algorithm>
#include <torch/torch.h>

#include <torch/csrc/distributed/c10d/Backend.hpp>

#include <functional>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/segs_neural.h"
#include "../../../include/segs_train.h"

namespace segs_host {

class AnchorDensifier;
class KeyframeExchange;

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

// Anchors + MLPs in one flat parameter bucket (plus same-shaped gradient and Adam-moment buckets): the reference's six
// anchor tensors and its Sequential MLPs (src/gaussian_model.cpp:60-150, 327-381) as the fused Adam's operand.  The anchor
// segments are laid out for `capacity` >= A anchors so that densification appends and prunes rows in place; views cover
// the first A rows.  C++ twin of neural_gaussians.py::ScaffoldModel.
struct ScaffoldModelState {
  ScaffoldModelState(int64_t num_anchors, const ScaffoldDims& dims, torch::Device device, int64_t capacity = 0);
  void reserve(int64_t capacity);   // grow the buckets, keeping the first A rows of every segment and the MLP block
  int64_t width(const std::string& name) const;
  torch::Tensor view(const torch::Tensor& bucket, const std::string& name, int64_t rows = -1) const;
  torch::Tensor param(const std::string& name, int64_t rows = -1) const { return view(params, name, rows); }
  torch::Tensor grad(const std::string& name, int64_t rows = -1) const { return view(grads, name, rows); }
  torch::Tensor mlp_params() const { return params.slice(0, mlp_offset, n_params); }
  float* seg_ptr(const torch::Tensor& bucket, const std::string& name) const { return bucket.data_ptr<float>() + seg_offset.at(name); }
  // (offset, count, lr slot) of the Adam groups in the reference's order (src/gaussian_model.cpp:632-690): four anchor groups,
  // then the MLP groups; `kind` of an MLP group: 0 opacity, 1 cov, 2 color, 3 appearance, 4 feature bank
  std::vector<std::pair<int64_t, int64_t>> anchor_groups() const;

  ScaffoldDims dims;
  segs_neural_dims cdims;
  torch::Device dev;
  int64_t A = 0, capacity = 0, n_params = 0, mlp_offset = 0, mlp_total = 0;
  std::map<std::string, int64_t> seg_offset;
  std::vector<std::pair<int64_t, int64_t>> mlp_group_rel;   // (offset inside the MLP block, count)
  std::vector<int> mlp_group_kind;
  torch::Tensor params, grads, exp_avg, exp_avg_sq, rotation, opacity;

 private:
  void allocate(int64_t capacity);
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
                      double spatial_lr_scale = 1.0, int64_t capacity = 0);
  ~GaussianTrainerStep();

  ScaffoldModelState& model() { return model_; }
  // views into the flat parameter bucket (the Adam operand): "anchor" (A,3), "offset" (A,n_offsets,3),
  // "anchor_feat" (A,32), "scaling" (A,6); the MLP block in the order of segs_neural_param_layout
  torch::Tensor param(const std::string& name) { return model_.param(name); }
  torch::Tensor mlp_params() { return model_.mlp_params(); }
  torch::Tensor params_flat() { return model_.params; }
  torch::Tensor grads_flat() { return model_.grads; }

  // training_statis / adjust_anchor on the schedule of trainForOneIteration (src/gaussian_mapper.cpp:961-968); the random keep
  // masks come from a CPU generator seeded identically on every rank.  The step does not own the densifier.
  void enable_densification(AnchorDensifier* densifier, uint64_t seed);
  // keyframe-parallel training: N > 1 ranks each render their own keyframe, the gradient bucket is exchanged before the
  // optimizer (keyframe_exchange.h).  `pg` null = single process.
  void set_process_group(c10::intrusive_ptr<c10d::Backend> pg, bool sharded_optimizer = true, bool single_rank_collectives = false);
  int world() const;
  int rank() const;

  // One iteration on one keyframe (the caller picks keyframe (iteration * world + rank) mod n, or the mapper's walk); returns
  // the L1/SSIM part of the loss as a 1-element device tensor (the scaling regulariser's value is in scaling_reg()).
  // Nothing in it waits for the device except the first, calibrating pass and the adjust_anchor iterations.
  torch::Tensor trainingOnce(const KeyframeView& kf, const torch::Tensor& gt_image);

  // The mapper's frequency regulariser (src/gaussian_mapper.cpp:930-945; Mapper.* keys of the configuration): between
  // iterations `start` and `until` (exclusive, as :938) loss += lambda_high * multi_scale_loss(image, gt, scales) -- or
  // high_frequency_loss when not multi_resolution -- through the plan entry points of segs_train.h (hipFFT inside the library,
  // |FFT(gt)| cached per target tensor).  low_freq_loss has a zero gradient in the reference and
  // lambda_low = 0 in every shipped configuration: not evaluated here.
  void enable_frequency_regularization(float lambda_high, const std::vector<float>& scales, int64_t start, int64_t until,
                                       bool multi_resolution = true);
  torch::Tensor frequency_loss() { return freq_value_; }   // value of the regulariser in the last iteration (1 device float)
  // how many targets' |FFT| tables (and the targets themselves) the step keeps: 6.5 + 9.8 MB each at 1200x680
  void set_frequency_target_cache(size_t max_targets) { freq_.max_targets = max_targets ? max_targets : 1; }
  size_t frequency_targets_cached() const { return freq_.targets.size(); }

  // SURVEY 8f n3 (default on): once the rasterizer's resident scratch is calibrated, the neural forward runs the per-Gaussian
  // projection (K1) and prefilter_voxel itself (segs_neural_forward_projected) and the rasterizer starts at the binning
  // (segs_rasterize_forward_resident_projected).  Same image, radii and gradients bit for bit; colours and opacities of the
  // candidates are then not materialised as arrays.
  void set_fuse_projection(bool on) { fuse_projection_ = on; }
  // Run an iteration the device dropped (a resident-capacity overflow on ANY rank) again before the next one (trainingOnce; on by
  // default).  finish(): resolve the LAST iteration's word and redo it if needed -- call once after the last trainingOnce of a run.
  void set_redo_dropped_steps(bool on) { redo_dropped_steps_ = on; }
  int64_t redone_steps() const { return redone_steps_; }
  void finish();
#ifdef SEGS_TESTING
  // Test support, compiled in only for the drivers under tests/ (trainer_test is built with -DSEGS_TESTING; the class layout does
  // not depend on the macro).  dL/dimage is multiplied by this (H,W) mask before the raster backward (the parity tests blank the
  // pixels whose compositing decisions sit on a threshold, on both sides); a callback that sees the gradient bucket exactly as the
  // optimizer receives it (after the exchange, before Adam clears it); and a way to make the next resident forward overflow
  // (capacity = last instance count / divisor; the buffers stay as large as they are).
  void set_image_gradient_mask(const torch::Tensor& mask_hw) { dL_mask_ = mask_hw; }
  void set_on_gradients(std::function<void(const torch::Tensor&)> fn) { on_gradients_ = std::move(fn); }
  void debug_shrink_capacity(int divisor) { resolve_status(); if (capacity_ > 0) capacity_ = std::max(num_rendered_ / divisor, 1024); }
#endif

  torch::Tensor image() { return out_color_; }
  torch::Tensor scaling_reg() { return scaling_reg_; }
  int64_t iteration() const { return iteration_; }
  int64_t steps_taken();        // optimizer steps really taken by the MLP groups (device-side count; synchronises)
  int64_t anchor_steps_taken(); // ... by the anchor groups (they skip the adjust_anchor iterations, src/gaussian_model.cpp:1677)
  // candidate-domain state of the last pass, for the statistics (anchor_densifier.h)
  torch::Tensor neural_opacity() { return neural_opacity_; }
  torch::Tensor visible_radii() { return visible_radii_; }
  torch::Tensor radii() { return radii_; }
  torch::Tensor dL_dmean2D() { return dL_dmean2D_; }
  bool last_pass_resident() const { return last_resident_; }

 private:
  struct StepCount {   // torch::optim::Adam's per-group step count, on the device: two int64 words used in turn
    torch::Tensor words;
    int calls = 0;
  };
  void learning_rates(int64_t it, std::vector<double>& lr_of_group) const;
  void prefilter(const KeyframeView& kf);
  const torch::Tensor& anchor_rotations();
  void render(const KeyframeView& kf);
  bool resolve_status();
  void redo_if_dropped();
  std::function<void(const torch::Tensor&)> on_gradients_;
  torch::Tensor iteration_body(const KeyframeView& kf, const torch::Tensor& gt_image);
  void forward_backward(const KeyframeView& kf, const torch::Tensor& gt_image);
  void adam(const std::vector<segs_adam_segment>& groups, StepCount& count, const uint32_t* guard);
  void allocate_candidate_buffers();
  KeyframeExchange& exchange();

  ScaffoldModelState model_;
  ScaffoldOptimization opt_;
  int W_, H_;
  torch::Device dev_;
  float reg_weight_;
  double spatial_lr_scale_;
  int64_t iteration_ = 0;
  int64_t cand_capacity_ = 0;                                       // anchors the candidate-domain buffers are sized for
  torch::Tensor rot_normalized_;
  int64_t rot_rows_ = -1;
  StepCount mlp_count_, anchor_count_;
  bool anchor_count_split_ = false;
  AnchorDensifier* densifier_ = nullptr;
  at::Generator densify_generator_;
  c10::intrusive_ptr<c10d::Backend> pg_;
  bool sharded_optimizer_ = true, single_rank_collectives_ = false;
  std::unique_ptr<KeyframeExchange> ex_;

  // candidate-domain buffers of the neural Gaussians and their gradients
  torch::Tensor means3D_, colors_, opacity_, scales_, rotations_, neural_opacity_, neural_temp_, visible_radii_;
  torch::Tensor g_means3D_, g_colors_, g_opacity_, g_scales_, g_rotations_, dL_dmean2D_;
  // rasterizer state
  torch::Tensor out_color_, radii_, bg_, geom_, binning_, img_, geom_r_, binning_r_, img_r_, status_, status_host_;
  int num_rendered_ = 0, capacity_ = 0;
  bool last_resident_ = false, status_pending_ = false, fuse_projection_ = true;
  bool redo_dropped_steps_ = true, have_last_ = false;
  int64_t redone_steps_ = 0;
  KeyframeView last_kf_;
  torch::Tensor last_gt_;
  void* status_event_ = nullptr;
  // loss
  torch::Tensor loss_temp_, loss_out_, dL_dimage_, scaling_reg_, dL_mask_, freq_value_;
  struct FreqReg {
    bool on = false, multi = true;
    float lambda_high = 0.f;
    std::vector<float> scales;
    int64_t start = 0, until = 0;
    segs_freq_plan* plan = nullptr;
    // |FFT(gt)| tables per target tensor (a keyframe's image does not change).  An entry is keyed on the target's address AND
    // its version counter and KEEPS THE TARGET ALIVE: the caching allocator hands a freed image's address to the next one (the
    // reference's mapper makes a fresh `gt_image * mask_rgb` every iteration, src/gaussian_mapper.cpp:921), and an in-place
    // refresh of a staging buffer bumps the version.  Least recently used entries go first (set_frequency_target_cache).
    struct Target { torch::Tensor gt, table; uint32_t version; };
    std::list<Target> targets;       // most recently used first
    size_t max_targets = 64;
  } freq_;
};

}  // namespace segs_host
