// gaussian_trainer.cpp -- see gaussian_trainer.h.  Reference: src/gaussian_trainer.cpp:47-117 (loop body of trainingOnce),
// src/gaussian_mapper.cpp:861-1030 (the mapper's variant with the scaling regulariser), src/gaussian_model.cpp:620-690,
// 874-915 (Adam groups, learning-rate schedule), src/gaussian_renderer.cpp:131-199 (prefilter_voxel).
#include "gaussian_trainer.h"

#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <cmath>

#include "../../../include/segs_raster.h"
#include "../../../include/segs_train.h"

namespace segs_host {
namespace {

void* cur_stream(const torch::Device& d) { return (void*)c10::hip::getCurrentHIPStream(d.index()).stream(); }
void check(int status, const char* what) {
  if (status != SEGS_OK) AT_ERROR(what, " failed (", status, "): ", segs_last_error());
}
void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) AT_ERROR(what, ": ", hipGetErrorString(e));
}
float* fp(const torch::Tensor& t) { return t.data_ptr<float>(); }

// allocator callback of the reference-shaped forward (src/rasterize_points.cu:28-34): grow-only byte tensor
char* grow_cb(void* ctx, size_t n) {
  auto* t = static_cast<torch::Tensor*>(ctx);
  if ((size_t)t->numel() < n) *t = torch::empty({(int64_t)(n + n / 4 + 4096)}, t->options());
  return reinterpret_cast<char*>(t->data_ptr());
}

// getExponLrFunc (src/gaussian_model.cpp:1393-1409) with lr_delay_steps = 0
double expon_lr(int64_t step, double lr_init, double lr_final, int max_steps) {
  if (step < 0 || (lr_init == 0.0 && lr_final == 0.0)) return 0.0;
  const double t = std::min(std::max((double)step / max_steps, 0.0), 1.0);
  return std::exp(std::log(lr_init) * (1 - t) + std::log(lr_final) * t);
}

constexpr uint32_t FLAG_SKIP_NONPOSITIVE_OPACITY = 1u;   // SEGS_RASTER_SKIP_NONPOSITIVE_OPACITY

}  // namespace

GaussianTrainerStep::GaussianTrainerStep(int64_t num_anchors, const ScaffoldDims& dims, int width, int height, torch::Device device,
                                         const ScaffoldOptimization& opt, float scaling_reg_weight, double spatial_lr_scale)
    : dims_(dims), cdims_(dims.c()), opt_(opt), A_(num_anchors), P_(num_anchors * dims.n_offsets), W_(width), H_(height),
      dev_(device), reg_weight_(scaling_reg_weight), spatial_lr_scale_(spatial_lr_scale) {
  auto f = torch::TensorOptions().dtype(torch::kFloat32).device(dev_);
  auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(dev_);
  auto u8 = torch::TensorOptions().dtype(torch::kUInt8).device(dev_);
  // ---- one flat bucket: anchor | offset | anchor_feat | scaling | MLP block (the reference's Adam groups 0-2, 4, 6-11;
  // _opacity and _rotation never receive a gradient and stay outside)
  int64_t pos = 0;
  const std::pair<const char*, int64_t> widths[4] = {{"anchor", 3}, {"offset", 3 * dims.n_offsets}, {"anchor_feat", dims.feat_dim},
                                                     {"scaling", 6}};
  for (auto& w : widths) { seg_[w.first] = {pos, A_ * w.second}; pos += A_ * w.second; }
  mlp_offset_ = pos;
  int64_t offs[SEGS_NEURAL_MAX_TENSORS], cnts[SEGS_NEURAL_MAX_TENSORS], total = 0;
  int nt = 0;
  check(segs_neural_param_layout(&cdims_, offs, cnts, &nt, &total), "segs_neural_param_layout");
  mlp_total_ = total;
  n_params_ = pos + total;
  // Adam groups of the MLP block: 4 tensors each for opacity / cov / color, 2 for the appearance Linear, 4 for the feature bank
  int t = 0;
  auto add_group = [&](int ntens, int kind) {
    mlp_group_.push_back({mlp_offset_ + offs[t], 0});
    for (int k = 0; k < ntens; k++) mlp_group_.back().second += cnts[t + k];
    mlp_group_kind_.push_back(kind);
    t += ntens;
  };
  add_group(4, 0); add_group(4, 1); add_group(4, 2);
  if (dims.appearance_dim > 0) add_group(2, 3);
  if (dims.use_feat_bank) add_group(4, 4);
  TORCH_CHECK(t == nt, "unexpected MLP tensor count");
  params_ = torch::zeros({n_params_}, f);
  grads_ = torch::zeros({n_params_}, f);
  exp_avg_ = torch::zeros({n_params_}, f);
  exp_avg_sq_ = torch::zeros({n_params_}, f);
  step_words_ = torch::zeros({2}, torch::TensorOptions().dtype(torch::kInt64).device(dev_));
  rotation_ = torch::zeros({A_, 4}, f);          // _rotation: identity, never trained (src/gaussian_model.cpp:372)
  rotation_.select(1, 0).fill_(1.0f);
  rot_normalized_ = torch::nn::functional::normalize(rotation_).contiguous();
  // ---- candidate-domain buffers (A * n_offsets rows)
  means3D_ = torch::zeros({P_, 3}, f); colors_ = torch::zeros({P_, 3}, f); opacity_ = torch::zeros({P_, 1}, f);
  scales_ = torch::zeros({P_, 3}, f); rotations_ = torch::zeros({P_, 4}, f); neural_opacity_ = torch::zeros({P_, 1}, f);
  g_means3D_ = torch::zeros({P_, 3}, f); g_colors_ = torch::zeros({P_, 3}, f); g_opacity_ = torch::zeros({P_, 1}, f);
  g_scales_ = torch::zeros({P_, 3}, f); g_rotations_ = torch::zeros({P_, 4}, f); dL_dmean2D_ = torch::zeros({P_, 3}, f);
  neural_temp_ = torch::empty({(int64_t)segs_neural_temp_bytes(&cdims_, (int)A_)}, u8);
  visible_radii_ = torch::zeros({A_}, i32);
  // ---- rasterizer and loss
  out_color_ = torch::zeros({3, H_, W_}, f);
  radii_ = torch::zeros({P_}, i32);
  bg_ = torch::zeros({3}, f);
  geom_ = torch::empty({0}, u8); binning_ = torch::empty({0}, u8); img_ = torch::empty({0}, u8);
  loss_temp_ = torch::empty({(int64_t)segs_l1_ssim_temp_bytes(H_, W_)}, u8);
  loss_out_ = torch::zeros({3}, f);
  dL_dimage_ = torch::empty({3, H_, W_}, f);
  scaling_reg_ = torch::zeros({1}, f);
}

torch::Tensor GaussianTrainerStep::param(const std::string& name) {
  auto it = seg_.find(name);
  TORCH_CHECK(it != seg_.end(), "unknown parameter segment ", name);
  auto flat = params_.slice(0, it->second.first, it->second.first + it->second.second);
  if (name == "anchor") return flat.view({A_, 3});
  if (name == "offset") return flat.view({A_, dims_.n_offsets, 3});
  if (name == "anchor_feat") return flat.view({A_, dims_.feat_dim});
  return flat.view({A_, 6});
}

// updateLearningRate (src/gaussian_model.cpp:874-915); anchor / offset scaled by spatial_lr_scale (:637,640)
void GaussianTrainerStep::learning_rates(int64_t it, std::vector<double>& lr) const {
  const auto& o = opt_;
  lr.clear();
  lr.push_back(expon_lr(it, o.position_lr_init * spatial_lr_scale_, o.position_lr_final * spatial_lr_scale_, o.position_lr_max_steps));
  lr.push_back(expon_lr(it, o.offset_lr_init * spatial_lr_scale_, o.offset_lr_final * spatial_lr_scale_, o.offset_lr_max_steps));
  lr.push_back(o.feature_lr);
  lr.push_back(o.scaling_lr);
  const double by_kind[5] = {expon_lr(it, o.mlp_opacity_lr_init, o.mlp_opacity_lr_final, o.mlp_opacity_lr_max_steps),
                             expon_lr(it, o.mlp_cov_lr_init, o.mlp_cov_lr_final, o.mlp_cov_lr_max_steps),
                             expon_lr(it, o.mlp_color_lr_init, o.mlp_color_lr_final, o.mlp_color_lr_max_steps),
                             expon_lr(it, o.appearance_lr_init, o.appearance_lr_final, o.appearance_lr_max_steps),
                             expon_lr(it, o.mlp_featurebank_lr_init, o.mlp_featurebank_lr_final, o.mlp_featurebank_lr_max_steps)};
  for (int k : mlp_group_kind_) lr.push_back(by_kind[k]);
}

// prefilter_voxel (src/gaussian_renderer.cpp:131-199): radii of the anchors drawn as Gaussians with exp(scaling[:, :3])
void GaussianTrainerStep::prefilter(const KeyframeView& kf) {
  auto scales = torch::exp(param("scaling").slice(1, 0, 3)).contiguous();
  check(segs_visible_filter((int)A_, 0, W_, H_, fp(param("anchor")), fp(scales), 1.0f, fp(rot_normalized_), nullptr, fp(kf.view),
                            fp(kf.proj), kf.tanfovx, kf.tanfovy, 0, visible_radii_.data_ptr<int>(), cur_stream(dev_)),
        "segs_visible_filter");
}

// the asynchronous status read-back of the previous resident forward: an overflow sends the next pass through the
// synchronising, re-sizing path (the pass that overflowed was dropped on the device by the guarded optimizer)
void GaussianTrainerStep::resolve_status() {
  if (!status_pending_) return;
  hip_check(hipEventSynchronize((hipEvent_t)status_event_), "hipEventSynchronize");
  status_pending_ = false;
  const int32_t* h = status_host_.data_ptr<int32_t>();
  num_rendered_ = h[0];
  if (h[3] != 0) capacity_ = 0;
}

void GaussianTrainerStep::render(const KeyframeView& kf) {
  prefilter(kf);
  void* st = cur_stream(dev_);
  check(segs_neural_forward(&cdims_, (int)A_, fp(param("anchor")), fp(param("offset")), fp(param("anchor_feat")), fp(param("scaling")),
                            visible_radii_.data_ptr<int>(), fp(mlp_params()), fp(kf.campos), fp(kf.pose7), fp(means3D_), fp(colors_),
                            fp(opacity_), fp(scales_), fp(rotations_), fp(neural_opacity_), (char*)neural_temp_.data_ptr(), st),
        "segs_neural_forward");
  resolve_status();
  const uint32_t old_flags = segs_raster_set_flags(FLAG_SKIP_NONPOSITIVE_OPACITY);
  if (capacity_ > 0) {
    segs_raster_set_status_mirror((uint32_t*)status_host_.data_ptr<int32_t>());
    const int rc = segs_rasterize_forward_resident((char*)geom_r_.data_ptr(), (char*)binning_r_.data_ptr(), (char*)img_r_.data_ptr(),
                                                   capacity_, (int)P_, (int)P_, 0, 0, fp(bg_), W_, H_, fp(means3D_), nullptr, fp(colors_),
                                                   fp(opacity_), fp(scales_), 1.0f, fp(rotations_), nullptr, fp(kf.view), fp(kf.proj),
                                                   fp(kf.campos), kf.tanfovx, kf.tanfovy, fp(out_color_), radii_.data_ptr<int>(),
                                                   (uint32_t*)status_.data_ptr<int32_t>(), st);
    segs_raster_set_status_mirror(nullptr);
    segs_raster_set_flags(old_flags);
    check(rc, "segs_rasterize_forward_resident");
    hip_check(hipEventRecord((hipEvent_t)status_event_, (hipStream_t)st), "hipEventRecord");
    status_pending_ = true;
    last_resident_ = true;
    return;
  }
  // calibrating pass: the reference-shaped forward blocks on R once; the resident scratch is then sized for 1.25 R
  int R = 0;
  const int rc = segs_rasterize_forward(grow_cb, &geom_, grow_cb, &binning_, grow_cb, &img_, (int)P_, 0, 0, fp(bg_), W_, H_, fp(means3D_),
                                        nullptr, fp(colors_), fp(opacity_), fp(scales_), 1.0f, fp(rotations_), nullptr, fp(kf.view),
                                        fp(kf.proj), fp(kf.campos), kf.tanfovx, kf.tanfovy, 0, fp(out_color_), radii_.data_ptr<int>(), st, &R);
  segs_raster_set_flags(old_flags);
  check(rc, "segs_rasterize_forward");
  num_rendered_ = R;
  last_resident_ = false;
  capacity_ = (int)(R * 1.25) + 65536;
  auto u8 = torch::TensorOptions().dtype(torch::kUInt8).device(dev_);
  geom_r_ = torch::zeros({(int64_t)segs_geometry_bytes((int)P_)}, u8);     // zero-filled: the resident backward keeps it clean
  img_r_ = torch::empty({(int64_t)segs_image_bytes(W_, H_)}, u8);
  binning_r_ = torch::empty({(int64_t)segs_resident_binning_bytes((int)P_, capacity_)}, u8);
  status_ = torch::zeros({4}, torch::TensorOptions().dtype(torch::kInt32).device(dev_));
  status_host_ = torch::zeros({4}, torch::TensorOptions().dtype(torch::kInt32)).pin_memory();
  if (!status_event_) {
    hipEvent_t e;
    hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
    status_event_ = e;
  }
}

torch::Tensor GaussianTrainerStep::trainingOnce(const KeyframeView& kf, const torch::Tensor& gt_image) {
  TORCH_CHECK(gt_image.is_contiguous() && gt_image.sizes() == out_color_.sizes() && gt_image.device() == dev_, "gt_image must be a contiguous (3,H,W) tensor on the step's device");
  iteration_ += 1;
  std::vector<double> lr;
  learning_rates(iteration_, lr);
  void* st = cur_stream(dev_);
  render(kf);
  check(segs_l1_ssim_loss(fp(out_color_), fp(gt_image), H_, W_, (float)opt_.lambda_dssim, fp(loss_out_), fp(dL_dimage_),
                          (char*)loss_temp_.data_ptr(), st),
        "segs_l1_ssim_loss");
  if (last_resident_) {
    check(segs_rasterize_backward_resident((char*)geom_r_.data_ptr(), (char*)binning_r_.data_ptr(), (char*)img_r_.data_ptr(), capacity_,
                                           (int)P_, (int)P_, 0, 0, fp(bg_), W_, H_, fp(means3D_), nullptr, fp(scales_), 1.0f, fp(rotations_),
                                           nullptr, fp(kf.view), fp(kf.proj), fp(kf.campos), kf.tanfovx, kf.tanfovy, radii_.data_ptr<int>(),
                                           fp(dL_dimage_), fp(dL_dmean2D_), nullptr, fp(g_opacity_), fp(g_colors_), fp(g_means3D_), nullptr,
                                           nullptr, fp(g_scales_), fp(g_rotations_), st),
          "segs_rasterize_backward_resident");
  } else {
    check(segs_rasterize_backward((int)P_, 0, 0, num_rendered_, fp(bg_), W_, H_, fp(means3D_), nullptr, fp(colors_), fp(scales_), 1.0f,
                                  fp(rotations_), nullptr, fp(kf.view), fp(kf.proj), fp(kf.campos), kf.tanfovx, kf.tanfovy,
                                  radii_.data_ptr<int>(), (char*)geom_.data_ptr(), (char*)binning_.data_ptr(), (char*)img_.data_ptr(),
                                  fp(dL_dimage_), fp(dL_dmean2D_), nullptr, fp(g_opacity_), fp(g_colors_), fp(g_means3D_), nullptr, nullptr,
                                  fp(g_scales_), fp(g_rotations_), st),
          "segs_rasterize_backward");
  }
  auto seg_ptr = [&](torch::Tensor& bucket, const char* name) { return fp(bucket) + seg_[name].first; };
  check(segs_neural_backward(&cdims_, (int)A_, seg_ptr(params_, "anchor"), seg_ptr(params_, "offset"), seg_ptr(params_, "anchor_feat"),
                             seg_ptr(params_, "scaling"), fp(params_) + mlp_offset_, fp(kf.campos), fp(kf.pose7), fp(g_means3D_),
                             fp(g_colors_), fp(g_opacity_), fp(g_scales_), fp(g_rotations_), seg_ptr(grads_, "anchor"),
                             seg_ptr(grads_, "offset"), seg_ptr(grads_, "anchor_feat"), seg_ptr(grads_, "scaling"), fp(grads_) + mlp_offset_,
                             reg_weight_, reg_weight_ != 0.f ? fp(scaling_reg_) : nullptr, (char*)neural_temp_.data_ptr(), st),
        "segs_neural_backward");
  // optimizer_->step(); optimizer_->zero_grad(true)  (src/gaussian_trainer.cpp:115-116): one fused launch over the bucket,
  // guarded by the rasterizer's overflow word, step count on the device
  std::vector<segs_adam_segment> groups;
  const char* names[4] = {"anchor", "offset", "anchor_feat", "scaling"};
  for (int k = 0; k < 4; k++) groups.push_back({seg_[names[k]].first, seg_[names[k]].second, lr[k]});
  for (size_t k = 0; k < mlp_group_.size(); k++) groups.push_back({mlp_group_[k].first, mlp_group_[k].second, lr[4 + k]});
  const uint32_t* guard = last_resident_ ? (const uint32_t*)status_.data_ptr<int32_t>() + 3 : nullptr;
  check(segs_adam_step_device(fp(params_), fp(grads_), fp(exp_avg_), fp(exp_avg_sq_), groups.data(), (int)groups.size(), opt_.beta1,
                              opt_.beta2, opt_.eps, step_words_.data_ptr<int64_t>(), adam_calls_, 1.0f, 1, guard, st),
        "segs_adam_step_device");
  adam_calls_ += 1;
  return loss_out_.slice(0, 0, 1);
}

int64_t GaussianTrainerStep::steps_taken() { return step_words_[adam_calls_ & 1].item<int64_t>(); }

}  // namespace segs_host
