// gaussian_trainer.cpp -- see gaussian_trainer.h.  C++ twin of segs-slam_amd/neural_gaussians.py (ScaffoldModel,
// ScaffoldTrainerStep): same names, same order of operations.  Reference: src/gaussian_trainer.cpp:47-117 (loop body of trainingOnce),
// src/gaussian_mapper.cpp:861-1030 (the mapper's variant with the scaling regulariser), src/gaussian_model.cpp:620-690,
// 874-915 (Adam groups, learning-rate schedule), src/gaussian_renderer.cpp:131-199 (prefilter_voxel).
#include "gaussian_trainer.h"

#include "anchor_densifier.h"
#include "keyframe_exchange.h"

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

// ---------------------------------------------------------------------------------------------------------------------
ScaffoldModelState::ScaffoldModelState(int64_t num_anchors, const ScaffoldDims& d, torch::Device device, int64_t cap)
    : dims(d), cdims(d.c()), dev(device), A(num_anchors) {
  int64_t offs[SEGS_NEURAL_MAX_TENSORS], cnts[SEGS_NEURAL_MAX_TENSORS], total = 0;
  int nt = 0;
  check(segs_neural_param_layout(&cdims, offs, cnts, &nt, &total), "segs_neural_param_layout");
  mlp_total = total;
  // Adam groups of the MLP block: 4 tensors each for opacity / cov / color, 2 for the appearance Linear, 4 for the feature bank
  int t = 0;
  auto add_group = [&](int ntens, int kind) {
    mlp_group_rel.push_back({offs[t], 0});
    for (int k = 0; k < ntens; k++) mlp_group_rel.back().second += cnts[t + k];
    mlp_group_kind.push_back(kind);
    t += ntens;
  };
  add_group(4, 0); add_group(4, 1); add_group(4, 2);
  if (dims.appearance_dim > 0) add_group(2, 3);
  if (dims.use_feat_bank) add_group(4, 4);
  TORCH_CHECK(t == nt, "unexpected MLP tensor count");
  allocate(std::max<int64_t>(std::max(cap, A), 1));
}

int64_t ScaffoldModelState::width(const std::string& name) const {
  if (name == "anchor") return 3;
  if (name == "offset") return 3 * dims.n_offsets;
  if (name == "anchor_feat") return dims.feat_dim;
  TORCH_CHECK(name == "scaling", "unknown parameter segment ", name);
  return 6;
}

// one flat bucket: anchor | offset | anchor_feat | scaling (each laid out for `capacity` rows) | MLP block -- the reference's
// Adam groups 0-2, 4, 6-11; _opacity and _rotation never receive a gradient and stay outside
void ScaffoldModelState::allocate(int64_t cap) {
  capacity = cap;
  auto f = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
  int64_t pos = 0;
  for (const char* name : {"anchor", "offset", "anchor_feat", "scaling"}) { seg_offset[name] = pos; pos += capacity * width(name); }
  mlp_offset = pos;
  n_params = pos + mlp_total;
  params = torch::zeros({n_params}, f);
  grads = torch::zeros({n_params}, f);
  exp_avg = torch::zeros({n_params}, f);
  exp_avg_sq = torch::zeros({n_params}, f);
  rotation = torch::zeros({capacity, 4}, f);   // _rotation: identity, never trained (src/gaussian_model.cpp:372)
  rotation.select(1, 0).fill_(1.0f);
  opacity = torch::zeros({capacity, 1}, f);    // _opacity: unused by the forward
}

void ScaffoldModelState::reserve(int64_t cap) {
  if (cap <= capacity) return;
  torch::Tensor old[4] = {params, grads, exp_avg, exp_avg_sq};
  auto old_off = seg_offset;
  const int64_t old_mlp = mlp_offset, old_n = n_params;
  auto old_rot = rotation, old_op = opacity;
  allocate(cap);
  torch::Tensor* now[4] = {&params, &grads, &exp_avg, &exp_avg_sq};
  for (int b = 0; b < 4; b++) {
    for (const char* name : {"anchor", "offset", "anchor_feat", "scaling"}) {
      const int64_t n = A * width(name);
      now[b]->slice(0, seg_offset[name], seg_offset[name] + n).copy_(old[b].slice(0, old_off[name], old_off[name] + n));
    }
    now[b]->slice(0, mlp_offset, n_params).copy_(old[b].slice(0, old_mlp, old_n));
  }
  rotation.slice(0, 0, A).copy_(old_rot.slice(0, 0, A));
  opacity.slice(0, 0, A).copy_(old_op.slice(0, 0, A));
}

torch::Tensor ScaffoldModelState::view(const torch::Tensor& bucket, const std::string& name, int64_t rows) const {
  if (rows < 0) rows = A;
  const int64_t o = seg_offset.at(name);
  auto flat = bucket.slice(0, o, o + rows * width(name));
  if (name == "anchor") return flat.view({rows, 3});
  if (name == "offset") return flat.view({rows, dims.n_offsets, 3});
  if (name == "anchor_feat") return flat.view({rows, dims.feat_dim});
  return flat.view({rows, 6});
}

std::vector<std::pair<int64_t, int64_t>> ScaffoldModelState::anchor_groups() const {
  std::vector<std::pair<int64_t, int64_t>> out;
  for (const char* name : {"anchor", "offset", "anchor_feat", "scaling"}) out.push_back({seg_offset.at(name), A * width(name)});
  return out;
}

// ---------------------------------------------------------------------------------------------------------------------
GaussianTrainerStep::GaussianTrainerStep(int64_t num_anchors, const ScaffoldDims& dims, int width, int height, torch::Device device,
                                         const ScaffoldOptimization& opt, float scaling_reg_weight, double spatial_lr_scale, int64_t capacity)
    : model_(num_anchors, dims, device, capacity), opt_(opt), W_(width), H_(height), dev_(device), reg_weight_(scaling_reg_weight),
      spatial_lr_scale_(spatial_lr_scale) {
  auto f = torch::TensorOptions().dtype(torch::kFloat32).device(dev_);
  auto u8 = torch::TensorOptions().dtype(torch::kUInt8).device(dev_);
  auto i64 = torch::TensorOptions().dtype(torch::kInt64).device(dev_);
  mlp_count_.words = torch::zeros({2}, i64);
  anchor_count_.words = torch::zeros({2}, i64);
  allocate_candidate_buffers();
  out_color_ = torch::zeros({3, H_, W_}, f);
  bg_ = torch::zeros({3}, f);
  geom_ = torch::empty({0}, u8); binning_ = torch::empty({0}, u8); img_ = torch::empty({0}, u8);
  loss_temp_ = torch::empty({(int64_t)segs_l1_ssim_temp_bytes(H_, W_)}, u8);
  loss_out_ = torch::zeros({3}, f);
  dL_dimage_ = torch::empty({3, H_, W_}, f);
  scaling_reg_ = torch::zeros({1}, f);
}

GaussianTrainerStep::~GaussianTrainerStep() {
  if (status_event_) (void)hipEventDestroy((hipEvent_t)status_event_);
  if (freq_.plan) segs_freq_plan_destroy(freq_.plan);
}

void GaussianTrainerStep::enable_frequency_regularization(float lambda_high, const std::vector<float>& scales, int64_t start,
                                                          int64_t until, bool multi_resolution) {
  if (freq_.plan) { segs_freq_plan_destroy(freq_.plan); freq_.plan = nullptr; }
  freq_.targets.clear();
  freq_.on = lambda_high != 0.f;
  freq_.lambda_high = lambda_high;
  freq_.multi = multi_resolution;
  freq_.scales = multi_resolution ? scales : std::vector<float>{1.0f};   // high_frequency_loss = the s = 1 term with weight 1
  freq_.start = start;
  freq_.until = until;
  if (!freq_.on) return;
  check(segs_freq_plan_create(H_, W_, (int)freq_.scales.size(), freq_.scales.data(), lambda_high, &freq_.plan), "segs_freq_plan_create");
  freq_value_ = torch::zeros({1}, torch::TensorOptions().dtype(torch::kFloat32).device(dev_));
}

// candidate-domain buffers (capacity * n_offsets rows) and the rasterizer's per-Gaussian outputs; re-made when the map
// outgrew them (the resident rasterizer scratch is then re-calibrated by the next forward)
void GaussianTrainerStep::allocate_candidate_buffers() {
  auto f = torch::TensorOptions().dtype(torch::kFloat32).device(dev_);
  auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(dev_);
  auto u8 = torch::TensorOptions().dtype(torch::kUInt8).device(dev_);
  cand_capacity_ = model_.capacity;
  const int64_t Pc = cand_capacity_ * model_.dims.n_offsets;
  means3D_ = torch::zeros({Pc, 3}, f); colors_ = torch::zeros({Pc, 3}, f); opacity_ = torch::zeros({Pc, 1}, f);
  scales_ = torch::zeros({Pc, 3}, f); rotations_ = torch::zeros({Pc, 4}, f); neural_opacity_ = torch::zeros({Pc, 1}, f);
  g_means3D_ = torch::zeros({Pc, 3}, f); g_colors_ = torch::zeros({Pc, 3}, f); g_opacity_ = torch::zeros({Pc, 1}, f);
  g_scales_ = torch::zeros({Pc, 3}, f); g_rotations_ = torch::zeros({Pc, 4}, f); dL_dmean2D_ = torch::zeros({Pc, 3}, f);
  neural_temp_ = torch::empty({(int64_t)segs_neural_temp_bytes(&model_.cdims, (int)cand_capacity_)}, u8);
  visible_radii_ = torch::zeros({cand_capacity_}, i32);
  radii_ = torch::zeros({Pc}, i32);
  capacity_ = 0;   // resident scratch: calibrate anew
  status_pending_ = false;
}

void GaussianTrainerStep::enable_densification(AnchorDensifier* densifier, uint64_t seed) {
  densifier_ = densifier;
  densify_generator_ = at::detail::createCPUGenerator(seed);
}

void GaussianTrainerStep::set_process_group(c10::intrusive_ptr<c10d::Backend> pg, bool sharded_optimizer, bool single_rank_collectives) {
  pg_ = std::move(pg);
  sharded_optimizer_ = sharded_optimizer;
  single_rank_collectives_ = single_rank_collectives;
  ex_.reset();
}
int GaussianTrainerStep::world() const { return pg_ ? pg_->getSize() : 1; }
int GaussianTrainerStep::rank() const { return pg_ ? pg_->getRank() : 0; }

// the step's exchange over the model's flat bucket (rebuilt when densification re-sized the bucket)
KeyframeExchange& GaussianTrainerStep::exchange() {
  if (!ex_ || ex_->size() != model_.n_params)
    ex_ = std::make_unique<KeyframeExchange>(model_.n_params, dev_, pg_, sharded_optimizer_, single_rank_collectives_);
  return *ex_;
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
  for (int k : model_.mlp_group_kind) lr.push_back(by_kind[k]);
}

// prefilter_voxel (src/gaussian_renderer.cpp:131-199): radii of the anchors drawn as Gaussians with exp(scaling[:, :3])
const torch::Tensor& GaussianTrainerStep::anchor_rotations() {
  // _rotation is never trained: normalise it again only when densification rewrote rows
  if (rot_rows_ != model_.A || !rot_normalized_.defined()) {
    rot_normalized_ = torch::nn::functional::normalize(model_.rotation.slice(0, 0, model_.A)).contiguous();
    rot_rows_ = model_.A;
  }
  return rot_normalized_;
}

void GaussianTrainerStep::prefilter(const KeyframeView& kf) {
  // (exp(_scaling[:, :3]) is formed inside the kernel from the rows of 6 log-scales: no intermediate tensor)
  check(segs_visible_filter_log_scales((int)model_.A, W_, H_, fp(model_.param("anchor")), model_.seg_ptr(model_.params, "scaling"), 6,
                                       fp(anchor_rotations()), fp(kf.view), fp(kf.proj), kf.tanfovx, kf.tanfovy,
                                       visible_radii_.data_ptr<int>(), cur_stream(dev_)),
        "segs_visible_filter_log_scales");
}

// the asynchronous status read-back of the previous resident forward: an overflow sends the next pass through the
// synchronising, re-sizing path (the pass that overflowed was dropped on the device by the guarded optimizer)
// Returns true when the pass it resolves had overflowed (that pass was dropped on the device).
bool GaussianTrainerStep::resolve_status() {
  if (!status_pending_) return false;
  hip_check(hipEventSynchronize((hipEvent_t)status_event_), "hipEventSynchronize");
  status_pending_ = false;
  const int32_t* h = status_host_.data_ptr<int32_t>();
  num_rendered_ = h[0];
  if (h[3] != 0) { capacity_ = 0; return true; }
  return false;
}

void GaussianTrainerStep::render(const KeyframeView& kf) {
  if (model_.capacity > cand_capacity_) allocate_candidate_buffers();   // the map outgrew the buffers
  void* st = cur_stream(dev_);
  const int64_t A = model_.A, P = A * model_.dims.n_offsets, rows = cand_capacity_ * model_.dims.n_offsets;
  if (fuse_projection_) {
    resolve_status();
    if (capacity_ > 0) {
      // prefilter, neural Gaussians and the per-Gaussian projection in the neural kernels; the rasterizer bins and renders
      const uint32_t old = segs_raster_set_flags(FLAG_SKIP_NONPOSITIVE_OPACITY);
      segs_projection_targets tg;
      int rc = segs_resident_projection_targets((char*)geom_r_.data_ptr(), (char*)binning_r_.data_ptr(), (char*)img_r_.data_ptr(), capacity_,
                                                (int)rows, (int)P, W_, H_, radii_.data_ptr<int>(), (uint32_t*)status_.data_ptr<int32_t>(), &tg);
      if (rc == 0)
        rc = segs_neural_forward_projected(&model_.cdims, (int)A, fp(model_.param("anchor")), fp(model_.param("offset")),
                                           fp(model_.param("anchor_feat")), fp(model_.param("scaling")), visible_radii_.data_ptr<int>(),
                                           fp(anchor_rotations()), fp(model_.mlp_params()), fp(kf.campos), fp(kf.pose7), fp(means3D_),
                                           fp(scales_), fp(rotations_), fp(neural_opacity_), &tg, fp(kf.view), fp(kf.proj), W_, H_,
                                           kf.tanfovx, kf.tanfovy, 1.0f, (char*)neural_temp_.data_ptr(), st);
      if (rc == 0) {
        segs_raster_set_status_mirror((uint32_t*)status_host_.data_ptr<int32_t>());
        rc = segs_rasterize_forward_resident_projected((char*)geom_r_.data_ptr(), (char*)binning_r_.data_ptr(), (char*)img_r_.data_ptr(),
                                                       capacity_, (int)rows, (int)P, fp(bg_), W_, H_, fp(out_color_),
                                                       (uint32_t*)status_.data_ptr<int32_t>(), st);
        segs_raster_set_status_mirror(nullptr);
      }
      segs_raster_set_flags(old);
      check(rc, "segs_neural_forward_projected / segs_rasterize_forward_resident_projected");
      hip_check(hipEventRecord((hipEvent_t)status_event_, (hipStream_t)st), "hipEventRecord");
      status_pending_ = true;
      last_resident_ = true;
      return;
    }
  }
  prefilter(kf);
  check(segs_neural_forward(&model_.cdims, (int)A, fp(model_.param("anchor")), fp(model_.param("offset")), fp(model_.param("anchor_feat")),
                            fp(model_.param("scaling")), visible_radii_.data_ptr<int>(), fp(model_.mlp_params()), fp(kf.campos), fp(kf.pose7),
                            fp(means3D_), fp(colors_), fp(opacity_), fp(scales_), fp(rotations_), fp(neural_opacity_),
                            (char*)neural_temp_.data_ptr(), st),
        "segs_neural_forward");
  resolve_status();
  const uint32_t old_flags = segs_raster_set_flags(FLAG_SKIP_NONPOSITIVE_OPACITY);
  if (capacity_ > 0) {
    segs_raster_set_status_mirror((uint32_t*)status_host_.data_ptr<int32_t>());
    const int rc = segs_rasterize_forward_resident((char*)geom_r_.data_ptr(), (char*)binning_r_.data_ptr(), (char*)img_r_.data_ptr(),
                                                   capacity_, (int)rows, (int)P, 0, 0, fp(bg_), W_, H_, fp(means3D_), nullptr, fp(colors_),
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
  const int rc = segs_rasterize_forward(grow_cb, &geom_, grow_cb, &binning_, grow_cb, &img_, (int)P, 0, 0, fp(bg_), W_, H_, fp(means3D_),
                                        nullptr, fp(colors_), fp(opacity_), fp(scales_), 1.0f, fp(rotations_), nullptr, fp(kf.view),
                                        fp(kf.proj), fp(kf.campos), kf.tanfovx, kf.tanfovy, 0, fp(out_color_), radii_.data_ptr<int>(), st, &R);
  segs_raster_set_flags(old_flags);
  check(rc, "segs_rasterize_forward");
  num_rendered_ = R;
  last_resident_ = false;
  capacity_ = (int)(R * 1.25) + 65536;
  auto u8 = torch::TensorOptions().dtype(torch::kUInt8).device(dev_);
  geom_r_ = torch::zeros({(int64_t)segs_geometry_bytes((int)rows)}, u8);     // zero-filled: the resident backward keeps it clean
  img_r_ = torch::empty({(int64_t)segs_image_bytes(W_, H_)}, u8);
  binning_r_ = torch::empty({(int64_t)segs_resident_binning_bytes((int)rows, capacity_)}, u8);
  status_ = torch::zeros({4}, torch::TensorOptions().dtype(torch::kInt32).device(dev_));
  status_host_ = torch::zeros({4}, torch::TensorOptions().dtype(torch::kInt32)).pin_memory();
  if (!status_event_) {
    hipEvent_t e;
    hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
    status_event_ = e;
  }
}

// render -> [overflow word's all-reduce starts] -> L1/SSIM -> raster backward -> neural backward (gradients into the bucket)
void GaussianTrainerStep::forward_backward(const KeyframeView& kf, const torch::Tensor& gt_image) {
  void* st = cur_stream(dev_);
  const int64_t A = model_.A, P = A * model_.dims.n_offsets, rows = cand_capacity_ * model_.dims.n_offsets;
  if (A == 0) {
    // every anchor was pruned: the reference's rasterizer short-circuits P == 0 to a zero image (src/rasterize_points.cu:81)
    out_color_.zero_();
    last_resident_ = false;
    exchange().reduce_flag_async(torch::Tensor());
    check(segs_l1_ssim_loss(fp(out_color_), fp(gt_image), H_, W_, (float)opt_.lambda_dssim, fp(loss_out_), fp(dL_dimage_),
                            (char*)loss_temp_.data_ptr(), st), "segs_l1_ssim_loss");
    return;
  }
  render(kf);
  // the overflow word is final once the forward's binning has run: its all-reduce hides behind loss and backward
  exchange().reduce_flag_async(last_resident_ ? status_.slice(0, 3, 4) : torch::Tensor());
  check(segs_l1_ssim_loss(fp(out_color_), fp(gt_image), H_, W_, (float)opt_.lambda_dssim, fp(loss_out_), fp(dL_dimage_),
                          (char*)loss_temp_.data_ptr(), st),
        "segs_l1_ssim_loss");
  if (freq_.on && iteration_ > freq_.start && iteration_ < freq_.until) {   // src/gaussian_mapper.cpp:938
    const uint32_t version = gt_image._version();
    auto hit = freq_.targets.begin();
    while (hit != freq_.targets.end() && !(hit->gt.data_ptr() == gt_image.data_ptr() && hit->version == version)) ++hit;
    if (hit == freq_.targets.end()) {
      while (freq_.targets.size() >= freq_.max_targets) freq_.targets.pop_back();
      auto tab = torch::empty({(int64_t)segs_freq_target_floats(freq_.plan)}, torch::TensorOptions().dtype(torch::kFloat32).device(dev_));
      check(segs_freq_target(freq_.plan, fp(gt_image), fp(tab), st), "segs_freq_target");
      freq_.targets.push_front({gt_image, tab, version});
    } else if (hit != freq_.targets.begin()) {
      freq_.targets.splice(freq_.targets.begin(), freq_.targets, hit);
    }
    check(segs_freq_loss(freq_.plan, fp(out_color_), fp(freq_.targets.front().table), fp(dL_dimage_), fp(freq_value_), fp(loss_out_), st), "segs_freq_loss");
  }
  if (dL_mask_.defined()) dL_dimage_.mul_(dL_mask_);
  if (last_resident_) {
    check(segs_rasterize_backward_resident((char*)geom_r_.data_ptr(), (char*)binning_r_.data_ptr(), (char*)img_r_.data_ptr(), capacity_,
                                           (int)rows, (int)P, 0, 0, fp(bg_), W_, H_, fp(means3D_), nullptr, fp(scales_), 1.0f, fp(rotations_),
                                           nullptr, fp(kf.view), fp(kf.proj), fp(kf.campos), kf.tanfovx, kf.tanfovy, radii_.data_ptr<int>(),
                                           fp(dL_dimage_), fp(dL_dmean2D_), nullptr, fp(g_opacity_), fp(g_colors_), fp(g_means3D_), nullptr,
                                           nullptr, fp(g_scales_), fp(g_rotations_), st),
          "segs_rasterize_backward_resident");
  } else {
    check(segs_rasterize_backward((int)P, 0, 0, num_rendered_, fp(bg_), W_, H_, fp(means3D_), nullptr, fp(colors_), fp(scales_), 1.0f,
                                  fp(rotations_), nullptr, fp(kf.view), fp(kf.proj), fp(kf.campos), kf.tanfovx, kf.tanfovy,
                                  radii_.data_ptr<int>(), (char*)geom_.data_ptr(), (char*)binning_.data_ptr(), (char*)img_.data_ptr(),
                                  fp(dL_dimage_), fp(dL_dmean2D_), nullptr, fp(g_opacity_), fp(g_colors_), fp(g_means3D_), nullptr, nullptr,
                                  fp(g_scales_), fp(g_rotations_), st),
          "segs_rasterize_backward");
  }
  auto& m = model_;
  check(segs_neural_backward(&m.cdims, (int)A, m.seg_ptr(m.params, "anchor"), m.seg_ptr(m.params, "offset"), m.seg_ptr(m.params, "anchor_feat"),
                             m.seg_ptr(m.params, "scaling"), fp(m.params) + m.mlp_offset, fp(kf.campos), fp(kf.pose7), fp(g_means3D_),
                             fp(g_colors_), fp(g_opacity_), fp(g_scales_), fp(g_rotations_), m.seg_ptr(m.grads, "anchor"),
                             m.seg_ptr(m.grads, "offset"), m.seg_ptr(m.grads, "anchor_feat"), m.seg_ptr(m.grads, "scaling"),
                             fp(m.grads) + m.mlp_offset, reg_weight_, reg_weight_ != 0.f ? fp(scaling_reg_) : nullptr,
                             (char*)neural_temp_.data_ptr(), st),
        "segs_neural_backward");
}

// optimizer_->step(); optimizer_->zero_grad(true)  (src/gaussian_trainer.cpp:115-116): one fused launch over the groups
// (restricted to this rank's shard of the bucket when the optimizer is sharded), guarded by the summed overflow word, step
// count on the device
void GaussianTrainerStep::adam(const std::vector<segs_adam_segment>& groups_in, StepCount& count, const uint32_t* guard) {
  auto groups = exchange().clip_segments(groups_in);
  if (groups.empty()) groups.push_back({0, 0, 0.0});   // the launch always happens (an empty shard still advances the count)
  auto& m = model_;
  check(segs_adam_step_device(fp(m.params), fp(m.grads), fp(m.exp_avg), fp(m.exp_avg_sq), groups.data(), (int)groups.size(), opt_.beta1,
                              opt_.beta2, opt_.eps, count.words.data_ptr<int64_t>(), count.calls, 1.0f / (float)world(), 1, guard,
                              cur_stream(dev_)),
        "segs_adam_step_device");
  count.calls += 1;
}

// One mapper iteration (neural_gaussians.py::ScaffoldTrainerStep.training_once; src/gaussian_mapper.cpp:823-1032).  A pass
// whose instance count outgrew the rasterizer's resident capacity on ANY rank is dropped on the device by every rank --
// statistics and optimizer are guarded by the all-reduced overflow word, the Adam step counts live on the device and do not
// advance -- and the rank that overflowed re-sizes its scratch at its next forward.
//
// With one rank a dropped iteration is not lost (set_redo_dropped_steps, on by default): the host learns of it when it resolves
// that forward's status word -- before anything of the next iteration is queued -- and runs the same keyframe with the same
// iteration number again right there (its forward re-calibrates, so it cannot overflow).  Parameters, moments and step counts
// are what the dropped pass found, so the optimizer takes every step the reference takes, in the same order.  The keyframe's
// and the target's tensors must therefore stay unchanged until the next call.
torch::Tensor GaussianTrainerStep::trainingOnce(const KeyframeView& kf, const torch::Tensor& gt_image) {
  TORCH_CHECK(gt_image.is_contiguous() && gt_image.sizes() == out_color_.sizes() && gt_image.device() == dev_, "gt_image must be a contiguous (3,H,W) tensor on the step's device");
  redo_if_dropped();
  iteration_ += 1;
  last_kf_ = kf;
  last_gt_ = gt_image;
  have_last_ = true;
  return iteration_body(kf, gt_image);
}

// An iteration the device dropped is run again -- same keyframe, same iteration number (iteration_ still holds it) -- before the
// next one is queued.  One rank: this rank's own status word.  N > 1: the summed word every rank mirrored to its host after the
// gradient exchange (KeyframeExchange::mirror_flag), so all ranks redo the same iteration together; the rank that overflowed
// re-calibrates in its forward (resolve_status), and a redo that another rank's overflow drops again is redone again.
void GaussianTrainerStep::redo_if_dropped() {
  if (!redo_dropped_steps_ || !have_last_) return;
  have_last_ = false;
  for (int tries = 0;; tries++) {
    bool dropped;
    if (world() == 1) dropped = resolve_status();
    else { dropped = exchange().step_dropped(); resolve_status(); }
    if (!dropped) return;
    TORCH_CHECK(tries < 4, "an iteration kept being dropped by the device");
    redone_steps_ += 1;
    iteration_body(last_kf_, last_gt_);
    if (world() == 1) return;   // (a re-calibrating forward cannot overflow)
  }
}

void GaussianTrainerStep::finish() { redo_if_dropped(); }

torch::Tensor GaussianTrainerStep::iteration_body(const KeyframeView& kf, const torch::Tensor& gt_image) {
  std::vector<double> lr;
  learning_rates(iteration_, lr);
  AnchorDensifier* d = densifier_;
  const bool in_stat_window = d && model_.A > 0 && d->params().start_stat < iteration_ && iteration_ < d->params().update_until;   // gaussian_mapper.cpp:961-968
  const bool adjust_now = in_stat_window && iteration_ > d->params().update_from && iteration_ % d->params().update_interval == 0;
  forward_backward(kf, gt_image);
  torch::Tensor flag = exchange().wait_flag();
  if (adjust_now) {
    // adjust_anchor reads tensor sizes on the host and must see a valid pass on every rank: resolve the summed overflow word
    // here (the one synchronisation of a densification iteration) and redo the pass while it is set
    int tries = 0;
    while (flag.item<int32_t>() != 0) {
      TORCH_CHECK(++tries <= 3, "resident rasterizer kept overflowing its re-sized scratch");
      resolve_status();
      model_.grads.zero_();
      forward_backward(kf, gt_image);
      flag = exchange().wait_flag();
    }
  }
  const uint32_t* guard = (const uint32_t*)flag.data_ptr<int32_t>();
  // a densification may re-size the bucket, so the shard partition the optimizer clips to below is not the one a
  // reduce-scatter would have summed for: every element gets the full sum on those steps
  exchange().reduce_gradients(model_.grads, adjust_now);
  if (exchange().active() && redo_dropped_steps_ && !adjust_now) exchange().mirror_flag();   // (an adjust_anchor iteration has resolved its word above)
  if (on_gradients_) on_gradients_(model_.grads);
  bool adjusted = false;
  if (in_stat_window) {
    d->training_statis(neural_opacity_, visible_radii_, radii_, dL_dmean2D_, guard, world() > 1, cur_stream(dev_));
    if (adjust_now) {
      if (exchange().sharded()) {   // moments are only current inside each rank's shard: make them whole before rows move
        exchange().gather(model_.exp_avg);
        exchange().gather(model_.exp_avg_sq);
      }
      d->reduce_statistics(&exchange());
      d->adjust_anchor(densify_generator_, world());
      adjusted = true;
    }
  }
  std::vector<segs_adam_segment> anchor_groups, mlp_groups;
  {
    auto ag = model_.anchor_groups();
    for (size_t k = 0; k < ag.size(); k++) anchor_groups.push_back({ag[k].first, ag[k].second, lr[k]});
    for (size_t k = 0; k < model_.mlp_group_rel.size(); k++)
      mlp_groups.push_back({model_.mlp_offset + model_.mlp_group_rel[k].first, model_.mlp_group_rel[k].second, lr[4 + k]});
  }
  if (adjusted) {
    // the six anchor tensors were re-created by adjust_anchor: no gradient, skipped by Adam this iteration
    // (src/gaussian_model.cpp:1677), so from here on their step count lags the MLPs'
    for (const char* name : {"anchor", "offset", "anchor_feat", "scaling"}) model_.grad(name).zero_();
    if (!anchor_count_split_) {
      anchor_count_.words.copy_(mlp_count_.words);
      anchor_count_.calls = mlp_count_.calls;
      anchor_count_split_ = true;
    }
    adam(mlp_groups, mlp_count_, guard);
  } else if (!anchor_count_split_) {
    auto all = anchor_groups;
    all.insert(all.end(), mlp_groups.begin(), mlp_groups.end());
    adam(all, mlp_count_, guard);
  } else {
    adam(anchor_groups, anchor_count_, guard);
    adam(mlp_groups, mlp_count_, guard);
  }
  if (exchange().sharded()) {
    exchange().gather(model_.params);
    model_.grads.zero_();   // outside this rank's shard the bucket still holds its own contribution
  }
  return loss_out_.slice(0, 0, 1);
}

int64_t GaussianTrainerStep::steps_taken() { return mlp_count_.words[mlp_count_.calls & 1].item<int64_t>(); }
int64_t GaussianTrainerStep::anchor_steps_taken() {
  return anchor_count_split_ ? anchor_count_.words[anchor_count_.calls & 1].item<int64_t>() : steps_taken();
}

}  // namespace segs_host
