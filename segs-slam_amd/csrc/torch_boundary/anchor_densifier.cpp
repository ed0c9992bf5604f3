// anchor_densifier.cpp -- see anchor_densifier.h.  Reference: src/gaussian_model.cpp:1459-1762.
#include "anchor_densifier.h"

#include <c10/hip/HIPStream.h>

#include <cmath>

#include "../../../include/segs_densify.h"
#include "../../../include/segs_raster.h"
#include "keyframe_exchange.h"

namespace segs_host {
namespace {
const char* const kStatNames[4] = {"opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"};
void check(int status, const char* what) {
  if (status != SEGS_OK) AT_ERROR(what, " failed (", status, "): ", segs_last_error());
}
float* fp(const torch::Tensor& t) { return t.data_ptr<float>(); }
void* cur_stream(const torch::Device& d) { return (void*)c10::hip::getCurrentHIPStream(d.index()).stream(); }
}  // namespace

AnchorDensifier::AnchorDensifier(ScaffoldModelState& model, const DensifyParams& params) : m_(model), p_(params) {
  alloc_stats(m_.capacity);
}

// The four accumulators are segments of ONE flat tensor, and so is their keyframe-parallel shadow: with N ranks every rank
// accumulates its own keyframe's increments there and reduce_statistics() folds the sum over ranks into the replicated
// accumulators with one all-reduce right before adjust_anchor.
void AnchorDensifier::alloc_stats(int64_t capacity) {
  auto f = torch::TensorOptions().dtype(torch::kFloat32).device(m_.dev);
  const int64_t no = m_.dims.n_offsets;
  auto old = stats_, old_delta = delta_;
  stats_capacity_ = capacity;
  const int64_t sizes[4] = {capacity, capacity, capacity * no, capacity * no};
  stats_flat_ = torch::zeros({sizes[0] + sizes[1] + sizes[2] + sizes[3]}, f);
  delta_flat_ = torch::zeros_like(stats_flat_);
  int64_t off = 0;
  for (int k = 0; k < 4; k++) {
    stats_[kStatNames[k]] = stats_flat_.slice(0, off, off + sizes[k]);
    delta_[kStatNames[k]] = delta_flat_.slice(0, off, off + sizes[k]);
    off += sizes[k];
  }
  for (auto& kv : old) stats_[kv.first].slice(0, 0, kv.second.numel()).copy_(kv.second);
  for (auto& kv : old_delta) delta_[kv.first].slice(0, 0, kv.second.numel()).copy_(kv.second);
}

torch::Tensor AnchorDensifier::stat(const std::string& name) {
  const int64_t no = m_.dims.n_offsets;
  const int64_t n = (name == "opacity_accum" || name == "anchor_demon") ? m_.A : m_.A * no;
  return stats_.at(name).slice(0, 0, n).view({-1, 1});
}

void AnchorDensifier::training_statis(const torch::Tensor& neural_opacity, const torch::Tensor& visible_radii, const torch::Tensor& radii,
                                      const torch::Tensor& dL_dmean2D, const uint32_t* skip_flag, bool into_delta, void* stream) {
  auto& s = into_delta ? delta_ : stats_;
  check(segs_training_statis_guarded((int)m_.A, m_.dims.n_offsets, fp(neural_opacity), visible_radii.data_ptr<int>(), radii.data_ptr<int>(),
                                     fp(dL_dmean2D), fp(s["opacity_accum"]), fp(s["anchor_demon"]), fp(s["offset_gradient_accum"]),
                                     fp(s["offset_denom"]), skip_flag, stream),
        "segs_training_statis_guarded");
}

void AnchorDensifier::reduce_statistics(KeyframeExchange* exchange) {
  if (exchange && exchange->world() > 1) exchange->all_reduce_sum(delta_flat_);
  stats_flat_ += delta_flat_;
  delta_flat_.zero_();
}

// :1623-1696: concatenate the new rows to the six tensors, zero-extend the Adam moments and the counters
void AnchorDensifier::append(const torch::Tensor& new_anchor, const torch::Tensor& new_feat, float cur_size) {
  const int64_t n_new = new_anchor.size(0);
  const int64_t A0 = m_.A, A1 = m_.A + n_new;
  if (A1 > m_.capacity) m_.reserve((int64_t)(A1 * 1.5) + 1024);
  if (A1 > stats_capacity_) alloc_stats(m_.capacity);
  m_.A = A1;
  const char* names[4] = {"anchor", "offset", "anchor_feat", "scaling"};
  for (const torch::Tensor* bucket : {&m_.grads, &m_.exp_avg, &m_.exp_avg_sq})
    for (const char* name : names) m_.view(*bucket, name).slice(0, A0, A1).zero_();
  m_.param("anchor").slice(0, A0, A1).copy_(new_anchor);
  m_.param("offset").slice(0, A0, A1).zero_();
  m_.param("anchor_feat").slice(0, A0, A1).copy_(new_feat);
  auto f = torch::TensorOptions().dtype(torch::kFloat32).device(m_.dev);
  m_.param("scaling").slice(0, A0, A1).copy_(torch::log(torch::ones({n_new, 6}, f) * cur_size));
  m_.rotation.slice(0, A0, A1).zero_();
  m_.rotation.slice(0, A0, A1).select(1, 0).fill_(1.0f);
  auto x = 0.1f * torch::ones({n_new, 1}, f);
  m_.opacity.slice(0, A0, A1).copy_(torch::log(x / (1 - x)));   // general_utils::inverse_sigmoid(0.1)
  stats_["anchor_demon"].slice(0, A0, A1).zero_();
  stats_["opacity_accum"].slice(0, A0, A1).zero_();
}

// :1559-1699.  grads (A_init*no), offset_mask (A_init*no) bool, rands[i] (A_init*no) in [0,1)
void AnchorDensifier::anchor_growing(const torch::Tensor& grads_in, double threshold, const torch::Tensor& offset_mask,
                                     const std::vector<torch::Tensor>& rands) {
  const int no = m_.dims.n_offsets;
  const int64_t A_init = m_.A;
  auto mask_u8 = offset_mask.to(torch::kUInt8).contiguous();
  auto grads = grads_in.contiguous();
  auto f = torch::TensorOptions().dtype(torch::kFloat32).device(m_.dev);
  auto n_new_dev = torch::zeros({1}, torch::TensorOptions().dtype(torch::kInt32).device(m_.dev));
  void* st = cur_stream(m_.dev);
  for (int i = 0; i < p_.update_depth; i++) {
    const float cur_threshold = (float)(threshold * std::pow(std::floor(p_.update_hierachy_factor / 2), i));
    const double size_factor = std::floor(p_.update_init_factor / std::pow((double)p_.update_hierachy_factor, i));
    const float cur_size = (float)(p_.voxel_size * size_factor);
    if (m_.A == A_init && i > 0) continue;   // :1573-1577
    const int64_t max_new = A_init * no;
    auto temp = torch::empty({(int64_t)segs_anchor_growing_temp_bytes((int)m_.A, (int)(A_init * no))},
                             torch::TensorOptions().dtype(torch::kUInt8).device(m_.dev));
    auto new_anchor = torch::empty({max_new, 3}, f);
    auto new_feat = torch::empty({max_new, m_.dims.feat_dim}, f);
    auto rnd = rands[i].contiguous();
    check(segs_anchor_growing_level((int)m_.A, (int)A_init, no, m_.dims.feat_dim, fp(m_.param("anchor")), fp(m_.param("offset")),
                                    fp(m_.param("scaling")), fp(m_.param("anchor_feat")), fp(grads), mask_u8.data_ptr<uint8_t>(), fp(rnd),
                                    cur_threshold, (float)std::pow(0.5, i + 1), cur_size, (int)max_new, fp(new_anchor), fp(new_feat),
                                    n_new_dev.data_ptr<int>(), (char*)temp.data_ptr(), st),
          "segs_anchor_growing_level");
    const int64_t n_new = n_new_dev.item<int>();
    if (n_new > 0) append(new_anchor.slice(0, 0, n_new), new_feat.slice(0, 0, n_new), cur_size);
  }
}

torch::Tensor AnchorDensifier::adjust_anchor(at::Generator generator, int views_per_iteration) {
  const int64_t no = m_.dims.n_offsets;
  const double check_interval = (double)p_.update_interval * views_per_iteration;
  const int64_t A_init = m_.A;
  std::vector<torch::Tensor> rands;
  for (int i = 0; i < p_.update_depth; i++)
    rands.push_back(torch::rand({A_init * no}, generator, torch::TensorOptions().dtype(torch::kFloat32)).to(m_.dev));
  auto accum = stat("offset_gradient_accum"), denom = stat("offset_denom");
  auto grads = accum / denom;
  grads.masked_fill_(grads.isnan(), 0.0);
  auto grads_norm = torch::linalg_vector_norm(grads, 2, {-1});
  auto offset_mask = (denom > (float)(check_interval * p_.success_threshold * 0.5)).squeeze(1);
  anchor_growing(grads_norm, p_.densify_grad_threshold, offset_mask, rands);
  // counters of the offsets that were eligible restart; rows of the new anchors start at zero (:1714-1724)
  auto& s = stats_;
  s["offset_denom"].slice(0, 0, A_init * no).masked_fill_(offset_mask, 0.0);
  s["offset_gradient_accum"].slice(0, 0, A_init * no).masked_fill_(offset_mask, 0.0);
  s["offset_denom"].slice(0, A_init * no, m_.A * no).zero_();
  s["offset_gradient_accum"].slice(0, A_init * no, m_.A * no).zero_();
  const int64_t A = m_.A;
  auto opacity_accum = s["opacity_accum"].slice(0, 0, A), anchor_demon = s["anchor_demon"].slice(0, 0, A);
  auto prune_mask = opacity_accum < (float)p_.min_opacity * anchor_demon;
  auto anchors_mask = anchor_demon > (float)(check_interval * p_.success_threshold);
  prune_mask = prune_mask & anchors_mask;
  opacity_accum.masked_fill_(anchors_mask, 0.0);   // :1738-1748
  anchor_demon.masked_fill_(anchors_mask, 0.0);
  if (A > 0) prune_anchor(prune_mask);
  return prune_mask;
}

// :1505-1558 plus the row filtering of the counters (:1730-1754): stable compaction of every per-anchor row
void AnchorDensifier::prune_anchor(const torch::Tensor& mask) {
  const int64_t no = m_.dims.n_offsets;
  const int64_t A = m_.A;
  auto keep = torch::nonzero(~mask).squeeze(1);
  const int64_t A1 = keep.numel();
  const char* names[4] = {"anchor", "offset", "anchor_feat", "scaling"};
  for (const torch::Tensor* bucket : {&m_.params, &m_.exp_avg, &m_.exp_avg_sq, &m_.grads})
    for (const char* name : names) {
      auto v = m_.view(*bucket, name, A);
      v.slice(0, 0, A1).copy_(v.index_select(0, keep));
    }
  m_.rotation.slice(0, 0, A1).copy_(m_.rotation.slice(0, 0, A).index_select(0, keep));
  m_.opacity.slice(0, 0, A1).copy_(m_.opacity.slice(0, 0, A).index_select(0, keep));
  for (const char* k : {"opacity_accum", "anchor_demon"})
    stats_[k].slice(0, 0, A1).copy_(stats_[k].slice(0, 0, A).index_select(0, keep));
  for (const char* k : {"offset_gradient_accum", "offset_denom"})
    stats_[k].slice(0, 0, A1 * no).copy_(stats_[k].slice(0, 0, A * no).view({A, no}).index_select(0, keep).reshape({-1}));
  m_.A = A1;
  auto sc = m_.param("scaling");
  sc.slice(1, 3, 6).copy_(torch::clamp_max(sc.slice(1, 3, 6), 0.05));   // :1525-1532
}

}  // namespace segs_host
