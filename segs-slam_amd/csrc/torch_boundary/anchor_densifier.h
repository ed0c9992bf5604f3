// anchor_densifier.h -- anchor statistics and densification of the Scaffold model, C++/LibTorch-ROCm host.
//
// Host side of GaussianModel::training_statis / adjust_anchor / anchor_growing / prune_anchor
// (src/gaussian_model.cpp:1459-1503, 1701-1762, 1559-1699, 1505-1558) over the C ABI of include/segs_densify.h, on the
// candidate-domain layout of segs_neural.h: the per-iteration statistics and each growing level are fused HIP kernels
// (segs_training_statis_guarded, segs_anchor_growing_level); the tensor bookkeeping of adjust_anchor -- appending rows,
// resetting counters, pruning by boolean mask together with the Adam moments (densificationPostfix / the optimizer-state
// migration of :1034-1052, 1511-1524, 1663-1687, 1788-1811) -- is done with LibTorch device ops inside the model's
// capacity-sized buckets.  C++ twin of segs-slam_amd/densify.py::AnchorDensifier (same names, same order of operations).
#pragma once
#include <torch/torch.h>

#include <map>
#include <string>
#include <vector>

#include "gaussian_trainer.h"

namespace segs_host {

// Model.* / Optimization.* keys of cfg/gaussian_mapper/RGB-D/Replica/office0.yaml:13-18,130-137
struct DensifyParams {
  double voxel_size = 0.001;
  int update_depth = 3, update_init_factor = 16, update_hierachy_factor = 4;
  int64_t start_stat = 500, update_from = 1500, update_interval = 100, update_until = 25500;
  double min_opacity = 0.005, success_threshold = 0.8, densify_grad_threshold = 0.0002;
};

class AnchorDensifier {
 public:
  AnchorDensifier(ScaffoldModelState& model, const DensifyParams& params = DensifyParams());
  const DensifyParams& params() const { return p_; }

  // src/gaussian_model.cpp:1459-1503 in the candidate domain.  skip_flag: device address of the (all-reduced) overflow word:
  // the pass is then dropped on the device.  into_delta: keyframe-parallel ranks accumulate into the shadow that
  // reduce_statistics() sums over ranks.
  void training_statis(const torch::Tensor& neural_opacity, const torch::Tensor& visible_radii, const torch::Tensor& radii,
                       const torch::Tensor& dL_dmean2D, const uint32_t* skip_flag, bool into_delta, void* stream);
  // fold every rank's increments since the last call into the replicated accumulators (one all-reduce of the flat shadow)
  void reduce_statistics(KeyframeExchange* exchange);
  // :1701-1762.  The reference's torch::rand_like (:1568) is drawn from `generator` (a CPU generator seeded identically on
  // every rank).  views_per_iteration: N keyframes are accumulated per iteration under keyframe parallelism, so the
  // "seen in more than this fraction of the window" thresholds count views.  Returns the prune mask.
  torch::Tensor adjust_anchor(at::Generator generator, int views_per_iteration = 1);
  // views over the live rows, shaped like the reference's tensors: (A,1) / (A*n_offsets,1)
  torch::Tensor stat(const std::string& name);

 private:
  void alloc_stats(int64_t capacity);
  void append(const torch::Tensor& new_anchor, const torch::Tensor& new_feat, float cur_size);
  void anchor_growing(const torch::Tensor& grads, double threshold, const torch::Tensor& offset_mask, const std::vector<torch::Tensor>& rands);
  void prune_anchor(const torch::Tensor& mask);

  ScaffoldModelState& m_;
  DensifyParams p_;
  int64_t stats_capacity_ = 0;
  torch::Tensor stats_flat_, delta_flat_;
  std::map<std::string, torch::Tensor> stats_, delta_;
};

}  // namespace segs_host
