// keyframe_exchange.h -- the one exchange of a keyframe-parallel mapper iteration in C++, over a c10d backend
// (c10d::ProcessGroupNCCL = RCCL over xGMI in production; any c10d::Backend works, tests rehearse with Gloo).
//
// C++ counterpart of segs-slam_amd/keyframe_parallel.py::BucketExchange -- same method names and the same two exchanges,
// WITHOUT three refinements of the Python side: the overflow word always takes its own (asynchronous) all-reduce instead of
// riding in the dense gradient all-reduce, `sharded` is an explicit bool (no "auto" by bucket size), and a frozen anchor
// segment is exchanged with the rest (no `offset`).  Results are the same; the fixed cost per step is the larger of the two
// measured in DESIGN.md section 6.
//   * FLAG: the resident rasterizer's overflow word of every rank, summed asynchronously right after the forward; the summed
//     word guards statistics and optimizer on the device (segs_training_statis_guarded, segs_adam_step_device), so every
//     replica drops the same steps and no rank synchronises with its device to find out;
//   * GRADIENT bucket: dense all-reduce + full Adam on every rank, or reduce-scatter -> Adam on the rank's flat shard
//     [r L/N, (r+1) L/N) -> all-gather of the parameters (same bytes per link, 1/N of the optimizer traffic).
//     `dense = true` on the iterations whose shard partition changes before the optimizer runs (a densification that
//     re-sizes the bucket): every element must then hold the sum.
// The reference has no multi-GPU path; what it would all-reduce is what optimizer_->step() consumes
// (src/gaussian_mapper.cpp:1027-1030, src/gaussian_trainer.cpp:115-116).
#pragma once
#include <torch/torch.h>
#include <torch/csrc/distributed/c10d/Backend.hpp>

#include <utility>
#include <vector>

#include "../../../include/segs_train.h"

namespace segs_host {

class KeyframeExchange {
 public:
  // `pg` may be null (single process, nothing is exchanged).  single_rank_collectives: issue the collectives even with one
  // rank (exercises the RCCL calls on a one-GPU box).
  KeyframeExchange(int64_t n, torch::Device device, c10::intrusive_ptr<c10d::Backend> pg, bool sharded = true,
                   bool single_rank_collectives = false);
  ~KeyframeExchange();
  KeyframeExchange(const KeyframeExchange&) = delete;
  KeyframeExchange& operator=(const KeyframeExchange&) = delete;

  int world() const { return world_; }
  int rank() const { return rank_; }
  bool sharded() const { return sharded_; }
  bool active() const { return active_; }
  int64_t size() const { return n_; }

  // [lo, hi) of the bucket rank r (default: this rank) updates; the whole bucket when not sharded
  std::pair<int64_t, int64_t> shard_range(int r = -1) const;
  // Adam segments intersected with this rank's shard
  std::vector<segs_adam_segment> clip_segments(const std::vector<segs_adam_segment>& segments) const;

  // overflow word: start its all-reduce (local_flag: 1-element int32 device tensor, or undefined = 0) / make the current
  // stream wait for it and return the device word
  void reduce_flag_async(const torch::Tensor& local_flag);
  torch::Tensor wait_flag();

  // The SUMMED word on the host, one step late, so that no iteration is lost with N > 1 ranks either: mirror_flag() queues a
  // copy of the summed word into pinned host memory (call after wait_flag()); step_dropped() waits for that copy and says
  // whether some rank's pass was invalid (false when nothing was mirrored since the last call).  Every rank mirrors the same
  // word, so every rank takes the same decision without another collective.
  void mirror_flag();
  bool step_dropped();

  void reduce_gradients(torch::Tensor grads, bool dense = false);
  void gather(torch::Tensor bucket);
  // plain sum over ranks (the densification statistics' shadow)
  void all_reduce_sum(torch::Tensor t);

 private:
  c10::intrusive_ptr<c10d::Backend> pg_;
  int world_ = 1, rank_ = 0;
  int64_t n_, shard_len_;
  torch::Device dev_;
  bool active_ = false, sharded_ = false, emulate_ = false;
  torch::Tensor flag_, local_, send_, shard_, full_;
  c10::intrusive_ptr<c10d::Work> flag_work_;
  bool local_set_ = false;
  torch::Tensor mirror_host_;
  void* mirror_event_ = nullptr;
  bool mirror_pending_ = false;
};

}  // namespace segs_host
