// keyframe_exchange.cpp -- see keyframe_exchange.h (C++ twin of segs-slam_amd/keyframe_parallel.py::BucketExchange).
#include "keyframe_exchange.h"

#include <algorithm>

#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime.h>

namespace segs_host {
namespace {
constexpr int64_t ALIGN = 4;   // shard boundaries fall on 16-byte boundaries (float4 accesses of the fused Adam)
}

KeyframeExchange::KeyframeExchange(int64_t n, torch::Device device, c10::intrusive_ptr<c10d::Backend> pg, bool sharded,
                                   bool single_rank_collectives)
    : pg_(std::move(pg)), n_(n), dev_(device) {
  if (pg_) { world_ = pg_->getSize(); rank_ = pg_->getRank(); }
  active_ = pg_ && (world_ > 1 || single_rank_collectives);
  sharded_ = sharded && active_;
  const int64_t per = (n_ + world_ - 1) / world_;
  shard_len_ = (per + ALIGN - 1) / ALIGN * ALIGN;
  flag_ = torch::zeros({1}, torch::TensorOptions().dtype(torch::kInt32).device(dev_));
  // Gloo (only used to rehearse the N > 1 path with every rank on one GPU) has no tensor-shaped reduce-scatter / all-gather
  // for device tensors: same RESULT from an all-reduce + slice and a list all-gather
  emulate_ = active_ && dev_.is_cuda() && pg_->getBackendName() != "nccl";
  if (sharded_) {
    auto f = torch::TensorOptions().dtype(torch::kFloat32).device(dev_);
    if (shard_len_ * world_ != n_) {   // the collectives need world * shard_len elements, the bucket has n
      send_ = torch::zeros({shard_len_ * world_}, f);
      full_ = torch::zeros({shard_len_ * world_}, f);
    }
    shard_ = torch::zeros({shard_len_}, f);
  }
}

std::pair<int64_t, int64_t> KeyframeExchange::shard_range(int r) const {
  if (!sharded_) return {0, n_};
  const int64_t rr = r < 0 ? rank_ : r;
  const int64_t lo = std::min(rr * shard_len_, n_);
  return {lo, std::min(lo + shard_len_, n_)};
}

std::vector<segs_adam_segment> KeyframeExchange::clip_segments(const std::vector<segs_adam_segment>& segments) const {
  const auto [lo, hi] = shard_range();
  std::vector<segs_adam_segment> out;
  for (const auto& s : segments) {
    const int64_t a = std::max<int64_t>(s.offset, lo), b = std::min<int64_t>(s.offset + s.count, hi);
    if (b > a) out.push_back({a, b - a, s.lr});
  }
  return out;
}

void KeyframeExchange::reduce_flag_async(const torch::Tensor& local_flag) {
  if (!active_) {   // with one rank the word itself is the guard: no copy, no launch
    local_ = local_flag;
    local_set_ = local_flag.defined();
    return;
  }
  if (local_flag.defined()) flag_.copy_(local_flag.reshape({1}));
  else flag_.zero_();
  std::vector<at::Tensor> ts{flag_};
  flag_work_ = pg_->allreduce(ts);
}

torch::Tensor KeyframeExchange::wait_flag() {
  if (!active_) {
    if (local_set_) return local_;
    return flag_;   // never written with one rank: stays zero
  }
  if (flag_work_) { flag_work_->wait(); flag_work_.reset(); }
  return flag_;
}

KeyframeExchange::~KeyframeExchange() {
  if (mirror_event_) (void)hipEventDestroy((hipEvent_t)mirror_event_);
}

void KeyframeExchange::mirror_flag() {
  torch::Tensor w = wait_flag();
  if (!dev_.is_cuda()) {
    mirror_host_ = w.to(torch::kCPU).clone();
    mirror_pending_ = true;
    return;
  }
  if (!mirror_host_.defined()) {
    mirror_host_ = torch::zeros({1}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
    hipEvent_t e;
    TORCH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "hipEventCreateWithFlags");
    mirror_event_ = e;
  }
  hipStream_t st = c10::hip::getCurrentHIPStream(dev_.index()).stream();
  TORCH_CHECK(hipMemcpyAsync(mirror_host_.data_ptr(), w.data_ptr(), 4, hipMemcpyDeviceToHost, st) == hipSuccess, "hipMemcpyAsync");
  TORCH_CHECK(hipEventRecord((hipEvent_t)mirror_event_, st) == hipSuccess, "hipEventRecord");
  mirror_pending_ = true;
}

bool KeyframeExchange::step_dropped() {
  if (!mirror_pending_) return false;
  mirror_pending_ = false;
  if (mirror_event_) TORCH_CHECK(hipEventSynchronize((hipEvent_t)mirror_event_) == hipSuccess, "hipEventSynchronize");
  return mirror_host_.data_ptr<int32_t>()[0] != 0;
}

void KeyframeExchange::all_reduce_sum(torch::Tensor t) {
  if (!active_) return;
  std::vector<at::Tensor> ts{t};
  pg_->allreduce(ts)->wait();
}

void KeyframeExchange::reduce_gradients(torch::Tensor grads, bool dense) {
  TORCH_CHECK(grads.numel() == n_, "gradient bucket size changed: rebuild the exchange");
  if (!active_) return;
  if (dense || !sharded_) { all_reduce_sum(grads); return; }
  const auto [lo, hi] = shard_range();
  if (emulate_) {
    // the reduce-scatter's result, outside the shard included: there the bucket keeps this rank's own contribution
    auto own = grads.clone();
    all_reduce_sum(grads);
    own.slice(0, lo, hi).copy_(grads.slice(0, lo, hi));
    grads.copy_(own);
    return;
  }
  torch::Tensor src = grads;
  if (send_.defined()) { send_.slice(0, 0, n_).copy_(grads); src = send_; }
  pg_->_reduce_scatter_base(shard_, src)->wait();
  grads.slice(0, lo, hi).copy_(shard_.slice(0, 0, hi - lo));
}

void KeyframeExchange::gather(torch::Tensor bucket) {
  if (!sharded_) return;
  TORCH_CHECK(bucket.numel() == n_, "bucket size changed: rebuild the exchange");
  const auto [lo, hi] = shard_range();
  if (hi - lo < shard_len_) shard_.slice(0, hi - lo, shard_len_).zero_();
  shard_.slice(0, 0, hi - lo).copy_(bucket.slice(0, lo, hi));
  if (emulate_) {
    std::vector<std::vector<at::Tensor>> outs(1);
    for (int r = 0; r < world_; r++) outs[0].push_back(torch::empty_like(shard_));
    std::vector<at::Tensor> ins{shard_};
    pg_->allgather(outs, ins)->wait();
    for (int r = 0; r < world_; r++) {
      const auto [a, b] = shard_range(r);
      bucket.slice(0, a, b).copy_(outs[0][r].slice(0, 0, b - a));
    }
  } else if (!full_.defined()) {
    pg_->_allgather_base(bucket, shard_)->wait();
  } else {
    pg_->_allgather_base(full_, shard_)->wait();
    bucket.copy_(full_.slice(0, 0, n_));
  }
}

}  // namespace segs_host
