// trainer_test.cpp -- drives segs_host::GaussianTrainerStep (gaussian_trainer.h) on a model read from a file; the Python tests
// tests/test_cpp_trainer.py run segs-slam_amd/neural_gaussians.py::ScaffoldTrainerStep on the same model and compare.
//
//   trainer_test <in.bin> <out.bin>
//     in.bin : int32 A, W, H, appearance_dim, use_feat_bank, n_steps ; float tanfovx, tanfovy, scaling_reg_weight ; float32 arrays
//              anchor(A,3) offset(A,10,3) anchor_feat(A,32) scaling(A,6) mlp(block) view(16) proj(16) campos(3) pose7(7) gt(3,H,W)
//     out.bin: float32 loss[n_steps] ; scaling_reg[n_steps] ; steps_taken ; resident passes ; params(flat bucket) ; image(3,H,W)
//
//   trainer_test --mapper <in.bin> <out.bin> [--world N --rank r --store FILE [--dense] | --nccl1 FILE]
//     the mapper loop with densification (anchor_densifier.h) and, with --world, the keyframe-parallel exchange
//     (keyframe_exchange.h) between N processes of this program, rank r rendering keyframe (iteration * N + r) mod n_keyframes.
//     in.bin : int32 A, W, H, appearance_dim, use_feat_bank, n_steps, n_keyframes, start_stat, update_from, update_interval, seed ;
//              float tanfovx, tanfovy, scaling_reg_weight, voxel_size, densify_grad_threshold ; model arrays as above ;
//              per keyframe: view(16) proj(16) campos(3) pose7(7) gt(3,H,W)
//     out.bin: int32 A_final, capacity, mlp_steps, anchor_steps, A after every step [n_steps] ; float32 loss[n_steps] ;
//              live rows of anchor / offset / anchor_feat / scaling, the MLP block, the four statistics (live rows)
//     --world: the collectives run over a c10d::Backend defined HERE on top of c10d::FileStore (host staging, rank-ordered sums:
//     a rehearsal transport for a one-GPU box, where RCCL refuses two ranks on one device); --nccl1: a ONE-rank
//     c10d::ProcessGroupNCCL with every collective forced (reduce_scatter / all_gather through RCCL from C++).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <torch/csrc/distributed/c10d/FileStore.hpp>
#ifdef SEGS_TEST_WITH_NCCL
#include <torch/csrc/distributed/c10d/ProcessGroupNCCL.hpp>
#endif

#include "anchor_densifier.h"
#include "gaussian_trainer.h"
#include "../../../include/segs_raster.h"
#include "keyframe_exchange.h"

static torch::Tensor rd(std::ifstream& f, std::vector<int64_t> shape) {
  int64_t n = 1;
  for (auto s : shape) n *= s;
  torch::Tensor t = torch::empty(shape, torch::kFloat32);
  f.read(reinterpret_cast<char*>(t.data_ptr<float>()), n * 4);
  return t.to(torch::kCUDA);
}
static void wr(std::ofstream& f, const torch::Tensor& t) {
  auto c = t.detach().to(torch::kFloat32).contiguous().cpu();
  f.write(reinterpret_cast<const char*>(c.data_ptr<float>()), c.numel() * 4);
}

namespace {
// A c10d::Backend over a Store: every rank publishes its tensor's bytes, reads every rank's, and combines them in rank order
// (so every rank computes the same words).  Test transport only.
class StoreWork : public c10d::Work {
 public:
  StoreWork() : c10d::Work(-1, c10d::OpType::UNKNOWN) { finish(); }
  bool wait(std::chrono::milliseconds) override { return true; }
};
class StoreBackend : public c10d::Backend {
 public:
  StoreBackend(c10::intrusive_ptr<c10d::Store> store, int rank, int size) : c10d::Backend(rank, size), store_(std::move(store)) {}
  const std::string getBackendName() const override { return "store"; }
  c10::intrusive_ptr<c10d::Work> allreduce(std::vector<at::Tensor>& ts, const c10d::AllreduceOptions& = c10d::AllreduceOptions()) override {
    auto parts = exchange(ts[0]);
    auto sum = parts[0].clone();
    for (int r = 1; r < getSize(); r++) sum += parts[r];
    ts[0].copy_(sum.to(ts[0].device()));
    return c10::make_intrusive<StoreWork>();
  }
  c10::intrusive_ptr<c10d::Work> allgather(std::vector<std::vector<at::Tensor>>& outs, std::vector<at::Tensor>& ins,
                                          const c10d::AllgatherOptions& = c10d::AllgatherOptions()) override {
    auto parts = exchange(ins[0]);
    for (int r = 0; r < getSize(); r++) outs[0][r].copy_(parts[r].to(outs[0][r].device()));
    return c10::make_intrusive<StoreWork>();
  }

 private:
  std::vector<at::Tensor> exchange(const at::Tensor& t) {
    auto host = t.detach().contiguous().cpu();
    const size_t bytes = host.numel() * host.element_size();
    const std::string base = "x" + std::to_string(seq_++) + "_";
    std::vector<uint8_t> mine((const uint8_t*)host.data_ptr(), (const uint8_t*)host.data_ptr() + bytes);
    store_->set(base + std::to_string(getRank()), mine);
    std::vector<at::Tensor> parts;
    for (int r = 0; r < getSize(); r++) {
      auto v = store_->get(base + std::to_string(r));
      TORCH_CHECK(v.size() == bytes, "store exchange: size mismatch");
      auto p = torch::empty_like(host);
      std::memcpy(p.data_ptr(), v.data(), bytes);
      parts.push_back(p);
    }
    return parts;
  }
  c10::intrusive_ptr<c10d::Store> store_;
  int64_t seq_ = 0;
};
}  // namespace

static int run_single(const char* in, const char* out) {
  std::ifstream f(in, std::ios::binary);
  int32_t hdr[6]; float tf[3];
  f.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
  f.read(reinterpret_cast<char*>(tf), sizeof(tf));
  const int A = hdr[0], W = hdr[1], H = hdr[2], n_steps = hdr[5];
  segs_host::ScaffoldDims dims;
  dims.appearance_dim = hdr[3];
  dims.use_feat_bank = hdr[4] != 0;
  segs_host::GaussianTrainerStep step(A, dims, W, H, torch::Device(torch::kCUDA, 0), segs_host::ScaffoldOptimization(), tf[2]);
  if (std::getenv("SEGS_TRAINER_TEST_UNFUSED")) step.set_fuse_projection(false);   // A/B of tests/test_cpp_trainer.py: K1 and the prefilter as kernels of their own
  step.param("anchor").copy_(rd(f, {A, 3}));
  step.param("offset").copy_(rd(f, {A, dims.n_offsets, 3}));
  step.param("anchor_feat").copy_(rd(f, {A, dims.feat_dim}));
  step.param("scaling").copy_(rd(f, {A, 6}));
  step.mlp_params().copy_(rd(f, {step.mlp_params().numel()}));
  segs_host::KeyframeView kf;
  kf.view = rd(f, {4, 4}); kf.proj = rd(f, {4, 4}); kf.campos = rd(f, {3}); kf.pose7 = rd(f, {7});
  kf.tanfovx = tf[0]; kf.tanfovy = tf[1];
  auto gt = rd(f, {3, H, W});
  std::vector<float> losses, regs;
  int resident = 0;
  // test hooks (tests/test_cpp_trainer.py): SEGS_TRAINER_TEST_OVERFLOW_AT=k shrinks the resident capacity in front of
  // iteration k so that its forward overflows and the device drops it; SEGS_TRAINER_TEST_NO_REDO keeps it dropped
  const int overflow_at = std::getenv("SEGS_TRAINER_TEST_OVERFLOW_AT") ? std::atoi(std::getenv("SEGS_TRAINER_TEST_OVERFLOW_AT")) : -1;
  if (std::getenv("SEGS_TRAINER_TEST_NO_REDO")) step.set_redo_dropped_steps(false);
  for (int it = 0; it < n_steps; it++) {
    if (it == overflow_at) step.debug_shrink_capacity(3);
    auto loss = step.trainingOnce(kf, gt);
    losses.push_back(loss.item<float>());
    regs.push_back(step.scaling_reg().item<float>());
    resident += step.last_pass_resident() ? 1 : 0;
  }
  torch::cuda::synchronize();
  std::ofstream o(out, std::ios::binary);
  o.write(reinterpret_cast<const char*>(losses.data()), losses.size() * 4);
  o.write(reinterpret_cast<const char*>(regs.data()), regs.size() * 4);
  const float extra[2] = {(float)step.steps_taken(), (float)resident};
  o.write(reinterpret_cast<const char*>(extra), 8);
  wr(o, step.params_flat());
  wr(o, step.image());
  std::printf("trainer_test ok A=%d %dx%d steps=%d (resident passes %d) redone %d\n", A, W, H, n_steps, resident, (int)step.redone_steps());
  return 0;
}

// trainer_test --chain <in.bin> <out.bin>: what tests/test_cpp_trainer.py needs to hold the C++ host against the float64 /
// oracle chain DIRECTLY (not through the Python step): in.bin as for the plain mode, followed by a (H,W) float mask that
// dL/dimage is multiplied with (the oracle's stable pixels) and int32 freq_start, freq_until + float lambda_high (0 = no
// frequency regulariser; scales 1, 1/2, 1/4).  out.bin per step: loss, scaling_reg, freq_loss ; then per step the gradient
// bucket as the optimizer received it and the parameter bucket after the step ; visible_radii (A) as float, image of the
// FIRST step (3,H,W).
static int run_chain(const char* in, const char* out) {
  std::ifstream f(in, std::ios::binary);
  int32_t hdr[6]; float tf[3];
  f.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
  f.read(reinterpret_cast<char*>(tf), sizeof(tf));
  const int A = hdr[0], W = hdr[1], H = hdr[2], n_steps = hdr[5];
  segs_host::ScaffoldDims dims;
  dims.appearance_dim = hdr[3];
  dims.use_feat_bank = hdr[4] != 0;
  segs_host::GaussianTrainerStep step(A, dims, W, H, torch::Device(torch::kCUDA, 0), segs_host::ScaffoldOptimization(), tf[2]);
  step.param("anchor").copy_(rd(f, {A, 3}));
  step.param("offset").copy_(rd(f, {A, dims.n_offsets, 3}));
  step.param("anchor_feat").copy_(rd(f, {A, dims.feat_dim}));
  step.param("scaling").copy_(rd(f, {A, 6}));
  step.mlp_params().copy_(rd(f, {step.mlp_params().numel()}));
  segs_host::KeyframeView kf;
  kf.view = rd(f, {4, 4}); kf.proj = rd(f, {4, 4}); kf.campos = rd(f, {3}); kf.pose7 = rd(f, {7});
  kf.tanfovx = tf[0]; kf.tanfovy = tf[1];
  auto gt = rd(f, {3, H, W});
  step.set_image_gradient_mask(rd(f, {H, W}));
  int32_t fr[2]; float lam;
  f.read(reinterpret_cast<char*>(fr), sizeof(fr));
  f.read(reinterpret_cast<char*>(&lam), 4);
  if (lam != 0.f) step.enable_frequency_regularization(lam, {1.0f, 0.5f, 0.25f}, fr[0], fr[1], true);
  std::vector<torch::Tensor> grads, params;
  step.set_on_gradients([&](const torch::Tensor& g) { grads.push_back(g.clone()); });
  std::vector<float> scal;
  torch::Tensor first_image, first_radii;
  for (int it = 0; it < n_steps; it++) {
    auto loss = step.trainingOnce(kf, gt);
    scal.push_back(loss.item<float>());
    scal.push_back(step.scaling_reg().item<float>());
    scal.push_back(lam != 0.f ? step.frequency_loss().item<float>() : 0.f);
    params.push_back(step.params_flat().clone());
    if (it == 0) { first_image = step.image().clone(); first_radii = step.visible_radii().slice(0, 0, A).clone(); }
  }
  torch::cuda::synchronize();
  std::ofstream o(out, std::ios::binary);
  o.write(reinterpret_cast<const char*>(scal.data()), scal.size() * 4);
  for (int it = 0; it < n_steps; it++) { wr(o, grads[it]); wr(o, params[it]); }
  wr(o, first_radii); wr(o, first_image);
  std::printf("trainer_test --chain ok A=%d %dx%d steps=%d\n", A, W, H, n_steps);
  return 0;
}

// trainer_test --freq-cache <in.bin> <out.bin>: the |FFT(target)| cache of the frequency regulariser against the host pattern of
// the reference's mapper (src/gaussian_mapper.cpp:845,921: a FRESH target tensor every iteration, the previous one freed).  Three
// iterations: target A; A freed and target B allocated (the caching allocator would hand B the address of A); B refreshed IN PLACE
// with image A.  After each, the regulariser is re-evaluated here with a table made from the target the step was given -- on the
// image the step rendered in that iteration -- and must equal the step's value bit for bit.  out.bin: 3 x (step value, direct value).
static int run_freq_cache(const char* in, const char* out) {
  std::ifstream f(in, std::ios::binary);
  int32_t hdr[6]; float tf[3];
  f.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
  f.read(reinterpret_cast<char*>(tf), sizeof(tf));
  const int A = hdr[0], W = hdr[1], H = hdr[2];
  segs_host::ScaffoldDims dims;
  dims.appearance_dim = hdr[3];
  dims.use_feat_bank = hdr[4] != 0;
  segs_host::GaussianTrainerStep step(A, dims, W, H, torch::Device(torch::kCUDA, 0), segs_host::ScaffoldOptimization(), tf[2]);
  step.param("anchor").copy_(rd(f, {A, 3}));
  step.param("offset").copy_(rd(f, {A, dims.n_offsets, 3}));
  step.param("anchor_feat").copy_(rd(f, {A, dims.feat_dim}));
  step.param("scaling").copy_(rd(f, {A, 6}));
  step.mlp_params().copy_(rd(f, {step.mlp_params().numel()}));
  segs_host::KeyframeView kf;
  kf.view = rd(f, {4, 4}); kf.proj = rd(f, {4, 4}); kf.campos = rd(f, {3}); kf.pose7 = rd(f, {7});
  kf.tanfovx = tf[0]; kf.tanfovy = tf[1];
  auto img_a = rd(f, {3, H, W}).cpu();
  auto img_b = img_a.flip({2}).mul(0.5f).add(0.25f).contiguous();          // another image of the same size
  const std::vector<float> scales{1.0f, 0.5f, 0.25f};
  const float lam = 0.01f;
  step.enable_frequency_regularization(lam, scales, -1, 1 << 30, true);
  step.set_frequency_target_cache(2);
  segs_freq_plan* plan = nullptr;
  if (segs_freq_plan_create(H, W, 3, scales.data(), lam, &plan) != 0) { std::fprintf(stderr, "segs_freq_plan_create: %s\n", segs_last_error()); return 1; }
  auto fopt = torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA);
  auto table = torch::empty({(int64_t)segs_freq_target_floats(plan)}, fopt);
  auto direct = torch::zeros({1}, fopt), scratch = torch::zeros({3, H, W}, fopt);
  void* st = (void*)c10::hip::getCurrentHIPStream().stream();
  std::vector<float> vals;
  const void* addr_a = nullptr;
  int same_address = 0;
  auto once = [&](const torch::Tensor& gt) {
    step.trainingOnce(kf, gt);
    if (segs_freq_target(plan, gt.data_ptr<float>(), table.data_ptr<float>(), st) != 0 ||
        segs_freq_loss(plan, step.image().data_ptr<float>(), table.data_ptr<float>(), scratch.data_ptr<float>(), direct.data_ptr<float>(), nullptr, st) != 0) {
      std::fprintf(stderr, "direct evaluation: %s\n", segs_last_error());
      std::exit(1);
    }
    vals.push_back(step.frequency_loss().item<float>());
    vals.push_back(direct.item<float>());
  };
  {
    auto gt = img_a.to(torch::kCUDA);
    addr_a = gt.data_ptr();
    once(gt);
  }                                            // the host's handle on target A is gone
  auto gt2 = img_b.to(torch::kCUDA);           // without an owning cache entry this lands on A's address
  same_address = gt2.data_ptr() == addr_a;
  once(gt2);
  gt2.copy_(img_a.to(torch::kCUDA));           // same tensor, same address, new contents
  once(gt2);
  torch::cuda::synchronize();
  segs_freq_plan_destroy(plan);
  std::ofstream o(out, std::ios::binary);
  o.write(reinterpret_cast<const char*>(vals.data()), vals.size() * 4);
  std::printf("trainer_test --freq-cache ok second target at the first one's address: %d, cached %d\n", same_address, (int)step.frequency_targets_cached());
  return 0;
}

static int run_mapper(int argc, char** argv) {
  const char* in = argv[2];
  const char* out = argv[3];
  int world = 1, rank = 0;
  bool dense = false, nccl1 = false;
  std::string store_path;
  for (int i = 4; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--world" && i + 1 < argc) world = std::atoi(argv[++i]);
    else if (a == "--rank" && i + 1 < argc) rank = std::atoi(argv[++i]);
    else if (a == "--store" && i + 1 < argc) store_path = argv[++i];
    else if (a == "--nccl1" && i + 1 < argc) { nccl1 = true; store_path = argv[++i]; }
    else if (a == "--dense") dense = true;
  }
  std::ifstream f(in, std::ios::binary);
  int32_t hdr[11]; float tf[5];
  f.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
  f.read(reinterpret_cast<char*>(tf), sizeof(tf));
  const int A = hdr[0], W = hdr[1], H = hdr[2], n_steps = hdr[5], n_kf = hdr[6];
  segs_host::ScaffoldDims dims;
  dims.appearance_dim = hdr[3];
  dims.use_feat_bank = hdr[4] != 0;
  const torch::Device dev(torch::kCUDA, 0);
  segs_host::GaussianTrainerStep step(A, dims, W, H, dev, segs_host::ScaffoldOptimization(), tf[2]);
  step.param("anchor").copy_(rd(f, {A, 3}));
  step.param("offset").copy_(rd(f, {A, dims.n_offsets, 3}));
  step.param("anchor_feat").copy_(rd(f, {A, dims.feat_dim}));
  step.param("scaling").copy_(rd(f, {A, 6}));
  step.mlp_params().copy_(rd(f, {step.mlp_params().numel()}));
  std::vector<segs_host::KeyframeView> kfs(n_kf);
  std::vector<torch::Tensor> gts(n_kf);
  for (int k = 0; k < n_kf; k++) {
    kfs[k].view = rd(f, {4, 4}); kfs[k].proj = rd(f, {4, 4}); kfs[k].campos = rd(f, {3}); kfs[k].pose7 = rd(f, {7});
    kfs[k].tanfovx = tf[0]; kfs[k].tanfovy = tf[1];
    gts[k] = rd(f, {3, H, W});
  }
  segs_host::DensifyParams dp;
  dp.start_stat = hdr[7]; dp.update_from = hdr[8]; dp.update_interval = hdr[9]; dp.update_until = 1000000000;
  dp.voxel_size = (double)tf[3]; dp.densify_grad_threshold = (double)tf[4];
  segs_host::AnchorDensifier dens(step.model(), dp);
  step.enable_densification(&dens, (uint64_t)hdr[10]);
  c10::intrusive_ptr<c10d::Backend> pg;
  if (nccl1) {
#ifdef SEGS_TEST_WITH_NCCL
    auto store = c10::make_intrusive<c10d::FileStore>(store_path, 1);
    auto opts = c10d::ProcessGroupNCCL::Options::create();
    pg = c10::make_intrusive<c10d::ProcessGroupNCCL>(store, 0, 1, opts);
    step.set_process_group(pg, !dense, /*single_rank_collectives=*/true);
#else
    std::fprintf(stderr, "built without SEGS_TEST_WITH_NCCL\n");
    return 3;
#endif
  } else if (world > 1) {
    auto store = c10::make_intrusive<c10d::FileStore>(store_path, world);
    pg = c10::make_intrusive<StoreBackend>(store, rank, world);
    step.set_process_group(pg, !dense);
  }
  std::vector<float> losses;
  std::vector<int32_t> sizes;
  // test hook (tests/test_cpp_trainer.py): SEGS_TRAINER_TEST_OVERFLOW_AT=k with SEGS_TRAINER_TEST_OVERFLOW_RANK=r shrinks rank r's
  // resident capacity in front of loop index k, so that ITS forward overflows and every rank's device drops that pass
  const int overflow_at = std::getenv("SEGS_TRAINER_TEST_OVERFLOW_AT") ? std::atoi(std::getenv("SEGS_TRAINER_TEST_OVERFLOW_AT")) : -1;
  const int overflow_rank = std::getenv("SEGS_TRAINER_TEST_OVERFLOW_RANK") ? std::atoi(std::getenv("SEGS_TRAINER_TEST_OVERFLOW_RANK")) : 0;
  for (int it = 0; it < n_steps; it++) {
    const int k = (it * world + rank) % n_kf;
    if (it == overflow_at && rank == overflow_rank) step.debug_shrink_capacity(3);
    auto loss = step.trainingOnce(kfs[k], gts[k]);
    losses.push_back(loss.item<float>());
    sizes.push_back((int32_t)step.model().A);
  }
  step.finish();
  torch::cuda::synchronize();
  auto& m = step.model();
  std::ofstream o(out, std::ios::binary);
  const int32_t head[4] = {(int32_t)m.A, (int32_t)m.capacity, (int32_t)step.steps_taken(), (int32_t)step.anchor_steps_taken()};
  o.write(reinterpret_cast<const char*>(head), sizeof(head));
  o.write(reinterpret_cast<const char*>(sizes.data()), sizes.size() * 4);
  o.write(reinterpret_cast<const char*>(losses.data()), losses.size() * 4);
  for (const char* name : {"anchor", "offset", "anchor_feat", "scaling"}) wr(o, m.param(name));
  wr(o, m.mlp_params());
  for (const char* name : {"opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"}) wr(o, dens.stat(name));
  std::printf("trainer_test --mapper ok rank %d/%d A %d -> %d capacity %d steps %d/%d redone %d\n", rank, world, A, (int)m.A, (int)m.capacity,
              (int)step.steps_taken(), (int)step.anchor_steps_taken(), (int)step.redone_steps());
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 4 && std::string(argv[1]) == "--mapper") return run_mapper(argc, argv);
  if (argc >= 4 && std::string(argv[1]) == "--chain") return run_chain(argv[2], argv[3]);
  if (argc >= 4 && std::string(argv[1]) == "--freq-cache") return run_freq_cache(argv[2], argv[3]);
  if (argc < 3) { std::fprintf(stderr, "usage: trainer_test in.bin out.bin | trainer_test --mapper in.bin out.bin [...]\n"); return 2; }
  return run_single(argv[1], argv[2]);
}
