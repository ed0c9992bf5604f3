// trainer_test.cpp -- drives segs_host::GaussianTrainerStep (gaussian_trainer.h) on a model read from a file; the Python test
// tests/test_cpp_trainer.py runs segs-slam_amd/neural_gaussians.py::ScaffoldTrainerStep on the same model and compares.
//   trainer_test <in.bin> <out.bin>
// in.bin : int32 A, W, H, appearance_dim, use_feat_bank, n_steps ; float tanfovx, tanfovy, scaling_reg_weight ; float32 arrays
//          anchor(A,3) offset(A,10,3) anchor_feat(A,32) scaling(A,6) mlp(block) view(16) proj(16) campos(3) pose7(7) gt(3,H,W)
// out.bin: float32 loss[n_steps] ; scaling_reg[n_steps] ; steps_taken ; resident passes ; params(flat bucket) ; image(3,H,W)
#include <cstdio>
#include <fstream>
#include <vector>

#include "gaussian_trainer.h"

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

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: trainer_test in.bin out.bin\n"); return 2; }
  std::ifstream f(argv[1], std::ios::binary);
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
  std::vector<float> losses, regs;
  int resident = 0;
  for (int it = 0; it < n_steps; it++) {
    auto loss = step.trainingOnce(kf, gt);
    losses.push_back(loss.item<float>());
    regs.push_back(step.scaling_reg().item<float>());
    resident += step.last_pass_resident() ? 1 : 0;
  }
  torch::cuda::synchronize();
  std::ofstream o(argv[2], std::ios::binary);
  o.write(reinterpret_cast<const char*>(losses.data()), losses.size() * 4);
  o.write(reinterpret_cast<const char*>(regs.data()), regs.size() * 4);
  const float extra[2] = {(float)step.steps_taken(), (float)resident};
  o.write(reinterpret_cast<const char*>(extra), 8);
  wr(o, step.params_flat());
  wr(o, step.image());
  std::printf("trainer_test ok A=%d %dx%d steps=%d (resident passes %d)\n", A, W, H, n_steps, resident);
  return 0;
}
