// boundary_test.cpp -- exercises the C++/LibTorch drop-in layer end to end on a GPU (driven by tests/test_cpp_boundary.py):
//   boundary_test <in.bin> <out.bin>
// in.bin : int32 P, W, H ; float tanfovx, tanfovy ; then float32 arrays bg(3) means3D(P,3) colors(P,3) opacity(P)
//          scales(P,3) rotations(P,4) view(16) proj(16) campos(3) dL(3,H,W) points_for_knn reuse means3D
// out.bin: int32 R ; image(3,H,W) ; radii(P) as float ; grads: means3D(P,3) means2D(P,3) opacity(P) scales(P,3) rotations(P,4)
//          colors(P,3) ; dist2(P) ; visible_filter radii(P) as float
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../../include/segs_raster.h"
#include "gaussian_rasterizer.h"

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

// boundary_test --bench <in.bin> <steps> <warmup> <flags>: throughput of the drop-in path exactly as an unchanged caller
// drives it -- GaussianRasterizer::forward (fresh output tensors, the three scratch byte tensors grown by the allocator
// callback, one host synchronisation on num_rendered per call, src/rasterize_points.cu:36-114) and the autograd backward
// (src/gaussian_rasterizer.cpp:91-154), gradients reset to undefined between iterations like optimizer.zero_grad(true)
// (src/gaussian_mapper.cpp:1029).  `flags`: segs_raster_set_flags bits (0 = the reference's lists bit for bit, 32 =
// SEGS_RASTER_TIGHT_BINNING).  Prints one JSON object.
static int bench_main(int argc, char** argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: boundary_test --bench in.bin steps warmup flags\n"); return 2; }
  std::ifstream f(argv[2], std::ios::binary);
  if (!f) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
  const int steps = std::atoi(argv[3]), warmup = std::atoi(argv[4]);
  const unsigned flags = (unsigned)std::strtoul(argv[5], nullptr, 0);
  int32_t hdr[3]; float tf[2];
  f.read(reinterpret_cast<char*>(hdr), 12);
  f.read(reinterpret_cast<char*>(tf), 8);
  const int P = hdr[0], W = hdr[1], H = hdr[2];
  auto bg = rd(f, {3}), means3D = rd(f, {P, 3}), colors = rd(f, {P, 3}), opacity = rd(f, {P, 1}), scales = rd(f, {P, 3}),
       rotations = rd(f, {P, 4}), view = rd(f, {4, 4}), proj = rd(f, {4, 4}), campos = rd(f, {3}), dL = rd(f, {3, H, W});
  std::vector<torch::Tensor*> leaves = {&means3D, &colors, &opacity, &scales, &rotations};
  for (auto* t : leaves) t->set_requires_grad(true);
  auto means2D = torch::zeros_like(means3D).set_requires_grad(true);
  GaussianRasterizationSettings rs(H, W, tf[0], tf[1], bg, 1.0f, view, proj, 0, campos, false);
  GaussianRasterizer rast(rs);
  torch::Tensor none;
  segs_raster_set_flags(flags);
  int64_t visible = 0;
  auto one = [&]() {
    auto out = rast.forward(means3D, means2D, opacity, false, true, true, true, false, none, colors, scales, rotations, none);
    std::get<0>(out).backward(dL);
    for (auto* t : leaves) t->mutable_grad() = torch::Tensor();
    means2D.mutable_grad() = torch::Tensor();
    return std::get<1>(out);
  };
  for (int i = 0; i < warmup; i++) one();
  torch::cuda::synchronize();
  const auto t0 = std::chrono::steady_clock::now();
  torch::Tensor radii;
  for (int i = 0; i < steps; i++) radii = one();
  torch::cuda::synchronize();
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  visible = (radii > 0).sum().item<int64_t>();
  std::printf("{\"iters_per_s\": %.3f, \"ms_per_step\": %.5f, \"steps\": %d, \"warmup\": %d, \"flags\": %u, \"P\": %d, "
              "\"P_visible\": %lld, \"width\": %d, \"height\": %d}\n", steps / sec, sec / steps * 1e3, steps, warmup, flags, P,
              (long long)visible, W, H);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::string(argv[1]) == "--bench") return bench_main(argc, argv);
  if (argc < 3) { std::fprintf(stderr, "usage: boundary_test in.bin out.bin | boundary_test --bench in.bin steps warmup flags\n"); return 2; }
  std::ifstream f(argv[1], std::ios::binary);
  int32_t hdr[3]; float tf[2];
  f.read(reinterpret_cast<char*>(hdr), 12);
  f.read(reinterpret_cast<char*>(tf), 8);
  const int P = hdr[0], W = hdr[1], H = hdr[2];
  auto bg = rd(f, {3}), means3D = rd(f, {P, 3}), colors = rd(f, {P, 3}), opacity = rd(f, {P, 1}), scales = rd(f, {P, 3}),
       rotations = rd(f, {P, 4}), view = rd(f, {4, 4}), proj = rd(f, {4, 4}), campos = rd(f, {3}), dL = rd(f, {3, H, W});
  for (auto* t : {&means3D, &opacity, &scales, &rotations}) t->set_requires_grad(true);
  // colours arrive as a strided view of a wider tensor (src/gaussian_renderer.cpp:319-324)
  auto wide = torch::cat({colors, torch::zeros({P, 19}, colors.options())}, 1).set_requires_grad(true);
  auto col_view = wide.index({torch::indexing::Slice(), torch::indexing::Slice(0, 3)});
  auto means2D = torch::zeros_like(means3D).set_requires_grad(true);
  GaussianRasterizationSettings rs(H, W, tf[0], tf[1], bg, 1.0f, view, proj, 0, campos, false);
  GaussianRasterizer rast(rs);
  torch::Tensor none;
  auto out = rast.forward(means3D, means2D, opacity, false, true, true, true, false, none, col_view, scales, rotations, none);
  auto image = std::get<0>(out), radii = std::get<1>(out);
  (image * dL).sum().backward();
  bool threw = false;
  try { rast.forward(means3D, means2D, opacity, true, true, true, true, false, none, col_view, scales, rotations, none); }
  catch (const std::runtime_error&) { threw = true; }
  auto vis = rast.visible_filter(means3D.detach(), true, true, false, scales.detach(), rotations.detach(), none);
  auto d2 = distCUDA2(means3D.detach());
  torch::cuda::synchronize();
  std::ofstream o(argv[2], std::ios::binary);
  int32_t R = threw ? 1 : 0;  // first word: the exactly-one-of check fired
  o.write(reinterpret_cast<const char*>(&R), 4);
  wr(o, image); wr(o, radii);
  wr(o, means3D.grad()); wr(o, means2D.grad()); wr(o, opacity.grad()); wr(o, scales.grad()); wr(o, rotations.grad());
  wr(o, wide.grad().index({torch::indexing::Slice(), torch::indexing::Slice(0, 3)}));
  wr(o, d2); wr(o, vis);
  std::printf("boundary_test ok P=%d %dx%d\n", P, W, H);
  return 0;
}
