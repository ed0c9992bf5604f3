// params_driver.cpp -- thin C-ABI driver around three more pieces of the REFERENCE that compile from their own files against
// LibTorch (CPU) alone.  TEST INFRASTRUCTURE ONLY; nothing of the reference is copied: the headers are #included and
// src/gaussian_parameters.cpp is compiled from where they lie (oracle/Makefile), outputs go to oracle/_ref/ (git-ignored);
// tests/golden/make_params_golden.py turns the results into committed fixtures (tests/golden/reference_params.json, .npz).
//   * GaussianModelParams / GaussianOptimizationParams / GaussianPipelineParams default construction
//     (include/gaussian_parameters.h, src/gaussian_parameters.cpp) -> the values a key absent from no configuration would take
//   * general_utils::inverse_sigmoid (include/general_utils.h:26-29; build_rotation :31-60 allocates on kCUDA and cannot run here)
//   * sh_utils::eval_sh / RGB2SH / SH2RGB (include/sh_utils.h)
#include <torch/torch.h>

#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>

#include "include/gaussian_parameters.h"
#include "include/general_utils.h"
#include "include/sh_utils.h"

extern "C" int ref_default_params_json(char* buf, int cap) {
  GaussianModelParams m;
  GaussianOptimizationParams o;
  GaussianPipelineParams p;
  std::ostringstream s;
  s.precision(9);
#define KV(name, val) s << "\"" << name << "\": " << (val) << ", "
  s << "{\"model\": {";
  KV("sh_degree", m.sh_degree_); KV("white_background", (int)m.white_background_); KV("feat_dim", m.feat_dim); KV("n_offsets", m.n_offsets);
  KV("voxel_size", m.voxel_size); KV("update_depth", m.update_depth); KV("update_init_factor", m.update_init_factor);
  KV("update_hierachy_factor", m.update_hierachy_factor); KV("use_feat_bank", (int)m.use_feat_bank); KV("appearance_dim", m.appearance_dim);
  KV("add_opacity_dist", (int)m.add_opacity_dist); KV("add_cov_dist", (int)m.add_cov_dist); KV("add_color_dist", (int)m.add_color_dist);
  KV("embedding_dim", m.embedding_dim); KV("use_coarse_anchor", (int)m.use_coarse_anchor);
  s << "\"ratio\": " << m.ratio << "}, \"optimization\": {";
  KV("iterations", o.iterations_); KV("position_lr_init", o.position_lr_init_); KV("position_lr_final", o.position_lr_final_);
  KV("position_lr_delay_mult", o.position_lr_delay_mult_); KV("position_lr_max_steps", o.position_lr_max_steps_);
  KV("offset_lr_init", o.offset_lr_init); KV("offset_lr_final", o.offset_lr_final); KV("offset_lr_max_steps", o.offset_lr_max_steps);
  KV("feature_lr", o.feature_lr_); KV("opacity_lr", o.opacity_lr_); KV("scaling_lr", o.scaling_lr_); KV("rotation_lr", o.rotation_lr_);
  KV("mlp_opacity_lr_init", o.mlp_opacity_lr_init); KV("mlp_opacity_lr_final", o.mlp_opacity_lr_final); KV("mlp_opacity_lr_max_steps", o.mlp_opacity_lr_max_steps);
  KV("mlp_cov_lr_init", o.mlp_cov_lr_init); KV("mlp_cov_lr_final", o.mlp_cov_lr_final); KV("mlp_cov_lr_max_steps", o.mlp_cov_lr_max_steps);
  KV("mlp_color_lr_init", o.mlp_color_lr_init); KV("mlp_color_lr_final", o.mlp_color_lr_final); KV("mlp_color_lr_max_steps", o.mlp_color_lr_max_steps);
  KV("mlp_featurebank_lr_init", o.mlp_featurebank_lr_init); KV("mlp_featurebank_lr_final", o.mlp_featurebank_lr_final);
  KV("mlp_featurebank_lr_max_steps", o.mlp_featurebank_lr_max_steps);
  KV("appearance_lr_init", o.appearance_lr_init); KV("appearance_lr_final", o.appearance_lr_final); KV("appearance_lr_max_steps", o.appearance_lr_max_steps);
  KV("percent_dense", o.percent_dense_); KV("lambda_dssim", o.lambda_dssim_); KV("start_stat", o.start_stat); KV("update_from", o.update_from);
  KV("update_interval", o.update_interval); KV("update_until", o.update_until); KV("min_opacity", o.min_opacity);
  KV("success_threshold", o.success_threshold);
  s << "\"densify_grad_threshold\": " << o.densify_grad_threshold << "}, \"pipeline\": {";
  KV("convert_SHs", (int)p.convert_SHs_);
  s << "\"compute_cov3D\": " << (int)p.compute_cov3D_ << "}}";
#undef KV
  const std::string out = s.str();
  if ((int)out.size() + 1 > cap) return -1;
  std::memcpy(buf, out.c_str(), out.size() + 1);
  return (int)out.size();
}

extern "C" int ref_inverse_sigmoid(const float* x, int n, float* out) {
  auto t = torch::from_blob(const_cast<float*>(x), {n}, torch::kFloat32).clone();
  auto r = general_utils::inverse_sigmoid(t).contiguous();
  std::memcpy(out, r.data_ptr<float>(), sizeof(float) * (size_t)n);
  return 0;
}

// sh (P,3,K) [channel-major, coefficient last, as eval_sh indexes it], dirs (P,3) -> out (P,3) = eval_sh(deg, sh, dirs)
extern "C" int ref_eval_sh(int deg, const float* sh, const float* dirs, int P, int K, float* out) {
  try {
    auto s = torch::from_blob(const_cast<float*>(sh), {P, 3, K}, torch::kFloat32).clone();
    auto d = torch::from_blob(const_cast<float*>(dirs), {P, 3}, torch::kFloat32).clone();
    auto r = sh_utils::eval_sh(deg, s, d).contiguous();
    std::memcpy(out, r.data_ptr<float>(), sizeof(float) * (size_t)P * 3);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ref_eval_sh: %s\n", e.what());
    return 1;
  }
}

extern "C" int ref_rgb2sh(const float* rgb, int n, float* out, float* sh2rgb_of_out) {
  auto t = torch::from_blob(const_cast<float*>(rgb), {n}, torch::kFloat32).clone();
  auto r = sh_utils::RGB2SH(t).contiguous();
  std::memcpy(out, r.data_ptr<float>(), sizeof(float) * (size_t)n);
  for (int i = 0; i < n; i++) sh2rgb_of_out[i] = sh_utils::SH2RGB(out[i]);
  return 0;
}
