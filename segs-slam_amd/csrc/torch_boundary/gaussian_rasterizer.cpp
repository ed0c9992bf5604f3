// gaussian_rasterizer.cpp -- mirror of the reference's src/gaussian_rasterizer.cpp:19-307: autograd Function
// (what is saved, gradient order :143-153), exactly-one-of argument checks (:175-181), "absent" = empty tensor.
#include "gaussian_rasterizer.h"

namespace {
void check_inputs(bool has_shs, bool has_colors_precomp, bool has_scales, bool has_rotations, bool has_cov3D_precomp) {
  if ((!has_shs && !has_colors_precomp) || (has_shs && has_colors_precomp))
    throw std::runtime_error("Please provide excatly one of either SHs or precomputed colors!");
  if (((!has_scales || !has_rotations) && !has_cov3D_precomp) || ((has_scales || has_rotations) && has_cov3D_precomp))
    throw std::runtime_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
}
torch::Tensor absent(const torch::Tensor& like) { return torch::empty({0}, like.options().dtype(torch::kFloat32)); }
}  // namespace

torch::Tensor GaussianRasterizer::markVisibleGaussians(torch::Tensor& positions) {
  torch::NoGradGuard no_grad;
  return markVisible(positions, raster_settings_.viewmatrix_, raster_settings_.projmatrix_);
}

torch::autograd::tensor_list GaussianRasterizerFunction::forward(torch::autograd::AutogradContext* ctx, torch::Tensor means3D,
                                                                 torch::Tensor means2D, torch::Tensor sh,
                                                                 torch::Tensor colors_precomp, torch::Tensor opacities,
                                                                 torch::Tensor scales, torch::Tensor rotations,
                                                                 torch::Tensor cov3Ds_precomp,
                                                                 GaussianRasterizationSettings rs) {
  (void)means2D;
  auto res = RasterizeGaussiansCUDA(rs.bg_, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier_, cov3Ds_precomp,
                                    rs.viewmatrix_, rs.projmatrix_, rs.tanfovx_, rs.tanfovy_, rs.image_height_, rs.image_width_, sh,
                                    rs.sh_degree_, rs.campos_, rs.prefiltered_);
  auto color = std::get<1>(res);
  auto radii = std::get<2>(res);
  ctx->saved_data["num_rendered"] = std::get<0>(res);
  ctx->saved_data["scale_modifier"] = rs.scale_modifier_;
  ctx->saved_data["tanfovx"] = rs.tanfovx_;
  ctx->saved_data["tanfovy"] = rs.tanfovy_;
  ctx->saved_data["sh_degree"] = rs.sh_degree_;
  ctx->save_for_backward({rs.bg_, rs.viewmatrix_, rs.projmatrix_, rs.campos_, colors_precomp, means3D, scales, rotations,
                          cov3Ds_precomp, radii, sh, std::get<3>(res), std::get<4>(res), std::get<5>(res)});
  return {color, radii};
}

torch::autograd::tensor_list GaussianRasterizerFunction::backward(torch::autograd::AutogradContext* ctx,
                                                                  torch::autograd::tensor_list grad_outputs) {
  auto num_rendered = ctx->saved_data["num_rendered"].toInt();
  auto scale_modifier = static_cast<float>(ctx->saved_data["scale_modifier"].toDouble());
  auto tanfovx = static_cast<float>(ctx->saved_data["tanfovx"].toDouble());
  auto tanfovy = static_cast<float>(ctx->saved_data["tanfovy"].toDouble());
  auto sh_degree = ctx->saved_data["sh_degree"].toInt();
  auto s = ctx->get_saved_variables();
  auto r = RasterizeGaussiansBackwardCUDA(s[0], s[5], s[9], s[4], s[6], s[7], scale_modifier, s[8], s[1], s[2], tanfovx, tanfovy,
                                          grad_outputs[0], s[10], sh_degree, s[3], s[11], num_rendered, s[12], s[13]);
  // means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings
  return {std::get<3>(r), std::get<0>(r), std::get<5>(r), std::get<1>(r), std::get<2>(r),
          std::get<6>(r), std::get<7>(r), std::get<4>(r), torch::Tensor()};
}

std::tuple<torch::Tensor, torch::Tensor> GaussianRasterizer::forward(torch::Tensor means3D, torch::Tensor means2D,
                                                                     torch::Tensor opacities, bool has_shs, bool has_colors_precomp,
                                                                     bool has_scales, bool has_rotations, bool has_cov3D_precomp,
                                                                     torch::Tensor shs, torch::Tensor colors_precomp,
                                                                     torch::Tensor scales, torch::Tensor rotations,
                                                                     torch::Tensor cov3D_precomp) {
  check_inputs(has_shs, has_colors_precomp, has_scales, has_rotations, has_cov3D_precomp);
  if (!has_shs) shs = absent(means3D);
  if (!has_colors_precomp) colors_precomp = absent(means3D);
  if (!has_scales) scales = absent(means3D);
  if (!has_rotations) rotations = absent(means3D);
  if (!has_cov3D_precomp) cov3D_precomp = absent(means3D);
  auto result = rasterizeGaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, raster_settings_);
  return std::make_tuple(result[0], result[1]);
}

torch::Tensor GaussianRasterizer::visible_filter(torch::Tensor means3D, bool has_scales, bool has_rotations, bool has_cov3D_precomp,
                                                 torch::Tensor scales, torch::Tensor rotations, torch::Tensor cov3D_precomp) {
  auto& rs = raster_settings_;
  if (!has_scales) scales = absent(means3D);
  if (!has_rotations) rotations = absent(means3D);
  if (!has_cov3D_precomp) cov3D_precomp = absent(means3D);
  torch::NoGradGuard no_grad;
  return RasterizeGaussiansfilterCUDA(means3D, scales, rotations, rs.scale_modifier_, cov3D_precomp, rs.viewmatrix_, rs.projmatrix_,
                                      rs.tanfovx_, rs.tanfovy_, rs.image_height_, rs.image_width_, rs.prefiltered_, false);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> GaussianRasterizer::project2_image(
    torch::Tensor means3D, torch::Tensor means2D, torch::Tensor opacities, bool has_shs, bool has_colors_precomp, bool has_scales,
    bool has_rotations, bool has_cov3D_precomp, torch::Tensor shs, torch::Tensor colors_precomp, torch::Tensor scales,
    torch::Tensor rotations, torch::Tensor cov3D_precomp) {
  (void)means2D;
  check_inputs(has_shs, has_colors_precomp, has_scales, has_rotations, has_cov3D_precomp);
  auto& rs = raster_settings_;
  if (!has_shs) shs = absent(means3D);
  if (!has_colors_precomp) colors_precomp = absent(means3D);
  if (!has_scales) scales = absent(means3D);
  if (!has_rotations) rotations = absent(means3D);
  if (!has_cov3D_precomp) cov3D_precomp = absent(means3D);
  auto result = RasterizeGaussiansprojectCUDA(rs.bg_, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier_,
                                              cov3D_precomp, rs.viewmatrix_, rs.projmatrix_, rs.tanfovx_, rs.tanfovy_,
                                              rs.image_height_, rs.image_width_, shs, rs.sh_degree_, rs.campos_, rs.prefiltered_);
  return std::make_tuple(std::get<0>(result), std::get<1>(result), std::get<2>(result));
}
