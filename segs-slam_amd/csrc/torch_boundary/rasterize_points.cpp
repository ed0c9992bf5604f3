// rasterize_points.cpp -- LibTorch-ROCm host layer of the drop-in boundary: the reference's tensor-typed entry points
// (src/rasterize_points.cu, src/operate_points.cu, src/stereo_vision.cu, third_party/simple-knn/spatial.cu) on top of the
// C ABI (include/segs_raster.h, include/segs_points.h).  LibTorch is plumbing here: tensor allocation + current stream.
#include "rasterize_points.h"

#include <c10/hip/HIPStream.h>

#include "../../../include/segs_points.h"
#include "../../../include/segs_raster.h"

namespace {
constexpr int NUM_CHANNELS = 3;  // cuda_rasterizer/config.h:15

void* cur_stream(const torch::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
void check(int status, const char* what) {
  if (status != SEGS_OK) AT_ERROR(what, " failed (", status, "): ", segs_last_error());
}
// "absent" = 0-element tensor => null pointer (src/rasterize_points.cu:95-105)
template <typename T = float>
T* ptr(const torch::Tensor& t) { return t.numel() == 0 ? nullptr : t.data_ptr<T>(); }
torch::Tensor f32c(const torch::Tensor& t) { return t.numel() == 0 ? t : t.contiguous().to(torch::kFloat32); }

// resizeFunctional (src/rasterize_points.cu:28-34) as a plain C callback
char* resize_cb(void* ctx, size_t n) {
  auto* t = static_cast<torch::Tensor*>(ctx);
  t->resize_({(long long)n});
  return reinterpret_cast<char*>(t->contiguous().data_ptr());
}
}  // namespace

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                       const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                       const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                       const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
                       const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                       const bool prefiltered) {
  if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
  const int P = means3D.size(0), H = image_height, W = image_width;
  auto float_opts = means3D.options().dtype(torch::kFloat32);
  // The reference fills both with zeros (:68-69), which only shows when P == 0 (:81 leaves the zero image): with Gaussians
  // the render kernel writes every pixel and the per-Gaussian kernel every radius, so the fills (37 MB at 3 M / 1080p) go
  torch::Tensor out_color = P != 0 ? torch::empty({NUM_CHANNELS, H, W}, float_opts) : torch::full({NUM_CHANNELS, H, W}, 0.0, float_opts);
  torch::Tensor radii = P != 0 ? torch::empty({P}, means3D.options().dtype(torch::kInt32)) : torch::full({P}, 0, means3D.options().dtype(torch::kInt32));
  auto byte_opts = torch::TensorOptions(torch::kByte).device(means3D.device());
  torch::Tensor geomBuffer = torch::empty({0}, byte_opts), binningBuffer = torch::empty({0}, byte_opts),
                imgBuffer = torch::empty({0}, byte_opts);
  int rendered = 0;
  if (P != 0) {
    int M = sh.size(0) != 0 ? sh.size(1) : 0;
    auto bg = f32c(background), m3 = f32c(means3D), shc = f32c(sh), col = f32c(colors), op = f32c(opacity), sc = f32c(scales),
         rot = f32c(rotations), cov = f32c(cov3D_precomp), view = f32c(viewmatrix), proj = f32c(projmatrix), cam = f32c(campos);
    check(segs_rasterize_forward(resize_cb, &geomBuffer, resize_cb, &binningBuffer, resize_cb, &imgBuffer, P, degree, M, ptr(bg), W,
                                 H, ptr(m3), ptr(shc), ptr(col), ptr(op), ptr(sc), scale_modifier, ptr(rot), ptr(cov), ptr(view),
                                 ptr(proj), ptr(cam), tan_fovx, tan_fovy, prefiltered, ptr(out_color), ptr<int>(radii),
                                 cur_stream(means3D), &rendered),
          "segs_rasterize_forward");
  }
  return std::make_tuple(rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                               const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                               const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree,
                               const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                               const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer) {
  const int P = means3D.size(0), H = dL_dout_color.size(1), W = dL_dout_color.size(2);
  int M = sh.size(0) != 0 ? sh.size(1) : 0;
  auto o = means3D.options().dtype(torch::kFloat32);
  // every row is written by the kernels: empty() instead of the reference's nine torch::zeros (:149-157)
  torch::Tensor dL_dmeans3D = torch::empty({P, 3}, o), dL_dmeans2D = torch::empty({P, 3}, o), dL_dcolors = torch::empty({P, NUM_CHANNELS}, o),
                dL_dopacity = torch::empty({P, 1}, o), dL_dcov3D = torch::empty({P, 6}, o),   // (dL_dconic :153 is internal: not materialised)
                dL_dsh = torch::zeros({P, M, 3}, o);
  const bool has_sr = scales.numel() != 0;
  torch::Tensor dL_dscales = has_sr ? torch::empty({P, 3}, o) : torch::zeros({P, 3}, o);
  torch::Tensor dL_drotations = has_sr ? torch::empty({P, 4}, o) : torch::zeros({P, 4}, o);
  if (P != 0) {
    auto bg = f32c(background), m3 = f32c(means3D), shc = f32c(sh), col = f32c(colors), sc = f32c(scales), rot = f32c(rotations),
         cov = f32c(cov3D_precomp), view = f32c(viewmatrix), proj = f32c(projmatrix), cam = f32c(campos), dL = f32c(dL_dout_color);
    auto rad = radii.contiguous();
    check(segs_rasterize_backward(P, degree, M, R, ptr(bg), W, H, ptr(m3), ptr(shc), ptr(col), ptr(sc), scale_modifier, ptr(rot),
                                  ptr(cov), ptr(view), ptr(proj), ptr(cam), tan_fovx, tan_fovy, ptr<int>(rad),
                                  reinterpret_cast<char*>(geomBuffer.contiguous().data_ptr()),
                                  reinterpret_cast<char*>(binningBuffer.contiguous().data_ptr()),
                                  reinterpret_cast<char*>(imageBuffer.contiguous().data_ptr()), ptr(dL), ptr(dL_dmeans2D),
                                  nullptr, ptr(dL_dopacity), ptr(dL_dcolors), ptr(dL_dmeans3D), ptr(dL_dcov3D), ptr(dL_dsh),
                                  has_sr ? ptr(dL_dscales) : nullptr, has_sr ? ptr(dL_drotations) : nullptr, cur_stream(means3D)),
          "segs_rasterize_backward");
  }
  return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix) {
  const int P = means3D.size(0);
  torch::Tensor present = torch::full({P}, false, means3D.options().dtype(at::kBool));
  if (P != 0) {
    auto m3 = f32c(means3D), view = f32c(viewmatrix), proj = f32c(projmatrix);
    check(segs_mark_visible(P, ptr(m3), ptr(view), ptr(proj), reinterpret_cast<uint8_t*>(present.data_ptr<bool>()), cur_stream(means3D)),
          "segs_mark_visible");
  }
  return present;
}

torch::Tensor RasterizeGaussiansfilterCUDA(const torch::Tensor& means3D, const torch::Tensor& scales, const torch::Tensor& rotations,
                                           const float scale_modifier, const torch::Tensor& cov3D_precomp,
                                           const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                                           const float tan_fovy, const int image_height, const int image_width,
                                           const bool prefiltered, const bool /*debug*/) {
  if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
  const int P = means3D.size(0);
  torch::Tensor radii = torch::full({P}, 0, means3D.options().dtype(torch::kInt32));
  if (P != 0) {
    auto m3 = f32c(means3D), sc = f32c(scales), rot = f32c(rotations), cov = f32c(cov3D_precomp), view = f32c(viewmatrix),
         proj = f32c(projmatrix);
    check(segs_visible_filter(P, 0, image_width, image_height, ptr(m3), ptr(sc), scale_modifier, ptr(rot), ptr(cov), ptr(view),
                              ptr(proj), tan_fovx, tan_fovy, prefiltered, ptr<int>(radii), cur_stream(means3D)),
          "segs_visible_filter");
  }
  return radii;
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansprojectCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                              const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                              const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                              const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
                              const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                              const bool prefiltered) {
  (void)background;
  if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
  const int P = means3D.size(0);
  auto o = means3D.options().dtype(torch::kFloat32);
  torch::Tensor out_color = torch::full({P, NUM_CHANNELS}, 0.0, o);
  torch::Tensor radii = torch::full({P}, 0, means3D.options().dtype(torch::kInt32));
  torch::Tensor points_image = torch::full({P, 2}, 0, o);
  if (P != 0) {
    int M = sh.size(0) != 0 ? sh.size(1) : 0;
    auto m3 = f32c(means3D), shc = f32c(sh), col = f32c(colors), op = f32c(opacity), sc = f32c(scales), rot = f32c(rotations),
         cov = f32c(cov3D_precomp), view = f32c(viewmatrix), proj = f32c(projmatrix), cam = f32c(campos);
    check(segs_project2_image(P, degree, M, image_width, image_height, ptr(m3), ptr(shc), ptr(col), ptr(op), ptr(sc), scale_modifier,
                              ptr(rot), ptr(cov), ptr(view), ptr(proj), ptr(cam), tan_fovx, tan_fovy, prefiltered, ptr(out_color),
                              ptr(points_image), ptr<int>(radii), cur_stream(means3D)),
          "segs_project2_image");
  }
  return std::make_tuple(points_image, radii, out_color);  // src/rasterize_points.cu:361
}

torch::Tensor distCUDA2(const torch::Tensor& points) {
  const int P = points.size(0);
  torch::Tensor means = torch::full({P}, 0.0, points.options().dtype(torch::kFloat32));
  if (P != 0) {
    auto pts = f32c(points);
    torch::Tensor temp = torch::empty({(long long)segs_knn_temp_bytes(P)}, torch::TensorOptions(torch::kByte).device(points.device()));
    check(segs_knn_mean_dist2(P, ptr(pts), ptr(means), reinterpret_cast<char*>(temp.data_ptr()), cur_stream(points)), "segs_knn_mean_dist2");
  }
  return means;
}

void transformPoints(torch::Tensor& points, torch::Tensor& transformmatrix) {
  if (points.ndimension() != 2 || points.size(1) != 3) AT_ERROR("points must have dimensions (num_points, 3)");
  const int P = points.size(0);
  torch::Tensor transformed_points = torch::zeros_like(points);
  if (P != 0) {
    auto pts = f32c(points), m = f32c(transformmatrix);
    check(segs_transform_points(P, ptr(pts), ptr(m), ptr(transformed_points), cur_stream(points)), "segs_transform_points");
    points = transformed_points;
  }
}

void scaleAndTransformThenMarkVisiblePoints(torch::Tensor& points, torch::Tensor& rots, torch::Tensor& point_not_transformed_mask,
                                            torch::Tensor& point_unstable_mask, torch::Tensor& transformmatrix,
                                            torch::Tensor& viewmatrix, torch::Tensor& projmatrix, int& num_transformed,
                                            const float scale) {
  if (points.ndimension() != 2 || points.size(1) != 3) AT_ERROR("points must have dimensions (num_points, 3)");
  torch::Tensor present = markVisible(points, viewmatrix, projmatrix);
  auto num_points = present.size(0);
  if (point_not_transformed_mask.size(0) != num_points || point_unstable_mask.size(0) != num_points)
    AT_ERROR("points_mask must have dimensions (num_points)");
  torch::Tensor final_mask = torch::logical_and(torch::logical_and(point_not_transformed_mask, point_unstable_mask), present);
  num_transformed += final_mask.sum().item<int>();
  const int P = points.size(0);
  if (P != 0) {
    torch::Tensor transformed_points = torch::zeros_like(points), transformed_rots = torch::zeros_like(rots);
    auto pts = f32c(points), rr = f32c(rots), m = f32c(transformmatrix);
    auto mask_u8 = final_mask.to(torch::kUInt8).contiguous();
    check(segs_scale_and_transform_points(P, scale, ptr(pts), ptr(rr), ptr(m), mask_u8.data_ptr<uint8_t>(), ptr(transformed_points),
                                          ptr(transformed_rots), cur_stream(points)),
          "segs_scale_and_transform_points");
    points.index_put_({final_mask}, transformed_points.index({final_mask}));
    rots.index_put_({final_mask}, transformed_rots.index({final_mask}));
    point_not_transformed_mask.index_put_({final_mask},
                                          torch::full({P}, false, point_not_transformed_mask.options()).index({final_mask}));
  }
}

torch::Tensor reprojectDepthPinhole(torch::Tensor& depth, torch::Tensor& mask, std::vector<float>& intr, int width) {
  if (depth.ndimension() != 1) AT_ERROR("points must have dimensions (num_points)");
  const int P = depth.size(0);
  torch::Tensor points;
  if (P != 0) {
    points = torch::zeros({P, 3}, depth.options());
    auto d = f32c(depth);
    auto k = mask.to(torch::kUInt8).contiguous();
    check(segs_reproject_depths_pinhole(P, width, intr[0], intr[1], intr[2], intr[3], ptr(d), k.data_ptr<uint8_t>(), ptr(points),
                                        cur_stream(depth)),
          "segs_reproject_depths_pinhole");
  }
  return points;
}

std::tuple<torch::Tensor, torch::Tensor> monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints(
    torch::Tensor& kps_pixel, torch::Tensor& kps_has3D, torch::Tensor& kps_point_local, torch::Tensor& colors,
    float max_pixel_dist, std::vector<float>& intr, int width) {
  if (kps_pixel.ndimension() != 2 || kps_pixel.size(1) != 2) AT_ERROR("kps_pixel must have dimensions (num_points, 2)");
  if (kps_has3D.ndimension() != 1) AT_ERROR("kps_has3D must have dimensions (num_points)");
  if (kps_point_local.ndimension() != 2 || kps_point_local.size(1) != 3) AT_ERROR("kps_point_local must have dimensions (num_points, 3)");
  int N = kps_pixel.size(0);
  torch::Tensor result_pt, result_color;
  if (N != 0) {
    result_pt = torch::zeros_like(kps_point_local);
    result_color = torch::zeros_like(kps_point_local);
    auto px = f32c(kps_pixel), p3 = f32c(kps_point_local), col = f32c(colors);
    auto h = kps_has3D.to(torch::kUInt8).contiguous();
    check(segs_search_neighborhood_depth(N, width, intr[0], intr[1], intr[2], intr[3], max_pixel_dist, ptr(px), h.data_ptr<uint8_t>(),
                                         ptr(p3), ptr(col), ptr(result_pt), ptr(result_color), cur_stream(kps_pixel)),
          "segs_search_neighborhood_depth");
    torch::Tensor valid = result_pt.index({torch::indexing::Slice(), 2}) > 0.0f;
    result_pt = result_pt.index({valid});
    result_color = result_color.index({valid});
  }
  return std::make_tuple(result_pt, result_color);
}
