// rasterize_points.h -- drop-in declarations of the reference's tensor-typed rasterizer entry points
// (same names, parameter order and return tuples as /root/reference include/rasterize_points.h:18-102), implemented
// on top of the C ABI include/segs_raster.h.  A SEGS-SLAM build links this library in place of its own
// libcuda_rasterizer.so; gaussian_rasterizer.cpp / gaussian_renderer.cpp / gaussian_mapper.cpp compile unchanged.
#pragma once
#include <torch/torch.h>
#include <tuple>

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                       const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                       const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                       const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
                       const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                       const bool prefiltered);

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                               const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                               const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree,
                               const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                               const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer);

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix);

torch::Tensor RasterizeGaussiansfilterCUDA(const torch::Tensor& means3D, const torch::Tensor& scales, const torch::Tensor& rotations,
                                           const float scale_modifier, const torch::Tensor& cov3D_precomp,
                                           const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                                           const float tan_fovy, const int image_height, const int image_width,
                                           const bool prefiltered, const bool debug);

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansprojectCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                              const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                              const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                              const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
                              const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                              const bool prefiltered);

// third_party/simple-knn/spatial.h:13
torch::Tensor distCUDA2(const torch::Tensor& points);

// include/operate_points.h:27-40
void transformPoints(torch::Tensor& points, torch::Tensor& transformmatrix);
void scaleAndTransformThenMarkVisiblePoints(torch::Tensor& points, torch::Tensor& rots, torch::Tensor& point_not_transformed_mask,
                                            torch::Tensor& point_unstable_mask, torch::Tensor& transformmatrix,
                                            torch::Tensor& viewmatrix, torch::Tensor& projmatrix, int& num_transformed,
                                            const float scale = 1.0f);

// include/stereo_vision.h:26-40
torch::Tensor reprojectDepthPinhole(torch::Tensor& depth, torch::Tensor& mask, std::vector<float>& intr, int width);
std::tuple<torch::Tensor, torch::Tensor> monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints(
    torch::Tensor& kps_pixel, torch::Tensor& kps_has3D, torch::Tensor& kps_point_local, torch::Tensor& colors,
    float max_pixel_dist, std::vector<float>& intr, int width);
