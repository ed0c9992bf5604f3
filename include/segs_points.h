/* segs_points.h -- C ABI of the point-set helpers named by the north star next to the rasterizer
 * (part of libsegs_raster.so): simple-knn, operate_points, stereo_vision.
 *
 * Reference interfaces (C++ ABI over torch::Tensor, no extern "C"):
 *   distCUDA2 / SimpleKNN::knn              third_party/simple-knn/spatial.h:13, simple_knn.h:14-18, simple_knn.cu:185-220
 *   transformPoints, scaleAndTransformThenMarkVisiblePoints   include/operate_points.h:27-40, src/operate_points.cu:73-143
 *   reprojectDepthPinhole, monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints
 *                                           include/stereo_vision.h:26-40, src/stereo_vision.cu:136-213
 * All pointers are device pointers; `bool` tensors are passed as bytes; status as in segs_raster.h.
 */
#ifndef SEGS_POINTS_H_
#define SEGS_POINTS_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* SimpleKNN::knn (simple_knn.cu:185-220): mean_dists[i] = mean of the 3 smallest squared distances from point i
 * to the other points.  `temp` holds segs_knn_temp_bytes(P) bytes.  No host synchronisation (the reference copies
 * the bounding box to the host twice and cudaMallocs per call). */
size_t segs_knn_temp_bytes(int P);
int segs_knn_mean_dist2(int P, const float* points /*P,3*/, float* mean_dists /*P*/, char* temp, void* stream);

/* transform_points (operate_points.cu:38-50): out = M * p with the transposed-layout 4x4 of auxiliary.h:59-67. */
int segs_transform_points(int P, const float* points, const float* transformmatrix, float* out_points, void* stream);

/* scale_and_transform_points (operate_points.cu:52-71): for mask[i] != 0, out_points[i] = M * (scale * p_i) and
 * out_rots[i] = quaternion of (M_rot * R(q_i)) (Shoemake).  The reference's insert_rot_to_rots writes element +2
 * twice and never +3 (cuda_rasterizer/operate_points.h:175-178); that behaviour is kept: out_rots[i] = (w, x, z, untouched). */
int segs_scale_and_transform_points(int P, float scale, const float* points, const float* rots, const float* transformmatrix,
                                    const uint8_t* mask, float* out_points, float* out_rots, void* stream);

/* reproject_depths_pinhole (stereo_vision.cu:39-61): pixel idx = v*width+u -> ((u-cx)*d/fx, (v-cy)*d/fy, d) where mask. */
int segs_reproject_depths_pinhole(int P, int width, float fx, float fy, float cx, float cy, const float* depths,
                                  const uint8_t* mask, float* points, void* stream);

/* search_neighborhood_to_estimate_depth_and_reproject_pinhole (stereo_vision.cu:63-134), quirks kept: the colour index
 * is v*width+u (not x3) and max_pixel_dist is compared with the SQUARED pixel distance. */
int segs_search_neighborhood_depth(int N, int width, float fx, float fy, float cx, float cy, float max_pixel_dist,
                                   const float* pixels /*N,2*/, const uint8_t* has3D, const float* point3D_orig /*N,3*/,
                                   const float* colors, float* point3D_result /*N,3*/, float* colors_result /*N,3*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGS_POINTS_H_ */
