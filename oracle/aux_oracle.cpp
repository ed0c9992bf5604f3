// aux_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code) for the point-set helpers:
// simple-knn, operate_points, stereo_vision.  Parity status: "parity unpinned" by the reference (no tests or
// fixtures exist for them upstream); restated line by line, -ffp-contract=off, see gs_oracle.cpp's header.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>

extern "C" {

// SimpleKNN::knn (third_party/simple-knn/simple_knn.cu:185-220): the Morton ordering and the box pruning only decide
// WHICH candidates are visited; the value is the exact mean of the three smallest squared distances to the other
// points (updateKBest, :131-145; final mean :182), so the oracle is the brute-force scan with the same arithmetic.
void gso_knn_mean_dist2(int P, const float* pts, float* out) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    const float rx = pts[3 * i], ry = pts[3 * i + 1], rz = pts[3 * i + 2];
    for (int j = 0; j < P; j++) {
      if (j == i) continue;
      const float dx = pts[3 * j] - rx, dy = pts[3 * j + 1] - ry, dz = pts[3 * j + 2] - rz;
      float dist = dx * dx + dy * dy + dz * dz;
      for (int k = 0; k < 3; k++)
        if (best[k] > dist) { float t = best[k]; best[k] = dist; dist = t; }
    }
    out[i] = (best[0] + best[1] + best[2]) / 3.0f;
  }
}

static inline void tp4x3(const float* p, const float* M, float* o) {  // cuda_rasterizer/auxiliary.h:59-67
  o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
  o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
  o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
}

// transform_points (src/operate_points.cu:38-50)
void gso_transform_points(int P, const float* pts, const float* M, float* out) {
  for (int i = 0; i < P; i++) tp4x3(pts + 3 * i, M, out + 3 * i);
}

// scale_and_transform_points (src/operate_points.cu:52-71) with transfrom_quaternion_using_matrix and the
// insert_rot_to_rots quirk (cuda_rasterizer/operate_points.h:69-178).  Outputs are only touched where mask != 0.
void gso_scale_and_transform_points(int P, float scale, const float* pts, const float* rots, const float* M,
                                    const uint8_t* mask, float* out_pts, float* out_rots) {
  for (int idx = 0; idx < P; idx++) {
    if (!mask[idx]) continue;
    float p[3] = { pts[3 * idx], pts[3 * idx + 1], pts[3 * idx + 2] };
    p[0] *= scale; p[1] *= scale; p[2] *= scale;
    tp4x3(p, M, out_pts + 3 * idx);
    const float qx = rots[4 * idx + 1], qy = rots[4 * idx + 2], qz = rots[4 * idx + 3], qw = rots[4 * idx];
    float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
    float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
    float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    float R00 = 1.0f - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
    float R10 = txy + twz, R11 = 1.0f - (txx + tzz), R12 = tyz - twx;
    float R20 = txz - twy, R21 = tyz + twx, R22 = 1.0f - (txx + tyy);
    float R[3][3];
    R[0][0] = M[0] * R00 + M[4] * R10 + M[8] * R20; R[0][1] = M[0] * R01 + M[4] * R11 + M[8] * R21; R[0][2] = M[0] * R02 + M[4] * R12 + M[8] * R22;
    R[1][0] = M[1] * R00 + M[5] * R10 + M[9] * R20; R[1][1] = M[1] * R01 + M[5] * R11 + M[9] * R21; R[1][2] = M[1] * R02 + M[5] * R12 + M[9] * R22;
    R[2][0] = M[2] * R00 + M[6] * R10 + M[10] * R20; R[2][1] = M[2] * R01 + M[6] * R11 + M[10] * R21; R[2][2] = M[2] * R02 + M[6] * R12 + M[10] * R22;
    float ox, oy, oz, ow;
    float t = R[0][0] + R[1][1] + R[2][2];
    if (t > 0.0f) {
      t = std::sqrt(t + 1.0f);
      ow = 0.5f * t;
      t = 0.5f / t;
      ox = (R[2][1] - R[1][2]) * t; oy = (R[0][2] - R[2][0]) * t; oz = (R[1][0] - R[0][1]) * t;
    } else {
      int i = 0;
      if (R[1][1] > R[0][0]) i = 1;
      if (R[2][2] > R[i][i]) i = 2;
      int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0f);
      float xyz[3];
      xyz[i] = 0.5f * t;
      t = 0.5f / t;
      ow = (R[k][j] - R[j][k]) * t;
      xyz[j] = (R[j][i] + R[i][j]) * t;
      xyz[k] = (R[k][i] + R[i][k]) * t;
      ox = xyz[0]; oy = xyz[1]; oz = xyz[2];
    }
    out_rots[4 * idx] = ow; out_rots[4 * idx + 1] = ox; out_rots[4 * idx + 2] = oy; out_rots[4 * idx + 2] = oz;  // sic
  }
}

static inline void reproject(int u, int v, float depth, float fx, float fy, float cx, float cy, float* o) {
  o[0] = (u - cx) * depth / fx; o[1] = (v - cy) * depth / fy; o[2] = depth;  // cuda_rasterizer/stereo_vision.h:39-54
}

// reproject_depths_pinhole (src/stereo_vision.cu:39-61)
void gso_reproject_depths_pinhole(int P, int width, float fx, float fy, float cx, float cy, const float* depths,
                                  const uint8_t* mask, float* points) {
  for (int idx = 0; idx < P; idx++) {
    if (!mask[idx]) continue;
    int v = idx / width, u = idx - v * width;
    reproject(u, v, depths[idx], fx, fy, cx, cy, points + 3 * idx);
  }
}

// search_neighborhood_to_estimate_depth_and_reproject_pinhole (src/stereo_vision.cu:63-134)
void gso_search_neighborhood_depth(int N, int width, float fx, float fy, float cx, float cy, float max_pixel_dist,
                                   const float* pixels, const uint8_t* has3D, const float* p3d, const float* colors,
                                   float* out_p, float* out_c) {
  for (int idx = 0; idx < N; idx++) {
    float u = pixels[2 * idx], v = pixels[2 * idx + 1];
    int ptidx = idx * 3;
    int pxidx_in_image = v * width + u;
    if (has3D[idx]) {
      out_p[ptidx] = p3d[ptidx]; out_p[ptidx + 1] = p3d[ptidx + 1]; out_p[ptidx + 2] = p3d[ptidx + 2];
      out_c[ptidx] = colors[pxidx_in_image]; out_c[ptidx + 1] = colors[pxidx_in_image + 1]; out_c[ptidx + 2] = colors[pxidx_in_image + 2];
      continue;
    }
    float min_dist = 3.402823466e+38f, depth = -1.0f;
    for (int i = 0; i < N; ++i) {
      if (!has3D[i] || i == idx) continue;
      float u_uu = u - pixels[2 * i], v_vv = v - pixels[2 * i + 1];
      float dist = u_uu * u_uu + v_vv * v_vv;
      if (dist > max_pixel_dist || dist >= min_dist) continue;
      min_dist = dist;
      depth = p3d[i * 3 + 2];
    }
    if (depth > 0.0f) {
      reproject((int)u, (int)v, depth, fx, fy, cx, cy, out_p + ptidx);
      out_c[ptidx] = colors[pxidx_in_image]; out_c[ptidx + 1] = colors[pxidx_in_image + 1]; out_c[ptidx + 2] = colors[pxidx_in_image + 2];
    } else {
      out_p[ptidx + 2] = -1.0f;
    }
  }
}

}  // extern "C"
