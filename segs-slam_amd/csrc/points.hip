// points.hip -- simple-knn, operate_points, stereo_vision kernels (include/segs_points.h) for gfx950.
//   K16/K17 reference: third_party/simple-knn/simple_knn.cu:45-220 (what it computes: the mean of the three smallest squared
//           distances from every point to the others -- an exact 3-NN query, so any exact search gives the same numbers)
//   K18     reference: src/operate_points.cu:38-71, cuda_rasterizer/operate_points.h:39-178
//   K19     reference: src/stereo_vision.cu:39-134, cuda_rasterizer/stereo_vision.h:39-54
// simple-knn here is a workgroup-cooperative search over LDS tiles (see knn_query_kernel); the rest are cold elementwise
// kernels.  Built with -ffp-contract=off: distances are formed as dx*dx + dy*dy + dz*dz like the reference's, so the
// results are bit-identical to the CPU oracle.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include "../../include/segs_raster.h"
#include "../../include/segs_points.h"
#include "gs_layout.h"

#pragma clang fp contract(off)

namespace {

// ---------------------------------------------------------------- simple-knn
// bounding box with the reference's init {0,0,0} for both reductions (simple_knn.cu:191-199): the box always
// contains the origin.  Two stages, no host copy.
__global__ void __launch_bounds__(256) bbox_partial_kernel(int P, const float* __restrict__ pts, float* __restrict__ partial) {
  __shared__ float red[6][256];
  float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
    for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
  for (int k = 0; k < 3; k++) { red[k][threadIdx.x] = mn[k]; red[3 + k][threadIdx.x] = mx[k]; }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int k = 0; k < 3; k++) {
        red[k][threadIdx.x] = fminf(red[k][threadIdx.x], red[k][threadIdx.x + off]);
        red[3 + k][threadIdx.x] = fmaxf(red[3 + k][threadIdx.x], red[3 + k][threadIdx.x + off]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ void bbox_final_kernel(int nb, const float* __restrict__ partial, float* __restrict__ bbox) {
  if (threadIdx.x < 6) {
    float v = 0.f;
    for (int b = 0; b < nb; b++) v = threadIdx.x < 3 ? fminf(v, partial[b * 6 + threadIdx.x]) : fmaxf(v, partial[b * 6 + threadIdx.x]);
    bbox[threadIdx.x] = v;
  }
}
__device__ __forceinline__ uint32_t prepMorton(uint32_t x) {  // simple_knn.cu:45-52
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}
__global__ void __launch_bounds__(256) morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ bbox,
                                                     uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float px = pts[3 * (size_t)idx], py = pts[3 * (size_t)idx + 1], pz = pts[3 * (size_t)idx + 2];
  // simple_knn.cu:54-61 (float -> uint32 conversion truncates)
  const uint32_t x = prepMorton((uint32_t)(((px - bbox[0]) / (bbox[3] - bbox[0])) * ((1 << 10) - 1)));
  const uint32_t y = prepMorton((uint32_t)(((py - bbox[1]) / (bbox[4] - bbox[1])) * ((1 << 10) - 1)));
  const uint32_t z = prepMorton((uint32_t)(((pz - bbox[2]) / (bbox[5] - bbox[2])) * ((1 << 10) - 1)));
  keys[idx] = (uint64_t)(x | (y << 1) | (z << 2));
  vals[idx] = (uint32_t)idx;
}
// ---- exact 3-nearest-neighbour search, MI355X-shaped.
// The reference gives every point its own serial walk over all 1024-point boxes and, for each box it cannot reject, over the
// box's points in global memory (simple_knn.cu:147-183).  Here the points are cut into boxes of 256 consecutive points of the
// Morton order -- one box = one workgroup = one LDS tile -- and a WORKGROUP searches for its 256 query points together:
//   (A) every lane takes its three best among the workgroup's own tile; R = the largest third-best of the workgroup;
//   (B) the lanes test 256 candidate boxes at a time (AABB-to-AABB gap against R, one box per lane), the hits are compacted
//       with ballots, and each hit box is pulled into LDS ONCE with coalesced 16-byte loads and scanned by all lanes as
//       broadcast reads, each lane first checking the box against its own third-best; R is re-tightened as it goes.
// Spatially coherent queries share almost all their candidate boxes, so a box's points are fetched once per workgroup, not
// once per point.  The pruning bounds are deflated by 1e-6 so that rounding can only make a lane look at a box too many.
constexpr int KNN_BOX = 256;
constexpr float KNN_FAR = 3.0e38f;

// gather the points into Morton order (float4: xyz + original index bits) and take each 256-point box's bounds
__global__ void __launch_bounds__(KNN_BOX) knn_gather_boxes_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                                   float4* __restrict__ sorted, float* __restrict__ boxes) {
  __shared__ float red[4][6];
  const int idx = blockIdx.x * KNN_BOX + threadIdx.x;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (idx < P) {
    const uint32_t o = order[idx];
    const float x = pts[3 * (size_t)o], y = pts[3 * (size_t)o + 1], z = pts[3 * (size_t)o + 2];
    sorted[idx] = make_float4(x, y, z, __uint_as_float(o));
    lo[0] = hi[0] = x; lo[1] = hi[1] = y; lo[2] = hi[2] = z;
  }
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
    }
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 3; k++) { red[threadIdx.x >> 6][k] = lo[k]; red[threadIdx.x >> 6][3 + k] = hi[k]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    const float a = red[0][k], b = red[1][k], c = red[2][k], d = red[3][k];
    boxes[blockIdx.x * 6 + k] = k < 3 ? fminf(fminf(a, b), fminf(c, d)) : fmaxf(fmaxf(a, b), fmaxf(c, d));
  }
}

__device__ __forceinline__ void keep_best3(float dist, float* best) {   // best[0] <= best[1] <= best[2]
#pragma unroll
  for (int j = 0; j < 3; j++)
    if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
}
__device__ __forceinline__ float gap2_point_box(const float* b, float x, float y, float z) {   // squared distance point -> AABB
  const float dx = fmaxf(0.f, fmaxf(b[0] - x, x - b[3])), dy = fmaxf(0.f, fmaxf(b[1] - y, y - b[4])), dz = fmaxf(0.f, fmaxf(b[2] - z, z - b[5]));
  return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ float block_max(float v, float* s_red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

__global__ void __launch_bounds__(KNN_BOX) knn_query_kernel(int P, const float4* __restrict__ sorted, const float* __restrict__ boxes,
                                                            int nboxes, float* __restrict__ dists) {
  __shared__ float4 tile[KNN_BOX];
  __shared__ float s_red[4];
  __shared__ int s_hits[KNN_BOX];
  __shared__ int s_wave_n[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int qb = blockIdx.x;
  const int q = qb * KNN_BOX + tid;
  const bool valid = q < P;
  const float4 me = valid ? sorted[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};

  auto scan_tile = [&](int count, int self) {     // all lanes read the same LDS word: broadcast
    for (int j = 0; j < count; j++) {
      const float4 p = tile[j];
      const float dx = p.x - me.x, dy = p.y - me.y, dz = p.z - me.z;
      const float d = dx * dx + dy * dy + dz * dz;
      if (j != self) keep_best3(d, best);
    }
  };
  // (A) the workgroup's own box
  tile[tid] = me;
  __syncthreads();
  const int own = min(KNN_BOX, P - qb * KNN_BOX);
  if (valid) scan_tile(own, tid);
  float R = block_max(valid ? best[2] : 0.f, s_red);
  float qbox[6];
#pragma unroll
  for (int k = 0; k < 6; k++) qbox[k] = boxes[qb * 6 + k];

  // (B) every other box that can hold a point closer than R to some point of this workgroup's box
  for (int base = 0; base < nboxes; base += KNN_BOX) {
    const int b = base + tid;
    bool hit = false;
    if (b < nboxes && b != qb) {
      const float* bb = boxes + 6 * (size_t)b;
      const float gx = fmaxf(0.f, fmaxf(bb[0] - qbox[3], qbox[0] - bb[3])), gy = fmaxf(0.f, fmaxf(bb[1] - qbox[4], qbox[1] - bb[4])),
                  gz = fmaxf(0.f, fmaxf(bb[2] - qbox[5], qbox[2] - bb[5]));
      hit = (gx * gx + gy * gy + gz * gz) * 0.999999f <= R;
    }
    const uint64_t m = __ballot(hit);
    if (lane == 0) s_wave_n[wv] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
    for (int w = 0; w < 4; w++) { if (w < wv) off += s_wave_n[w]; total += s_wave_n[w]; }
    if (hit) s_hits[off + __popcll(m & ((1ull << lane) - 1ull))] = b;
    __syncthreads();
    for (int h = 0; h < total; h++) {
      const int hb = s_hits[h];
      const int src = hb * KNN_BOX + tid;
      const float4 mine = src < P ? sorted[src] : make_float4(KNN_FAR, KNN_FAR, KNN_FAR, 0.f);
      __syncthreads();          // everyone is done with the previous tile
      tile[tid] = mine;
      __syncthreads();
      if (valid && gap2_point_box(boxes + 6 * (size_t)hb, me.x, me.y, me.z) * 0.999999f <= best[2])
        scan_tile(min(KNN_BOX, P - hb * KNN_BOX), -1);
    }
    if (total > 0) R = block_max(valid ? best[2] : 0.f, s_red);   // the bound only ever tightens
    __syncthreads();
  }
  if (valid) dists[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

// ---------------------------------------------------------------- operate_points
__device__ __forceinline__ float3 transformPoint4x3(float3 p, const float* M) {  // auxiliary.h:59-67
  return make_float3(M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12],
                     M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
                     M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14]);
}
__global__ void __launch_bounds__(256) transform_points_kernel(int P, const float* __restrict__ pts, const float* __restrict__ M,
                                                               float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float3 p = make_float3(pts[3 * (size_t)idx], pts[3 * (size_t)idx + 1], pts[3 * (size_t)idx + 2]);
  const float3 t = transformPoint4x3(p, M);
  out[3 * (size_t)idx] = t.x; out[3 * (size_t)idx + 1] = t.y; out[3 * (size_t)idx + 2] = t.z;
}
__global__ void __launch_bounds__(256) scale_and_transform_points_kernel(
    int P, float scale, const float* __restrict__ pts, const float* __restrict__ rots, const float* __restrict__ M,
    const uint8_t* __restrict__ mask, float* __restrict__ out_pts, float* __restrict__ out_rots) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P || !mask[idx]) return;
  float3 p = make_float3(pts[3 * (size_t)idx], pts[3 * (size_t)idx + 1], pts[3 * (size_t)idx + 2]);
  p.x *= scale; p.y *= scale; p.z *= scale;  // operate_points.h:61-63
  const float3 t = transformPoint4x3(p, M);
  out_pts[3 * (size_t)idx] = t.x; out_pts[3 * (size_t)idx + 1] = t.y; out_pts[3 * (size_t)idx + 2] = t.z;
  // transfrom_quaternion_using_matrix (operate_points.h:69-160); input stored (w, x, y, z)
  const float qx = rots[4 * (size_t)idx + 1], qy = rots[4 * (size_t)idx + 2], qz = rots[4 * (size_t)idx + 3], qw = rots[4 * (size_t)idx];
  const float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
  const float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const float R00 = 1.0f - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
  const float R10 = txy + twz, R11 = 1.0f - (txx + tzz), R12 = tyz - twx;
  const float R20 = txz - twy, R21 = tyz + twx, R22 = 1.0f - (txx + tyy);
  float R[3][3];
  R[0][0] = M[0] * R00 + M[4] * R10 + M[8] * R20; R[0][1] = M[0] * R01 + M[4] * R11 + M[8] * R21; R[0][2] = M[0] * R02 + M[4] * R12 + M[8] * R22;
  R[1][0] = M[1] * R00 + M[5] * R10 + M[9] * R20; R[1][1] = M[1] * R01 + M[5] * R11 + M[9] * R21; R[1][2] = M[1] * R02 + M[5] * R12 + M[9] * R22;
  R[2][0] = M[2] * R00 + M[6] * R10 + M[10] * R20; R[2][1] = M[2] * R01 + M[6] * R11 + M[10] * R21; R[2][2] = M[2] * R02 + M[6] * R12 + M[10] * R22;
  float ox, oy, oz, ow;
  float t0 = R[0][0] + R[1][1] + R[2][2];
  if (t0 > 0.0f) {
    t0 = sqrtf(t0 + 1.0f);
    ow = 0.5f * t0;
    t0 = 0.5f / t0;
    ox = (R[2][1] - R[1][2]) * t0; oy = (R[0][2] - R[2][0]) * t0; oz = (R[1][0] - R[0][1]) * t0;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t0 = sqrtf(R[i][i] - R[j][j] - R[k][k] + 1.0f);
    float xyz[3];
    xyz[i] = 0.5f * t0;
    t0 = 0.5f / t0;
    ow = (R[k][j] - R[j][k]) * t0;
    xyz[j] = (R[j][i] + R[i][j]) * t0;
    xyz[k] = (R[k][i] + R[i][k]) * t0;
    ox = xyz[0]; oy = xyz[1]; oz = xyz[2];
  }
  // insert_rot_to_rots (operate_points.h:168-178): +2 is written twice (y then z), +3 never.
  out_rots[4 * (size_t)idx] = ow; out_rots[4 * (size_t)idx + 1] = ox; out_rots[4 * (size_t)idx + 2] = oz;
  (void)oy;
}

// ---------------------------------------------------------------- stereo_vision
__device__ __forceinline__ float3 reproject_depth_pinhole(int u, int v, float depth, float fx, float fy, float cx, float cy) {
  return make_float3((u - cx) * depth / fx, (v - cy) * depth / fy, depth);  // stereo_vision.h:39-54
}
__global__ void __launch_bounds__(256) reproject_depths_pinhole_kernel(int P, int width, float fx, float fy, float cx, float cy,
                                                                       const float* __restrict__ depths, const uint8_t* __restrict__ mask,
                                                                       float* __restrict__ points) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P || !mask[idx]) return;
  const int v = idx / width, u = idx - v * width;
  const float3 p = reproject_depth_pinhole(u, v, depths[idx], fx, fy, cx, cy);
  points[3 * (size_t)idx] = p.x; points[3 * (size_t)idx + 1] = p.y; points[3 * (size_t)idx + 2] = p.z;
}
__global__ void __launch_bounds__(256) search_neighborhood_kernel(int N, int width, float fx, float fy, float cx, float cy,
                                                                  float max_pixel_dist, const float* __restrict__ pixels,
                                                                  const uint8_t* __restrict__ has3D, const float* __restrict__ p3d,
                                                                  const float* __restrict__ colors, float* __restrict__ out_p,
                                                                  float* __restrict__ out_c) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N) return;
  const float u = pixels[2 * (size_t)idx], v = pixels[2 * (size_t)idx + 1];
  const size_t pt = 3 * (size_t)idx;
  const int pxidx_in_image = (int)(v * width + u);  // stereo_vision.cu:86 (not scaled by 3: reference quirk)
  if (has3D[idx]) {
    out_p[pt] = p3d[pt]; out_p[pt + 1] = p3d[pt + 1]; out_p[pt + 2] = p3d[pt + 2];
    out_c[pt] = colors[pxidx_in_image]; out_c[pt + 1] = colors[pxidx_in_image + 1]; out_c[pt + 2] = colors[pxidx_in_image + 2];
    return;
  }
  float min_dist = 3.402823466e+38f, depth = -1.0f;  // MAXFLOAT
  for (int i = 0; i < N; ++i) {
    if (!has3D[i] || i == idx) continue;
    const float du = u - pixels[2 * (size_t)i], dv = v - pixels[2 * (size_t)i + 1];
    const float dist = du * du + dv * dv;
    if (dist > max_pixel_dist || dist >= min_dist) continue;  // squared distance vs max_pixel_dist: reference quirk (:113)
    min_dist = dist;
    depth = p3d[3 * (size_t)i + 2];
  }
  if (depth > 0.0f) {
    const float3 r = reproject_depth_pinhole((int)u, (int)v, depth, fx, fy, cx, cy);
    out_p[pt] = r.x; out_p[pt + 1] = r.y; out_p[pt + 2] = r.z;
    out_c[pt] = colors[pxidx_in_image]; out_c[pt + 1] = colors[pxidx_in_image + 1]; out_c[pt + 2] = colors[pxidx_in_image + 2];
  } else {
    out_p[pt + 2] = -1.0f;
  }
}

struct KnnLayout { size_t bbox, partial, keys, vals, keys_o, vals_o, sorted, boxes, sort_temp, total; };
KnnLayout knn_layout(int P) {
  using segs::align_up;
  KnnLayout l{};
  const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
  size_t o = 0;
  l.bbox = o;      o = align_up(o + 64);
  l.partial = o;   o = align_up(o + 256 * 6 * 4);
  l.keys = o;      o = align_up(o + (size_t)P * 8);
  l.vals = o;      o = align_up(o + (size_t)P * 4);
  l.keys_o = o;    o = align_up(o + (size_t)P * 8);
  l.vals_o = o;    o = align_up(o + (size_t)P * 4);
  l.sorted = o;    o = align_up(o + (size_t)P * 16);
  l.boxes = o;     o = align_up(o + (size_t)(nboxes + 1) * 6 * 4);
  l.sort_temp = o; o = align_up(o + segs_binning_bytes(P));
  l.total = o + segs::ALIGN;
  return l;
}
#define LAUNCH_OK() do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return (int)_e; } while (0)
}  // namespace

extern "C" {

size_t segs_knn_temp_bytes(int P) { return knn_layout(P < 0 ? 0 : P).total; }

int segs_knn_mean_dist2(int P, const float* points, float* mean_dists, char* temp, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (P < 0) return SEGS_ERR_INVALID_ARGUMENT;
  if (P == 0) return SEGS_OK;
  if (!points || !mean_dists || !temp) return SEGS_ERR_INVALID_ARGUMENT;
  const KnnLayout L = knn_layout(P);
  char* base = segs::align_ptr(temp);
  float* bbox = (float*)(base + L.bbox);
  float* partial = (float*)(base + L.partial);
  const int nb = min(256, (P + 255) / 256);
  bbox_partial_kernel<<<nb, 256, 0, st>>>(P, points, partial); LAUNCH_OK();
  bbox_final_kernel<<<1, 64, 0, st>>>(nb, partial, bbox); LAUNCH_OK();
  morton_kernel<<<(P + 255) / 256, 256, 0, st>>>(P, points, bbox, (uint64_t*)(base + L.keys), (uint32_t*)(base + L.vals)); LAUNCH_OK();
  int rc = segs_sort_pairs((const uint64_t*)(base + L.keys), (const uint32_t*)(base + L.vals), (uint64_t*)(base + L.keys_o),
                           (uint32_t*)(base + L.vals_o), P, 30, base + L.sort_temp, stream);
  if (rc) return rc;
  const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
  knn_gather_boxes_kernel<<<nboxes, KNN_BOX, 0, st>>>(P, points, (const uint32_t*)(base + L.vals_o), (float4*)(base + L.sorted),
                                                      (float*)(base + L.boxes)); LAUNCH_OK();
  knn_query_kernel<<<nboxes, KNN_BOX, 0, st>>>(P, (const float4*)(base + L.sorted), (const float*)(base + L.boxes), nboxes, mean_dists); LAUNCH_OK();
  return SEGS_OK;
}

int segs_transform_points(int P, const float* points, const float* transformmatrix, float* out_points, void* stream) {
  if (P < 0) return SEGS_ERR_INVALID_ARGUMENT;
  if (P == 0) return SEGS_OK;
  if (!points || !transformmatrix || !out_points) return SEGS_ERR_INVALID_ARGUMENT;
  transform_points_kernel<<<(P + 255) / 256, 256, 0, (hipStream_t)stream>>>(P, points, transformmatrix, out_points); LAUNCH_OK();
  return SEGS_OK;
}

int segs_scale_and_transform_points(int P, float scale, const float* points, const float* rots, const float* transformmatrix,
                                    const uint8_t* mask, float* out_points, float* out_rots, void* stream) {
  if (P < 0) return SEGS_ERR_INVALID_ARGUMENT;
  if (P == 0) return SEGS_OK;
  if (!points || !rots || !transformmatrix || !mask || !out_points || !out_rots) return SEGS_ERR_INVALID_ARGUMENT;
  scale_and_transform_points_kernel<<<(P + 255) / 256, 256, 0, (hipStream_t)stream>>>(P, scale, points, rots, transformmatrix, mask,
                                                                                     out_points, out_rots); LAUNCH_OK();
  return SEGS_OK;
}

int segs_reproject_depths_pinhole(int P, int width, float fx, float fy, float cx, float cy, const float* depths,
                                  const uint8_t* mask, float* points, void* stream) {
  if (P < 0 || width <= 0) return SEGS_ERR_INVALID_ARGUMENT;
  if (P == 0) return SEGS_OK;
  if (!depths || !mask || !points) return SEGS_ERR_INVALID_ARGUMENT;
  reproject_depths_pinhole_kernel<<<(P + 255) / 256, 256, 0, (hipStream_t)stream>>>(P, width, fx, fy, cx, cy, depths, mask, points); LAUNCH_OK();
  return SEGS_OK;
}

int segs_search_neighborhood_depth(int N, int width, float fx, float fy, float cx, float cy, float max_pixel_dist,
                                   const float* pixels, const uint8_t* has3D, const float* point3D_orig, const float* colors,
                                   float* point3D_result, float* colors_result, void* stream) {
  if (N < 0 || width <= 0) return SEGS_ERR_INVALID_ARGUMENT;
  if (N == 0) return SEGS_OK;
  if (!pixels || !has3D || !point3D_orig || !colors || !point3D_result || !colors_result) return SEGS_ERR_INVALID_ARGUMENT;
  search_neighborhood_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, width, fx, fy, cx, cy, max_pixel_dist, pixels, has3D,
                                                                              point3D_orig, colors, point3D_result, colors_result); LAUNCH_OK();
  return SEGS_OK;
}

}  // extern "C"
