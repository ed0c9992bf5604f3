// project_gaussian.h -- the per-Gaussian forward geometry of K1 / K2 (cuda_rasterizer/forward.cu:74-152,185-236; auxiliary.h:41-78)
// and the 64-byte record the tile kernels read, as device functions shared by preprocess.hip (preprocess_fwd_kernel,
// visible_filter_kernel, preprocess_bwd_kernel's recompute) and neural.hip (the neural forward's epilogue projects the Gaussians
// it generates: SURVEY 8f n3, round 4).
//
// Arithmetic: every expression in the reference's order with one binary32 rounding per operation -- NO FMA contraction -- so
// that radii, rects, depths (hence tile / sort keys) are bit-identical to the CPU oracle whichever translation unit the code is
// inlined into: the pragma below covers this header and the including file restores its own setting after the #include
// (preprocess.hip: off again, it is built with -ffp-contract=off; neural.hip: fast, hipcc's default).
// glm semantics: mat3 is column-major, m[c][r] = column c, row r.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gs_layout.h"

#pragma clang fp contract(off)

namespace segs {

struct mat3 { float m[3][3]; };

__device__ __forceinline__ mat3 mk(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
  mat3 r; r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c; r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f; r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i; return r;
}
__device__ __forceinline__ mat3 mul(const mat3& A, const mat3& B) {
  mat3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++)
      R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
  return R;
}
__device__ __forceinline__ mat3 transpose(const mat3& A) {
  mat3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
  return R;
}
__device__ __forceinline__ mat3 smul(float s, const mat3& A) {
  mat3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.m[c][r] = s * A.m[c][r];
  return R;
}

// auxiliary.h:41-45 -- double-literal arithmetic, narrowed on return.
__device__ __forceinline__ float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:47-57 -- (int) truncation first, then clamp to the tile grid.
__device__ __forceinline__ void getRect(float px, float py, int max_radius, uint32_t& minx, uint32_t& miny,
                                        uint32_t& maxx, uint32_t& maxy, uint32_t gx, uint32_t gy) {
  minx = min(gx, (uint32_t)max(0, (int)((px - max_radius) / TILE_X)));
  miny = min(gy, (uint32_t)max(0, (int)((py - max_radius) / TILE_Y)));
  maxx = min(gx, (uint32_t)max(0, (int)((px + max_radius + TILE_X - 1) / TILE_X)));
  maxy = min(gy, (uint32_t)max(0, (int)((py + max_radius + TILE_Y - 1) / TILE_Y)));
}

// auxiliary.h:59-78
__device__ __forceinline__ float3 transformPoint4x3(float3 p, const float* M) {
  return make_float3(M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12],
                     M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
                     M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14]);
}
__device__ __forceinline__ float4 transformPoint4x4(float3 p, const float* M) {
  return make_float4(M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12],
                     M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
                     M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14],
                     M[3] * p.x + M[7] * p.y + M[11] * p.z + M[15]);
}

__device__ __forceinline__ mat3 quat_to_R(float4 rot) {  // forward.cu:127-139 (un-normalised, F5b)
  const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
  return mk(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
            2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
            2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

// forward.cu:118-152
__device__ __forceinline__ void computeCov3D(float3 scale, float mod, float4 rot, float* cov3D) {
  mat3 S = mk(1, 0, 0, 0, 1, 0, 0, 0, 1);
  S.m[0][0] = mod * scale.x; S.m[1][1] = mod * scale.y; S.m[2][2] = mod * scale.z;
  mat3 R = quat_to_R(rot);
  mat3 M = mul(S, R);
  mat3 Sigma = mul(transpose(M), M);
  cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
  cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

struct Cov2DTerms { mat3 T, W, Vrk; float3 t; float txtz, tytz; };

// forward.cu:74-113 (also the recompute at backward.cu:160-199); returns (a, b, c) with the 0.3 dilation.
__device__ __forceinline__ float3 computeCov2D(float3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                                               const float* cov3D, const float* view, Cov2DTerms* out) {
  float3 t = transformPoint4x3(mean, view);
  const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
  const float txtz = t.x / t.z, tytz = t.y / t.z;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  mat3 J = mk(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z),
              0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z),
              0, 0, 0);
  mat3 W = mk(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
  mat3 T = mul(W, J);
  mat3 Vrk = mk(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
  mat3 cov = mul(mul(transpose(T), transpose(Vrk)), T);
  cov.m[0][0] += 0.3f; cov.m[1][1] += 0.3f;
  if (out) { out->T = T; out->W = W; out->Vrk = Vrk; out->t = t; out->txtz = txtz; out->tytz = tytz; }
  return make_float3(cov.m[0][0], cov.m[0][1], cov.m[1][1]);
}

struct Projected {
  int radius;          // 0 = rejected
  float depth, px, py; // view z, pixel centre
  float3 conic;
  float cov_a, cov_c;  // diagonal of the dilated 2D covariance
  uint32_t minx, miny, maxx, maxy;
};

// Shared geometry of K1/K2 (forward.cu:185-236 and :282-330).
__device__ __forceinline__ Projected project_gaussian(float3 p, float3 scale, float mod, float4 rot, const float* cov3D_precomp_row,
                                                     const float* view, const float* proj, int W, int H, float tan_fovx,
                                                     float tan_fovy, float focal_x, float focal_y, uint32_t gx, uint32_t gy) {
  Projected o; o.radius = 0;
  float3 p_view = transformPoint4x3(p, view);
  if (p_view.z <= 0.2f) return o;  // auxiliary.h:155-156 (x/y frustum test removed in this fork)
  float4 p_hom = transformPoint4x4(p, proj);
  float p_w = 1.0f / (p_hom.w + 0.0000001f);
  float3 p_proj = make_float3(p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w);
  float cov3D[6];
  if (cov3D_precomp_row) {
#pragma unroll
    for (int k = 0; k < 6; k++) cov3D[k] = cov3D_precomp_row[k];
  } else {
    computeCov3D(scale, mod, rot, cov3D);
  }
  float3 cov = computeCov2D(p, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view, nullptr);
  float det = (cov.x * cov.z - cov.y * cov.y);
  if (det == 0.0f) return o;
  float det_inv = 1.f / det;
  o.conic = make_float3(cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv);
  o.cov_a = cov.x; o.cov_c = cov.z;
  float mid = 0.5f * (cov.x + cov.z);
  float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
  float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
  o.px = ndc2Pix(p_proj.x, W); o.py = ndc2Pix(p_proj.y, H);
  getRect(o.px, o.py, (int)my_radius, o.minx, o.miny, o.maxx, o.maxy, gx, gy);
  if ((o.maxx - o.minx) * (o.maxy - o.miny) == 0) return o;
  o.depth = p_view.z;
  o.radius = (int)my_radius;
  return o;
}



// What K1 leaves per binned Gaussian besides its radius: the tile rectangle it is binned over, the 64-byte record, the depth bits.
struct BinnedRecord {
  uint32_t touched, depth_bits, rect_min, rect_max;
  float4 q0, q1, q2, q3;     // x y A2 B2 | C2 o r g | b depth rect_min rect_max | A B C -   (gs_layout.h)
};
constexpr uint32_t PROJECT_TIGHT_RECT = 0x80000000u;   // = PREPROCESS_TIGHT_RECT (kernels.h)

// The conic as the tile kernels take it: alpha = o * exp2(A2 dx^2 + B2 dx dy + C2 dy^2)
__device__ __forceinline__ float4 scaled_conic(float A, float B, float C) {
  return make_float4((-0.5f * LOG2E) * A, (-LOG2E) * B, (-0.5f * LOG2E) * C, 0.f);
}

// Second half of K1 for a Gaussian with g.radius > 0: optionally shrink the rectangle to the alpha >= 1/255 ellipse's bounding box
// (resident / tight mode), count the tiles, assemble the record.  `col` is the Gaussian's colour (precomputed or from SH).
__device__ __forceinline__ BinnedRecord make_record(Projected& g, float opacity, float3 col, uint32_t flags) {
  BinnedRecord b;
  if (flags & PROJECT_TIGHT_RECT) {
    // Bin only the tiles the alpha >= 1/255 ellipse can reach.  d^T Q d <= k = 2 ln(255 o) has the axis-aligned half extents
    // sqrt(k cov_xx), sqrt(k cov_yy) (cov = Q^-1, the dilated 2D covariance); the rect is the reference's (3 sigma square,
    // getRect) INTERSECTED with that box, never larger, so exactly the reference's contributing pairs remain.  k is inflated
    // like the emitter's (binning.hip), the extents once more.
    if (opacity * 255.0f > 1.0f) {
      const float k = 2.0f * __logf(255.0f * opacity) * 1.0001f + 1e-3f;
      const float hx = sqrtf(k * g.cov_a) * 1.0001f + 0.01f, hy = sqrtf(k * g.cov_c) * 1.0001f + 0.01f;
      const int tx0 = (int)floorf((g.px - hx) * (1.0f / TILE_X)), tx1 = (int)floorf((g.px + hx) * (1.0f / TILE_X)) + 1;
      const int ty0 = (int)floorf((g.py - hy) * (1.0f / TILE_Y)), ty1 = (int)floorf((g.py + hy) * (1.0f / TILE_Y)) + 1;
      g.minx = (uint32_t)max((int)g.minx, tx0); g.maxx = (uint32_t)max((int)g.minx, min((int)g.maxx, tx1));
      g.miny = (uint32_t)max((int)g.miny, ty0); g.maxy = (uint32_t)max((int)g.miny, min((int)g.maxy, ty1));
    } else {
      g.maxx = g.minx;   // alpha >= 1/255 is impossible: no instance at all
    }
  }
  b.touched = (g.maxy - g.miny) * (g.maxx - g.minx);
  b.depth_bits = __float_as_uint(g.depth);
  b.rect_min = g.minx | (g.miny << 16);
  b.rect_max = g.maxx | (g.maxy << 16);
  const float4 sq = scaled_conic(g.conic.x, g.conic.y, g.conic.z);
  b.q0 = make_float4(g.px, g.py, sq.x, sq.y);
  b.q1 = make_float4(sq.z, opacity, col.x, col.y);
  b.q2 = make_float4(col.z, g.depth, __uint_as_float(b.rect_min), __uint_as_float(b.rect_max));   // [10..11]: the emitter's rectangle
  b.q3 = make_float4(g.conic.x, g.conic.y, g.conic.z, 0.f);
  return b;
}

}  // namespace segs
