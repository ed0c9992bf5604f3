// preprocess.hip -- per-Gaussian kernels of the rasterizer (gfx950 / MI355X).
//
//   preprocess_fwd_kernel   K1  reference: preprocessCUDA          cuda_rasterizer/forward.cu:155-256
//   visible_filter_kernel   K2  reference: filter_preprocessCUDA   cuda_rasterizer/forward.cu:259-334
//   mark_visible_kernel     K4  reference: checkFrustum            cuda_rasterizer/rasterizer_impl.cu:54-66
//   preprocess_bwd_kernel   K12+K13 fused; reference: computeCov2DCUDA backward.cu:144-274,
//                           preprocessCUDA backward.cu:346-396, computeCov3D backward.cu:278-341
//
// All are HBM-bound streaming kernels (one Gaussian per lane, 256-thread workgroups):
//  * the (P,3) AoS inputs are fetched as three fully coalesced dword sweeps per workgroup and
//    transposed through LDS (stride-3 LDS reads are conflict free), rotations as 16 B/lane;
//  * everything the tile kernels need is emitted as ONE 64-byte record per Gaussian (gs_layout.h);
//  * the per-workgroup sum of tiles_touched is produced here so the prefix sum needs no extra pass
//    over the per-Gaussian data;
//  * the backward fuses the reference's two kernels and its 108 B/Gaussian of zero-fills (K14): every
//    output row is written exactly once (zeros for culled Gaussians), cov3D is recomputed instead of
//    being stored and re-read.
//
// Arithmetic: this translation unit is built with -ffp-contract=off and evaluates every expression
// in the reference's order with one binary32 rounding per operation, so that radii, rects, depths
// (hence tile/sort keys) are bit-identical to the CPU oracle.  glm semantics: mat3 is column-major,
// m[c][r] = column c, row r.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gs_layout.h"
#include "kernels.h"

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

// Workgroup-coalesced fetch of row `tid` of a (P,3) float array: three dword sweeps + LDS transpose.
__device__ __forceinline__ float3 load_row3(const float* __restrict__ a, int P, float* lds /*768 floats*/) {
  const int tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 768;
  const size_t lim = (size_t)P * 3;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const size_t i = base + k * 256 + tid;
    lds[k * 256 + tid] = i < lim ? a[i] : 0.f;
  }
  __syncthreads();
  float3 v = make_float3(lds[3 * tid], lds[3 * tid + 1], lds[3 * tid + 2]);
  __syncthreads();
  return v;
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


// ---- spherical harmonics: SH -> RGB (forward.cu:20-71) and its backward (backward.cu:20-139).  Off the live
// SEGS-SLAM path (the renderer always passes colors_precomp, src/gaussian_renderer.cpp:86-99); kept for API parity.
// glm::vec3 semantics: componentwise ops, dot summed left to right.
__device__ const float SH_C0 = 0.28209479177387814f;
__device__ const float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                  -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
struct v3 { float x, y, z; };
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3 operator*(float s, v3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float length(v3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ v3 ldv3(const float* p) { return {p[0], p[1], p[2]}; }

__device__ v3 sh_to_rgb(int idx, int deg, int max_coeffs, float3 mean, const float* campos_, const float* shs, uint32_t* clamp_bits) {
  v3 pos = {mean.x, mean.y, mean.z};
  v3 campos = {campos_[0], campos_[1], campos_[2]};
  v3 dir = pos - campos;
  dir = dir / length(dir);
  const float* shp = shs + (size_t)idx * max_coeffs * 3;
#define SH(k) ldv3(shp + 3 * (k))
  v3 result = SH_C0 * SH(0);
  if (deg > 0) {
    float x = dir.x, y = dir.y, z = dir.z;
    result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
               SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
      if (deg > 2) {
        result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                 SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                 SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                 SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
      }
    }
  }
  result = result + v3{0.5f, 0.5f, 0.5f};
  *clamp_bits = (result.x < 0 ? 1u : 0u) | (result.y < 0 ? 2u : 0u) | (result.z < 0 ? 4u : 0u);
  return {fmaxf(result.x, 0.0f), fmaxf(result.y, 0.0f), fmaxf(result.z, 0.0f)};
}

__device__ __forceinline__ v3 dnormvdv(v3 v, v3 dv) {  // auxiliary.h:109-120
  float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  v3 r;
  r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return r;
}

// returns the contribution to dL_dmean (backward.cu:138); writes the active rows of dL_dsh
__device__ v3 sh_to_rgb_backward(int idx, int deg, int max_coeffs, float3 mean, const float* campos_, const float* shs,
                                 uint32_t clamp_bits, v3 dL_dRGB, float* dL_dshs) {
  v3 pos = {mean.x, mean.y, mean.z};
  v3 campos = {campos_[0], campos_[1], campos_[2]};
  v3 dir_orig = pos - campos;
  v3 dir = dir_orig / length(dir_orig);
  const float* shp = shs + (size_t)idx * max_coeffs * 3;
  dL_dRGB.x *= (clamp_bits & 1u) ? 0 : 1; dL_dRGB.y *= (clamp_bits & 2u) ? 0 : 1; dL_dRGB.z *= (clamp_bits & 4u) ? 0 : 1;
  v3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
  float x = dir.x, y = dir.y, z = dir.z;
  float* out = dL_dshs + (size_t)idx * max_coeffs * 3;
#define ST(k, val) do { const v3 _v = (val); out[3 * (k)] = _v.x; out[3 * (k) + 1] = _v.y; out[3 * (k) + 2] = _v.z; } while (0)
  float dRGBdsh0 = SH_C0;
  ST(0, dRGBdsh0 * dL_dRGB);
  if (deg > 0) {
    float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
    ST(1, dRGBdsh1 * dL_dRGB); ST(2, dRGBdsh2 * dL_dRGB); ST(3, dRGBdsh3 * dL_dRGB);
    dRGBdx = -SH_C1 * SH(3); dRGBdy = -SH_C1 * SH(1); dRGBdz = SH_C1 * SH(2);
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      float dRGBdsh4 = SH_C2[0] * xy, dRGBdsh5 = SH_C2[1] * yz, dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
      float dRGBdsh7 = SH_C2[3] * xz, dRGBdsh8 = SH_C2[4] * (xx - yy);
      ST(4, dRGBdsh4 * dL_dRGB); ST(5, dRGBdsh5 * dL_dRGB); ST(6, dRGBdsh6 * dL_dRGB); ST(7, dRGBdsh7 * dL_dRGB); ST(8, dRGBdsh8 * dL_dRGB);
      dRGBdx = dRGBdx + (SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8));
      dRGBdy = dRGBdy + (SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8));
      dRGBdz = dRGBdz + (SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7));
      if (deg > 2) {
        float dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy), dRGBdsh10 = SH_C3[1] * xy * z, dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
        float dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
        float dRGBdsh14 = SH_C3[5] * z * (xx - yy), dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
        ST(9, dRGBdsh9 * dL_dRGB); ST(10, dRGBdsh10 * dL_dRGB); ST(11, dRGBdsh11 * dL_dRGB); ST(12, dRGBdsh12 * dL_dRGB);
        ST(13, dRGBdsh13 * dL_dRGB); ST(14, dRGBdsh14 * dL_dRGB); ST(15, dRGBdsh15 * dL_dRGB);
        dRGBdx = dRGBdx + (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                           SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                           SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy));
        dRGBdy = dRGBdy + (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz + SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                           SH_C3[3] * SH(12) * -3.f * 2.f * yz + SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz +
                           SH_C3[6] * SH(15) * -3.f * 2.f * xy);
        dRGBdz = dRGBdz + (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz + SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                           SH_C3[4] * SH(13) * 4.f * 2.f * xz + SH_C3[5] * SH(14) * (xx - yy));
      }
    }
  }
#undef ST
#undef SH
  v3 dL_ddir = {dot(dRGBdx, dL_dRGB), dot(dRGBdy, dL_dRGB), dot(dRGBdz, dL_dRGB)};
  return dnormvdv(dir_orig, dL_ddir);
}

__global__ void __launch_bounds__(256) preprocess_fwd_kernel(
    int P, const float* __restrict__ means3D, const float* __restrict__ scales, float mod,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ colors,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y, uint32_t gx, uint32_t gy,
    int* __restrict__ radii, float* __restrict__ rec, BinInfo* __restrict__ bin, uint32_t* __restrict__ block_sums,
    uint32_t* __restrict__ depth_range /* per workgroup: [b] = max(depth_bits), [nblocks + b] = max(~depth_bits), visible only */,
    const float* __restrict__ shs, int D, int M, const float* __restrict__ cam_pos, uint32_t* __restrict__ clamped,
    uint32_t flags, uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ depth_vals, uint2* __restrict__ ranges,
    int num_tiles, uint32_t* __restrict__ depth_overflow /* resident: set when a binned depth leaves the 27-bit key range */,
    uint32_t* __restrict__ touched_dense) {
  __shared__ float lds[768];
  __shared__ uint32_t wave_sums[4], wave_dmax[4], wave_dnmin[4];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const float3 p = load_row3(means3D, P, lds);
  float3 sc = make_float3(0, 0, 0);
  float4 rot = make_float4(0, 0, 0, 0);
  if (scales) {
    sc = load_row3(scales, P, lds);
    if (idx < P) rot = reinterpret_cast<const float4*>(rotations)[idx];
  }
  float3 col = make_float3(0.f, 0.f, 0.f);
  if (colors) col = load_row3(colors, P, lds);

  uint32_t touched = 0, dbits_mine = 0;
  if (idx < P) {
    Projected g = project_gaussian(p, sc, mod, rot, cov3D_precomp ? cov3D_precomp + (size_t)6 * idx : nullptr,
                                   viewmatrix, projmatrix, W, H, tan_fovx, tan_fovy, focal_x, focal_y, gx, gy);
    // SEGS_RASTER_SKIP_NONPOSITIVE_OPACITY: such a Gaussian can never pass alpha >= 1/255; dropping it here equals
    // the reference's compaction of masked-out neural Gaussians before the rasterizer (gaussian_renderer.cpp:320)
    if ((flags & 1u) && !(opacities[idx] > 0.f)) g.radius = 0;
    BinInfo b{0u, 0u, 0u, 0u};
    if (g.radius > 0) {
      if (flags & PREPROCESS_TIGHT_RECT) {
        // Resident mode: bin only the tiles the alpha >= 1/255 ellipse can reach.  d^T Q d <= k = 2 ln(255 o) has the
        // axis-aligned half extents sqrt(k cov_xx), sqrt(k cov_yy) (cov = Q^-1, the dilated 2D covariance); the rect is
        // the reference's (3 sigma square, getRect) INTERSECTED with that box, never larger, so exactly the reference's
        // contributing pairs remain.  k is inflated like the emitter's (binning.hip), the extents once more.
        const float op0 = opacities[idx];
        if (op0 * 255.0f > 1.0f) {
          const float k = 2.0f * __logf(255.0f * op0) * 1.0001f + 1e-3f;
          const float hx = sqrtf(k * g.cov_a) * 1.0001f + 0.01f, hy = sqrtf(k * g.cov_c) * 1.0001f + 0.01f;
          const int tx0 = (int)floorf((g.px - hx) * (1.0f / TILE_X)), tx1 = (int)floorf((g.px + hx) * (1.0f / TILE_X)) + 1;
          const int ty0 = (int)floorf((g.py - hy) * (1.0f / TILE_Y)), ty1 = (int)floorf((g.py + hy) * (1.0f / TILE_Y)) + 1;
          g.minx = (uint32_t)max((int)g.minx, tx0); g.maxx = (uint32_t)max((int)g.minx, min((int)g.maxx, tx1));
          g.miny = (uint32_t)max((int)g.miny, ty0); g.maxy = (uint32_t)max((int)g.miny, min((int)g.maxy, ty1));
        } else {
          g.maxx = g.minx;   // alpha >= 1/255 is impossible: no instance at all
        }
      }
      touched = (g.maxy - g.miny) * (g.maxx - g.minx);
      b.depth_bits = __float_as_uint(g.depth);
      dbits_mine = b.depth_bits;
      b.rect_min = g.minx | (g.miny << 16);
      b.rect_max = g.maxx | (g.maxy << 16);
      b.tiles_touched = touched;
      const float op = opacities[idx];
      if (!colors) {  // forward.cu:241-247
        uint32_t cb;
        const v3 c3 = sh_to_rgb(idx, D, M, p, cam_pos, shs, &cb);
        col = make_float3(c3.x, c3.y, c3.z);
        clamped[idx] = cb;
      }
      float4* r4 = reinterpret_cast<float4*>(rec + (size_t)idx * REC_DWORDS);
      // A2/B2/C2: conic pre-scaled so the tile kernels evaluate alpha = o * exp2(A2 dx^2 + B2 dx dy + C2 dy^2)
      r4[0] = make_float4(g.px, g.py, (-0.5f * LOG2E) * g.conic.x, (-LOG2E) * g.conic.y);
      r4[1] = make_float4((-0.5f * LOG2E) * g.conic.z, op, col.x, col.y);
      r4[2] = make_float4(col.z, g.conic.x, g.conic.y, g.conic.z);
      r4[3] = make_float4(g.depth, 0.f, 0.f, 0.f);
    }
    radii[idx] = g.radius;
    reinterpret_cast<uint4*>(bin)[idx] = make_uint4(b.depth_bits, b.rect_min, b.rect_max, b.tiles_touched);
    touched_dense[idx] = b.tiles_touched;
    if (depth_keys) {   // resident mode: what make_depth_keys_kernel would write (32-bit depth keys, culled = 0xFFFFFFFF)
      // The resident depth sort looks at DEPTH_KEY_BITS bits above the near plane's bit pattern (three 9-bit passes).
      // Gaussians without instances carry the all-ones key: the FIRST pass gives them no histogram count and no rank, so the
      // later passes, the depth-ordered prefix and the emitter's searches only see the binned ones (30 % fewer at 3 M).  A
      // binned depth beyond the key range -- 0.2 * 2^16 = 13 107 -- flags the step; the host redoes it through the exact path.
      depth_keys[idx] = b.tiles_touched ? b.depth_bits : 0xFFFFFFFFu;
      (void)depth_vals;   // the sort's first pass takes the index itself as the value (no iota array)
      if (depth_overflow && b.tiles_touched && (b.depth_bits - DEPTH_KEY_MIN) >= ((1u << DEPTH_KEY_BITS) - 1u)) *depth_overflow = 1u;
    }
  }
  if (ranges)
    for (int t = idx; t < num_tiles; t += gridDim.x * 256) ranges[t] = make_uint2(0u, 0u);   // rasterizer_impl.cu:310
  // workgroup sum of tiles_touched -> block_sums[blockIdx] (feeds the prefix sum, K5); depth range for the sort
  uint32_t s = touched;
  uint32_t dmax = touched ? dbits_mine : 0u, dnmin = touched ? ~dbits_mine : 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    dmax = max(dmax, (uint32_t)__shfl_down((int)dmax, off, 64));
    dnmin = max(dnmin, (uint32_t)__shfl_down((int)dnmin, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    wave_sums[threadIdx.x >> 6] = s; wave_dmax[threadIdx.x >> 6] = dmax; wave_dnmin[threadIdx.x >> 6] = dnmin;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    block_sums[blockIdx.x] = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
    depth_range[blockIdx.x] = max(max(wave_dmax[0], wave_dmax[1]), max(wave_dmax[2], wave_dmax[3]));
    depth_range[gridDim.x + blockIdx.x] = max(max(wave_dnmin[0], wave_dnmin[1]), max(wave_dnmin[2], wave_dnmin[3]));
  }
}

__global__ void __launch_bounds__(256) visible_filter_kernel(
    int P, const float* __restrict__ means3D, const float* __restrict__ scales, float mod,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, int W, int H,
    float tan_fovx, float tan_fovy, float focal_x, float focal_y, uint32_t gx, uint32_t gy, int* __restrict__ radii) {
  __shared__ float lds[768];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const float3 p = load_row3(means3D, P, lds);
  float3 sc = make_float3(0, 0, 0);
  float4 rot = make_float4(0, 0, 0, 0);
  if (scales) {
    sc = load_row3(scales, P, lds);
    if (idx < P) rot = reinterpret_cast<const float4*>(rotations)[idx];
  }
  if (idx >= P) return;
  Projected g = project_gaussian(p, sc, mod, rot, cov3D_precomp ? cov3D_precomp + (size_t)6 * idx : nullptr,
                                 viewmatrix, projmatrix, W, H, tan_fovx, tan_fovy, focal_x, focal_y, gx, gy);
  radii[idx] = g.radius;
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ viewmatrix, uint8_t* __restrict__ present) {
  __shared__ float lds[768];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const float3 p = load_row3(means3D, P, lds);
  if (idx >= P) return;
  float3 p_view = transformPoint4x3(p, viewmatrix);
  present[idx] = (p_view.z <= 0.2f) ? 0 : 1;
}

// Fused K12 + K13.  gacc rows hold the tile kernel's per-Gaussian sums (gs_layout.h); if gacc is null the
// caller supplied dL_dmean2D / dL_dconic directly (test hook for the bit-exactness check).
__global__ void __launch_bounds__(256) preprocess_bwd_kernel(
    int P, const float* __restrict__ means3D, const int* __restrict__ radii, const float* __restrict__ scales,
    const float* __restrict__ rotations, float mod, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ view, const float* __restrict__ proj, float h_x, float h_y, float tan_fovx, float tan_fovy,
    float* __restrict__ gacc, const float* __restrict__ rec_in, float img_w, float img_h,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D,
    float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
    const float* __restrict__ shs, int D, int M, const float* __restrict__ cam_pos, const uint32_t* __restrict__ clamped,
    float* __restrict__ dL_dsh, int clean_gacc /* write zeros back over the consumed accumulator row (resident backward) */) {
  __shared__ float lds[768];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const float3 mean = load_row3(means3D, P, lds);
  float3 scale = make_float3(0, 0, 0);
  if (scales) scale = load_row3(scales, P, lds);
  if (idx >= P) return;

  float g2x, g2y, gcx, gcy, gcw;
  float dcol0 = 0.f, dcol1 = 0.f, dcol2 = 0.f;
  if (gacc) {
    // Rows hold raw moments from the tile kernel (render.hip): Mx My Mxx Mxy | Myy S0 Sr Sg | Sb, sums over the
    // (pixel, Gaussian) pairs of w=dL_dG*G times 1, dx, dy, ...; the reference's per-pair terms
    // (backward.cu:541-554) are linear in them:
    //   dL_dmean2D.x = sum dL_dG*(-G dx A - G dy B)*(W/2) = -(A Mx + B My) W/2,  .y = -(C My + B Mx) H/2
    //   dL_dconic    = -0.5 (Mxx, Mxy, Myy);   dL_dopacity = sum G dL_dalpha = S0 / o
    const float4* a = reinterpret_cast<const float4*>(gacc + (size_t)idx * GACC_DWORDS);
    const float4 a0 = a[0], a1 = a[1];
    const float a8 = gacc[(size_t)idx * GACC_DWORDS + 8];
    if (clean_gacc && radii[idx] > 0) {   // only binned Gaussians' rows can have been touched by the tile kernel
      float4* z = reinterpret_cast<float4*>(gacc + (size_t)idx * GACC_DWORDS);
      z[0] = make_float4(0.f, 0.f, 0.f, 0.f); z[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      gacc[(size_t)idx * GACC_DWORDS + 8] = 0.f;
    }
    float dop = 0.f;
    g2x = 0.f; g2y = 0.f; gcx = 0.f; gcy = 0.f; gcw = 0.f;
    if (radii[idx] > 0) {
      const float4 q2 = reinterpret_cast<const float4*>(rec_in + (size_t)idx * REC_DWORDS)[2];  // b, A, B, C
      const float op = rec_in[(size_t)idx * REC_DWORDS + REC_O];
      g2x = -(q2.y * a0.x + q2.z * a0.y) * (0.5f * img_w);
      g2y = -(q2.w * a0.y + q2.z * a0.x) * (0.5f * img_h);
      gcx = -0.5f * a0.z; gcy = -0.5f * a0.w; gcw = -0.5f * a1.x;
      dop = a1.y != 0.f ? a1.y / op : 0.f;
    }
    dL_dmean2D[3 * (size_t)idx + 0] = g2x; dL_dmean2D[3 * (size_t)idx + 1] = g2y; dL_dmean2D[3 * (size_t)idx + 2] = 0.f;
    if (dL_dconic) reinterpret_cast<float4*>(dL_dconic)[idx] = make_float4(gcx, gcy, 0.f, gcw);   // internal product: optional
    dL_dopacity[idx] = dop;
    dL_dcolor[3 * (size_t)idx + 0] = a1.z; dL_dcolor[3 * (size_t)idx + 1] = a1.w; dL_dcolor[3 * (size_t)idx + 2] = a8;
    dcol0 = a1.z; dcol1 = a1.w; dcol2 = a8;
  } else {
    g2x = dL_dmean2D[3 * (size_t)idx + 0]; g2y = dL_dmean2D[3 * (size_t)idx + 1];
    gcx = dL_dconic[4 * (size_t)idx + 0]; gcy = dL_dconic[4 * (size_t)idx + 1]; gcw = dL_dconic[4 * (size_t)idx + 3];
  }

  float out_mean[3] = {0, 0, 0}, out_cov[6] = {0, 0, 0, 0, 0, 0}, out_scale[3] = {0, 0, 0}, out_rot[4] = {0, 0, 0, 0};
  if (radii[idx] > 0) {
    float4 rot = make_float4(0, 0, 0, 0);
    float cov3D[6];
    if (cov3D_precomp) {
#pragma unroll
      for (int k = 0; k < 6; k++) cov3D[k] = cov3D_precomp[(size_t)6 * idx + k];
    } else {
      rot = reinterpret_cast<const float4*>(rotations)[idx];
      computeCov3D(scale, mod, rot, cov3D);
    }
    // ---- K12: backward.cu:160-273
    const float3 dL_dconic3 = make_float3(gcx, gcy, gcw);
    Cov2DTerms q;
    float3 abc = computeCov2D(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view, &q);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float x_grad_mul = q.txtz < -limx || q.txtz > limx ? 0 : 1;
    const float y_grad_mul = q.tytz < -limy || q.tytz > limy ? 0 : 1;
    const mat3& T = q.T; const mat3& Wm = q.W; const mat3& Vrk = q.Vrk; const float3 t = q.t;
    float a = abc.x, b = abc.y, c = abc.z;
    float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    if (denom2inv != 0) {
      dL_da = denom2inv * (-c * c * dL_dconic3.x + 2 * b * c * dL_dconic3.y + (denom - a * c) * dL_dconic3.z);
      dL_dc = denom2inv * (-a * a * dL_dconic3.z + 2 * a * b * dL_dconic3.y + (denom - a * c) * dL_dconic3.x);
      dL_db = denom2inv * 2 * (b * c * dL_dconic3.x - (denom + 2 * b * b) * dL_dconic3.y + a * b * dL_dconic3.z);
      out_cov[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
      out_cov[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
      out_cov[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
      out_cov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
      out_cov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
      out_cov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
    }
    float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_da +
                    (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_db;
    float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_da +
                    (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_db;
    float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_da +
                    (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_db;
    float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc +
                    (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_db;
    float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc +
                    (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_db;
    float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc +
                    (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_db;
    float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
    float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
    float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
    float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
    float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
    // transformVec4x3Transpose (auxiliary.h:90-98)
    float k12x = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
    float k12y = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
    float k12z = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;
    // ---- K13: backward.cu:370-387
    float4 m_hom = transformPoint4x4(mean, proj);
    float m_w = 1.0f / (m_hom.w + 0.0000001f);
    float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
    float dmx = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    float dmy = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    float dmz = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    out_mean[0] = k12x + dmx; out_mean[1] = k12y + dmy; out_mean[2] = k12z + dmz;  // "dL_dmeans[idx] += dL_dmean" (:387)
    if (shs && gacc) {  // backward.cu:390-391 (needs the summed dL_dcolor of this Gaussian)
      const v3 g = sh_to_rgb_backward(idx, D, M, mean, cam_pos, shs, clamped[idx], v3{dcol0, dcol1, dcol2}, dL_dsh);
      out_mean[0] = out_mean[0] + g.x; out_mean[1] = out_mean[1] + g.y; out_mean[2] = out_mean[2] + g.z;
    }
    // ---- computeCov3D backward: backward.cu:278-341 (no quaternion-normalisation Jacobian, F5b)
    if (scales) {
      const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
      mat3 Rm = quat_to_R(rot);
      mat3 S = mk(1, 0, 0, 0, 1, 0, 0, 0, 1);
      const float3 s = make_float3(mod * scale.x, mod * scale.y, mod * scale.z);
      S.m[0][0] = s.x; S.m[1][1] = s.y; S.m[2][2] = s.z;
      mat3 M = mul(S, Rm);
      const float* g = out_cov;
      mat3 dL_dSigma = mk(g[0], 0.5f * g[1], 0.5f * g[2], 0.5f * g[1], g[3], 0.5f * g[4], 0.5f * g[2], 0.5f * g[4], g[5]);
      mat3 dL_dM = mul(smul(2.0f, M), dL_dSigma);
      mat3 Rt = transpose(Rm);
      mat3 dL_dMt = transpose(dL_dM);
      out_scale[0] = Rt.m[0][0] * dL_dMt.m[0][0] + Rt.m[0][1] * dL_dMt.m[0][1] + Rt.m[0][2] * dL_dMt.m[0][2];
      out_scale[1] = Rt.m[1][0] * dL_dMt.m[1][0] + Rt.m[1][1] * dL_dMt.m[1][1] + Rt.m[1][2] * dL_dMt.m[1][2];
      out_scale[2] = Rt.m[2][0] * dL_dMt.m[2][0] + Rt.m[2][1] * dL_dMt.m[2][1] + Rt.m[2][2] * dL_dMt.m[2][2];
#pragma unroll
      for (int k = 0; k < 3; k++) { dL_dMt.m[0][k] *= s.x; dL_dMt.m[1][k] *= s.y; dL_dMt.m[2][k] *= s.z; }
      out_rot[0] = 2 * z * (dL_dMt.m[0][1] - dL_dMt.m[1][0]) + 2 * y * (dL_dMt.m[2][0] - dL_dMt.m[0][2]) + 2 * x * (dL_dMt.m[1][2] - dL_dMt.m[2][1]);
      out_rot[1] = 2 * y * (dL_dMt.m[1][0] + dL_dMt.m[0][1]) + 2 * z * (dL_dMt.m[2][0] + dL_dMt.m[0][2]) + 2 * r * (dL_dMt.m[1][2] - dL_dMt.m[2][1]) - 4 * x * (dL_dMt.m[2][2] + dL_dMt.m[1][1]);
      out_rot[2] = 2 * x * (dL_dMt.m[1][0] + dL_dMt.m[0][1]) + 2 * r * (dL_dMt.m[2][0] - dL_dMt.m[0][2]) + 2 * z * (dL_dMt.m[1][2] + dL_dMt.m[2][1]) - 4 * y * (dL_dMt.m[2][2] + dL_dMt.m[0][0]);
      out_rot[3] = 2 * r * (dL_dMt.m[0][1] - dL_dMt.m[1][0]) + 2 * x * (dL_dMt.m[2][0] + dL_dMt.m[0][2]) + 2 * y * (dL_dMt.m[1][2] + dL_dMt.m[2][1]) - 4 * z * (dL_dMt.m[1][1] + dL_dMt.m[0][0]);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) dL_dmean3D[3 * (size_t)idx + k] = out_mean[k];
  if (dL_dcov3D) {   // only a caller that passed cov3D_precomp has a use for it
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)idx + k] = out_cov[k];
  }
  if (dL_dscale) {
#pragma unroll
    for (int k = 0; k < 3; k++) dL_dscale[3 * (size_t)idx + k] = out_scale[k];
  }
  if (dL_drot) reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(out_rot[0], out_rot[1], out_rot[2], out_rot[3]);
}

}  // namespace segs
