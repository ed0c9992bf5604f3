// gs_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// A plain C++17/OpenMP restatement of the reference's differentiable 3D-Gaussian
// tile rasterizer (leaner-forever/SEGS-SLAM, cuda_rasterizer/).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product path (segs-slam_amd/csrc) never links, includes or calls it.
//
// PARITY STATUS: "parity unpinned" by the reference -- the reference ships no
// tests, golden vectors or KATs for this path and cannot be built
// here (CUDA, glm, CUB absent).  The oracle is pinned instead by
//   (a) an independent PyTorch-autograd formulation (oracle/torch_ref.py),
//   (b) exact integer invariants checked in tests/test_oracle.py,
//   (c) the camera-matrix known-answer data of check_colmap.md (tests/golden/).
//
// Arithmetic convention (stated because nvcc's FMA contraction is unobservable
// here): every expression is evaluated exactly as written in the reference, one
// IEEE-754 binary32 rounding per operation, NO fused multiply-add contraction
// (build with -ffp-contract=off).  glm semantics restated: mat3(a..i) fills
// COLUMNS, M[c][r] is column c row r, (A*B)[c][r] = A[0][r]*B[c][0] +
// A[1][r]*B[c][1] + A[2][r]*B[c][2] summed left to right.
//
// Each function cites the reference file:line it follows (paths relative to
// /root/reference).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int BLOCK_X = 16;   // cuda_rasterizer/config.h:16
constexpr int BLOCK_Y = 16;   // cuda_rasterizer/config.h:17
constexpr int NCH = 3;        // cuda_rasterizer/config.h:15

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct u2 { uint32_t x, y; };

// glm::mat3 stand-in: m[c][r] = column c, row r.
struct mat3 {
  float m[3][3];
  float* operator[](int c) { return m[c]; }
  const float* operator[](int c) const { return m[c]; }
};
// glm::mat3(a,b,c, d,e,f, g,h,i): columns (a,b,c), (d,e,f), (g,h,i).
static inline mat3 mk(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
  mat3 r; r.m[0][0]=a; r.m[0][1]=b; r.m[0][2]=c; r.m[1][0]=d; r.m[1][1]=e; r.m[1][2]=f; r.m[2][0]=g; r.m[2][1]=h; r.m[2][2]=i; return r;
}
static inline mat3 mul(const mat3& A, const mat3& B) {
  mat3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++)
      R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
  return R;
}
static inline mat3 transpose(const mat3& A) {
  mat3 R;
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
  return R;
}
static inline mat3 smul(float s, const mat3& A) {
  mat3 R;
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) R.m[c][r] = s * A.m[c][r];
  return R;
}

// cuda_rasterizer/auxiliary.h:41-45 (double-literal arithmetic, narrowed on return)
static inline float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// cuda_rasterizer/auxiliary.h:47-57
static inline void getRect(f2 p, int max_radius, u2& rmin, u2& rmax, uint32_t gx, uint32_t gy) {
  rmin.x = std::min(gx, (uint32_t)std::max(0, (int)((p.x - max_radius) / BLOCK_X)));
  rmin.y = std::min(gy, (uint32_t)std::max(0, (int)((p.y - max_radius) / BLOCK_Y)));
  rmax.x = std::min(gx, (uint32_t)std::max(0, (int)((p.x + max_radius + BLOCK_X - 1) / BLOCK_X)));
  rmax.y = std::min(gy, (uint32_t)std::max(0, (int)((p.y + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

// cuda_rasterizer/auxiliary.h:59-78
static inline f3 transformPoint4x3(f3 p, const float* M) {
  return { M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12],
           M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
           M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14] };
}
static inline f4 transformPoint4x4(f3 p, const float* M) {
  return { M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12],
           M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
           M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14],
           M[3] * p.x + M[7] * p.y + M[11] * p.z + M[15] };
}
// cuda_rasterizer/auxiliary.h:90-98
static inline f3 transformVec4x3Transpose(f3 p, const float* M) {
  return { M[0] * p.x + M[1] * p.y + M[2] * p.z,
           M[4] * p.x + M[5] * p.y + M[6] * p.z,
           M[8] * p.x + M[9] * p.y + M[10] * p.z };
}

// cuda_rasterizer/auxiliary.h:140-166 (x/y frustum test removed in this fork: F5a)
static inline bool in_frustum(int idx, const float* pts, const float* view, f3& p_view) {
  f3 p = { pts[3 * idx], pts[3 * idx + 1], pts[3 * idx + 2] };
  p_view = transformPoint4x3(p, view);
  return !(p_view.z <= 0.2f);
}

// cuda_rasterizer/forward.cu:118-152 (quaternion used un-normalised: F5b)
static inline void computeCov3D(f3 scale, float mod, f4 rot, float* cov3D) {
  mat3 S = mk(1, 0, 0, 0, 1, 0, 0, 0, 1);
  S[0][0] = mod * scale.x; S[1][1] = mod * scale.y; S[2][2] = mod * scale.z;
  float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
  mat3 R = mk(
    1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
    2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
    2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
  mat3 M = mul(S, R);
  mat3 Sigma = mul(transpose(M), M);
  cov3D[0] = Sigma[0][0]; cov3D[1] = Sigma[0][1]; cov3D[2] = Sigma[0][2];
  cov3D[3] = Sigma[1][1]; cov3D[4] = Sigma[1][2]; cov3D[5] = Sigma[2][2];
}

// cuda_rasterizer/forward.cu:74-113
static inline f3 computeCov2D(f3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                              const float* cov3D, const float* view) {
  f3 t = transformPoint4x3(mean, view);
  const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
  const float txtz = t.x / t.z, tytz = t.y / t.z;
  t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
  t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
  mat3 J = mk(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z),
              0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z),
              0, 0, 0);
  mat3 W = mk(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
  mat3 T = mul(W, J);
  mat3 Vrk = mk(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
  mat3 cov = mul(mul(transpose(T), transpose(Vrk)), T);
  cov[0][0] += 0.3f; cov[1][1] += 0.3f;
  return { cov[0][0], cov[0][1], cov[1][1] };
}

// cuda_rasterizer/rasterizer_impl.cu:35-50
static uint32_t getHigherMsb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4, step = msb;
  while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
  if (n >> msb) msb++;
  return msb;
}


// ---- spherical harmonics (cuda_rasterizer/auxiliary.h:22-40 constants; glm::vec3 semantics: componentwise,
// scalar*vec, dot = x*x' + y*y' + z*z' summed left to right, length = sqrt(dot(v,v))) ----
const float SH_C0 = 0.28209479177387814f;
const float SH_C1 = 0.4886025119029199f;
const float SH_C2[] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f };
const float SH_C3[] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                        -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };
struct v3 { float x, y, z; };
static inline v3 operator+(v3 a, v3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline v3 operator-(v3 a, v3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline v3 operator*(float s, v3 a) { return { s * a.x, s * a.y, s * a.z }; }
static inline v3 operator*(v3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static inline v3 operator/(v3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float length(v3 a) { return std::sqrt(dot(a, a)); }

// forward.cu:20-71
static inline v3 computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos_, const float* shs,
                                    uint8_t* clamped) {
  v3 pos = { means[3 * idx], means[3 * idx + 1], means[3 * idx + 2] };
  v3 campos = { campos_[0], campos_[1], campos_[2] };
  v3 dir = pos - campos;
  dir = dir / length(dir);
  const v3* sh = reinterpret_cast<const v3*>(shs) + (size_t)idx * max_coeffs;
  v3 result = SH_C0 * sh[0];
  if (deg > 0) {
    float x = dir.x, y = dir.y, z = dir.z;
    result = result - SH_C1 * y * sh[1] + SH_C1 * z * sh[2] - SH_C1 * x * sh[3];
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      result = result + SH_C2[0] * xy * sh[4] + SH_C2[1] * yz * sh[5] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[6] +
               SH_C2[3] * xz * sh[7] + SH_C2[4] * (xx - yy) * sh[8];
      if (deg > 2) {
        result = result + SH_C3[0] * y * (3.0f * xx - yy) * sh[9] + SH_C3[1] * xy * z * sh[10] +
                 SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11] + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12] +
                 SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13] + SH_C3[5] * z * (xx - yy) * sh[14] +
                 SH_C3[6] * x * (xx - 3.0f * yy) * sh[15];
      }
    }
  }
  result = result + v3{ 0.5f, 0.5f, 0.5f };
  clamped[3 * idx + 0] = (result.x < 0); clamped[3 * idx + 1] = (result.y < 0); clamped[3 * idx + 2] = (result.z < 0);
  return { std::max(result.x, 0.0f), std::max(result.y, 0.0f), std::max(result.z, 0.0f) };
}

// auxiliary.h:109-120
static inline v3 dnormvdv(v3 v, v3 dv) {
  float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  float invsum32 = 1.0f / std::sqrt(sum2 * sum2 * sum2);
  v3 r;
  r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return r;
}

// backward.cu:20-139; returns the mean-gradient contribution (added to dL_dmeans at :138), writes dL_dsh.
static inline v3 computeColorFromSH_backward(int idx, int deg, int max_coeffs, const float* means, const float* campos_,
                                             const float* shs, const uint8_t* clamped, const float* dL_dcolor, float* dL_dshs) {
  v3 pos = { means[3 * idx], means[3 * idx + 1], means[3 * idx + 2] };
  v3 campos = { campos_[0], campos_[1], campos_[2] };
  v3 dir_orig = pos - campos;
  v3 dir = dir_orig / length(dir_orig);
  const v3* sh = reinterpret_cast<const v3*>(shs) + (size_t)idx * max_coeffs;
  v3 dL_dRGB = { dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2] };
  dL_dRGB.x *= clamped[3 * idx + 0] ? 0 : 1; dL_dRGB.y *= clamped[3 * idx + 1] ? 0 : 1; dL_dRGB.z *= clamped[3 * idx + 2] ? 0 : 1;
  v3 dRGBdx = { 0, 0, 0 }, dRGBdy = { 0, 0, 0 }, dRGBdz = { 0, 0, 0 };
  float x = dir.x, y = dir.y, z = dir.z;
  v3* dL_dsh = reinterpret_cast<v3*>(dL_dshs) + (size_t)idx * max_coeffs;
  float dRGBdsh0 = SH_C0;
  dL_dsh[0] = dRGBdsh0 * dL_dRGB;
  if (deg > 0) {
    float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
    dL_dsh[1] = dRGBdsh1 * dL_dRGB; dL_dsh[2] = dRGBdsh2 * dL_dRGB; dL_dsh[3] = dRGBdsh3 * dL_dRGB;
    dRGBdx = -SH_C1 * sh[3]; dRGBdy = -SH_C1 * sh[1]; dRGBdz = SH_C1 * sh[2];
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      float dRGBdsh4 = SH_C2[0] * xy, dRGBdsh5 = SH_C2[1] * yz, dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
      float dRGBdsh7 = SH_C2[3] * xz, dRGBdsh8 = SH_C2[4] * (xx - yy);
      dL_dsh[4] = dRGBdsh4 * dL_dRGB; dL_dsh[5] = dRGBdsh5 * dL_dRGB; dL_dsh[6] = dRGBdsh6 * dL_dRGB;
      dL_dsh[7] = dRGBdsh7 * dL_dRGB; dL_dsh[8] = dRGBdsh8 * dL_dRGB;
      dRGBdx = dRGBdx + (SH_C2[0] * y * sh[4] + SH_C2[2] * 2.f * -x * sh[6] + SH_C2[3] * z * sh[7] + SH_C2[4] * 2.f * x * sh[8]);
      dRGBdy = dRGBdy + (SH_C2[0] * x * sh[4] + SH_C2[1] * z * sh[5] + SH_C2[2] * 2.f * -y * sh[6] + SH_C2[4] * 2.f * -y * sh[8]);
      dRGBdz = dRGBdz + (SH_C2[1] * y * sh[5] + SH_C2[2] * 2.f * 2.f * z * sh[6] + SH_C2[3] * x * sh[7]);
      if (deg > 2) {
        float dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy), dRGBdsh10 = SH_C3[1] * xy * z, dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
        float dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
        float dRGBdsh14 = SH_C3[5] * z * (xx - yy), dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
        dL_dsh[9] = dRGBdsh9 * dL_dRGB; dL_dsh[10] = dRGBdsh10 * dL_dRGB; dL_dsh[11] = dRGBdsh11 * dL_dRGB;
        dL_dsh[12] = dRGBdsh12 * dL_dRGB; dL_dsh[13] = dRGBdsh13 * dL_dRGB; dL_dsh[14] = dRGBdsh14 * dL_dRGB;
        dL_dsh[15] = dRGBdsh15 * dL_dRGB;
        dRGBdx = dRGBdx + (SH_C3[0] * sh[9] * 3.f * 2.f * xy + SH_C3[1] * sh[10] * yz + SH_C3[2] * sh[11] * -2.f * xy +
                           SH_C3[3] * sh[12] * -3.f * 2.f * xz + SH_C3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) +
                           SH_C3[5] * sh[14] * 2.f * xz + SH_C3[6] * sh[15] * 3.f * (xx - yy));
        dRGBdy = dRGBdy + (SH_C3[0] * sh[9] * 3.f * (xx - yy) + SH_C3[1] * sh[10] * xz + SH_C3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) +
                           SH_C3[3] * sh[12] * -3.f * 2.f * yz + SH_C3[4] * sh[13] * -2.f * xy + SH_C3[5] * sh[14] * -2.f * yz +
                           SH_C3[6] * sh[15] * -3.f * 2.f * xy);
        dRGBdz = dRGBdz + (SH_C3[1] * sh[10] * xy + SH_C3[2] * sh[11] * 4.f * 2.f * yz + SH_C3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) +
                           SH_C3[4] * sh[13] * 4.f * 2.f * xz + SH_C3[5] * sh[14] * (xx - yy));
      }
    }
  }
  v3 dL_ddir = { dot(dRGBdx, dL_dRGB), dot(dRGBdy, dL_dRGB), dot(dRGBdz, dL_dRGB) };
  return dnormvdv(dir_orig, dL_ddir);
}
// glm: (vec3 * float) as used in the degree-3 derivative sums (sh[9] * 3.f ...)

struct Ctx {
  int P = 0, W = 0, H = 0, R = 0;
  uint32_t gx = 0, gy = 0;
  // GeometryState (cuda_rasterizer/rasterizer_impl.h:30-45)
  std::vector<float> depths, cov3D, rgb;
  std::vector<uint8_t> clamped;
  int D = 0, M = 0;
  std::vector<int> radii;
  std::vector<f2> means2D;
  std::vector<f4> conic_opacity;
  std::vector<uint32_t> tiles_touched, point_offsets;
  // BinningState (:56-66)
  std::vector<uint64_t> keys_unsorted, keys;
  std::vector<uint32_t> vals_unsorted, vals;
  // ImageState (:47-54)
  std::vector<u2> ranges;
  std::vector<uint32_t> n_contrib;
  std::vector<float> accum_alpha, out_color;
  // gradients
  std::vector<float> dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, dL_dsh;
};

// Shared body of preprocessCUDA (forward.cu:155-256) and filter_preprocessCUDA (:259-334).
// Returns radius (0 = rejected).  When `full`, fills the per-Gaussian state.
static inline int preprocess_one(Ctx* c, int idx, bool full, const float* means3D, const float* scales,
                                 float mod, const float* rots, const float* opac, const float* cov3D_precomp,
                                 const float* view, const float* proj, int W, int H,
                                 float tan_fovx, float tan_fovy, float focal_x, float focal_y, float* cov_scratch) {
  f3 p_view;
  if (!in_frustum(idx, means3D, view, p_view)) return 0;
  f3 p = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
  f4 p_hom = transformPoint4x4(p, proj);
  float p_w = 1.0f / (p_hom.w + 0.0000001f);
  f3 p_proj = { p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w };
  const float* cov3D;
  if (cov3D_precomp) cov3D = cov3D_precomp + 6 * idx;
  else {
    computeCov3D({ scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2] }, mod,
                 { rots[4 * idx], rots[4 * idx + 1], rots[4 * idx + 2], rots[4 * idx + 3] }, cov_scratch);
    cov3D = cov_scratch;
  }
  f3 cov = computeCov2D(p, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view);
  float det = (cov.x * cov.z - cov.y * cov.y);
  if (det == 0.0f) return 0;
  float det_inv = 1.f / det;
  f3 conic = { cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv };
  float mid = 0.5f * (cov.x + cov.z);
  float lambda1 = mid + std::sqrt(std::max(0.1f, mid * mid - det));
  float lambda2 = mid - std::sqrt(std::max(0.1f, mid * mid - det));
  float my_radius = std::ceil(3.f * std::sqrt(std::max(lambda1, lambda2)));
  f2 pix = { ndc2Pix(p_proj.x, W), ndc2Pix(p_proj.y, H) };
  u2 rmin, rmax;
  getRect(pix, (int)my_radius, rmin, rmax, c->gx, c->gy);
  if ((rmax.x - rmin.x) * (rmax.y - rmin.y) == 0) return 0;
  if (full) {
    c->depths[idx] = p_view.z;
    c->means2D[idx] = pix;
    c->conic_opacity[idx] = { conic.x, conic.y, conic.z, opac[idx] };
    c->tiles_touched[idx] = (rmax.y - rmin.y) * (rmax.x - rmin.x);
  }
  return (int)my_radius;
}

// Stable LSD radix sort of (key,value) pairs on key bits [0,end_bit) -- the semantics of
// cub::DeviceRadixSort::SortPairs(..., 0, 32+bit) at rasterizer_impl.cu:303-308.
static void radix_sort_pairs(std::vector<uint64_t>& k_in, std::vector<uint32_t>& v_in,
                             std::vector<uint64_t>& k_out, std::vector<uint32_t>& v_out, int end_bit) {
  const size_t n = k_in.size();
  k_out.resize(n); v_out.resize(n);
  if (n == 0) return;
  std::vector<uint64_t> ka(k_in), kb(n);
  std::vector<uint32_t> va(v_in), vb(n);
  int nth = 1;
#ifdef _OPENMP
  nth = omp_get_max_threads();
#endif
  std::vector<size_t> hist((size_t)nth * 256);
  for (int shift = 0; shift < end_bit; shift += 8) {
    const int bits = std::min(8, end_bit - shift);
    const uint64_t mask = (1ull << bits) - 1;
    std::fill(hist.begin(), hist.end(), 0);
#pragma omp parallel num_threads(nth)
    {
#ifdef _OPENMP
      int t = omp_get_thread_num();
#else
      int t = 0;
#endif
      size_t lo = n * t / nth, hi = n * (t + 1) / nth;
      size_t* h = &hist[(size_t)t * 256];
      for (size_t i = lo; i < hi; i++) h[(ka[i] >> shift) & mask]++;
#pragma omp barrier
#pragma omp single
      {
        size_t run = 0;
        for (int d = 0; d < 256; d++)
          for (int tt = 0; tt < nth; tt++) { size_t cnt = hist[(size_t)tt * 256 + d]; hist[(size_t)tt * 256 + d] = run; run += cnt; }
      }
      for (size_t i = lo; i < hi; i++) {
        size_t pos = h[(ka[i] >> shift) & mask]++;
        kb[pos] = ka[i]; vb[pos] = va[i];
      }
    }
    ka.swap(kb); va.swap(vb);
  }
  k_out = ka; v_out = va;
}

} // namespace

extern "C" {

void* gso_create() { return new Ctx(); }
void gso_destroy(void* h) { delete (Ctx*)h; }
int gso_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// CudaRasterizer::Rasterizer::markVisible -- rasterizer_impl.cu:54-66,141-153
void gso_mark_visible(int P, const float* means3D, const float* view, const float* /*proj*/, uint8_t* present) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) { f3 pv; present[i] = in_frustum(i, means3D, view, pv) ? 1 : 0; }
}

// CudaRasterizer::Rasterizer::visible_filter -- rasterizer_impl.cu:339-393, forward.cu:259-334
void gso_visible_filter(void* h, int P, int W, int H, const float* means3D, const float* scales, float mod,
                        const float* rots, const float* cov3D_precomp, const float* view, const float* proj,
                        float tan_fovx, float tan_fovy, int* radii) {
  Ctx* c = (Ctx*)h;
  c->gx = (W + BLOCK_X - 1) / BLOCK_X; c->gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  const float focal_y = H / (2.0f * tan_fovy), focal_x = W / (2.0f * tan_fovx);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    float cs[6];
    radii[i] = preprocess_one(c, i, false, means3D, scales, mod, rots, nullptr, cov3D_precomp, view, proj, W, H,
                              tan_fovx, tan_fovy, focal_x, focal_y, cs);
  }
}

// CudaRasterizer::Rasterizer::forward -- rasterizer_impl.cu:198-336.  Returns num_rendered.
// stage_mask: bit0 preprocess+binning (K1,K5,K7,K8,K9), bit1 render (K10).
int gso_forward(void* h, int P, const float* bg, int W, int H, const float* means3D, const float* colors,
                const float* opac, const float* scales, float mod, const float* rots, const float* cov3D_precomp,
                const float* view, const float* proj, float tan_fovx, float tan_fovy,
                const float* shs, int D, int M, const float* campos) {
  Ctx* c = (Ctx*)h;
  c->P = P; c->W = W; c->H = H;
  c->gx = (W + BLOCK_X - 1) / BLOCK_X; c->gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  const float focal_y = H / (2.0f * tan_fovy), focal_x = W / (2.0f * tan_fovx);
  c->depths.assign(P, 0.f); c->cov3D.assign((size_t)6 * P, 0.f); c->radii.assign(P, 0);
  c->means2D.assign(P, { 0, 0 }); c->conic_opacity.assign(P, { 0, 0, 0, 0 });
  c->tiles_touched.assign(P, 0); c->point_offsets.assign(P, 0);
  c->D = D; c->M = M;
  c->clamped.assign((size_t)3 * P, 0);
  if (colors) c->rgb.assign(colors, colors + (size_t)3 * P);  // colors_precomp branch (forward.cu:241; rasterizer_impl.cu:321)
  else c->rgb.assign((size_t)3 * P, 0.f);

  // K1 preprocessCUDA
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    c->radii[i] = preprocess_one(c, i, true, means3D, scales, mod, rots, opac, cov3D_precomp, view, proj, W, H,
                                 tan_fovx, tan_fovy, focal_x, focal_y, &c->cov3D[(size_t)6 * i]);
    if (!colors && c->radii[i] > 0) {  // forward.cu:241-247 (only for Gaussians that passed every reject)
      v3 col = computeColorFromSH(i, D, M, means3D, campos, shs, c->clamped.data());
      c->rgb[3 * i] = col.x; c->rgb[3 * i + 1] = col.y; c->rgb[3 * i + 2] = col.z;
    }
  }
  // K5 InclusiveSum (rasterizer_impl.cu:276-277) and K6 num_rendered (:281)
  uint32_t run = 0;
  for (int i = 0; i < P; i++) { run += c->tiles_touched[i]; c->point_offsets[i] = run; }
  const int R = P > 0 ? (int)c->point_offsets[P - 1] : 0;
  c->R = R;
  // K7 duplicateWithKeys (rasterizer_impl.cu:70-111)
  c->keys_unsorted.assign(R, 0); c->vals_unsorted.assign(R, 0);
#pragma omp parallel for schedule(dynamic, 1024)
  for (int i = 0; i < P; i++) {
    if (c->radii[i] > 0) {
      uint32_t off = (i == 0) ? 0 : c->point_offsets[i - 1];
      u2 rmin, rmax;
      getRect(c->means2D[i], c->radii[i], rmin, rmax, c->gx, c->gy);
      uint32_t dbits; std::memcpy(&dbits, &c->depths[i], 4);
      for (uint32_t y = rmin.y; y < rmax.y; y++)
        for (uint32_t x = rmin.x; x < rmax.x; x++) {
          uint64_t key = y * c->gx + x; key <<= 32; key |= dbits;
          c->keys_unsorted[off] = key; c->vals_unsorted[off] = (uint32_t)i; off++;
        }
    }
  }
  // K8 SortPairs over bits [0, 32+bit) (rasterizer_impl.cu:300-308)
  const int bit = (int)getHigherMsb(c->gx * c->gy);
  radix_sort_pairs(c->keys_unsorted, c->vals_unsorted, c->keys, c->vals, 32 + bit);
  // K9 memset + identifyTileRanges (rasterizer_impl.cu:310-318,116-138)
  c->ranges.assign((size_t)c->gx * c->gy, { 0, 0 });
  for (int i = 0; i < R; i++) {
    uint32_t cur = (uint32_t)(c->keys[i] >> 32);
    if (i == 0) c->ranges[cur].x = 0;
    else {
      uint32_t prev = (uint32_t)(c->keys[i - 1] >> 32);
      if (cur != prev) { c->ranges[prev].y = i; c->ranges[cur].x = i; }
    }
    if (i == R - 1) c->ranges[cur].y = R;
  }
  // K10 renderCUDA (forward.cu:339-452)
  const size_t HW = (size_t)W * H;
  c->out_color.assign(3 * HW, 0.f); c->accum_alpha.assign(HW, 0.f); c->n_contrib.assign(HW, 0);
  const int ntiles = (int)(c->gx * c->gy);
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < ntiles; tile++) {
    const uint32_t tx = tile % c->gx, ty = tile / c->gx;
    const u2 range = c->ranges[tile];
    for (uint32_t ly = 0; ly < BLOCK_Y; ly++)
      for (uint32_t lx = 0; lx < BLOCK_X; lx++) {
        const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
        const f2 pixf = { (float)px, (float)py };
        float T = 1.0f, C[NCH] = { 0, 0, 0 };
        uint32_t contributor = 0, last_contributor = 0;
        for (uint32_t s = range.x; s < range.y; s++) {
          contributor++;
          const uint32_t id = c->vals[s];
          const f2 xy = c->means2D[id];
          const f2 d = { xy.x - pixf.x, xy.y - pixf.y };
          const f4 con_o = c->conic_opacity[id];
          const float power = -0.5f * (con_o.x * d.x * d.x + con_o.z * d.y * d.y) - con_o.y * d.x * d.y;
          if (power > 0.0f) continue;
          const float alpha = std::min(0.99f, con_o.w * std::exp(power));
          if (alpha < 1.0f / 255.0f) continue;
          const float test_T = T * (1 - alpha);
          if (test_T < 0.0001f) break;  // done = true (forward.cu:421-425)
          for (int ch = 0; ch < NCH; ch++) C[ch] += c->rgb[(size_t)id * NCH + ch] * alpha * T;
          T = test_T;
          last_contributor = contributor;
        }
        const size_t pix_id = (size_t)W * py + px;
        c->accum_alpha[pix_id] = T;
        c->n_contrib[pix_id] = last_contributor;
        for (int ch = 0; ch < NCH; ch++) c->out_color[ch * HW + pix_id] = C[ch] + T * bg[ch];
      }
  }
  return R;
}

// CudaRasterizer::Rasterizer::backward -- rasterizer_impl.cu:397-490 (K11 -> K12 -> K13).
// Per-(pixel,Gaussian) terms are those of backward.cu:464-556; the reference sums them with float
// atomics in arbitrary order, the oracle sums them in double (per tile instance, then per Gaussian
// in sorted-instance order) and rounds once, so it is the order-independent value.
// If in_dL_dmean2D/in_dL_dconic are non-null, K11 is skipped and they are used as K12/K13 inputs
// (lets a test check the per-Gaussian backward bit-exactly).
void gso_backward(void* h, const float* bg, const float* means3D, const float* scales, float mod, const float* rots,
                  const float* cov3D_precomp, const float* view, const float* proj, float tan_fovx, float tan_fovy,
                  const float* dL_dpix, const float* in_dL_dmean2D, const float* in_dL_dconic,
                  const float* shs, const float* campos) {
  Ctx* c = (Ctx*)h;
  const int P = c->P, W = c->W, H = c->H, R = c->R;
  const size_t HW = (size_t)W * H;
  const float focal_y = H / (2.0f * tan_fovy), focal_x = W / (2.0f * tan_fovx);
  c->dL_dmean2D.assign((size_t)3 * P, 0.f); c->dL_dconic.assign((size_t)4 * P, 0.f);
  c->dL_dopacity.assign(P, 0.f); c->dL_dcolor.assign((size_t)3 * P, 0.f);
  c->dL_dmean3D.assign((size_t)3 * P, 0.f); c->dL_dcov3D.assign((size_t)6 * P, 0.f);
  c->dL_dscale.assign((size_t)3 * P, 0.f); c->dL_drot.assign((size_t)4 * P, 0.f);
  c->dL_dsh.assign((size_t)3 * P * std::max(c->M, 0), 0.f);

  if (in_dL_dmean2D && in_dL_dconic) {
    std::memcpy(c->dL_dmean2D.data(), in_dL_dmean2D, sizeof(float) * 3 * P);
    std::memcpy(c->dL_dconic.data(), in_dL_dconic, sizeof(float) * 4 * P);
  } else {
    // K11 renderCUDA backward (backward.cu:399-557)
    std::vector<double> slab((size_t)R * 9, 0.0);
    const int ntiles = (int)(c->gx * c->gy);
    const float ddelx_dx = 0.5 * W, ddely_dy = 0.5 * H;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < ntiles; tile++) {
      const uint32_t tx = tile % c->gx, ty = tile / c->gx;
      const u2 range = c->ranges[tile];
      const int toDo = (int)(range.y - range.x);
      for (uint32_t ly = 0; ly < BLOCK_Y; ly++)
        for (uint32_t lx = 0; lx < BLOCK_X; lx++) {
          const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
          if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
          const size_t pix_id = (size_t)W * py + px;
          const f2 pixf = { (float)px, (float)py };
          const float T_final = c->accum_alpha[pix_id];
          float T = T_final;
          uint32_t contributor = toDo;
          const int last_contributor = (int)c->n_contrib[pix_id];
          float accum_rec[NCH] = { 0 }, dL_dpixel[NCH], last_color[NCH] = { 0 };
          for (int i = 0; i < NCH; i++) dL_dpixel[i] = dL_dpix[i * HW + pix_id];
          float last_alpha = 0;
          for (int j = 0; j < toDo; j++) {
            contributor--;
            if ((int)contributor >= last_contributor) continue;
            const uint32_t s = range.y - 1 - j;
            const uint32_t id = c->vals[s];
            const f2 xy = c->means2D[id];
            const f2 d = { xy.x - pixf.x, xy.y - pixf.y };
            const f4 con_o = c->conic_opacity[id];
            const float power = -0.5f * (con_o.x * d.x * d.x + con_o.z * d.y * d.y) - con_o.y * d.x * d.y;
            if (power > 0.0f) continue;
            const float G = std::exp(power);
            const float alpha = std::min(0.99f, con_o.w * G);
            if (alpha < 1.0f / 255.0f) continue;
            T = T / (1.f - alpha);
            const float dchannel_dcolor = alpha * T;
            float dL_dalpha = 0.0f;
            double* g = &slab[(size_t)s * 9];
            for (int ch = 0; ch < NCH; ch++) {
              const float col = c->rgb[(size_t)id * NCH + ch];
              accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
              last_color[ch] = col;
              const float dL_dchannel = dL_dpixel[ch];
              dL_dalpha += (col - accum_rec[ch]) * dL_dchannel;
              g[6 + ch] += (double)(dchannel_dcolor * dL_dchannel);
            }
            dL_dalpha *= T;
            last_alpha = alpha;
            float bg_dot_dpixel = 0;
            for (int i = 0; i < NCH; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
            dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
            const float dL_dG = con_o.w * dL_dalpha;
            const float gdx = G * d.x, gdy = G * d.y;
            const float dG_ddelx = -gdx * con_o.x - gdy * con_o.y;
            const float dG_ddely = -gdy * con_o.z - gdx * con_o.y;
            g[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
            g[1] += (double)(dL_dG * dG_ddely * ddely_dy);
            g[2] += (double)(-0.5f * gdx * d.x * dL_dG);
            g[3] += (double)(-0.5f * gdx * d.y * dL_dG);
            g[4] += (double)(-0.5f * gdy * d.y * dL_dG);
            g[5] += (double)(G * dL_dalpha);
          }
        }
    }
    std::vector<double> acc((size_t)P * 9, 0.0);
    for (int s = 0; s < R; s++) {
      const uint32_t id = c->vals[s];
      for (int k = 0; k < 9; k++) acc[(size_t)id * 9 + k] += slab[(size_t)s * 9 + k];
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
      const double* a = &acc[(size_t)i * 9];
      c->dL_dmean2D[3 * i + 0] = (float)a[0]; c->dL_dmean2D[3 * i + 1] = (float)a[1];
      c->dL_dconic[4 * i + 0] = (float)a[2]; c->dL_dconic[4 * i + 1] = (float)a[3]; c->dL_dconic[4 * i + 3] = (float)a[4];
      c->dL_dopacity[i] = (float)a[5];
      c->dL_dcolor[3 * i + 0] = (float)a[6]; c->dL_dcolor[3 * i + 1] = (float)a[7]; c->dL_dcolor[3 * i + 2] = (float)a[8];
    }
  }

  // K12 computeCov2DCUDA (backward.cu:144-274)
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    if (!(c->radii[idx] > 0)) continue;
    const float* cov3D = cov3D_precomp ? cov3D_precomp + 6 * idx : &c->cov3D[(size_t)6 * idx];
    f3 mean = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
    f3 dL_dconic = { c->dL_dconic[4 * idx], c->dL_dconic[4 * idx + 1], c->dL_dconic[4 * idx + 3] };
    f3 t = transformPoint4x3(mean, view);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
    t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
    const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
    const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;
    const float h_x = focal_x, h_y = focal_y;
    mat3 J = mk(h_x / t.z, 0.0f, -(h_x * t.x) / (t.z * t.z), 0.0f, h_y / t.z, -(h_y * t.y) / (t.z * t.z), 0, 0, 0);
    mat3 Wm = mk(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    mat3 Vrk = mk(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 T = mul(Wm, J);
    mat3 cov2D = mul(mul(transpose(T), transpose(Vrk)), T);
    float a = cov2D[0][0] += 0.3f;
    float b = cov2D[0][1];
    float cc = cov2D[1][1] += 0.3f;
    float denom = a * cc - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float* dL_dcov = c->dL_dcov3D.data();
    if (denom2inv != 0) {
      dL_da = denom2inv * (-cc * cc * dL_dconic.x + 2 * b * cc * dL_dconic.y + (denom - a * cc) * dL_dconic.z);
      dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * cc) * dL_dconic.x);
      dL_db = denom2inv * 2 * (b * cc * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);
      dL_dcov[6 * idx + 0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
      dL_dcov[6 * idx + 3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
      dL_dcov[6 * idx + 5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
      dL_dcov[6 * idx + 1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
      dL_dcov[6 * idx + 2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
      dL_dcov[6 * idx + 4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
    } else {
      for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
    }
    float dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_da +
                    (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_db;
    float dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_da +
                    (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_db;
    float dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_da +
                    (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_db;
    float dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc +
                    (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_db;
    float dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc +
                    (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_db;
    float dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc +
                    (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_db;
    float dL_dJ00 = Wm[0][0] * dL_dT00 + Wm[0][1] * dL_dT01 + Wm[0][2] * dL_dT02;
    float dL_dJ02 = Wm[2][0] * dL_dT00 + Wm[2][1] * dL_dT01 + Wm[2][2] * dL_dT02;
    float dL_dJ11 = Wm[1][0] * dL_dT10 + Wm[1][1] * dL_dT11 + Wm[1][2] * dL_dT12;
    float dL_dJ12 = Wm[2][0] * dL_dT10 + Wm[2][1] * dL_dT11 + Wm[2][2] * dL_dT12;
    float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
    f3 dL_dmean = transformVec4x3Transpose({ dL_dtx, dL_dty, dL_dtz }, view);
    c->dL_dmean3D[3 * idx + 0] = dL_dmean.x; c->dL_dmean3D[3 * idx + 1] = dL_dmean.y; c->dL_dmean3D[3 * idx + 2] = dL_dmean.z;
  }

  // K13 preprocessCUDA backward (backward.cu:346-396) + computeCov3D backward (:278-341)
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    if (!(c->radii[idx] > 0)) continue;
    f3 m = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
    f4 m_hom = transformPoint4x4(m, proj);
    float m_w = 1.0f / (m_hom.w + 0.0000001f);
    const float gx2 = c->dL_dmean2D[3 * idx + 0], gy2 = c->dL_dmean2D[3 * idx + 1];
    float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    f3 dm;
    dm.x = (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
    dm.y = (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
    dm.z = (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
    c->dL_dmean3D[3 * idx + 0] += dm.x; c->dL_dmean3D[3 * idx + 1] += dm.y; c->dL_dmean3D[3 * idx + 2] += dm.z;
    if (shs) {  // backward.cu:390-391
      v3 g = computeColorFromSH_backward(idx, c->D, c->M, means3D, campos, shs, c->clamped.data(), c->dL_dcolor.data(), c->dL_dsh.data());
      c->dL_dmean3D[3 * idx + 0] += g.x; c->dL_dmean3D[3 * idx + 1] += g.y; c->dL_dmean3D[3 * idx + 2] += g.z;
    }
    if (scales) {
      f4 rot = { rots[4 * idx], rots[4 * idx + 1], rots[4 * idx + 2], rots[4 * idx + 3] };
      f3 scale = { scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2] };
      float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
      mat3 Rm = mk(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
      mat3 S = mk(1, 0, 0, 0, 1, 0, 0, 0, 1);
      f3 s = { mod * scale.x, mod * scale.y, mod * scale.z };
      S[0][0] = s.x; S[1][1] = s.y; S[2][2] = s.z;
      mat3 M = mul(S, Rm);
      const float* g = &c->dL_dcov3D[(size_t)6 * idx];
      mat3 dL_dSigma = mk(g[0], 0.5f * g[1], 0.5f * g[2], 0.5f * g[1], g[3], 0.5f * g[4], 0.5f * g[2], 0.5f * g[4], g[5]);
      mat3 dL_dM = mul(smul(2.0f, M), dL_dSigma);
      mat3 Rt = transpose(Rm);
      mat3 dL_dMt = transpose(dL_dM);
      auto dot3 = [](const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
      c->dL_dscale[3 * idx + 0] = dot3(Rt[0], dL_dMt[0]);
      c->dL_dscale[3 * idx + 1] = dot3(Rt[1], dL_dMt[1]);
      c->dL_dscale[3 * idx + 2] = dot3(Rt[2], dL_dMt[2]);
      for (int k = 0; k < 3; k++) { dL_dMt[0][k] *= s.x; dL_dMt[1][k] *= s.y; dL_dMt[2][k] *= s.z; }
      f4 dq;
      dq.x = 2 * z * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * y * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * x * (dL_dMt[1][2] - dL_dMt[2][1]);
      dq.y = 2 * y * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * z * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * r * (dL_dMt[1][2] - dL_dMt[2][1]) - 4 * x * (dL_dMt[2][2] + dL_dMt[1][1]);
      dq.z = 2 * x * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * r * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * z * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * y * (dL_dMt[2][2] + dL_dMt[0][0]);
      dq.w = 2 * r * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * x * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * y * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * z * (dL_dMt[1][1] + dL_dMt[0][0]);
      c->dL_drot[4 * idx + 0] = dq.x; c->dL_drot[4 * idx + 1] = dq.y; c->dL_drot[4 * idx + 2] = dq.z; c->dL_drot[4 * idx + 3] = dq.w;
    }
  }
}

// ---- accessors (copy state out into caller-owned arrays) ----
int gso_num_rendered(void* h) { return ((Ctx*)h)->R; }
int gso_sort_bits(void* h) { Ctx* c = (Ctx*)h; return 32 + (int)getHigherMsb(c->gx * c->gy); }
#define GSO_COPY(name, field, type) \
  void gso_get_##name(void* h, type* out) { Ctx* c = (Ctx*)h; if (!c->field.empty()) std::memcpy(out, c->field.data(), sizeof(c->field[0]) * c->field.size()); }
GSO_COPY(radii, radii, int)
GSO_COPY(depths, depths, float)
GSO_COPY(means2D, means2D, float)
GSO_COPY(conic_opacity, conic_opacity, float)
GSO_COPY(cov3D, cov3D, float)
GSO_COPY(tiles_touched, tiles_touched, uint32_t)
GSO_COPY(point_offsets, point_offsets, uint32_t)
GSO_COPY(keys_unsorted, keys_unsorted, uint64_t)
GSO_COPY(vals_unsorted, vals_unsorted, uint32_t)
GSO_COPY(keys, keys, uint64_t)
GSO_COPY(point_list, vals, uint32_t)
GSO_COPY(ranges, ranges, uint32_t)
GSO_COPY(n_contrib, n_contrib, uint32_t)
GSO_COPY(final_T, accum_alpha, float)
GSO_COPY(out_color, out_color, float)
GSO_COPY(dL_dmean2D, dL_dmean2D, float)
GSO_COPY(dL_dconic, dL_dconic, float)
GSO_COPY(dL_dopacity, dL_dopacity, float)
GSO_COPY(dL_dcolor, dL_dcolor, float)
GSO_COPY(dL_dmean3D, dL_dmean3D, float)
GSO_COPY(dL_dcov3D, dL_dcov3D, float)
GSO_COPY(dL_dscale, dL_dscale, float)
GSO_COPY(dL_drot, dL_drot, float)
GSO_COPY(dL_dsh, dL_dsh, float)
GSO_COPY(rgb, rgb, float)

// Per-pixel instability flags for tolerance tests: bit0 set if any (pixel,Gaussian) decision of the
// forward walk (power>0, alpha<1/255, test_T<1e-4) lies within `rel` of its threshold, i.e. a
// different-but-valid exp() could flip it.  Not part of the reference; test support only.
void gso_unstable_pixels(void* h, float rel, uint8_t* flags) {
  Ctx* c = (Ctx*)h;
  const int W = c->W, H = c->H;
  const int ntiles = (int)(c->gx * c->gy);
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < ntiles; tile++) {
    const uint32_t tx = tile % c->gx, ty = tile / c->gx;
    const u2 range = c->ranges[tile];
    for (uint32_t ly = 0; ly < BLOCK_Y; ly++)
      for (uint32_t lx = 0; lx < BLOCK_X; lx++) {
        const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
        const f2 pixf = { (float)px, (float)py };
        float T = 1.0f; uint8_t flag = 0;
        for (uint32_t s = range.x; s < range.y; s++) {
          const uint32_t id = c->vals[s];
          const f2 xy = c->means2D[id];
          const f2 d = { xy.x - pixf.x, xy.y - pixf.y };
          const f4 con_o = c->conic_opacity[id];
          const float power = -0.5f * (con_o.x * d.x * d.x + con_o.z * d.y * d.y) - con_o.y * d.x * d.y;
          if (std::fabs(power) < rel) flag = 1;
          if (power > 0.0f) continue;
          const float alpha = std::min(0.99f, con_o.w * std::exp(power));
          if (std::fabs(alpha - 1.0f / 255.0f) < rel * (1.0f / 255.0f)) flag = 1;
          if (alpha < 1.0f / 255.0f) continue;
          const float test_T = T * (1 - alpha);
          if (std::fabs(test_T - 0.0001f) < rel * 0.0001f) flag = 1;
          if (test_T < 0.0001f) break;
          T = test_T;
        }
        flags[(size_t)W * py + px] = flag;
      }
  }
}

} // extern "C"
