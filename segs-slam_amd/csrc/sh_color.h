// sh_color.h -- view-dependent colour from real spherical harmonics, degrees 0..3, and its backward.
//
// Off the live SEGS-SLAM path: GaussianRenderer always hands the rasterizer colors_precomp (src/gaussian_renderer.cpp:86-99),
// so this branch only runs when a caller of the C ABI passes `shs` instead (kept because the reference's entry points accept
// it: forward.cu:205,241, backward.cu:390-395).  What the reference computes there (forward.cu:20-71, backward.cu:20-139):
//     rgb = max(0, 0.5 + sum_k Y_k(dir) * sh_k),   dir = (mean - campos) / |mean - campos|
// with the 16 real SH basis functions Y_k in the usual graphics ordering and sign convention, the clamp remembered per
// channel for the backward.  Here the basis is evaluated ONCE into a table (and, for the backward, its gradient with respect to
// the direction into a second one); colour, dL/dsh and dL/ddir are then plain dot products over the active rows, and the
// normalisation is differentiated through the projector (I - d d^T) / |v|.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace segs {
namespace sh {

constexpr int MAX_COEFFS = 16;
__device__ __forceinline__ int rows_of(int degree) { return (degree + 1) * (degree + 1); }

// Normalisation constants of the real SH basis: sqrt((2l+1)/(4 pi) * (l-|m|)!/(l+|m|)!) folded with the polynomial factors.
constexpr float K0 = 0.28209479177387814f;   // l = 0
constexpr float K1 = 0.4886025119029199f;    // l = 1
constexpr float K2A = 1.0925484305920792f, K2B = 0.31539156525252005f, K2C = 0.5462742152960396f;                       // l = 2
constexpr float K3A = 0.5900435899266435f, K3B = 2.890611442640554f, K3C = 0.4570457994644658f, K3D = 0.3731763325901154f,
                K3E = 1.445305721320277f;                                                                                  // l = 3

// Y[k] for unit direction (x, y, z); rows beyond rows_of(degree) are left untouched.
__device__ __forceinline__ void basis(float x, float y, float z, int degree, float* Y) {
  Y[0] = K0;
  if (degree < 1) return;
  Y[1] = -K1 * y; Y[2] = K1 * z; Y[3] = -K1 * x;
  if (degree < 2) return;
  const float xx = x * x, yy = y * y, zz = z * z;
  Y[4] = K2A * (x * y); Y[5] = -K2A * (y * z); Y[6] = K2B * (2.f * zz - xx - yy); Y[7] = -K2A * (x * z); Y[8] = K2C * (xx - yy);
  if (degree < 3) return;
  Y[9] = -K3A * y * (3.f * xx - yy);
  Y[10] = K3B * (x * y) * z;
  Y[11] = -K3C * y * (4.f * zz - xx - yy);
  Y[12] = K3D * z * (2.f * zz - 3.f * xx - 3.f * yy);
  Y[13] = -K3C * x * (4.f * zz - xx - yy);
  Y[14] = K3E * z * (xx - yy);
  Y[15] = -K3A * x * (xx - 3.f * yy);
}

// dY[k][c] = d Y_k / d (x, y, z)_c, the polynomials differentiated as written above (dir treated as free variables).
__device__ __forceinline__ void basis_gradient(float x, float y, float z, int degree, float (*dY)[3]) {
  dY[0][0] = dY[0][1] = dY[0][2] = 0.f;
  if (degree < 1) return;
  dY[1][0] = 0.f; dY[1][1] = -K1; dY[1][2] = 0.f;
  dY[2][0] = 0.f; dY[2][1] = 0.f; dY[2][2] = K1;
  dY[3][0] = -K1; dY[3][1] = 0.f; dY[3][2] = 0.f;
  if (degree < 2) return;
  dY[4][0] = K2A * y;        dY[4][1] = K2A * x;        dY[4][2] = 0.f;
  dY[5][0] = 0.f;            dY[5][1] = -K2A * z;       dY[5][2] = -K2A * y;
  dY[6][0] = -2.f * K2B * x; dY[6][1] = -2.f * K2B * y; dY[6][2] = 4.f * K2B * z;
  dY[7][0] = -K2A * z;       dY[7][1] = 0.f;            dY[7][2] = -K2A * x;
  dY[8][0] = 2.f * K2C * x;  dY[8][1] = -2.f * K2C * y; dY[8][2] = 0.f;
  if (degree < 3) return;
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  dY[9][0] = -K3A * 6.f * xy;                    dY[9][1] = -K3A * 3.f * (xx - yy);              dY[9][2] = 0.f;
  dY[10][0] = K3B * yz;                          dY[10][1] = K3B * xz;                           dY[10][2] = K3B * xy;
  dY[11][0] = K3C * 2.f * xy;                    dY[11][1] = -K3C * (4.f * zz - xx - 3.f * yy);  dY[11][2] = -K3C * 8.f * yz;
  dY[12][0] = -K3D * 6.f * xz;                   dY[12][1] = -K3D * 6.f * yz;                    dY[12][2] = K3D * 3.f * (2.f * zz - xx - yy);
  dY[13][0] = -K3C * (4.f * zz - 3.f * xx - yy); dY[13][1] = K3C * 2.f * xy;                     dY[13][2] = -K3C * 8.f * xz;
  dY[14][0] = K3E * 2.f * xz;                    dY[14][1] = -K3E * 2.f * yz;                    dY[14][2] = K3E * (xx - yy);
  dY[15][0] = -K3A * 3.f * (xx - yy);            dY[15][1] = K3A * 6.f * xy;                     dY[15][2] = 0.f;
}

// Colour of Gaussian `idx` seen from campos; clamp_bits: bit c set if channel c was clamped at 0.
__device__ inline float3 to_rgb(int idx, int degree, int max_coeffs, float3 mean, const float* campos, const float* shs,
                                uint32_t* clamp_bits) {
  const float vx = mean.x - campos[0], vy = mean.y - campos[1], vz = mean.z - campos[2];
  const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
  float Y[MAX_COEFFS];
  basis(vx * inv, vy * inv, vz * inv, degree, Y);
  const float* c = shs + (size_t)idx * max_coeffs * 3;
  float r = 0.5f, g = 0.5f, b = 0.5f;
  const int n = rows_of(degree);
  for (int k = 0; k < n; k++) { r += Y[k] * c[3 * k]; g += Y[k] * c[3 * k + 1]; b += Y[k] * c[3 * k + 2]; }
  *clamp_bits = (r < 0.f ? 1u : 0u) | (g < 0.f ? 2u : 0u) | (b < 0.f ? 4u : 0u);
  return make_float3(fmaxf(r, 0.f), fmaxf(g, 0.f), fmaxf(b, 0.f));
}

// Backward: writes rows 0..rows_of(degree)-1 of dL_dsh for this Gaussian, returns dL/dmean (through the view direction).
__device__ inline float3 backward(int idx, int degree, int max_coeffs, float3 mean, const float* campos, const float* shs,
                                  uint32_t clamp_bits, float3 dL_drgb, float* dL_dshs) {
  const float vx = mean.x - campos[0], vy = mean.y - campos[1], vz = mean.z - campos[2];
  const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
  const float x = vx * inv, y = vy * inv, z = vz * inv;
  // a clamped channel passes no gradient (forward.cu:66-70)
  const float gr = (clamp_bits & 1u) ? 0.f : dL_drgb.x, gg = (clamp_bits & 2u) ? 0.f : dL_drgb.y, gb = (clamp_bits & 4u) ? 0.f : dL_drgb.z;
  float Y[MAX_COEFFS], dY[MAX_COEFFS][3];
  basis(x, y, z, degree, Y);
  basis_gradient(x, y, z, degree, dY);
  const float* c = shs + (size_t)idx * max_coeffs * 3;
  float* out = dL_dshs + (size_t)idx * max_coeffs * 3;
  float dx = 0.f, dy = 0.f, dz = 0.f;   // dL / d(unit direction)
  const int n = rows_of(degree);
  for (int k = 0; k < n; k++) {
    out[3 * k] = Y[k] * gr; out[3 * k + 1] = Y[k] * gg; out[3 * k + 2] = Y[k] * gb;
    const float w = c[3 * k] * gr + c[3 * k + 1] * gg + c[3 * k + 2] * gb;   // dL/dY_k
    dx += dY[k][0] * w; dy += dY[k][1] * w; dz += dY[k][2] * w;
  }
  // d = v / |v|  =>  dL/dv = (g - d (d . g)) / |v|
  const float dg = x * dx + y * dy + z * dz;
  return make_float3((dx - x * dg) * inv, (dy - y * dg) * inv, (dz - z * dg) * inv);
}

}  // namespace sh
}  // namespace segs
