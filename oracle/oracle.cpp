// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Scalar CPU restatement of glaze's render path (reference: davidepi/glaze v0.3.0).  The reference
// has no CPU tracer: BVH build/traversal/ray-triangle live in the Vulkan driver and all shading is
// GLSL, and it can be neither compiled nor run here (no rustc/cargo/glslc/Vulkan,
// SURVEY F3).  This file therefore restates the GLSL shaders and the Rust host logic line by line;
// every function cites the reference file:line it follows.
//
// PARITY UNPINNED: no reference test pins a ray hit, a BSDF value or a pixel.  What
// IS pinned by reference KATs and checked in tests/: Spectrum::from_rgb/to_xyz/luminance
// (spectrum.rs:762-794), ColorXYZ<->ColorRGB (color.rs:329-345), fovx->fovy (camera.rs:296-307) and
// the .glaze byte format via resources/mattest.glaze (oracle/glaze_v1.py).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// The product (glaze_amd/csrc) never links, includes or calls anything in oracle/.
//
// Shared with the product on purpose (interface / data / math spec, not implementation):
//   include/glaze_abi.h   -- POD scene description types
//   include/glz_tables.h  -- numeric tables (metal n,k; CIE bins; Smits bases)
//   include/glz_detmath.h -- deterministic sin/cos/acos/atan2 (see its header)
//
// Build: g++ -O2 -ffp-contract=off -fno-fast-math -shared -fPIC  (oracle/Makefile)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "glaze_abi.h"
#include "glz_detmath.h"
#include "glz_tables.h"

namespace {

// ------------------------------------------------------------------------------------------
// GLSL built-ins with the exact definitions of the GLSL 4.60 spec ([ext])
// ------------------------------------------------------------------------------------------
const float PI = 3.1415926f, INV_PI = 0.3183099f, TWO_PI = 6.2831853f;   // constants.glsl:4-6
const float DEFAULT_IOR = 1.000293f, INV_2PI = 0.1591549f;               // constants.glsl:7, :9
const float INF = INFINITY;

inline float gmin(float x, float y) { return y < x ? y : x; }
inline float gmax(float x, float y) { return x < y ? y : x; }
inline float gstep(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
inline float gsign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
inline float gmix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline float gclampf(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
inline float checknan(float x) { return isnan(x) ? 0.0f : x; }   // raytrace_commons.glsl:7
inline float checkinf(float x) { return isinf(x) ? 0.0f : x; }   // raytrace_commons.glsl:8

struct V3 {
  float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// normalize(v) = v * (1/sqrt(dot(v,v)))  -- the build's definition of the GLSL built-in ([ext])
inline V3 normalize(V3 a) {
  float inv = 1.0f / sqrtf(dot(a, a));
  return v3(a.x * inv, a.y * inv, a.z * inv);
}
inline V3 reflect(V3 I, V3 N) { return I - (2.0f * dot(N, I)) * N; }
inline V3 refract(V3 I, V3 N, float eta) {
  float d = dot(N, I);
  float k = 1.0f - eta * eta * (1.0f - d * d);
  if (k < 0.0f) return v3(0, 0, 0);
  return eta * I - (eta * d + sqrtf(k)) * N;
}

// mat4 (column-major) * (x,y,z,w): r = c0*x + c1*y + c2*z + c3*w, evaluated left to right
inline V3 mat_point(const float* m, V3 p) {   // w = 1
  return v3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
inline V3 mat_dir(const float* m, V3 p) {     // w = 0
  return v3(m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z,
            m[2] * p.x + m[6] * p.y + m[10] * p.z);
}
// transpose(upper 3x3 of m) * n  ==  (n^T * M3)^T
inline V3 mat_tdir(const float* m, V3 n) {
  return v3(m[0] * n.x + m[1] * n.y + m[2] * n.z, m[4] * n.x + m[5] * n.y + m[6] * n.z,
            m[8] * n.x + m[9] * n.y + m[10] * n.z);
}

// ------------------------------------------------------------------------------------------
// Spectrum, device flavour (spectrum.glsl): 16 bins as 4 x vec4; bin i lives in col[i/4][i%4]
// ------------------------------------------------------------------------------------------
struct Sp {
  float w[16];
};
inline Sp sp_uniform(float f) { Sp s; for (int i = 0; i < 16; ++i) s.w[i] = f; return s; }
inline Sp sp_add(const Sp& a, const Sp& b) { Sp r; for (int i = 0; i < 16; ++i) r.w[i] = a.w[i] + b.w[i]; return r; }
inline Sp sp_add(const Sp& a, float f) { Sp r; for (int i = 0; i < 16; ++i) r.w[i] = a.w[i] + f; return r; }
inline Sp sp_sub(const Sp& a, const Sp& b) { Sp r; for (int i = 0; i < 16; ++i) r.w[i] = a.w[i] - b.w[i]; return r; }
inline Sp sp_mul(const Sp& a, const Sp& b) { Sp r; for (int i = 0; i < 16; ++i) r.w[i] = a.w[i] * b.w[i]; return r; }
inline Sp sp_mul(const Sp& a, float f) { Sp r; for (int i = 0; i < 16; ++i) r.w[i] = a.w[i] * f; return r; }
inline Sp sp_div(const Sp& a, const Sp& b) { Sp r; for (int i = 0; i < 16; ++i) r.w[i] = a.w[i] / b.w[i]; return r; }
inline Sp sp_div(const Sp& a, float f) { Sp r; for (int i = 0; i < 16; ++i) r.w[i] = a.w[i] / f; return r; }
inline Sp sp_mix(const Sp& a, const Sp& b, float t) { Sp r; for (int i = 0; i < 16; ++i) r.w[i] = gmix(a.w[i], b.w[i], t); return r; }

const float DEV_X[16] = GLZ_DEV_CIE_X, DEV_Y[16] = GLZ_DEV_CIE_Y, DEV_Z[16] = GLZ_DEV_CIE_Z;

// sum of "sp.col0*c0 + sp.col1*c1 + sp.col2*c2 + sp.col3*c3" then ".x+.y+.z+.w" (spectrum.glsl:45-46, :65-70)
inline float weighted_sum(const Sp& sp, const float* c) {
  float lane[4];
  for (int j = 0; j < 4; ++j)
    lane[j] = ((sp.w[j] * c[j] + sp.w[4 + j] * c[4 + j]) + sp.w[8 + j] * c[8 + j]) + sp.w[12 + j] * c[12 + j];
  return ((lane[0] + lane[1]) + lane[2]) + lane[3];
}
inline float sp_luminance(const Sp& sp) { return weighted_sum(sp, DEV_Y) * 0.17557178f; }   // spectrum.glsl:39-48
inline V3 sp_xyz(const Sp& sp) {                                                              // spectrum.glsl:50-72
  return v3(weighted_sum(sp, DEV_X) * 0.17557178f, weighted_sum(sp, DEV_Y) * 0.17557178f, weighted_sum(sp, DEV_Z) * 0.17557178f);
}
inline V3 xyz_to_rgb(V3 c) {                                                                  // spectrum.glsl:74-81
  V3 r;
  r.x = (3.240479f * c.x - 1.537150f * c.y) - 0.498535f * c.z;
  r.y = (-0.969256f * c.x + 1.875991f * c.y) + 0.041556f * c.z;
  r.z = (0.055648f * c.x - 0.204043f * c.y) + 1.057311f * c.z;
  return r;
}
inline V3 sp_rgb(const Sp& sp) { return xyz_to_rgb(sp_xyz(sp)); }                             // spectrum.glsl:83-86

struct Bases { Sp white, cyan, magenta, yellow, red, green, blue; };
const Bases SURF = {{GLZ_DEV_SURF_WHITE}, {GLZ_DEV_SURF_CYAN}, {GLZ_DEV_SURF_MAGENTA}, {GLZ_DEV_SURF_YELLOW},
                    {GLZ_DEV_SURF_RED}, {GLZ_DEV_SURF_GREEN}, {GLZ_DEV_SURF_BLUE}};
const Bases ILLUM = {{GLZ_DEV_ILLUM_WHITE}, {GLZ_DEV_ILLUM_CYAN}, {GLZ_DEV_ILLUM_MAGENTA}, {GLZ_DEV_ILLUM_YELLOW},
                     {GLZ_DEV_ILLUM_RED}, {GLZ_DEV_ILLUM_GREEN}, {GLZ_DEV_ILLUM_BLUE}};

// GENERATE_COLOR_TO_SPECTRUM, spectrum.glsl:158-200 (no clamping on the device, Q10)
inline Sp color_to_spectrum(V3 rgb, const Bases& B) {
  Sp res;
  if (rgb.x <= rgb.y && rgb.x <= rgb.z) {
    res = sp_mul(B.white, rgb.x);
    if (rgb.y <= rgb.z) {
      res = sp_add(res, sp_mul(B.cyan, rgb.y - rgb.x));
      res = sp_add(res, sp_mul(B.blue, rgb.z - rgb.y));
    } else {
      res = sp_add(res, sp_mul(B.cyan, rgb.z - rgb.x));
      res = sp_add(res, sp_mul(B.green, rgb.y - rgb.z));
    }
  } else if (rgb.y <= rgb.x && rgb.y <= rgb.z) {
    res = sp_mul(B.white, rgb.y);
    if (rgb.x <= rgb.z) {
      res = sp_add(res, sp_mul(B.magenta, rgb.x - rgb.y));
      res = sp_add(res, sp_mul(B.blue, rgb.z - rgb.x));
    } else {
      res = sp_add(res, sp_mul(B.magenta, rgb.z - rgb.y));
      res = sp_add(res, sp_mul(B.red, rgb.x - rgb.z));
    }
  } else {
    res = sp_mul(B.white, rgb.z);
    if (rgb.x <= rgb.y) {
      res = sp_add(res, sp_mul(B.yellow, rgb.x - rgb.z));
      res = sp_add(res, sp_mul(B.green, rgb.y - rgb.x));
    } else {
      res = sp_add(res, sp_mul(B.yellow, rgb.y - rgb.z));
      res = sp_add(res, sp_mul(B.red, rgb.x - rgb.y));
    }
  }
  return res;
}
inline Sp from_surface_color(V3 rgb) { return sp_mul(color_to_spectrum(rgb, SURF), 0.94f); }        // spectrum.glsl:202-242
inline Sp from_illuminant_color(V3 rgb) { return sp_mul(color_to_spectrum(rgb, ILLUM), 0.86445f); } // spectrum.glsl:244-284

// ------------------------------------------------------------------------------------------
// Host-side colour math (geometry/spectrum.rs, geometry/color.rs) -- pinned by reference KATs
// ------------------------------------------------------------------------------------------
struct HostBases { const float *white, *cyan, *magenta, *yellow, *red, *green, *blue; };
const HostBases H_SURF = {GLZ_HOST_SPECTRUM_WHITE, GLZ_HOST_SPECTRUM_CYAN, GLZ_HOST_SPECTRUM_MAGENTA, GLZ_HOST_SPECTRUM_YELLOW,
                          GLZ_HOST_SPECTRUM_RED, GLZ_HOST_SPECTRUM_GREEN, GLZ_HOST_SPECTRUM_BLUE};
const HostBases H_ILLUM = {GLZ_HOST_SPECTRUM_WHITEL, GLZ_HOST_SPECTRUM_CYANL, GLZ_HOST_SPECTRUM_MAGENTAL, GLZ_HOST_SPECTRUM_YELLOWL,
                           GLZ_HOST_SPECTRUM_REDL, GLZ_HOST_SPECTRUM_GREENL, GLZ_HOST_SPECTRUM_BLUEL};

// Spectrum::from_rgb (spectrum.rs:82-141): `res += sp[k] * c` per term, final scale, clamp to [0,1]
Sp host_from_rgb(float r, float g, float b, bool is_light) {
  const HostBases& B = is_light ? H_ILLUM : H_SURF;
  Sp res = sp_uniform(0.0f);
  auto acc = [&](const float* basis, float c) { for (int i = 0; i < 16; ++i) res.w[i] += basis[i] * c; };
  if (r <= g && r <= b) {
    acc(B.white, r);
    if (g <= b) { acc(B.cyan, g - r); acc(B.blue, b - g); } else { acc(B.cyan, b - r); acc(B.green, g - b); }
  } else if (g <= r && g <= b) {
    acc(B.white, g);
    if (r <= b) { acc(B.magenta, r - g); acc(B.blue, b - r); } else { acc(B.magenta, b - g); acc(B.red, r - b); }
  } else {
    acc(B.white, b);
    if (r <= g) { acc(B.yellow, r - b); acc(B.green, g - r); } else { acc(B.yellow, g - b); acc(B.red, r - g); }
  }
  float scale = is_light ? 0.86445f : 0.94f;
  for (int i = 0; i < 16; ++i) {
    res.w[i] *= scale;
    res.w[i] = res.w[i] < 0.0f ? 0.0f : (res.w[i] > 1.0f ? 1.0f : res.w[i]);   // f32::clamp
  }
  return res;
}
const float INVY_SUM = 0.17557178f;                                             // spectrum.rs:215
// Spectrum::to_xyz (spectrum.rs:144-162)
void host_to_xyz(const Sp& s, float out[3]) {
  float x = 0, y = 0, z = 0;
  for (int i = 0; i < 16; ++i) {
    x += s.w[i] * GLZ_HOST_X[i];
    y += s.w[i] * GLZ_HOST_Y[i];
    z += s.w[i] * GLZ_HOST_Z[i];
  }
  x *= 100.0f * INVY_SUM;
  y *= 100.0f * INVY_SUM;
  z *= 100.0f * INVY_SUM;
  out[0] = fmaxf(x, 0.0f); out[1] = fmaxf(y, 0.0f); out[2] = fmaxf(z, 0.0f);
}
// Spectrum::luminance (spectrum.rs:168-174)
float host_luminance(const Sp& s) {
  float y = 0;
  for (int i = 0; i < 16; ++i) y += s.w[i] * GLZ_HOST_Y[i];
  y *= INVY_SUM;
  return y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
}
// From<ColorXYZ> for ColorRGB (color.rs:104-137)
void host_xyz_to_rgb(const float c[3], float out[3]) {
  const float EXP = 1.0f / 2.4f, INV_100 = 1.0f / 100.0f;
  float x = c[0] * INV_100, y = c[1] * INV_100, z = c[2] * INV_100;
  float r = x * 3.2404542f + y * -1.5371385f + z * -0.4985314f;
  float g = x * -0.969266f + y * 1.8760108f + z * 0.0415560f;
  float b = x * 0.0556434f + y * -0.2040259f + z * 1.0572252f;
  r = r > 0.0031308f ? 1.055f * powf(r, EXP) - 0.055f : r * 12.92f;
  g = g > 0.0031308f ? 1.055f * powf(g, EXP) - 0.055f : g * 12.92f;
  b = b > 0.0031308f ? 1.055f * powf(b, EXP) - 0.055f : b * 12.92f;
  out[0] = fmaxf(r, 0.0f); out[1] = fmaxf(g, 0.0f); out[2] = fmaxf(b, 0.0f);
}
// From<ColorRGB> for ColorXYZ (color.rs:140-168)
void host_rgb_to_xyz(const float c[3], float out[3]) {
  const float INV = 1.0f / 12.92f;
  float v[3];
  for (int i = 0; i < 3; ++i) v[i] = (c[i] > 0.04045f ? powf((c[i] + 0.055f) / 1.055f, 2.4f) : c[i] * INV) * 100.0f;
  out[0] = fmaxf(v[0] * 0.4124564f + v[1] * 0.3575761f + v[2] * 0.1804375f, 0.0f);
  out[1] = fmaxf(v[0] * 0.2126729f + v[1] * 0.7151522f + v[2] * 0.0721750f, 0.0f);
  out[2] = fmaxf(v[0] * 0.0193339f + v[1] * 0.119192f + v[2] * 0.9503041f, 0.0f);
}
// Spectrum::from_blackbody (spectrum.rs:44-72)
Sp host_from_blackbody(float temperature) {
  if (temperature <= 0.0f) return sp_uniform(0.0f);
  const float PLANCK_H = 6.62606957e-34f, BOLTZMANN_K = 1.38064852e-23f, C = 299792458.0f;
  float cur = 400.0f * 1e-9f, maxval = -3.40282347e+38f;
  Sp s;
  for (int i = 0; i < 16; ++i) {
    float c5 = cur * cur * cur * cur * cur;   // powi(5)
    float first = 2.0f * PLANCK_H * C * C / c5;
    float expo = PLANCK_H * C / (cur * temperature * BOLTZMANN_K);
    s.w[i] = first * 1.0f / expm1f(expo);
    maxval = fmaxf(s.w[i], maxval);
    cur += 20.0f * 1e-9f;
  }
  float inv = 1.0f / maxval;
  for (int i = 0; i < 16; ++i) {
    s.w[i] *= inv;
    s.w[i] = s.w[i] < 0.0f ? 0.0f : (s.w[i] > 1.0f ? 1.0f : s.w[i]);
  }
  return s;
}

// ------------------------------------------------------------------------------------------
// 4x4 matrices in double for the host-side camera / sky math (cgmath 0.18 [ext])
// ------------------------------------------------------------------------------------------
struct M4 {
  double m[16];  // column-major: m[c*4+r]
};
M4 m4_identity() { M4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1; return r; }
M4 m4_mul(const M4& a, const M4& b) {
  M4 r{};
  for (int c = 0; c < 4; ++c)
    for (int rr = 0; rr < 4; ++rr) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a.m[k * 4 + rr] * b.m[c * 4 + k];
      r.m[c * 4 + rr] = s;
    }
  return r;
}
// General inverse by cofactors (the classic 16-cofactor expansion), in double.
bool m4_invert(const M4& a, M4& out) {
  const double* m = a.m;
  double inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (det == 0) return false;
  det = 1.0 / det;
  for (int i = 0; i < 16; ++i) out.m[i] = inv[i] * det;
  return true;
}
void m4_to_f32(const M4& a, float* out) { for (int i = 0; i < 16; ++i) out[i] = (float)a.m[i]; }
M4 m4_from_f32(const float* f) { M4 r; for (int i = 0; i < 16; ++i) r.m[i] = f[i]; return r; }

// PerspectiveCam::fovy (camera.rs:24-28), in f32 like the reference (KAT camera.rs:296-307)
float cam_fovy(float fovx, float aspect_ratio) { return 2.0f * atanf(tanf(fovx * 0.5f) / aspect_ratio); }

// build_push_constants (raytracer.rs:1098-1120): camera2world = look_at_rh^-1,
// screen2camera = (projection with [1][1] *= -1)^-1.  cgmath formulas [ext]; f32 inputs, f64 math.
void push_constants(const glz_camera& cam, uint32_t width, uint32_t height, float out[32]) {
  // Matrix4::look_at_rh(eye, center, up) = look_to_rh(eye, center-eye, up)
  double ex = cam.position[0], ey = cam.position[1], ez = cam.position[2];
  double fx = cam.target[0] - ex, fy = cam.target[1] - ey, fz = cam.target[2] - ez;
  double fl = sqrt(fx * fx + fy * fy + fz * fz);
  fx /= fl; fy /= fl; fz /= fl;
  double ux = cam.up[0], uy = cam.up[1], uz = cam.up[2];
  double sx = fy * uz - fz * uy, sy = fz * ux - fx * uz, sz = fx * uy - fy * ux;
  double sl = sqrt(sx * sx + sy * sy + sz * sz);
  sx /= sl; sy /= sl; sz /= sl;
  double vx = sy * fz - sz * fy, vy = sz * fx - sx * fz, vz = sx * fy - sy * fx;
  M4 view{};
  view.m[0] = sx; view.m[1] = vx; view.m[2] = -fx; view.m[3] = 0;
  view.m[4] = sy; view.m[5] = vy; view.m[6] = -fy; view.m[7] = 0;
  view.m[8] = sz; view.m[9] = vz; view.m[10] = -fz; view.m[11] = 0;
  view.m[12] = -(ex * sx + ey * sy + ez * sz);
  view.m[13] = -(ex * vx + ey * vy + ez * vz);
  view.m[14] = (ex * fx + ey * fy + ez * fz);
  view.m[15] = 1;
  M4 proj{};
  if (cam.type == GLZ_CAMERA_PERSPECTIVE) {   // cgmath::perspective(fovy, aspect, near, far)
    float ar = (float)width / (float)height;
    double fovy = cam_fovy(cam.fovx_or_scale, ar);
    double f = 1.0 / tan(fovy / 2.0);
    double n = cam.near_plane, fa = cam.far_plane;
    proj.m[0] = f / (double)ar;
    proj.m[5] = f;
    proj.m[10] = (fa + n) / (n - fa);
    proj.m[11] = -1;
    proj.m[14] = (2.0 * fa * n) / (n - fa);
  } else {                                     // cgmath::ortho(-s, s, -s, s, -far, far)  (camera.rs:136-139)
    double s = cam.fovx_or_scale, n = -(double)cam.far_plane, fa = cam.far_plane;
    proj.m[0] = 2.0 / (s - -s);
    proj.m[5] = 2.0 / (s - -s);
    proj.m[10] = -2.0 / (fa - n);
    proj.m[12] = -(s + -s) / (s - -s);
    proj.m[13] = -(s + -s) / (s - -s);
    proj.m[14] = -(fa + n) / (fa - n);
    proj.m[15] = 1;
  }
  proj.m[5] *= -1.0;                           // raytracer.rs:1102
  M4 vi, pi;
  if (!m4_invert(view, vi)) vi = m4_identity();
  if (!m4_invert(proj, pi)) pi = m4_identity();
  m4_to_f32(vi, out);
  m4_to_f32(pi, out + 16);
}

// Light::rotation_matrix (light.rs:195-199): R_y(yaw) * R_z(pitch) * R_x(roll), cgmath [ext]
M4 sky_rotation(float yaw_deg, float pitch_deg, float roll_deg) {
  auto rad = [](float d) { return (double)d * (3.14159265358979323846 / 180.0); };
  double cy = cos(rad(yaw_deg)), sy = sin(rad(yaw_deg));
  double cz = cos(rad(pitch_deg)), sz = sin(rad(pitch_deg));
  double cx = cos(rad(roll_deg)), sx = sin(rad(roll_deg));
  M4 ry = m4_identity(), rz = m4_identity(), rx = m4_identity();
  ry.m[0] = cy; ry.m[2] = -sy; ry.m[8] = sy; ry.m[10] = cy;     // from_angle_y
  rz.m[0] = cz; rz.m[1] = sz; rz.m[4] = -sz; rz.m[5] = cz;      // from_angle_z
  rx.m[5] = cx; rx.m[6] = sx; rx.m[9] = -sx; rx.m[10] = cx;     // from_angle_x
  return m4_mul(m4_mul(ry, rz), rx);
}

// ------------------------------------------------------------------------------------------
// Device-layout structs (raytrace_structures.rs) as the oracle keeps them
// ------------------------------------------------------------------------------------------
struct RTInstance { uint32_t index_offset, index_count, material_id, transform_id; };
struct RTMaterial {
  float diffuse_mul[4], emissive_col[4];
  Sp metal_ior, metal_fresnel;
  uint32_t diffuse, roughness, metalness, opacity, normal, bsdf_index;
  float roughness_mul, metalness_mul, anisotropy, ior_dielectric;
  uint32_t is_specular, is_emissive;
};
static_assert(sizeof(RTMaterial) == 208, "RTMaterial layout");
struct RTLight {
  Sp color;
  float pos[4], dir[4];
  uint32_t shader, instance_id;
  float intensity;
  uint32_t delta;
};
static_assert(sizeof(RTLight) == 112, "RTLight layout");

struct Tex {
  uint32_t format, w, h;
  std::vector<uint8_t> px;
  // mip levels 1.. (build_mip_chain below), only used by the opt-in texture level of detail
  std::vector<Tex> mips;
};

struct Tri {   // world-space triangle for intersection
  V3 v0, v1, v2;
  uint32_t world_id, instance, prim;
  bool non_opaque;
};

struct BNode {
  float lo[3], hi[3];
  int left, right;     // children (internal) ...
  int first, count;    // ... or triangle range (leaf: count > 0)
};

struct Scene {
  std::vector<glz_vertex> vertices;
  std::vector<uint32_t> indices;
  std::vector<glz_mesh> meshes;
  std::vector<glz_transform> transforms;
  std::vector<M4> w2o_d;
  std::vector<std::vector<float>> w2o;   // inverse transforms as f32[16]
  std::vector<glz_mesh_instance> mesh_instances;
  std::vector<glz_material> materials;
  std::vector<glz_light> lights;         // after reorder_lights
  std::vector<Tex> textures;
  glz_camera camera;
  glz_meta meta;
  // derived
  std::vector<RTInstance> instances;
  std::vector<RTMaterial> rt_materials;
  std::vector<RTLight> rt_lights;
  uint32_t lights_no = 0;
  std::vector<float> derivatives;        // 12 floats per object triangle
  float srgb_lut[256];
  // sky
  float sky_obj2world[16], sky_world2obj[16];
  uint32_t sky_tex_id = 0;
  float sky_intensity = 0;
  uint32_t marginal_cdf_count = 0, conditional_integral_offset = 0, conditional_cdf_count = 0;
  float marginal_integral = 0;
  std::vector<float> marginal;           // cdf | values | conditional integrals
  std::vector<float> cond_values, cond_cdf;
  uint32_t sky_w = 0, sky_h = 0;
  // acceleration
  std::vector<Tri> tris;                 // in BVH order
  std::vector<BNode> nodes;
  // optional externally supplied BVH4 (the product's tree) for work counting
  std::vector<uint32_t> ext_nodes;       // 16 words per node (BvhNode4 layout)
  std::vector<float> ext_tris;           // 12 floats per tri (BvhTri layout)
  float ext_grid[9] = {0};               // BvhGrid: lo[3], cell[3], inv_cell[3]
};

// ------------------------------------------------------------------------------------------
// Textures: bilinear, REPEAT, level 0, sRGB decode (Vulkan sampler semantics [ext])
// ------------------------------------------------------------------------------------------
struct V4 { float x, y, z, w; };
inline V4 texel(const Scene& sc, const Tex& t, int x, int y) {
  if (t.format == GLZ_TEX_GRAY) {
    float g = (float)t.px[(size_t)y * t.w + x] / 255.0f;
    return V4{g, 0.0f, 0.0f, 1.0f};
  }
  const uint8_t* p = &t.px[((size_t)y * t.w + x) * 4];
  if (t.format == GLZ_TEX_RGBA_SRGB) return V4{sc.srgb_lut[p[0]], sc.srgb_lut[p[1]], sc.srgb_lut[p[2]], (float)p[3] / 255.0f};
  return V4{(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f};
}
inline int wrapi(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }
inline float lerp1(float a, float b, float t) { return a + (b - a) * t; }
V4 texture_bilinear_level(const Scene& sc, const Tex& t, float u, float v);
V4 texture_bilinear(const Scene& sc, uint32_t id, float u, float v) { return texture_bilinear_level(sc, sc.textures[id], u, v); }
V4 texture_bilinear_level(const Scene& sc, const Tex& t, float u, float v) {
  float fu = u * (float)t.w - 0.5f, fv = v * (float)t.h - 0.5f;
  float iu = glz_floorf(fu), iv = glz_floorf(fv);
  float ax = fu - iu, ay = fv - iv;
  int x0 = wrapi((int)iu, (int)t.w), y0 = wrapi((int)iv, (int)t.h);
  int x1 = wrapi((int)iu + 1, (int)t.w), y1 = wrapi((int)iv + 1, (int)t.h);
  V4 a = texel(sc, t, x0, y0), b = texel(sc, t, x1, y0), c = texel(sc, t, x0, y1), d = texel(sc, t, x1, y1);
  V4 r;
  r.x = lerp1(lerp1(a.x, b.x, ax), lerp1(c.x, d.x, ax), ay);
  r.y = lerp1(lerp1(a.y, b.y, ax), lerp1(c.y, d.y, ax), ay);
  r.z = lerp1(lerp1(a.z, b.z, ax), lerp1(c.z, d.z, ax), ay);
  r.w = lerp1(lerp1(a.w, b.w, ax), lerp1(c.w, d.w, ax), ay);
  return r;
}

// ------------------------------------------------------------------------------------------
// Texture level of detail (build-defined and opt-in: the reference's ray-tracing stages sample level 0 [ext]).
// Mip chain as load_texture_to_gpu builds it when the texture brings none (scene.rs:1012-1263): level l is
// max(1, w >> l) x max(1, h >> l), 1 + floor(log2(max(w, h))) levels, each from the one before by vkCmdBlitImage(LINEAR) --
// stated as: destination texel centre (x + 0.5) * (sw / dw) - 0.5 in the source, bilinear blend of the four texels around it
// with clamp-to-edge, in linear light for the colour channels of sRGB textures, rounded to the nearest code (sRGB codes by
// the threshold rule of to_srgb8 below).  Double precision, the operations in exactly this order.
// ------------------------------------------------------------------------------------------
uint8_t to_srgb8(float c);
void build_mip_chain(Tex& t) {
  t.mips.clear();
  uint32_t levels = 1;
  for (uint32_t m = std::max(t.w, t.h); m > 1; m >>= 1) ++levels;
  double eotf[256];
  for (int i = 0; i < 256; ++i) {
    double c = i / 255.0;
    eotf[i] = c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4);
  }
  const uint32_t bpp = t.format == GLZ_TEX_GRAY ? 1u : 4u;
  const Tex* src = &t;
  t.mips.reserve(levels);
  for (uint32_t l = 1; l < levels; ++l) {
    Tex d;
    d.format = t.format;
    d.w = std::max(1u, t.w >> l);
    d.h = std::max(1u, t.h >> l);
    d.px.assign((size_t)d.w * d.h * bpp, 0);
    const double rx = (double)src->w / (double)d.w, ry = (double)src->h / (double)d.h;
    for (uint32_t y = 0; y < d.h; ++y) {
      double sy = ((double)y + 0.5) * ry - 0.5, fy0 = floor(sy), fy = sy - fy0;
      long y0 = std::min<long>(std::max<long>((long)fy0, 0), (long)src->h - 1), y1 = std::min<long>(std::max<long>((long)fy0 + 1, 0), (long)src->h - 1);
      for (uint32_t x = 0; x < d.w; ++x) {
        double sx = ((double)x + 0.5) * rx - 0.5, fx0 = floor(sx), fx = sx - fx0;
        long x0 = std::min<long>(std::max<long>((long)fx0, 0), (long)src->w - 1), x1 = std::min<long>(std::max<long>((long)fx0 + 1, 0), (long)src->w - 1);
        for (uint32_t ch = 0; ch < bpp; ++ch) {
          bool srgb = t.format == GLZ_TEX_RGBA_SRGB && ch < 3;
          auto tx = [&](long ax, long ay) { uint8_t v = src->px[((size_t)ay * src->w + (size_t)ax) * bpp + ch]; return srgb ? eotf[v] : (double)v / 255.0; };
          double a = tx(x0, y0), b = tx(x1, y0), c = tx(x0, y1), e = tx(x1, y1);
          double v = (a * (1.0 - fx) + b * fx) * (1.0 - fy) + (c * (1.0 - fx) + e * fx) * fy;
          uint8_t q;
          if (srgb) q = to_srgb8((float)v);
          else { int r = (int)(v * 255.0 + 0.5); q = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); }
          d.px[((size_t)y * d.w + x) * bpp + ch] = q;
        }
      }
    }
    t.mips.push_back(std::move(d));
    src = &t.mips.back();
  }
}
// LINEAR mip filtering: the two nearest levels blended (sampler of scene.rs:716-749).  fp.lod_base = the texture-independent part of
// the ray-cone level (raygen below); NO_LOD or a level <= 0 is texture_bilinear().  fp.taps > 1 (anisotropic footprint, lod mode 2):
// that many trilinear probes spread evenly over the footprint's long axis (du, dv) around (u, v), averaged -- the scheme of the
// reference sampler's anisotropy (scene.rs:716-749 enables it at the device's maximum; only the raster viewer uses that sampler with
// derivatives, the ray-tracing stages sample level 0).
constexpr float NO_LOD = -1e30f;
struct TexFootprint { float lod_base = NO_LOD; float du = 0.0f, dv = 0.0f; uint32_t taps = 1; };
V4 texture_lod(const Scene& sc, uint32_t id, float u, float v, const TexFootprint& fp) {
  const Tex& t = sc.textures[id];
  if (!(fp.lod_base > -1e29f) || t.mips.empty()) return texture_bilinear_level(sc, t, u, v);
  float lam = fp.lod_base + 0.5f * glz_log2f((float)t.w * (float)t.h);
  lam = lam > 0.0f ? lam : 0.0f;
  float top = (float)t.mips.size();
  lam = lam < top ? lam : top;
  float fl = glz_floorf(lam);
  uint32_t l0 = (uint32_t)fl;
  float frac = lam - fl;
  V4 sum{0.0f, 0.0f, 0.0f, 0.0f};
  for (uint32_t k = 0; k < fp.taps; ++k) {
    float uu = u, vv = v;
    if (fp.taps > 1) {
      float s = ((float)k + 0.5f) / (float)fp.taps - 0.5f;
      uu = u + s * fp.du;
      vv = v + s * fp.dv;
    }
    V4 a = texture_bilinear_level(sc, l0 == 0 ? t : t.mips[l0 - 1], uu, vv);
    if (frac > 0.0f) {
      V4 b = texture_bilinear_level(sc, t.mips[l0], uu, vv);
      a = V4{lerp1(a.x, b.x, frac), lerp1(a.y, b.y, frac), lerp1(a.z, b.z, frac), lerp1(a.w, b.w, frac)};
    }
    sum = k == 0 ? a : V4{sum.x + a.x, sum.y + a.y, sum.z + a.z, sum.w + a.w};
  }
  if (fp.taps > 1) {
    float n = (float)fp.taps;
    sum = V4{sum.x / n, sum.y / n, sum.z / n, sum.w / n};
  }
  return sum;
}

// ------------------------------------------------------------------------------------------
// Scene preparation (vulkan/scene.rs RayTraceScene::new and helpers)
// ------------------------------------------------------------------------------------------
uint32_t sbt_index(uint8_t mtype) {   // material.rs:244-258
  switch (mtype) {
    case GLZ_MAT_FLAT: case GLZ_MAT_LAMBERT: return 4;
    case GLZ_MAT_MIRROR: return 6;
    case GLZ_MAT_GLASS: return 8;
    case GLZ_MAT_METAL: return 10;
    case GLZ_MAT_FROSTED: return 12;
    default: return 14;
  }
}

// load_raytrace_materials_to_gpu (scene.rs:1821-1860), col_int_to_f32 (:1930-1937)
void build_rt_materials(Scene& sc) {
  sc.rt_materials.clear();
  for (const glz_material& m : sc.materials) {
    RTMaterial r{};
    for (int i = 0; i < 3; ++i) r.diffuse_mul[i] = (float)m.diffuse_mul[i] / 255.0f;
    r.diffuse_mul[3] = 1.0f;
    for (int i = 0; i < 3; ++i) r.emissive_col[i] = m.has_emissive ? (float)m.emissive_col[i] / 255.0f : 0.0f;
    r.emissive_col[3] = 1.0f;
    unsigned metal = m.metal < GLZ_METAL_COUNT ? m.metal : 0;
    for (int i = 0; i < 16; ++i) {
      float n = GLZ_METAL_N[metal][i], k = GLZ_METAL_K[metal][i];
      r.metal_ior.w[i] = n;
      r.metal_fresnel.w[i] = (n * n) + (k * k);
    }
    r.diffuse = m.diffuse; r.roughness = m.roughness; r.metalness = m.metalness; r.opacity = m.opacity; r.normal = m.normal;
    r.bsdf_index = sbt_index(m.mtype);
    r.roughness_mul = m.roughness_mul; r.metalness_mul = m.metalness_mul; r.anisotropy = m.anisotropy; r.ior_dielectric = m.ior;
    r.is_specular = (m.mtype == GLZ_MAT_MIRROR || m.mtype == GLZ_MAT_GLASS) ? 1 : 0;   // material.rs:103-114
    r.is_emissive = m.has_emissive ? 1 : 0;
    sc.rt_materials.push_back(r);
  }
}

// load_raytrace_instances_to_gpu (scene.rs:1784-1818)
void build_rt_instances(Scene& sc) {
  sc.instances.clear();
  for (const glz_mesh_instance& in : sc.mesh_instances) {
    const glz_mesh* found = nullptr;
    for (const glz_mesh& m : sc.meshes)
      if (m.id == in.mesh_id) found = &m;   // FnvHashMap collect: the last mesh with that id wins
    if (!found) continue;
    sc.instances.push_back(RTInstance{found->index_offset, found->index_count, found->material, in.transform_id});
  }
}

// reorder_lights (scene.rs:628-635) + load_raytrace_lights_to_gpu (scene.rs:1863-1927)
void build_rt_lights(Scene& sc, const std::vector<glz_light>& parsed) {
  sc.lights.clear();
  const glz_light* sky = nullptr;
  for (const glz_light& l : parsed)
    if (l.ltype == GLZ_LIGHT_SKY && !sky) sky = &l;
  for (const glz_light& l : parsed)
    if (l.ltype != GLZ_LIGHT_SKY) sc.lights.push_back(l);
  if (sky) sc.lights.push_back(*sky);
  sc.lights_no = (uint32_t)sc.lights.size();     // scene.rs:1549: lights.len(), NOT the expanded count
  sc.rt_lights.clear();
  for (const glz_light& l : sc.lights) {
    float dx = l.direction[0], dy = l.direction[1], dz = l.direction[2];
    if (dx == 0.0f && dy == 0.0f && dz == 0.0f) dy = -1.0f;
    // `dir.normalize();` discards its result (Q14): the direction stays un-normalised
    RTLight r{};
    memcpy(r.color.w, l.color, 64);
    r.pos[0] = l.position[0]; r.pos[1] = l.position[1]; r.pos[2] = l.position[2]; r.pos[3] = 0;
    r.dir[0] = dx; r.dir[1] = dy; r.dir[2] = dz; r.dir[3] = 0;
    r.shader = l.ltype;                            // light.rs:111-119 (stride 1)
    r.instance_id = 0xFFFFFFFFu;
    r.intensity = l.intensity;
    r.delta = (l.ltype == GLZ_LIGHT_OMNI || l.ltype == GLZ_LIGHT_SUN) ? 1 : 0;
    if (l.ltype == GLZ_LIGHT_AREA) {
      // map_materials_to_instances (scene.rs:1764-1781): instance ids whose mesh uses this material
      uint16_t material_id = (uint16_t)l.resource_id;
      std::vector<uint32_t> ids;
      for (size_t i = 0; i < sc.mesh_instances.size(); ++i) {
        const glz_mesh* found = nullptr;
        for (const glz_mesh& m : sc.meshes)
          if (m.id == sc.mesh_instances[i].mesh_id) found = &m;
        if (found && found->material == material_id) ids.push_back((uint32_t)(uint16_t)i);
      }
      if (ids.empty()) ids.push_back(0);
      for (uint32_t id : ids) { r.instance_id = id; sc.rt_lights.push_back(r); }
    } else {
      sc.rt_lights.push_back(r);
    }
  }
  if (sc.rt_lights.empty()) {
    RTLight r{};
    r.instance_id = 0xFFFFFFFFu; r.intensity = 1.0f; r.delta = 1;
    sc.rt_lights.push_back(r);
  }
}

// generate_derivatives.comp:23-64
void build_derivatives(Scene& sc) {
  size_t ntri = 0;
  for (const glz_mesh& m : sc.meshes) ntri = std::max<size_t>(ntri, ((size_t)m.index_offset + m.index_count) / 3);   // scene.rs:2123-2128
  sc.derivatives.assign(ntri * 12, 0.0f);
  for (size_t t = 0; t < ntri; ++t) {
    const glz_vertex& a = sc.vertices[sc.indices[3 * t]];
    const glz_vertex& b = sc.vertices[sc.indices[3 * t + 1]];
    const glz_vertex& c = sc.vertices[sc.indices[3 * t + 2]];
    V3 p0 = v3(a.vv[0], a.vv[1], a.vv[2]), p1 = v3(b.vv[0], b.vv[1], b.vv[2]), p2 = v3(c.vv[0], c.vv[1], c.vv[2]);
    float duv02x = a.vt[0] - c.vt[0], duv02y = a.vt[1] - c.vt[1];
    float duv12x = b.vt[0] - c.vt[0], duv12y = b.vt[1] - c.vt[1];
    float det = duv02x * duv12y - duv02y * duv12x;
    V3 dp20 = p2 - p0, dp10 = p1 - p0;
    V3 n = normalize(cross(dp10, dp20));
    V3 dpdu, dpdv;
    if (det == 0.0f) {
      if (fabsf(n.x) > fabsf(n.y)) dpdu = v3(-n.z, 0.0f, n.x) / sqrtf(n.x * n.x + n.z * n.z);
      else dpdu = v3(0.0f, n.z, -n.y) / sqrtf(n.y * n.y + n.z * n.z);
      dpdv = cross(n, dpdu);
    } else {
      V3 dp02 = p0 - p2, dp12 = p1 - p2;
      float invdet = 1.0f / det;
      dpdu = (duv12y * dp02 - duv02y * dp12) * invdet;
      dpdv = ((-duv12x) * dp02 + duv02x * dp12) * invdet;
    }
    float* o = &sc.derivatives[t * 12];
    o[0] = n.x; o[1] = n.y; o[2] = n.z; o[3] = 0;
    o[4] = dpdu.x; o[5] = dpdu.y; o[6] = dpdu.z; o[7] = 0;
    o[8] = dpdv.x; o[9] = dpdv.y; o[10] = dpdv.z; o[11] = 0;
  }
}

// Distribution1D::new (distribution.rs:14-38), appended to flat arrays
void distribution1d(const float* values, size_t n, std::vector<float>& cdf_out, float& integral) {
  const float nf = (float)n;
  size_t base = cdf_out.size();
  cdf_out.resize(base + n + 1);
  float* cdf = &cdf_out[base];
  cdf[0] = 0.0f;
  for (size_t i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + values[i - 1] / nf;
  integral = cdf[n];
  if (integral == 0.0f) {
    for (size_t i = 1; i < n + 1; ++i) cdf[i] = (float)i / nf;
  } else {
    for (size_t i = 1; i < n + 1; ++i) cdf[i] = cdf[i] / integral;
  }
}

// calculate_skymap_distributions + build_sky_raytrace_buffers (scene.rs:2191-2313)
void build_sky(Scene& sc) {
  const glz_light* sky = (!sc.lights.empty() && sc.lights.back().ltype == GLZ_LIGHT_SKY) ? &sc.lights.back() : nullptr;   // scene.rs:1500
  glz_light dflt{};   // Light::default(): OMNI, resource 0, angles 0, intensity 1
  dflt.intensity = 1.0f;
  const glz_light& l = sky ? *sky : dflt;
  M4 rot = sky_rotation(l.yaw_deg, l.pitch_deg, l.roll_deg);
  // the reference builds the matrix in f32 (cgmath Matrix4<f32>) and inverts that
  float rotf[16];
  m4_to_f32(rot, rotf);
  M4 rot32 = m4_from_f32(rotf), inv;
  if (!m4_invert(rot32, inv)) inv = m4_identity();
  memcpy(sc.sky_obj2world, rotf, 64);
  m4_to_f32(inv, sc.sky_world2obj);
  sc.sky_tex_id = l.resource_id;
  sc.sky_intensity = l.intensity;
  const Tex& map = sc.textures[sc.sky_tex_id < sc.textures.size() ? sc.sky_tex_id : 0];
  const uint32_t W = map.w, H = map.h;
  const size_t bpp = map.format == GLZ_TEX_GRAY ? 1 : 4;
  std::vector<float> values((size_t)W * H);
  const float PI_F = 3.14159265358979323846f;   // std::f32::consts::PI
  for (uint32_t y = 0; y < H; ++y) {
    float sint = sinf(PI_F * ((float)y + 0.5f) / (float)H);
    for (uint32_t x = 0; x < W; ++x) {
      const uint8_t* p = &map.px[((size_t)y * W + x) * bpp];
      float r = (float)p[0] / 255.0f, g = (float)p[bpp > 1 ? 1 : 0] / 255.0f, b = (float)p[bpp > 1 ? 2 : 0] / 255.0f;
      values[(size_t)y * W + x] = host_luminance(host_from_rgb(r, g, b, true)) * sint;
    }
  }
  sc.sky_w = W; sc.sky_h = H;
  sc.cond_values = values;
  sc.cond_cdf.clear();
  std::vector<float> integrals(H);
  for (uint32_t y = 0; y < H; ++y) distribution1d(&values[(size_t)y * W], W, sc.cond_cdf, integrals[y]);
  std::vector<float> mcdf;
  distribution1d(integrals.data(), H, mcdf, sc.marginal_integral);
  sc.marginal_cdf_count = H + 1;
  sc.conditional_integral_offset = H + (H + 1);
  sc.conditional_cdf_count = W + 1;
  sc.marginal = mcdf;                                                       // cdf (H+1)
  sc.marginal.insert(sc.marginal.end(), integrals.begin(), integrals.end()); // marginal values (H) = row integrals
  sc.marginal.insert(sc.marginal.end(), integrals.begin(), integrals.end()); // conditional integrals (H)
}

// ---- acceleration: flatten instances to world space, median-split BVH ----------------------
void build_accel(Scene& sc) {
  sc.tris.clear();
  uint32_t world_id = 0;
  for (size_t i = 0; i < sc.instances.size(); ++i) {
    const RTInstance& in = sc.instances[i];
    const float* M = sc.transforms[in.transform_id].m;
    const bool non_opaque = sc.materials[in.material_id].opacity != 0;    // acceleration.rs:136-141
    for (uint32_t p = 0; p < in.index_count / 3; ++p, ++world_id) {
      const uint32_t* ix = &sc.indices[in.index_offset + 3 * p];
      V3 v[3];
      for (int k = 0; k < 3; ++k) {
        const glz_vertex& vt = sc.vertices[ix[k]];
        v[k] = mat_point(M, v3(vt.vv[0], vt.vv[1], vt.vv[2]));
      }
      Tri t;
      t.v0 = v[0]; t.v1 = v[1]; t.v2 = v[2];
      t.world_id = world_id; t.instance = (uint32_t)i; t.prim = p; t.non_opaque = non_opaque;
      sc.tris.push_back(t);
    }
  }
  sc.nodes.clear();
  const size_t n = sc.tris.size();
  if (n == 0) return;
  std::vector<float> lo(n * 3), hi(n * 3), ce(n * 3);
  auto bounds_of = [&](const Tri& t, float* l, float* h) {
    V3 a = t.v0, b = t.v1, c = t.v2;
    const float* A = &a.x; const float* B = &b.x; const float* C = &c.x;
    for (int k = 0; k < 3; ++k) {
      l[k] = fminf(A[k], fminf(B[k], C[k]));
      h[k] = fmaxf(A[k], fmaxf(B[k], C[k]));
      // conservative padding: the ray / triangle test may accept points a few ulps outside the exact bounds
      float pad = 1e-5f * fmaxf(fmaxf(fabsf(l[k]), fabsf(h[k])), 1e-3f);
      l[k] -= pad; h[k] += pad;
    }
  };
  std::vector<uint32_t> order(n);
  for (size_t i = 0; i < n; ++i) {
    order[i] = (uint32_t)i;
    bounds_of(sc.tris[i], &lo[i * 3], &hi[i * 3]);
    for (int k = 0; k < 3; ++k) ce[i * 3 + k] = 0.5f * (lo[i * 3 + k] + hi[i * 3 + k]);
  }
  struct Work { int node; size_t first, count; };
  std::vector<Work> stack;
  sc.nodes.push_back(BNode{});
  stack.push_back(Work{0, 0, n});
  while (!stack.empty()) {
    Work w = stack.back();
    stack.pop_back();
    BNode nd{};
    float clo[3] = {INF, INF, INF}, chi[3] = {-INF, -INF, -INF};
    for (int k = 0; k < 3; ++k) { nd.lo[k] = INF; nd.hi[k] = -INF; }
    for (size_t i = w.first; i < w.first + w.count; ++i) {
      uint32_t t = order[i];
      for (int k = 0; k < 3; ++k) {
        nd.lo[k] = fminf(nd.lo[k], lo[t * 3 + k]); nd.hi[k] = fmaxf(nd.hi[k], hi[t * 3 + k]);
        clo[k] = fminf(clo[k], ce[t * 3 + k]); chi[k] = fmaxf(chi[k], ce[t * 3 + k]);
      }
    }
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    if (w.count <= 4 || !(chi[axis] > clo[axis])) {
      nd.first = (int)w.first; nd.count = (int)w.count; nd.left = nd.right = -1;
      sc.nodes[w.node] = nd;
      continue;
    }
    size_t mid = w.first + w.count / 2;
    std::nth_element(order.begin() + w.first, order.begin() + mid, order.begin() + w.first + w.count,
                     [&](uint32_t a, uint32_t b) { return ce[a * 3 + axis] < ce[b * 3 + axis]; });
    nd.count = 0; nd.first = 0;
    nd.left = (int)sc.nodes.size();
    nd.right = nd.left + 1;
    sc.nodes[w.node] = nd;
    sc.nodes.push_back(BNode{});
    sc.nodes.push_back(BNode{});
    stack.push_back(Work{nd.left, w.first, mid - w.first});
    stack.push_back(Work{nd.right, mid, w.first + w.count - mid});
  }
  std::vector<Tri> sorted(n);
  for (size_t i = 0; i < n; ++i) sorted[i] = sc.tris[order[i]];
  sc.tris.swap(sorted);
}

// ------------------------------------------------------------------------------------------
// Ray / triangle -- the build's definition of what the Vulkan driver does for traceRayEXT ([ext], parity unpinned;
// path_trace.rgen:169, :106-109 on the acceleration structure of acceleration.rs:319-345).  Candidate accepted iff
// tmin < t < tmax.  No culling (TRIANGLE_FACING_CULL_DISABLE, acceleration.rs:335-345).
//
// WATERTIGHT, as the Vulkan specification requires of the driver's intersector: a ray cannot pass between triangles that share
// an edge or a vertex.  The rule is the one of Woop, Benthin, Wald, "Watertight Ray/Triangle Intersection" (JCGT 2013), with
// the exact tie-break done in single precision:
//   * per RAY: the axis kz of the direction's largest component and the shear that takes d to (0, 0, 1):
//     Sz = 1 / d[kz], Sx = d[kx] * Sz, Sy = d[ky] * Sz  (kx, ky the next two axes; no winding swap -- nothing is culled);
//   * per VERTEX P: A = P - o, then its image (A[kx] - Sx A[kz], A[ky] - Sy A[kz], Sz A[kz]).  The image is a function of the
//     ray and of the vertex's floats only, so every triangle that uses the vertex sees the SAME 2-D point (the records hold
//     the vertices themselves for that reason, not a vertex and two edges);
//   * per EDGE: the 2-D edge function  U = Cx By - Cy Bx  (V, W alike) from two separately rounded products.  Rounding is
//     monotonic, so the sign of fl(Cx By) - fl(Cy Bx) is the sign of the exact value unless the two rounded products are
//     equal; then the exact value is the difference of the two rounding errors, each of which one fma delivers exactly
//     (fma(a, b, -fl(a b)) = a b - fl(a b)).  Every sign is therefore the EXACT orientation of the origin against two fixed 2-D
//     points, and exact predicates on consistent points cannot disagree: across a shared edge one triangle sees e, the other
//     -e; around a shared vertex the fan's spokes cannot all lie on one side of the origin unless the origin is outside the fan.
//     (A product that underflows loses this exactness; scenes are not built out of 1e-20-sized coordinates.)
//   * inside iff no two of U, V, W have opposite signs; zero counts as inside for BOTH neighbours, the tie on t then goes to
//     the smaller world id.  det = U + V + W;  u = V / det, v = W / det (weights of v1, v2);  t = (U Az + V Bz + W Cz) / det.
// Moeller-Trumbore, which north_star names and rounds 1-2 ran, evaluates u, v and 1 - u - v by three differently rounded
// expressions per triangle and leaks (found by tests/test_analytic_render.py: the parallel rays of an orthographic camera
// through a square's diagonal); scalar triple products d . (P x Q) made antisymmetric by unfused cross products close the
// edges but not the vertices (5 % of the rays aimed at the vertices of a closed mesh got out: tests/test_watertight.py).
// ------------------------------------------------------------------------------------------
// Fused operations are stated explicitly (fmaf is correctly rounded, so every implementation of this statement gives the same
// bits); everything else is one rounding per operation (-ffp-contract=off).
inline float dot_fma(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
struct RayShear { int kx, ky, kz; float sx, sy, sz; };
inline RayShear ray_shear(V3 d) {
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  RayShear r;
  r.kz = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
  r.kx = (r.kz + 1) % 3;
  r.ky = (r.kz + 2) % 3;
  const float* c = &d.x;
  r.sz = 1.0f / c[r.kz];
  r.sx = c[r.kx] * r.sz;
  r.sy = c[r.ky] * r.sz;
  return r;
}
inline V3 shear_vertex(const RayShear& r, V3 p, V3 o) {
  const V3 a = p - o;
  const float* c = &a.x;
  return v3(fmaf(-r.sx, c[r.kz], c[r.kx]), fmaf(-r.sy, c[r.kz], c[r.ky]), r.sz * c[r.kz]);
}
// a*b - c*d with the exact sign: the difference of the rounded products, or of their rounding errors when those are equal
inline float edge_fn(float a, float b, float c, float d) {
  const float p = a * b, q = c * d;
  const float e = p - q;
  return e != 0.0f ? e : fmaf(a, b, -p) - fmaf(c, d, -q);
}
inline bool ray_tri(const Tri& tr, V3 o, V3 d, float tmin, float tmax, float& t, float& u, float& v) {
  const RayShear rs = ray_shear(d);
  const V3 A = shear_vertex(rs, tr.v0, o), B = shear_vertex(rs, tr.v1, o), C = shear_vertex(rs, tr.v2, o);
  const float U = edge_fn(C.x, B.y, C.y, B.x), V = edge_fn(A.x, C.y, A.y, C.x), W = edge_fn(B.x, A.y, B.y, A.x);
  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
  const float det = (U + V) + W;
  // The ray lies in the triangle's plane, or the triangle has no area in this projection: det is zero -- or, in floating point, NOT
  // DISTINGUISHABLE from zero: it is the sum of three differences of products, and when it is smaller than 2^-19 of the first product
  // of each (about 2^-20 of all six) it is their rounding.  A shadow ray towards a point of a light that lies in the plane of the
  // surface it leaves is exactly that case; every coplanar triangle along it passes the (exact) inside test, and the distance that
  // comes out of a noise-sized det is anything -- "occluded" at a point outside the triangle, or not, depending on the boxes of
  // whoever walks the scene (tools/gpu_fuzz_classify.py: every one of the 16 scenes in 150 000 that still differed was this, with
  // |det| between 2e-8 and 2.4e-7 of the products' sum, everything else above 1e-6).  Such a candidate does not count.
  const float noise = ((fabsf(C.x * B.y) + fabsf(A.x * C.y)) + fabsf(B.x * A.y)) * 1.9073486e-6f;
  if (!(fabsf(det) > noise)) return false;
  const float inv = 1.0f / det;
  u = V * inv;
  v = W * inv;
  t = fmaf(W, C.z, fmaf(V, B.z, U * A.z)) * inv;
  return t > tmin && t < tmax;      // a NaN anywhere above ends here
}

// raytrace_hit.rahit:24-39: ignore the candidate when the opacity texture's red channel < 0.5
bool alpha_pass(const Scene& sc, const Tri& tr, float u, float v) {
  const RTInstance& in = sc.instances[tr.instance];
  const uint32_t* ix = &sc.indices[(in.index_offset / 3 + tr.prim) * 3];
  const glz_vertex &a = sc.vertices[ix[0]], &b = sc.vertices[ix[1]], &c = sc.vertices[ix[2]];
  float bx = 1.0f - u - v;
  float tu = (a.vt[0] * bx + b.vt[0] * u) + c.vt[0] * v;
  float tv = (a.vt[1] * bx + b.vt[1] * u) + c.vt[1] * v;
  float alpha = texture_bilinear(sc, sc.rt_materials[in.material_id].opacity, tu, tv).x;
  return !(alpha < 0.5f);
}

struct Hit {
  float t, u, v;
  uint32_t tri;    // index into sc.tris
  bool valid;
};

struct Counters { uint64_t nodes = 0, tris = 0; };

// Conservative slab test of the oracle's own hierarchy (an accelerator only: orc_trace_closest_brute is the definition).  The plane
// distances are widened by the rounding they can carry -- a difference of operands of size |plane| + |origin| times 1 / d, three
// roundings -- so that a box is never left out because its entry rounds past the distance of a hit already found.  (Without it a
// triangle lying in a plane near coordinate 0, seen from an origin a unit away, lost exact ties against a coincident triangle of
// another instance: the 1e-5 relative pad of a near-zero coordinate vanishes in the subtraction.  tools/gpu_fuzz_parity.py.)
inline bool ray_box(const float* lo, const float* hi, V3 o, V3 inv, float tmin, float tmax) {
  float t0 = tmin, t1 = tmax;
  const float* O = &o.x; const float* I = &inv.x;
  for (int k = 0; k < 3; ++k) {
    float a = (lo[k] - O[k]) * I[k], b = (hi[k] - O[k]) * I[k];
    float e = ((fmaxf(fabsf(lo[k]), fabsf(hi[k])) + fabsf(O[k])) * fabsf(I[k])) * 3.6e-7f;
    float n = fminf(a, b) - e, f = fmaxf(a, b) + e;   // fminf/fmaxf below drop NaNs (0*inf, inf-inf on slab-parallel rays)
    t0 = fmaxf(t0, n);
    t1 = fminf(t1, f * 1.0000005f);
  }
  return t0 <= t1;
}

// The far end of the interval a box is tested against: the distance of the hit in hand (or the ray's tmax), widened by what the
// TRIANGLE test's distance can be off -- a depth in sheared coordinates, a few ulps of the largest coordinates of origin and hit point,
// whichever axis the box is thin in.  (ray_box's own widening covers the slab's rounding, in the units of the slab's axis: for a ray
// that starts 0.0075 above a flat mesh at x = 0.32 that was 6e-9, the triangle's t 1.5e-8 below the plane's, and the coincident
// triangle of another instance lost its tie; tools/gpu_fuzz_parity.py, seed 60378 with its second camera.)
inline float far_bound(V3 o, float t) {
  const float omax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
  return t + 4.8e-7f * (2.0f * omax + t);   // inf stays inf
}
// closest hit: smallest t; ties broken by smallest world triangle id (traversal-order independent)
// ORC_NO_HIERARCHY=1 in the environment: every trace walks ALL triangles (the definition itself, for looking at a difference between
// the HIP path and the oracle without the oracle's hierarchy in the picture; tools/gpu_fuzz_diag.py)
static bool no_hierarchy() {
  static const bool off = [] { const char* e = getenv("ORC_NO_HIERARCHY"); return e && *e && *e != '0'; }();
  return off;
}
Hit trace_closest(const Scene& sc, V3 o, V3 d, float tmin, float tmax) {
  Hit best{tmax, 0, 0, 0, false};
  if (sc.nodes.empty()) return best;
  V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  int stack[128];
  int sp = 0;
  stack[sp++] = 0;
  const bool all = no_hierarchy();
  BNode everything{};
  everything.first = 0;
  everything.count = (int)sc.tris.size();
  while (sp) {
    const BNode& nd = all ? everything : sc.nodes[stack[--sp]];
    if (all) sp = 0;
    if (!all && !ray_box(nd.lo, nd.hi, o, inv, tmin, far_bound(o, best.t))) continue;
    if (nd.count > 0) {
      for (int i = nd.first; i < nd.first + nd.count; ++i) {
        float t, u, v;
        // accept t == best.t too so that the id tie-break sees every candidate
        if (!ray_tri(sc.tris[i], o, d, tmin, INF, t, u, v)) continue;
        if (!(t < tmax)) continue;
        bool better = !best.valid ? true : (t < best.t || (t == best.t && sc.tris[i].world_id < sc.tris[best.tri].world_id));
        if (!better) continue;
        if (sc.tris[i].non_opaque && !alpha_pass(sc, sc.tris[i], u, v)) continue;
        best = Hit{t, u, v, (uint32_t)i, true};
      }
    } else {
      stack[sp++] = nd.left;
      stack[sp++] = nd.right;
    }
  }
  return best;
}

bool trace_any(const Scene& sc, V3 o, V3 d, float tmin, float tmax) {
  if (sc.nodes.empty()) return false;
  V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  int stack[128];
  int sp = 0;
  stack[sp++] = 0;
  const bool all = no_hierarchy();
  BNode everything{};
  everything.first = 0;
  everything.count = (int)sc.tris.size();
  while (sp) {
    const BNode& nd = all ? everything : sc.nodes[stack[--sp]];
    if (all) sp = 0;
    if (!all && !ray_box(nd.lo, nd.hi, o, inv, tmin, far_bound(o, tmax))) continue;
    if (nd.count > 0) {
      for (int i = nd.first; i < nd.first + nd.count; ++i) {
        float t, u, v;
        if (!ray_tri(sc.tris[i], o, d, tmin, tmax, t, u, v)) continue;
        if (sc.tris[i].non_opaque && !alpha_pass(sc, sc.tris[i], u, v)) continue;
        return true;
      }
    } else {
      stack[sp++] = nd.left;
      stack[sp++] = nd.right;
    }
  }
  return false;
}

// Traversal of an externally supplied BVH4 in the product's node/leaf layout (64-byte nodes with 16-bit grid boxes,
// DESIGN.md section 2), counting node and triangle visits: the "counted on the shared LBVH" figures of SURVEY 8(d).
// Ordered (near child first, ties -> child0) traversal with pruning against the current best t -- exactly the visit
// rule of the HIP tracer, so the instrumented kernels' counters must equal these counts.
inline uint32_t fbits(float f);
struct ExtNode { uint32_t w[16]; };   // BvhNode4
struct ExtTri { float v0[3]; uint32_t world_id; float v1[3]; uint32_t instance; float v2[3]; uint32_t prim_flags; };
inline float box_entry_q(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 ig, V3 cg, float tmin, float tmax) {
  // plane distances as one correctly rounded fma each (device/wavefront.h box_key: v_pk_fma_f32).  The kernel's byte permute hands the
  // fma the float 32768 + q (exact: q < 32768) and the addend carries the - 32768 ig (grid_addend): restated here operation for operation
  const float M = 32768.0f;
  const float ax = fmaf(M + lox, ig.x, cg.x), bx = fmaf(M + hix, ig.x, cg.x);
  const float ay = fmaf(M + loy, ig.y, cg.y), by = fmaf(M + hiy, ig.y, cg.y);
  const float az = fmaf(M + loz, ig.z, cg.z), bz = fmaf(M + hiz, ig.z, cg.z);
  const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
  const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));   // (the kernel picks the near / far plane by the sign of ig: same values)
  return t0 <= fminf(t1, tmax) ? t0 : INF;
}
void ext_trace(const Scene& sc, V3 o, V3 d, float tmin, float tmax, bool any, Counters& c, float& t_out, uint32_t& id_out) {
  const ExtNode* nodes = (const ExtNode*)sc.ext_nodes.data();
  const ExtTri* tris = (const ExtTri*)sc.ext_tris.data();
  t_out = INF; id_out = 0xFFFFFFFFu;
  if (sc.ext_nodes.empty()) return;
  const float* G = sc.ext_grid;   // lo[3], cell[3], inv_cell[3]
  const V3 og = v3((o.x - G[0]) * G[6], (o.y - G[1]) * G[7], (o.z - G[2]) * G[8]);
  {   // kernels_render.hip ray_is_finite: such a ray is a miss without traversal
    const float s = ((o.x + o.y) + o.z) + ((d.x + d.y) + d.z);
    if (!(s - s == 0.0f)) return;
  }
  auto inv_dir = [](float x) { const float i = 1.0f / x; return fabsf(i) <= 1e30f ? i : copysignf(1e30f, i); };   // grid_inv_dir
  const V3 ig = v3(inv_dir(d.x) * G[3], inv_dir(d.y) * G[4], inv_dir(d.z) * G[5]);
  const V3 cg = v3(fmaf(-32768.0f, ig.x, -(og.x * ig.x)), fmaf(-32768.0f, ig.y, -(og.y * ig.y)), fmaf(-32768.0f, ig.z, -(og.z * ig.z)));   // grid_addend
  float best = tmax; uint32_t best_id = 0xFFFFFFFFu; bool found = false;
  int stack[512]; int sp = 0; int cur = 0;
  for (;;) {
    if (cur >= 0) {
      const uint32_t* w = nodes[cur].w;
      c.nodes++;
      uint32_t key[4];
      for (uint32_t k = 0; k < 4; ++k) {
        // one word per axis: lo | hi << 16
        const float e = box_entry_q((float)(w[3 * k] & 0xFFFFu), (float)(w[3 * k + 1] & 0xFFFFu), (float)(w[3 * k + 2] & 0xFFFFu), (float)(w[3 * k] >> 16),
                                    (float)(w[3 * k + 1] >> 16), (float)(w[3 * k + 2] >> 16), ig, cg, tmin, best);
        key[k] = (e < INF && w[12 + k] != 0x7FFFFFFFu) ? ((fbits(e) & 0xFFFFFC00u) | (k << 8)) : 0xFFFFFFFFu;   // box_key, device/wavefront.h: distance bits, child index in bits 8..9
      }
      std::sort(key, key + 4);
      if (key[0] != 0xFFFFFFFFu) {
        for (int k = 3; k >= 1; --k)
          if (key[k] != 0xFFFFFFFFu) stack[sp++] = (int)w[12 + ((key[k] >> 8) & 3u)];
        cur = (int)w[12 + ((key[0] >> 8) & 3u)];
        continue;
      }
    } else {
      // a leaf is one triangle or two adjacent ones (bit 30 of the first one's flags, kTriHasPartner): both are tested in one visit
      const uint32_t first = (uint32_t)~cur;
      const uint32_t count = (tris[first].prim_flags & 0x40000000u) ? 2u : 1u;
      bool stop = false;
      for (uint32_t s = first; s < first + count; ++s) {
        const ExtTri& et = tris[s];
        c.tris++;
        Tri tr; tr.v0 = v3(et.v0[0], et.v0[1], et.v0[2]); tr.v1 = v3(et.v1[0], et.v1[1], et.v1[2]); tr.v2 = v3(et.v2[0], et.v2[1], et.v2[2]);
        tr.instance = et.instance; tr.prim = et.prim_flags & 0x3FFFFFFFu; tr.non_opaque = (et.prim_flags >> 31) != 0;
        float t, u, v;
        if (ray_tri(tr, o, d, tmin, INF, t, u, v) && t < tmax) {
          bool better = !found ? true : (t < best || (t == best && et.world_id < best_id));
          if (better && (!tr.non_opaque || alpha_pass(sc, tr, u, v))) {
            found = true; best = t; best_id = et.world_id;
            if (any) stop = true;
          }
        }
      }
      if (stop) break;
    }
    if (!sp) break;
    cur = stack[--sp];
  }
  if (found) { t_out = best; id_out = best_id; }
}

// ------------------------------------------------------------------------------------------
// Shading space, Fresnel, microfacets
// ------------------------------------------------------------------------------------------
struct ShadingSpace { V3 s, t, n; };
inline ShadingSpace new_shading_space(V3 dpdu, V3 n) {                 // shading_space.glsl:11-16
  V3 s = normalize(dpdu - n * dot(n, dpdu));
  V3 t = cross(n, s);
  return ShadingSpace{s, t, n};
}
inline V3 to_world_space(V3 v, const ShadingSpace& m) {                 // shading_space.glsl:18-24
  V3 r = v3((m.s.x * v.x + m.t.x * v.y) + m.n.x * v.z, (m.s.y * v.x + m.t.y * v.y) + m.n.y * v.z, (m.s.z * v.x + m.t.z * v.y) + m.n.z * v.z);
  return normalize(r);
}
inline V3 to_shading_space(V3 w, const ShadingSpace& m) {               // shading_space.glsl:26-30
  return normalize(v3(dot(w, m.s), dot(w, m.t), dot(w, m.n)));
}

Sp fresnel_conductor(float cosin, const Sp& ior, const Sp& ior2abs2) {  // fresnel.glsl:7-17
  float cosin2 = cosin * cosin;
  Sp etacosin2 = sp_mul(ior, cosin * 2.0f);
  Sp etacosin2plus = sp_add(etacosin2, cosin2);
  Sp etacosin2plusplus = sp_add(etacosin2, 1.0f);
  Sp rperpsq = sp_div(sp_sub(ior2abs2, etacosin2plus), sp_add(ior2abs2, etacosin2plus));
  Sp tmp = sp_mul(ior2abs2, cosin2);
  Sp rparsq = sp_div(sp_sub(tmp, etacosin2plusplus), sp_add(tmp, etacosin2plusplus));
  return sp_div(sp_add(rperpsq, rparsq), 2.0f);
}

float fresnel_dielectric(float costi, float etai, float etat) {         // fresnel.glsl:19-35
  float sin2ti = gmax(0.0f, 1.0f - costi * costi);
  float sin2tt = etai * etai / (etat * etat) * sin2ti;
  if (sin2tt >= 1.0f) return 1.0f;
  float costt = sqrtf(gmax(0.0f, 1.0f - sin2tt));
  float etatcostt = etat * costt, etatcosti = etat * costi, etaicosti = etai * costi, etaicostt = etai * costt;
  float rparl = (etatcosti - etaicostt) / (etatcosti + etaicostt);
  float rperp = (etaicosti - etatcostt) / (etaicosti + etatcostt);
  return (rparl * rparl + rperp * rperp) / 2.0f;
}

struct V2 { float x, y; };

V2 ggx_sample_p22(float cost, V2 r) {                                    // microfacets.glsl:23-55
  if (cost > 0.999f) {
    float rr = sqrtf(r.x / (1.0f - r.x));
    float phi = TWO_PI * r.y;
    return V2{rr * glz_cosf(phi), rr * glz_sinf(phi)};
  }
  float cos2t = cost * cost;
  float sin2t = gmax(0.0f, 1.0f - cos2t);
  float tan2t = checkinf(sin2t / cos2t);
  float tant = sqrtf(tan2t);
  float a2 = 1.0f / tan2t;
  float G1 = 2.0f / (1.0f + sqrtf(1.0f + 1.0f / a2));
  float A = 2.0f * r.x / G1 - 1.0f;
  float B = tant;
  float invA2m1 = 1.0f / (A * A - 1.0f);
  float sqrt_term = sqrtf(gmax(0.0f, B * B * invA2m1 * invA2m1 - (A * A - B * B) * invA2m1));
  float sx1 = B * invA2m1 - sqrt_term;
  float sx2 = B * invA2m1 + sqrt_term;
  float sx = (A < 0.0f || sx2 > 1.0f / tant) ? sx1 : sx2;
  float stepval = gstep(0.5f, r.y);
  float s = gmix(1.0f, -1.0f, stepval);
  float u = gmix(2.0f * (r.y - 0.5f), 2.0f * (0.5f - r.y), stepval);
  float z = (u * (u * (u * -0.3657289f + 0.7902350f) - 0.4249658f) + 0.0001529f) /
            (u * (u * (u * (u * 0.1695078f - 0.3972035f) - 0.2325005f) + 1.0f) - 0.5398259f);
  float sy = s * z * sqrtf(1.0f + sx * sx);
  return V2{sx, sy};
}

float ggx_d(V3 wh, V2 a) {                                               // microfacets.glsl:57-69
  float cost = wh.z;
  float cos2t = cost * cost;
  float cos4t = cos2t * cos2t;
  float sin2t = gmax(0.0f, 1.0f - cos2t);
  float tan2t = sin2t / cos2t;
  float cos2p = wh.x * wh.x / sin2t;
  float sin2p = wh.y * wh.y / sin2t;
  float eplus1 = 1.0f + ((cos2p / (a.x * a.x) + sin2p / (a.y * a.y)) * tan2t);
  float d = 1.0f / (PI * a.x * a.y * cos4t * eplus1 * eplus1);
  return isinf(tan2t) ? 0.0f : d;
}

float ggx_lambda(V3 v, V2 a) {                                           // microfacets.glsl:71-82
  float cost = v.z;
  float cos2t = cost * cost;
  float sin2t = gmax(0.0f, 1.0f - cos2t);
  float tan2t = sin2t / cos2t;
  float cos2p = gmax(0.0f, v.x * v.x / sin2t);
  float sin2p = gmax(0.0f, v.y * v.y / sin2t);
  float alpha2 = cos2p * a.x * a.x + sin2p * a.y * a.y;
  float lambda = (-1.0f + sqrtf(1.0f + tan2t * alpha2)) * 0.5f;
  return isinf(tan2t) ? 0.0f : lambda;
}
float ggx_g(V3 wo, V3 wi, V2 a) { return 1.0f / (1.0f + ggx_lambda(wo, a) + ggx_lambda(wi, a)); }   // :84-87
float ggx_g1(V3 v, V2 a) { return 1.0f / (1.0f + ggx_lambda(v, a)); }                                // :89-92
float ggx_pdf(float d, V2 a, V3 wo, V3 wh) { return d * ggx_g1(wh, a) * fabsf(dot(wo, wh)) / fabsf(wh.z); }   // :94-99 (Q6)

V3 ggx_sample_wh(V3 wo, V2 r, V2 a) {                                    // microfacets.glsl:102-120
  float flip = gsign(wo.z);
  V3 wi = flip * wo;
  V3 ws = normalize(v3(wi.x * a.x, wi.y * a.y, wi.z));
  float cost = ws.z;
  V2 slope = ggx_sample_p22(cost, r);
  float cos2t = cost * cost;
  float sin2t = gmax(0.0f, 1.0f - cos2t);
  float cosp = sqrtf(ws.x * ws.x / sin2t);
  float sinp = sqrtf(ws.y * ws.y / sin2t);
  float sx = cosp * slope.x - sinp * slope.y;
  float sy = sinp * slope.x + cosp * slope.y;
  return flip * normalize(v3(-a.x * sx, -a.y * sy, 1.0f));
}
inline V2 to_anisotropic(float a, float anis) { return V2{a * (1.0f + anis), a * (1.0f - anis)}; }   // :122-125

// ------------------------------------------------------------------------------------------
// BSDF callables (mat_*_value.rcall / mat_*_sample_value.rcall)
// ------------------------------------------------------------------------------------------
struct BsdfIn { V3 woW, wiW; V2 uv; ShadingSpace sh; uint32_t material_id; TexFootprint fp; };
inline V3 tex_rgb(const Scene& sc, uint32_t id, V2 uv, const TexFootprint& fp = TexFootprint()) { V4 t = texture_lod(sc, id, uv.x, uv.y, fp); return v3(t.x, t.y, t.z); }
inline float tex_r(const Scene& sc, uint32_t id, V2 uv, const TexFootprint& fp = TexFootprint()) { return texture_lod(sc, id, uv.x, uv.y, fp).x; }

// Oren-Nayar term shared by mat_uber_value.rcall:56-73 and mat_uber_sample_value.rcall:66-81
float oren_nayar_term(float roughness, V3 wo, V3 wi) {
  float sigma = roughness * 0.5f;
  float sigma2 = sigma * sigma;
  float A = 1.0f - sigma2 / (2.0f * (sigma2 + 0.33f));
  float B = 0.45f * sigma2 / (sigma2 + 0.09f);
  float sinto = sqrtf(gmax(0.0f, 1.0f - wo.z * wo.z));
  float sinti = sqrtf(gmax(0.0f, 1.0f - wi.z * wi.z));
  float sinpi = wi.y / sinti, cospi = wi.x / sinti;
  float sinpo = wo.y / sinto, cospo = wo.x / sinto;
  float maxcos = gmax(0.0f, cospi * cospo + sinpi * sinpo);
  float dotwi_g_dotwo = gstep(fabsf(wo.z), fabsf(wi.z));
  float sinalpha = gmix(sinto, sinti, dotwi_g_dotwo);
  float tanbeta = gmix(sinti / fabsf(wi.z), sinto / fabsf(wo.z), dotwi_g_dotwo);
  return INV_PI * (A + B * maxcos * sinalpha * tanbeta);
}

// Specular lobe shared by frosted-reflect and uber (mat_frosted_value.rcall:35-47, mat_uber_value.rcall:39-52)
struct SpecTerms { float d, g, pdf, costi, costwo, costwi; };
SpecTerms spec_terms(V3 wo, V3 wi, V3 wh, V2 a) {
  SpecTerms s;
  float dotwowh = dot(wo, wh), dotwiwh = dot(wi, wh);
  s.costi = dot(wi, gsign(dot(wh, v3(0, 0, 1))) * wh);
  s.costwo = fabsf(wo.z);
  s.costwi = fabsf(wi.z);
  s.d = gstep(0.0f, wo.z) * ggx_d(wh, a);
  s.g = gstep(0.0f, dotwowh) * gstep(0.0f, dotwiwh) * ggx_g(wo, wi, a);
  s.pdf = ggx_pdf(s.d, a, wo, wh) / (4.0f * dotwowh);
  return s;
}

// eval: returns pdf, fills value (value untouched when the shader leaves it stale)
float bsdf_value(const Scene& sc, const BsdfIn& in, float rand_sample, Sp& value) {
  const RTMaterial& mat = sc.rt_materials[in.material_id];
  switch (mat.bsdf_index) {
    case 4: {   // mat_lambert_value.rcall:23-34
      V3 wo = to_shading_space(in.woW, in.sh), wi = to_shading_space(in.wiW, in.sh);
      float same_hemi = gstep(0.0f, wo.z * wi.z);
      V3 tx = tex_rgb(sc, mat.diffuse, in.uv, in.fp);
      V3 dm = v3(mat.diffuse_mul[0], mat.diffuse_mul[1], mat.diffuse_mul[2]);
      value = from_surface_color((tx * dm) * INV_PI);
      return same_hemi * fabsf(wi.z) * INV_PI;
    }
    case 6: return 0.0f;   // mat_mirror_value.rcall:8-11
    case 8: return 0.0f;   // mat_glass_value.rcall:8-11
    case 10: {  // mat_metal_value.rcall:19-44
      V3 wo = to_shading_space(in.woW, in.sh), wi = to_shading_space(in.wiW, in.sh);
      V3 wh = normalize(wo + wi);
      float costwo = fabsf(wo.z), costwi = fabsf(wi.z);
      if (wo.z * wi.z > 0.0f) {
        Sp F = fresnel_conductor(dot(wi, wh), mat.metal_ior, mat.metal_fresnel);
        float rough = tex_r(sc, mat.roughness, in.uv, in.fp);
        V2 a = to_anisotropic(rough * mat.roughness_mul, mat.anisotropy);
        float d = ggx_d(wh, a);
        float g = ggx_g(wo, wi, a);
        float term = d * g / (4.0f * costwo * costwi);
        float pdf = ggx_pdf(d, a, wo, wh) / (4.0f * dot(wo, wh));
        value = sp_mul(F, term);
        return checknan(pdf);
      }
      return 0.0f;
    }
    case 12: {  // mat_frosted_value.rcall:19-66
      V3 wo = to_shading_space(in.woW, in.sh), wi = to_shading_space(in.wiW, in.sh);
      float rough = tex_r(sc, mat.roughness, in.uv, in.fp);
      V2 a = to_anisotropic(rough * mat.roughness_mul, mat.anisotropy);
      bool same_hemi = wo.z * wi.z > 0.0f;
      float from_outside = gstep(0.0f, wo.z);
      float etai = gmix(mat.ior_dielectric, DEFAULT_IOR, from_outside);
      float etat = gmix(DEFAULT_IOR, mat.ior_dielectric, from_outside);
      float eta = etai / etat;
      if (same_hemi) {
        V3 wh = normalize(wo + wi);
        SpecTerms s = spec_terms(wo, wi, wh, a);
        float f = fresnel_dielectric(s.costi, etai, etat);
        float term = s.d * s.g * f / (4.0f * s.costwo * s.costwi);
        value = sp_uniform(term);
        return checknan(s.pdf);
      } else {
        V3 wh = normalize(wo + eta * wi);
        wh = wh * gsign(wo.z);
        float dotwowh = dot(wo, wh), dotwiwh = dot(wi, wh);
        float f = fresnel_dielectric(dotwowh, etai, etat);
        float costwo = fabsf(wo.z), costwi = fabsf(wi.z);
        float denom = dotwowh + eta * dotwiwh;
        float d = ggx_d(wh, a);
        float g = ggx_g(wo, wi, a);
        float pdf = ggx_pdf(d, a, wo, wh) * fabsf(eta * eta * dotwiwh) / (denom * denom);
        float term = d * g * (1.0f - f) * fabsf(dotwiwh) * fabsf(dotwowh) / (denom * denom * costwo * costwi);
        value = sp_uniform(term);
        return checknan(pdf);
      }
    }
    default: {  // 14: mat_uber_value.rcall:20-77
      V3 wo = to_shading_space(in.woW, in.sh), wi = to_shading_space(in.wiW, in.sh);
      float rough_tex = tex_r(sc, mat.roughness, in.uv, in.fp);
      float roughness = rough_tex * mat.roughness_mul;
      float same_hemi = gstep(0.0f, wo.z * wi.z);
      if (rand_sample < 0.5f) {
        V2 a = to_anisotropic(roughness * mat.roughness_mul, mat.anisotropy);   // Q5: roughness_mul twice
        V3 wh = normalize(wo + wi);
        float metalness = tex_r(sc, mat.metalness, in.uv, in.fp) * mat.metalness_mul;
        float from_outside = gstep(0.0f, wo.z);
        float etai = gmix(mat.ior_dielectric, DEFAULT_IOR, from_outside);
        float etat = gmix(DEFAULT_IOR, mat.ior_dielectric, from_outside);
        SpecTerms s = spec_terms(wo, wi, wh, a);
        Sp fd = sp_uniform(fresnel_dielectric(s.costi, etai, etat));
        Sp fc = fresnel_conductor(s.costi, mat.metal_ior, mat.metal_fresnel);
        Sp f = sp_mix(fd, fc, metalness);
        float term = s.d * s.g / (4.0f * s.costwo * s.costwi);
        value = sp_mul(f, term);
        return checknan(same_hemi * 0.5f * s.pdf);
      } else {
        V3 tx = tex_rgb(sc, mat.diffuse, in.uv, in.fp);
        V3 dm = v3(mat.diffuse_mul[0], mat.diffuse_mul[1], mat.diffuse_mul[2]);
        float term = oren_nayar_term(roughness, wo, wi);
        value = from_surface_color((tx * dm) * term);
        return checknan(same_hemi * 0.5f * fabsf(wi.z) * INV_PI);
      }
    }
  }
}

// cosine-weighted hemisphere sample (mat_lambert_sample_value.rcall:19-29, mat_uber_sample_value.rcall:58-63)
inline V3 cosine_sample(float rx, float ry, float woz) {
  float t = TWO_PI * rx;
  float r = sqrtf(ry);
  V3 wi;
  wi.x = r * glz_cosf(t);
  wi.y = r * glz_sinf(t);
  wi.z = sqrtf(gmax(0.0f, 1.0f - wi.x * wi.x - wi.y * wi.y));
  wi.z *= gsign(woz);
  return wi;
}

// sample: returns pdf, fills value and wiW (left untouched where the shader leaves them stale)
float bsdf_sample(const Scene& sc, const BsdfIn& in, V3 r, Sp& value, V3& wiW) {
  const RTMaterial& mat = sc.rt_materials[in.material_id];
  switch (mat.bsdf_index) {
    case 4: {   // mat_lambert_sample_value.rcall:31-41
      V3 wo = to_shading_space(in.woW, in.sh);
      V3 wi = cosine_sample(r.x, r.y, wo.z);
      float pdf = fabsf(wi.z) * INV_PI;
      wiW = normalize(to_world_space(wi, in.sh));
      V3 tx = tex_rgb(sc, mat.diffuse, in.uv, in.fp);
      V3 dm = v3(mat.diffuse_mul[0], mat.diffuse_mul[1], mat.diffuse_mul[2]);
      value = from_surface_color((tx * dm) * INV_PI);
      return pdf;
    }
    case 6: {   // mat_mirror_sample_value.rcall:16-34
      V3 wo = to_shading_space(in.woW, in.sh);
      V3 wi = v3(-wo.x, -wo.y, wo.z);
      Sp fresnel = fresnel_conductor(wo.z, mat.metal_ior, mat.metal_fresnel);
      wiW = normalize(to_world_space(wi, in.sh));
      value = sp_div(fresnel, fabsf(wo.z));
      return 1.0f;
    }
    case 8: {   // mat_glass_sample_value.rcall:34-56
      V3 wo = to_shading_space(in.woW, in.sh);
      float costi = wo.z;
      float from_outside = gstep(0.0f, costi);
      float etai = gmix(mat.ior_dielectric, DEFAULT_IOR, from_outside);
      float etat = gmix(DEFAULT_IOR, mat.ior_dielectric, from_outside);
      costi = gmix(fabsf(costi), costi, from_outside);
      float fresnel = fresnel_dielectric(costi, etai, etat);
      V3 wi;
      float pdf;
      if (r.z < fresnel) {
        wi = v3(-wo.x, -wo.y, wo.z);
        float eval = fresnel / fabsf(wi.z);
        value = sp_mul(sp_uniform(1.0f), eval);
        pdf = fresnel;
      } else {
        wi = refract(wo, v3(0.0f, 0.0f, gsign(wo.z)), etai / etat);   // Q7
        float eval = (1.0f - fresnel) * (etai * etai) / (etat * etat * fabsf(wi.z));
        value = sp_mul(sp_uniform(1.0f), eval);
        pdf = 1.0f - fresnel;
      }
      wiW = to_world_space(wi, in.sh);
      return pdf;
    }
    case 10: {  // mat_metal_sample_value.rcall:21-49
      V3 wo = to_shading_space(in.woW, in.sh);
      float rough = tex_r(sc, mat.roughness, in.uv, in.fp);
      V2 a = to_anisotropic(rough * mat.roughness_mul, mat.anisotropy);
      V3 wh = normalize(ggx_sample_wh(wo, V2{r.x, r.y}, a));
      V3 wi = -normalize(reflect(wo, wh));
      float costwo = fabsf(wo.z), costwi = fabsf(wi.z);
      if (wo.z * wi.z > 0.0f) {
        float d = ggx_d(wh, a);
        float g = ggx_g(wo, wi, a);
        float ggxpdf = ggx_pdf(d, a, wo, wh);
        Sp F = fresnel_conductor(dot(wi, wh), mat.metal_ior, mat.metal_fresnel);
        float term = d * g / (4.0f * costwo * costwi);
        float pdf = ggxpdf / (4.0f * dot(wo, wh));
        value = sp_mul(F, term);
        wiW = normalize(to_world_space(wi, in.sh));
        return checknan(pdf);
      }
      return 0.0f;
    }
    case 12: {  // mat_frosted_sample_value.rcall:21-71
      V3 wo = to_shading_space(in.woW, in.sh);
      float rough = tex_r(sc, mat.roughness, in.uv, in.fp);
      V2 a = to_anisotropic(rough * mat.roughness_mul, mat.anisotropy);
      V3 wh = normalize(ggx_sample_wh(wo, V2{r.x, r.y}, a));
      float from_outside = gstep(0.0f, wo.z);
      float etai = gmix(mat.ior_dielectric, DEFAULT_IOR, from_outside);
      float etat = gmix(DEFAULT_IOR, mat.ior_dielectric, from_outside);
      float eta = etai / etat;
      V3 wi;
      float pdf_out;
      if (r.z < 0.5f) {
        wi = -normalize(reflect(wo, wh));
        SpecTerms s = spec_terms(wo, wi, wh, a);
        float f = fresnel_dielectric(s.costi, etai, etat);
        float term = s.d * s.g * f / (4.0f * s.costwo * s.costwi);
        value = sp_uniform(term);
        pdf_out = checknan(0.5f * s.pdf);
      } else {
        wi = normalize(refract(wo, wh, eta));
        float dotwowh = dot(wo, wh), dotwiwh = dot(wi, wh);
        float f = fresnel_dielectric(dotwowh, etai, etat);
        float costwo = fabsf(wo.z), costwi = fabsf(wi.z);
        float denom = dotwowh + eta * dotwiwh;
        float d = ggx_d(wh, a);
        float g = ggx_g(wo, wi, a);
        float pdf = ggx_pdf(d, a, wo, wh) * fabsf(eta * eta * dotwiwh) / (denom * denom);
        float term = d * g * (1.0f - f) * fabsf(dotwiwh) * fabsf(dotwowh) / (denom * denom * costwo * costwi);
        value = sp_uniform(term);
        pdf_out = dotwowh * dotwiwh < 0.0f ? checknan(0.5f * pdf) : 0.0f;
      }
      wiW = normalize(to_world_space(wi, in.sh));
      return pdf_out;
    }
    default: {  // 14: mat_uber_sample_value.rcall:21-86
      V3 wo = to_shading_space(in.woW, in.sh);
      float rough_tex = tex_r(sc, mat.roughness, in.uv, in.fp);
      float roughness = rough_tex * mat.roughness_mul;
      V3 wi;
      float pdf_out;
      if (r.z < 0.5f) {
        V2 a = to_anisotropic(roughness * mat.roughness_mul, mat.anisotropy);
        V3 wh = normalize(ggx_sample_wh(wo, V2{r.x, r.y}, a));
        float metalness = tex_r(sc, mat.metalness, in.uv, in.fp) * mat.metalness_mul;
        float from_outside = gstep(0.0f, wo.z);
        float etai = gmix(mat.ior_dielectric, DEFAULT_IOR, from_outside);
        float etat = gmix(DEFAULT_IOR, mat.ior_dielectric, from_outside);
        wi = -normalize(reflect(wo, wh));
        SpecTerms s = spec_terms(wo, wi, wh, a);
        Sp fd = sp_uniform(fresnel_dielectric(s.costi, etai, etat));
        Sp fc = fresnel_conductor(s.costi, mat.metal_ior, mat.metal_fresnel);
        Sp f = sp_mix(fd, fc, metalness);
        float term = s.d * s.g / (4.0f * s.costwo * s.costwi);
        value = sp_mul(f, term);
        pdf_out = checknan(0.5f * s.pdf);
      } else {
        wi = cosine_sample(r.x, r.y, wo.z);
        V3 tx = tex_rgb(sc, mat.diffuse, in.uv, in.fp);
        V3 dm = v3(mat.diffuse_mul[0], mat.diffuse_mul[1], mat.diffuse_mul[2]);
        float term = oren_nayar_term(roughness, wo, wi);
        value = from_surface_color((tx * dm) * term);
        pdf_out = 0.5f * fabsf(wi.z) * INV_PI;
      }
      wiW = normalize(to_world_space(wi, in.sh));
      return pdf_out;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Light callables (light_*_sample_visible.rcall)
// ------------------------------------------------------------------------------------------
struct SampledLight { Sp emission; float pdf; V3 wiW; float distance; };

// sample_marginal / sample_conditional (light_sky_sample_visible.rcall:31-98).  The conditional
// lookups pass INTEGER texel coordinates to a normalised-coordinate REPEAT/NEAREST sampler, so every
// fetch returns texel (0,0) (Q3); cond_lookup() restates exactly that.
inline float cond_lookup(const std::vector<float>& img, int /*x*/, uint32_t /*row*/) { return img[0]; }

void light_sample(const Scene& sc, uint32_t light_index, V3 position, V3 r, float scene_radius, SampledLight& sam) {
  const RTLight& light = sc.rt_lights[light_index];
  switch (light.shader) {
    case 0: {   // light_omni_sample_visible.rcall:14-25
      V3 lp = v3(light.pos[0], light.pos[1], light.pos[2]);
      sam.wiW = normalize(lp - position);
      float d2 = ((lp.x - position.x) * (lp.x - position.x) + (lp.y - position.y) * (lp.y - position.y)) + (lp.z - position.z) * (lp.z - position.z);
      sam.distance = sqrtf(d2);
      sam.pdf = 1.0f;
      sam.emission = sp_div(light.color, d2 / light.intensity);
      return;
    }
    case 1: {   // light_sun_sample_visible.rcall:22-29
      sam.wiW = v3(-light.dir[0], -light.dir[1], -light.dir[2]);
      sam.pdf = 1.0f;
      sam.distance = 2.0f * scene_radius + 1.0f;
      sam.emission = sp_mul(light.color, light.intensity);
      return;
    }
    case 2: {   // light_area_sample_visible.rcall:29-64
      const RTInstance& in = sc.instances[light.instance_id];
      uint32_t ntri = in.index_count / 3;
      uint32_t triangle_id = (uint32_t)gmin(r.x * (float)in.index_count / 3.0f, (float)(ntri - 1));   // (rand*index_count)/3, float math
      triangle_id += in.index_offset / 3;
      const uint32_t* ix = &sc.indices[triangle_id * 3];
      auto pos = [&](uint32_t i) { const glz_vertex& v = sc.vertices[i]; return v3(v.vv[0], v.vv[1], v.vv[2]); };
      V3 v0 = pos(ix[0]), v1 = pos(ix[1]), v2 = pos(ix[2]);
      float triangle_area = 0.5f * 3.0f;   // Q1: vec3.length() is the component count
      float sqr_u = sqrtf(r.y);
      float ru = 1.0f - sqr_u, rv = r.z * sqr_u;
      V3 rp = (ru * v0 + rv * v1) + (1.0f - ru - rv) * v2;
      rp = mat_point(sc.transforms[in.transform_id].m, rp);
      sam.wiW = normalize(position - rp);   // Q2: points away from the light
      float d2 = ((rp.x - position.x) * (rp.x - position.x) + (rp.y - position.y) * (rp.y - position.y)) + (rp.z - position.z) * (rp.z - position.z);
      sam.distance = sqrtf(d2);
      const RTMaterial& mat = sc.rt_materials[in.material_id];
      sam.emission = sp_div(from_surface_color(v3(mat.diffuse_mul[0], mat.diffuse_mul[1], mat.diffuse_mul[2])), d2 / light.intensity);
      float select_pdf = 1.0f / (float)ntri;
      float area_pdf = 1.0f / triangle_area;
      sam.pdf = select_pdf * area_pdf;
      return;
    }
    default: {  // 3: light_sky_sample_visible.rcall:100-135
      float v_pdf, u_pdf;
      uint32_t conditional_index;
      float v, u;
      {   // sample_marginal(rand_sample.y)
        float rnd = r.y;
        int size = (int)sc.marginal_cdf_count, first = 0, len = size;
        while (len > 0) {
          int halff = len >> 1, middle = first + halff;
          if (sc.marginal[middle] <= rnd) { first = middle + 1; len -= halff + 1; } else { len = halff; }
        }
        int off = first - 1; off = off < 0 ? 0 : (off > size - 2 ? size - 2 : off);
        uint32_t offset = (uint32_t)off;
        float cur = sc.marginal[offset], next = sc.marginal[offset + 1];
        float du = rnd - cur;
        if (next - cur > 0.0f) du /= next - cur;
        v_pdf = sc.marginal[sc.marginal_cdf_count + offset] / sc.marginal_integral;
        conditional_index = offset;
        v = ((float)offset + du) / (float)sc.marginal_cdf_count;
      }
      {   // sample_conditional(rand_sample.x, conditional_index)
        float rnd = r.x;
        int size = (int)sc.conditional_cdf_count, first = 0, len = size;
        while (len > 0) {
          int halff = len >> 1, middle = first + halff;
          if (cond_lookup(sc.cond_cdf, middle, conditional_index) <= rnd) { first = middle + 1; len -= halff + 1; } else { len = halff; }
        }
        int off = first - 1; off = off < 0 ? 0 : (off > size - 2 ? size - 2 : off);
        uint32_t offset = (uint32_t)off;
        float cur = cond_lookup(sc.cond_cdf, (int)offset, conditional_index);
        float next = cond_lookup(sc.cond_cdf, (int)offset + 1, conditional_index);
        float du = rnd - cur;
        if (next - cur > 0.0f) du /= next - cur;
        u_pdf = cond_lookup(sc.cond_values, (int)offset, conditional_index) / sc.marginal[sc.conditional_integral_offset + conditional_index];
        u = ((float)offset + du) / (float)sc.conditional_cdf_count;
      }
      float pdf = u_pdf * v_pdf;
      float theta = v * PI;
      float sint = glz_sinf(theta);
      if (pdf > 0.0f && sint != 0.0f) {
        float phi = u * TWO_PI;
        float cost = glz_cosf(theta), cosp = glz_cosf(phi), sinp = glz_sinf(phi);
        sam.pdf = pdf / (2.0f * PI * PI * sint);
        V3 wi = v3(sint * cosp, sint * sinp, cost);
        sam.wiW = normalize(mat_dir(sc.sky_obj2world, wi));   // vec4 normalize with w = 0
        sam.distance = 2.0f * scene_radius + 1.0f;
        V3 tx = tex_rgb(sc, sc.sky_tex_id, V2{u, v}) * sc.sky_intensity;
        sam.emission = from_illuminant_color(tx);
      } else {
        sam.pdf = 0.0f;
      }
      return;
    }
  }
}

// ------------------------------------------------------------------------------------------
// RNG (random.glsl:7-57)
// ------------------------------------------------------------------------------------------
inline uint32_t pcg_hash(uint32_t seed) {
  uint32_t state = seed * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float rnd(uint32_t& state) {
  state = pcg_hash(state);
  uint32_t flt = 0x3F800000u | (state & 0x007FFFFFu);
  float f; memcpy(&f, &flt, 4);
  return f - 1.0f;
}
inline V3 rnd3(uint32_t& s) { float a = rnd(s), b = rnd(s), c = rnd(s); return v3(a, b, c); }

// host seed stream (build-defined): xoshiro128++ seeded through SplitMix64
struct Xoshiro128pp {
  uint32_t s[4];
  explicit Xoshiro128pp(uint64_t seed) {
    auto next = [&seed]() { seed += 0x9E3779B97F4A7C15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    uint64_t a = next(), b = next();
    s[0] = (uint32_t)a; s[1] = (uint32_t)(a >> 32); s[2] = (uint32_t)b; s[3] = (uint32_t)(b >> 32);
  }
  static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
  uint32_t next_u32() {
    uint32_t result = rotl(s[0] + s[3], 7) + s[0];
    uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 11);
    return result;
  }
};

// WorkScheduler (raytracer.rs:1168-1206)
struct WorkScheduler {
  struct Area { float a0, a1, b0, b1; };   // (area.0, area.1)
  std::vector<Area> current, next;
  WorkScheduler() { current.push_back(Area{0, 0, 1, 1}); }
  void take(float out[2]) {
    for (;;) {
      if (!current.empty()) {
        Area ar = current.back();
        current.pop_back();
        float mx = (ar.a0 + ar.b0) / 2.0f, my = (ar.a1 + ar.b1) / 2.0f;
        next.push_back(Area{ar.a0, ar.a1, mx, my});
        next.push_back(Area{mx, my, ar.b0, ar.b1});
        next.push_back(Area{mx, ar.a1, ar.b0, my});
        next.push_back(Area{ar.a0, my, mx, ar.b1});
        out[0] = mx; out[1] = my;
        return;
      }
      current.insert(current.end(), next.begin(), next.end());
      next.clear();
    }
  }
};

// ------------------------------------------------------------------------------------------
// Renderer (raytracer.rs draw/draw_frame + path_trace.rgen)
// ------------------------------------------------------------------------------------------
struct PTLastVertex { Sp importance; float wi[4]; float hit[4]; };

struct Renderer {
  Scene* scene;
  uint32_t w, h;
  int integrator = GLZ_PATH_TRACE;
  uint32_t pt_steps = 6;
  float exposure;
  uint64_t seed = 0;
  glz_camera camera;
  float push[32];
  std::vector<PTLastVertex> last;
  std::vector<float> cone;   // ray-cone width at the ray origin, per pixel (texture LOD)
  int lod_mode = 0;          // 0 = level 0 (the reference), 1 = ray cones, 2 = ray cones with an anisotropic footprint
  std::vector<float> cumulative, out32;
  Xoshiro128pp rng{0};
  WorkScheduler sched;
  uint64_t launches = 0;
  bool request_new_frame = true;
  int threads = 1;
  bool count = false;
  // Optional subset of 64x64-pixel tiles to render (row-major tile ids; empty = the whole frame).  Pixels are independent
  // (RNG from absolute pixel coordinates, path_trace.rgen:143-147), so a tile rendered alone equals the same tile of a full
  // render: full-size frames are spot-checked against the product on a sample of tiles in seconds.
  std::vector<uint32_t> tiles;
  std::atomic<uint64_t> c_closest_nodes{0}, c_closest_tris{0}, c_shadow_nodes{0}, c_shadow_tris{0}, c_hits{0}, c_closest_rays{0}, c_shadow_rays{0};
};

struct FrameConsts { uint32_t seed; float off[2]; };

struct HitData { V3 point, shading_normal, geometric_normal, dpdu, dpdv; V2 uv; uint32_t material_id; float distance; };

// raytrace_hit.rchit:30-71
// Ray cones (Akenine-Moeller et al.): the cone is `width` wide where the ray meets the triangle; a texture of W x H texels over a
// triangle with texture-space area A_uv and world area A_w is minified by sqrt(A_uv W H / A_w) texels per unit length and the
// footprint on the surface is width / |cos|:  level = 0.5 log2(A_uv / A_w * width^2 / cos^2) + 0.5 log2(W H).
// Edges and the geometric normal are taken to world space when the instance's transform is not the identity (bitwise).
// mode 2 (anisotropic): the footprint is `width` across and width / |cos| along the projection m of the ray direction onto the
// surface; taps = ceil(min(1 / |cos|, 16)) probes along m, each at the level of a footprint width / |cos| / taps wide; m is written
// in the triangle's edges (least squares: it lies in their plane) to get the footprint's long axis in texture space.
TexFootprint ray_cone_footprint(const Scene& sc, const Tri& tr, V3 direction, float width, int mode) {
  TexFootprint fp;
  const RTInstance& in = sc.instances[tr.instance];
  uint32_t triangle_id = in.index_offset / 3 + tr.prim;
  const uint32_t* ix = &sc.indices[triangle_id * 3];
  const glz_vertex &a = sc.vertices[ix[0]], &b = sc.vertices[ix[1]], &c = sc.vertices[ix[2]];
  V3 e1 = v3(b.vv[0], b.vv[1], b.vv[2]) - v3(a.vv[0], a.vv[1], a.vv[2]), e2 = v3(c.vv[0], c.vv[1], c.vv[2]) - v3(a.vv[0], a.vv[1], a.vv[2]);
  const float* dv = &sc.derivatives[(size_t)triangle_id * 12];
  V3 n = v3(dv[0], dv[1], dv[2]);
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float* M = sc.transforms[in.transform_id].m;
  if (memcmp(M, ident, 64) != 0) {
    e1 = mat_dir(M, e1);
    e2 = mat_dir(M, e2);
    n = mat_tdir(sc.w2o[in.transform_id].data(), n);
  }
  V3 cr = cross(e1, e2);
  float area2 = sqrtf(dot(cr, cr));
  float uva2 = fabsf((b.vt[0] - a.vt[0]) * (c.vt[1] - a.vt[1]) - (c.vt[0] - a.vt[0]) * (b.vt[1] - a.vt[1]));
  float nn = dot(n, n), nd = dot(n, direction);
  float cosv = fabsf(nd) / sqrtf(nn);
  float x = ((uva2 / area2) * (width * width)) / (cosv * cosv);
  if (!(x >= 1.17549435e-38f && x <= 3.4e38f)) return fp;
  fp.lod_base = 0.5f * glz_log2f(x);
  if (mode != 2) return fp;
  float ratio = 1.0f / cosv;
  ratio = ratio < 16.0f ? ratio : 16.0f;
  float taps = -glz_floorf(-ratio);            // ceil
  V3 m = direction - n * (nd / nn);
  float mm = dot(m, m);
  if (!(taps > 1.0f) || !(mm > 0.0f)) return fp;
  float g11 = dot(e1, e1), g12 = dot(e1, e2), g22 = dot(e2, e2), r1 = dot(m, e1), r2 = dot(m, e2);
  float det = g11 * g22 - g12 * g12;
  float ca = (r1 * g22 - r2 * g12) / det, cb = (r2 * g11 - r1 * g12) / det;
  float len = (width / cosv) / sqrtf(mm);
  float du = (ca * (b.vt[0] - a.vt[0]) + cb * (c.vt[0] - a.vt[0])) * len;
  float dvv = (ca * (b.vt[1] - a.vt[1]) + cb * (c.vt[1] - a.vt[1])) * len;
  if (!(fabsf(du) <= 3.4e38f) || !(fabsf(dvv) <= 3.4e38f)) return fp;
  fp.du = du;
  fp.dv = dvv;
  fp.taps = (uint32_t)taps;
  fp.lod_base = fp.lod_base - glz_log2f(taps);
  return fp;
}

// fp: the default (the reference: level 0) or the ray-cone footprint (ray_cone_footprint above)
void closest_hit_shader(const Scene& sc, const Tri& tr, float t, float u, float v, HitData& hit, const TexFootprint& fp = TexFootprint()) {
  const RTInstance& in = sc.instances[tr.instance];
  uint32_t triangle_id = in.index_offset / 3 + tr.prim;
  float b0 = 1.0f - u - v, b1 = u, b2 = v;
  const uint32_t* ix = &sc.indices[triangle_id * 3];
  const glz_vertex &a = sc.vertices[ix[0]], &b = sc.vertices[ix[1]], &c = sc.vertices[ix[2]];
  auto mix3 = [&](const float* x, const float* y, const float* z) { return (v3(x[0], x[1], x[2]) * b0 + v3(y[0], y[1], y[2]) * b1) + v3(z[0], z[1], z[2]) * b2; };
  hit.point = mix3(a.vv, b.vv, c.vv);
  hit.uv = V2{(a.vt[0] * b0 + b.vt[0] * b1) + c.vt[0] * b2, (a.vt[1] * b0 + b.vt[1] * b1) + c.vt[1] * b2};
  const float* dv = &sc.derivatives[(size_t)triangle_id * 12];
  hit.geometric_normal = v3(dv[0], dv[1], dv[2]);
  hit.dpdu = v3(dv[4], dv[5], dv[6]);
  hit.dpdv = v3(dv[8], dv[9], dv[10]);
  hit.shading_normal = mix3(a.vn, b.vn, c.vn);
  hit.material_id = in.material_id;
  const RTMaterial& mat = sc.rt_materials[hit.material_id];
  if (mat.normal != 0) {
    V4 tx = texture_lod(sc, mat.normal, hit.uv.x, hit.uv.y, fp);
    ShadingSpace old;
    old.s = normalize(hit.dpdu);
    old.n = hit.shading_normal;
    old.t = normalize(cross(old.n, old.s));
    V3 nt = v3(tx.x * 2.0f - 1.0f, tx.y * 2.0f - 1.0f, tx.z * 2.0f - 1.0f);
    hit.shading_normal = normalize(to_world_space(nt, old));
    hit.shading_normal = hit.shading_normal * gsign(dot(hit.geometric_normal, hit.shading_normal));
  }
  hit.distance = t;
  const float* M = sc.transforms[in.transform_id].m;
  const float* Wi = sc.w2o[in.transform_id].data();
  hit.point = mat_point(M, hit.point);
  hit.dpdu = mat_point(M, hit.dpdu);   // Q8: w = 1
  hit.dpdv = mat_point(M, hit.dpdv);
  hit.geometric_normal = mat_tdir(Wi, hit.geometric_normal);
  hit.shading_normal = mat_tdir(Wi, hit.shading_normal);
}

// ORC_DEBUG_PIXEL="x,y" in the environment: the rays that pixel traces and what comes back, one line each on stderr, floats as
// hexadecimal literals (tools/gpu_fuzz_diag.py feeds the same rays to the HIP tracer's debug hooks)
static bool debug_pixel(uint32_t px, uint32_t py) {
  static int want[2] = {-2, -2};
  if (want[0] == -2) {
    want[0] = want[1] = -1;
    if (const char* e = getenv("ORC_DEBUG_PIXEL")) {
      if (!strcmp(e, "all")) want[0] = want[1] = -3;   // every pixel (single-threaded renders only: the lines of different pixels would mix)
      else sscanf(e, "%d,%d", &want[0], &want[1]);
    }
  }
  if (want[0] == -3) return true;
  return (int)px == want[0] && (int)py == want[1];
}
void trace_ray_closest(Renderer& R, V3 o, V3 d, Hit& h) {
  const Scene& sc = *R.scene;
  h = trace_closest(sc, o, d, 0.0001f, INF);
  if (R.count) {
    R.c_closest_rays++;
    if (!sc.ext_nodes.empty()) {
      Counters c; float t; uint32_t id;
      ext_trace(sc, o, d, 0.0001f, INF, false, c, t, id);
      R.c_closest_nodes += c.nodes; R.c_closest_tris += c.tris;
    }
    if (h.valid) R.c_hits++;
  }
}
bool trace_ray_shadow(Renderer& R, V3 o, V3 d, float tmax) {
  const Scene& sc = *R.scene;
  bool occluded = trace_any(sc, o, d, 0.001f, tmax);
  if (R.count) {
    R.c_shadow_rays++;
    if (!sc.ext_nodes.empty()) {
      Counters c; float t; uint32_t id;
      ext_trace(sc, o, d, 0.001f, tmax, true, c, t, id);
      R.c_shadow_nodes += c.nodes; R.c_shadow_tris += c.tris;
    }
  }
  return occluded;
}

// path_trace.rgen:135-239 for one pixel of one launch
void raygen(Renderer& R, const FrameConsts& fc, uint32_t px, uint32_t py) {
  const Scene& sc = *R.scene;
  const uint32_t lights_no = sc.lights_no;
  if (lights_no == 0) return;                                            // :137-141
  const bool direct_only = R.integrator == GLZ_DIRECT;
  const size_t path_id = (size_t)py * R.w + px;
  float* cum = &R.cumulative[path_id * 4];
  float* out = &R.out32[path_id * 4];
  auto update_result = [&](const Sp& radiance) {                         // :126-133
    V3 c = sp_rgb(radiance);
    cum[0] += c.x; cum[1] += c.y; cum[2] += c.z;
    out[0] = cum[0] * R.exposure / cum[3];
    out[1] = cum[1] * R.exposure / cum[3];
    out[2] = cum[2] * R.exposure / cum[3];
    out[3] = 1.0f;
  };
  cum[3] += 1.0f;                                                        // update_count :119-124
  uint32_t rng = pcg_hash(fbits((float)fc.seed) ^ pcg_hash(fbits((float)px) ^ pcg_hash(fbits((float)py))));   // :143, Q11
  PTLastVertex& last = R.last[path_id];
  const float pxf = (float)px + fc.off[0], pyf = (float)py + fc.off[1];
  const float uvx = pxf / (float)R.w, uvy = pyf / (float)R.h;
  const float ndcx = -1.0f + 2.0f * uvx, ndcy = -1.0f + 2.0f * uvy;
  Sp importance;
  V3 origin, direction;
  if (direct_only || last.hit[3] == 0.0f) {
    const float* c2w = R.push;
    const float* s2c = R.push + 16;
    const bool persp = R.camera.type == GLZ_CAMERA_PERSPECTIVE;
    {   // ray_origin :47-56
      float is_ortho = gstep(0.5f, persp ? 0.0f : 1.0f);
      float fx = ndcx * is_ortho, fy = ndcy * is_ortho;
      // c2w * vec4(fx, fy, 0, 1); the z column contributes an exact zero and is dropped
      origin = v3((c2w[0] * fx + c2w[4] * fy) + c2w[12], (c2w[1] * fx + c2w[5] * fy) + c2w[13], (c2w[2] * fx + c2w[6] * fy) + c2w[14]);
    }
    {   // ray_dir :58-73
      float is_persp = gstep(0.5f, persp ? 1.0f : 0.0f);
      float fx = ndcx * is_persp, fy = ndcy * is_persp;
      V3 target = v3(((s2c[0] * fx + s2c[4] * fy) + s2c[8]) + s2c[12], ((s2c[1] * fx + s2c[5] * fy) + s2c[9]) + s2c[13],
                     ((s2c[2] * fx + s2c[6] * fy) + s2c[10]) + s2c[14]);
      V3 nt = normalize(target);
      // normalize(vec4(c2w * vec4(nt, 0))): the 4th component is row 3 of c2w . (nt,0)
      float dx = (c2w[0] * nt.x + c2w[4] * nt.y) + c2w[8] * nt.z;
      float dy = (c2w[1] * nt.x + c2w[5] * nt.y) + c2w[9] * nt.z;
      float dz = (c2w[2] * nt.x + c2w[6] * nt.y) + c2w[10] * nt.z;
      float dw = (c2w[3] * nt.x + c2w[7] * nt.y) + c2w[11] * nt.z;
      float inv = 1.0f / sqrtf(((dx * dx + dy * dy) + dz * dz) + dw * dw);
      direction = v3(dx * inv, dy * inv, dz * inv);
    }
    importance = sp_uniform(1.0f);
  } else {
    origin = v3(last.hit[0], last.hit[1], last.hit[2]);
    direction = v3(last.wi[0], last.wi[1], last.wi[2]);
    importance = last.importance;
  }
  Hit h;
  trace_ray_closest(R, origin, direction, h);                            // :169
  const bool dbg = debug_pixel(px, py);
  if (dbg)
    fprintf(stderr, "orc closest o %a %a %a d %a %a %a bounce %g -> valid %d t %a u %a v %a world_id %u instance %u pixel %u %u\n", origin.x, origin.y, origin.z, direction.x, direction.y,
            direction.z, last.hit[3], (int)h.valid, h.t, h.u, h.v, h.valid ? sc.tris[h.tri].world_id : 0u, h.valid ? sc.tris[h.tri].instance : 0u, px, py);
  if (!h.valid) {                                                        // :170-179
    if ((last.hit[3] == 0.0f || last.wi[3] == 1.0f) && sc.sky_tex_id > 0) {
      V3 wv = normalize(mat_dir(sc.sky_world2obj, direction));           // sky_radiance :75-82
      float phi = glz_atan2f(wv.y, wv.x);
      float theta = glz_acosf(wv.z);
      V3 texel = tex_rgb(sc, sc.sky_tex_id, V2{phi * INV_2PI, theta * INV_PI});
      update_result(sp_mul(importance, from_illuminant_color(texel)));
    }
    last.hit[3] = 0.0f;
    return;
  }
  HitData hit;
  TexFootprint fp;
  float cone_w = 0.0f;
  if (R.lod_mode != 0) {
    // one pixel of the image plane at unit distance (perspective: the cone's spread) or in world units (orthographic: its width)
    const float pixel = 2.0f * fabsf(R.push[16 + 5]) / (float)R.h;
    const bool persp = R.camera.type == GLZ_CAMERA_PERSPECTIVE;
    const float spread = persp ? pixel : 0.0f, width0 = persp ? 0.0f : pixel;
    const bool fresh = direct_only || last.hit[3] == 0.0f;
    cone_w = (fresh ? width0 : R.cone[path_id]) + spread * h.t;
    fp = ray_cone_footprint(sc, sc.tris[h.tri], direction, cone_w, R.lod_mode);
  }
  closest_hit_shader(sc, sc.tris[h.tri], h.t, h.u, h.v, hit, fp);
  const RTMaterial& material = sc.rt_materials[hit.material_id];
  V3 woW = -direction;
  ShadingSpace matrix = new_shading_space(hit.dpdu, hit.shading_normal);
  BsdfIn bin;
  bin.woW = woW; bin.uv = hit.uv; bin.sh = matrix; bin.material_id = hit.material_id; bin.fp = fp;
  if (material.is_specular == 0) {                                       // :183-189, direct_light :84-117
    Sp radiance_light = sp_uniform(0.0f);
    float weight_light = 1.0f;
    uint32_t light_index = (uint32_t)gmin(rnd(rng) * (float)lights_no, (float)(lights_no - 1));
    V3 r3 = rnd3(rng);
    SampledLight sam;
    sam.pdf = 0.0f;
    light_sample(sc, light_index, hit.point, r3, sc.meta.scene_radius, sam);
    if (sam.pdf > 0.0f) {
      bin.wiW = sam.wiW;
      float rs = rnd(rng);
      Sp value = sp_uniform(0.0f);
      float bpdf = bsdf_value(sc, bin, rs, value);
      if (dbg) fprintf(stderr, "orc light %u pdf %a distance %a bsdf pdf %a material %u\n", light_index, sam.pdf, sam.distance, bpdf, hit.material_id);
      if (bpdf > 0.0f) {
        bool shadow_ray_hit = trace_ray_shadow(R, hit.point, sam.wiW, sam.distance - 1e-3f);
        if (dbg)
          fprintf(stderr, "orc shadow o %a %a %a d %a %a %a tmax %a -> occluded %d\n", hit.point.x, hit.point.y, hit.point.z, sam.wiW.x, sam.wiW.y, sam.wiW.z,
                  sam.distance - 1e-3f, (int)shadow_ray_hit);
        weight_light *= shadow_ray_hit ? 0.0f : 1.0f;
        weight_light *= fabsf(dot(sam.wiW, hit.shading_normal)) / sam.pdf;
        radiance_light = sp_mul(value, sam.emission);
      }
    }
    Sp radiance = sp_mul(radiance_light, weight_light);
    radiance = sp_mul(radiance, (float)lights_no);
    radiance = sp_mul(radiance, importance);
    update_result(radiance);
    last.wi[3] = 0.0f;
  } else {
    last.wi[3] = 1.0f;
  }
  if (direct_only) return;
  if (last.hit[3] > (float)(R.pt_steps / 2)) {                           // RUSSIAN_ROULETTE :197-210
    float kill_prob = gmax(0.05f, 1.0f - sp_luminance(importance));
    float roll = rnd(rng);
    if (roll < kill_prob) { last.hit[3] = 0.0f; return; }
    importance = sp_mul(importance, 1.0f / (1.0f - kill_prob));
  }
  V3 r3 = rnd3(rng);
  Sp value = sp_uniform(0.0f);
  V3 wiW = v3(0, 0, 0);
  float pdf = bsdf_sample(sc, bin, r3, value, wiW);                      // :212-218
  if (pdf == 0.0f) { last.hit[3] = 0.0f; return; }
  float weight = fabsf(dot(wiW, hit.shading_normal));
  weight /= pdf;
  last.importance = sp_mul(importance, sp_mul(value, weight));           // :226
  last.hit[0] = hit.point.x; last.hit[1] = hit.point.y; last.hit[2] = hit.point.z;
  last.wi[0] = wiW.x; last.wi[1] = wiW.y; last.wi[2] = wiW.z;
  if (last.hit[3] < (float)R.pt_steps) last.hit[3] += 1.0f; else last.hit[3] = 0.0f;   // :230-237
  if (R.lod_mode != 0) R.cone[path_id] = cone_w;
}

void renderer_reset(Renderer& R) {
  R.last.assign((size_t)R.w * R.h, PTLastVertex{});
  R.cone.assign((size_t)R.w * R.h, 0.0f);
  R.cumulative.assign((size_t)R.w * R.h * 4, 0.0f);
  R.out32.assign((size_t)R.w * R.h * 4, 0.0f);
  R.rng = Xoshiro128pp(R.seed);
  R.sched = WorkScheduler();
  R.launches = 0;
  R.request_new_frame = false;
}

void renderer_launch(Renderer& R) {
  if (R.request_new_frame) renderer_reset(R);
  FrameConsts fc;
  fc.seed = R.rng.next_u32();
  R.sched.take(fc.off);
  int nt = std::max(1, R.threads);
  // work items: rows of the frame, or (tile, row) pairs when only a subset of tiles is rendered
  const uint32_t tiles_x = (R.w + 63) / 64;
  const uint32_t n_items = R.tiles.empty() ? R.h : (uint32_t)R.tiles.size() * 64u;
  auto run_item = [&](uint32_t item) {
    if (R.tiles.empty()) {
      for (uint32_t x = 0; x < R.w; ++x) raygen(R, fc, x, item);
      return;
    }
    const uint32_t t = R.tiles[item / 64u], y = (t / tiles_x) * 64u + item % 64u, x0 = (t % tiles_x) * 64u;
    if (y >= R.h) return;
    for (uint32_t x = x0; x < std::min(R.w, x0 + 64u); ++x) raygen(R, fc, x, y);
  };
  if (nt == 1) {
    for (uint32_t i = 0; i < n_items; ++i) run_item(i);
  } else {
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t)
      pool.emplace_back([&] {
        for (;;) {
          uint32_t i = next.fetch_add(1);
          if (i >= n_items) break;
          run_item(i);
        }
      });
    for (auto& th : pool) th.join();
  }
  R.launches++;
}

// linear -> sRGB 8 bit, what the R8G8B8A8_SRGB blit + export does (raytracer.rs:576-584, memory.rs:269-483) [ext].
// Stated as a rule that needs no pow() per pixel so that every implementation agrees on every byte:
// q = round(255 * OETF(c)) = the number of k in 1..255 whose threshold T_k = (float) EOTF((k - 0.5) / 255) is <= c.
// This is a plain count over the rule; the product searches an uploaded table instead.
struct Srgb8Rule {
  float t[256];
  Srgb8Rule() {
    t[0] = 0.0f;
    for (int k = 1; k <= 255; ++k) {
      double v = ((double)k - 0.5) / 255.0;
      t[k] = (float)(v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4));
    }
  }
};
uint8_t to_srgb8(float c) {
  static const Srgb8Rule rule;
  if (!(c > 0.0f)) return 0;
  int q = 0;
  for (int k = 1; k <= 255; ++k)
    if (c >= rule.t[k]) ++q;
  return (uint8_t)q;
}

void scene_finish(Scene& sc, const std::vector<glz_light>& parsed_lights) {
  for (int i = 0; i < 256; ++i) {
    double c = i / 255.0;
    sc.srgb_lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
  }
  sc.w2o.clear();
  for (const glz_transform& t : sc.transforms) {
    M4 inv;
    if (!m4_invert(m4_from_f32(t.m), inv)) inv = m4_identity();
    std::vector<float> f(16);
    m4_to_f32(inv, f.data());
    sc.w2o.push_back(f);
  }
  build_rt_instances(sc);
  build_rt_materials(sc);
  build_rt_lights(sc, parsed_lights);
  build_derivatives(sc);
  build_sky(sc);
  build_accel(sc);
}

}  // namespace

// ==========================================================================================
// C interface (ctypes)
// ==========================================================================================
extern "C" {

void* orc_scene_create(const glz_scene_desc* d) {
  Scene* sc = new Scene();
  sc->vertices.assign(d->vertices, d->vertices + d->n_vertices);
  sc->indices.assign(d->indices, d->indices + d->n_indices);
  sc->meshes.assign(d->meshes, d->meshes + d->n_meshes);
  if (d->n_transforms) sc->transforms.assign(d->transforms, d->transforms + d->n_transforms);
  else { glz_transform t{}; t.m[0] = t.m[5] = t.m[10] = t.m[15] = 1; sc->transforms.push_back(t); }
  sc->mesh_instances.assign(d->instances, d->instances + d->n_instances);
  if (d->n_materials) sc->materials.assign(d->materials, d->materials + d->n_materials);
  else { glz_material m{}; m.mtype = GLZ_MAT_LAMBERT; m.diffuse_mul[0] = m.diffuse_mul[1] = m.diffuse_mul[2] = 255; m.ior = 1.46f; m.roughness_mul = 1.0f; sc->materials.push_back(m); }
  for (uint32_t i = 0; i < d->n_textures; ++i) {
    Tex t; t.format = d->textures[i].format; t.w = d->textures[i].width; t.h = d->textures[i].height;
    size_t n = (size_t)t.w * t.h * (t.format == GLZ_TEX_GRAY ? 1 : 4);
    t.px.assign(d->textures[i].pixels, d->textures[i].pixels + n);
    sc->textures.push_back(t);
  }
  if (sc->textures.empty()) { Tex t; t.format = GLZ_TEX_RGBA_SRGB; t.w = t.h = 1; t.px.assign(4, 255); sc->textures.push_back(t); }
  if (d->camera) sc->camera = *d->camera;
  else { glz_camera c{}; c.type = 0; c.target[2] = 100; c.up[1] = 1; c.fovx_or_scale = 90.0f * (3.14159265358979323846f / 180.0f); c.near_plane = 1e-3f; c.far_plane = 1e3f; sc->camera = c; }
  if (d->meta) sc->meta = *d->meta;
  else { glz_meta m{}; m.scene_radius = 100; m.exposure = 1; sc->meta = m; }
  std::vector<glz_light> lights(d->lights, d->lights + d->n_lights);
  scene_finish(*sc, lights);
  return sc;
}
void orc_scene_destroy(void* s) { delete (Scene*)s; }

uint32_t orc_scene_lights_no(void* s) { return ((Scene*)s)->lights_no; }
uint64_t orc_scene_world_tris(void* s) { return ((Scene*)s)->tris.size(); }

int64_t orc_read_derivatives(void* s, float* out, int64_t cap_tris) {
  Scene* sc = (Scene*)s;
  int64_t n = (int64_t)sc->derivatives.size() / 12;
  if (out) memcpy(out, sc->derivatives.data(), (size_t)std::min(n, cap_tris) * 48);
  return n;
}
int64_t orc_read_rt_materials(void* s, void* out, int64_t cap) {
  Scene* sc = (Scene*)s;
  int64_t n = (int64_t)sc->rt_materials.size() * 208;
  if (out) memcpy(out, sc->rt_materials.data(), (size_t)std::min(n, cap));
  return n;
}
int64_t orc_read_rt_lights(void* s, void* out, int64_t cap) {
  Scene* sc = (Scene*)s;
  int64_t n = (int64_t)sc->rt_lights.size() * 112;
  if (out) memcpy(out, sc->rt_lights.data(), (size_t)std::min(n, cap));
  return n;
}
// RTSky (36 floats: obj2world, world2obj, tex_id bits, intensity, 2 pad) | header (4) | marginal arrays
int64_t orc_read_sky(void* s, float* out, int64_t cap) {
  Scene* sc = (Scene*)s;
  std::vector<float> buf;
  buf.insert(buf.end(), sc->sky_obj2world, sc->sky_obj2world + 16);
  buf.insert(buf.end(), sc->sky_world2obj, sc->sky_world2obj + 16);
  float f; memcpy(&f, &sc->sky_tex_id, 4); buf.push_back(f);
  buf.push_back(sc->sky_intensity); buf.push_back(0); buf.push_back(0);
  memcpy(&f, &sc->marginal_cdf_count, 4); buf.push_back(f);
  memcpy(&f, &sc->conditional_integral_offset, 4); buf.push_back(f);
  memcpy(&f, &sc->conditional_cdf_count, 4); buf.push_back(f);
  buf.push_back(sc->marginal_integral);
  buf.insert(buf.end(), sc->marginal.begin(), sc->marginal.end());
  if (out) memcpy(out, buf.data(), (size_t)std::min<int64_t>(buf.size(), cap) * 4);
  return (int64_t)buf.size();
}
int64_t orc_read_sky_cond(void* s, float* values, float* cdf) {
  Scene* sc = (Scene*)s;
  if (values) memcpy(values, sc->cond_values.data(), sc->cond_values.size() * 4);
  if (cdf) memcpy(cdf, sc->cond_cdf.data(), sc->cond_cdf.size() * 4);
  return (int64_t)sc->cond_values.size();
}

// product LBVH import for work counting (nodes: 8 words each, tris: 12 floats each, grid: lo[3], cell[3])
void orc_scene_set_ext_bvh(void* s, const uint32_t* nodes, uint64_t n_nodes, const float* tris, uint64_t n_tris, const float* grid_lo, const float* grid_cell) {
  Scene* sc = (Scene*)s;
  sc->ext_nodes.assign(nodes, nodes + n_nodes * 16);
  sc->ext_tris.assign(tris, tris + n_tris * 12);
  for (int k = 0; k < 3; ++k) {
    sc->ext_grid[k] = grid_lo[k];
    sc->ext_grid[3 + k] = grid_cell[k];
    sc->ext_grid[6 + k] = 1.0f / grid_cell[k];   // k_grid_params computes inv_cell the same way
  }
}

void orc_trace_closest(void* s, const float* o, const float* d, uint64_t n, float tmin, float* t_out, uint32_t* tri_out,
                       uint32_t* inst_out, float* u_out, float* v_out) {
  Scene* sc = (Scene*)s;
  for (uint64_t i = 0; i < n; ++i) {
    Hit h = trace_closest(*sc, v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmin, INF);
    t_out[i] = h.valid ? h.t : INF;
    tri_out[i] = h.valid ? sc->tris[h.tri].world_id : 0xFFFFFFFFu;
    inst_out[i] = h.valid ? sc->tris[h.tri].instance : 0xFFFFFFFFu;
    u_out[i] = h.valid ? h.u : 0.0f;
    v_out[i] = h.valid ? h.v : 0.0f;
  }
}
void orc_trace_any(void* s, const float* o, const float* d, const float* tmax, uint64_t n, float tmin, uint8_t* hit_out) {
  Scene* sc = (Scene*)s;
  for (uint64_t i = 0; i < n; ++i)
    hit_out[i] = trace_any(*sc, v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmin, tmax[i]) ? 1 : 0;
}
// brute force over every triangle (no BVH): validates the oracle's own BVH
void orc_trace_closest_brute(void* s, const float* o, const float* d, uint64_t n, float tmin, float* t_out, uint32_t* tri_out) {
  Scene* sc = (Scene*)s;
  for (uint64_t i = 0; i < n; ++i) {
    V3 oo = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dd = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    float best = INF; uint32_t id = 0xFFFFFFFFu;
    for (const Tri& tr : sc->tris) {
      float t, u, v;
      if (!ray_tri(tr, oo, dd, tmin, INF, t, u, v)) continue;
      if (t < best || (t == best && tr.world_id < id)) {
        if (tr.non_opaque && !alpha_pass(*sc, tr, u, v)) continue;
        best = t; id = tr.world_id;
      }
    }
    t_out[i] = best; tri_out[i] = id;
  }
}

void* orc_renderer_create(void* s, uint32_t w, uint32_t h) {
  Renderer* R = new Renderer();
  R->scene = (Scene*)s;
  R->w = w; R->h = h;
  R->camera = R->scene->camera;
  R->exposure = R->scene->meta.exposure;
  push_constants(R->camera, w, h, R->push);
  R->request_new_frame = true;
  return R;
}
void orc_renderer_destroy(void* r) { delete (Renderer*)r; }
void orc_renderer_set_integrator(void* r, int i) { Renderer* R = (Renderer*)r; R->integrator = i; R->request_new_frame = true; }
void orc_renderer_set_depth(void* r, uint32_t d) { Renderer* R = (Renderer*)r; R->pt_steps = d; R->request_new_frame = true; }
void orc_renderer_set_seed(void* r, uint64_t s) { Renderer* R = (Renderer*)r; R->seed = s; R->request_new_frame = true; }
void orc_renderer_set_exposure(void* r, float e) { ((Renderer*)r)->exposure = e; }
void orc_renderer_set_threads(void* r, int t) { ((Renderer*)r)->threads = t; }
void orc_renderer_set_counting(void* r, int on) { ((Renderer*)r)->count = on != 0; }
void orc_renderer_update_camera(void* r, const glz_camera* c) {
  Renderer* R = (Renderer*)r; R->camera = *c; push_constants(R->camera, R->w, R->h, R->push); R->request_new_frame = true;
}
uint32_t orc_renderer_steps_per_sample(void* r) { Renderer* R = (Renderer*)r; return R->integrator == GLZ_DIRECT ? 1 : R->pt_steps; }
void orc_renderer_restart(void* r) { ((Renderer*)r)->request_new_frame = true; }
void orc_renderer_step(void* r, uint32_t n) { for (uint32_t i = 0; i < n; ++i) renderer_launch(*(Renderer*)r); }
// draw(spp): raytracer.rs:615-687
void orc_renderer_draw(void* r, uint64_t spp) {
  Renderer* R = (Renderer*)r;
  R->request_new_frame = true;
  uint64_t n = spp * orc_renderer_steps_per_sample(r);
  for (uint64_t i = 0; i < n; ++i) renderer_launch(*R);
}
void orc_renderer_read_hdr(void* r, float* out) { Renderer* R = (Renderer*)r; if (R->request_new_frame) renderer_reset(*R); memcpy(out, R->cumulative.data(), R->cumulative.size() * 4); }
void orc_renderer_read_result(void* r, float* out) { Renderer* R = (Renderer*)r; if (R->request_new_frame) renderer_reset(*R); memcpy(out, R->out32.data(), R->out32.size() * 4); }
uint8_t orc_to_srgb8(float c) { return to_srgb8(c); }
// texture level of detail: 0 = level 0 (the reference), 1 = ray cones, 2 = anisotropic ray cones (the mip chains are built here).  Restarts.
void orc_renderer_set_texture_lod(void* r, int mode) {
  Renderer* R = (Renderer*)r;
  R->lod_mode = mode;
  if (mode != 0)
    for (Tex& t : R->scene->textures)
      if (t.mips.empty() && (t.w > 1 || t.h > 1)) build_mip_chain(t);
  R->request_new_frame = true;
}
// one level of a texture's generated mip chain (level 0 = the texture): returns the byte count, 0 past the last level
int64_t orc_texture_level(void* s, uint32_t texture, uint32_t level, uint8_t* out, uint32_t* w, uint32_t* h) {
  Scene* sc = (Scene*)s;
  if (texture >= sc->textures.size()) return 0;
  Tex& t = sc->textures[texture];
  if (level > 0 && t.mips.empty() && (t.w > 1 || t.h > 1)) build_mip_chain(t);
  if (level > t.mips.size()) return 0;
  const Tex& l = level == 0 ? t : t.mips[level - 1];
  if (w) *w = l.w;
  if (h) *h = l.h;
  if (out) memcpy(out, l.px.data(), l.px.size());
  return (int64_t)l.px.size();
}
// restrict rendering to `n` 64x64 tiles (row-major ids over ceil(w/64) x ceil(h/64)); n = 0 = whole frame.  Restarts.
void orc_renderer_set_tiles(void* r, const uint32_t* tiles, uint32_t n) {
  Renderer* R = (Renderer*)r;
  const uint32_t total = ((R->w + 63) / 64) * ((R->h + 63) / 64);
  R->tiles.clear();
  for (uint32_t i = 0; i < n; ++i)
    if (tiles[i] < total) R->tiles.push_back(tiles[i]);
  R->request_new_frame = true;
}
void orc_renderer_read_rgba8(void* r, uint8_t* out) {
  Renderer* R = (Renderer*)r;
  if (R->request_new_frame) renderer_reset(*R);
  for (size_t i = 0; i < (size_t)R->w * R->h; ++i) {
    out[4 * i] = to_srgb8(R->out32[4 * i]); out[4 * i + 1] = to_srgb8(R->out32[4 * i + 1]); out[4 * i + 2] = to_srgb8(R->out32[4 * i + 2]);
    out[4 * i + 3] = (uint8_t)(R->out32[4 * i + 3] >= 1.0f ? 255 : 0);
  }
}
void orc_renderer_read_state(void* r, float* out24) { Renderer* R = (Renderer*)r; memcpy(out24, R->last.data(), R->last.size() * sizeof(PTLastVertex)); }
void orc_renderer_push_constants(void* r, float* out32) { memcpy(out32, ((Renderer*)r)->push, 128); }
// constants of launch `i` counted from a restart (fresh streams, does not disturb the renderer)
void orc_launch_constants(uint64_t seed, uint32_t launch, uint32_t* seed_out, float* off) {
  Xoshiro128pp rng(seed);
  WorkScheduler ws;
  uint32_t s = 0; float o[2] = {0, 0};
  for (uint32_t i = 0; i <= launch; ++i) { s = rng.next_u32(); ws.take(o); }
  *seed_out = s; off[0] = o[0]; off[1] = o[1];
}
void orc_renderer_counters(void* r, uint64_t* out7) {
  Renderer* R = (Renderer*)r;
  out7[0] = R->c_closest_rays; out7[1] = R->c_shadow_rays; out7[2] = R->c_closest_nodes; out7[3] = R->c_closest_tris;
  out7[4] = R->c_shadow_nodes; out7[5] = R->c_shadow_tris; out7[6] = R->c_hits;
}

// ---- KAT hooks for the reference's own unit tests -----------------------------------------
void orc_spectrum_from_rgb(float r, float g, float b, int is_light, float* out16) { Sp s = host_from_rgb(r, g, b, is_light != 0); memcpy(out16, s.w, 64); }
void orc_spectrum_to_xyz(const float* sp16, float* out3) { Sp s; memcpy(s.w, sp16, 64); host_to_xyz(s, out3); }
float orc_spectrum_luminance(const float* sp16) { Sp s; memcpy(s.w, sp16, 64); return host_luminance(s); }
void orc_spectrum_from_blackbody(float t, float* out16) { Sp s = host_from_blackbody(t); memcpy(out16, s.w, 64); }
void orc_xyz_to_rgb(const float* xyz, float* rgb) { host_xyz_to_rgb(xyz, rgb); }
void orc_rgb_to_xyz(const float* rgb, float* xyz) { host_rgb_to_xyz(rgb, xyz); }
float orc_fovy(float fovx, float ar) { return cam_fovy(fovx, ar); }
void orc_spectrum_white(float* out16) { memcpy(out16, GLZ_HOST_SPECTRUM_WHITE, 64); }
// device-flavour colour math
void orc_dev_from_surface_color(const float* rgb, float* out16) { Sp s = from_surface_color(v3(rgb[0], rgb[1], rgb[2])); memcpy(out16, s.w, 64); }
void orc_dev_from_illuminant_color(const float* rgb, float* out16) { Sp s = from_illuminant_color(v3(rgb[0], rgb[1], rgb[2])); memcpy(out16, s.w, 64); }
void orc_dev_rgb(const float* sp16, float* out3) { Sp s; memcpy(s.w, sp16, 64); V3 c = sp_rgb(s); out3[0] = c.x; out3[1] = c.y; out3[2] = c.z; }
float orc_dev_luminance(const float* sp16) { Sp s; memcpy(s.w, sp16, 64); return sp_luminance(s); }
// deterministic math
void orc_detmath(int fn, const float* x, const float* y, float* out, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) {
    switch (fn) {
      case 0: out[i] = glz_sinf(x[i]); break;
      case 1: out[i] = glz_cosf(x[i]); break;
      case 2: out[i] = glz_acosf(x[i]); break;
      case 4: out[i] = glz_log2f(x[i]); break;
      default: out[i] = glz_atan2f(y[i], x[i]); break;
    }
  }
}
uint32_t orc_pcg_hash(uint32_t x) { return pcg_hash(x); }
// ---- shading routines on their own, for the tests that check this restatement against mathematics (tests/test_oracle_math.py) ----
// n evaluations of the material's BSDF in the canonical frame (s, t, n) = (x, y, z), so world directions ARE shading-space
// directions: value (16 bins) and pdf of mat_*_value.rcall for (wo, wi, uv, rand)
void orc_bsdf_value(void* s, uint32_t material_id, const float* wo3, const float* wi3, const float* uv2, const float* rand1, uint64_t n,
                    float* value16, float* pdf) {
  Scene* sc = (Scene*)s;
  ShadingSpace sh{v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
  for (uint64_t i = 0; i < n; ++i) {
    BsdfIn in{v3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), v3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), V2{uv2[0], uv2[1]}, sh, material_id};
    Sp value = sp_uniform(0.0f);
    pdf[i] = bsdf_value(*sc, in, rand1[i], value);
    memcpy(value16 + 16 * i, value.w, 64);
  }
}
// n samples of mat_*_sample_value.rcall for (wo, uv, rand3): sampled direction, value (16 bins), pdf
void orc_bsdf_sample(void* s, uint32_t material_id, const float* wo3, const float* uv2, const float* rand3, uint64_t n, float* wi3, float* value16,
                     float* pdf) {
  Scene* sc = (Scene*)s;
  ShadingSpace sh{v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
  for (uint64_t i = 0; i < n; ++i) {
    BsdfIn in{v3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), v3(0, 0, 0), V2{uv2[0], uv2[1]}, sh, material_id};
    Sp value = sp_uniform(0.0f);
    V3 wi = v3(0, 0, 0);
    pdf[i] = bsdf_sample(*sc, in, v3(rand3[3 * i], rand3[3 * i + 1], rand3[3 * i + 2]), value, wi);
    wi3[3 * i] = wi.x; wi3[3 * i + 1] = wi.y; wi3[3 * i + 2] = wi.z;
    memcpy(value16 + 16 * i, value.w, 64);
  }
}
// n samples of light_*_sample_visible.rcall for RTLight `light_index`: direction, distance, pdf, emission (16 bins)
void orc_light_sample(void* s, uint32_t light_index, const float* pos3, const float* rand3, uint64_t n, float scene_radius, float* wi3, float* dist,
                      float* pdf, float* emission16) {
  Scene* sc = (Scene*)s;
  for (uint64_t i = 0; i < n; ++i) {
    SampledLight sam;
    sam.emission = sp_uniform(0.0f);
    sam.pdf = 0.0f;
    sam.wiW = v3(0, 0, 0);
    sam.distance = 0.0f;
    light_sample(*sc, light_index, v3(pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]), v3(rand3[3 * i], rand3[3 * i + 1], rand3[3 * i + 2]), scene_radius, sam);
    wi3[3 * i] = sam.wiW.x; wi3[3 * i + 1] = sam.wiW.y; wi3[3 * i + 2] = sam.wiW.z;
    dist[i] = sam.distance;
    pdf[i] = sam.pdf;
    memcpy(emission16 + 16 * i, sam.emission.w, 64);
  }
}
uint32_t orc_scene_rt_light_count(void* s) { return (uint32_t)((Scene*)s)->rt_lights.size(); }

void orc_rand_stream(uint32_t seed, uint32_t px, uint32_t py, float* out, uint32_t n) {
  uint32_t rng = pcg_hash(fbits((float)seed) ^ pcg_hash(fbits((float)px) ^ pcg_hash(fbits((float)py))));
  for (uint32_t i = 0; i < n; ++i) out[i] = rnd(rng);
}

}  // extern "C"
