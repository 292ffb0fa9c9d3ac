// Host-side math of the render path: 4x4 matrices (cgmath 0.18 stand-ins [ext]), camera push
// constants, sky rotation, CPU-side spectrum/luminance and the piecewise-constant distributions.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "glaze_abi.h"
#include "glz_tables.h"

namespace glz {
namespace host {

struct Mat4d {
  double m[16];   // column-major, m[col * 4 + row]
  static Mat4d identity() {
    Mat4d r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
    return r;
  }
  static Mat4d from_f32(const float* f) {
    Mat4d r;
    for (int i = 0; i < 16; ++i) r.m[i] = f[i];
    return r;
  }
  void to_f32(float* f) const {
    for (int i = 0; i < 16; ++i) f[i] = (float)m[i];
  }
};

inline Mat4d operator*(const Mat4d& a, const Mat4d& b) {
  Mat4d r{};
  for (int c = 0; c < 4; ++c)
    for (int row = 0; row < 4; ++row) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a.m[k * 4 + row] * b.m[c * 4 + k];
      r.m[c * 4 + row] = s;
    }
  return r;
}

// Inverse through the adjugate (cofactor expansion); false if singular.
inline bool invert(const Mat4d& a, Mat4d& out) {
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
  if (det == 0.0) return false;
  det = 1.0 / det;
  for (int i = 0; i < 16; ++i) out.m[i] = inv[i] * det;
  return true;
}

// PerspectiveCam::fovy (geometry/camera.rs:24-28) -- f32 like the reference
inline float fovy(float fovx, float aspect) { return 2.0f * atanf(tanf(fovx * 0.5f) / aspect); }

// Matrix4::look_at_rh (cgmath [ext])
inline Mat4d look_at_rh(const glz_camera& c) {
  const double ex = c.position[0], ey = c.position[1], ez = c.position[2];
  double fx = c.target[0] - ex, fy = c.target[1] - ey, fz = c.target[2] - ez;
  const double fl = std::sqrt(fx * fx + fy * fy + fz * fz);
  fx /= fl; fy /= fl; fz /= fl;
  const double ux = c.up[0], uy = c.up[1], uz = c.up[2];
  double sx = fy * uz - fz * uy, sy = fz * ux - fx * uz, sz = fx * uy - fy * ux;
  const double sl = std::sqrt(sx * sx + sy * sy + sz * sz);
  sx /= sl; sy /= sl; sz /= sl;
  const double vx = sy * fz - sz * fy, vy = sz * fx - sx * fz, vz = sx * fy - sy * fx;
  Mat4d v{};
  v.m[0] = sx; v.m[1] = vx; v.m[2] = -fx;
  v.m[4] = sy; v.m[5] = vy; v.m[6] = -fy;
  v.m[8] = sz; v.m[9] = vz; v.m[10] = -fz;
  v.m[12] = -(ex * sx + ey * sy + ez * sz);
  v.m[13] = -(ex * vx + ey * vy + ez * vz);
  v.m[14] = ex * fx + ey * fy + ez * fz;
  v.m[15] = 1.0;
  return v;
}

// Camera::projection (geometry/camera.rs:127-142): cgmath::perspective / cgmath::ortho [ext]
inline Mat4d projection(const glz_camera& c, uint32_t width, uint32_t height) {
  Mat4d p{};
  if (c.type == GLZ_CAMERA_PERSPECTIVE) {
    const float ar = (float)width / (float)height;
    const double f = 1.0 / std::tan((double)fovy(c.fovx_or_scale, ar) / 2.0);
    const double n = c.near_plane, fa = c.far_plane;
    p.m[0] = f / (double)ar;
    p.m[5] = f;
    p.m[10] = (fa + n) / (n - fa);
    p.m[11] = -1.0;
    p.m[14] = (2.0 * fa * n) / (n - fa);
  } else {
    const double l = -(double)c.fovx_or_scale, r = c.fovx_or_scale, b = l, t = r, n = -(double)c.far_plane, fa = c.far_plane;
    p.m[0] = 2.0 / (r - l);
    p.m[5] = 2.0 / (t - b);
    p.m[10] = -2.0 / (fa - n);
    p.m[12] = -(r + l) / (r - l);
    p.m[13] = -(t + b) / (t - b);
    p.m[14] = -(fa + n) / (fa - n);
    p.m[15] = 1.0;
  }
  return p;
}

// build_push_constants (vulkan/raytracer.rs:1098-1120)
inline void push_constants(const glz_camera& c, uint32_t width, uint32_t height, float camera2world[16], float screen2camera[16]) {
  Mat4d view_inv, proj_inv;
  if (!invert(look_at_rh(c), view_inv)) view_inv = Mat4d::identity();
  Mat4d proj = projection(c, width, height);
  proj.m[5] *= -1.0;
  if (!invert(proj, proj_inv)) proj_inv = Mat4d::identity();
  view_inv.to_f32(camera2world);
  proj_inv.to_f32(screen2camera);
}

// Light::rotation_matrix (geometry/light.rs:195-199): from_angle_y(yaw) * from_angle_z(pitch) * from_angle_x(roll)
inline Mat4d sky_rotation(float yaw_deg, float pitch_deg, float roll_deg) {
  const double k = 3.14159265358979323846 / 180.0;
  const double cy = std::cos(yaw_deg * k), sy = std::sin(yaw_deg * k);
  const double cz = std::cos(pitch_deg * k), sz = std::sin(pitch_deg * k);
  const double cx = std::cos(roll_deg * k), sx = std::sin(roll_deg * k);
  Mat4d ry = Mat4d::identity(), rz = Mat4d::identity(), rx = Mat4d::identity();
  ry.m[0] = cy; ry.m[2] = -sy; ry.m[8] = sy; ry.m[10] = cy;
  rz.m[0] = cz; rz.m[1] = sz; rz.m[4] = -sz; rz.m[5] = cz;
  rx.m[5] = cx; rx.m[6] = sx; rx.m[9] = -sx; rx.m[10] = cx;
  return (ry * rz) * rx;
}

// Spectrum::from_rgb(c, true).luminance() (geometry/spectrum.rs:82-141, :168-174) -- the CPU-side
// colour math the sky distribution is built with (f64-literal tables, clamped; Q10).
inline float illuminant_luminance(float r, float g, float b) {
  const float* W = GLZ_HOST_SPECTRUM_WHITEL;
  const float *A, *B;
  float k0, k1, k2;
  if (r <= g && r <= b) {
    k0 = r; A = GLZ_HOST_SPECTRUM_CYANL;
    if (g <= b) { k1 = g - r; B = GLZ_HOST_SPECTRUM_BLUEL; k2 = b - g; } else { k1 = b - r; B = GLZ_HOST_SPECTRUM_GREENL; k2 = g - b; }
  } else if (g <= r && g <= b) {
    k0 = g; A = GLZ_HOST_SPECTRUM_MAGENTAL;
    if (r <= b) { k1 = r - g; B = GLZ_HOST_SPECTRUM_BLUEL; k2 = b - r; } else { k1 = b - g; B = GLZ_HOST_SPECTRUM_REDL; k2 = r - b; }
  } else {
    k0 = b; A = GLZ_HOST_SPECTRUM_YELLOWL;
    if (r <= g) { k1 = r - b; B = GLZ_HOST_SPECTRUM_GREENL; k2 = g - r; } else { k1 = g - b; B = GLZ_HOST_SPECTRUM_REDL; k2 = r - g; }
  }
  float y = 0.0f;
  for (int i = 0; i < 16; ++i) {
    float w = 0.0f;
    w += W[i] * k0;
    w += A[i] * k1;
    w += B[i] * k2;
    w *= 0.86445f;
    w = w < 0.0f ? 0.0f : (w > 1.0f ? 1.0f : w);
    y += w * GLZ_HOST_Y[i];
  }
  y *= 0.17557178f;
  return y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
}

// Distribution1D::new (geometry/distribution.rs:14-38): appends n+1 cdf entries, returns the integral
inline float distribution1d(const float* values, size_t n, std::vector<float>& cdf_out) {
  const float nf = (float)n;
  const size_t base = cdf_out.size();
  cdf_out.resize(base + n + 1);
  float* cdf = cdf_out.data() + base;
  cdf[0] = 0.0f;
  for (size_t i = 1; i <= n; ++i) cdf[i] = cdf[i - 1] + values[i - 1] / nf;
  const float integral = cdf[n];
  if (integral == 0.0f) {
    for (size_t i = 1; i <= n; ++i) cdf[i] = (float)i / nf;
  } else {
    for (size_t i = 1; i <= n; ++i) cdf[i] = cdf[i] / integral;
  }
  return integral;
}

// host seed stream (build-defined: the reference seeds from OS entropy, raytracer.rs:779):
// xoshiro128++ whose 128-bit state is filled from SplitMix64(seed), as rand_xoshiro's seed_from_u64 does
struct SeedStream {
  uint32_t s[4];
  explicit SeedStream(uint64_t seed = 0) { reseed(seed); }
  void reseed(uint64_t seed) {
    uint64_t z[2];
    for (int i = 0; i < 2; ++i) {
      seed += 0x9E3779B97F4A7C15ull;
      uint64_t x = seed;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      z[i] = x ^ (x >> 31);
    }
    s[0] = (uint32_t)z[0]; s[1] = (uint32_t)(z[0] >> 32); s[2] = (uint32_t)z[1]; s[3] = (uint32_t)(z[1] >> 32);
  }
  static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
  uint32_t next() {
    const uint32_t result = rotl(s[0] + s[3], 7) + s[0];
    const uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 11);
    return result;
  }
};

// WorkScheduler (vulkan/raytracer.rs:1168-1206): hierarchical midpoint subdivision of the pixel
// area, LIFO over the current level, four children queued for the next.
class WorkScheduler {
 public:
  WorkScheduler() { rewind(); }
  void rewind() {
    current_.clear();
    next_.clear();
    current_.push_back(Area{{0.0f, 0.0f}, {1.0f, 1.0f}});
  }
  void next(float out[2]) {
    while (current_.empty()) {
      current_.swap(next_);
      if (current_.empty()) rewind();
    }
    const Area a = current_.back();
    current_.pop_back();
    const float mx = (a.lo[0] + a.hi[0]) / 2.0f, my = (a.lo[1] + a.hi[1]) / 2.0f;
    next_.push_back(Area{{a.lo[0], a.lo[1]}, {mx, my}});
    next_.push_back(Area{{mx, my}, {a.hi[0], a.hi[1]}});
    next_.push_back(Area{{mx, a.lo[1]}, {a.hi[0], my}});
    next_.push_back(Area{{a.lo[0], my}, {mx, a.hi[1]}});
    out[0] = mx;
    out[1] = my;
  }
  // what the next call of next() will return, without taking it (the launch after this one: FrameData::next_pixel_offset)
  void peek(float out[2]) const {
    const Area full{{0.0f, 0.0f}, {1.0f, 1.0f}};
    const Area& a = !current_.empty() ? current_.back() : (!next_.empty() ? next_.back() : full);
    out[0] = (a.lo[0] + a.hi[0]) / 2.0f;
    out[1] = (a.lo[1] + a.hi[1]) / 2.0f;
  }

 private:
  struct Area {
    float lo[2], hi[2];
  };
  std::vector<Area> current_, next_;
};

// Thresholds of the 8-bit sRGB quantiser (the R8G8B8A8_SRGB blit of raytracer.rs:576-584 [ext]): a linear value c encodes to
// q = #{k in 1..255 : c >= thr[k]}, thr[k] = (float) EOTF((k - 0.5) / 255) -- round(255 * OETF(c)) without evaluating pow()
// per pixel, so the device and any other implementation of this rule agree on every byte.  thr[0] = 0.
inline void srgb8_thresholds(float thr[256]) {
  thr[0] = 0.0f;
  for (int k = 1; k < 256; ++k) {
    const double v = ((double)k - 0.5) / 255.0;
    thr[k] = (float)(v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4));
  }
}
}  // namespace host
}  // namespace glz
