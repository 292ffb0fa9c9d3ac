// Device-side scalar/vector helpers.  Every operation is spelled out in the evaluation order the
// build defines for the GLSL built-ins ([ext]); with -ffp-contract=off this is
// what makes the kernels reproduce the CPU restatement bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "glz_detmath.h"

namespace glz {
namespace dev {

#define GLZ_D __device__ __forceinline__

constexpr float kPi = 3.1415926f;        // constants.glsl:4
constexpr float kInvPi = 0.3183099f;     // constants.glsl:5
constexpr float kTwoPi = 6.2831853f;     // constants.glsl:6
constexpr float kDefaultIor = 1.000293f; // constants.glsl:7
constexpr float kInv2Pi = 0.1591549f;    // constants.glsl:9

struct vec3 {
  float x, y, z;
};
struct vec2 {
  float x, y;
};

GLZ_D vec3 mk3(float x, float y, float z) { return vec3{x, y, z}; }
GLZ_D vec3 operator+(vec3 a, vec3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
GLZ_D vec3 operator-(vec3 a, vec3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
GLZ_D vec3 operator-(vec3 a) { return mk3(-a.x, -a.y, -a.z); }
GLZ_D vec3 operator*(vec3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
GLZ_D vec3 operator*(float s, vec3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
GLZ_D vec3 operator*(vec3 a, vec3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
GLZ_D vec3 operator/(vec3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
GLZ_D float dot3(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GLZ_D vec3 cross3(vec3 a, vec3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
GLZ_D vec3 normalize3(vec3 a) {
  float inv = 1.0f / sqrtf(dot3(a, a));
  return mk3(a.x * inv, a.y * inv, a.z * inv);
}

// GLSL 4.60 definitions
GLZ_D float gl_min(float x, float y) { return y < x ? y : x; }
GLZ_D float gl_max(float x, float y) { return x < y ? y : x; }
GLZ_D float gl_step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
GLZ_D float gl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
GLZ_D float gl_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
GLZ_D float check_nan(float x) { return isnan(x) ? 0.0f : x; }   // raytrace_commons.glsl:7
GLZ_D float check_inf(float x) { return isinf(x) ? 0.0f : x; }   // raytrace_commons.glsl:8
GLZ_D vec3 gl_reflect(vec3 I, vec3 N) { return I - (2.0f * dot3(N, I)) * N; }
GLZ_D vec3 gl_refract(vec3 I, vec3 N, float eta) {
  float d = dot3(N, I);
  float k = 1.0f - eta * eta * (1.0f - d * d);
  if (k < 0.0f) return mk3(0.0f, 0.0f, 0.0f);
  return eta * I - (eta * d + sqrtf(k)) * N;
}

// column-major mat4 applied to a point (w = 1), a direction (w = 0), and the transposed 3x3
GLZ_D vec3 xform_point(const float* __restrict__ m, vec3 p) {
  return mk3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
             m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
GLZ_D vec3 xform_dir(const float* __restrict__ m, vec3 p) {
  return mk3(m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z, m[2] * p.x + m[6] * p.y + m[10] * p.z);
}
GLZ_D vec3 xform_tdir(const float* __restrict__ m, vec3 n) {
  return mk3(m[0] * n.x + m[1] * n.y + m[2] * n.z, m[4] * n.x + m[5] * n.y + m[6] * n.z, m[8] * n.x + m[9] * n.y + m[10] * n.z);
}

// PCG hash RNG, random.glsl:7-57
GLZ_D uint32_t pcg(uint32_t seed) {
  uint32_t state = seed * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
GLZ_D float rand01(uint32_t& state) {
  state = pcg(state);
  return __uint_as_float(0x3F800000u | (state & 0x007FFFFFu)) - 1.0f;
}

}  // namespace dev
}  // namespace glz
