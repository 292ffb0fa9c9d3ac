/* glz_detmath.h -- deterministic single-precision elementary functions.
 *
 * A SPECIFICATION shared by the HIP kernels and the CPU oracle: sin, cos, acos and atan2 built
 * from +, -, *, / and sqrt only (all correctly rounded on x86-64/SSE and on gfx950 when compiled
 * with -ffp-contract=off), so that the two sides produce bit-identical results and discrete
 * decisions of the path tracer (Russian roulette, Fresnel branch, light pick) cannot flip on
 * last-bit differences between glibc's libm and the device's ocml.  The reference leaves these to
 * the GLSL built-ins of the Vulkan driver ([ext], SURVEY 8c: parity unpinned).
 *
 * Coefficients: the classic Cephes single-precision minimax polynomials (sinf/cosf/asinf/atanf/logf).
 * Accuracy: <= 2 ulp over the ranges the renderer uses (checked against numpy in tests/).
 */
#ifndef GLZ_DETMATH_H
#define GLZ_DETMATH_H

#if defined(__HIPCC__)
#define GLZ_HD __host__ __device__ inline
#else
#define GLZ_HD inline
#endif

#if defined(__HIPCC__) || defined(__cplusplus)
#include <math.h>
#endif

/* floor for |x| < 2^31 without libm dependence on the device */
GLZ_HD float glz_floorf(float x) {
  float t = (float)(int)x;
  return t > x ? t - 1.0f : t;
}

/* Reduces x to r in [-pi/4, pi/4] with x = k*(pi/2) + r; returns k & 3.  Cody-Waite, 3 terms. */
GLZ_HD int glz_reduce_pio2(float x, float* r) {
  const float two_over_pi = 0.636619772367581343f;
  const float p1 = 1.5703125f;                 /* pi/2 split into three parts */
  const float p2 = 4.837512969970703125e-4f;
  const float p3 = 7.54978995489188216e-8f;
  float kf = glz_floorf(x * two_over_pi + 0.5f);
  float y = x - kf * p1;
  y = y - kf * p2;
  y = y - kf * p3;
  *r = y;
  return ((int)kf) & 3;
}

GLZ_HD float glz_sin_poly(float r) {
  float z = r * r;
  float p = -1.9515295891e-4f * z + 8.3321608736e-3f;
  p = p * z - 1.6666654611e-1f;
  return r + r * z * p;
}

GLZ_HD float glz_cos_poly(float r) {
  float z = r * r;
  float p = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
  p = p * z + 4.166664568298827e-2f;
  return (1.0f - 0.5f * z) + z * z * p;
}

GLZ_HD float glz_sinf(float x) {
  float r;
  int q = glz_reduce_pio2(x, &r);
  float s = (q & 1) ? glz_cos_poly(r) : glz_sin_poly(r);
  return (q & 2) ? -s : s;
}

GLZ_HD float glz_cosf(float x) {
  float r;
  int q = glz_reduce_pio2(x, &r);
  float c = (q & 1) ? glz_sin_poly(r) : glz_cos_poly(r);
  return ((q + 1) & 2) ? -c : c;
}

/* asin on [0, 0.5] by polynomial, on (0.5, 1] through asin(x) = pi/2 - 2 asin(sqrt((1-x)/2)) */
GLZ_HD float glz_asin_poly(float x) {
  float z = x * x;
  float p = 4.2163199048e-2f * z + 2.4181311049e-2f;
  p = p * z + 4.5470025998e-2f;
  p = p * z + 7.4953002686e-2f;
  p = p * z + 1.6666752422e-1f;
  return x + x * z * p;
}

GLZ_HD float glz_acosf(float x) {
  const float pi = 3.14159265358979323846f;
  const float pio2 = 1.57079632679489661923f;
  if (x != x) return x;
  if (x >= 1.0f) return 0.0f;
  if (x <= -1.0f) return pi;
  float a = x < 0.0f ? -x : x;
  if (a <= 0.5f) return pio2 - (x < 0.0f ? -glz_asin_poly(a) : glz_asin_poly(a));
  float s = sqrtf((1.0f - a) * 0.5f);
  float t = 2.0f * glz_asin_poly(s);
  return x < 0.0f ? pi - t : t;
}

/* atan on [0, inf) */
GLZ_HD float glz_atan_pos(float x) {
  const float pio2 = 1.57079632679489661923f;
  const float pio4 = 0.78539816339744830962f;
  float y;
  if (x > 2.414213562373095f) { /* tan(3pi/8) */
    y = pio2;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) { /* tan(pi/8) */
    y = pio4;
    x = (x - 1.0f) / (x + 1.0f);
  } else {
    y = 0.0f;
  }
  float z = x * x;
  float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
  p = p * z + 1.99777106478e-1f;
  p = p * z - 3.33329491539e-1f;
  return y + (p * z * x + x);
}

/* atan2(y, x), GLSL atan(y, x).  Zero/zero returns 0. */
GLZ_HD float glz_atan2f(float y, float x) {
  const float pi = 3.14159265358979323846f;
  const float pio2 = 1.57079632679489661923f;
  if (x != x || y != y) return x + y;
  if (y == 0.0f) return x < 0.0f ? pi : 0.0f;
  if (x == 0.0f) return y > 0.0f ? pio2 : -pio2;
  float ay = y < 0.0f ? -y : y;
  float ax = x < 0.0f ? -x : x;
  float a = glz_atan_pos(ay / ax);
  if (x < 0.0f) a = pi - a;
  return y < 0.0f ? -a : a;
}

/* log2(x) for finite x > 0 (normal numbers): exponent and mantissa split by bit operations, log(1 + t) on
 * [sqrt(1/2) - 1, sqrt(2) - 1] by the Cephes logf polynomial, scaled by log2(e).  Used for the texture level of detail. */
GLZ_HD float glz_log2f(float x) {
  union { float f; unsigned u; } c;
  c.f = x;
  int e = (int)((c.u >> 23) & 0xFFu) - 127;
  c.u = (c.u & 0x007FFFFFu) | 0x3F800000u;
  float m = c.f;                              /* [1, 2) */
  if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
  float t = m - 1.0f;
  float z = t * t;
  float p = 7.0376836292e-2f * t - 1.1514610310e-1f;
  p = p * t + 1.1676998740e-1f;
  p = p * t - 1.2420140846e-1f;
  p = p * t + 1.4249322787e-1f;
  p = p * t - 1.6668057665e-1f;
  p = p * t + 2.0000714765e-1f;
  p = p * t - 2.4999993993e-1f;
  p = p * t + 3.3333331174e-1f;
  float y = (p * t) * z - 0.5f * z;
  return (t + y) * 1.44269504088896340736f + (float)e;
}

#endif /* GLZ_DETMATH_H */
