/*
 * pt_glsl_math.h — this build's DEFINITION of the GLSL built-ins whose results the
 * reference leaves to the Vulkan driver (SURVEY.md §8c "third-party arithmetic (ii)"):
 * sin, cos, pow (via exp2/log2).  main.comp uses them at main.comp:249,256,361,374,
 * 382,398,498,506,617.  The reference pins none of them ("parity unpinned"), so the
 * build fixes ONE definition, written with + - * and integer bit operations only, and
 * compiles it with floating-point contraction OFF on both the host (gcc) and the
 * device (hipcc, gfx950).  That makes every transcendental bit-identical between the
 * CPU oracle and the HIP kernels, which turns the parity tests from "statistically
 * close" into "bit-exact per pixel".
 *
 * The polynomials are the classic single-precision minimax sets published with the
 * Cephes math library (sinf/cosf/logf/expf); accuracy is 1–2 ulp on the ranges the
 * path tracer uses, checked against libm in tests/test_glsl_math.py.
 *
 * Plain C99; usable from C, C++ and HIP device code (PT_HD expands to
 * __host__ __device__ under hipcc).  No FMA may be formed from this code: build with
 * -ffp-contract=off.
 */
#ifndef PT_GLSL_MATH_H
#define PT_GLSL_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ static inline
#else
#define PT_HD static inline
#endif

PT_HD uint32_t pt_f2u(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  return c.u;
}

PT_HD float pt_u2f(uint32_t u) {
  union { float f; uint32_t u; } c;
  c.u = u;
  return c.f;
}

/* round-half-away for non-negative inputs is all the callers need; written with a
 * float->int truncation so host and device agree without relying on rintf(). */
PT_HD int32_t pt_round_nonneg(float x) { return (int32_t)(x + 0.5f); }

/* sin and cos of x (radians), |x| < ~8000.  Octant reduction with a three-term
 * Cody–Waite split of pi/4. */
PT_HD void pt_sincos(float x, float* s_out, float* c_out) {
  const float FOPI = 1.27323954473516f; /* 4/pi */
  const float DP1 = 0.78515625f;
  const float DP2 = 2.4187564849853515625e-4f;
  const float DP3 = 3.77489497744594108e-8f;

  int sin_neg = 0;
  if (x < 0.0f) {
    x = -x;
    sin_neg = 1;
  }

  uint32_t j = (uint32_t)(x * FOPI);
  j = (j + 1u) & ~1u; /* round odd octants up: y is an even multiple of pi/4 */
  float y = (float)j;

  float r = ((x - y * DP1) - y * DP2) - y * DP3;
  float z = r * r;

  /* sin(r), cos(r) on [-pi/4, pi/4] */
  float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
             - 0.5f * z + 1.0f;

  uint32_t q = (j >> 1) & 3u; /* quadrant: x = r + q*pi/2 */
  float s, c;
  if (q == 0u) { s = ps;  c = pc;  }
  else if (q == 1u) { s = pc;  c = -ps; }
  else if (q == 2u) { s = -ps; c = -pc; }
  else { s = -pc; c = ps;  }

  *s_out = sin_neg ? -s : s;
  *c_out = c;
}

PT_HD float pt_sin(float x) {
  float s, c;
  pt_sincos(x, &s, &c);
  return s;
}

PT_HD float pt_cos(float x) {
  float s, c;
  pt_sincos(x, &s, &c);
  return c;
}

/* log2 of a positive, finite, normal float. log2(1) == 0 exactly. */
PT_HD float pt_log2_pos(float x) {
  uint32_t u = pt_f2u(x);
  int32_t e = (int32_t)(u >> 23) - 127;
  float m = pt_u2f((u & 0x007FFFFFu) | 0x3F800000u); /* [1,2) */
  if (m > 1.41421356237f) {
    m = m * 0.5f;
    e += 1;
  }
  float f = m - 1.0f; /* [-0.293, 0.414] */
  float z = f * f;
  float p = 7.0376836292e-2f;
  p = p * f - 1.1514610310e-1f;
  p = p * f + 1.1676998740e-1f;
  p = p * f - 1.2420140846e-1f;
  p = p * f + 1.4249322787e-1f;
  p = p * f - 1.6668057665e-1f;
  p = p * f + 2.0000714765e-1f;
  p = p * f - 2.4999993993e-1f;
  p = p * f + 3.3333331174e-1f;
  float ln = (p * f * z - 0.5f * z) + f; /* ln(1+f) */
  return ln * 1.44269504088896341f + (float)e;
}

/* 2^x; flushes to 0 below 2^-126 and saturates to +inf above 2^128. */
PT_HD float pt_exp2(float x) {
  if (!(x > -126.0f)) {
    if (x != x) return x;
    return 0.0f;
  }
  if (x >= 128.0f) return pt_u2f(0x7F800000u);
  float fl = (float)(int32_t)x; /* trunc */
  if (fl > x) fl -= 1.0f;       /* floor */
  float r = x - fl;             /* [0,1) */
  if (r > 0.5f) {
    r -= 1.0f;
    fl += 1.0f;
  }
  float t = r * 0.693147180559945309f; /* [-0.347, 0.347] */
  float z = t * t;
  float p = 1.9875691500e-4f;
  p = p * t + 1.3981999507e-3f;
  p = p * t + 8.3334519073e-3f;
  p = p * t + 4.1665795894e-2f;
  p = p * t + 1.6666665459e-1f;
  p = p * t + 5.0000001201e-1f;
  float v = (p * z + t) + 1.0f;
  int32_t n = (int32_t)fl; /* [-126, 128] */
  if (n > 127) { /* r was rounded up at the very top of the range */
    v = v * 2.0f;
    n = 127;
  }
  return v * pt_u2f((uint32_t)(n + 127) << 23);
}

/* GLSL pow(x, y) for the uses in main.comp: x >= 0 expected, y > 0 constant.
 * x == 0 -> 0, x == 1 -> 1 exactly, x < 0 or NaN -> NaN (GLSL: undefined), +inf -> +inf. */
PT_HD float pt_pow(float x, float y) {
  if (x == 0.0f) return 0.0f;
  if (!(x > 0.0f)) return pt_u2f(0x7FC00000u);
  if (x == pt_u2f(0x7F800000u)) return x;
  if (x < 1.17549435e-38f) return 0.0f; /* denormal base: below anything a colour carries */
  return pt_exp2(y * pt_log2_pos(x));
}

#endif /* PT_GLSL_MATH_H */
