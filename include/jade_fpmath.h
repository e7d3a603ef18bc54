/*
 * jade_fpmath.h — deterministic fp32 math shared by every backend of jade_rt.
 *
 * The reference integrator (PathTrace.cu) leans on three closed third-party
 * pieces whose results cannot be reproduced outside an NVIDIA toolchain:
 * CUDA libm (powf/sinf/cosf/atan2f/asinf/norm3df), nvcc's default FMA
 * contraction, and cuRAND XORWOW shared racily between threads
 * (PathTrace.cu:38, 664-667, 1430-1431).  A Monte-Carlo path only stays the
 * same path if every comparison (hit/miss, u < 0.5, side-of-normal flips)
 * sees the same bits, so this header pins ONE evaluation order for the
 * arithmetic those pieces supplied, usable from C (gcc: the oracle), C++
 * (g++: host library) and HIP device code (hipcc, gfx950):
 *
 *   - only + - * / sqrtf and explicit __builtin_fmaf, every translation unit
 *     that includes this file MUST be built with -ffp-contract=off (and, for
 *     hipcc, the default correctly rounded fp32 divide/sqrt and denormals on);
 *     jade_fp_selftest() detects a build that contracted behind our back;
 *   - FMA is used exactly where nvcc contracts the reference's vec3 helpers
 *     (dot / cross / mixed_product, PathTrace.cu:257-289): a*b + c*d + e*f
 *     becomes fma(e,f, fma(c,d, a*b));
 *   - transcendental functions are small Cody-Waite + polynomial routines
 *     (cephes-style coefficients), a few ulp from the true value, identical
 *     to the bit on CPU and GPU;
 *   - the RNG is the reference's own deterministic generator from its GLSL
 *     integrator (shaders/fshader_render.fsh:82-98): a Wang hash iterated on
 *     a 32-bit state seeded per pixel and per sample.
 *
 * Nothing here is an intersection routine, a traversal or a shading rule:
 * those are restated independently by the oracle and by the HIP kernels.
 */
#ifndef JADE_FPMATH_H
#define JADE_FPMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define JADE_HD __host__ __device__ static inline
#else
#define JADE_HD static inline
#endif

/* The reference's constants, as written (PathTrace.cu:35-37). */
#define JADE_PI_D 3.1415926
#define JADE_E_D 2.71828182846
#define JADE_RR_RATE_D 0.9
#define JADE_SSS_RATE_D 0.5

typedef struct jvec3 {
  float x, y, z;
} jvec3;

/* ------------------------------------------------------------------ bits */

JADE_HD uint32_t jade_f2u(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  return c.u;
}
JADE_HD float jade_u2f(uint32_t u) {
  union { float f; uint32_t u; } c;
  c.u = u;
  return c.f;
}
JADE_HD int jade_isnan(float f) { return (jade_f2u(f) & 0x7fffffffu) > 0x7f800000u; }
JADE_HD float jade_fabs(float f) { return jade_u2f(jade_f2u(f) & 0x7fffffffu); }
JADE_HD float jade_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
JADE_HD float jade_sqrt(float a) { return __builtin_sqrtf(a); }

/* CUDA's min(float,float)/max(float,float) overloads are fminf/fmaxf: a NaN
 * operand is dropped (used for the scalar reductions in hitAABB,
 * PathTrace.cu:767-768).  Written with compares so that every backend agrees
 * on signed zeros too: ties return the FIRST operand. */
JADE_HD float jade_fminf(float a, float b) {
  if (jade_isnan(a)) return b;
  if (jade_isnan(b)) return a;
  return (b < a) ? b : a;
}
JADE_HD float jade_fmaxf(float a, float b) {
  if (jade_isnan(a)) return b;
  if (jade_isnan(b)) return a;
  return (b > a) ? b : a;
}

/* ------------------------------------------------------------------ vec3 */

JADE_HD jvec3 jv(float x, float y, float z) {
  jvec3 r;
  r.x = x; r.y = y; r.z = z;
  return r;
}
JADE_HD jvec3 jv_add(jvec3 a, jvec3 b) { return jv(a.x + b.x, a.y + b.y, a.z + b.z); }
JADE_HD jvec3 jv_sub(jvec3 a, jvec3 b) { return jv(a.x - b.x, a.y - b.y, a.z - b.z); }
JADE_HD jvec3 jv_mul(jvec3 a, jvec3 b) { return jv(a.x * b.x, a.y * b.y, a.z * b.z); }
JADE_HD jvec3 jv_div(jvec3 a, jvec3 b) { return jv(a.x / b.x, a.y / b.y, a.z / b.z); }
JADE_HD jvec3 jv_scale(jvec3 a, float s) { return jv(a.x * s, a.y * s, a.z * s); }
JADE_HD jvec3 jv_divs(jvec3 a, float s) { return jv(a.x / s, a.y / s, a.z / s); }
JADE_HD jvec3 jv_neg(jvec3 a) { return jv(a.x * -1.0f, a.y * -1.0f, a.z * -1.0f); }

/* dot / cross / mixed_product with nvcc's contraction pattern
 * (PathTrace.cu:257-266, 283-289). */
JADE_HD float jv_dot(jvec3 a, jvec3 b) {
  return jade_fma(a.z, b.z, jade_fma(a.y, b.y, a.x * b.x));
}
/* a*b - c*d  ->  fma(a, b, -(c*d)) */
JADE_HD float jade_diffprod(float a, float b, float c, float d) {
  return jade_fma(a, b, -(c * d));
}
JADE_HD jvec3 jv_cross(jvec3 b, jvec3 c) {
  return jv(jade_diffprod(b.y, c.z, b.z, c.y), jade_diffprod(b.z, c.x, b.x, c.z),
            jade_diffprod(b.x, c.y, b.y, c.x));
}
JADE_HD float jv_mixed(jvec3 a, jvec3 b, jvec3 c) {
  float t = a.x * jade_diffprod(b.y, c.z, b.z, c.y);
  t = jade_fma(a.y, jade_diffprod(b.z, c.x, b.x, c.z), t);
  return jade_fma(a.z, jade_diffprod(b.x, c.y, b.y, c.x), t);
}
/* norm3df + "1.0 / norm" of vec3_dv::normalize (PathTrace.cu:278-281): the
 * fp64 quotient rounded to fp32 equals the fp32 quotient (53 >= 2*24+2). */
JADE_HD float jv_len(jvec3 a) { return jade_sqrt(jv_dot(a, a)); }
JADE_HD jvec3 jv_normalize(jvec3 a) {
  float rev = 1.0f / jv_len(a);
  return jv(a.x * rev, a.y * rev, a.z * rev);
}
/* transform(v, f4, mat4) with mat4[col][row] (PathTrace.cu:268-276); m is the
 * 16 floats in the reference's memory order, m[4*c + r]. */
JADE_HD jvec3 jade_transform(jvec3 v, float f4, const float* m) {
  jvec3 r;
  r.x = jade_fma(m[12], f4, jade_fma(m[8], v.z, jade_fma(m[4], v.y, m[0] * v.x)));
  r.y = jade_fma(m[13], f4, jade_fma(m[9], v.z, jade_fma(m[5], v.y, m[1] * v.x)));
  r.z = jade_fma(m[14], f4, jade_fma(m[10], v.z, jade_fma(m[6], v.y, m[2] * v.x)));
  return r;
}

/* ------------------------------------------------------------------- RNG */

/* shaders/fshader_render.fsh:82-98.  One 32-bit state per (pixel, sample): the
 * frame term advances with the sample index (see JADE_SAMPLE_LANES in jade_rt.h). */
JADE_HD uint32_t jade_rng_seed(uint32_t px, uint32_t py, uint32_t frame) {
  return (px * 1973u + py * 9277u + frame * 26699u) | 1u;
}
JADE_HD uint32_t jade_wang(uint32_t* seed) {
  uint32_t s = *seed;
  s = (s ^ 61u) ^ (s >> 16);
  s *= 9u;
  s = s ^ (s >> 4);
  s *= 0x27d4eb2du;
  s = s ^ (s >> 15);
  *seed = s;
  return s;
}
/* float(uint)/2^32, round-to-nearest-even conversion: u in [0, 1]. */
JADE_HD float jade_rand(uint32_t* seed) {
  return (float)jade_wang(seed) * 2.3283064365386963e-10f;
}

/* ------------------------------------------------------- transcendentals */

/* floor for |x| < 2^31, without libm. */
JADE_HD float jade_floorf(float x) {
  float t;
  if (!(jade_fabs(x) < 2147483648.0f)) return x; /* huge, inf, NaN */
  t = (float)(int32_t)x;
  return (t > x) ? (t - 1.0f) : t;
}

/* sin and cos of x (radians), |x| up to a few thousand; 3-term Cody-Waite
 * reduction by pi/2 then the cephes sinf/cosf kernels on [-pi/4, pi/4]. */
JADE_HD void jade_sincosf(float x, float* sn, float* cs) {
  float kf, r, z, ps, pc;
  int32_t q;
  if (!(jade_fabs(x) < 1.0e6f)) { /* out of the supported range, inf, NaN */
    *sn = jade_u2f(0x7fc00000u);
    *cs = jade_u2f(0x7fc00000u);
    return;
  }
  kf = jade_floorf(jade_fma(x, 0.63661977236758134f, 0.5f));
  r = jade_fma(-kf, 1.5703125f, x);
  r = jade_fma(-kf, 4.837512969970703125e-4f, r);
  r = jade_fma(-kf, 7.54978995489188216e-8f, r);
  z = r * r;
  ps = jade_fma(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = jade_fma(ps, z, -1.6666654611e-1f);
  ps = jade_fma(ps * z, r, r);
  pc = jade_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = jade_fma(pc, z, 4.166664568298827e-2f);
  pc = jade_fma(pc * z, z, jade_fma(-0.5f, z, 1.0f));
  q = (int32_t)kf & 3;
  if (q == 0) { *sn = ps; *cs = pc; }
  else if (q == 1) { *sn = pc; *cs = -ps; }
  else if (q == 2) { *sn = -ps; *cs = -pc; }
  else { *sn = -pc; *cs = ps; }
}
JADE_HD float jade_sinf(float x) { float s, c; jade_sincosf(x, &s, &c); return s; }
JADE_HD float jade_cosf(float x) { float s, c; jade_sincosf(x, &s, &c); return c; }

/* log2(a) = e + t with e an integer and |t| <= 0.5 kept apart, so a caller
 * can form b*log2(a) without losing the low bits of t behind a large e.
 * Only for finite a > 0. */
JADE_HD void jade_log2_split(float a, float* ef, float* tf) {
  uint32_t u = jade_f2u(a);
  int32_t e = 0;
  float m, f, z, p, r;
  const float l2e_hi = 1.4426950216293335f; /* float(log2 e) */
  const float l2e_lo = 1.9259629911e-8f;    /* log2 e - l2e_hi */
  if (u < 0x00800000u) { /* subnormal: scale by 2^23 */
    a = a * 8388608.0f;
    u = jade_f2u(a);
    e = -23;
  }
  e += (int32_t)(u >> 23) - 127;
  m = jade_u2f((u & 0x007fffffu) | 0x3f800000u); /* [1,2) */
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; } /* [sqrt(1/2), sqrt 2) */
  f = m - 1.0f;
  z = f * f;
  /* cephes logf: log(1+f) = f - z/2 + f*z*P(f) */
  p = jade_fma(7.0376836292e-2f, f, -1.1514610310e-1f);
  p = jade_fma(p, f, 1.1676998740e-1f);
  p = jade_fma(p, f, -1.2420140846e-1f);
  p = jade_fma(p, f, 1.4249322787e-1f);
  p = jade_fma(p, f, -1.6668057665e-1f);
  p = jade_fma(p, f, 2.0000714765e-1f);
  p = jade_fma(p, f, -2.4999993993e-1f);
  p = jade_fma(p, f, 3.3333331174e-1f);
  p = p * f * z;
  p = jade_fma(-0.5f, z, p);
  /* (f + p) * log2(e), split so the leading term stays accurate */
  r = jade_fma(p, l2e_hi, f * l2e_lo);
  r = jade_fma(f, l2e_hi, r);
  *ef = (float)e;
  *tf = r;
}

/* log2(a).  a < 0 -> NaN, 0 -> -inf, inf -> inf. */
JADE_HD float jade_log2f(float a) {
  uint32_t u = jade_f2u(a);
  float e, t;
  if (jade_isnan(a)) return a;
  if ((u << 1) == 0u) return jade_u2f(0xff800000u); /* +-0 */
  if (u >> 31) return jade_u2f(0x7fc00000u);        /* negative */
  if (u == 0x7f800000u) return a;                   /* +inf */
  jade_log2_split(a, &e, &t);
  return t + e;
}

/* 2^(yh + yl), |yl| small.  >= 128 -> inf, < -150 -> 0, NaN -> NaN. */
JADE_HD float jade_exp2_split(float yh, float yl) {
  float nf, f, p, s1, s2;
  int32_t n, n1, n2;
  if (jade_isnan(yh) || jade_isnan(yl)) return jade_u2f(0x7fc00000u);
  if (yh >= 128.0f) return jade_u2f(0x7f800000u);
  if (yh < -150.0f) return 0.0f;
  nf = jade_floorf((yh + yl) + 0.5f);
  f = (yh - nf) + yl; /* about [-0.5, 0.5]; yh - nf is exact */
  n = (int32_t)nf;
  p = jade_fma(1.535336188319500e-4f, f, 1.339887440266574e-3f);
  p = jade_fma(p, f, 9.618437357674640e-3f);
  p = jade_fma(p, f, 5.550332471162809e-2f);
  p = jade_fma(p, f, 2.402264791363012e-1f);
  p = jade_fma(p, f, 6.931472028550421e-1f);
  p = jade_fma(p, f, 1.0f);
  /* scale by 2^n in two exact steps so subnormal results round once */
  n1 = n / 2;
  n2 = n - n1;
  s1 = jade_u2f((uint32_t)(n1 + 127) << 23);
  s2 = jade_u2f((uint32_t)(n2 + 127) << 23);
  return (p * s1) * s2;
}
JADE_HD float jade_exp2f(float y) { return jade_exp2_split(y, 0.0f); }

/* powf(a, b) as the integrator uses it: e^x profiles, rate^distance, gamma.
 * a < 0 -> NaN (the reference only ever raises to non-integer powers). */
JADE_HD float jade_powf(float a, float b) {
  float e, t, yh, yl;
  if (b == 0.0f) return 1.0f;
  if (a == 1.0f) return 1.0f;
  if (jade_isnan(a) || jade_isnan(b)) return jade_u2f(0x7fc00000u);
  if (a == 0.0f) return (b > 0.0f) ? 0.0f : jade_u2f(0x7f800000u);
  if (a < 0.0f) return jade_u2f(0x7fc00000u);
  if (jade_f2u(a) == 0x7f800000u) return (b > 0.0f) ? a : 0.0f;
  if ((jade_f2u(b) & 0x7fffffffu) == 0x7f800000u) { /* b = +-inf */
    int grow = (a > 1.0f) == (b > 0.0f);
    return grow ? jade_u2f(0x7f800000u) : 0.0f;
  }
  jade_log2_split(a, &e, &t);
  /* b*(e + t) = yh + yl with the rounding error of b*e recovered by FMA */
  yh = b * e;
  yl = jade_fma(b, e, -yh) + b * t;
  if (e == 0.0f) { yh = b * t; yl = jade_fma(b, t, -yh); }
  return jade_exp2_split(yh, yl);
}

/* atan on the whole line (cephes atanf). */
JADE_HD float jade_atanf(float t) {
  float a = jade_fabs(t), y, z, p;
  if (jade_isnan(t)) return t;
  if (a > 2.414213562373095f) { /* tan(3pi/8) */
    y = 1.5707963267948966f;
    a = -(1.0f / a);
  } else if (a > 0.4142135623730950f) { /* tan(pi/8) */
    y = 0.7853981633974483f;
    a = (a - 1.0f) / (a + 1.0f);
  } else {
    y = 0.0f;
  }
  z = a * a;
  p = jade_fma(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = jade_fma(p, z, 1.99777106478e-1f);
  p = jade_fma(p, z, -3.33329491539e-1f);
  p = jade_fma(p * z, a, a);
  y = y + p;
  return (jade_f2u(t) >> 31) ? -y : y;
}

/* atan2f(y, x), result in [-pi, pi]. */
JADE_HD float jade_atan2f(float y, float x) {
  const float pi = 3.14159265358979323846f;
  const float pio2 = 1.57079632679489661923f;
  float w;
  if (jade_isnan(x) || jade_isnan(y)) return jade_u2f(0x7fc00000u);
  if (x == 0.0f) {
    if (y == 0.0f) return 0.0f;
    return (y > 0.0f) ? pio2 : -pio2;
  }
  if (y == 0.0f) return (x > 0.0f) ? 0.0f : pi;
  if (x > 0.0f) w = 0.0f;
  else w = (y > 0.0f) ? pi : -pi;
  return w + jade_atanf(y / x);
}

/* asinf on [-1, 1] (cephes asinf); |x| > 1 -> NaN. */
JADE_HD float jade_asinf(float x) {
  float a = jade_fabs(x), z, s, p;
  int big;
  if (jade_isnan(x)) return x;
  if (a > 1.0f) return jade_u2f(0x7fc00000u);
  big = a > 0.5f;
  if (big) {
    z = 0.5f * (1.0f - a);
    s = jade_sqrt(z);
  } else {
    s = a;
    z = a * a;
  }
  p = jade_fma(4.2163199048e-2f, z, 2.4181311049e-2f);
  p = jade_fma(p, z, 4.5470025998e-2f);
  p = jade_fma(p, z, 7.4953002686e-2f);
  p = jade_fma(p, z, 1.6666752422e-1f);
  p = jade_fma(p * z, s, s);
  if (big) p = 1.5707963267948966f - (p + p);
  return (jade_f2u(x) >> 31) ? -p : p;
}

/* ------------------------------------------------------------- self-test */

/* Returns 0 when this translation unit evaluates a*b+c with two roundings
 * (i.e. it really was built with -ffp-contract=off) and explicit FMA with
 * one.  `one` must be passed as 1.0f from a place the optimiser cannot see
 * through (a volatile, a kernel argument). */
JADE_HD int jade_fp_selftest(float one) {
  float x = one + 1.220703125e-4f;           /* 1 + 2^-13 */
  float ref = one + 2.44140625e-4f;          /* 1 + 2^-12 */
  float two_roundings = x * x - ref;         /* 0 unless contracted */
  float fused = jade_fma(x, x, -ref);        /* 2^-26 */
  int bad = 0;
  if (two_roundings != 0.0f) bad |= 1;
  if (fused != 1.4901161193847656e-8f) bad |= 2;
  return bad;
}

#endif /* JADE_FPMATH_H */
