/*
 * ofp_math.h -- the arithmetic spec of the onset-detection path.
 *
 * The reference computes the rectified-dB signal and the relative envelope with
 * numpy float32 ufuncs:  20*np.log10(np.abs(x + 1e-10))  (detection.py:747) and
 * 10 ** (rel / 20) - 1e-10  (detection.py:753).  numpy's float32 log10/power are
 * libm or SVML results that are NOT correctly rounded and change with the host
 * CPU, so "the reference's bits" are not a fixed
 * target.  This header fixes the canon used by BOTH the HIP kernels and the CPU
 * oracle: evaluate in fp64 with a fixed sequence of IEEE mul/add operations
 * (no FMA contraction, no library calls, no division), round once to fp32.
 * The result is the correctly rounded fp32 value except when the exact value
 * lies within ~2^-50 (relative) of an fp32 rounding boundary, and it is
 * bit-identical on x86-64 and gfx950 by construction.
 *
 * Every other operation on the path is a plain IEEE fp32 (or the one fp64 add of
 * envelope_follower.c:16) operation in the reference's own order; those are
 * spelled out here too so that there is exactly one definition.
 *
 * Compile every translation unit that includes this header with
 * -ffp-contract=off.
 */
#ifndef OFP_MATH_H
#define OFP_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OFP_HD __host__ __device__ inline __attribute__((always_inline))
#define OFP_D __device__ inline __attribute__((always_inline))
#if defined(__HIP_DEVICE_COMPILE__)
#define OFP_TABLE_QUAL static __device__ const
#else
#define OFP_TABLE_QUAL static const
#endif
#pragma clang fp contract(off)
#else
#define OFP_HD static inline
#define OFP_D static inline
#define OFP_TABLE_QUAL static const
#endif

#include "ofp_math_tables.h"

OFP_HD uint64_t ofp_d2u(double d) {
    union { double d; uint64_t u; } c;
    c.d = d;
    return c.u;
}
OFP_HD double ofp_u2d(uint64_t u) {
    union { double d; uint64_t u; } c;
    c.u = u;
    return c.d;
}
OFP_HD uint32_t ofp_f2u(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    return c.u;
}
OFP_HD float ofp_u2f(uint32_t u) {
    union { float f; uint32_t u; } c;
    c.u = u;
    return c.f;
}

/* log10 of a non-negative fp32 value, fp64-evaluated, rounded once to fp32. */
OFP_HD float ofp_log10f(float a) {
    if (a != a) return a;
    if (a == 0.0f) return ofp_u2f(0xff800000u); /* -inf */
    if (a < 0.0f) return ofp_u2f(0x7fc00000u);  /* nan  */
    if (ofp_f2u(a) == 0x7f800000u) return a;    /* +inf */
    uint64_t u = ofp_d2u((double)a); /* exact, always a normal double */
    int k = (int)((u >> 52) & 0x7ff) - 1023;
    uint64_t mant = u & 0x000fffffffffffffull;
    int i = (int)(mant >> 45);
    uint64_t ebits = 0x3ff0000000000000ull;
    if (i >= 53) { /* m' = m/2 in [0.707,1) */
        ebits = 0x3fe0000000000000ull;
        k += 1;
    }
    double m = ofp_u2d(ebits | mant);
    double z = m * OFP_LOG_R[i] - 1.0; /* exact product, exact difference */
    /* log1p(z), |z| < 0.0082: z - z^2/2 + ... - z^10/10 (Horner, mul+add) */
    double p = -0.1;
    p = p * z + (1.0 / 9.0);
    p = p * z - 0.125;
    p = p * z + (1.0 / 7.0);
    p = p * z - (1.0 / 6.0);
    p = p * z + 0.2;
    p = p * z - 0.25;
    p = p * z + (1.0 / 3.0);
    p = p * z - 0.5;
    p = p * z + 1.0;
    p = p * z;
    double r = (double)k * OFP_LOG10_2 + OFP_LOG_T[i];
    r = r + p * OFP_LOG10_E;
    return (float)r;
}

/* 10**v for fp32 v, fp64-evaluated, rounded once to fp32. */
OFP_HD float ofp_exp10f(float v) {
    if (v != v) return v;
    if (v >= 39.0f) return ofp_u2f(0x7f800000u); /* > FLT_MAX */
    if (v <= -46.0f) return 0.0f;                /* < half the least denormal */
    double t = (double)v * OFP_LOG2_10;
    double tn = t * 32.0;
    int n = (int)(tn + (tn >= 0.0 ? 0.5 : -0.5));
    double f = t - (double)n * 0.03125; /* exact */
    double y = f * OFP_LN2;
    /* e^y, |y| <= 0.011: Taylor to y^7 */
    double p = 1.0 / 5040.0;
    p = p * y + (1.0 / 720.0);
    p = p * y + (1.0 / 120.0);
    p = p * y + (1.0 / 24.0);
    p = p * y + (1.0 / 6.0);
    p = p * y + 0.5;
    p = p * y + 1.0;
    p = p * y + 1.0;
    int e = n >> 5; /* floor(n/32), arithmetic shift */
    double s = ofp_u2d((uint64_t)(1023 + e) << 52); /* 2^e, e in [-153,130] */
    double r = OFP_EXP2_T[n & 31] * p;
    r = r * s; /* exact scaling */
    return (float)r;
}

/* Rectified dB with floor: detection.py:747-748
 *   x = 20 * np.log10(np.abs(x + 1e-10)); x = x.clip(floor)
 * (all float32 under NumPy-2 promotion; clip keeps NaN). */
OFP_HD float ofp_rect_db(float x, float floor_db) {
    float a = x + 1e-10f;
    a = a < 0.0f ? -a : a;
    float v = 20.0f * ofp_log10f(a);
    return v < floor_db ? floor_db : v;
}

/* Relative envelope back to linear: detection.py:753-754
 *   rel = 10 ** (rel / 20) - 1e-10; rel = rel.clip(0, -floor) */
OFP_HD float ofp_rel_linear(float dif, float floor_db) {
    float v = ofp_exp10f(dif / 20.0f) - 1e-10f;
    float hi = -floor_db;
    v = v < 0.0f ? 0.0f : v;
    v = v > hi ? hi : v;
    return v;
}

/* One step of the attack/release follower: envelope_follower.c:15-22.
 * `diff = xi - yi + 1e-10` is an fp32 subtract, an fp64 add of the double
 * literal, and a round back to fp32. */
OFP_HD float ofp_ar_step(float x, float y, float attack, float release) {
    float s = x - y;
    float d = (float)((double)s + 1e-10);
    float g = d > 0.0f ? attack : release;
    return y + g * d;
}

/* One step of the EMA min tracker: envelope_follower.c:39-45. */
OFP_HD float ofp_min_step(float x, float mn, float ialpha, float alpha, float minmin) {
    /* branch-free form of: if (x < minmin) minmin; else if (x < mn) x; else ema */
    float a = mn * ialpha;
    float b = x * alpha;
    float r = a + b;
    r = x < mn ? x : r;
    r = x < minmin ? minmin : r;
    return r;
}

/* One step of the EMA max tracker: envelope_follower.c:47-51. */
OFP_HD float ofp_max_step(float x, float mx, float ialpha, float alpha) {
    /* branch-free form of: if (x > mx) x; else ema */
    float a = mx * ialpha;
    float b = x * alpha;
    float r = a + b;
    return x > mx ? x : r;
}

/* (float)(1.0 - alpha): envelope_follower.c:31-32, fp64 subtract then fp32 store */
OFP_HD float ofp_ialpha(float alpha) { return (float)(1.0 - (double)alpha); }

/* One sample of a 4th-order direct-form-II-transposed IIR in fp32, the
 * operation order of scipy.signal.lfilter's float kernel (detection.py:499-501;
 * b,a already divided by a[0] in fp32 as scipy does). z[4] is updated. */
OFP_HD float ofp_df2t4_step(float x, const float* b, const float* a, float* z) {
    float y = z[0] + b[0] * x;
    z[0] = (z[1] + x * b[1]) - y * a[1];
    z[1] = (z[2] + x * b[2]) - y * a[2];
    z[2] = (z[3] + x * b[3]) - y * a[3];
    z[3] = x * b[4] - y * a[4];
    return y;
}

#endif /* OFP_MATH_H */
