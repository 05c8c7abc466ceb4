/*
 * orc_libm.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The five transcendental functions the path uses (sin, cos, log, log2, exp; tan only in
 * proj_slope) come from ONE of two places:
 *
 *   default              glibc libm.  This is the variant pinned BIT-EXACT against the
 *                        reference's own headers (tests/golden/ref_math.bin), liborc.so.
 *   -DORC_PORTABLE_LIBM  the "pl_" functions below (Cephes single-precision algorithms,
 *                        S. Moshier, restated with IEEE fp32 + - * / and integer bit
 *                        operations only, no FMA).  Built -ffp-contract=off they give the
 *                        same bits on gcc/x86 and on hipcc/gfx950, which is what lets the
 *                        GPU parity tests demand BIT-EXACT images, hits and bakes.  This
 *                        variant (liborc_p.so) is pinned against the same golden vectors
 *                        to a few ulp (tests/test_oracle_golden.py), i.e. it differs from
 *                        the reference by no more than two libm's differ from each other
 *                        (the reference itself is built -ffast-math, CMakeLists.txt:168).
 *
 * The HIP product carries its own statement of the same algorithms
 * (pim_amd/csrc/pt_math.h); the specification both follow is DESIGN.md "portable libm".
 */
#ifndef ORC_LIBM_H
#define ORC_LIBM_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float pl_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t pl_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* shared octant reduction of |x| by pi/4 (Cephes sinf.c): returns reduced x, writes octant j (0..7) */
static inline float pl_reduce_pio4(float x, int32_t* jOut)
{
    int32_t j = (int32_t)(1.27323954473516f * x);   /* x / (pi/4), truncated */
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    *jOut = j & 7;
    if (x > 8192.0f)
    {
        return x - y * 0.78539816339744830962f;
    }
    return ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
}
static inline float pl_sin_poly(float x, float z)
{
    float y = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x;
    return y + x;
}
static inline float pl_cos_poly(float z)
{
    float y = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}
static inline float pl_sinf(float xx)
{
    float sign = 1.0f, x = xx;
    if (xx < 0.0f) { sign = -1.0f; x = -xx; }
    if (!(x <= 16777215.0f)) return (x != x) ? x : 0.0f;
    int32_t j;
    x = pl_reduce_pio4(x, &j);
    if (j > 3) { sign = -sign; j -= 4; }
    float z = x * x;
    float y = (j == 1 || j == 2) ? pl_cos_poly(z) : pl_sin_poly(x, z);
    return (sign < 0.0f) ? -y : y;
}
static inline float pl_cosf(float xx)
{
    float sign = 1.0f, x = (xx < 0.0f) ? -xx : xx;
    if (!(x <= 16777215.0f)) return (x != x) ? x : 0.0f;
    int32_t j;
    x = pl_reduce_pio4(x, &j);
    if (j > 3) { j -= 4; sign = -sign; }
    if (j > 1) sign = -sign;
    float z = x * x;
    float y = (j == 1 || j == 2) ? pl_sin_poly(x, z) : pl_cos_poly(z);
    return (sign < 0.0f) ? -y : y;
}
static inline float pl_tanf(float x) { return pl_sinf(x) / pl_cosf(x); }

/* frexp for positive normal floats: x = m * 2^e, m in [0.5, 1) */
static inline float pl_frexp(float x, int32_t* e)
{
    uint32_t u = pl_asuint(x);
    *e = (int32_t)((u >> 23) & 0xffu) - 126;
    return pl_asfloat((u & 0x007fffffu) | 0x3f000000u);
}
static inline float pl_log_poly(float x)
{
    return ((((((((7.0376836292e-2f * x - 1.1514610310e-1f) * x + 1.1676998740e-1f) * x - 1.2420140846e-1f) * x
        + 1.4249322787e-1f) * x - 1.6668057665e-1f) * x + 2.0000714765e-1f) * x - 2.4999993993e-1f) * x + 3.3333331174e-1f);
}
/* special cases shared by log and log2: returns 1 and writes *r when x is not a positive normal finite float */
static inline int pl_log_special(float x, float* r)
{
    uint32_t u = pl_asuint(x);
    if (x != x) { *r = x; return 1; }
    if ((u & 0x7f800000u) == 0u) { *r = -INFINITY; return 1; }   /* +-0 and denormals (DAZ) */
    if (u & 0x80000000u) { *r = NAN; return 1; }
    if (u == 0x7f800000u) { *r = x; return 1; }
    return 0;
}
static inline float pl_logf(float x)
{
    float r;
    if (pl_log_special(x, &r)) return r;
    int32_t e;
    x = pl_frexp(x, &e);
    if (x < 0.707106781186547524f) { e -= 1; x = (x + x) - 1.0f; }
    else { x = x - 1.0f; }
    float z = x * x;
    float y = pl_log_poly(x) * x * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = x + y;
    z = z + 0.693359375f * fe;
    return z;
}
static inline float pl_log2f(float x)
{
    float r;
    if (pl_log_special(x, &r)) return r;
    int32_t e;
    x = pl_frexp(x, &e);
    if (x < 0.707106781186547524f) { e -= 1; x = (x + x) - 1.0f; }
    else { x = x - 1.0f; }
    float z = x * x;
    float y = pl_log_poly(x) * x * z;
    y = y - 0.5f * z;
    z = y * 0.44269504088896340735992f;
    z = z + x * 0.44269504088896340735992f;
    z = z + y;
    z = z + x;
    z = z + (float)e;
    return z;
}
static inline float pl_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -87.33654475055310898657f) return 0.0f;     /* results below 2^-126 flush to zero (FTZ) */
    float t = 1.44269504088896341f * x + 0.5f;
    float z = (float)(int32_t)t;
    if (z > t) z = z - 1.0f;                            /* floor */
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int32_t n = (int32_t)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x
        + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    int32_t n0 = n / 2, n1 = n - n0;                    /* ldexp in two exact steps */
    z = z * pl_asfloat((uint32_t)(n0 + 127) << 23);
    z = z * pl_asfloat((uint32_t)(n1 + 127) << 23);
    return z;
}

#ifdef ORC_PORTABLE_LIBM
#define o_sinf  pl_sinf
#define o_cosf  pl_cosf
#define o_tanf  pl_tanf
#define o_logf  pl_logf
#define o_log2f pl_log2f
#define o_expf  pl_expf
#else
#define o_sinf  sinf
#define o_cosf  cosf
#define o_tanf  tanf
#define o_logf  logf
#define o_log2f log2f
#define o_expf  expf
#endif

#endif /* ORC_LIBM_H */
