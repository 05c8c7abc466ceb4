/*
 * pt_math.h -- host+device math of the path-tracing hot path (product code).
 *
 * Scalar restatement of the subset of Pim's header-only math the path uses; each function
 * names the reference file:line it follows (paths relative to the Pim source tree, src/).
 * Arithmetic ORDER follows the reference.  The translation units that include this header are
 * built with -ffp-contract=off (no implicit FMA), IEEE divide/sqrt and flush-to-zero
 * denormals (threading/task.c:73-74 sets FTZ/DAZ on every worker), so a result computed on
 * gfx950 equals the same expression evaluated in IEEE fp32 on the host, bit for bit.
 *
 * Transcendental functions: the path needs sin, cos, log, log2 and exp.  They are the "pl_"
 * functions below (Cephes single-precision algorithms restated with fp32 + - * / and integer
 * bit operations only), NOT the device library's, so that their results do not depend on the
 * ROCm version and can be reproduced exactly by a CPU checker (DESIGN.md "portable libm").
 */
#ifndef PT_MATH_H
#define PT_MATH_H

#include <stdint.h>
#include <hip/hip_runtime.h>

#define PT_FN static __host__ __device__ __forceinline__

namespace pt {

/* math/scalar.h:21-28 */
#define PT_PI          3.1415926535897932384626433832795f
#define PT_TAU         6.283185307179586476925286766559f
#define PT_RCP_EPS     8388608.0f
#define PT_EPS         (1.0f / ((1u << 23) - 1u))
#define PT_EPS_SQ      (1.0f / 70368744177664.0f)   /* 1/(2^46-1) as the reference writes it; the integer rounds to 2^46 in fp32 */
#define PT_LOG2_EPS    (-23.0f)
#define PT_SQRT5_CONJ  0.2360679774997896964091736687313f
#define PT_MILLI       1e-3f
#define PT_KILO        1e3f
#define PT_DECI        1e-1f
/* math/lighting.h:36-39 */
#define PT_MIN_ALPHA   (1.0f / (1 << 10))
/* rendering/r_config.h:113 */
#define PT_EMISSION_SCALE 100.0f

PT_FN float pt_sqrt(float x) { return __builtin_sqrtf(x); }

/* ---- portable transcendental functions ---- */
PT_FN float pl_asfloat(uint32_t u) { return __builtin_bit_cast(float, u); }
PT_FN uint32_t pl_asuint(float f) { return __builtin_bit_cast(uint32_t, f); }

/* shared octant reduction of |x| by pi/4 (Cephes sinf.c): returns reduced x, writes octant j (0..7) */
PT_FN float pl_reduce_pio4(float x, int32_t* jOut)
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
PT_FN float pl_sin_poly(float x, float z)
{
    float y = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x;
    return y + x;
}
PT_FN float pl_cos_poly(float z)
{
    float y = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}
PT_FN float pl_sinf(float xx)
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
PT_FN float pl_cosf(float xx)
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
PT_FN float pl_tanf(float x) { return pl_sinf(x) / pl_cosf(x); }

/* frexp for positive normal floats: x = m * 2^e, m in [0.5, 1) */
PT_FN float pl_frexp(float x, int32_t* e)
{
    uint32_t u = pl_asuint(x);
    *e = (int32_t)((u >> 23) & 0xffu) - 126;
    return pl_asfloat((u & 0x007fffffu) | 0x3f000000u);
}
PT_FN float pl_log_poly(float x)
{
    return ((((((((7.0376836292e-2f * x - 1.1514610310e-1f) * x + 1.1676998740e-1f) * x - 1.2420140846e-1f) * x
        + 1.4249322787e-1f) * x - 1.6668057665e-1f) * x + 2.0000714765e-1f) * x - 2.4999993993e-1f) * x + 3.3333331174e-1f);
}
/* special cases shared by log and log2: returns 1 and writes *r when x is not a positive normal finite float */
PT_FN int pl_log_special(float x, float* r)
{
    uint32_t u = pl_asuint(x);
    if (x != x) { *r = x; return 1; }
    if ((u & 0x7f800000u) == 0u) { *r = -__builtin_inff(); return 1; }   /* +-0 and denormals (DAZ) */
    if (u & 0x80000000u) { *r = __builtin_nanf(""); return 1; }
    if (u == 0x7f800000u) { *r = x; return 1; }
    return 0;
}
PT_FN float pl_logf(float x)
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
PT_FN float pl_log2f(float x)
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
PT_FN float pl_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return __builtin_inff();
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


typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { v4 c0, c1, c2; } m33;
typedef struct { v4 c0, c1, c2, c3; } m44;
typedef struct { uint32_t x, y, z, w; } u4;



/* ---- scalar: math/scalar.h:65-302 ---- */
PT_FN float f1_sat(float x) { x = (x > 0.0f) ? x : 0.0f; x = (x < 1.0f) ? x : 1.0f; return x; }   /* :66-72 NaN -> 0 */
PT_FN float f1_min(float a, float b) { return a < b ? a : b; }                                   /* :104 */
PT_FN float f1_max(float a, float b) { return a > b ? a : b; }                                   /* :109 */
PT_FN float f1_clamp(float x, float lo, float hi) { return f1_min(f1_max(x, lo), hi); }            /* :114 */
PT_FN float f1_abs(float x) { return f1_max(x, -x); }                                             /* :124 */
PT_FN float f1_sign(float x) { return ((x > 0.0f) ? 1.0f : 0.0f) - ((x < 0.0f) ? 1.0f : 0.0f); } /* :119 */
PT_FN float f1_trunc(float x) { int32_t i = (int32_t)x; return (float)i; }                       /* :129-137 */
PT_FN float f1_floor(float x) { float i = f1_trunc(x); return i - (i > x); }                      /* :139-147 */
PT_FN float f1_ceil(float x) { float i = f1_trunc(x); return i + (x > i); }                       /* :149-157 */
PT_FN float f1_frac(float x) { return x - f1_floor(x); }                                          /* :159 */
PT_FN float f1_lerp(float a, float b, float t) { return a + (b - a) * t; }                       /* :183 */
PT_FN float f1_distance(float a, float b) { return f1_abs(b - a); }                               /* :267 */
PT_FN float f1_radians(float x) { return x * (PT_TAU / 360.0f); }                               /* :74 */
PT_FN float f1_gauss_invcdf(float p, float u, float s) { return u + s * pt_sqrt(-pl_logf(f1_max(1.0f - p, PT_EPS))); } /* :299-302 */
PT_FN int32_t i1_min(int32_t a, int32_t b) { return a < b ? a : b; }
PT_FN int32_t i1_max(int32_t a, int32_t b) { return a > b ? a : b; }
PT_FN int32_t i1_clamp(int32_t x, int32_t lo, int32_t hi) { return i1_min(hi, i1_max(lo, x)); } /* :364 */

/* ---- float2: math/float2_funcs.h ---- */
PT_FN v2 f2_v(float x, float y) { v2 v = { x, y }; return v; }
PT_FN v2 f2_add(v2 a, v2 b) { return f2_v(a.x + b.x, a.y + b.y); }
PT_FN v2 f2_sub(v2 a, v2 b) { return f2_v(a.x - b.x, a.y - b.y); }
PT_FN v2 f2_mul(v2 a, v2 b) { return f2_v(a.x * b.x, a.y * b.y); }
PT_FN v2 f2_muls(v2 a, float s) { return f2_v(a.x * s, a.y * s); }
PT_FN v2 f2_lerps(v2 a, v2 b, float t) { return f2_add(a, f2_muls(f2_sub(b, a), t)); }          /* :337 */
PT_FN v2 f2_bilerp(v2 tl, v2 tr, v2 bl, v2 br, v2 x) { return f2_lerps(f2_lerps(tl, tr, x.x), f2_lerps(bl, br, x.x), x.y); } /* :359 */
PT_FN v2 f2_blend(v2 a, v2 b, v2 c, v4 wuvt)                                                    /* :607-613 */
{
    v2 p = f2_muls(a, wuvt.x);
    p = f2_add(p, f2_muls(b, wuvt.y));
    p = f2_add(p, f2_muls(c, wuvt.z));
    return p;
}

/* ---- float4: math/float4_funcs.h ---- */
PT_FN v4 f4_v(float x, float y, float z, float w) { v4 v = { x, y, z, w }; return v; }
PT_FN v4 f4_s(float s) { return f4_v(s, s, s, s); }
PT_FN v4 f4_add(v4 a, v4 b) { return f4_v(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }          /* :133 */
PT_FN v4 f4_adds(v4 a, float s) { return f4_v(a.x + s, a.y + s, a.z + s, a.w + s); }              /* :139 */
PT_FN v4 f4_sub(v4 a, v4 b) { return f4_v(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }          /* :151 */
PT_FN v4 f4_mul(v4 a, v4 b) { return f4_v(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }          /* :169 */
PT_FN v4 f4_muls(v4 a, float s) { return f4_v(a.x * s, a.y * s, a.z * s, a.w * s); }              /* :175 */
PT_FN v4 f4_divs(v4 a, float s) { return f4_muls(a, 1.0f / s); }                                /* :187 multiply by reciprocal */
PT_FN v4 f4_neg(v4 a) { return f4_v(-a.x, -a.y, -a.z, -a.w); }                                    /* :121 */
PT_FN v4 f4_rcp(v4 a) { return f4_v(1.0f / a.x, 1.0f / a.y, 1.0f / a.z, 1.0f / a.w); }            /* :127 */
PT_FN v4 f4_inv(v4 a) { return f4_sub(f4_s(1.0f), a); }                                         /* :198 */
PT_FN v4 f4_zxy(v4 v) { return f4_v(v.z, v.x, v.y, 0.0f); }                                       /* :109 */
PT_FN float f4_sum3(v4 v) { return v.x + v.y + v.z; }                                           /* :400 */
PT_FN float f4_dot3(v4 a, v4 b) { return f4_sum3(f4_mul(a, b)); }                               /* :546 */
PT_FN float f4_dotsat(v4 a, v4 b) { return f1_sat(f4_dot3(a, b)); }                              /* :551 */
PT_FN v4 f4_cross3(v4 a, v4 b) { return f4_zxy(f4_sub(f4_mul(f4_zxy(a), b), f4_mul(a, f4_zxy(b)))); } /* :556 */
PT_FN float f4_length3(v4 x) { return pt_sqrt(f1_max(PT_EPS_SQ, f4_dot3(x, x))); }                /* :568 */
PT_FN v4 f4_normalize3(v4 x) { return f4_divs(x, f4_length3(x)); }                              /* :577 */
PT_FN float f4_lengthsq3(v4 x) { return f4_dot3(x, x); }                                        /* :595 */
PT_FN v4 f4_min(v4 a, v4 b) { return f4_v(f1_min(a.x, b.x), f1_min(a.y, b.y), f1_min(a.z, b.z), f1_min(a.w, b.w)); }
PT_FN v4 f4_max(v4 a, v4 b) { return f4_v(f1_max(a.x, b.x), f1_max(a.y, b.y), f1_max(a.z, b.z), f1_max(a.w, b.w)); }
PT_FN float f4_hmax3(v4 v) { return f1_max(v.x, f1_max(v.y, v.z)); }                              /* :519 */
PT_FN float f4_hmin3(v4 v) { return f1_min(v.x, f1_min(v.y, v.z)); }                              /* :510 */
PT_FN v4 f4_abs(v4 v) { return f4_v(f1_abs(v.x), f1_abs(v.y), f1_abs(v.z), f1_abs(v.w)); }            /* :727 */
PT_FN v4 f4_lerp(v4 a, v4 b, v4 t) { return f4_add(a, f4_mul(f4_sub(b, a), t)); }               /* :609 */
PT_FN v4 f4_lerps(v4 a, v4 b, float t) { return f4_add(a, f4_muls(f4_sub(b, a), t)); }          /* :613 f4_lerpvs */
PT_FN v4 f4_lerpsv(float a, float b, v4 t) { v4 m = f4_muls(t, b - a); return f4_v(a + m.x, a + m.y, a + m.z, a + m.w); } /* :617 */
PT_FN v4 f4_bilerp(v4 tl, v4 tr, v4 bl, v4 br, v2 x) { return f4_lerps(f4_lerps(tl, tr, x.x), f4_lerps(bl, br, x.x), x.y); } /* :636 */
PT_FN v4 f4_blend(v4 a, v4 b, v4 c, v4 wuvt)                                                    /* :995-1001 */
{
    v4 p = f4_muls(a, wuvt.x);
    p = f4_add(p, f4_muls(b, wuvt.y));
    p = f4_add(p, f4_muls(c, wuvt.z));
    return p;
}
PT_FN v4 f4_reflect3(v4 i, v4 n)                                                                /* :707-711 */
{
    v4 nidn = f4_muls(n, f4_dot3(i, n));
    return f4_sub(i, f4_muls(nidn, 2.0f));
}
PT_FN v4 f4_refract3(v4 i, v4 n, float k)                                                       /* :713-719 */
{
    float cosTheta = f1_min(1.0f, f4_dot3(f4_neg(i), n));
    v4 perp = f4_muls(f4_add(f4_muls(n, cosTheta), i), k);
    v4 para = f4_muls(n, -pt_sqrt(f1_abs(1.0f - f4_lengthsq3(perp))));
    return f4_normalize3(f4_add(perp, para));
}
PT_FN v4 f4_floor(v4 v) { return f4_v(f1_floor(v.x), f1_floor(v.y), f1_floor(v.z), f1_floor(v.w)); }  /* :901 */
PT_FN v4 f4_ceil(v4 v) { return f4_v(f1_ceil(v.x), f1_ceil(v.y), f1_ceil(v.z), f1_ceil(v.w)); }       /* :913 */
PT_FN v4 f4_frac(v4 v) { return f4_sub(v, f4_floor(v)); }                                       /* :925 */
PT_FN v4 f4_unormstep(v4 t) { return f4_mul(f4_mul(f4_adds(f4_muls(t, -2.0f), 3.0f), t), t); }  /* :672 */
PT_FN float f4_avglum(v4 x) { return (x.x + x.y + x.z) * (1.0f / 3.0f); }                       /* math/color.h:261 */

PT_FN v3 f3_v(float x, float y, float z) { v3 v = { x, y, z }; return v; }
PT_FN v3 f3_lerps(v3 a, v3 b, float t)                                                          /* math/float3_funcs.h:373 */
{
    return f3_v(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t);
}
PT_FN v3 f3_bilerp(v3 tl, v3 tr, v3 bl, v3 br, v2 x) { return f3_lerps(f3_lerps(tl, tr, x.x), f3_lerps(bl, br, x.x), x.y); } /* :395 */

/* ---- matrices: math/float3x3_funcs.h, float4x4_funcs.h, quat_funcs.h ---- */
PT_FN v4 m33_mul_col(m33 m, v4 c)                                                               /* float3x3_funcs.h:150-157 */
{
    return f4_add(f4_muls(m.c0, c.x), f4_add(f4_muls(m.c1, c.y), f4_muls(m.c2, c.z)));
}
PT_FN m33 m33_transpose(m33 a)                                                                  /* :54-69 */
{
    m33 b;
    b.c0 = f4_v(a.c0.x, a.c1.x, a.c2.x, 0.0f);
    b.c1 = f4_v(a.c0.y, a.c1.y, a.c2.y, 0.0f);
    b.c2 = f4_v(a.c0.z, a.c1.z, a.c2.z, 0.0f);
    return b;
}
PT_FN m33 m33_inverse(m33 m)                                                                    /* :71-105, [column][row] accessors */
{
    float m00 = m.c0.x, m01 = m.c0.y, m02 = m.c0.z;
    float m10 = m.c1.x, m11 = m.c1.y, m12 = m.c1.z;
    float m20 = m.c2.x, m21 = m.c2.y, m22 = m.c2.z;
    float a = m00 * m11 - m10 * m01;
    float b = m00 * m12 - m10 * m02;
    float c = m00 * m21 - m20 * m01;
    float d = m00 * m22 - m20 * m02;
    float e = m01 * m12 - m11 * m02;
    float f = m01 * m22 - m21 * m02;
    float g = m10 * m21 - m20 * m11;
    float h = m10 * m22 - m20 * m12;
    float i = m11 * m22 - m21 * m12;
    float d0 = m00 * i;
    float d1 = -m10 * f;
    float d2 = m20 * e;
    float det = d0 + d1 + d2;
    float r = 1.0f / det;
    m33 inv;
    inv.c0 = f4_v(r * i, -r * f, r * e, 0.0f);
    inv.c1 = f4_v(-r * h, r * d, -r * b, 0.0f);
    inv.c2 = f4_v(r * g, -r * c, r * a, 0.0f);
    return inv;
}
PT_FN m33 m44_IM(m44 M)                                                                         /* f3x3_IM :107-110 */
{
    m33 t; t.c0 = M.c0; t.c1 = M.c1; t.c2 = M.c2;
    t.c0.w = t.c1.w = t.c2.w = 0.0f;
    return m33_inverse(m33_transpose(t));
}
PT_FN v4 m44_mul_dir(m44 m, v4 d)                                                               /* float4x4_funcs.h:78-84 */
{
    v4 a = f4_muls(m.c0, d.x), b = f4_muls(m.c1, d.y), c = f4_muls(m.c2, d.z);
    return f4_add(a, f4_add(b, c));
}
PT_FN v4 m44_mul_pt(m44 m, v4 p)                                                                /* :87-93 */
{
    v4 a = f4_muls(m.c0, p.x), b = f4_muls(m.c1, p.y), c = f4_muls(m.c2, p.z);
    return f4_add(f4_add(a, b), f4_add(c, m.c3));
}
PT_FN m33 quat_to_m33(v4 v)                                                                     /* quat_f3x3, quat_funcs.h:131-158 */
{
    float xx = v.x * v.x, xy = v.x * v.y, xz = v.x * v.z;
    float yy = v.y * v.y, yz = v.y * v.z, zz = v.z * v.z;
    float wx = v.w * v.x, wy = v.w * v.y, wz = v.w * v.z;
    m33 m;
    m.c0 = f4_v(1.0f - 2.0f * (yy + zz), 2.0f * (xy - wz), 2.0f * (xz + wy), 0.0f);
    m.c1 = f4_v(2.0f * (xy + wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz - wx), 0.0f);
    m.c2 = f4_v(2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (xx + yy), 0.0f);
    return m;
}
PT_FN m44 m44_trs(v4 t, v4 q, v4 s)                                                             /* f4x4_trs, float4x4_funcs.h:128-192 */
{
    m44 T;
    T.c0 = f4_v(1, 0, 0, 0); T.c1 = f4_v(0, 1, 0, 0); T.c2 = f4_v(0, 0, 1, 0); T.c3 = f4_v(t.x, t.y, t.z, 1.0f);
    m33 rot = quat_to_m33(q);
    m44 R;
    R.c0 = m44_mul_dir(T, rot.c0); R.c1 = m44_mul_dir(T, rot.c1); R.c2 = m44_mul_dir(T, rot.c2); R.c3 = T.c3;
    m44 S;
    S.c0 = f4_muls(R.c0, s.x); S.c1 = f4_muls(R.c1, s.y); S.c2 = f4_muls(R.c2, s.z); S.c3 = R.c3;
    return S;
}
PT_FN v4 quat_mul_dir(v4 q, v4 dir)                                                             /* quat_funcs.h:86-97 */
{
    v4 uv = f4_cross3(q, dir);
    v4 uuv = f4_cross3(q, uv);
    v4 y = f4_muls(uv, q.w);
    y = f4_add(y, uuv);
    y = f4_muls(y, 2.0f);
    y = f4_add(y, dir);
    return y;
}
PT_FN v4 quat_fwd(v4 q) { return quat_mul_dir(q, f4_v(0.0f, 0.0f, -1.0f, 0.0f)); }                /* :99 */
PT_FN v4 quat_up(v4 q) { return quat_mul_dir(q, f4_v(0.0f, 1.0f, 0.0f, 0.0f)); }                  /* :105 */
PT_FN v4 quat_right(v4 q) { return quat_mul_dir(q, f4_v(1.0f, 0.0f, 0.0f, 0.0f)); }               /* :111 */
PT_FN v4 m33_to_quat(m33 m)                                                                     /* f3x3_quat :160-221 */
{
    float m00 = m.c0.x, m01 = m.c0.y, m02 = m.c0.z;
    float m10 = m.c1.x, m11 = m.c1.y, m12 = m.c1.z;
    float m20 = m.c2.x, m21 = m.c2.y, m22 = m.c2.z;
    float x = m00 - m11 - m22, y = m11 - m00 - m22, z = m22 - m00 - m11, w = m00 + m11 + m22;
    float b = f1_max(f1_max(x, y), f1_max(z, w));
    int i = 0;
    i = (b == x) ? 1 : i;
    i = (b == y) ? 2 : i;
    i = (b == z) ? 3 : i;
    b = pt_sqrt(f1_max(b + 1.0f, 4.0f * PT_EPS_SQ)) * 0.5f;
    float mult = 0.25f / b;
    v4 r = f4_v(0, 0, 0, 1);
    switch (i)
    {
    case 0: r.w = b; r.x = (m12 - m21) * mult; r.y = (m20 - m02) * mult; r.z = (m01 - m10) * mult; break;
    case 1: r.w = (m12 - m21) * mult; r.x = b; r.y = (m01 + m10) * mult; r.z = (m20 + m02) * mult; break;
    case 2: r.w = (m20 - m02) * mult; r.x = (m01 + m10) * mult; r.y = b; r.z = (m12 + m21) * mult; break;
    default: r.w = (m01 - m10) * mult; r.x = (m20 + m02) * mult; r.y = (m12 + m21) * mult; r.z = b; break;
    }
    return r;
}
PT_FN v4 quat_lookat(v4 forward, v4 up)                                                         /* :223-233 */
{
    forward = f4_neg(forward);
    v4 right = f4_normalize3(f4_cross3(up, forward));
    up = f4_normalize3(f4_cross3(forward, right));
    m33 m; m.c2 = forward; m.c0 = right; m.c1 = up;
    return m33_to_quat(m);
}

/* ---- frustum: math/frustum.h:26-47 ---- */
PT_FN v2 proj_slope(float fovy, float aspect) { float t = pl_tanf(fovy * 0.5f); return f2_v(aspect * t, t); }
PT_FN v4 proj_dir(v4 right, v4 up, v4 fwd, v2 slope, v2 coord)
{
    coord = f2_mul(coord, slope);
    right = f4_muls(right, coord.x);
    up = f4_muls(up, coord.y);
    fwd = f4_add(fwd, right);
    fwd = f4_add(fwd, up);
    return f4_normalize3(fwd);
}

/* ---- RNG: math/pcg.h:16-171, common/random.h:13-146 ---- */
PT_FN uint32_t pcg1(uint32_t v)
{
    v = v * 747796405u + 2891336453u;
    v = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    v = (v >> 22) ^ v;
    return v;
}
PT_FN void pcg2(uint32_t* px, uint32_t* py)
{
    uint32_t x = *px, y = *py;
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    x += y * 1664525u; y += x * 1664525u;
    x ^= x >> 16; y ^= y >> 16;
    x += y * 1664525u; y += x * 1664525u;
    *px = x; *py = y;
}
PT_FN void pcg3(uint32_t* px, uint32_t* py, uint32_t* pz)
{
    uint32_t x = *px, y = *py, z = *pz;
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
    *px = x; *py = y; *pz = z;
}
PT_FN u4 pcg4(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    u4 r = { x, y, z, w };
    return r;
}
PT_FN float rng_tofloat(uint32_t x) { return (x >> 8) * (1.0f / (1 << 24)); }                   /* random.h:108 */
PT_FN uint32_t rng_u32(u4* s) { s->x = pcg1(s->x); return s->x; }                               /* :37 */
PT_FN float rng_f32(u4* s) { s->x = pcg1(s->x); return rng_tofloat(s->x); }                     /* :112 */
PT_FN v2 rng_f2(u4* s) { pcg2(&s->x, &s->y); return f2_v(rng_tofloat(s->x), rng_tofloat(s->y)); } /* :117 */
PT_FN v4 rng_f4(u4* s)                                                                          /* :136 */
{
    *s = pcg4(s->x, s->y, s->z, s->w);
    return f4_v(rng_tofloat(s->x), rng_tofloat(s->y), rng_tofloat(s->z), rng_tofloat(s->w));
}

/* ---- sampling: math/sampling.h ---- */
PT_FN m33 NormalToTBN(v4 N)                                                                     /* :26-60 */
{
    float s = (N.z < 0.0f) ? -1.0f : 1.0f;
    const float a = -1.0f / (s + N.z);
    const float b = N.x * N.y * a;
    m33 t;
    t.c0 = f4_v(1.0f + s * N.x * N.x * a, s * b, -s * N.x, 0.0f);
    t.c1 = f4_v(b, s + N.y * N.y * a, -N.y, 0.0f);
    t.c2 = N;
    return t;
}
PT_FN v4 TbnToWorld(m33 TBN, v4 nTS) { return m33_mul_col(TBN, nTS); }                          /* :62 */
PT_FN v4 TanToWorld(v4 nWS, v4 nTS) { return TbnToWorld(NormalToTBN(nWS), nTS); }               /* :69 */
PT_FN float RadicalInverseBase2(uint32_t bits)                                                  /* :75-83 */
{
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)(bits) * 2.3283064365386963e-10f;
}
PT_FN v2 Hammersley2D(uint32_t i, uint32_t N) { return f2_v((i + 0.5f) / N, RadicalInverseBase2(i)); } /* :86 */
PT_FN float PowerHeuristic(float f, float g) { return (f * f) / f1_max(f * f + g * g, PT_EPS); } /* :93 */
PT_FN v2 MapSquareToDisk(v2 Xi)                                                                 /* :100-118 */
{
    const float lo = PT_EPS, hi = 1.0f - PT_EPS;
    Xi = f2_v(lo + (hi - lo) * Xi.x, lo + (hi - lo) * Xi.y);   /* f2_lerpsv = a + (b-a)*t */
    float phi, r;
    float a = 2.0f * Xi.x - 1.0f;
    float b = 2.0f * Xi.y - 1.0f;
    if ((a * a) > (b * b)) { r = a; phi = (PT_PI / 4.0f) * (b / a); }
    else { r = b; phi = (PT_PI / 2.0f) - (PT_PI / 4.0f) * (a / b); }
    return f2_v(r * pl_cosf(phi), r * pl_sinf(phi));
}
PT_FN v4 SampleBaryCoord(v2 Xi)                                                                 /* :120-128 */
{
    float r1 = pt_sqrt(f1_max(Xi.x, PT_EPS_SQ));
    float r2 = Xi.y;
    float u = r1 * (1.0f - r2);
    float v = r2 * r1;
    float w = 1.0f - (u + v);
    return f4_v(w, u, v, 0.0f);
}
PT_FN v2 SampleNGon(v2 Xi, uint32_t side, uint32_t N, float rot)                                /* :130-139 */
{
    side %= N;
    const float R = PT_TAU / N;
    const float a = rot + (1 + side) * R;
    const float b = rot + (2 + side) * R;
    const v2 A = { pl_cosf(a), pl_sinf(a) };
    const v2 B = { pl_cosf(b), pl_sinf(b) };
    return f2_blend(f2_v(0.0f, 0.0f), A, B, SampleBaryCoord(Xi));
}
PT_FN v2 SamplePentagram(v2 Xi, uint32_t side)                                                  /* :141-156 */
{
    const float R = PT_TAU / 5.0f;
    const float s = PT_PI * 0.1f;
    const float q = (1.0f - PT_SQRT5_CONJ) * 0.5f;
    side %= 5;
    const float a = s + (1.0f + side) * R;
    const float b = s + (1.5f + side) * R;
    const float c = s + (2.0f + side) * R;
    const v2 A = { q * pl_cosf(a), q * pl_sinf(a) };
    const v2 B = { pl_cosf(b), pl_sinf(b) };
    const v2 C = { q * pl_cosf(c), q * pl_sinf(c) };
    return f2_bilerp(A, B, f2_v(0.0f, 0.0f), C, Xi);
}
PT_FN v4 SphericalToCartesian(float cosTheta, float phi)                                        /* :158-165 */
{
    float sinTheta = pt_sqrt(f1_max(1.0f - cosTheta * cosTheta, 0.0f));
    return f4_v(sinTheta * pl_cosf(phi), sinTheta * pl_sinf(phi), cosTheta, 0.0f);
}
PT_FN v4 SampleUnitSphere(v2 Xi) { return SphericalToCartesian(Xi.y * 2.0f - 1.0f, PT_TAU * Xi.x); } /* :168 */
PT_FN v4 SampleUnitHemisphere(v2 Xi) { return SphericalToCartesian(Xi.y, PT_TAU * Xi.x); }     /* :176 */
PT_FN v4 SampleCosineHemisphere(v2 Xi)                                                          /* :271-276 */
{
    Xi = MapSquareToDisk(Xi);
    float z = pt_sqrt(f1_max(1.0f - (Xi.x * Xi.x + Xi.y * Xi.y), PT_EPS_SQ));
    return f4_v(Xi.x, Xi.y, z, 0.0f);
}
PT_FN v4 SampleGGXMicrofacet(v2 Xi, float alpha)                                                /* :280-287 */
{
    float a2 = alpha * alpha;
    float phi = PT_TAU * Xi.x;
    float b = f1_max(1.0f + (a2 - 1.0f) * Xi.y, PT_EPS);
    float cosTheta = pt_sqrt(f1_max((1.0f - Xi.y) / b, PT_EPS_SQ));
    return SphericalToCartesian(cosTheta, phi);
}
PT_FN float LambertPdf(float NoL) { return NoL * (1.0f / PT_PI); }                             /* :301 */
PT_FN float LightPdf(float area, float cosTheta, float distSq) { float t = cosTheta * area; return distSq / f1_max(t, PT_EPS); } /* :321 */
PT_FN v2 SampleGaussPixelFilter(v2 Xi, float stddev)                                            /* :327-335 */
{
    float angle = Xi.x * PT_TAU;
    float radius = f1_gauss_invcdf(Xi.y, 0.0f, stddev);
    Xi.x = pl_cosf(angle);
    Xi.y = pl_sinf(angle);
    return f2_muls(Xi, radius);
}
PT_FN float SampleFreePath(float Xi, float mfp) { return -pl_logf(f1_max(1.0f - Xi, PT_EPS)) * mfp; } /* :340 */

/* ---- BRDF terms: math/lighting.h ---- */
PT_FN float BrdfAlpha(float roughness) { return f1_max(roughness * roughness, PT_MIN_ALPHA); }   /* :57 */
PT_FN v4 F_0(v4 albedo, float metallic) { return f4_lerps(f4_s(0.04f), albedo, metallic); }     /* :69 */
PT_FN float F_90(v4 F0) { return f1_sat(50.0f * f4_dot3(F0, f4_s(0.33f))); }                     /* :75 */
PT_FN float F_Schlick1(float f0, float f90, float cosTheta)                                     /* :97 */
{
    float t = 1.0f - cosTheta;
    float t5 = t * t * t * t * t;
    return f1_lerp(f0, f90, t5);
}
PT_FN float F_Dielectric(float cosThetaI, float etaI, float etaT)                               /* :138-162 */
{
    cosThetaI = f1_clamp(cosThetaI, -1.0f, 1.0f);
    if (cosThetaI < 0.0f)
    {
        cosThetaI = -cosThetaI;
        float tmp = etaI; etaI = etaT; etaT = tmp;
    }
    float sinThetaI = pt_sqrt(f1_max(1.0f - cosThetaI * cosThetaI, PT_EPS_SQ));
    float sinThetaT = (etaI / etaT) * sinThetaI;
    if (sinThetaT >= 1.0f)
    {
        return 1.0f;
    }
    float cosThetaT = pt_sqrt(f1_max(1.0f - sinThetaT * sinThetaT, PT_EPS_SQ));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return f1_sat((Rparl * Rparl + Rperp * Rperp) * 0.5f);
}
PT_FN v4 SigmaAFromReflectance(v4 albedo, float beta_n)                                         /* :193-206 */
{
    float r2 = beta_n * beta_n;
    float r3 = r2 * beta_n;
    float r4 = r3 * beta_n;
    float r5 = r4 * beta_n;
    float t = 5.969f - 0.215f * beta_n + 2.532f * r2 - 10.73f * r3 + 5.574f * r4 + 0.245f * r5;
    v4 s = f4_v(pl_logf(albedo.x), pl_logf(albedo.y), pl_logf(albedo.z), pl_logf(albedo.w));
    s = f4_divs(s, f1_max(t, PT_EPS));
    return f4_mul(s, s);
}
PT_FN v4 AlbedoToTransmittance(v4 albedo, float roughness, float thickness)                     /* :208-212 */
{
    v4 s = f4_muls(SigmaAFromReflectance(albedo, roughness), -thickness);
    return f4_v(pl_expf(s.x), pl_expf(s.y), pl_expf(s.z), pl_expf(s.w));
}
PT_FN float D_GTR(float NoH, float alpha)                                                       /* :218-224 */
{
    float a2 = alpha * alpha;
    float f = f1_lerp(1.0f, a2, NoH * NoH);
    f = f * f * PT_PI;
    return a2 / f1_max(f, PT_EPS);
}
PT_FN float GGXPdf(float NoH, float HoV, float alpha)                                           /* sampling.h:311-315 */
{
    float d = D_GTR(NoH, alpha);
    return (d * NoH) / f1_max(4.0f * HoV, PT_EPS);
}
PT_FN float V_SmithCorrelated(float NoL, float NoV, float alpha)                                /* :246-253 */
{
    float a2 = alpha * alpha;
    float v = NoL * pt_sqrt(f1_max(a2 + (NoV - NoV * a2) * NoV, PT_EPS_SQ));
    float l = NoV * pt_sqrt(f1_max(a2 + (NoL - NoL * a2) * NoL, PT_EPS_SQ));
    float t = v + l;
    return 0.5f / f1_max(t, PT_EPS);
}
PT_FN float Fd_Burley(float NoL, float NoV, float HoV, float roughness)                         /* :266-276 */
{
    float fd90 = 0.5f + 2.0f * HoV * HoV * roughness;
    float lightScatter = F_Schlick1(1.0f, fd90, NoL);
    float viewScatter = F_Schlick1(1.0f, fd90, NoV);
    return (lightScatter * viewScatter) / PT_PI;
}

/* one sample of BakeBrdfFn's loop body (math/lighting.c:53-80) for LUT texel (cx, cy): the sample's weighted
 * contribution (D*G*NoL*dx*Fc, D*G*NoL*dx), zero when the reflected direction is below the horizon */
PT_FN v2 BrdfLutSample(int32_t cx, int32_t cy, float rcpWidth, float rcpHeight, float dx, v2 XiAA, float XiPhi, v2 XiH)
{
    const v2 aa = SampleGaussPixelFilter(XiAA, 0.5f);
    const float NoV = f1_clamp(((float)cx + aa.x) * rcpWidth, PT_EPS, 1.0f - PT_EPS);
    const float alpha = f1_clamp(((float)cy + aa.y) * rcpHeight, PT_MIN_ALPHA, 1.0f);
    const v4 V = SphericalToCartesian(NoV, XiPhi * PT_TAU);
    const v4 H = SampleGGXMicrofacet(XiH, alpha);                      /* N = (0, 0, 1) */
    const v4 L = f4_reflect3(f4_neg(V), H);
    const float NoL = L.z, NoH = H.z;
    const float HoV = f4_dot3(H, V);
    const float pdf = GGXPdf(NoH, HoV, alpha);
    v2 r = f2_v(0.0f, 0.0f);
    if ((NoL > PT_EPS) && (pdf > PT_EPS))
    {
        const float D = D_GTR(NoH, alpha) / pdf;
        const float G = V_SmithCorrelated(NoL, NoV, alpha);
        const float Fc = F_Dielectric(HoV, 1.000293f, 1.52f);
        const float DG_NoL_dx = D * G * NoL * dx;
        r.x = DG_NoL_dx * Fc;
        r.y = DG_NoL_dx;
    }
    return r;
}

/* ---- colour / texel decode: math/color.h ---- */
PT_FN v4 rgba8_to_f4(uint32_t c)                                                                /* :113-117 (r = low byte) */
{
    const float s = 1.0f / 255.0f;
    return f4_v((c & 0xff) * s, ((c >> 8) & 0xff) * s, ((c >> 16) & 0xff) * s, ((c >> 24) & 0xff) * s);
}
PT_FN uint32_t f4_to_rgba8(v4 v)                                                                /* :102-112 */
{
    v = f4_v(f1_clamp(v.x, 0.0f, 1.0f), f1_clamp(v.y, 0.0f, 1.0f), f1_clamp(v.z, 0.0f, 1.0f), f1_clamp(v.w, 0.0f, 1.0f));
    v = f4_adds(f4_muls(v, 255.0f), 0.5f);
    return ((uint32_t)v.x & 0xff) | (((uint32_t)v.y & 0xff) << 8) | (((uint32_t)v.z & 0xff) << 16) | (((uint32_t)v.w & 0xff) << 24);
}
PT_FN v4 sRGB_EOTF_Fit(v4 V)                                                                    /* :193-196 */
{
    return f4_add(f4_add(f4_muls(V, 0.020883f), f4_muls(f4_mul(V, V), 0.656075f)), f4_muls(f4_mul(V, f4_mul(V, V)), 0.324285f));
}
PT_FN v4 sRGB_InverseEOTF_Fit(v4 L)                                                             /* :212-218 */
{
    v4 l1 = f4_v(pt_sqrt(L.x), pt_sqrt(L.y), pt_sqrt(L.z), pt_sqrt(L.w));
    v4 l2 = f4_v(pt_sqrt(l1.x), pt_sqrt(l1.y), pt_sqrt(l1.z), pt_sqrt(l1.w));
    v4 l3 = f4_v(pt_sqrt(l2.x), pt_sqrt(l2.y), pt_sqrt(l2.z), pt_sqrt(l2.w));
    return f4_add(f4_add(f4_muls(l1, 0.658444f), f4_muls(l2, 0.643378f)), f4_muls(l3, -0.298148f));
}
PT_FN uint32_t GammaEncode_rgba8(v4 lin) { return f4_to_rgba8(sRGB_InverseEOTF_Fit(lin)); }     /* :251 */
PT_FN v4 GammaDecode_rgba8(uint32_t c) { return sRGB_EOTF_Fit(rgba8_to_f4(c)); }                /* :256 */
PT_FN v4 Xy16ToNormalTs(uint32_t packed)                                                        /* :241-249 (short2: x low half) */
{
    int16_t sx = (int16_t)(packed & 0xffff), sy = (int16_t)(packed >> 16);
    v4 n;
    n.x = sx * (1.0f / (1 << 15));
    n.y = sy * (1.0f / (1 << 15));
    n.z = pt_sqrt(f1_max(PT_EPS_SQ, 1.0f - (n.x * n.x + n.y * n.y)));
    n.w = 0.0f;
    return n;
}
PT_FN v4 UnpackEmission(v4 albedo, float e) { return f4_muls(albedo, (e * e) * PT_EMISSION_SCALE); } /* :588 */

/* ---- bilinear addressing: rendering/sampler.h:176-229 ---- */
typedef struct { int32_t a, b, c, d; v2 frac; } Bilinear;
PT_FN void LinearClamp(int32_t size, float u, int32_t* a, int32_t* b, float* t)                 /* :176-184 */
{
    float x = f1_sat(u) * (size - 1);
    *a = (int32_t)x;
    *t = x - (float)*a;
    *b = i1_min(*a + 1, size - 1);
}
PT_FN Bilinear BilinearClamp(int32_t sx, int32_t sy, v2 uv)                                 /* :210-222 */
{
    Bilinear r;
    int32_t xa, xb, ya, yb;
    LinearClamp(sx, uv.x, &xa, &xb, &r.frac.x);
    LinearClamp(sy, uv.y, &ya, &yb, &r.frac.y);
    r.a = ya * sx + xa; r.b = ya * sx + xb; r.c = yb * sx + xa; r.d = yb * sx + xb;
    return r;
}
PT_FN Bilinear BilinearWrap(int32_t sx, int32_t sy, v2 uv)                                  /* :223-229 */
{
    uv.x = (uv.x >= 0.0f) ? uv.x : (1.0f - uv.x);
    uv.y = (uv.y >= 0.0f) ? uv.y : (1.0f - uv.y);
    uv = f2_v(f1_frac(uv.x), f1_frac(uv.y));
    return BilinearClamp(sx, sy, uv);
}
PT_FN v4 UvBilinearWrap_c32(const uint32_t* tex, int32_t sx, int32_t sy, v2 uv)                 /* :375-379, :240-249 */
{
    /* a flat texture (GenFlatTexture's 1x1, render_system.c:1119-1134: every material of the Cornell scenes): the four taps are
     * the one texel and lerp(a, a, t) = a + (a - a) * t = a exactly, so one fetch and one decode give the same bits */
    if ((sx | sy) == 1) return GammaDecode_rgba8(tex[0]);
    Bilinear bi = BilinearWrap(sx, sy, uv);
    return f4_bilerp(GammaDecode_rgba8(tex[bi.a]), GammaDecode_rgba8(tex[bi.b]), GammaDecode_rgba8(tex[bi.c]), GammaDecode_rgba8(tex[bi.d]), bi.frac);
}
PT_FN v4 UvBilinearWrap_xy16(const uint32_t* tex, int32_t sx, int32_t sy, v2 uv)                /* :388-392, :250-260 */
{
    if ((sx | sy) == 1) return f4_normalize3(Xy16ToNormalTs(tex[0]));          /* flat: as above */
    Bilinear bi = BilinearWrap(sx, sy, uv);
    v4 r = f4_bilerp(Xy16ToNormalTs(tex[bi.a]), Xy16ToNormalTs(tex[bi.b]), Xy16ToNormalTs(tex[bi.c]), Xy16ToNormalTs(tex[bi.d]), bi.frac);
    return f4_normalize3(r);
}
PT_FN v2 UvBilinearClamp_f2(const v2* tex, int32_t sx, int32_t sy, v2 uv)                       /* :347-351 */
{
    Bilinear bi = BilinearClamp(sx, sy, uv);
    return f2_bilerp(tex[bi.a], tex[bi.b], tex[bi.c], tex[bi.d], bi.frac);
}
PT_FN v3 UvBilinearClamp_f3(const v3* tex, int32_t sx, int32_t sy, v2 uv)                       /* :337-341 */
{
    Bilinear bi = BilinearClamp(sx, sy, uv);
    return f3_bilerp(tex[bi.a], tex[bi.b], tex[bi.c], tex[bi.d], bi.frac);
}
PT_FN int32_t PointClamp1D(int32_t size, float u) { return (int32_t)(f1_sat(u) * (size - 1) + 0.5f); } /* :105 */
PT_FN int32_t PointWrap1D(int32_t size, float u)                                                /* :109-114 */
{
    u = (u >= 0.0f) ? u : (1.0f - u);
    u = f1_frac(u);
    return PointClamp1D(size, u);
}

/* ---- cubemap addressing: rendering/cubemap.h:71-100, cubemap.c:24-42 ---- */
PT_FN int Cubemap_CalcUv(v4 dir, v2* uvOut)
{
    v4 absdir = f4_abs(dir);
    float v = f4_hmax3(absdir);
    float ma = 0.5f / v;
    int face;
    if (v == absdir.x) face = dir.x < 0.0f ? 1 : 0;
    else if (v == absdir.y) face = dir.y < 0.0f ? 3 : 2;
    else face = dir.z < 0.0f ? 5 : 4;
    /* Cubemap_kRights / Cubemap_kUps, cubemap.c:24-42 */
    v4 right = f4_v((face == 2 || face == 4) ? 1.0f : ((face == 3 || face == 5) ? -1.0f : 0.0f), 0.0f,
        (face == 0) ? -1.0f : ((face == 1) ? 1.0f : 0.0f), 0.0f);
    v4 up = f4_v(0.0f, (face == 2 || face == 3) ? 0.0f : 1.0f, (face == 2 || face == 3) ? -1.0f : 0.0f, 0.0f);
    v2 uv;
    uv.x = f4_dot3(right, dir);
    uv.y = f4_dot3(up, dir);
    uv = f2_v(uv.x * ma + 0.5f, uv.y * ma + 0.5f);
    *uvOut = uv;
    return face;
}

/* ---- the convolved mip chain of a cubemap face: rendering/sampler.h:23-51,85-88, rendering/cubemap.h:60-69,117-129,169-188 ---- */
PT_FN int32_t i1_log2(int32_t x) { return x > 0 ? 31 - __builtin_clz((uint32_t)x) : -1; }         /* math/scalar.h:392-422 */
PT_FN int32_t CalcMipCount(int32_t size) { return 1 + i1_log2(size); }
PT_FN int32_t CalcMipSize(int32_t size, int32_t m) { const int32_t y = size >> m; return y > 1 ? y : 1; }
PT_FN int32_t CalcMipOffset(int32_t size, int32_t m)
{
    const int32_t last = CalcMipCount(size) - 1;
    m = m < 0 ? 0 : (m > last ? last : m);
    int32_t y = 0;
    for (int32_t i = 0; i < m; ++i) { const int32_t ms = CalcMipSize(size, i); y += ms * ms; }
    return y;
}
PT_FN int32_t CubemapElemCount(int32_t size) { return CalcMipOffset(size, CalcMipCount(size)) + 1; }  /* Cubemap_New, cubemap.c:105-106 */
#define PT_CUBEMAP_MAX_MIP 6.0f                                                                  /* cubemap.h:12 */
PT_FN float MipToRoughness(float mip) { return mip / PT_CUBEMAP_MAX_MIP; }
PT_FN v3 Cubemap_ReadColor(const v3* const* color, int32_t size, v4 dir)
{
    v2 uv; const int face = Cubemap_CalcUv(dir, &uv);
    return UvBilinearClamp_f3(color[face], size, size, uv);
}

/* f2_tent, math/float2_funcs.h:630-636.  As written in the reference: the y lane's upper branch reads
 * Xi.x AFTER it was mapped (2 - Xi.x, not 2 - Xi.y); kept, the reference's golden vectors pin it. */
PT_FN v2 f2_tent(v2 Xi)
{
    Xi = f2_v(Xi.x * 2.0f, Xi.y * 2.0f);
    Xi.x = Xi.x < 1.0f ? pt_sqrt(Xi.x) - 1.0f : 1.0f - pt_sqrt(2.0f - Xi.x);
    Xi.y = Xi.y < 1.0f ? pt_sqrt(Xi.y) - 1.0f : 1.0f - pt_sqrt(2.0f - Xi.x);
    return Xi;
}

/* Cubemap_CalcDir, rendering/cubemap.h:190-207 (BilinearInvert / LinearInvert rendering/sampler.h:192-236,
 * f2_snorm math/float2_funcs.h:620, Cubemap_kForwards / kRights / kUps rendering/cubemap.c:14-42) */
PT_FN v4 Cubemap_CalcDir(int32_t size, int face, int32_t cx, int32_t cy, v2 Xi)
{
    v2 uv = f2_v((float)cx / (float)(size - 1), (float)cy / (float)(size - 1));
    const float rcp = 1.0f / size;
    Xi = f2_v(Xi.x * rcp, Xi.y * rcp);
    uv = f2_v(uv.x + Xi.x, uv.y + Xi.y);
    uv = f2_v(uv.x * 2.0f - 1.0f, uv.y * 2.0f - 1.0f);
    const v4 fwd = f4_v(face == 0 ? 1.0f : (face == 1 ? -1.0f : 0.0f), face == 2 ? 1.0f : (face == 3 ? -1.0f : 0.0f),
        face == 4 ? 1.0f : (face == 5 ? -1.0f : 0.0f), 0.0f);
    const v4 right = f4_v((face == 2 || face == 4) ? 1.0f : ((face == 3 || face == 5) ? -1.0f : 0.0f), 0.0f,
        (face == 0) ? -1.0f : ((face == 1) ? 1.0f : 0.0f), 0.0f);
    const v4 up = f4_v(0.0f, (face == 2 || face == 3) ? 0.0f : 1.0f, (face == 2 || face == 3) ? -1.0f : 0.0f, 0.0f);
    return proj_dir(right, up, fwd, f2_v(1.0f, 1.0f), uv);
}

/* AmbCube_Fit, math/ambcube.h:45-67 (AmbCube_t math/types.h:513-517) */
struct AmbCube { v4 Values[6]; };
PT_FN void AmbCube_Fit(AmbCube& c, float weight, v4 dir, v4 rad)
{
    const bool fx = dir.x > 0.0f, fy = dir.y > 0.0f, fz = dir.z > 0.0f;
    dir = f4_mul(dir, dir);
    dir = f4_muls(dir, weight);
    if (fx) c.Values[0] = f4_lerps(c.Values[0], rad, dir.x); else c.Values[1] = f4_lerps(c.Values[1], rad, dir.x);
    if (fy) c.Values[2] = f4_lerps(c.Values[2], rad, dir.y); else c.Values[3] = f4_lerps(c.Values[3], rad, dir.y);
    if (fz) c.Values[4] = f4_lerps(c.Values[4], rad, dir.z); else c.Values[5] = f4_lerps(c.Values[5], rad, dir.z);
}

/* ---- lightmap embedding helpers: math/sdf.h:99-156, rendering/lightmap.c:1000-1019 ---- */
PT_FN float f2_dot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }                                  /* float2_funcs.h:303 */
PT_FN v2 f2_min(v2 a, v2 b) { return f2_v(f1_min(a.x, b.x), f1_min(a.y, b.y)); }                  /* :253 */
PT_FN float sdEdge2D(v2 A, v2 B, v2 P)                                                            /* sdf.h:99-108 */
{
    v2 AP = f2_sub(P, A), AB = f2_sub(B, A);
    float e = AP.y * AB.x - AP.x * AB.y;
    return 0.5f * e;
}
PT_FN v4 bary2D(v2 a, v2 b, v2 c, float rcpArea, v2 p)                                            /* :112-123 */
{
    float w = sdEdge2D(b, c, p) * rcpArea;
    float u = sdEdge2D(c, a, p) * rcpArea;
    float v = sdEdge2D(a, b, p) * rcpArea;
    return f4_v(w, u, v, 0.0f);
}
PT_FN float sdTriangle2D(v2 a, v2 b, v2 c, v2 p)                                                  /* :125-156 */
{
    v2 ba = f2_sub(b, a), cb = f2_sub(c, b), ac = f2_sub(a, c);
    v2 pa = f2_sub(p, a), pb = f2_sub(p, b), pc = f2_sub(p, c);
    float h0 = f1_sat(f2_dot(pa, ba) / f2_dot(ba, ba));
    float h1 = f1_sat(f2_dot(pb, cb) / f2_dot(cb, cb));
    float h2 = f1_sat(f2_dot(pc, ac) / f2_dot(ac, ac));
    v2 pq0 = f2_sub(pa, f2_muls(ba, h0));
    v2 pq1 = f2_sub(pb, f2_muls(cb, h1));
    v2 pq2 = f2_sub(pc, f2_muls(ac, h2));
    float s = f1_sign(ba.x * ac.y - ba.y * ac.x);
    v2 t0 = f2_v(f2_dot(pq0, pq0), s * (pa.x * ba.y - pa.y * ba.x));
    v2 t1 = f2_v(f2_dot(pq1, pq1), s * (pb.x * cb.y - pb.y * cb.x));
    v2 t2 = f2_v(f2_dot(pq2, pq2), s * (pc.x * ac.y - pc.y * ac.x));
    v2 d = f2_min(t0, f2_min(t1, t2));
    float dySign = (d.y >= 0.0f) ? 1.0f : -1.0f;
    return -pt_sqrt(f1_max(d.x, PT_EPS_SQ)) * dySign;
}
#define PT_LM_FILL_PADDING 2.0f                                                                   /* lightmap.c:40 */
/* the fragment EmbedTaskFn keeps for a texel once triangle (A, B, C) (texel units) has won it, lightmap.c:1007-1014 */
PT_FN void LmEmbedFragment(v2 A, v2 B, v2 C, v2 px, const v4* pos, const v4* nor, v4* fragPos, v4* fragNor)
{
    float area = sdEdge2D(A, B, C);
    area = f1_max(area, 1e-5f);
    v4 wuv = bary2D(A, B, C, 1.0f / area, px);
    wuv = f4_divs(wuv, wuv.x + wuv.y + wuv.z);
    *fragPos = f4_blend(pos[0], pos[1], pos[2], wuv);
    *fragNor = f4_normalize3(f4_blend(nor[0], nor[1], nor[2], wuv));
}

/* ---- geometry helpers: math/area.h:28, math/sdf.h:158-189 ---- */
PT_FN float TriArea3D(v4 A, v4 B, v4 C) { return 0.5f * f4_length3(f4_cross3(f4_sub(B, A), f4_sub(C, A))); }
/* ---- lightmap chart nodes: math/sdf.h:251-257, rendering/lightmap.c:284-320 ---- */
PT_FN v4 triToPlane(v4 A, v4 B, v4 C)                                                             /* sdf.h:251-257 */
{
    v4 pl = f4_normalize3(f4_cross3(f4_sub(B, A), f4_sub(C, A)));
    pl.w = f4_dot3(pl, A);
    return pl;
}
struct ChartGeom { v4 plane; v2 a, b, c; float area; };
/* chartnode_new (lightmap.c:302-320): plane, the triangle projected on its own plane's tangent frame (ProjTri :291-300)
 * in texels, area */
PT_FN ChartGeom ChartNodeGeom(v4 A, v4 B, v4 C, float texelsPerUnit)
{
    ChartGeom g;
    g.plane = triToPlane(A, B, C);
    const m33 TBN = NormalToTBN(triToPlane(A, B, C));
    g.a = f2_muls(f2_v(f4_dot3(TBN.c0, A), f4_dot3(TBN.c1, A)), texelsPerUnit);
    g.b = f2_muls(f2_v(f4_dot3(TBN.c0, B), f4_dot3(TBN.c1, B)), texelsPerUnit);
    g.c = f2_muls(f2_v(f4_dot3(TBN.c0, C), f4_dot3(TBN.c1, C)), texelsPerUnit);
    g.area = TriArea3D(A, B, C);
    return g;
}
#define PT_LM_MASK_PADDING 1.0f                                                                   /* lightmap.c:39 */

PT_FN float sdTriangle3D(v4 a, v4 b, v4 c, v4 p)
{
    v4 ba = f4_sub(b, a); v4 pa = f4_sub(p, a);
    v4 cb = f4_sub(c, b); v4 pb = f4_sub(p, b);
    v4 ac = f4_sub(a, c); v4 pc = f4_sub(p, c);
    ba.w = pa.w = cb.w = pb.w = ac.w = pc.w = 0.0f;
    v4 nor = f4_cross3(ba, ac);
    float t0 = f1_sign(f4_dot3(f4_cross3(ba, nor), pa));
    float t1 = f1_sign(f4_dot3(f4_cross3(cb, nor), pb));
    float t2 = f1_sign(f4_dot3(f4_cross3(ac, nor), pc));
    float t;
    if ((t0 + t1 + t2) < 2.0f)
    {
        t0 = f1_sat(f4_dot3(ba, pa) / f4_lengthsq3(ba));
        t1 = f1_sat(f4_dot3(cb, pb) / f4_lengthsq3(cb));
        t2 = f1_sat(f4_dot3(ac, pc) / f4_lengthsq3(ac));
        t0 = f4_lengthsq3(f4_sub(f4_muls(ba, t0), pa));
        t1 = f4_lengthsq3(f4_sub(f4_muls(cb, t1), pb));
        t2 = f4_lengthsq3(f4_sub(f4_muls(ac, t2), pc));
        t = f1_min(t0, f1_min(t1, t2));
    }
    else
    {
        t0 = f4_dot3(nor, pa);
        t = t0 * t0 / f4_lengthsq3(nor);
    }
    return pt_sqrt(f1_max(t, PT_EPS_SQ));
}

/* ---- participating media helpers: math/atmosphere.h:36-43, math/noise.h:16-83 ---- */
PT_FN float MiePhase(float cosTheta, float g)
{
    float k = (3.0f / (8.0f * PT_PI)) * (1.0f - g * g) / (2.0f + g * g);
    float l = 1.0f + g * g - 2.0f * g * cosTheta;
    l = l * pt_sqrt(f1_max(PT_EPS_SQ, l));
    return k * (1.0f + cosTheta * cosTheta) / f1_max(PT_EPS, l);
}
PT_FN v4 GradientCell3(int32_t cx, int32_t cy, int32_t cz, uint32_t seed)
{
    uint32_t i = pcg4((uint32_t)cx, (uint32_t)cy, (uint32_t)cz, seed).x;
    return f4_v((i & (1u << 31)) ? 1.0f : -1.0f, (i & (1u << 30)) ? 1.0f : -1.0f, (i & (1u << 29)) ? 1.0f : -1.0f, 0.0f);
}
PT_FN float GradientNoise3(v4 p, uint32_t seed)
{
    v4 fl = f4_floor(p);
    int32_t ix = (int32_t)fl.x, iy = (int32_t)fl.y, iz = (int32_t)fl.z;
    v4 f = f4_frac(p);
    float c000 = f4_dot3(GradientCell3(ix + 0, iy + 0, iz + 0, seed), f4_sub(f, f4_v(0, 0, 0, 0)));
    float c001 = f4_dot3(GradientCell3(ix + 0, iy + 0, iz + 1, seed), f4_sub(f, f4_v(0, 0, 1, 0)));
    float c010 = f4_dot3(GradientCell3(ix + 0, iy + 1, iz + 0, seed), f4_sub(f, f4_v(0, 1, 0, 0)));
    float c011 = f4_dot3(GradientCell3(ix + 0, iy + 1, iz + 1, seed), f4_sub(f, f4_v(0, 1, 1, 0)));
    float c100 = f4_dot3(GradientCell3(ix + 1, iy + 0, iz + 0, seed), f4_sub(f, f4_v(1, 0, 0, 0)));
    float c101 = f4_dot3(GradientCell3(ix + 1, iy + 0, iz + 1, seed), f4_sub(f, f4_v(1, 0, 1, 0)));
    float c110 = f4_dot3(GradientCell3(ix + 1, iy + 1, iz + 0, seed), f4_sub(f, f4_v(1, 1, 0, 0)));
    float c111 = f4_dot3(GradientCell3(ix + 1, iy + 1, iz + 1, seed), f4_sub(f, f4_v(1, 1, 1, 0)));
    v4 u = f4_unormstep(f);
    v4 c0vz = f4_lerps(f4_v(c000, c010, c100, c110), f4_v(c001, c011, c101, c111), u.z);
    float c0yz = f1_lerp(c0vz.x, c0vz.y, u.y);
    float c1yz = f1_lerp(c0vz.z, c0vz.w, u.y);
    return f1_lerp(c0yz, c1yz, u.x);
}
PT_FN float FbmGradientNoise3(v4 p, float lacunarity, float gain, int32_t octaves, uint32_t seed)
{
    float sum = 0.0f, freq = 1.0f, ampl = 1.0f;
    for (int32_t i = 0; i < octaves; ++i)
    {
        float noise = GradientNoise3(f4_muls(p, freq), seed + (uint32_t)i + 1u);
        sum += noise * ampl;
        freq *= lacunarity;
        ampl *= gain;
    }
    return sum;
}

/* ---- spherical gaussians: math/sphgauss.h:21-35, sphgauss.c:19-58 ---- */
PT_FN float SG_BasisEval(v4 axis, v4 dir) { return pl_expf(axis.w * (f4_dot3(dir, axis) - 1.0f)); }
PT_FN void SG_Accumulate(float sampleWeight, v4 dir, v4 rad, const v4* axii, v4* amplitudes, int32_t length)
{
    if (sampleWeight == 1.0f)
    {
        for (int32_t i = 0; i < length; ++i) amplitudes[i] = f4_s(0.0f);
    }
    v4 estimate = f4_s(0.0f);
    for (int32_t i = 0; i < length; ++i)
    {
        estimate = f4_add(estimate, f4_muls(amplitudes[i], SG_BasisEval(axii[i], dir)));
    }
    for (int32_t i = 0; i < length; ++i)
    {
        v4 axis = axii[i];
        float basis = SG_BasisEval(axis, dir);
        if (basis > 0.0f)
        {
            v4 amplitude = amplitudes[i];
            float weight = amplitude.w;
            weight = f1_lerp(weight, basis, sampleWeight);
            v4 otherLobes = f4_sub(estimate, f4_muls(amplitude, basis));
            v4 thisLobe = f4_muls(f4_sub(rad, otherLobes), basis / weight);
            amplitude = f4_lerps(amplitude, thisLobe, sampleWeight);
            amplitude = f4_max(amplitude, f4_s(0.0f));
            amplitude.w = weight;
            amplitudes[i] = amplitude;
        }
    }
}

} /* namespace pt */

#endif /* PT_MATH_H */
