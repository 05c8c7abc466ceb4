/*
 * orc_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Scalar C restatement of the subset of Pim's header-only math that the path
 * tracer hot path uses.  Every function names the reference file:line it
 * follows (paths relative to /root/reference/src).  Arithmetic ORDER follows the
 * reference so that, built with -ffp-contract=off and without -ffast-math, the
 * results are bit-identical to the reference headers built the same way
 * (tests/test_oracle_golden.py pins that against tests/golden/ref_math.bin,
 * which oracle/ref_golden.c generates by #including the reference headers).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <stdint.h>
#include <stdbool.h>
#include "orc_libm.h"

typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { v4 c0, c1, c2; } m33;
typedef struct { v4 c0, c1, c2, c3; } m44;
typedef struct { uint32_t x, y, z, w; } u4;

/* math/scalar.h:21-28 */
#define ORC_PI          3.1415926535897932384626433832795f
#define ORC_TAU         6.283185307179586476925286766559f
#define ORC_RCP_EPS     8388608.0f
#define ORC_EPS         (1.0f / ((1u << 23) - 1u))
#define ORC_EPS_SQ      (1.0f / ((1ull << 46) - 1ull))
#define ORC_LOG2_EPS    (-23.0f)
#define ORC_SQRT5_CONJ  0.2360679774997896964091736687313f
#define ORC_MILLI       1e-3f
#define ORC_KILO        1e3f
#define ORC_DECI        1e-1f
/* math/lighting.h:36-39 */
#define ORC_MIN_ALPHA   (1.0f / (1 << 10))
/* rendering/r_config.h:113 */
#define ORC_EMISSION_SCALE 100.0f

#define ORC_INL static inline

/* ---- scalar: math/scalar.h:65-302 ---- */
ORC_INL float o_sat(float x) { x = (x > 0.0f) ? x : 0.0f; x = (x < 1.0f) ? x : 1.0f; return x; }   /* :66-72 NaN -> 0 */
ORC_INL float o_min(float a, float b) { return a < b ? a : b; }                                   /* :104 */
ORC_INL float o_max(float a, float b) { return a > b ? a : b; }                                   /* :109 */
ORC_INL float o_clamp(float x, float lo, float hi) { return o_min(o_max(x, lo), hi); }            /* :114 */
ORC_INL float o_abs(float x) { return o_max(x, -x); }                                             /* :124 */
ORC_INL float o_sign(float x) { return ((x > 0.0f) ? 1.0f : 0.0f) - ((x < 0.0f) ? 1.0f : 0.0f); } /* :119 */
ORC_INL float o_trunc(float x) { int32_t i = (int32_t)x; return (float)i; }                       /* :129-137 */
ORC_INL float o_floor(float x) { float i = o_trunc(x); return i - (i > x); }                      /* :139-147 */
ORC_INL float o_ceil(float x) { float i = o_trunc(x); return i + (x > i); }                       /* :149-157 */
ORC_INL float o_frac(float x) { return x - o_floor(x); }                                          /* :159 */
ORC_INL float o_lerp(float a, float b, float t) { return a + (b - a) * t; }                       /* :183 */
ORC_INL float o_distance(float a, float b) { return o_abs(b - a); }                               /* :267 */
ORC_INL float o_radians(float x) { return x * (ORC_TAU / 360.0f); }                               /* :74 */
ORC_INL float o_gauss_invcdf(float p, float u, float s) { return u + s * sqrtf(-o_logf(o_max(1.0f - p, ORC_EPS))); } /* :299-302 */
ORC_INL int32_t oi_min(int32_t a, int32_t b) { return a < b ? a : b; }
ORC_INL int32_t oi_max(int32_t a, int32_t b) { return a > b ? a : b; }
ORC_INL int32_t oi_clamp(int32_t x, int32_t lo, int32_t hi) { return oi_min(hi, oi_max(lo, x)); } /* :364 */

/* ---- float2: math/float2_funcs.h ---- */
ORC_INL v2 o2(float x, float y) { v2 v = { x, y }; return v; }
ORC_INL v2 o2_add(v2 a, v2 b) { return o2(a.x + b.x, a.y + b.y); }
ORC_INL v2 o2_sub(v2 a, v2 b) { return o2(a.x - b.x, a.y - b.y); }
ORC_INL v2 o2_mul(v2 a, v2 b) { return o2(a.x * b.x, a.y * b.y); }
ORC_INL v2 o2_muls(v2 a, float s) { return o2(a.x * s, a.y * s); }
ORC_INL v2 o2_lerps(v2 a, v2 b, float t) { return o2_add(a, o2_muls(o2_sub(b, a), t)); }          /* :337 */
ORC_INL v2 o2_bilerp(v2 tl, v2 tr, v2 bl, v2 br, v2 x) { return o2_lerps(o2_lerps(tl, tr, x.x), o2_lerps(bl, br, x.x), x.y); } /* :359 */
ORC_INL v2 o2_blend(v2 a, v2 b, v2 c, v4 wuvt)                                                    /* :607-613 */
{
    v2 p = o2_muls(a, wuvt.x);
    p = o2_add(p, o2_muls(b, wuvt.y));
    p = o2_add(p, o2_muls(c, wuvt.z));
    return p;
}

/* ---- float4: math/float4_funcs.h ---- */
ORC_INL v4 o4(float x, float y, float z, float w) { v4 v = { x, y, z, w }; return v; }
ORC_INL v4 o4_s(float s) { return o4(s, s, s, s); }
ORC_INL v4 o4_add(v4 a, v4 b) { return o4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }          /* :133 */
ORC_INL v4 o4_adds(v4 a, float s) { return o4(a.x + s, a.y + s, a.z + s, a.w + s); }              /* :139 */
ORC_INL v4 o4_sub(v4 a, v4 b) { return o4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }          /* :151 */
ORC_INL v4 o4_mul(v4 a, v4 b) { return o4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }          /* :169 */
ORC_INL v4 o4_muls(v4 a, float s) { return o4(a.x * s, a.y * s, a.z * s, a.w * s); }              /* :175 */
ORC_INL v4 o4_divs(v4 a, float s) { return o4_muls(a, 1.0f / s); }                                /* :187 multiply by reciprocal */
ORC_INL v4 o4_neg(v4 a) { return o4(-a.x, -a.y, -a.z, -a.w); }                                    /* :121 */
ORC_INL v4 o4_rcp(v4 a) { return o4(1.0f / a.x, 1.0f / a.y, 1.0f / a.z, 1.0f / a.w); }            /* :127 */
ORC_INL v4 o4_inv(v4 a) { return o4_sub(o4_s(1.0f), a); }                                         /* :198 */
ORC_INL v4 o4_zxy(v4 v) { return o4(v.z, v.x, v.y, 0.0f); }                                       /* :109 */
ORC_INL float o4_sum3(v4 v) { return v.x + v.y + v.z; }                                           /* :400 */
ORC_INL float o4_dot3(v4 a, v4 b) { return o4_sum3(o4_mul(a, b)); }                               /* :546 */
ORC_INL float o4_dotsat(v4 a, v4 b) { return o_sat(o4_dot3(a, b)); }                              /* :551 */
ORC_INL v4 o4_cross3(v4 a, v4 b) { return o4_zxy(o4_sub(o4_mul(o4_zxy(a), b), o4_mul(a, o4_zxy(b)))); } /* :556 */
ORC_INL float o4_length3(v4 x) { return sqrtf(o_max(ORC_EPS_SQ, o4_dot3(x, x))); }                /* :568 */
ORC_INL v4 o4_normalize3(v4 x) { return o4_divs(x, o4_length3(x)); }                              /* :577 */
ORC_INL float o4_lengthsq3(v4 x) { return o4_dot3(x, x); }                                        /* :595 */
ORC_INL v4 o4_min(v4 a, v4 b) { return o4(o_min(a.x, b.x), o_min(a.y, b.y), o_min(a.z, b.z), o_min(a.w, b.w)); }
ORC_INL v4 o4_max(v4 a, v4 b) { return o4(o_max(a.x, b.x), o_max(a.y, b.y), o_max(a.z, b.z), o_max(a.w, b.w)); }
ORC_INL float o4_hmax3(v4 v) { return o_max(v.x, o_max(v.y, v.z)); }                              /* :519 */
ORC_INL float o4_hmin3(v4 v) { return o_min(v.x, o_min(v.y, v.z)); }                              /* :510 */
ORC_INL v4 o4_abs(v4 v) { return o4(o_abs(v.x), o_abs(v.y), o_abs(v.z), o_abs(v.w)); }            /* :727 */
ORC_INL v4 o4_lerp(v4 a, v4 b, v4 t) { return o4_add(a, o4_mul(o4_sub(b, a), t)); }               /* :609 */
ORC_INL v4 o4_lerps(v4 a, v4 b, float t) { return o4_add(a, o4_muls(o4_sub(b, a), t)); }          /* :613 f4_lerpvs */
ORC_INL v4 o4_lerpsv(float a, float b, v4 t) { v4 m = o4_muls(t, b - a); return o4(a + m.x, a + m.y, a + m.z, a + m.w); } /* :617 */
ORC_INL v4 o4_bilerp(v4 tl, v4 tr, v4 bl, v4 br, v2 x) { return o4_lerps(o4_lerps(tl, tr, x.x), o4_lerps(bl, br, x.x), x.y); } /* :636 */
ORC_INL v4 o4_blend(v4 a, v4 b, v4 c, v4 wuvt)                                                    /* :995-1001 */
{
    v4 p = o4_muls(a, wuvt.x);
    p = o4_add(p, o4_muls(b, wuvt.y));
    p = o4_add(p, o4_muls(c, wuvt.z));
    return p;
}
ORC_INL v4 o4_reflect3(v4 i, v4 n)                                                                /* :707-711 */
{
    v4 nidn = o4_muls(n, o4_dot3(i, n));
    return o4_sub(i, o4_muls(nidn, 2.0f));
}
ORC_INL v4 o4_refract3(v4 i, v4 n, float k)                                                       /* :713-719 */
{
    float cosTheta = o_min(1.0f, o4_dot3(o4_neg(i), n));
    v4 perp = o4_muls(o4_add(o4_muls(n, cosTheta), i), k);
    v4 para = o4_muls(n, -sqrtf(o_abs(1.0f - o4_lengthsq3(perp))));
    return o4_normalize3(o4_add(perp, para));
}
ORC_INL v4 o4_floor(v4 v) { return o4(o_floor(v.x), o_floor(v.y), o_floor(v.z), o_floor(v.w)); }  /* :901 */
ORC_INL v4 o4_ceil(v4 v) { return o4(o_ceil(v.x), o_ceil(v.y), o_ceil(v.z), o_ceil(v.w)); }       /* :913 */
ORC_INL v4 o4_frac(v4 v) { return o4_sub(v, o4_floor(v)); }                                       /* :925 */
ORC_INL v4 o4_unormstep(v4 t) { return o4_mul(o4_mul(o4_adds(o4_muls(t, -2.0f), 3.0f), t), t); }  /* :672 */
ORC_INL float o4_avglum(v4 x) { return (x.x + x.y + x.z) * (1.0f / 3.0f); }                       /* math/color.h:261 */

ORC_INL v3 o3(float x, float y, float z) { v3 v = { x, y, z }; return v; }
ORC_INL v3 o3_lerps(v3 a, v3 b, float t)                                                          /* math/float3_funcs.h:373 */
{
    return o3(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t);
}
ORC_INL v3 o3_bilerp(v3 tl, v3 tr, v3 bl, v3 br, v2 x) { return o3_lerps(o3_lerps(tl, tr, x.x), o3_lerps(bl, br, x.x), x.y); } /* :395 */

/* ---- matrices: math/float3x3_funcs.h, float4x4_funcs.h, quat_funcs.h ---- */
ORC_INL v4 m33_mul_col(m33 m, v4 c)                                                               /* float3x3_funcs.h:150-157 */
{
    return o4_add(o4_muls(m.c0, c.x), o4_add(o4_muls(m.c1, c.y), o4_muls(m.c2, c.z)));
}
ORC_INL m33 m33_transpose(m33 a)                                                                  /* :54-69 */
{
    m33 b;
    b.c0 = o4(a.c0.x, a.c1.x, a.c2.x, 0.0f);
    b.c1 = o4(a.c0.y, a.c1.y, a.c2.y, 0.0f);
    b.c2 = o4(a.c0.z, a.c1.z, a.c2.z, 0.0f);
    return b;
}
ORC_INL m33 m33_inverse(m33 m)                                                                    /* :71-105, [column][row] accessors */
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
    inv.c0 = o4(r * i, -r * f, r * e, 0.0f);
    inv.c1 = o4(-r * h, r * d, -r * b, 0.0f);
    inv.c2 = o4(r * g, -r * c, r * a, 0.0f);
    return inv;
}
ORC_INL m33 m44_IM(m44 M)                                                                         /* f3x3_IM :107-110 */
{
    m33 t; t.c0 = M.c0; t.c1 = M.c1; t.c2 = M.c2;
    t.c0.w = t.c1.w = t.c2.w = 0.0f;
    return m33_inverse(m33_transpose(t));
}
ORC_INL v4 m44_mul_dir(m44 m, v4 d)                                                               /* float4x4_funcs.h:78-84 */
{
    v4 a = o4_muls(m.c0, d.x), b = o4_muls(m.c1, d.y), c = o4_muls(m.c2, d.z);
    return o4_add(a, o4_add(b, c));
}
ORC_INL v4 m44_mul_pt(m44 m, v4 p)                                                                /* :87-93 */
{
    v4 a = o4_muls(m.c0, p.x), b = o4_muls(m.c1, p.y), c = o4_muls(m.c2, p.z);
    return o4_add(o4_add(a, b), o4_add(c, m.c3));
}
ORC_INL m33 quat_to_m33(v4 v)                                                                     /* quat_f3x3, quat_funcs.h:131-158 */
{
    float xx = v.x * v.x, xy = v.x * v.y, xz = v.x * v.z;
    float yy = v.y * v.y, yz = v.y * v.z, zz = v.z * v.z;
    float wx = v.w * v.x, wy = v.w * v.y, wz = v.w * v.z;
    m33 m;
    m.c0 = o4(1.0f - 2.0f * (yy + zz), 2.0f * (xy - wz), 2.0f * (xz + wy), 0.0f);
    m.c1 = o4(2.0f * (xy + wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz - wx), 0.0f);
    m.c2 = o4(2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (xx + yy), 0.0f);
    return m;
}
ORC_INL m44 m44_trs(v4 t, v4 q, v4 s)                                                             /* f4x4_trs, float4x4_funcs.h:128-192 */
{
    m44 T;
    T.c0 = o4(1, 0, 0, 0); T.c1 = o4(0, 1, 0, 0); T.c2 = o4(0, 0, 1, 0); T.c3 = o4(t.x, t.y, t.z, 1.0f);
    m33 rot = quat_to_m33(q);
    m44 R;
    R.c0 = m44_mul_dir(T, rot.c0); R.c1 = m44_mul_dir(T, rot.c1); R.c2 = m44_mul_dir(T, rot.c2); R.c3 = T.c3;
    m44 S;
    S.c0 = o4_muls(R.c0, s.x); S.c1 = o4_muls(R.c1, s.y); S.c2 = o4_muls(R.c2, s.z); S.c3 = R.c3;
    return S;
}
ORC_INL v4 quat_mul_dir(v4 q, v4 dir)                                                             /* quat_funcs.h:86-97 */
{
    v4 uv = o4_cross3(q, dir);
    v4 uuv = o4_cross3(q, uv);
    v4 y = o4_muls(uv, q.w);
    y = o4_add(y, uuv);
    y = o4_muls(y, 2.0f);
    y = o4_add(y, dir);
    return y;
}
ORC_INL v4 quat_fwd(v4 q) { return quat_mul_dir(q, o4(0.0f, 0.0f, -1.0f, 0.0f)); }                /* :99 */
ORC_INL v4 quat_up(v4 q) { return quat_mul_dir(q, o4(0.0f, 1.0f, 0.0f, 0.0f)); }                  /* :105 */
ORC_INL v4 quat_right(v4 q) { return quat_mul_dir(q, o4(1.0f, 0.0f, 0.0f, 0.0f)); }               /* :111 */
ORC_INL v4 m33_to_quat(m33 m)                                                                     /* f3x3_quat :160-221 */
{
    float m00 = m.c0.x, m01 = m.c0.y, m02 = m.c0.z;
    float m10 = m.c1.x, m11 = m.c1.y, m12 = m.c1.z;
    float m20 = m.c2.x, m21 = m.c2.y, m22 = m.c2.z;
    float x = m00 - m11 - m22, y = m11 - m00 - m22, z = m22 - m00 - m11, w = m00 + m11 + m22;
    float b = o_max(o_max(x, y), o_max(z, w));
    int i = 0;
    i = (b == x) ? 1 : i;
    i = (b == y) ? 2 : i;
    i = (b == z) ? 3 : i;
    b = sqrtf(o_max(b + 1.0f, 4.0f * ORC_EPS_SQ)) * 0.5f;
    float mult = 0.25f / b;
    v4 r = o4(0, 0, 0, 1);
    switch (i)
    {
    case 0: r.w = b; r.x = (m12 - m21) * mult; r.y = (m20 - m02) * mult; r.z = (m01 - m10) * mult; break;
    case 1: r.w = (m12 - m21) * mult; r.x = b; r.y = (m01 + m10) * mult; r.z = (m20 + m02) * mult; break;
    case 2: r.w = (m20 - m02) * mult; r.x = (m01 + m10) * mult; r.y = b; r.z = (m12 + m21) * mult; break;
    default: r.w = (m01 - m10) * mult; r.x = (m20 + m02) * mult; r.y = (m12 + m21) * mult; r.z = b; break;
    }
    return r;
}
ORC_INL v4 quat_lookat(v4 forward, v4 up)                                                         /* :223-233 */
{
    forward = o4_neg(forward);
    v4 right = o4_normalize3(o4_cross3(up, forward));
    up = o4_normalize3(o4_cross3(forward, right));
    m33 m; m.c2 = forward; m.c0 = right; m.c1 = up;
    return m33_to_quat(m);
}

/* ---- frustum: math/frustum.h:26-47 ---- */
ORC_INL v2 proj_slope(float fovy, float aspect) { float t = o_tanf(fovy * 0.5f); return o2(aspect * t, t); }
ORC_INL v4 proj_dir(v4 right, v4 up, v4 fwd, v2 slope, v2 coord)
{
    coord = o2_mul(coord, slope);
    right = o4_muls(right, coord.x);
    up = o4_muls(up, coord.y);
    fwd = o4_add(fwd, right);
    fwd = o4_add(fwd, up);
    return o4_normalize3(fwd);
}

/* ---- RNG: math/pcg.h:16-171, common/random.h:13-146 ---- */
ORC_INL uint32_t pcg1(uint32_t v)
{
    v = v * 747796405u + 2891336453u;
    v = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    v = (v >> 22) ^ v;
    return v;
}
ORC_INL void pcg2(uint32_t* px, uint32_t* py)
{
    uint32_t x = *px, y = *py;
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    x += y * 1664525u; y += x * 1664525u;
    x ^= x >> 16; y ^= y >> 16;
    x += y * 1664525u; y += x * 1664525u;
    *px = x; *py = y;
}
ORC_INL void pcg3(uint32_t* px, uint32_t* py, uint32_t* pz)
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
ORC_INL u4 pcg4(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
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
ORC_INL float rng_tofloat(uint32_t x) { return (x >> 8) * (1.0f / (1 << 24)); }                   /* random.h:108 */
ORC_INL uint32_t rng_u32(u4* s) { s->x = pcg1(s->x); return s->x; }                               /* :37 */
ORC_INL float rng_f32(u4* s) { s->x = pcg1(s->x); return rng_tofloat(s->x); }                     /* :112 */
ORC_INL v2 rng_f2(u4* s) { pcg2(&s->x, &s->y); return o2(rng_tofloat(s->x), rng_tofloat(s->y)); } /* :117 */
ORC_INL v4 rng_f4(u4* s)                                                                          /* :136 */
{
    *s = pcg4(s->x, s->y, s->z, s->w);
    return o4(rng_tofloat(s->x), rng_tofloat(s->y), rng_tofloat(s->z), rng_tofloat(s->w));
}

/* ---- sampling: math/sampling.h ---- */
ORC_INL m33 NormalToTBN(v4 N)                                                                     /* :26-60 */
{
    float s = (N.z < 0.0f) ? -1.0f : 1.0f;
    const float a = -1.0f / (s + N.z);
    const float b = N.x * N.y * a;
    m33 t;
    t.c0 = o4(1.0f + s * N.x * N.x * a, s * b, -s * N.x, 0.0f);
    t.c1 = o4(b, s + N.y * N.y * a, -N.y, 0.0f);
    t.c2 = N;
    return t;
}
ORC_INL v4 TbnToWorld(m33 TBN, v4 nTS) { return m33_mul_col(TBN, nTS); }                          /* :62 */
ORC_INL v4 TanToWorld(v4 nWS, v4 nTS) { return TbnToWorld(NormalToTBN(nWS), nTS); }               /* :69 */
ORC_INL float RadicalInverseBase2(uint32_t bits)                                                  /* :75-83 */
{
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)(bits) * 2.3283064365386963e-10f;
}
ORC_INL v2 Hammersley2D(uint32_t i, uint32_t N) { return o2((i + 0.5f) / N, RadicalInverseBase2(i)); } /* :86 */
ORC_INL float PowerHeuristic(float f, float g) { return (f * f) / o_max(f * f + g * g, ORC_EPS); } /* :93 */
ORC_INL v2 MapSquareToDisk(v2 Xi)                                                                 /* :100-118 */
{
    const float lo = ORC_EPS, hi = 1.0f - ORC_EPS;
    Xi = o2(lo + (hi - lo) * Xi.x, lo + (hi - lo) * Xi.y);   /* f2_lerpsv = a + (b-a)*t */
    float phi, r;
    float a = 2.0f * Xi.x - 1.0f;
    float b = 2.0f * Xi.y - 1.0f;
    if ((a * a) > (b * b)) { r = a; phi = (ORC_PI / 4.0f) * (b / a); }
    else { r = b; phi = (ORC_PI / 2.0f) - (ORC_PI / 4.0f) * (a / b); }
    return o2(r * o_cosf(phi), r * o_sinf(phi));
}
ORC_INL v4 SampleBaryCoord(v2 Xi)                                                                 /* :120-128 */
{
    float r1 = sqrtf(o_max(Xi.x, ORC_EPS_SQ));
    float r2 = Xi.y;
    float u = r1 * (1.0f - r2);
    float v = r2 * r1;
    float w = 1.0f - (u + v);
    return o4(w, u, v, 0.0f);
}
ORC_INL v2 SampleNGon(v2 Xi, uint32_t side, uint32_t N, float rot)                                /* :130-139 */
{
    side %= N;
    const float R = ORC_TAU / N;
    const float a = rot + (1 + side) * R;
    const float b = rot + (2 + side) * R;
    const v2 A = { o_cosf(a), o_sinf(a) };
    const v2 B = { o_cosf(b), o_sinf(b) };
    return o2_blend(o2(0.0f, 0.0f), A, B, SampleBaryCoord(Xi));
}
ORC_INL v2 SamplePentagram(v2 Xi, uint32_t side)                                                  /* :141-156 */
{
    const float R = ORC_TAU / 5.0f;
    const float s = ORC_PI * 0.1f;
    const float q = (1.0f - ORC_SQRT5_CONJ) * 0.5f;
    side %= 5;
    const float a = s + (1.0f + side) * R;
    const float b = s + (1.5f + side) * R;
    const float c = s + (2.0f + side) * R;
    const v2 A = { q * o_cosf(a), q * o_sinf(a) };
    const v2 B = { o_cosf(b), o_sinf(b) };
    const v2 C = { q * o_cosf(c), q * o_sinf(c) };
    return o2_bilerp(A, B, o2(0.0f, 0.0f), C, Xi);
}
ORC_INL v4 SphericalToCartesian(float cosTheta, float phi)                                        /* :158-165 */
{
    float sinTheta = sqrtf(o_max(1.0f - cosTheta * cosTheta, 0.0f));
    return o4(sinTheta * o_cosf(phi), sinTheta * o_sinf(phi), cosTheta, 0.0f);
}
ORC_INL v4 SampleUnitSphere(v2 Xi) { return SphericalToCartesian(Xi.y * 2.0f - 1.0f, ORC_TAU * Xi.x); } /* :168 */
ORC_INL v4 SampleUnitHemisphere(v2 Xi) { return SphericalToCartesian(Xi.y, ORC_TAU * Xi.x); }     /* :176 */
ORC_INL v4 SampleCosineHemisphere(v2 Xi)                                                          /* :271-276 */
{
    Xi = MapSquareToDisk(Xi);
    float z = sqrtf(o_max(1.0f - (Xi.x * Xi.x + Xi.y * Xi.y), ORC_EPS_SQ));
    return o4(Xi.x, Xi.y, z, 0.0f);
}
ORC_INL v4 SampleGGXMicrofacet(v2 Xi, float alpha)                                                /* :280-287 */
{
    float a2 = alpha * alpha;
    float phi = ORC_TAU * Xi.x;
    float b = o_max(1.0f + (a2 - 1.0f) * Xi.y, ORC_EPS);
    float cosTheta = sqrtf(o_max((1.0f - Xi.y) / b, ORC_EPS_SQ));
    return SphericalToCartesian(cosTheta, phi);
}
ORC_INL float LambertPdf(float NoL) { return NoL * (1.0f / ORC_PI); }                             /* :301 */
ORC_INL float LightPdf(float area, float cosTheta, float distSq) { float t = cosTheta * area; return distSq / o_max(t, ORC_EPS); } /* :321 */
ORC_INL v2 SampleGaussPixelFilter(v2 Xi, float stddev)                                            /* :327-335 */
{
    float angle = Xi.x * ORC_TAU;
    float radius = o_gauss_invcdf(Xi.y, 0.0f, stddev);
    Xi.x = o_cosf(angle);
    Xi.y = o_sinf(angle);
    return o2_muls(Xi, radius);
}
ORC_INL float SampleFreePath(float Xi, float mfp) { return -o_logf(o_max(1.0f - Xi, ORC_EPS)) * mfp; } /* :340 */

/* ---- BRDF terms: math/lighting.h ---- */
ORC_INL float BrdfAlpha(float roughness) { return o_max(roughness * roughness, ORC_MIN_ALPHA); }   /* :57 */
ORC_INL v4 F_0(v4 albedo, float metallic) { return o4_lerps(o4_s(0.04f), albedo, metallic); }     /* :69 */
ORC_INL float F_90(v4 F0) { return o_sat(50.0f * o4_dot3(F0, o4_s(0.33f))); }                     /* :75 */
ORC_INL float F_Schlick1(float f0, float f90, float cosTheta)                                     /* :97 */
{
    float t = 1.0f - cosTheta;
    float t5 = t * t * t * t * t;
    return o_lerp(f0, f90, t5);
}
ORC_INL float F_Dielectric(float cosThetaI, float etaI, float etaT)                               /* :138-162 */
{
    cosThetaI = o_clamp(cosThetaI, -1.0f, 1.0f);
    if (cosThetaI < 0.0f)
    {
        cosThetaI = -cosThetaI;
        float tmp = etaI; etaI = etaT; etaT = tmp;
    }
    float sinThetaI = sqrtf(o_max(1.0f - cosThetaI * cosThetaI, ORC_EPS_SQ));
    float sinThetaT = (etaI / etaT) * sinThetaI;
    if (sinThetaT >= 1.0f)
    {
        return 1.0f;
    }
    float cosThetaT = sqrtf(o_max(1.0f - sinThetaT * sinThetaT, ORC_EPS_SQ));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return o_sat((Rparl * Rparl + Rperp * Rperp) * 0.5f);
}
ORC_INL v4 SigmaAFromReflectance(v4 albedo, float beta_n)                                         /* :193-206 */
{
    float r2 = beta_n * beta_n;
    float r3 = r2 * beta_n;
    float r4 = r3 * beta_n;
    float r5 = r4 * beta_n;
    float t = 5.969f - 0.215f * beta_n + 2.532f * r2 - 10.73f * r3 + 5.574f * r4 + 0.245f * r5;
    v4 s = o4(o_logf(albedo.x), o_logf(albedo.y), o_logf(albedo.z), o_logf(albedo.w));
    s = o4_divs(s, o_max(t, ORC_EPS));
    return o4_mul(s, s);
}
ORC_INL v4 AlbedoToTransmittance(v4 albedo, float roughness, float thickness)                     /* :208-212 */
{
    v4 s = o4_muls(SigmaAFromReflectance(albedo, roughness), -thickness);
    return o4(o_expf(s.x), o_expf(s.y), o_expf(s.z), o_expf(s.w));
}
ORC_INL float D_GTR(float NoH, float alpha)                                                       /* :218-224 */
{
    float a2 = alpha * alpha;
    float f = o_lerp(1.0f, a2, NoH * NoH);
    f = f * f * ORC_PI;
    return a2 / o_max(f, ORC_EPS);
}
ORC_INL float GGXPdf(float NoH, float HoV, float alpha)                                           /* sampling.h:311-315 */
{
    float d = D_GTR(NoH, alpha);
    return (d * NoH) / o_max(4.0f * HoV, ORC_EPS);
}
ORC_INL float V_SmithCorrelated(float NoL, float NoV, float alpha)                                /* :246-253 */
{
    float a2 = alpha * alpha;
    float v = NoL * sqrtf(o_max(a2 + (NoV - NoV * a2) * NoV, ORC_EPS_SQ));
    float l = NoV * sqrtf(o_max(a2 + (NoL - NoL * a2) * NoL, ORC_EPS_SQ));
    float t = v + l;
    return 0.5f / o_max(t, ORC_EPS);
}
ORC_INL float Fd_Burley(float NoL, float NoV, float HoV, float roughness)                         /* :266-276 */
{
    float fd90 = 0.5f + 2.0f * HoV * HoV * roughness;
    float lightScatter = F_Schlick1(1.0f, fd90, NoL);
    float viewScatter = F_Schlick1(1.0f, fd90, NoV);
    return (lightScatter * viewScatter) / ORC_PI;
}

/* one sample of BakeBrdfFn's loop body (math/lighting.c:53-80) for LUT texel (cx, cy): the sample's weighted
 * contribution (D*G*NoL*dx*Fc, D*G*NoL*dx), zero when the reflected direction is below the horizon */
ORC_INL v2 BrdfLutSample(int32_t cx, int32_t cy, float rcpWidth, float rcpHeight, float dx, v2 XiAA, float XiPhi, v2 XiH)
{
    const v2 aa = SampleGaussPixelFilter(XiAA, 0.5f);
    const float NoV = o_clamp(((float)cx + aa.x) * rcpWidth, ORC_EPS, 1.0f - ORC_EPS);
    const float alpha = o_clamp(((float)cy + aa.y) * rcpHeight, ORC_MIN_ALPHA, 1.0f);
    const v4 V = SphericalToCartesian(NoV, XiPhi * ORC_TAU);
    const v4 H = SampleGGXMicrofacet(XiH, alpha);                      /* N = (0, 0, 1) */
    const v4 L = o4_reflect3(o4_neg(V), H);
    const float NoL = L.z, NoH = H.z;
    const float HoV = o4_dot3(H, V);
    const float pdf = GGXPdf(NoH, HoV, alpha);
    v2 r = o2(0.0f, 0.0f);
    if ((NoL > ORC_EPS) && (pdf > ORC_EPS))
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
ORC_INL v4 rgba8_to_f4(uint32_t c)                                                                /* :113-117 (r = low byte) */
{
    const float s = 1.0f / 255.0f;
    return o4((c & 0xff) * s, ((c >> 8) & 0xff) * s, ((c >> 16) & 0xff) * s, ((c >> 24) & 0xff) * s);
}
ORC_INL uint32_t f4_to_rgba8(v4 v)                                                                /* :102-112 */
{
    v = o4(o_clamp(v.x, 0.0f, 1.0f), o_clamp(v.y, 0.0f, 1.0f), o_clamp(v.z, 0.0f, 1.0f), o_clamp(v.w, 0.0f, 1.0f));
    v = o4_adds(o4_muls(v, 255.0f), 0.5f);
    return ((uint32_t)v.x & 0xff) | (((uint32_t)v.y & 0xff) << 8) | (((uint32_t)v.z & 0xff) << 16) | (((uint32_t)v.w & 0xff) << 24);
}
ORC_INL v4 sRGB_EOTF_Fit(v4 V)                                                                    /* :193-196 */
{
    return o4_add(o4_add(o4_muls(V, 0.020883f), o4_muls(o4_mul(V, V), 0.656075f)), o4_muls(o4_mul(V, o4_mul(V, V)), 0.324285f));
}
ORC_INL v4 sRGB_InverseEOTF_Fit(v4 L)                                                             /* :212-218 */
{
    v4 l1 = o4(sqrtf(L.x), sqrtf(L.y), sqrtf(L.z), sqrtf(L.w));
    v4 l2 = o4(sqrtf(l1.x), sqrtf(l1.y), sqrtf(l1.z), sqrtf(l1.w));
    v4 l3 = o4(sqrtf(l2.x), sqrtf(l2.y), sqrtf(l2.z), sqrtf(l2.w));
    return o4_add(o4_add(o4_muls(l1, 0.658444f), o4_muls(l2, 0.643378f)), o4_muls(l3, -0.298148f));
}
ORC_INL uint32_t GammaEncode_rgba8(v4 lin) { return f4_to_rgba8(sRGB_InverseEOTF_Fit(lin)); }     /* :251 */
ORC_INL v4 GammaDecode_rgba8(uint32_t c) { return sRGB_EOTF_Fit(rgba8_to_f4(c)); }                /* :256 */
ORC_INL v4 Xy16ToNormalTs(uint32_t packed)                                                        /* :241-249 (short2: x low half) */
{
    int16_t sx = (int16_t)(packed & 0xffff), sy = (int16_t)(packed >> 16);
    v4 n;
    n.x = sx * (1.0f / (1 << 15));
    n.y = sy * (1.0f / (1 << 15));
    n.z = sqrtf(o_max(ORC_EPS_SQ, 1.0f - (n.x * n.x + n.y * n.y)));
    n.w = 0.0f;
    return n;
}
ORC_INL v4 UnpackEmission(v4 albedo, float e) { return o4_muls(albedo, (e * e) * ORC_EMISSION_SCALE); } /* :588 */

/* ---- bilinear addressing: rendering/sampler.h:176-229 ---- */
typedef struct { int32_t a, b, c, d; v2 frac; } orc_bilinear;
ORC_INL void LinearClamp(int32_t size, float u, int32_t* a, int32_t* b, float* t)                 /* :176-184 */
{
    float x = o_sat(u) * (size - 1);
    *a = (int32_t)x;
    *t = x - (float)*a;
    *b = oi_min(*a + 1, size - 1);
}
ORC_INL orc_bilinear BilinearClamp(int32_t sx, int32_t sy, v2 uv)                                 /* :210-222 */
{
    orc_bilinear r;
    int32_t xa, xb, ya, yb;
    LinearClamp(sx, uv.x, &xa, &xb, &r.frac.x);
    LinearClamp(sy, uv.y, &ya, &yb, &r.frac.y);
    r.a = ya * sx + xa; r.b = ya * sx + xb; r.c = yb * sx + xa; r.d = yb * sx + xb;
    return r;
}
ORC_INL orc_bilinear BilinearWrap(int32_t sx, int32_t sy, v2 uv)                                  /* :223-229 */
{
    uv.x = (uv.x >= 0.0f) ? uv.x : (1.0f - uv.x);
    uv.y = (uv.y >= 0.0f) ? uv.y : (1.0f - uv.y);
    uv = o2(o_frac(uv.x), o_frac(uv.y));
    return BilinearClamp(sx, sy, uv);
}
ORC_INL v4 UvBilinearWrap_c32(const uint32_t* tex, int32_t sx, int32_t sy, v2 uv)                 /* :375-379, :240-249 */
{
    orc_bilinear bi = BilinearWrap(sx, sy, uv);
    return o4_bilerp(GammaDecode_rgba8(tex[bi.a]), GammaDecode_rgba8(tex[bi.b]), GammaDecode_rgba8(tex[bi.c]), GammaDecode_rgba8(tex[bi.d]), bi.frac);
}
ORC_INL v4 UvBilinearWrap_xy16(const uint32_t* tex, int32_t sx, int32_t sy, v2 uv)                /* :388-392, :250-260 */
{
    orc_bilinear bi = BilinearWrap(sx, sy, uv);
    v4 r = o4_bilerp(Xy16ToNormalTs(tex[bi.a]), Xy16ToNormalTs(tex[bi.b]), Xy16ToNormalTs(tex[bi.c]), Xy16ToNormalTs(tex[bi.d]), bi.frac);
    return o4_normalize3(r);
}
ORC_INL v2 UvBilinearClamp_f2(const v2* tex, int32_t sx, int32_t sy, v2 uv)                       /* :347-351 */
{
    orc_bilinear bi = BilinearClamp(sx, sy, uv);
    return o2_bilerp(tex[bi.a], tex[bi.b], tex[bi.c], tex[bi.d], bi.frac);
}

/* BrdfLut_Sample, math/lighting.h:52-55; GGXEnergyCompensation, :294-303 (the table is g_BrdfLut there, an argument here) */
ORC_INL v2 BrdfLut_Sample(const v2* lut, int32_t sx, int32_t sy, float NoV, float alpha) { return UvBilinearClamp_f2(lut, sx, sy, o2(NoV, alpha)); }
ORC_INL v4 GGXEnergyCompensationLut(const v2* lut, int32_t sx, int32_t sy, v4 f0, float NoV, float alpha)
{
    v2 dvf_dv = BrdfLut_Sample(lut, sx, sy, NoV, alpha);
    float t = (1.0f / o_max(dvf_dv.y, ORC_EPS)) - 1.0f;
    return o4_adds(o4_muls(f0, t), 1.0f);
}
ORC_INL v3 UvBilinearClamp_f3(const v3* tex, int32_t sx, int32_t sy, v2 uv)                       /* :337-341 */
{
    orc_bilinear bi = BilinearClamp(sx, sy, uv);
    return o3_bilerp(tex[bi.a], tex[bi.b], tex[bi.c], tex[bi.d], bi.frac);
}
ORC_INL int32_t PointClamp1D(int32_t size, float u) { return (int32_t)(o_sat(u) * (size - 1) + 0.5f); } /* :105 */
ORC_INL int32_t PointWrap1D(int32_t size, float u)                                                /* :109-114 */
{
    u = (u >= 0.0f) ? u : (1.0f - u);
    u = o_frac(u);
    return PointClamp1D(size, u);
}

/* ---- cubemap addressing: rendering/cubemap.h:71-100, cubemap.c:24-42 ---- */
ORC_INL int Cubemap_CalcUv(v4 dir, v2* uvOut)
{
    static const float kUps[6][3] = { {0,1,0},{0,1,0},{0,0,-1},{0,0,-1},{0,1,0},{0,1,0} };
    static const float kRights[6][3] = { {0,0,-1},{0,0,1},{1,0,0},{-1,0,0},{1,0,0},{-1,0,0} };
    v4 absdir = o4_abs(dir);
    float v = o4_hmax3(absdir);
    float ma = 0.5f / v;
    int face;
    if (v == absdir.x) face = dir.x < 0.0f ? 1 : 0;
    else if (v == absdir.y) face = dir.y < 0.0f ? 3 : 2;
    else face = dir.z < 0.0f ? 5 : 4;
    v2 uv;
    uv.x = o4_dot3(o4(kRights[face][0], kRights[face][1], kRights[face][2], 0.0f), dir);
    uv.y = o4_dot3(o4(kUps[face][0], kUps[face][1], kUps[face][2], 0.0f), dir);
    uv = o2(uv.x * ma + 0.5f, uv.y * ma + 0.5f);
    *uvOut = uv;
    return face;
}

/* ---- the convolved mip chain of a cubemap face: rendering/sampler.h:23-51,85-88, rendering/cubemap.h:60-69,117-129,169-188 ---- */
ORC_INL int32_t i1_log2(int32_t x)                                                               /* math/scalar.h:392-422: floor(log2 x), -1 for x <= 0 */
{
    int32_t y = (x > 0) ? 0 : -1;
    if (x >= (1 << 16)) { y += 16; x >>= 16; }
    if (x >= (1 << 8)) { y += 8; x >>= 8; }
    if (x >= (1 << 4)) { y += 4; x >>= 4; }
    if (x >= (1 << 2)) { y += 2; x >>= 2; }
    if (x >= (1 << 1)) { y += 1; }
    return y;
}
ORC_INL int32_t CalcMipCount(int32_t size) { return 1 + i1_log2(size); }                            /* square faces: max(size.x, size.y) = size */
ORC_INL int32_t CalcMipSize(int32_t size, int32_t m) { int32_t y = size >> m; return y > 1 ? y : 1; }
ORC_INL int32_t CalcMipOffset(int32_t size, int32_t m)
{
    const int32_t last = CalcMipCount(size) - 1;
    m = m < 0 ? 0 : (m > last ? last : m);
    int32_t y = 0;
    for (int32_t i = 0; i < m; ++i) { const int32_t ms = CalcMipSize(size, i); y += ms * ms; }
    return y;
}
ORC_INL int32_t CubemapElemCount(int32_t size) { return CalcMipOffset(size, CalcMipCount(size)) + 1; } /* Cubemap_New, cubemap.c:105-106 */
#define ORC_CUBEMAP_MAX_MIP 6.0f                                                                 /* cubemap.h:12 */
ORC_INL float RoughnessToMip(float roughness) { return roughness * ORC_CUBEMAP_MAX_MIP; }
ORC_INL float MipToRoughness(float mip) { return mip / ORC_CUBEMAP_MAX_MIP; }
ORC_INL v3 Cubemap_ReadColor(const v3* const color[6], int32_t size, v4 dir)
{
    v2 uv; const int face = Cubemap_CalcUv(dir, &uv);
    return UvBilinearClamp_f3(color[face], size, size, uv);
}
/* returns the element the blend landed on */
ORC_INL int32_t Cubemap_BlendMip(v4* const convolved[6], int32_t size, int face, int32_t cx, int32_t cy, int32_t mip, v4 value, float weight)
{
    const int32_t at = CalcMipOffset(size, mip) + CalcMipSize(size, mip) * cy + cx;
    convolved[face][at] = o4_lerps(convolved[face][at], value, weight);
    return at;
}

/* f2_tent, math/float2_funcs.h:630-636.  As written in the reference: the y lane's upper branch reads
 * Xi.x AFTER it was mapped (2 - Xi.x, not 2 - Xi.y); restated as is, the golden vectors pin it. */
ORC_INL v2 f2_tent(v2 Xi)
{
    Xi = o2(Xi.x * 2.0f, Xi.y * 2.0f);
    Xi.x = Xi.x < 1.0f ? sqrtf(Xi.x) - 1.0f : 1.0f - sqrtf(2.0f - Xi.x);
    Xi.y = Xi.y < 1.0f ? sqrtf(Xi.y) - 1.0f : 1.0f - sqrtf(2.0f - Xi.x);
    return Xi;
}

/* Cubemap_CalcDir, rendering/cubemap.h:190-207 (BilinearInvert / LinearInvert rendering/sampler.h:192-236,
 * f2_snorm math/float2_funcs.h:620, Cubemap_kForwards / kRights / kUps rendering/cubemap.c:14-42) */
ORC_INL v4 Cubemap_CalcDir(int32_t size, int face, int32_t cx, int32_t cy, v2 Xi)
{
    static const float kFwd[6][3] = { {1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1} };
    static const float kUps[6][3] = { {0,1,0},{0,1,0},{0,0,-1},{0,0,-1},{0,1,0},{0,1,0} };
    static const float kRights[6][3] = { {0,0,-1},{0,0,1},{1,0,0},{-1,0,0},{1,0,0},{-1,0,0} };
    v2 uv = o2((float)cx / (float)(size - 1), (float)cy / (float)(size - 1));
    const float rcp = 1.0f / size;
    Xi = o2(Xi.x * rcp, Xi.y * rcp);
    uv = o2_add(uv, Xi);
    uv = o2(uv.x * 2.0f - 1.0f, uv.y * 2.0f - 1.0f);
    return proj_dir(o4(kRights[face][0], kRights[face][1], kRights[face][2], 0.0f), o4(kUps[face][0], kUps[face][1], kUps[face][2], 0.0f),
        o4(kFwd[face][0], kFwd[face][1], kFwd[face][2], 0.0f), o2(1.0f, 1.0f), uv);
}

/* AmbCube_Fit, math/ambcube.h:45-67 (AmbCube_t math/types.h:513-517) */
typedef struct { v4 Values[6]; } OrcAmbCube;
ORC_INL void AmbCube_Fit(OrcAmbCube* c, float weight, v4 dir, v4 rad)
{
    const int fx = dir.x > 0.0f, fy = dir.y > 0.0f, fz = dir.z > 0.0f;
    dir = o4_mul(dir, dir);
    dir = o4_muls(dir, weight);
    v4* vx = &c->Values[fx ? 0 : 1]; *vx = o4_lerps(*vx, rad, dir.x);
    v4* vy = &c->Values[fy ? 2 : 3]; *vy = o4_lerps(*vy, rad, dir.y);
    v4* vz = &c->Values[fz ? 4 : 5]; *vz = o4_lerps(*vz, rad, dir.z);
}

/* ---- geometry helpers: math/area.h:28, math/sdf.h:158-189 ---- */
ORC_INL float TriArea3D(v4 A, v4 B, v4 C) { return 0.5f * o4_length3(o4_cross3(o4_sub(B, A), o4_sub(C, A))); }
ORC_INL float sdTriangle3D(v4 a, v4 b, v4 c, v4 p)
{
    v4 ba = o4_sub(b, a); v4 pa = o4_sub(p, a);
    v4 cb = o4_sub(c, b); v4 pb = o4_sub(p, b);
    v4 ac = o4_sub(a, c); v4 pc = o4_sub(p, c);
    ba.w = pa.w = cb.w = pb.w = ac.w = pc.w = 0.0f;
    v4 nor = o4_cross3(ba, ac);
    float t0 = o_sign(o4_dot3(o4_cross3(ba, nor), pa));
    float t1 = o_sign(o4_dot3(o4_cross3(cb, nor), pb));
    float t2 = o_sign(o4_dot3(o4_cross3(ac, nor), pc));
    float t;
    if ((t0 + t1 + t2) < 2.0f)
    {
        t0 = o_sat(o4_dot3(ba, pa) / o4_lengthsq3(ba));
        t1 = o_sat(o4_dot3(cb, pb) / o4_lengthsq3(cb));
        t2 = o_sat(o4_dot3(ac, pc) / o4_lengthsq3(ac));
        t0 = o4_lengthsq3(o4_sub(o4_muls(ba, t0), pa));
        t1 = o4_lengthsq3(o4_sub(o4_muls(cb, t1), pb));
        t2 = o4_lengthsq3(o4_sub(o4_muls(ac, t2), pc));
        t = o_min(t0, o_min(t1, t2));
    }
    else
    {
        t0 = o4_dot3(nor, pa);
        t = t0 * t0 / o4_lengthsq3(nor);
    }
    return sqrtf(o_max(t, ORC_EPS_SQ));
}

/* ---- lightmap embedding helpers: math/sdf.h:99-156, rendering/lightmap.c:1000-1019 ---- */
ORC_INL float o2_dot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }                                /* float2_funcs.h:303 */
ORC_INL v2 o2_min(v2 a, v2 b) { return o2(o_min(a.x, b.x), o_min(a.y, b.y)); }                    /* :253 */
ORC_INL float sdEdge2D(v2 A, v2 B, v2 P)                                                          /* sdf.h:99-108 */
{
    v2 AP = o2_sub(P, A), AB = o2_sub(B, A);
    float e = AP.y * AB.x - AP.x * AB.y;
    return 0.5f * e;
}
ORC_INL v4 bary2D(v2 a, v2 b, v2 c, float rcpArea, v2 p)                                          /* :112-123 */
{
    float w = sdEdge2D(b, c, p) * rcpArea;
    float u = sdEdge2D(c, a, p) * rcpArea;
    float v = sdEdge2D(a, b, p) * rcpArea;
    return o4(w, u, v, 0.0f);
}
ORC_INL float sdTriangle2D(v2 a, v2 b, v2 c, v2 p)                                                /* :125-156 */
{
    v2 ba = o2_sub(b, a), cb = o2_sub(c, b), ac = o2_sub(a, c);
    v2 pa = o2_sub(p, a), pb = o2_sub(p, b), pc = o2_sub(p, c);
    float h0 = o_sat(o2_dot(pa, ba) / o2_dot(ba, ba));
    float h1 = o_sat(o2_dot(pb, cb) / o2_dot(cb, cb));
    float h2 = o_sat(o2_dot(pc, ac) / o2_dot(ac, ac));
    v2 pq0 = o2_sub(pa, o2_muls(ba, h0));
    v2 pq1 = o2_sub(pb, o2_muls(cb, h1));
    v2 pq2 = o2_sub(pc, o2_muls(ac, h2));
    float s = o_sign(ba.x * ac.y - ba.y * ac.x);
    v2 t0 = o2(o2_dot(pq0, pq0), s * (pa.x * ba.y - pa.y * ba.x));
    v2 t1 = o2(o2_dot(pq1, pq1), s * (pb.x * cb.y - pb.y * cb.x));
    v2 t2 = o2(o2_dot(pq2, pq2), s * (pc.x * ac.y - pc.y * ac.x));
    v2 d = o2_min(t0, o2_min(t1, t2));
    float dySign = (d.y >= 0.0f) ? 1.0f : -1.0f;
    return -sqrtf(o_max(d.x, ORC_EPS_SQ)) * dySign;
}
#define ORC_LM_FILL_PADDING 2.0f                                                                  /* lightmap.c:40 */
/* the fragment EmbedTaskFn keeps for a texel once triangle (A, B, C) (texel units) has won it, lightmap.c:1007-1014 */
ORC_INL void LmEmbedFragment(v2 A, v2 B, v2 C, v2 px, const v4* pos, const v4* nor, v4* fragPos, v4* fragNor)
{
    float area = sdEdge2D(A, B, C);
    area = o_max(area, 1e-5f);
    v4 wuv = bary2D(A, B, C, 1.0f / area, px);
    wuv = o4_divs(wuv, wuv.x + wuv.y + wuv.z);
    *fragPos = o4_blend(pos[0], pos[1], pos[2], wuv);
    *fragNor = o4_normalize3(o4_blend(nor[0], nor[1], nor[2], wuv));
}

/* ---- lightmap chart nodes: math/sdf.h:251-257, rendering/lightmap.c:284-320 ---- */
ORC_INL v4 triToPlane(v4 A, v4 B, v4 C)                                                           /* sdf.h:251-257 */
{
    v4 pl = o4_normalize3(o4_cross3(o4_sub(B, A), o4_sub(C, A)));
    pl.w = o4_dot3(pl, A);
    return pl;
}
typedef struct { v4 plane; v2 a, b, c; float area; } OrcChartGeom;
/* chartnode_new (lightmap.c:302-320): plane, the triangle projected on its own plane's tangent frame (ProjTri :291-300)
 * in texels, area */
ORC_INL OrcChartGeom ChartNodeGeom(v4 A, v4 B, v4 C, float texelsPerUnit)
{
    OrcChartGeom g;
    g.plane = triToPlane(A, B, C);
    const m33 TBN = NormalToTBN(triToPlane(A, B, C));
    g.a = o2_muls(o2(o4_dot3(TBN.c0, A), o4_dot3(TBN.c1, A)), texelsPerUnit);
    g.b = o2_muls(o2(o4_dot3(TBN.c0, B), o4_dot3(TBN.c1, B)), texelsPerUnit);
    g.c = o2_muls(o2(o4_dot3(TBN.c0, C), o4_dot3(TBN.c1, C)), texelsPerUnit);
    g.area = TriArea3D(A, B, C);
    return g;
}
#define ORC_LM_MASK_PADDING 1.0f                                                                  /* lightmap.c:39 */

/* ---- participating media helpers: math/atmosphere.h:36-43, math/noise.h:16-83 ---- */
ORC_INL float MiePhase(float cosTheta, float g)
{
    float k = (3.0f / (8.0f * ORC_PI)) * (1.0f - g * g) / (2.0f + g * g);
    float l = 1.0f + g * g - 2.0f * g * cosTheta;
    l = l * sqrtf(o_max(ORC_EPS_SQ, l));
    return k * (1.0f + cosTheta * cosTheta) / o_max(ORC_EPS, l);
}
ORC_INL v4 GradientCell3(int32_t cx, int32_t cy, int32_t cz, uint32_t seed)
{
    uint32_t i = pcg4((uint32_t)cx, (uint32_t)cy, (uint32_t)cz, seed).x;
    return o4((i & (1u << 31)) ? 1.0f : -1.0f, (i & (1u << 30)) ? 1.0f : -1.0f, (i & (1u << 29)) ? 1.0f : -1.0f, 0.0f);
}
ORC_INL float GradientNoise3(v4 p, uint32_t seed)
{
    v4 fl = o4_floor(p);
    int32_t ix = (int32_t)fl.x, iy = (int32_t)fl.y, iz = (int32_t)fl.z;
    v4 f = o4_frac(p);
    float c000 = o4_dot3(GradientCell3(ix + 0, iy + 0, iz + 0, seed), o4_sub(f, o4(0, 0, 0, 0)));
    float c001 = o4_dot3(GradientCell3(ix + 0, iy + 0, iz + 1, seed), o4_sub(f, o4(0, 0, 1, 0)));
    float c010 = o4_dot3(GradientCell3(ix + 0, iy + 1, iz + 0, seed), o4_sub(f, o4(0, 1, 0, 0)));
    float c011 = o4_dot3(GradientCell3(ix + 0, iy + 1, iz + 1, seed), o4_sub(f, o4(0, 1, 1, 0)));
    float c100 = o4_dot3(GradientCell3(ix + 1, iy + 0, iz + 0, seed), o4_sub(f, o4(1, 0, 0, 0)));
    float c101 = o4_dot3(GradientCell3(ix + 1, iy + 0, iz + 1, seed), o4_sub(f, o4(1, 0, 1, 0)));
    float c110 = o4_dot3(GradientCell3(ix + 1, iy + 1, iz + 0, seed), o4_sub(f, o4(1, 1, 0, 0)));
    float c111 = o4_dot3(GradientCell3(ix + 1, iy + 1, iz + 1, seed), o4_sub(f, o4(1, 1, 1, 0)));
    v4 u = o4_unormstep(f);
    v4 c0vz = o4_lerps(o4(c000, c010, c100, c110), o4(c001, c011, c101, c111), u.z);
    float c0yz = o_lerp(c0vz.x, c0vz.y, u.y);
    float c1yz = o_lerp(c0vz.z, c0vz.w, u.y);
    return o_lerp(c0yz, c1yz, u.x);
}
ORC_INL float FbmGradientNoise3(v4 p, float lacunarity, float gain, int32_t octaves, uint32_t seed)
{
    float sum = 0.0f, freq = 1.0f, ampl = 1.0f;
    for (int32_t i = 0; i < octaves; ++i)
    {
        float noise = GradientNoise3(o4_muls(p, freq), seed + (uint32_t)i + 1u);
        sum += noise * ampl;
        freq *= lacunarity;
        ampl *= gain;
    }
    return sum;
}

/* ---- spherical gaussians: math/sphgauss.h:21-35, sphgauss.c:19-58 ---- */
ORC_INL float SG_BasisEval(v4 axis, v4 dir) { return o_expf(axis.w * (o4_dot3(dir, axis) - 1.0f)); }
ORC_INL void SG_Accumulate(float sampleWeight, v4 dir, v4 rad, const v4* axii, v4* amplitudes, int32_t length)
{
    if (sampleWeight == 1.0f)
    {
        for (int32_t i = 0; i < length; ++i) amplitudes[i] = o4_s(0.0f);
    }
    v4 estimate = o4_s(0.0f);
    for (int32_t i = 0; i < length; ++i)
    {
        estimate = o4_add(estimate, o4_muls(amplitudes[i], SG_BasisEval(axii[i], dir)));
    }
    for (int32_t i = 0; i < length; ++i)
    {
        v4 axis = axii[i];
        float basis = SG_BasisEval(axis, dir);
        if (basis > 0.0f)
        {
            v4 amplitude = amplitudes[i];
            float weight = amplitude.w;
            weight = o_lerp(weight, basis, sampleWeight);
            v4 otherLobes = o4_sub(estimate, o4_muls(amplitude, basis));
            v4 thisLobe = o4_muls(o4_sub(rad, otherLobes), basis / weight);
            amplitude = o4_lerps(amplitude, thisLobe, sampleWeight);
            amplitude = o4_max(amplitude, o4_s(0.0f));
            amplitude.w = weight;
            amplitudes[i] = amplitude;
        }
    }
}

#endif /* ORC_MATH_H */
