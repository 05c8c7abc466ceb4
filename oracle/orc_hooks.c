/*
 * orc_hooks.c -- TEST INFRASTRUCTURE.  One uniform-signature hook per record of
 * tests/golden/ref_math.bin (written by oracle/ref_golden.c from the reference's own
 * headers): orc_hook_<record>(const uint32_t* in, uint32_t* out), words are IEEE-754
 * bit patterns or integers exactly as the generator stored them.
 * tests/test_oracle_golden.py calls every hook on every stored case.
 */
#include "orc_math.h"
#include <string.h>
#include <stdlib.h>

typedef union { float f; uint32_t u; int32_t i; } word;
static inline float F(const uint32_t* p, int i) { word w; w.u = p[i]; return w.f; }
static inline void PF(uint32_t* p, int i, float f) { word w; w.f = f; p[i] = w.u; }
static inline v4 F3(const uint32_t* p, int i, float w) { return o4(F(p, i), F(p, i + 1), F(p, i + 2), w); }
static inline void P3(uint32_t* p, int i, v4 v) { PF(p, i, v.x); PF(p, i + 1, v.y); PF(p, i + 2, v.z); }
static inline void P4(uint32_t* p, int i, v4 v) { P3(p, i, v); PF(p, i + 3, v.w); }

#define HOOK(name) void orc_hook_##name(const uint32_t* in, uint32_t* out)

HOOK(pcg1) { out[0] = pcg1(in[0]); }
HOOK(pcg2) { uint32_t x = in[0], y = in[1]; pcg2(&x, &y); out[0] = x; out[1] = y; }
HOOK(pcg3) { uint32_t x = in[0], y = in[1], z = in[2]; pcg3(&x, &y, &z); out[0] = x; out[1] = y; out[2] = z; }
HOOK(pcg4) { u4 r = pcg4(in[0], in[1], in[2], in[3]); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w; }
HOOK(prng_seq)
{
    u4 s = { in[0], in[1], in[2], in[3] };
    out[0] = rng_u32(&s);
    PF(out, 1, rng_f32(&s));
    v2 a = rng_f2(&s); PF(out, 2, a.x); PF(out, 3, a.y);
    v4 b = rng_f4(&s); P4(out, 4, b);
    out[8] = s.x; out[9] = s.y; out[10] = s.z; out[11] = s.w;
}
HOOK(NormalToTBN) { m33 t = NormalToTBN(F3(in, 0, 0.0f)); P3(out, 0, t.c0); P3(out, 3, t.c1); P3(out, 6, t.c2); }
HOOK(TanToWorld) { P3(out, 0, TanToWorld(F3(in, 0, 0.0f), F3(in, 3, 0.0f))); }
HOOK(MapSquareToDisk) { v2 r = MapSquareToDisk(o2(F(in, 0), F(in, 1))); PF(out, 0, r.x); PF(out, 1, r.y); }
HOOK(SampleCosineHemisphere) { P3(out, 0, SampleCosineHemisphere(o2(F(in, 0), F(in, 1)))); }
HOOK(SampleGGXMicrofacet) { P3(out, 0, SampleGGXMicrofacet(o2(F(in, 0), F(in, 1)), F(in, 2))); }
HOOK(SampleBaryCoord) { P3(out, 0, SampleBaryCoord(o2(F(in, 0), F(in, 1)))); }
HOOK(SampleUnitSphere) { P3(out, 0, SampleUnitSphere(o2(F(in, 0), F(in, 1)))); }
HOOK(SampleUnitHemisphere) { P3(out, 0, SampleUnitHemisphere(o2(F(in, 0), F(in, 1)))); }
HOOK(SampleNGon) { v2 r = SampleNGon(o2(F(in, 0), F(in, 1)), in[2], in[3], F(in, 4)); PF(out, 0, r.x); PF(out, 1, r.y); }
HOOK(SamplePentagram) { v2 r = SamplePentagram(o2(F(in, 0), F(in, 1)), in[2]); PF(out, 0, r.x); PF(out, 1, r.y); }
HOOK(SampleGaussPixelFilter) { v2 r = SampleGaussPixelFilter(o2(F(in, 0), F(in, 1)), F(in, 2)); PF(out, 0, r.x); PF(out, 1, r.y); }
HOOK(SampleFreePath) { PF(out, 0, SampleFreePath(F(in, 0), F(in, 1))); }
HOOK(PowerHeuristic) { PF(out, 0, PowerHeuristic(F(in, 0), F(in, 1))); }
HOOK(LightPdf) { PF(out, 0, LightPdf(F(in, 0), F(in, 1), F(in, 2))); }
HOOK(GGXPdf) { PF(out, 0, GGXPdf(F(in, 0), F(in, 1), F(in, 2))); }
HOOK(Hammersley2D) { v2 r = Hammersley2D(in[0], in[1]); PF(out, 0, r.x); PF(out, 1, r.y); }
HOOK(F_Dielectric) { PF(out, 0, F_Dielectric(F(in, 0), F(in, 1), F(in, 2))); }
HOOK(D_GTR) { PF(out, 0, D_GTR(F(in, 0), F(in, 1))); }
HOOK(V_SmithCorrelated) { PF(out, 0, V_SmithCorrelated(F(in, 0), F(in, 1), F(in, 2))); }
HOOK(Fd_Burley) { PF(out, 0, Fd_Burley(F(in, 0), F(in, 1), F(in, 2), F(in, 3))); }
HOOK(F_0_F_90) { v4 f0 = F_0(F3(in, 0, 1.0f), F(in, 3)); P3(out, 0, f0); PF(out, 3, F_90(f0)); }
HOOK(AlbedoToTransmittance) { P3(out, 0, AlbedoToTransmittance(F3(in, 0, 1.0f), F(in, 3), F(in, 4))); }
HOOK(BrdfAlpha) { PF(out, 0, BrdfAlpha(F(in, 0))); }
HOOK(reflect3) { P3(out, 0, o4_reflect3(F3(in, 0, 0.0f), F3(in, 3, 0.0f))); }
HOOK(refract3) { P3(out, 0, o4_refract3(F3(in, 0, 0.0f), F3(in, 3, 0.0f), F(in, 6))); }
HOOK(GammaDecode_rgba8) { P4(out, 0, GammaDecode_rgba8(in[0])); }
HOOK(GammaEncode_rgba8) { out[0] = GammaEncode_rgba8(o4(F(in, 0), F(in, 1), F(in, 2), F(in, 3))); }
HOOK(UnpackEmission) { P3(out, 0, UnpackEmission(F3(in, 0, 1.0f), F(in, 3))); }
HOOK(Xy16ToNormalTs) { P3(out, 0, Xy16ToNormalTs(in[0])); }
HOOK(Bilinear)
{
    int32_t sx = (int32_t)in[0], sy = (int32_t)in[1];
    v2 uv = o2(F(in, 2), F(in, 3));
    orc_bilinear w = BilinearWrap(sx, sy, uv), c = BilinearClamp(sx, sy, uv);
    out[0] = (uint32_t)w.a; out[1] = (uint32_t)w.b; out[2] = (uint32_t)w.c; out[3] = (uint32_t)w.d; PF(out, 4, w.frac.x); PF(out, 5, w.frac.y);
    out[6] = (uint32_t)c.a; out[7] = (uint32_t)c.b; out[8] = (uint32_t)c.c; out[9] = (uint32_t)c.d; PF(out, 10, c.frac.x); PF(out, 11, c.frac.y);
    out[12] = (uint32_t)PointWrap1D(sx, uv.x); out[13] = (uint32_t)PointWrap1D(sy, uv.y);
}
HOOK(UvBilinearWrap_4x4)
{
    v2 uv = o2(F(in, 16), F(in, 17));
    P4(out, 0, UvBilinearWrap_c32(in, 4, 4, uv));
    P3(out, 4, UvBilinearWrap_xy16(in, 4, 4, uv));
}
HOOK(UvBilinearClamp_4x4)
{
    float tex[48]; memcpy(tex, in, sizeof(tex));
    v2 uv = o2(F(in, 48), F(in, 49));
    v2 a = UvBilinearClamp_f2((const v2*)tex, 4, 4, uv);
    v3 b = UvBilinearClamp_f3((const v3*)tex, 4, 4, uv);
    PF(out, 0, a.x); PF(out, 1, a.y); PF(out, 2, b.x); PF(out, 3, b.y); PF(out, 4, b.z);
}
HOOK(Cubemap_CalcUv) { v2 uv; int f = Cubemap_CalcUv(F3(in, 0, 0.0f), &uv); out[0] = (uint32_t)f; PF(out, 1, uv.x); PF(out, 2, uv.y); }

HOOK(GGXEnergyCompensation)
{
    float tex[32]; memcpy(tex, in, sizeof(tex));
    v2 a = BrdfLut_Sample((const v2*)tex, 4, 4, F(in, 35), F(in, 36));
    v4 e = GGXEnergyCompensationLut((const v2*)tex, 4, 4, F3(in, 32, 0.0f), F(in, 35), F(in, 36));
    PF(out, 0, a.x); PF(out, 1, a.y); P3(out, 2, e);
}

HOOK(brdf_lut_sample)
{
    const float rcp = 1.0f / (float)(int32_t)in[2], dx = 1.0f / (float)(int32_t)in[3];
    v2 r = BrdfLutSample((int32_t)in[0], (int32_t)in[1], rcp, rcp, dx, o2(F(in, 4), F(in, 5)), F(in, 6), o2(F(in, 7), F(in, 8)));
    PF(out, 0, r.x); PF(out, 1, r.y);
}
HOOK(lm_embed_fragment)
{
    const float size = F(in, 6);
    const v2 A = o2_muls(o2(F(in, 0), F(in, 1)), size), B = o2_muls(o2(F(in, 2), F(in, 3)), size), C = o2_muls(o2(F(in, 4), F(in, 5)), size);
    const v2 px = o2(F(in, 7), F(in, 8));
    v4 pos[3] = { F3(in, 9, 1.0f), F3(in, 12, 1.0f), F3(in, 15, 1.0f) }, nor[3] = { F3(in, 18, 0.0f), F3(in, 21, 0.0f), F3(in, 24, 0.0f) };
    v4 fp, fn;
    LmEmbedFragment(A, B, C, px, pos, nor, &fp, &fn);
    fn = o4_normalize3(fn);
    PF(out, 0, sdTriangle2D(A, B, C, px)); PF(out, 1, sdEdge2D(A, B, C)); P3(out, 2, fp); P3(out, 5, fn);
}
HOOK(lm_chartnode)
{
    const OrcChartGeom g = ChartNodeGeom(F3(in, 0, 1.0f), F3(in, 3, 1.0f), F3(in, 6, 1.0f), F(in, 9));
    P4(out, 0, g.plane); PF(out, 4, g.a.x); PF(out, 5, g.a.y); PF(out, 6, g.b.x); PF(out, 7, g.b.y); PF(out, 8, g.c.x); PF(out, 9, g.c.y); PF(out, 10, g.area);
}
HOOK(Cubemap_MipChain)
{
    const int32_t size = (int32_t)in[0], mip = (int32_t)in[1];
    out[0] = (uint32_t)CalcMipCount(size); out[1] = (uint32_t)CubemapElemCount(size); out[2] = (uint32_t)CalcMipOffset(size, mip);
    out[3] = (uint32_t)CalcMipSize(size, mip < 0 ? 0 : mip);
    PF(out, 4, MipToRoughness((float)mip)); PF(out, 5, RoughnessToMip(MipToRoughness((float)mip)));
}
HOOK(Cubemap_ReadColor_4)
{
    /* the generator's 4x4 faces (oracle/ref_golden.c): data of the record */
    enum { CS = 4 };
    static v3 faces[6][CS * CS];
    const v3* color[6];
    for (int f = 0; f < 6; ++f)
    {
        for (int k = 0; k < CS * CS; ++k) { faces[f][k].x = 0.25f * f + 0.0625f * k; faces[f][k].y = 1.0f / (1 + k) + f; faces[f][k].z = (float)((f * 7 + k * 3) % 11); }
        color[f] = faces[f];
    }
    const v3 c = Cubemap_ReadColor(color, CS, F3(in, 0, 0.0f));
    PF(out, 0, c.x); PF(out, 1, c.y); PF(out, 2, c.z);
}
HOOK(Cubemap_BlendMip)
{
    const int32_t size = (int32_t)in[0], face = (int32_t)in[1], mip = (int32_t)in[2], cx = (int32_t)in[3], cy = (int32_t)in[4];
    const int32_t elems = CubemapElemCount(size);
    v4* conv[6];
    for (int f = 0; f < 6; ++f) { conv[f] = (v4*)malloc(sizeof(v4) * (size_t)elems); for (int k = 0; k < elems; ++k) conv[f][k] = o4_s(-7.0f); }
    const int32_t at0 = CalcMipOffset(size, mip) + CalcMipSize(size, mip) * cy + cx;
    conv[face][at0] = o4(F(in, 5), F(in, 6), F(in, 7), F(in, 8));
    const int32_t at = Cubemap_BlendMip(conv, size, face, cx, cy, mip, o4(F(in, 9), F(in, 10), F(in, 11), F(in, 12)), F(in, 13));
    out[0] = (uint32_t)face; out[1] = (uint32_t)at; P4(out, 2, conv[face][at]);
    for (int f = 0; f < 6; ++f) free(conv[f]);
}
HOOK(f2_tent) { v2 r = f2_tent(o2(F(in, 0), F(in, 1))); PF(out, 0, r.x); PF(out, 1, r.y); }
HOOK(Cubemap_CalcDir) { v4 d = Cubemap_CalcDir((int32_t)in[0], (int)in[1], (int32_t)in[2], (int32_t)in[3], o2(F(in, 4), F(in, 5))); PF(out, 0, d.x); PF(out, 1, d.y); PF(out, 2, d.z); }
HOOK(AmbCube_Fit)
{
    OrcAmbCube c;
    for (int k = 0; k < 6; ++k) c.Values[k] = o4(F(in, k * 4), F(in, k * 4 + 1), F(in, k * 4 + 2), F(in, k * 4 + 3));
    AmbCube_Fit(&c, F(in, 24), F3(in, 25, 0.0f), o4(F(in, 28), F(in, 29), F(in, 30), F(in, 31)));
    for (int k = 0; k < 6; ++k) { PF(out, k * 4, c.Values[k].x); PF(out, k * 4 + 1, c.Values[k].y); PF(out, k * 4 + 2, c.Values[k].z); PF(out, k * 4 + 3, c.Values[k].w); }
}

/* Grid, math/grid.h:8-49 (same arithmetic as pim_oracle.c's grid_*) */
HOOK(Grid)
{
    v4 lo = F3(in, 0, 0.0f), hi = F3(in, 3, 0.0f); float cpm = F(in, 6); v4 p = F3(in, 7, 0.0f);
    v4 sizef = o4_ceil(o4_muls(o4_sub(hi, lo), cpm));
    int32_t sx = (int32_t)sizef.x, sy = (int32_t)sizef.y, sz = (int32_t)sizef.z;
    v4 offset = o4_muls(o4_sub(p, lo), cpm);
    int32_t x = oi_clamp((int32_t)offset.x, 0, sx - 1), y = oi_clamp((int32_t)offset.y, 0, sy - 1), z = oi_clamp((int32_t)offset.z, 0, sz - 1);
    int32_t idx = x + y * sx + z * sx * sy;
    const float mpc = 1.0f / cpm;
    int32_t ix = idx % sx, iy = (idx / sx) % sy, iz = idx / (sx * sy);
    v4 c = o4_add(lo, o4((ix + 0.5f) * mpc, (iy + 0.5f) * mpc, (iz + 0.5f) * mpc, 0.0f));
    out[0] = (uint32_t)sx; out[1] = (uint32_t)sy; out[2] = (uint32_t)sz; out[3] = (uint32_t)idx; P3(out, 4, c);
}

/* Dist1D: the functions live in pim_oracle.c; exported test shims below */
float orc_test_dist_bake(float* pdf, float* cdf, int32_t len);
int32_t orc_test_dist_find(const float* cdf, int32_t size, float u);
void orc_test_dist_update(float* pdf, float* cdf, uint32_t* live, int32_t len, uint32_t* sum, float* integral);

HOOK(Dist1D_Bake_Sample)
{
    float pdf[12], cdf[13];
    for (int k = 0; k < 12; ++k) pdf[k] = F(in, k);
    float u = F(in, 12);
    float integral = orc_test_dist_bake(pdf, cdf, 12);
    for (int k = 0; k < 12; ++k) PF(out, k, pdf[k]);
    for (int k = 0; k < 13; ++k) PF(out, 12 + k, cdf[k]);
    PF(out, 25, integral);
    int32_t s = orc_test_dist_find(cdf, 13, u);
    out[26] = (uint32_t)s;
    PF(out, 27, pdf[s] / 12);
}
HOOK(Dist1D_Update)
{
    float pdf[12], cdf[13]; uint32_t live[12];
    for (int k = 0; k < 12; ++k) pdf[k] = F(in, k);
    float integral = orc_test_dist_bake(pdf, cdf, 12);
    for (int k = 0; k < 12; ++k) live[k] = in[12 + k];
    uint32_t sum = in[24];
    orc_test_dist_update(pdf, cdf, live, 12, &sum, &integral);
    for (int k = 0; k < 12; ++k) PF(out, k, pdf[k]);
    for (int k = 0; k < 13; ++k) PF(out, 12 + k, cdf[k]);
    for (int k = 0; k < 12; ++k) out[25 + k] = live[k];
    out[37] = sum; PF(out, 38, integral);
}

HOOK(SG_Accumulate6)
{
    static const float kAx[5][4] =
    {
        { 0.000000f, 0.000000f, 1.000000f, 4.999773f }, { 0.577350f, 0.577350f, 0.577350f, 4.999773f },
        { -0.577350f, 0.577350f, 0.577350f, 4.999773f }, { 0.577350f, -0.577350f, 0.577350f, 4.999773f },
        { -0.577350f, -0.577350f, 0.577350f, 4.999773f },
    };
    v4 N = F3(in, 0, 0.0f);
    m33 TBN = NormalToTBN(N);
    v4 axii[5], probes[5];
    for (int k = 0; k < 5; ++k) { v4 ax = o4(kAx[k][0], kAx[k][1], kAx[k][2], kAx[k][3]); float sh = ax.w; ax = TbnToWorld(TBN, ax); ax.w = sh; axii[k] = ax; probes[k] = o4_s(0.0f); }
    float sampleCount = 1.0f;
    for (int s = 0; s < 6; ++s)
    {
        v2 Xi = o2(F(in, 3 + s * 5), F(in, 4 + s * 5));
        v4 rad = F3(in, 5 + s * 5, 0.0f);
        v4 rd = TbnToWorld(TBN, SampleUnitHemisphere(Xi));
        float weight = 1.0f / sampleCount; sampleCount += 1.0f;
        SG_Accumulate(weight, rd, rad, axii, probes, 5);
    }
    for (int k = 0; k < 5; ++k) P4(out, k * 4, probes[k]);
}
HOOK(GradientNoise3) { PF(out, 0, GradientNoise3(F3(in, 0, 0.0f), in[3])); }
HOOK(FbmGradientNoise3) { PF(out, 0, FbmGradientNoise3(F3(in, 0, 0.0f), 2.0666f, 0.9f, (int32_t)in[3], 1)); }
HOOK(MiePhase) { PF(out, 0, MiePhase(F(in, 0), F(in, 1))); }
HOOK(proj_slope) { v2 s = proj_slope(F(in, 0), F(in, 1)); PF(out, 0, s.x); PF(out, 1, s.y); }
HOOK(camera_basis)
{
    v4 q = quat_lookat(F3(in, 0, 0.0f), F3(in, 3, 0.0f));
    v4 r = quat_right(q), u = quat_up(q), f = quat_fwd(q);
    v2 slope = proj_slope(o_radians(90.0f), 16.0f / 9.0f);
    v4 d = proj_dir(r, u, f, slope, o2(F(in, 6), F(in, 7)));
    P4(out, 0, q); P3(out, 4, r); P3(out, 7, u); P3(out, 10, f); P3(out, 13, d);
}
HOOK(TriArea_sdTriangle)
{
    v4 a = F3(in, 0, 1.0f), b = F3(in, 3, 1.0f), c = F3(in, 6, 1.0f), p = F3(in, 9, 1.0f);
    PF(out, 0, TriArea3D(a, b, c)); PF(out, 1, sdTriangle3D(a, b, c, p));
}
HOOK(trs_flatten)
{
    v4 t = F3(in, 0, 0.0f), q = o4(F(in, 3), F(in, 4), F(in, 5), F(in, 6)), s = F3(in, 7, 0.0f);
    v4 p = F3(in, 10, 1.0f), n = F3(in, 13, 0.0f);
    m44 M = m44_trs(t, q, s);
    m33 IM = m44_IM(M);
    P4(out, 0, m44_mul_pt(M, p));
    P3(out, 4, o4_normalize3(m33_mul_col(IM, n)));
}
HOOK(box_from_pts)
{
    v4 lo = o4_s((float)(1 << 23)), hi = o4_s(-(float)(1 << 23));
    for (int k = 0; k < 8; ++k) { v4 p = F3(in, k * 3, 1.0f); lo = o4_min(lo, p); hi = o4_max(hi, p); }
    P3(out, 0, lo); P3(out, 3, hi);
}
