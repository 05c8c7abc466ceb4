/*
 * ref_golden.c -- TEST INFRASTRUCTURE.  Golden-vector generator.
 *
 * Compiled ONLY in the build container, where /root/reference exists, by
 * oracle/Makefile target `_ref/ref_golden`:
 *
 *   gcc -std=c11 -O2 -ffp-contract=off -ffunction-sections -Wl,--gc-sections \
 *       -I/root/reference/src  oracle/ref_golden.c \
 *       /root/reference/src/math/dist1d.c /root/reference/src/math/sphgauss.c \
 *       /root/reference/src/rendering/cubemap.c -lm
 *
 * It #includes the reference's own header-only math (and links dist1d.c /
 * sphgauss.c where they lie; the allocator/console users in those files are
 * unreferenced and dropped by --gc-sections, so no stand-in is written) and
 * evaluates each pure function of the hot path on deterministic pseudo-random
 * inputs.  The inputs and the reference's outputs go to
 * tests/golden/ref_math.bin, which is committed; the reference sources are not
 * copied and do not travel.
 *
 * File format (little endian u32 words):
 *   "PIMG", version, nRecords, then per record:
 *   nameLen, name bytes padded to 4, nCases, nIn, nOut, nCases*(nIn+nOut) words
 *   (float values are stored as their IEEE-754 bit patterns).
 *
 * The reference is built WITHOUT -ffast-math here (its CMake adds it,
 * CMakeLists.txt:168) so that the vectors are IEEE-reproducible.
 */
#include "math/types.h"
#include "math/scalar.h"
#include "math/float2_funcs.h"
#include "math/float4_funcs.h"
#include "math/float3x3_funcs.h"
#include "math/float4x4_funcs.h"
#include "math/quat_funcs.h"
#include "math/sampling.h"
#include "math/lighting.h"
#include "math/dist1d.h"
#include "math/grid.h"
#include "math/pcg.h"
#include "math/noise.h"
#include "math/atmosphere.h"
#include "math/sphgauss.h"
#include "math/frustum.h"
#include "math/box.h"
#include "math/area.h"
#include "math/sdf.h"
#include "math/color.h"
#include "common/random.h"
#include "rendering/sampler.h"
#include "rendering/cubemap.h"
#include "math/ambcube.h"

/* rendering/lightmap.h cannot be included (it pulls vulkan/vkr.h -> <volk/volk.h>, an empty
 * submodule); the five lobe axes it defines (lightmap.h:14-21) are input DATA of the
 * SG_Accumulate record below and are stated here as such. */
#define kGiDirections 5
static const float4 kGiAxii[kGiDirections] =
{
    { 0.000000f, 0.000000f, 1.000000f, 4.999773f },
    { 0.577350f, 0.577350f, 0.577350f, 4.999773f },
    { -0.577350f, 0.577350f, 0.577350f, 4.999773f },
    { 0.577350f, -0.577350f, 0.577350f, 4.999773f },
    { -0.577350f, -0.577350f, 0.577350f, 4.999773f },
};

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the table BrdfLut_Sample reads (extern in math/lighting.h:45): data of the GGXEnergyCompensation record, see main() */
BrdfLut g_BrdfLut;


typedef union { float f; u32 u; i32 i; } word;

static u32 s_lcg = 0x12345u;
static u32 nextu(void) { s_lcg = s_lcg * 1664525u + 1013904223u; return Pcg1(s_lcg); }
static float rnd(void) { return (nextu() >> 8) * (1.0f / (1 << 24)); }
static float rnds(float lo, float hi) { return lo + (hi - lo) * rnd(); }
static float4 rndunit(void)
{
    float4 v;
    do { v = f4_v(rnds(-1, 1), rnds(-1, 1), rnds(-1, 1), 0.0f); } while (f4_dot3(v, v) < 0.01f || f4_dot3(v, v) > 1.0f);
    return f4_normalize3(v);
}

static FILE* s_out;
static u32 s_records;
static word* s_buf; static u32 s_len, s_cap;
static void put(word w) { if (s_len == s_cap) { s_cap = s_cap ? s_cap * 2 : 1024; s_buf = realloc(s_buf, s_cap * sizeof(word)); } s_buf[s_len++] = w; }
static void putf(float f) { word w; w.f = f; put(w); }
static void putu(u32 u) { word w; w.u = u; put(w); }
static void begin(void) { s_len = 0; }
static void end(const char* name, u32 nCases, u32 nIn, u32 nOut)
{
    if (s_len != nCases * (nIn + nOut)) { fprintf(stderr, "record %s: %u words, expected %u\n", name, s_len, nCases * (nIn + nOut)); exit(1); }
    u32 nameLen = (u32)strlen(name);
    char padded[64] = { 0 };
    memcpy(padded, name, nameLen);
    u32 hdr = nameLen;
    fwrite(&hdr, 4, 1, s_out);
    fwrite(padded, 1, (nameLen + 3) & ~3u, s_out);
    u32 h3[3] = { nCases, nIn, nOut };
    fwrite(h3, 4, 3, s_out);
    fwrite(s_buf, 4, s_len, s_out);
    ++s_records;
}
#define PUT3(v) do { putf((v).x); putf((v).y); putf((v).z); } while (0)
#define PUT4(v) do { putf((v).x); putf((v).y); putf((v).z); putf((v).w); } while (0)

#define N 256

int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "ref_math.bin";
    s_out = fopen(path, "wb");
    if (!s_out) { perror(path); return 1; }
    u32 hdr[3] = { 0x474D4950u /* "PIMG" */, 1u, 0u };
    fwrite(hdr, 4, 3, s_out);

    /* ---- integer hashes / RNG ---- */
    begin(); for (int i = 0; i < N; ++i) { u32 a = nextu(); putu(a); putu(Pcg1(a)); } end("pcg1", N, 1, 1);
    begin(); for (int i = 0; i < N; ++i) { u32 a = nextu(), b = nextu(); uint2 r = Pcg2(a, b); putu(a); putu(b); putu(r.x); putu(r.y); } end("pcg2", N, 2, 2);
    begin(); for (int i = 0; i < N; ++i) { u32 a = nextu(), b = nextu(), c = nextu(); uint3 r = Pcg3(a, b, c); putu(a); putu(b); putu(c); putu(r.x); putu(r.y); putu(r.z); } end("pcg3", N, 3, 3);
    begin(); for (int i = 0; i < N; ++i) { u32 a = nextu(), b = nextu(), c = nextu(), d = nextu(); uint4 r = Pcg4(a, b, c, d); putu(a); putu(b); putu(c); putu(d); putu(r.x); putu(r.y); putu(r.z); putu(r.w); } end("pcg4", N, 4, 4);
    begin();
    for (int i = 0; i < N; ++i)
    {
        Prng rng; rng.state = (uint4){ nextu(), nextu(), nextu(), nextu() };
        putu(rng.state.x); putu(rng.state.y); putu(rng.state.z); putu(rng.state.w);
        putu(Prng_u32(&rng));
        putf(Prng_f32(&rng));
        float2 f2 = Prng_float2(&rng); putf(f2.x); putf(f2.y);
        float4 f4 = Prng_float4(&rng); PUT4(f4);
        putu(rng.state.x); putu(rng.state.y); putu(rng.state.z); putu(rng.state.w);
    }
    end("prng_seq", N, 4, 12);

    /* ---- sampling warps ---- */
    begin(); for (int i = 0; i < N; ++i) { float4 n = rndunit(); if (i == 0) n = f4_v(0, 0, 1, 0); if (i == 1) n = f4_normalize3(f4_v(0.003f, -0.002f, -1.0f, 0)); float3x3 t = NormalToTBN(n); PUT3(n); PUT3(t.c0); PUT3(t.c1); PUT3(t.c2); } end("NormalToTBN", N, 3, 9);
    begin(); for (int i = 0; i < N; ++i) { float4 n = rndunit(), t = rndunit(); float4 r = TanToWorld(n, t); PUT3(n); PUT3(t); PUT3(r); } end("TanToWorld", N, 6, 3);
    begin(); for (int i = 0; i < N; ++i) { float2 Xi = { rnd(), rnd() }; if (i == 0) Xi = f2_v(0, 0); if (i == 1) Xi = f2_v(0.5f, 0.5f); float2 r = MapSquareToDisk(Xi); putf(Xi.x); putf(Xi.y); putf(r.x); putf(r.y); } end("MapSquareToDisk", N, 2, 2);
    begin(); for (int i = 0; i < N; ++i) { float2 Xi = { rnd(), rnd() }; float4 r = SampleCosineHemisphere(Xi); putf(Xi.x); putf(Xi.y); PUT3(r); } end("SampleCosineHemisphere", N, 2, 3);
    begin(); for (int i = 0; i < N; ++i) { float2 Xi = { rnd(), rnd() }; float a = rnds(kMinAlpha, 1.0f); float4 r = SampleGGXMicrofacet(Xi, a); putf(Xi.x); putf(Xi.y); putf(a); PUT3(r); } end("SampleGGXMicrofacet", N, 3, 3);
    begin(); for (int i = 0; i < N; ++i) { float2 Xi = { rnd(), rnd() }; if (i == 0) Xi.x = 0.0f; float4 r = SampleBaryCoord(Xi); putf(Xi.x); putf(Xi.y); PUT3(r); } end("SampleBaryCoord", N, 2, 3);
    begin(); for (int i = 0; i < N; ++i) { float2 Xi = { rnd(), rnd() }; float4 r = SampleUnitSphere(Xi); putf(Xi.x); putf(Xi.y); PUT3(r); } end("SampleUnitSphere", N, 2, 3);
    begin(); for (int i = 0; i < N; ++i) { float2 Xi = { rnd(), rnd() }; float4 r = SampleUnitHemisphere(Xi); putf(Xi.x); putf(Xi.y); PUT3(r); } end("SampleUnitHemisphere", N, 2, 3);
    begin(); for (int i = 0; i < N; ++i) { float2 Xi = { rnd(), rnd() }; u32 side = nextu(); u32 n = 3 + (nextu() % 8); float rot = rnds(0, kTau); float2 r = SampleNGon(Xi, side, n, rot); putf(Xi.x); putf(Xi.y); putu(side); putu(n); putf(rot); putf(r.x); putf(r.y); } end("SampleNGon", N, 5, 2);
    begin(); for (int i = 0; i < N; ++i) { float2 Xi = { rnd(), rnd() }; u32 side = nextu(); float2 r = SamplePentagram(Xi, side); putf(Xi.x); putf(Xi.y); putu(side); putf(r.x); putf(r.y); } end("SamplePentagram", N, 3, 2);
    begin(); for (int i = 0; i < N; ++i) { float2 Xi = { rnd(), rnd() }; float sd = (i & 1) ? 1.0f : 0.5f; float2 r = SampleGaussPixelFilter(Xi, sd); putf(Xi.x); putf(Xi.y); putf(sd); putf(r.x); putf(r.y); } end("SampleGaussPixelFilter", N, 3, 2);
    begin(); for (int i = 0; i < N; ++i) { float Xi = rnd(); float mfp = (i & 1) ? 11363.636f : rnds(0.1f, 100.0f); putf(Xi); putf(mfp); putf(SampleFreePath(Xi, mfp)); } end("SampleFreePath", N, 2, 1);
    begin(); for (int i = 0; i < N; ++i) { float f = rnds(0, 10), g = rnds(0, 10); if (i == 0) { f = 0; g = 0; } putf(f); putf(g); putf(PowerHeuristic(f, g)); } end("PowerHeuristic", N, 2, 1);
    begin(); for (int i = 0; i < N; ++i) { float a = rnds(0, 4), c = rnd(), d = rnds(0, 100); if (i == 0) c = 0; putf(a); putf(c); putf(d); putf(LightPdf(a, c, d)); } end("LightPdf", N, 3, 1);
    begin(); for (int i = 0; i < N; ++i) { float a = rnd(), b = rnd(), c = rnds(kMinAlpha, 1); putf(a); putf(b); putf(c); putf(GGXPdf(a, b, c)); } end("GGXPdf", N, 3, 1);
    begin(); for (int i = 0; i < N; ++i) { u32 a = (u32)i, n = 16 + (nextu() % 64); float2 r = Hammersley2D(a, n); putu(a); putu(n); putf(r.x); putf(r.y); } end("Hammersley2D", N, 2, 2);

    /* ---- BRDF terms ---- */
    begin(); for (int i = 0; i < N; ++i) { float c = rnds(-1.2f, 1.2f); float a = (i & 1) ? 1.0f : 1.000277f, b = (i & 2) ? 1.5f : rnds(1.0f, 2.5f); putf(c); putf(a); putf(b); putf(F_Dielectric(c, a, b)); } end("F_Dielectric", N, 3, 1);
    begin(); for (int i = 0; i < N; ++i) { float a = rnd(), b = rnds(kMinAlpha, 1); putf(a); putf(b); putf(D_GTR(a, b)); } end("D_GTR", N, 2, 1);
    begin(); for (int i = 0; i < N; ++i) { float a = rnd(), b = rnd(), c = rnds(kMinAlpha, 1); putf(a); putf(b); putf(c); putf(V_SmithCorrelated(a, b, c)); } end("V_SmithCorrelated", N, 3, 1);
    begin(); for (int i = 0; i < N; ++i) { float a = rnd(), b = rnd(), c = rnd(), d = rnd(); putf(a); putf(b); putf(c); putf(d); putf(Fd_Burley(a, b, c, d)); } end("Fd_Burley", N, 4, 1);
    begin(); for (int i = 0; i < N; ++i) { float4 al = f4_v(rnd(), rnd(), rnd(), 1.0f); float m = rnd(); float4 f0 = F_0(al, m); PUT3(al); putf(m); PUT3(f0); putf(F_90(f0)); } end("F_0_F_90", N, 4, 4);
    begin(); for (int i = 0; i < N; ++i) { float4 al = f4_v(rnds(0.01f, 1), rnds(0.01f, 1), rnds(0.01f, 1), 1.0f); float r = rnd(), th = rnds(0, 10); float4 t = AlbedoToTransmittance(al, r, th); PUT3(al); putf(r); putf(th); PUT3(t); } end("AlbedoToTransmittance", N, 5, 3);
    begin(); for (int i = 0; i < N; ++i) { float r = rnd(); if (i == 0) r = 0; putf(r); putf(BrdfAlpha(r)); } end("BrdfAlpha", N, 1, 1);
    begin(); for (int i = 0; i < N; ++i) { float4 a = rndunit(), n = rndunit(); float4 r = f4_reflect3(a, n); PUT3(a); PUT3(n); PUT3(r); } end("reflect3", N, 6, 3);
    begin(); for (int i = 0; i < N; ++i) { float4 a = rndunit(), n = rndunit(); if (f4_dot3(a, n) > 0) n = f4_neg(n); float k = (i & 1) ? 1.000277f / 1.5f : 1.5f / 1.000277f; float4 r = f4_refract3(a, n, k); PUT3(a); PUT3(n); putf(k); PUT3(r); } end("refract3", N, 7, 3);

    /* ---- texel decode / addressing ---- */
    begin(); for (int i = 0; i < N; ++i) { word w; w.u = nextu(); if (i == 0) w.u = 0xffffffffu; R8G8B8A8_t c; memcpy(&c, &w.u, 4); float4 r = GammaDecode_rgba8(c); put(w); PUT4(r); } end("GammaDecode_rgba8", N, 1, 4);
    begin(); for (int i = 0; i < N; ++i) { float4 v = f4_v(rnds(-0.1f, 1.1f), rnd(), rnd(), rnd()); if (i == 0) v = f4_s(1.0f); if (i == 1) v = f4_v(0.9f, 1.0f, 0.0f, 1.0f); R8G8B8A8_t c = GammaEncode_rgba8(v); word w; memcpy(&w.u, &c, 4); PUT4(v); put(w); } end("GammaEncode_rgba8", N, 4, 1);
    begin(); for (int i = 0; i < N; ++i) { float4 al = f4_v(rnd(), rnd(), rnd(), 1.0f); float e = rnd(); float4 r = UnpackEmission(al, e); PUT3(al); putf(e); PUT3(r); } end("UnpackEmission", N, 4, 3);
    begin(); for (int i = 0; i < N; ++i) { word w; w.u = nextu(); short2 s; memcpy(&s, &w.u, 4); float4 r = Xy16ToNormalTs(s); put(w); PUT3(r); } end("Xy16ToNormalTs", N, 1, 3);
    begin();
    for (int i = 0; i < N; ++i)
    {
        int2 size = { 1 + (i32)(nextu() % 9), 1 + (i32)(nextu() % 9) }; if (i < 4) size = i2_1;
        float2 uv = { rnds(-3, 3), rnds(-3, 3) };
        bilinear_t w = BilinearWrap(size, uv), c = BilinearClamp(size, uv);
        putu((u32)size.x); putu((u32)size.y); putf(uv.x); putf(uv.y);
        putu(w.a); putu(w.b); putu(w.c); putu(w.d); putf(w.frac.x); putf(w.frac.y);
        putu(c.a); putu(c.b); putu(c.c); putu(c.d); putf(c.frac.x); putf(c.frac.y);
        putu((u32)PointWrap1D(size.x, uv.x)); putu((u32)PointWrap1D(size.y, uv.y));
    }
    end("Bilinear", N, 4, 14);
    begin();
    for (int i = 0; i < N; ++i)
    {
        u32 tex[16]; for (int k = 0; k < 16; ++k) { tex[k] = nextu(); putu(tex[k]); }
        float2 uv = { rnds(-2, 2), rnds(-2, 2) }; putf(uv.x); putf(uv.y);
        float4 a = UvBilinearWrap_c32((const R8G8B8A8_t*)tex, i2_s(4), uv);
        float4 b = UvBilinearWrap_xy16((const short2*)tex, i2_s(4), uv);
        PUT4(a); PUT3(b);
    }
    end("UvBilinearWrap_4x4", N, 18, 7);
    begin();
    for (int i = 0; i < N; ++i)
    {
        float tex[48]; for (int k = 0; k < 48; ++k) { tex[k] = rnd(); putf(tex[k]); }
        float2 uv = { rnds(-0.2f, 1.2f), rnds(-0.2f, 1.2f) }; putf(uv.x); putf(uv.y);
        float2 a = UvBilinearClamp_f2((const float2*)tex, i2_s(4), uv);   /* first 32 floats */
        float3 b = UvBilinearClamp_f3((const float3*)tex, i2_s(4), uv);   /* all 48 floats */
        putf(a.x); putf(a.y); PUT3(b);
    }
    end("UvBilinearClamp_4x4", N, 50, 5);
    begin(); for (int i = 0; i < N; ++i) { float4 d = rndunit(); if (i < 6) { d = f4_0; ((float*)&d)[i >> 1] = (i & 1) ? -1.0f : 1.0f; } float2 uv; Cubeface f = Cubemap_CalcUv(d, &uv); PUT3(d); putu((u32)f); putf(uv.x); putf(uv.y); } end("Cubemap_CalcUv", N, 3, 3);

    /* ---- grid ---- */
    begin();
    for (int i = 0; i < N; ++i)
    {
        Box3D b; b.lo = f4_v(rnds(-10, 0), rnds(-10, 0), rnds(-10, 0), 0); b.hi = f4_add(b.lo, f4_v(rnds(0.1f, 20), rnds(0.1f, 20), rnds(0.1f, 20), 0));
        float cpm = 1.0f / ((i & 1) ? 1.5f : rnds(0.5f, 3.0f));
        float4 p = f4_v(rnds(-12, 12), rnds(-12, 12), rnds(-12, 12), 0);
        Grid g; Grid_New(&g, b, cpm);
        i32 idx = Grid_Index(&g, p);
        float4 c = Grid_Position(&g, idx);
        PUT3(b.lo); PUT3(b.hi); putf(cpm); PUT3(p);
        putu(g.size.x); putu(g.size.y); putu(g.size.z); putu((u32)idx); PUT3(c);
    }
    end("Grid", N, 10, 7);

    /* ---- Dist1D (math/dist1d.c linked where it lies) ---- */
    begin();
    for (int i = 0; i < N; ++i)
    {
        float pdf[12], cdf[13]; u32 live[12];
        Dist1D d; memset(&d, 0, sizeof(d)); d.pdf = pdf; d.cdf = cdf; d.live = live; d.length = 12;
        for (int k = 0; k < 12; ++k) { pdf[k] = (i == 0) ? 0.0f : ((nextu() & 3) ? (float)(nextu() % 17) / 16.0f : 0.0f); live[k] = 0; putf(pdf[k]); }
        float u = rnd(); putf(u);
        Dist1D_Bake(&d);
        for (int k = 0; k < 12; ++k) putf(pdf[k]);
        for (int k = 0; k < 13; ++k) putf(cdf[k]);
        putf(d.integral);
        i32 s = Dist1D_SampleD(&d, u); putu((u32)s); putf(Dist1D_PdfD(&d, s));
    }
    end("Dist1D_Bake_Sample", N, 13, 28);
    begin();
    for (int i = 0; i < N; ++i)
    {
        float pdf[12], cdf[13]; u32 live[12];
        Dist1D d; memset(&d, 0, sizeof(d)); d.pdf = pdf; d.cdf = cdf; d.live = live; d.length = 12;
        for (int k = 0; k < 12; ++k) { pdf[k] = (float)(nextu() % 17) / 16.0f; putf(pdf[k]); }
        Dist1D_Bake(&d);
        u32 scale = (i & 1) ? 3 : 300;
        for (int k = 0; k < 12; ++k) { live[k] = (nextu() & 1) ? nextu() % scale : 0; putu(live[k]); }
        d.sum = (i & 2) ? nextu() % 2000 : 0; putu(d.sum);
        Dist1D_Update(&d);
        for (int k = 0; k < 12; ++k) putf(pdf[k]);
        for (int k = 0; k < 13; ++k) putf(cdf[k]);
        for (int k = 0; k < 12; ++k) putu(live[k]);
        putu(d.sum); putf(d.integral);
    }
    end("Dist1D_Update", N, 25, 39);

    /* ---- spherical gaussians (math/sphgauss.c linked where it lies; kGiAxii from lightmap.h:14-21) ---- */
    begin();
    for (int i = 0; i < N; ++i)
    {
        float4 Nn = rndunit(); PUT3(Nn);
        float3x3 TBN = NormalToTBN(Nn);
        float4 axii[kGiDirections], probes[kGiDirections];
        for (int k = 0; k < kGiDirections; ++k) { float4 ax = kGiAxii[k]; float sh = ax.w; ax = TbnToWorld(TBN, ax); ax.w = sh; axii[k] = ax; probes[k] = f4_0; }
        float sampleCount = 1.0f;
        for (int s = 0; s < 6; ++s)
        {
            float2 Xi = { rnd(), rnd() }; float4 rad = f4_v(rnds(0, 5), rnds(0, 5), rnds(0, 5), 0.0f);
            putf(Xi.x); putf(Xi.y); PUT3(rad);
            float4 rd = TbnToWorld(TBN, SampleUnitHemisphere(Xi));
            float weight = 1.0f / sampleCount; sampleCount += 1.0f;
            SG_Accumulate(weight, rd, rad, axii, probes, kGiDirections);
        }
        for (int k = 0; k < kGiDirections; ++k) PUT4(probes[k]);
    }
    end("SG_Accumulate6", N, 3 + 6 * 5, 20);

    /* ---- media helpers ---- */
    begin(); for (int i = 0; i < N; ++i) { float4 p = f4_v(rnds(-50, 50), rnds(-50, 50), rnds(-50, 50), 0); u32 seed = 1 + (nextu() % 4); putf(p.x); putf(p.y); putf(p.z); putu(seed); putf(GradientNoise3(p, seed)); } end("GradientNoise3", N, 4, 1);
    begin(); for (int i = 0; i < N; ++i) { float4 p = f4_v(rnds(-50, 50), rnds(-50, 50), rnds(-50, 50), 0); i32 oct = 1 + (i32)(nextu() % 4); putf(p.x); putf(p.y); putf(p.z); putu((u32)oct); putf(FbmGradientNoise3(p, 2.0666f, 0.9f, oct, 1)); } end("FbmGradientNoise3", N, 4, 1);
    begin(); for (int i = 0; i < N; ++i) { float c = rnds(-1, 1), g = rnds(-0.99f, 0.99f); if (i & 1) g = 0.0f; putf(c); putf(g); putf(MiePhase(c, g)); } end("MiePhase", N, 2, 1);

    /* ---- camera ---- */
    begin(); for (int i = 0; i < N; ++i) { float fov = f1_radians(rnds(30, 120)), asp = rnds(0.5f, 2.5f); float2 s = proj_slope(fov, asp); putf(fov); putf(asp); putf(s.x); putf(s.y); } end("proj_slope", N, 2, 2);
    begin();
    for (int i = 0; i < N; ++i)
    {
        float4 fwd = rndunit(); float4 up = f4_v(0, 1, 0, 0);
        if (i == 0) fwd = f4_normalize3(f4_sub(f4_v(0, -1, 0, 0), f4_v(-4, 0, 4, 0)));
        if (f1_abs(fwd.y) > 0.99f) fwd = f4_normalize3(f4_v(0.3f, 0.5f, -0.4f, 0));
        quat q = quat_lookat(fwd, up);
        float4 r = quat_right(q), u = quat_up(q), f = quat_fwd(q);
        float2 slope = proj_slope(f1_radians(90.0f), 16.0f / 9.0f); float2 coord = { rnds(-1, 1), rnds(-1, 1) };
        float4 d = proj_dir(r, u, f, slope, coord);
        PUT3(fwd); PUT3(up); putf(coord.x); putf(coord.y);
        PUT4(q.v); PUT3(r); PUT3(u); PUT3(f); PUT3(d);
    }
    end("camera_basis", N, 8, 16);

    /* ---- geometry ---- */
    begin(); for (int i = 0; i < N; ++i) { float4 a = f4_v(rnds(-5, 5), rnds(-5, 5), rnds(-5, 5), 1), b = f4_v(rnds(-5, 5), rnds(-5, 5), rnds(-5, 5), 1), c = f4_v(rnds(-5, 5), rnds(-5, 5), rnds(-5, 5), 1), p = f4_v(rnds(-6, 6), rnds(-6, 6), rnds(-6, 6), 1.0f); PUT3(a); PUT3(b); PUT3(c); PUT3(p); putf(TriArea3D(a, b, c)); putf(sdTriangle3D(a, b, c, p)); } end("TriArea_sdTriangle", N, 12, 2);
    begin();
    for (int i = 0; i < N; ++i)
    {
        float4 t = f4_v(rnds(-5, 5), rnds(-5, 5), rnds(-5, 5), 0);
        float4 fwd = rndunit(); if (f1_abs(fwd.y) > 0.99f) fwd = f4_v(1, 0, 0, 0);
        quat q = quat_lookat(fwd, f4_v(0, 1, 0, 0));
        float4 s = f4_v(rnds(0.1f, 10), rnds(0.1f, 10), rnds(0.1f, 10), 0);
        float4 p = f4_v(rnds(-0.5f, 0.5f), rnds(-0.5f, 0.5f), rnds(-0.5f, 0.5f), 1.0f);
        float4 n = rndunit();
        float4x4 M = f4x4_trs(t, q, s);
        float3x3 IM = f3x3_IM(M);
        float4 wp = f4x4_mul_pt(M, p);
        float4 wn = f4_normalize3(f3x3_mul_col(IM, n));
        PUT3(t); PUT4(q.v); PUT3(s); PUT3(p); PUT3(n); PUT4(wp); PUT3(wn);
    }
    end("trs_flatten", N, 16, 7);
    begin();
    for (int i = 0; i < 32; ++i)
    {
        float4 pts[8]; for (int k = 0; k < 8; ++k) { pts[k] = f4_v(rnds(-9, 9), rnds(-9, 9), rnds(-9, 9), 1.0f); PUT3(pts[k]); }
        Box3D b = box_from_pts(pts, 8); PUT3(b.lo); PUT3(b.hi);
    }
    end("box_from_pts", 32, 24, 6);

    /* patch record count */
    /* ---- callers of Pt_TraceRay besides the image: Cubemap_Bake (cubemap.c:130-190), AmbCube_Bake (ambcube.c:5-32) ---- */
    begin(); for (int i = 0; i < N; ++i) { float2 Xi = { rnd(), rnd() }; if (i < 4) { Xi.x = (i & 1) ? 0.5f : 0.0f; Xi.y = (i & 2) ? 0.5f : 0.0f; } float2 r = f2_tent(Xi); putf(Xi.x); putf(Xi.y); putf(r.x); putf(r.y); } end("f2_tent", N, 2, 2);
    begin();
    for (int i = 0; i < N; ++i)
    {
        i32 size = (i & 1) ? 64 : 2 + (i32)(nextu() % 30u);
        i32 face = (i32)(nextu() % 6u);
        int2 coord = { (i32)(nextu() % (u32)size), (i32)(nextu() % (u32)size) };
        float2 Xi = { rnds(-1, 1), rnds(-1, 1) };
        float4 d = Cubemap_CalcDir(size, (Cubeface)face, coord, Xi);
        putu((u32)size); putu((u32)face); putu((u32)coord.x); putu((u32)coord.y); putf(Xi.x); putf(Xi.y); PUT3(d);
    }
    end("Cubemap_CalcDir", N, 6, 3);
    begin();
    for (int i = 0; i < N; ++i)
    {
        AmbCube_t c;
        for (int k = 0; k < 6; ++k) c.Values[k] = f4_v(rnds(0, 4), rnds(0, 4), rnds(0, 4), rnds(0, 1));
        float w = rnds(0.0f, 1.0f);
        float4 dir = rndunit(); float4 rad = f4_v(rnds(0, 8), rnds(0, 8), rnds(0, 8), 0.0f);
        for (int k = 0; k < 6; ++k) PUT4(c.Values[k]);
        putf(w); PUT3(dir); PUT4(rad);
        c = AmbCube_Fit(c, w, dir, rad);
        for (int k = 0; k < 6; ++k) PUT4(c.Values[k]);
    }
    end("AmbCube_Fit", N, 32, 24);

    /* ---- one sample of the BRDF table bake (math/lighting.c:53-80; BakeBrdfFn itself is a static function of a
     * translation unit that needs the Vulkan headers, so the record evaluates the reference's own header functions
     * in the order that loop calls them) ---- */
    begin();
    for (int i = 0; i < N; ++i)
    {
        const i32 size = (i & 1) ? 128 : 32;
        const i32 cx = (i32)(nextu() % (u32)size), cy = (i < 8) ? 0 : (i32)(nextu() % (u32)size);
        const i32 spp = 1 + (i32)(nextu() % 16u);
        const float2 XiAA = { rnd(), rnd() }, XiH = { rnd(), rnd() };
        const float XiPhi = rnd();
        const float rcp = 1.0f / size, wgt = 1.0f / spp;
        const float2 jit = SampleGaussPixelFilter(XiAA, 0.5f);
        const float cosV = f1_clamp((cx + jit.x) * rcp, kEpsilon, 1.0f - kEpsilon);
        const float a = f1_clamp((cy + jit.y) * rcp, kMinAlpha, 1.0f);
        const float4 view = SphericalToCartesian(cosV, XiPhi * kTau);
        const float4 half = SampleGGXMicrofacet(XiH, a);
        const float4 lightDir = f4_reflect3(f4_neg(view), half);
        const float hv = f4_dot3(half, view);
        const float p = GGXPdf(half.z, hv, a);
        float outA = 0.0f, outB = 0.0f;
        if ((lightDir.z > kEpsilon) && (p > kEpsilon))
        {
            const float term = (D_GTR(half.z, a) / p) * V_SmithCorrelated(lightDir.z, cosV, a) * lightDir.z * wgt;
            outA = term * F_Dielectric(hv, 1.000293f, 1.52f);
            outB = term;
        }
        putu((u32)cx); putu((u32)cy); putu((u32)size); putu((u32)spp);
        putf(XiAA.x); putf(XiAA.y); putf(XiPhi); putf(XiH.x); putf(XiH.y);
        putf(outA); putf(outB);
    }
    end("brdf_lut_sample", N, 9, 2);

    /* ---- lightmap embedding: sdf.h's 2D functions and the fragment EmbedTaskFn keeps (rendering/lightmap.c:1000-1025;
     * the function itself is static in a translation unit that needs the Vulkan headers, so the record evaluates the
     * reference's own header functions in the order that loop calls them).  Every 8th case has a zero-length edge. ---- */
    begin();
    for (int i = 0; i < N; ++i)
    {
        const float size = (i & 1) ? 1024.0f : 64.0f;
        float2 ua = { rnd(), rnd() }, ub = { rnd(), rnd() }, uc = { rnd(), rnd() };
        if ((i & 3) == 1) { ub.x = ua.x + rnds(-4, 4) / size; ub.y = ua.y + rnds(-4, 4) / size; uc.x = ua.x + rnds(-4, 4) / size; uc.y = ua.y + rnds(-4, 4) / size; }
        if ((i & 7) == 7) ub = ua;
        const float2 A = f2_mulvs(ua, size), B = f2_mulvs(ub, size), C = f2_mulvs(uc, size);
        float2 px = { (float)(i32)(rnd() * size) + 0.5f, (float)(i32)(rnd() * size) + 0.5f };
        if (i & 2) { px.x = (float)(i32)A.x + 0.5f + (float)((i32)(nextu() % 5u) - 2); px.y = (float)(i32)A.y + 0.5f + (float)((i32)(nextu() % 5u) - 2); }
        float4 pa = f4_v(rnds(-50, 50), rnds(-50, 50), rnds(-50, 50), 1.0f), pb = f4_v(rnds(-50, 50), rnds(-50, 50), rnds(-50, 50), 1.0f), pc = f4_v(rnds(-50, 50), rnds(-50, 50), rnds(-50, 50), 1.0f);
        float4 na = rndunit(), nb = rndunit(), nc = rndunit();
        const float dist = sdTriangle2D(A, B, C, px);
        float area = sdEdge2D(A, B, C);
        const float rawArea = area;
        area = pim_max(area, 1e-5f);
        float4 wuv = bary2D(A, B, C, 1.0f / area, px);
        wuv = f4_divvs(wuv, wuv.x + wuv.y + wuv.z);
        const float4 fragPos = f4_blend(pa, pb, pc, wuv);
        float4 fragNor = f4_normalize3(f4_blend(na, nb, nc, wuv));
        fragNor = f4_normalize3(fragNor);
        putf(ua.x); putf(ua.y); putf(ub.x); putf(ub.y); putf(uc.x); putf(uc.y); putf(size); putf(px.x); putf(px.y);
        PUT3(pa); PUT3(pb); PUT3(pc); PUT3(na); PUT3(nb); PUT3(nc);
        putf(dist); putf(rawArea); PUT3(fragPos); PUT3(fragNor);
    }
    end("lm_embed_fragment", N, 27, 8);

    /* ---- lightmap chart node: chartnode_new (rendering/lightmap.c:302-320, static there) from the reference's own header
     * functions in its order: triToPlane, ProjTri = NormalToTBN + two dot products per vertex, scale, TriArea3D ---- */
    begin();
    for (int i = 0; i < N; ++i)
    {
        float4 A = f4_v(rnds(-60, 60), rnds(-60, 60), rnds(-60, 60), 1.0f);
        float4 B = f4_v(A.x + rnds(-3, 3), A.y + rnds(-3, 3), A.z + rnds(-3, 3), 1.0f);
        float4 C = f4_v(A.x + rnds(-3, 3), A.y + rnds(-3, 3), A.z + rnds(-3, 3), 1.0f);
        if ((i & 7) == 3) { B.z = A.z; C.z = A.z; }                 /* axis-aligned faces: normals (0, 0, +-1) */
        if ((i & 7) == 5) { B.x = A.x; C.x = A.x; }
        const float tpu = (i & 1) ? 16.0f : 2.5f;
        Plane3D plane = triToPlane(A, B, C);
        float3x3 TBN = NormalToTBN(plane.value);
        float2 a = f2_mulvs(f2_v(f4_dot3(TBN.c0, A), f4_dot3(TBN.c1, A)), tpu);
        float2 b = f2_mulvs(f2_v(f4_dot3(TBN.c0, B), f4_dot3(TBN.c1, B)), tpu);
        float2 c = f2_mulvs(f2_v(f4_dot3(TBN.c0, C), f4_dot3(TBN.c1, C)), tpu);
        PUT3(A); PUT3(B); PUT3(C); putf(tpu);
        PUT4(plane.value); putf(a.x); putf(a.y); putf(b.x); putf(b.y); putf(c.x); putf(c.y); putf(TriArea3D(A, B, C));
    }
    end("lm_chartnode", N, 10, 11);

    /* appended last so that the records above keep their inputs (one shared pseudo-random stream) */
    /* BrdfLut_Sample / GGXEnergyCompensation (math/lighting.h:52-55,294-303) read the global table g_BrdfLut, which
     * math/lighting.c defines and bakes from a clock-seeded stream.  lighting.c is not linked: the table is INPUT DATA of
     * this record (a random 4x4 float2 table per case, stored with the case) and the global is defined below as such. */
    begin();
    for (int i = 0; i < N; ++i)
    {
        static float2 tex[16];
        for (int k = 0; k < 16; ++k) { tex[k].x = rnds(0.0f, 1.0f); tex[k].y = (i & 7) == 7 ? 0.0f : rnds(0.02f, 1.0f); putf(tex[k].x); putf(tex[k].y); }
        g_BrdfLut.size = i2_s(4); g_BrdfLut.texels = tex;
        float4 f0 = f4_v(rnd(), rnd(), rnd(), 0.0f); float NoV = rnds(-0.1f, 1.1f), alpha = rnds(0.0f, 1.05f);
        PUT3(f0); putf(NoV); putf(alpha);
        float2 a = BrdfLut_Sample(g_BrdfLut, NoV, alpha);
        float4 e = GGXEnergyCompensation(f0, NoV, alpha);
        putf(a.x); putf(a.y); PUT3(e);
    }
    end("GGXEnergyCompensation", N, 37, 5);

    /* ---- Cubemap_Convolve's own pieces (cubemap.c:192-303, cubemap.h:60-69,117-129,169-188, sampler.h:23-51): the mip chain's
     * layout, the roughness of a mip, the read of the colour faces and the blend into a mip texel, on a Cubemap made here
     * (the struct is public; its faces are plain arrays) ---- */
    begin();
    for (int i = 0; i < 64; ++i)
    {
        i32 size = i < 16 ? (1 << (i % 9)) : 1 + (i32)(nextu() % 300u);
        i32 mip = (i32)(nextu() % 12u) - 1;
        int2 s2 = { size, size };
        i32 mipCount = CalcMipCount(s2);
        putu((u32)size); putu((u32)mip);
        putu((u32)mipCount); putu((u32)(CalcMipOffset(s2, mipCount) + 1)); putu((u32)CalcMipOffset(s2, mip)); putu((u32)CalcMipSize(s2, mip < 0 ? 0 : mip).x);
        putf(MipToRoughness((float)mip)); putf(RoughnessToMip(MipToRoughness((float)mip)));
    }
    end("Cubemap_MipChain", 64, 2, 6);
    {
        enum { CS = 4 };
        Cubemap cm; memset(&cm, 0, sizeof(cm));
        cm.size = CS; cm.mipCount = CalcMipCount(i2_s(CS));
        for (int f = 0; f < 6; ++f)
        {
            cm.color[f] = malloc(sizeof(float3) * CS * CS);
            for (int k = 0; k < CS * CS; ++k) cm.color[f][k] = (float3){ 0.25f * f + 0.0625f * k, 1.0f / (1 + k) + f, (float)((f * 7 + k * 3) % 11) };
        }
        begin();
        for (int i = 0; i < N; ++i)
        {
            float4 d = rndunit(); if (i < 6) { d = f4_0; ((float*)&d)[i >> 1] = (i & 1) ? -1.0f : 1.0f; }
            float3 c = Cubemap_ReadColor(&cm, d);
            PUT3(d); putf(c.x); putf(c.y); putf(c.z);
        }
        end("Cubemap_ReadColor_4", N, 3, 3);
        for (int f = 0; f < 6; ++f) free(cm.color[f]);
    }
    begin();
    for (int i = 0; i < N; ++i)
    {
        i32 size = 1 << (nextu() % 7u);
        int2 s2 = { size, size };
        Cubemap cm; memset(&cm, 0, sizeof(cm));
        cm.size = size; cm.mipCount = CalcMipCount(s2);
        i32 elems = CalcMipOffset(s2, cm.mipCount) + 1;                 /* Cubemap_New, cubemap.c:105-106 */
        i32 face = (i32)(nextu() % 6u), mip = (i32)(nextu() % (u32)cm.mipCount);
        i32 msize = CalcMipSize(s2, mip).x;
        int2 coord = { (i32)(nextu() % (u32)msize), (i32)(nextu() % (u32)msize) };
        float4 old = f4_v(rnds(0, 4), rnds(0, 4), rnds(0, 4), rnds(0, 1)), val = f4_v(rnds(0, 9), rnds(0, 9), rnds(0, 9), 1.0f);
        float weight = (i & 7) == 0 ? 1.0f : rnds(0.0f, 1.0f);
        const float4 mark = f4_v(-7.0f, -7.0f, -7.0f, -7.0f);
        for (int f = 0; f < 6; ++f) { cm.convolved[f] = malloc(sizeof(float4) * elems); for (int k = 0; k < elems; ++k) cm.convolved[f][k] = mark; }
        i32 at = CalcMipOffset(s2, mip) + DecodeCoord2(CalcMipSize(s2, mip), coord);
        cm.convolved[face][at] = old;
        Cubemap_BlendMip(&cm, (Cubeface)face, coord, mip, val, weight);
        i32 where = -1, touched = 0;
        for (int f = 0; f < 6; ++f) for (int k = 0; k < elems; ++k) if (memcmp(&cm.convolved[f][k], &mark, sizeof(mark)) != 0) { where = f * elems + k; ++touched; }
        if (touched != 1) { fprintf(stderr, "Cubemap_BlendMip touched %d texels\n", touched); return 1; }
        float4 got = cm.convolved[where / elems][where % elems];
        putu((u32)size); putu((u32)face); putu((u32)mip); putu((u32)coord.x); putu((u32)coord.y); PUT4(old); PUT4(val); putf(weight);
        putu((u32)(where / elems)); putu((u32)(where % elems)); PUT4(got);
        for (int f = 0; f < 6; ++f) free(cm.convolved[f]);
    }
    end("Cubemap_BlendMip", N, 14, 6);

    fseek(s_out, 8, SEEK_SET);
    fwrite(&s_records, 4, 1, s_out);
    fclose(s_out);
    printf("wrote %u records to %s\n", s_records, path);
    return 0;
}
