/*
 * pt_scene_fn.h -- device functions of the path tracer that read the scene: light grid and
 * Dist1D, hit classification, surface fetch, BRDFs, light sampling, participating media.
 * Each follows the reference function named in its comment (rendering/path_tracer.c = pt.c).
 * The bounce loop itself (Pt_TraceRay, pt.c:2306-2416) is cut into wavefront stages in
 * integrate.hip; the functions here are the pieces between two ray queries.
 */
#ifndef PT_SCENE_FN_H
#define PT_SCENE_FN_H

#include "pt_device.h"
#include "traverse.h"

namespace pt {

/* ---- Grid, math/grid.h:8-49 ---- */
PT_FN v4 grid_position(const DGrid& g, int32_t index)
{
    const float metersPerCell = 1.0f / g.cellsPerMeter;
    const int32_t ix = index % g.sx;
    const int32_t iy = (index / g.sx) % g.sy;
    const int32_t iz = index / (g.sx * g.sy);
    const v4 offset = f4_v((ix + 0.5f) * metersPerCell, (iy + 0.5f) * metersPerCell, (iz + 0.5f) * metersPerCell, 0.0f);
    return f4_add(g.lo, offset);
}
PT_FN int32_t grid_index(const DGrid& g, v4 position)
{
    const v4 offset = f4_muls(f4_sub(position, g.lo), g.cellsPerMeter);
    const int32_t x = i1_clamp((int32_t)offset.x, 0, g.sx - 1);
    const int32_t y = i1_clamp((int32_t)offset.y, 0, g.sy - 1);
    const int32_t z = i1_clamp((int32_t)offset.z, 0, g.sz - 1);
    return x + y * g.sx + z * g.sx * g.sy;
}

/* ---- Dist1D on raw arrays, math/dist1d.c:33-163 ---- */
PT_FN float dist_bake(float* pdf, float* cdf, int32_t pdfLen)
{
    const int32_t cdfLen = pdfLen + 1;
    if (pdfLen <= 0) return 0.0f;
    const float rcpLen = 1.0f / pdfLen;
    cdf[0] = 0.0f;
    for (int32_t i = 1; i < cdfLen; ++i) cdf[i] = cdf[i - 1] + pdf[i - 1] * rcpLen;
    const float integral = cdf[cdfLen - 1];
    if (integral == 0.0f)
    {
        for (int32_t i = 1; i < cdfLen; ++i) cdf[i] = i * rcpLen;
    }
    else
    {
        const float rcpIntegral = 1.0f / integral;
        for (int32_t i = 1; i < cdfLen; ++i) cdf[i] = cdf[i] * rcpIntegral;
        for (int32_t i = 0; i < pdfLen; ++i) pdf[i] = pdf[i] * rcpIntegral;
    }
    return integral;
}
PT_FN int32_t dist_find_interval(const float* cdf, int32_t size, float u)
{
    int32_t first = 0, len = size;
    while (len > 0)
    {
        const int32_t half = len >> 1;
        const int32_t middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len = len - (half + 1); }
        else len = half;
    }
    return i1_clamp(first - 1, 0, size - 2);
}
PT_FN void dist_update(float* pdf, float* cdf, uint32_t* live, int32_t pdfLen, uint32_t* sumIo, float* integralIo)
{
    if (pdfLen <= 0) return;
    uint32_t sum = 0;
    for (int32_t i = 0; i < pdfLen; ++i) sum += live[i];
    if (sum < 30) return;
    const float scale = 1.0f / sum;
    const uint32_t prevSum = *sumIo;
    *sumIo = sum;
    float alpha = 0.5f;
    if (prevSum > 0)
    {
        const double ratio = (double)sum / (double)prevSum;
        alpha = f1_sat((float)ratio) * 0.9f;
        alpha = alpha * alpha;
    }
    for (int32_t i = 0; i < pdfLen; ++i)
    {
        const uint32_t ct = live[i];
        pdf[i] = f1_lerp(pdf[i], ct * scale, alpha);
        live[i] = (ct >> 1);
    }
    *integralIo = dist_bake(pdf, cdf, pdfLen);
}

/* ---- hits ---- */

/* PtRayHit in registers (rendering/path_tracer.h:24-31) */
struct RayHit { v4 wuvt; v4 normal; int32_t type; int32_t iVert; uint32_t flags; };

/* pt_intersect_local after the query, pt.c:1434-1463 */
__device__ __forceinline__ RayHit make_rayhit(const DScene& s, v4 rd, DHit h)
{
    RayHit hit;
    hit.wuvt = f4_v(0.0f, 0.0f, 0.0f, -1.0f);
    hit.normal = f4_s(0.0f);
    hit.type = PIM_HIT_NOTHING; hit.iVert = -1; hit.flags = 0;
    if (h.prim < 0 || h.t <= 0.0f) return hit;
    const DTriRec* r = s.tri + h.prim;
    const v4 p0 = r->p0;
    v4 e1 = f4_sub(r->p1, p0), e2 = f4_sub(r->p2, p0);
    e1.w = e2.w = 0.0f;
    v4 Ng = f4_cross3(e1, e2);
    hit.type = PIM_HIT_TRIANGLE;
    if (f4_dot3(Ng, rd) > 0.0f) { Ng = f4_neg(Ng); hit.type = PIM_HIT_BACKFACE; }
    hit.normal = f4_normalize3(Ng);
    hit.normal.w = 0.0f;
    const float u = f1_sat(h.u), v = f1_sat(h.v), w = f1_sat(1.0f - (u + v));
    hit.iVert = h.prim * 3;
    hit.wuvt = f4_v(w, u, v, h.t);
    hit.flags = r->flags;
    return hit;
}
__device__ __forceinline__ int32_t hit_type(const DScene& s, v4 rd, DHit h) { return make_rayhit(s, rd, h).type; }

/* per-path context: the RNG stream (PtContext, pt.c:168-176) + ray statistics.
 *
 * FAST / SLOW builds.  A few rarely taken parts of a stage need many registers and a traversal stack: the
 * medium's collision loop (with its in-scatter light ray), the transmittance loop, refraction (thickness ray).
 * A kernel's register allocation is the maximum over everything it can reach, so keeping them in the same
 * kernel would halve the occupancy of the common case.  Every stage is therefore built twice from one
 * template: the FAST build (SLOW = false) only DETECTS the rare case -- before any side effect -- sets
 * ctx.defer and the kernel drops the path's stage unfinished into a small "slow" list; the SLOW build runs
 * the complete code on that list from the same saved state, redrawing the same random numbers. */
struct Ctx { u4 rng; uint32_t shadowRays, otherRays; bool overflow; bool defer; LdsStack stack; /* SLOW builds: the calling kernel's traversal stack */ };
__device__ __forceinline__ float Sample1D(Ctx& c) { return rng_f32(&c.rng); }      /* pt.c:377 */
__device__ __forceinline__ v2 Sample2D(Ctx& c) { return rng_f2(&c.rng); }          /* pt.c:382 */
__device__ __forceinline__ Ctx ctx_make() { Ctx c; c.rng.x = c.rng.y = c.rng.z = c.rng.w = 0; c.shadowRays = c.otherRays = 0; c.overflow = false; c.defer = false; c.stack.lds = nullptr; c.stack.spill = nullptr; c.stack.ldsStride = 0; c.stack.spillStride = 0; c.stack.sp = 0; c.stack.cap = PT_STACK_MAX; return c; }

/* a closest-hit query traced inline by the calling lane (rare rays: media NEE, refraction thickness; SLOW builds only) */
__device__ __forceinline__ RayHit intersect_inline(Ctx& ctx, const DScene& s, v4 ro, v4 rd, float tNear, float tFar)
{
    /* the kernel's LDS + spill stack (k_shade_a / k_shade_b SLOW builds, k_tail): a pop from a private-memory stack is a
     * memory round trip on this lane's dependent chain */
    const DHit h = traverse<false>(s.bvh, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, tNear, tFar, ctx.stack, ctx.overflow);
    return make_rayhit(s, rd, h);
}

/* ---- surface fetch, pt.c:1180-1419 ---- */

struct SurfHit
{
    v4 P, M, N, albedo, emission, meanFreePath;
    float roughness, occlusion, metallic, ior;
    uint32_t flags;
    int32_t type;
};
struct Scatter { v4 pos, dir, attenuation, luminance; float pdf; };
struct LightSample { v4 direction, luminance, wuvt; float pdf; };
struct Media { v4 scattering, extinction; float g0, g1, gBlend; };

__device__ __forceinline__ v4 GetSky(const DScene& s, v4 rd)                       /* pt.c:1280-1291, cubemap.h:117-129 */
{
    if (s.skySize > 0)
    {
        v2 uv; const int face = Cubemap_CalcUv(rd, &uv);
        const v3 c = UvBilinearClamp_f3(s.sky[face], s.skySize, s.skySize, uv);
        return f4_v(c.x, c.y, c.z, 1.0f);
    }
    return f4_s(0.0f);
}
/* the hit triangle's record (pt_device.h): iVert == 3 * primID everywhere */
__device__ __forceinline__ const DTriRec* tri_of(const DScene& s, int32_t iVert) { return s.tri + (uint32_t)iVert / 3u; }
__device__ __forceinline__ v2 GetUV(const DScene& s, const RayHit& hit) { const DTriRec* r = tri_of(s, hit.iVert); return f2_blend(r->uv0, r->uv1, r->uv2, hit.wuvt); }
__device__ __forceinline__ v4 GetPosition(const DScene& s, const RayHit& hit) { const DTriRec* r = tri_of(s, hit.iVert); return f4_blend(r->p0, r->p1, r->p2, hit.wuvt); }
__device__ __forceinline__ v4 GetNormal(const DScene& s, const RayHit& hit)          /* pt.c:1192-1204 */
{
    const DTriRec* r = tri_of(s, hit.iVert);
    v4 N = f4_blend(r->n0, r->n1, r->n2, hit.wuvt);
    N = (f4_dot3(hit.normal, N) > 0.0f) ? N : f4_neg(N);
    return f4_normalize3(N);
}
__device__ __forceinline__ float GetArea(const DScene& s, int32_t iVert) { const DTriRec* r = tri_of(s, iVert); return TriArea3D(r->p0, r->p1, r->p2); }

__device__ __forceinline__ v4 tex_c32(const DScene& s, int32_t id, v2 uv, v4 dflt)
{
    if (id < 0) return dflt;
    const DTexture t = s.textures[id];
    return UvBilinearWrap_c32(t.texels, t.width, t.height, uv);
}

__device__ __forceinline__ v4 GetEmission(const DScene& s, v4 rd, const RayHit& hit)   /* pt.c:1293-1326 */
{
    if (hit.flags & PIM_MATFLAG_SKY) return GetSky(s, rd);
    const PimMaterial* mat = &s.materials[tri_of(s, hit.iVert)->matId];
    const int32_t albedoId = mat->albedo, romeId = mat->rome;
    const v2 uv = GetUV(s, hit);
    const v4 albedo = tex_c32(s, albedoId, uv, f4_s(1.0f));
    float e = 0.0f;
    if (romeId >= 0) e = tex_c32(s, romeId, uv, f4_s(0.0f)).w;
    return UnpackEmission(albedo, e);
}

__device__ __forceinline__ v4 FixShadingNormal(v4 M, v4 N) { return (f4_dot3(M, N) > 0.0f) ? N : f4_reflect3(N, M); }   /* pt.c:1356 */

__device__ __forceinline__ SurfHit GetSurface(const DScene& s, v4 rd, const RayHit& hit)   /* pt.c:1377-1419 */
{
    SurfHit surf;
    surf.type = hit.type;
    const PimMaterial mat = s.materials[tri_of(s, hit.iVert)->matId];
    surf.flags = mat.flags;
    surf.ior = mat.ior;
    const v2 uv = GetUV(s, hit);
    surf.M = GetNormal(s, hit);
    surf.N = surf.M;
    surf.P = GetPosition(s, hit);
    surf.P = f4_add(surf.P, f4_muls(surf.M, 0.01f * PT_MILLI));
    surf.meanFreePath = f4_v(mat.meanFreePath[0], mat.meanFreePath[1], mat.meanFreePath[2], mat.meanFreePath[3]);
    if (hit.flags & PIM_MATFLAG_SKY)
    {
        surf.albedo = f4_s(0.0f);
        surf.roughness = 1.0f; surf.occlusion = 0.0f; surf.metallic = 0.0f;
        surf.emission = GetSky(s, rd);
        surf.M = f4_neg(rd);
        surf.N = surf.M;
        surf.ior = 1.0f;
    }
    else
    {
        if (mat.normal >= 0)                  /* SampleNormal pt.c:1363-1375 */
        {
            const DTexture t = s.textures[mat.normal];
            const v4 Nts = UvBilinearWrap_xy16(t.texels, t.width, t.height, uv);
            surf.N = FixShadingNormal(surf.N, TanToWorld(surf.N, Nts));
        }
        surf.albedo = tex_c32(s, mat.albedo, uv, f4_s(1.0f));                      /* SampleAlbedo pt.c:1328 */
        const v4 rome = tex_c32(s, mat.rome, uv, f4_v(0.5f, 1.0f, 0.0f, 0.0f));    /* SampleRome pt.c:1341 */
        surf.emission = UnpackEmission(surf.albedo, rome.w);
        surf.roughness = rome.x; surf.occlusion = rome.y; surf.metallic = rome.z;
    }
    return surf;
}

/* ---- BRDF evaluation, pt.c:1476-1707 ---- */

__device__ __forceinline__ v4 GGXEnergyCompensation(const DScene& s, v4 f0, float NoV, float alpha)   /* lighting.h:294-303 */
{
    const v2 dvf_dv = UvBilinearClamp_f2(s.lut, s.lutSize, s.lutSize, f2_v(NoV, alpha));
    const float t = (1.0f / f1_max(dvf_dv.y, PT_EPS)) - 1.0f;
    return f4_adds(f4_muls(f0, t), 1.0f);
}

__device__ __forceinline__ v4 Eval_Diffuse(const SurfHit& surf, v4 I, v4 L)          /* pt.c:1476-1497 */
{
    v4 result = f4_s(0.0f);
    const v4 N = surf.N;
    const float NoL = f4_dot3(N, L);
    const float pdf = LambertPdf(NoL);
    if (pdf > PT_EPS)
    {
        const v4 V = f4_neg(I);
        const v4 H = f4_normalize3(f4_add(V, L));
        const float HoV = f4_dotsat(H, V);
        const float NoV = f4_dotsat(N, V);
        const v4 Fd = f4_muls(surf.albedo, Fd_Burley(NoL, NoV, HoV, surf.roughness));
        result = f4_muls(Fd, NoL);
        result.w = pdf;
    }
    return result;
}

__device__ __forceinline__ v4 Eval_Specular(const DScene& s, const SurfHit& surf, v4 I, v4 L)   /* pt.c:1516-1548 */
{
    v4 result = f4_s(0.0f);
    const v4 N = surf.N;
    const float NoL = f4_dot3(N, L);
    if (NoL > PT_EPS)
    {
        const float alpha = BrdfAlpha(surf.roughness);
        const v4 V = f4_neg(I);
        const v4 H = f4_normalize3(f4_add(V, L));
        const float NoH = f4_dot3(N, H);
        const float HoV = f4_dot3(H, V);
        const float pdf = GGXPdf(NoH, HoV, alpha);
        if (pdf > PT_EPS)
        {
            const float NoV = f4_dotsat(N, V);
            const float f = F_Dielectric(HoV, 1.0f, 1.5f);
            const v4 f0 = F_0(surf.albedo, surf.metallic);
            const v4 F = f4_lerps(f0, f4_s(F_90(f0)), f);
            const float D = D_GTR(NoH, alpha);
            const float G = V_SmithCorrelated(NoL, NoV, alpha);
            v4 Fr = f4_muls(F, D * G);
            Fr = f4_mul(Fr, GGXEnergyCompensation(s, f0, NoV, alpha));
            result = f4_muls(Fr, NoL);
            result.w = pdf;
        }
    }
    return result;
}

__device__ __forceinline__ Scatter Scatter_Diffuse(Ctx& ctx, const SurfHit& surf, v4 I)   /* pt.c:1499-1514 */
{
    Scatter r;
    r.luminance = f4_s(0.0f);
    const v4 L = TanToWorld(surf.N, SampleCosineHemisphere(Sample2D(ctx)));
    const v4 attenPdf = Eval_Diffuse(surf, I, L);
    r.dir = L; r.attenuation = attenPdf; r.pdf = attenPdf.w; r.pos = surf.P;
    return r;
}

__device__ __forceinline__ Scatter Scatter_Specular(Ctx& ctx, const DScene& s, const SurfHit& surf, v4 I)   /* pt.c:1550-1565 */
{
    Scatter r;
    r.luminance = f4_s(0.0f);
    v4 m = SampleGGXMicrofacet(Sample2D(ctx), BrdfAlpha(surf.roughness));
    m = TanToWorld(surf.N, m);
    m = FixShadingNormal(surf.M, m);
    r.dir = f4_reflect3(I, m);
    r.attenuation = Eval_Specular(s, surf, I, r.dir);
    r.pdf = r.attenuation.w;
    r.pos = surf.P;
    return r;
}

__device__ __forceinline__ Scatter Scatter_Refractive(Ctx& ctx, const DScene& s, const SurfHit& surf, v4 I)   /* pt.c:1576-1638 */
{
    const float etaI = 1.000277f;
    const float etaT = f1_max(1.0f, surf.ior);
    const float alpha = BrdfAlpha(surf.roughness);
    const v4 V = f4_neg(I);
    const v4 M = surf.M;
    v4 m = TanToWorld(surf.N, SampleGGXMicrofacet(Sample2D(ctx), alpha));
    m = FixShadingNormal(M, m);
    const bool entering = surf.type != PIM_HIT_BACKFACE;
    const float cosThetaI = f1_sat(f1_abs(f4_dot3(V, m)));
    float pdf = F_Dielectric(entering ? cosThetaI : -cosThetaI, etaI, etaT);
    v4 L;
    bool refract = false;
    if (Sample1D(ctx) < pdf)
    {
        L = f4_reflect3(I, m);
    }
    else
    {
        const float k = entering ? etaI / etaT : etaT / etaI;
        L = f4_refract3(I, m, k);
        pdf = 1.0f - pdf;
        refract = true;
    }
    v4 P = surf.P;
    if (f4_dot3(L, M) < 0.0f) P = f4_sub(P, f4_muls(M, PT_MILLI * 0.1f));
    Scatter r;
    r.luminance = f4_s(0.0f);
    r.pos = P; r.dir = L; r.pdf = pdf;
    if (refract && entering)
    {
        ctx.otherRays++;
        const RayHit hit = intersect_inline(ctx, s, P, L, 0.0f, PT_RCP_EPS);
        float thickness = f1_max(hit.wuvt.w, PT_EPS);
        if (hit.type == PIM_HIT_NOTHING) thickness = PT_RCP_EPS;
        const v4 tr = AlbedoToTransmittance(surf.albedo, surf.roughness, thickness);
        r.attenuation = f4_muls(tr, pdf);
    }
    else
    {
        r.attenuation = f4_s(pdf);
    }
    return r;
}

__device__ __forceinline__ v4 Eval_Principled(const DScene& s, const SurfHit& surf, v4 I, v4 L)   /* pt.c:1641-1668 */
{
    if (surf.flags & PIM_MATFLAG_REFRACTIVE) return f4_s(0.0f);
    if (!(f4_dot3(surf.N, L) > PT_EPS)) return f4_s(0.0f);
    const float amtA = f1_lerp(0.5f, 1.0f, surf.metallic);
    const float amtB = 1.0f - amtA;
    v4 evalA = Eval_Specular(s, surf, I, L);
    if (amtB > PT_EPS)
    {
        const v4 evalB = Eval_Diffuse(surf, I, L);
        evalA = f4_lerps(evalA, evalB, amtB);
    }
    return evalA;
}

template <bool SLOW>
__device__ __forceinline__ Scatter Scatter_Principled(Ctx& ctx, const DScene& s, const SurfHit& surf, v4 I)   /* pt.c:1670-1707 */
{
    if (surf.flags & PIM_MATFLAG_REFRACTIVE)
    {
        if constexpr (SLOW) return Scatter_Refractive(ctx, s, surf, I);
        else
        {
            ctx.defer = true;
            Scatter none; none.pos = none.dir = none.attenuation = none.luminance = f4_s(0.0f); none.pdf = 0.0f;
            return none;
        }
    }
    Scatter scatter;
    const float amtSpec = f1_lerp(0.5f, 1.0f, surf.metallic);
    const float amtDiffuse = 1.0f - amtSpec;
    if (Sample1D(ctx) < amtSpec)
    {
        scatter = Scatter_Specular(ctx, s, surf, I);
        if (amtDiffuse > PT_EPS)
        {
            const v4 eval = Eval_Diffuse(surf, I, scatter.dir);
            scatter.attenuation = f4_lerps(scatter.attenuation, eval, amtDiffuse);
            scatter.pdf = f1_lerp(scatter.pdf, eval.w, amtDiffuse);
        }
    }
    else
    {
        scatter = Scatter_Diffuse(ctx, surf, I);
        const v4 eval = Eval_Specular(s, surf, I, scatter.dir);
        scatter.attenuation = f4_lerps(scatter.attenuation, eval, amtSpec);
        scatter.pdf = f1_lerp(scatter.pdf, eval.w, amtSpec);
    }
    return scatter;
}

/* ---- light sampling, pt.c:1709-1942 ---- */

__device__ __forceinline__ void LightOnHit(const DScene& s, v4 ro, v4 lum4, int32_t iVert)   /* pt.c:1709-1733 */
{
    const float lum = f4_avglum(lum4);
    if (lum <= PT_EPS) return;
    float loglum = pl_log2f(lum) - PT_LOG2_EPS;
    loglum = f1_clamp(loglum, 0.0f, 46.0f);
    const uint32_t amt = (uint32_t)(loglum * (0xff / 46.0f) + 0.5f);
    const int32_t iGrid = grid_index(s.grid, ro);
    const int32_t iEmit = s.vertToEmit[iVert];
    if (iEmit >= 0 && s.cellLength[iGrid])
    {
        atomicAdd(&s.live[(size_t)iGrid * s.emissiveCount + iEmit], amt);
    }
}

__device__ __forceinline__ float Dist_PdfD(const DScene& s, int32_t iCell, int32_t iEmit) { return s.pdf[(size_t)iCell * s.emissiveCount + iEmit] / s.emissiveCount; }   /* dist1d.c:118 */

__device__ __forceinline__ bool LightSelect(Ctx& ctx, const DScene& s, v4 position, int32_t* iVertOut, float* pdfOut)   /* pt.c:1735-1764 */
{
    *iVertOut = -1; *pdfOut = 0.0f;
    if (s.emissiveCount == 0) return false;
    const int32_t iCell = grid_index(s.grid, position);
    if (!s.cellLength[iCell]) return false;
    const int32_t E = s.emissiveCount;
    const int32_t iEmit = dist_find_interval(s.cdf + (size_t)iCell * (E + 1), E + 1, Sample1D(ctx));
    const float pdf = Dist_PdfD(s, iCell, iEmit);
    *iVertOut = s.emitToVert[iEmit];
    *pdfOut = pdf;
    return pdf > PT_EPS;
}

__device__ __forceinline__ float LightSelectPdf(const DScene& s, int32_t iVert, v4 ro)   /* pt.c:1766-1783 */
{
    float selectPdf = 1.0f;
    const int32_t iGrid = grid_index(s.grid, ro);
    const int32_t iEmit = s.vertToEmit[iVert];
    if (iEmit >= 0 && s.cellLength[iGrid]) selectPdf = Dist_PdfD(s, iGrid, iEmit);
    return selectPdf;
}

__device__ __forceinline__ Media Media_Sample(const DMedia& desc, v4 P)               /* pt.c:2146-2181 */
{
    Media hit;
    hit.scattering = desc.constantMu;
    hit.g0 = desc.phaseDirA; hit.g1 = desc.phaseDirB; hit.gBlend = desc.phaseBlend;
    if (f1_distance(P.y, desc.noiseHeight) <= desc.noiseRange)
    {
        const float noise = FbmGradientNoise3(f4_muls(P, desc.noiseFreq), desc.noiseLacunarity, desc.noiseGain, desc.noiseOctaves, 1);
        const float noiseScale = desc.noiseScale;
        const float height = desc.noiseHeight + noiseScale * noise;
        const float dist = f1_distance(P.y, height) / noiseScale;
        const float heightDensity = f1_sat(1.0f - dist);
        hit.scattering = f4_add(hit.scattering, f4_muls(desc.noiseMu, heightDensity));
    }
    hit.extinction = f4_muls(hit.scattering, 1.0f + desc.absorption);
    return hit;
}

__device__ __forceinline__ float CalcPhase(const Media& m, float cosTheta) { return f1_lerp(MiePhase(cosTheta, m.g0), MiePhase(cosTheta, m.g1), m.gBlend); }   /* pt.c:2198 */

__device__ __forceinline__ v4 SamplePhaseDir(Ctx& ctx, const Media& media, v4 rd)     /* pt.c:2208-2221 */
{
    v4 L;
    do
    {
        L = SampleUnitSphere(Sample2D(ctx));
        L.w = CalcPhase(media, f4_dot3(rd, L));
    } while (Sample1D(ctx) > L.w);
    return L;
}

/* CalcTransmittance, pt.c:2223-2249.  The first free-path sample ends the loop for all but a few rays in a thin
 * medium (default mean free path 40 km); the FAST build defers the others. */
template <bool SLOW>
__device__ __forceinline__ v4 CalcTransmittance(Ctx& ctx, const DScene& s, v4 ro, v4 rd, float rayLen)
{
    const float rcpMaj = s.media.rcpMajorant;
    float t = 0.0f;
    v4 attenuation = f4_s(1.0f);
    while (true)
    {
        const float dt = SampleFreePath(Sample1D(ctx), rcpMaj);
        if ((t + dt) >= rayLen) break;
        if constexpr (!SLOW) { ctx.defer = true; break; }
        else
        {
            const Media media = Media_Sample(s.media, f4_add(ro, f4_muls(rd, t)));
            const v4 ratio = f4_inv(f4_muls(media.extinction, rcpMaj));
            attenuation = f4_mul(attenuation, ratio);
            t += dt;
        }
    }
    return attenuation;
}

/* SampleLight up to its ray, pt.c:1785-1812: the point on the emissive triangle and the ray towards it */
struct LightRay { v4 rd; float distSq, distance; };
__device__ __forceinline__ LightRay SampleLight_ray(Ctx& ctx, const DScene& s, v4 ro, int32_t iVert)
{
    LightRay lr;
    const v4 wuv = SampleBaryCoord(Sample2D(ctx));
    const DTriRec* lt = tri_of(s, iVert);
    const v4 A = lt->p0, B = lt->p1, C = lt->p2;
    const v4 pt = f4_blend(A, B, C, wuv);
    v4 rd = f4_sub(pt, ro);
    lr.distSq = f4_dot3(rd, rd);
    lr.distance = pt_sqrt(f1_max(lr.distSq, PT_EPS_SQ));
    lr.rd = f4_divs(rd, lr.distance);
    return lr;
}
/* SampleLight after its ray, pt.c:1813-1826 */
template <bool SLOW>
__device__ __forceinline__ LightSample SampleLight_finish(Ctx& ctx, const DScene& s, v4 ro, int32_t iVert, v4 rd, float distSq, const RayHit& hit)
{
    LightSample sample;
    sample.direction = rd; sample.luminance = f4_s(0.0f); sample.wuvt = f4_s(0.0f); sample.pdf = 0.0f;
    if ((hit.type != PIM_HIT_NOTHING) && (hit.iVert == iVert))
    {
        const float cosTheta = f1_abs(f4_dot3(rd, hit.normal));
        const float area = GetArea(s, iVert);
        sample.pdf = LightPdf(area, cosTheta, distSq);
        sample.luminance = GetEmission(s, rd, hit);
        if (f4_sum3(sample.luminance) > PT_EPS)
        {
            const v4 Tr = CalcTransmittance<SLOW>(ctx, s, ro, rd, hit.wuvt.w);
            sample.luminance = f4_mul(sample.luminance, Tr);
        }
    }
    return sample;
}

/* EvaluateLight, pt.c:1921-1942 (media in-scatter; its light ray is traced inline) */
__device__ __forceinline__ bool EvaluateLight(Ctx& ctx, const DScene& s, v4 P, v4* lightOut, v4* dirOut)
{
    int32_t iVert; float selectPdf;
    if (LightSelect(ctx, s, P, &iVert, &selectPdf))
    {
        const LightRay lr = SampleLight_ray(ctx, s, P, iVert);
        ctx.otherRays++;
        const RayHit hit = intersect_inline(ctx, s, P, lr.rd, 0.0f, lr.distance + 0.01f * PT_MILLI);
        const LightSample sample = SampleLight_finish<true>(ctx, s, P, iVert, lr.rd, lr.distSq, hit);
        if (sample.pdf > PT_EPS)
        {
            *lightOut = f4_divs(sample.luminance, sample.pdf * selectPdf);
            *dirOut = sample.direction;
            return true;
        }
    }
    return false;
}

/* ScatterRay, pt.c:2251-2304.  As with CalcTransmittance the first free-path sample usually reaches the surface;
 * the FAST build defers the paths that collide with the medium. */
template <bool SLOW>
__device__ __forceinline__ Scatter ScatterRay(Ctx& ctx, const DScene& s, v4 ro, v4 rd, float rayLen)
{
    Scatter result;
    result.pos = f4_s(0.0f); result.dir = f4_s(0.0f); result.luminance = f4_s(0.0f);
    result.pdf = 0.0f;
    const float rcpMaj = s.media.rcpMajorant;
    float t = 0.0f;
    result.attenuation = f4_s(1.0f);
    while (true)
    {
        const float dt = SampleFreePath(Sample1D(ctx), rcpMaj);
        t += dt;
        if (t >= rayLen) break;
        if constexpr (!SLOW) { ctx.defer = true; break; }
        else
        {
            const v4 P = f4_add(ro, f4_muls(rd, t));
            const Media media = Media_Sample(s.media, P);
            const v4 ratio = f4_inv(f4_muls(media.extinction, rcpMaj));
            result.attenuation = f4_mul(result.attenuation, ratio);
            const float scatterProb = f4_hmax3(media.scattering) * rcpMaj;
            if (Sample1D(ctx) < scatterProb)
            {
                v4 lum, L;
                if (EvaluateLight(ctx, s, P, &lum, &L))
                {
                    const float ph = CalcPhase(media, f4_dot3(rd, L));
                    lum = f4_muls(lum, ph * dt);
                    lum = f4_mul(result.attenuation, lum);
                    result.luminance = lum;
                }
                result.pos = P;
                result.dir = SamplePhaseDir(ctx, media, rd);
                result.pdf = result.dir.w;
                const float ph = CalcPhase(media, f4_dot3(rd, result.dir));
                result.attenuation = f4_muls(result.attenuation, ph);
                break;
            }
        }
    }
    return result;
}

} /* namespace pt */

#endif
