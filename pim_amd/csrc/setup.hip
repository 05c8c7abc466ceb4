/*
 * setup.hip -- scene set-up kernels: SetupEmissives (rendering/path_tracer.c:784-883),
 * SetupLightGrid (:885-1009) and UpdateDists (:2460-2483, math/dist1d.c:33-163).
 *
 * One lane per triangle (emissive detection) or per light-grid cell (everything else); the
 * per-cell Dist1D arrays are flat [cell][emissive] tables in HBM (pdf, cdf with E+1 entries, live).
 */
#include "pt_host.h"
#include "traverse.h"
#include "pt_scene_fn.h"

namespace pt {

/* EmissionPdf, path_tracer.c:784-822: fraction of `attempts` random points of the triangle whose
 * ROME texel has a non-zero emission byte */
__global__ __launch_bounds__(256) void k_emission_pdf(DScene s, int32_t attempts, uint32_t seed, float* __restrict__ pdfs)
{
    const int32_t iTri = blockIdx.x * blockDim.x + threadIdx.x;
    if (iTri >= s.triCount) return;
    const int32_t iVert = iTri * 3;
    const PimMaterial mat = s.materials[s.matIds[iVert]];
    float pdf = 0.0f;
    if (mat.flags & PIM_MATFLAG_SKY) pdf = 1.0f;
    else if (mat.rome >= 0)
    {
        const DTexture rome = s.textures[mat.rome];
        const v2 UA = s.uvs[iVert + 0], UB = s.uvs[iVert + 1], UC = s.uvs[iVert + 2];
        u4 rng = pcg4((uint32_t)iTri, 0u, seed, PT_TAG_EMIT);
        int32_t hits = 0;
        for (int32_t i = 0; i < attempts; ++i)
        {
            const v4 wuv = SampleBaryCoord(rng_f2(&rng));
            const v2 uv = f2_blend(UA, UB, UC, wuv);
            const int32_t cx = PointWrap1D(rome.width, uv.x), cy = PointWrap1D(rome.height, uv.y);
            const uint32_t sample = rome.texels[rome.width * cy + cx] >> 24;
            hits += sample ? 1 : 0;
        }
        pdf = (float)hits / (float)attempts;
    }
    pdfs[iTri] = pdf;
}

int setup_emission_pdfs(const DScene& s, int32_t attempts, uint32_t seed, float* pdfsDev, hipStream_t stream)
{
    if (s.triCount <= 0) return PIMGPU_OK;
    k_emission_pdf<<<(s.triCount + 255) / 256, 256, 0, stream>>>(s, attempts, seed, pdfsDev);
    PT_HIP(hipGetLastError());
    return PIMGPU_OK;
}

/* RtcPointQuery (path_tracer.c:555-616): distance to the nearest triangle inside the query radius,
 * kRcpEpsilon when there is none.  BVH walk with a fixed, slightly enlarged pruning radius; the
 * minimum over the surviving triangles does not depend on the visiting order. */
__device__ float point_query_distance(const DScene& s, v4 pt)
{
    const float radius = pt.w;
    const float prune = radius * 1.001f + 1e-6f;
    const float prune2 = prune * prune;
    float best = PT_RCP_EPS;
    int32_t stack[PT_STACK_MAX]; int sp = 0;
    int32_t cur = 0;
    for (;;)
    {
        if (cur >= 0)
        {
            const float4* q = s.bvh.nodes + (size_t)cur * 8;
            const float4 lox = q[0], loy = q[1], loz = q[2], hix = q[3], hiy = q[4], hiz = q[5], ch = q[6];
            const float lx[4] = { lox.x, lox.y, lox.z, lox.w }, ly[4] = { loy.x, loy.y, loy.z, loy.w }, lz[4] = { loz.x, loz.y, loz.z, loz.w };
            const float hx[4] = { hix.x, hix.y, hix.z, hix.w }, hy[4] = { hiy.x, hiy.y, hiy.z, hiy.w }, hz[4] = { hiz.x, hiz.y, hiz.z, hiz.w };
            const int32_t c[4] = { __float_as_int(ch.x), __float_as_int(ch.y), __float_as_int(ch.z), __float_as_int(ch.w) };
            for (int k = 0; k < 4; ++k)
            {
                if (c[k] == PT_BVH_EMPTY) continue;
                const float ex = fmaxf(fmaxf(lx[k] - pt.x, pt.x - hx[k]), 0.0f);
                const float ey = fmaxf(fmaxf(ly[k] - pt.y, pt.y - hy[k]), 0.0f);
                const float ez = fmaxf(fmaxf(lz[k] - pt.z, pt.z - hz[k]), 0.0f);
                if (ex * ex + ey * ey + ez * ez <= prune2 && sp < PT_STACK_MAX) stack[sp++] = c[k];
            }
        }
        else
        {
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k)
            {
                const float4* tp = s.bvh.tris + (size_t)(first + k) * 3;
                const float4 a = tp[0], b = tp[1], c = tp[2];
                const float d = f1_abs(sdTriangle3D(f4_v(a.x, a.y, a.z, 1.0f), f4_v(b.x, b.y, b.z, 1.0f), f4_v(c.x, c.y, c.z, 1.0f), pt));
                if (d < radius && d < best) best = d;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return best;
}

/* SetupLightGridFn, path_tracer.c:891-990 */
__global__ __launch_bounds__(64) void k_light_grid(DScene s, float metersPerCell, uint32_t seed, int32_t* __restrict__ cellLength,
    uint32_t* __restrict__ status)
{
    const int32_t iCell = blockIdx.x * blockDim.x + threadIdx.x;
    if (iCell >= s.gridLen) return;
    const int32_t E = s.emissiveCount;
    const float radius = metersPerCell * 0.666f;
    u4 rng = pcg4((uint32_t)iCell, 1u, seed, PT_TAG_GRID);
    cellLength[iCell] = 0;
    v4 position = grid_position(s.grid, iCell);
    position.w = radius + 0.01f * PT_MILLI;
    TrivialStack st; bool overflow = false;
    const float qd = point_query_distance(s, position);
    if (qd > radius)
    {
        float hitcount = 0.0f;
        for (int j = 0; j < 16; ++j)
        {
            const v4 dir = SampleUnitSphere(Hammersley2D((uint32_t)j, 16u));
            const DHit h = traverse<false>(s.bvh, position.x, position.y, position.z, dir.x, dir.y, dir.z, 0.0f, PT_RCP_EPS, st, overflow);
            if (hit_type(s, dir, h) == PIM_HIT_TRIANGLE) ++hitcount;
        }
        const float hitratio = hitcount / 16;
        if (hitratio < 0.5f) { if (overflow) atomicOr(status, 1u); return; }
    }
    if (E <= 0) return;
    float* pdf = s.pdf + (size_t)iCell * E;
    float* cdf = s.cdf + (size_t)iCell * (E + 1);
    for (int32_t iEmit = 0; iEmit < E; ++iEmit)
    {
        const int32_t iVert = s.emitToVert[iEmit];
        const v4 A = s.positions[iVert + 0], B = s.positions[iVert + 1], C = s.positions[iVert + 2];
        int32_t hits = 0;
        for (int k = 0; k < 16; ++k)
        {
            const v4 t = f4_muls(f4_lerpsv(-1.5f, 1.5f, rng_f4(&rng)), radius);
            v4 ro = f4_add(position, t);
            ro.w = 0.0f;
            const v4 at = f4_blend(A, B, C, SampleBaryCoord(rng_f2(&rng)));
            v4 rd = f4_sub(at, ro);
            const float dist = f4_length3(rd);
            rd = f4_divs(rd, dist);
            const float tFar = dist - 0.01f * PT_MILLI;
            const DHit h = traverse<true>(s.bvh, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, 0.0f, tFar, st, overflow);
            hits += (h.prim < 0) ? 1 : 0;
        }
        pdf[iEmit] = (float)hits / 16.0f;
    }
    s.cellIntegral[iCell] = dist_bake(pdf, cdf, E);
    cellLength[iCell] = E;
    if (overflow) atomicOr(status, 1u);
}

int setup_light_grid(const DScene& s, float metersPerCell, uint32_t seed, int32_t* cellLengthDev, uint32_t* statusDev, hipStream_t stream)
{
    if (s.gridLen <= 0) return PIMGPU_OK;
    k_light_grid<<<(s.gridLen + 63) / 64, 64, 0, stream>>>(s, metersPerCell, seed, cellLengthDev, statusDev);
    PT_HIP(hipGetLastError());
    return PIMGPU_OK;
}

/* UpdateDistsFn, path_tracer.c:2460-2469 */
__global__ __launch_bounds__(256) void k_update_dists(DScene s)
{
    const int32_t iCell = blockIdx.x * blockDim.x + threadIdx.x;
    if (iCell >= s.gridLen) return;
    if (!s.cellLength[iCell]) return;
    const int32_t E = s.emissiveCount;
    dist_update(s.pdf + (size_t)iCell * E, s.cdf + (size_t)iCell * (E + 1), s.live + (size_t)iCell * E, E,
        &s.cellSum[iCell], &s.cellIntegral[iCell]);
}

int update_dists(const DScene& s, hipStream_t stream)
{
    if (s.gridLen <= 0 || s.emissiveCount <= 0) return PIMGPU_OK;
    k_update_dists<<<(s.gridLen + 255) / 256, 256, 0, stream>>>(s);
    PT_HIP(hipGetLastError());
    return PIMGPU_OK;
}

} /* namespace pt */
