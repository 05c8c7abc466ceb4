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

/* the per-triangle records of the shading stages (pt_device.h, DTriRec): the same bits as the PtScene arrays, one line per triangle */
__global__ __launch_bounds__(256) void k_pack_trirecs(DScene s, DTriRec* __restrict__ out)
{
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= s.triCount) return;
    const int32_t iVert = t * 3;
    DTriRec r;
    r.p0 = s.positions[iVert + 0]; r.p1 = s.positions[iVert + 1]; r.p2 = s.positions[iVert + 2];
    r.n0 = s.normals[iVert + 0]; r.n1 = s.normals[iVert + 1]; r.n2 = s.normals[iVert + 2];
    r.uv0 = s.uvs[iVert + 0]; r.uv1 = s.uvs[iVert + 1]; r.uv2 = s.uvs[iVert + 2];
    r.matId = s.matIds[iVert];
    r.flags = s.materials[r.matId].flags;
    out[t] = r;
}

int pack_trirecs(const DScene& s, DTriRec* out, hipStream_t stream)
{
    if (s.triCount <= 0) return PIMGPU_OK;
    k_pack_trirecs<<<(s.triCount + 255) / 256, 256, 0, stream>>>(s, out);
    PT_HIP(hipGetLastError());
    return PIMGPU_OK;
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
__device__ float point_query_distance(const DScene& s, v4 pt, bool* overflow)
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
                if (ex * ex + ey * ey + ez * ez <= prune2)
                {
                    if (sp < PT_STACK_MAX) stack[sp++] = c[k];
                    else *overflow = true;          /* a dropped subtree could hold the nearest triangle: reported, scene_create fails */
                }
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

/*
 * SetupLightGridFn, path_tracer.c:891-990.  The reference gives a cell to one thread, which casts its 16 probe rays and
 * its 16 shadow rays per emissive one after the other; as one lane per cell with inline traversals that was 58 ms for the
 * 300 k cells of the 1 M-triangle scene.  The rays do not depend on each other's results, so they go through the batched
 * traversal kernels instead, a chunk of cells at a time:
 *   k_grid_probe   cell: point query; if the cell centre is not near a surface, its 16 closest-hit probe rays
 *   (closest-hit launch)
 *   k_grid_shadow  cell: culled if fewer than half of the probes saw a front face (the centre is inside a solid);
 *                  else its 16 shadow rays per emissive, drawn in the reference's order from the cell's stream
 *   (any-hit launch)
 *   k_grid_bake    cell: visible fraction per emissive -> pdf, Dist1D_Bake
 * Same rays, same counts: the tables equal the one-lane version's bit for bit.
 */
struct GridArgs
{
    float metersPerCell; uint32_t seed;
    int32_t cellFirst, cellCount;            /* this chunk */
    int32_t* cellLength;
    float4* rayO; float4* rayD;              /* probe rays: [cell][16]; shadow rays: [cell][E][16] */
    DHit* hits; int32_t* occluded;
    uint8_t* state;                          /* per cell of the chunk: 0 culled, 1 needs the probe verdict, 2 alive */
    uint32_t* status;                        /* TraceResources::status: bit 0 = a traversal stack overflowed */
};

__device__ __forceinline__ v4 grid_cell_position(const DScene& s, int32_t iCell, float radius)
{
    v4 position = grid_position(s.grid, iCell);
    position.w = radius + 0.01f * PT_MILLI;
    return position;
}

__global__ __launch_bounds__(64) void k_grid_probe(DScene s, GridArgs g)
{
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.cellCount) return;
    const int32_t iCell = g.cellFirst + c;
    const float radius = g.metersPerCell * 0.666f;
    g.cellLength[iCell] = 0;
    const v4 position = grid_cell_position(s, iCell, radius);
    bool overflow = false;
    const float qd = point_query_distance(s, position, &overflow);
    if (overflow) atomicOr(g.status, 1u);
    const bool probe = qd > radius;
    g.state[c] = probe ? 1 : 2;
    for (int j = 0; j < 16; ++j)
    {
        const v4 dir = SampleUnitSphere(Hammersley2D((uint32_t)j, 16u));
        g.rayO[(size_t)c * 16 + j] = make_float4(position.x, position.y, position.z, 0.0f);
        g.rayD[(size_t)c * 16 + j] = make_float4(dir.x, dir.y, dir.z, probe ? PT_RCP_EPS : -1.0f);     /* tFar < tNear: no ray */
    }
}

__global__ __launch_bounds__(64) void k_grid_shadow(DScene s, GridArgs g)
{
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.cellCount) return;
    const int32_t iCell = g.cellFirst + c;
    const int32_t E = s.emissiveCount;
    const float radius = g.metersPerCell * 0.666f;
    const v4 position = grid_cell_position(s, iCell, radius);
    bool alive = true;
    if (g.state[c] == 1)
    {
        float hitcount = 0.0f;
        for (int j = 0; j < 16; ++j)
        {
            const v4 dir = SampleUnitSphere(Hammersley2D((uint32_t)j, 16u));
            if (hit_type(s, dir, g.hits[(size_t)c * 16 + j]) == PIM_HIT_TRIANGLE) ++hitcount;
        }
        const float hitratio = hitcount / 16;
        alive = !(hitratio < 0.5f);
    }
    g.state[c] = alive ? 2 : 0;
    u4 rng = pcg4((uint32_t)iCell, 1u, g.seed, PT_TAG_GRID);
    for (int32_t iEmit = 0; iEmit < E; ++iEmit)
    {
        const size_t base = ((size_t)c * E + iEmit) * 16;
        if (!alive)
        {
            for (int k = 0; k < 16; ++k) { g.rayO[base + k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); g.rayD[base + k] = make_float4(0.0f, 0.0f, 1.0f, -1.0f); }
            continue;
        }
        const int32_t iVert = s.emitToVert[iEmit];
        const v4 A = s.positions[iVert + 0], B = s.positions[iVert + 1], C = s.positions[iVert + 2];
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
            g.rayO[base + k] = make_float4(ro.x, ro.y, ro.z, 0.0f);
            g.rayD[base + k] = make_float4(rd.x, rd.y, rd.z, tFar);
        }
    }
}

__global__ __launch_bounds__(64) void k_grid_bake(DScene s, GridArgs g)
{
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.cellCount) return;
    const int32_t iCell = g.cellFirst + c;
    const int32_t E = s.emissiveCount;
    if (g.state[c] == 0 || E <= 0) return;
    float* pdf = s.pdf + (size_t)iCell * E;
    float* cdf = s.cdf + (size_t)iCell * (E + 1);
    for (int32_t iEmit = 0; iEmit < E; ++iEmit)
    {
        const size_t base = ((size_t)c * E + iEmit) * 16;
        int32_t hits = 0;
        for (int k = 0; k < 16; ++k) hits += g.occluded[base + k] ? 0 : 1;
        pdf[iEmit] = (float)hits / 16.0f;
    }
    s.cellIntegral[iCell] = dist_bake(pdf, cdf, E);
    g.cellLength[iCell] = E;
}

int setup_light_grid(const DScene& s, TraceResources* res, float metersPerCell, uint32_t seed, int32_t* cellLengthDev, hipStream_t stream)
{
    if (s.gridLen <= 0) return PIMGPU_OK;
    const int64_t E = s.emissiveCount;
    const int64_t raysPerCell = 16 * std::max<int64_t>(E, 1);
    const int64_t kRayBudget = 1ll << 24;                        /* rays in flight per chunk: 512 MB of rays + 268 MB of results */
    const int32_t chunk = (int32_t)std::max<int64_t>(1, std::min<int64_t>(s.gridLen, kRayBudget / raysPerCell));
    const size_t rays = (size_t)chunk * (size_t)raysPerCell;
    float4* rayO = nullptr; float4* rayD = nullptr; DHit* hits = nullptr; uint8_t* state = nullptr;
    auto cleanup = [&]() { hipFree(rayO); hipFree(rayD); hipFree(hits); hipFree(state); };
    if (hipMalloc(&rayO, sizeof(float4) * rays) != hipSuccess || hipMalloc(&rayD, sizeof(float4) * rays) != hipSuccess ||
        hipMalloc(&hits, sizeof(DHit) * rays) != hipSuccess || hipMalloc(&state, (size_t)chunk) != hipSuccess)
    { cleanup(); set_error("light grid: out of device memory"); return PIMGPU_E_NOMEM; }
    int rc = PIMGPU_OK;
    for (int32_t first = 0; first < s.gridLen && rc == PIMGPU_OK; first += chunk)
    {
        GridArgs g;
        g.metersPerCell = metersPerCell; g.seed = seed; g.cellFirst = first; g.cellCount = std::min(chunk, s.gridLen - first);
        g.cellLength = cellLengthDev; g.rayO = rayO; g.rayD = rayD; g.hits = hits; g.occluded = (int32_t*)hits; g.state = state; g.status = res->status;
        const int blocks = (g.cellCount + 63) / 64;
        k_grid_probe<<<blocks, 64, 0, stream>>>(s, g);
        if ((rc = trace_closest(s.bvh, res, rayO, rayD, hits, g.cellCount * 16, nullptr, stream))) break;
        k_grid_shadow<<<blocks, 64, 0, stream>>>(s, g);
        if (E > 0 && (rc = trace_any(s.bvh, res, rayO, rayD, g.occluded, (int32_t)(g.cellCount * raysPerCell), stream))) break;
        k_grid_bake<<<blocks, 64, 0, stream>>>(s, g);
        if (hipGetLastError() != hipSuccess) { set_error("light grid: launch failed"); rc = PIMGPU_E_DEVICE; }
    }
    if (hipStreamSynchronize(stream) != hipSuccess && rc == PIMGPU_OK) { set_error("light grid: device error"); rc = PIMGPU_E_DEVICE; }
    cleanup();
    return rc;
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
