/*
 * prepare.hip -- the two producers of scene inputs that sit directly in front of PtScene_Init:
 *
 *   pimgpu_flatten         FlattenDrawables (rendering/path_tracer.c:692-782): entity meshes -> the
 *                          world-space PtScene vertex arrays
 *   pimgpu_brdf_lut_bake   BakeBrdfFn / BrdfLut_New / BrdfLut_Update (math/lighting.c:31-118): the
 *                          Monte-Carlo GGX energy table read by GGXEnergyCompensation
 *
 * Both are HBM-bound streaming kernels (one output vertex / one LUT texel per lane).
 */
#include "pt_host.h"
#include "pt_api.h"

#include <vector>
#include <cstring>
#include <cstdlib>
#include <algorithm>

namespace pt {

/* one entry per drawable that has a mesh (Mesh_Get() != NULL), in table order */
struct DFlatEntity
{
    m44 M;                      /* Entities.matrices[i], drawable.h:19 */
    const v4* positions;        /* Mesh arrays in HBM (mesh.h:25-33) */
    const v4* normals;
    const v4* uvs;
    int32_t meshLen;
    int32_t vertBack;           /* first output vertex, path_tracer.c:736 */
};

/*
 * One lane per output vertex; the entity of a vertex is found by bisection of the vertBack offsets
 * (entities are contiguous runs, so the lanes of a wave read the same one or two DFlatEntity records).
 * path_tracer.c:741-771: positions = M * p (w = 1), normals = normalize(IM * n) with
 * IM = inverse(transpose(M3x3)), uvs = (uv.x, uv.y) for whole triangles only (the tail of a mesh whose
 * length is not a multiple of 3 keeps the zeroes of Perm_Calloc), matIds = index of the entity's material.
 */
__global__ __launch_bounds__(256) void k_flatten(const DFlatEntity* __restrict__ ents, int32_t entCount, int32_t vertCount,
    v4* __restrict__ positions, v4* __restrict__ normals, v2* __restrict__ uvs, int32_t* __restrict__ matIds)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vertCount) return;
    int32_t lo = 0, hi = entCount - 1;
    while (lo < hi)
    {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (ents[mid].vertBack <= i) lo = mid; else hi = mid - 1;
    }
    const DFlatEntity e = ents[lo];
    const int32_t j = i - e.vertBack;
    const m33 IM = m44_IM(e.M);
    positions[i] = m44_mul_pt(e.M, e.positions[j]);
    normals[i] = f4_normalize3(m33_mul_col(IM, e.normals[j]));
    v2 uv = f2_v(0.0f, 0.0f);
    if (j < (e.meshLen / 3) * 3)
    {
        const v4 t = e.uvs[j];
        uv = f2_v(t.x, t.y);
    }
    uvs[i] = uv;
    matIds[i] = lo;
}

/*
 * BakeBrdfFn, math/lighting.c:31-84.  One lane per texel; the texel's samples [prevSamples, numSamples) are drawn
 * in order from the texel's own stream Pcg4(iTexel, prevSamples, seed, TAG) (the reference draws from the
 * clock-seeded stream of whichever host thread runs the texel) and summed in order.
 */
#define PT_TAG_LUT 0x4C555421u

__global__ __launch_bounds__(256) void k_brdf_lut(v2* __restrict__ texels, int32_t sx, int32_t sy, int32_t prevSamples, int32_t numSamples, uint32_t seed)
{
    const int32_t iTexel = blockIdx.x * blockDim.x + threadIdx.x;
    if (iTexel >= sx * sy) return;
    const int32_t deltaSamples = numSamples - prevSamples;
    const float dx = 1.0f / (float)deltaSamples;
    const float dt = (float)deltaSamples / (float)numSamples;
    const float rcpWidth = 1.0f / (float)sx, rcpHeight = 1.0f / (float)sy;
    const int32_t cx = iTexel % sx, cy = iTexel / sx;              /* EncodeCoord2 */
    u4 rng = pcg4((uint32_t)iTexel, (uint32_t)prevSamples, seed, PT_TAG_LUT);
    v2 result = f2_v(0.0f, 0.0f);
    for (int32_t iSample = prevSamples; iSample < numSamples; ++iSample)
    {
        const v2 XiAA = rng_f2(&rng);                                  /* draw order of lighting.c:53-60 */
        const float XiPhi = rng_f32(&rng);
        const v2 XiH = rng_f2(&rng);
        const v2 c = BrdfLutSample(cx, cy, rcpWidth, rcpHeight, dx, XiAA, XiPhi, XiH);
        result.x += c.x;
        result.y += c.y;
    }
    const v2 old = texels[iTexel];
    texels[iTexel] = f2_lerps(old, result, dt);                      /* f2_lerpvs */
}

/* ---------------------------------------------------------------------------------------------- */
/* EmbedAttributes / EmbedTaskFn, rendering/lightmap.c:941-1045                                    */

/*
 * The reference gives every texel a loop over every triangle of every drawable (O(texels x triangles)) and keeps the
 * nearest one within kFillPadding, the first in drawable/vertex order on a tie.  That winner is the minimum of the
 * pair (distance, order of the triangle in that loop), so the loop turns inside out: one wave per mapped triangle
 * walks only the texels its padded bounds cover and folds (distance, order) into the texel's 64-bit key with an
 * atomic min; a second pass, one lane per texel, rebuilds the winner's fragment.  Same fp32 operations in the same
 * order per (texel, triangle) pair, so the pages equal the scalar loop's bit for bit.
 */
struct DEmbedEntity
{
    const v4* positions; const v4* normals; const v4* uvs;   /* Mesh arrays in HBM; uvs.zw = lightmap coordinates */
    const int4* texIndices;                                  /* .w = lightmap texture id of the triangle */
    uint32_t triFirst;                                       /* order of the entity's first triangle in the reference's loop */
    int32_t triCount;
};

struct EmbedArgs
{
    const DEmbedEntity* entities; int32_t entityCount;
    uint32_t triTotal;
    const int32_t* pageTexIds; int32_t lmCount, lmSize;
    unsigned long long* keys;                                /* [lmCount * lmSize^2], starts all ones = no triangle */
    float* position; float* normal; float* sampleCounts;     /* float3 / float3 / float per texel, all pages */
};

__device__ __forceinline__ int32_t embed_entity_of(const EmbedArgs& a, uint32_t tri)
{
    int32_t lo = 0, hi = a.entityCount - 1;                  /* last entity whose triFirst <= tri */
    while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (a.entities[mid].triFirst <= tri) lo = mid; else hi = mid - 1; }
    return lo;
}

/* fp32 bits -> u32 that orders like the float (no NaN reaches it) */
__device__ __forceinline__ uint32_t embed_order_bits(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

static constexpr int kEmbedThreads = 256;

__global__ __launch_bounds__(kEmbedThreads) void k_embed_raster(EmbedArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wavesTotal = gridDim.x * (kEmbedThreads / 64);
    const float texelSize = (float)a.lmSize;
    const int64_t lmLen = (int64_t)a.lmSize * a.lmSize;
    for (uint32_t tri = blockIdx.x * (kEmbedThreads / 64) + (threadIdx.x >> 6); tri < a.triTotal; tri += wavesTotal)
    {
        const DEmbedEntity e = a.entities[embed_entity_of(a, tri)];
        const int32_t iVert = (int32_t)(tri - e.triFirst) * 3;
        const int32_t texId = e.texIndices[iVert].w;
        int32_t page = -1;
        for (int32_t i = 0; i < a.lmCount; ++i) if ((a.pageTexIds ? a.pageTexIds[i] : i) == texId) { page = i; break; }
        if (page < 0) continue;
        /* several pages may share an id only by caller error; the reference would then fill each: not supported, first wins */
        const v4 ua = e.uvs[iVert], ub = e.uvs[iVert + 1], uc = e.uvs[iVert + 2];
        const v2 A = f2_muls(f2_v(ua.z, ua.w), texelSize), B = f2_muls(f2_v(ub.z, ub.w), texelSize), C = f2_muls(f2_v(uc.z, uc.w), texelSize);
        /* Texels whose centre can pass the test below: the bounds grown by 3 texels (walked with one more of slack).
         * That holds unless the triangle is a sliver: sdTriangle2D calls a point inside when none of its three edge
         * cross products comes out negative; each is exact to a relative 5e-7 of |p - v| |edge|, and a point that far
         * outside is on the outer side of an edge line by an angle of at least half the sharpest corner, so the sign
         * holds while every corner is wider than ~1e-6 rad.  Below 1e-5 rad (zero area included: the reference then
         * gives EVERY texel a negative distance), or with coordinates that are not finite or huge: the whole page. */
        const v2 ba = f2_sub(B, A), cb = f2_sub(C, B), ac = f2_sub(A, C);
        const float area2 = f1_abs(ba.x * ac.y - ba.y * ac.x);
        const float lmax2 = f1_max(f2_dot(ba, ba), f1_max(f2_dot(cb, cb), f2_dot(ac, ac)));
        const float lox = f1_min(A.x, f1_min(B.x, C.x)), hix = f1_max(A.x, f1_max(B.x, C.x));
        const float loy = f1_min(A.y, f1_min(B.y, C.y)), hiy = f1_max(A.y, f1_max(B.y, C.y));
        const bool bounded = (area2 > 1.0e-5f * lmax2) && lox > -1.0e5f && loy > -1.0e5f && hix < 1.0e5f && hiy < 1.0e5f;
        int32_t x0 = 0, x1 = a.lmSize - 1, y0 = 0, y1 = a.lmSize - 1;
        if (bounded)
        {
            x0 = (int32_t)f1_max(lox - 4.0f, 0.0f); y0 = (int32_t)f1_max(loy - 4.0f, 0.0f);
            x1 = (int32_t)f1_min(hix + 4.0f, (float)(a.lmSize - 1)); y1 = (int32_t)f1_min(hiy + 4.0f, (float)(a.lmSize - 1));
        }
        const int32_t w = x1 - x0 + 1, h = y1 - y0 + 1;
        if (w <= 0 || h <= 0) continue;
        const int64_t cells = (int64_t)w * h;
        for (int64_t c = lane; c < cells; c += 64)
        {
            const int32_t x = x0 + (int32_t)(c % w), y = y0 + (int32_t)(c / w);
            const float dist = sdTriangle2D(A, B, C, f2_v(x + 0.5f, y + 0.5f));
            if (dist < PT_LM_FILL_PADDING)
            {
                const unsigned long long key = ((unsigned long long)embed_order_bits(dist) << 32) | (unsigned long long)tri;
                atomicMin(&a.keys[(int64_t)page * lmLen + (int64_t)y * a.lmSize + x], key);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_embed_resolve(EmbedArgs a)
{
    const int64_t lmLen = (int64_t)a.lmSize * a.lmSize;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lmLen * a.lmCount) return;
    const unsigned long long key = a.keys[i];
    v4 lmPos = f4_s(0.0f), lmNor = f4_s(0.0f);
    float count = 0.0f;
    if (key != ~0ull)
    {
        const uint32_t tri = (uint32_t)key;
        const DEmbedEntity e = a.entities[embed_entity_of(a, tri)];
        const int32_t iVert = (int32_t)(tri - e.triFirst) * 3;
        const int32_t iTexel = (int32_t)(i % lmLen);
        const float texelSize = (float)a.lmSize;
        const v4 ua = e.uvs[iVert], ub = e.uvs[iVert + 1], uc = e.uvs[iVert + 2];
        const v2 A = f2_muls(f2_v(ua.z, ua.w), texelSize), B = f2_muls(f2_v(ub.z, ub.w), texelSize), C = f2_muls(f2_v(uc.z, uc.w), texelSize);
        const v2 px = f2_v((iTexel % a.lmSize) + 0.5f, (iTexel / a.lmSize) + 0.5f);
        const v4 pos[3] = { e.positions[iVert], e.positions[iVert + 1], e.positions[iVert + 2] };
        const v4 nor[3] = { e.normals[iVert], e.normals[iVert + 1], e.normals[iVert + 2] };
        LmEmbedFragment(A, B, C, px, pos, nor, &lmPos, &lmNor);
        lmNor = f4_normalize3(lmNor);                                    /* :1023-1026 */
        count = 1.0f;
    }
    a.sampleCounts[i] = count;
    a.position[i * 3 + 0] = lmPos.x; a.position[i * 3 + 1] = lmPos.y; a.position[i * 3 + 2] = lmPos.z;
    a.normal[i * 3 + 0] = lmNor.x; a.normal[i * 3 + 1] = lmNor.y; a.normal[i * 3 + 2] = lmNor.z;
}

/* ---------------------------------------------------------------------------------------------- */
/* LmPack_Pack's data-parallel steps: chartnodes_create, ChartMaskFn (rendering/lightmap.c)        */

struct DChartEntity
{
    m44 M;                       /* Entities.matrices[d] */
    const v4* positions;
    int32_t drawable;            /* index in the draw table (chartnode_t.drawableIndex) */
    int32_t nodeBack;            /* first node of the entity, lightmap.c:751 */
    int32_t triCount;
};

/* chartnodes_create, lightmap.c:723-767: one lane per triangle */
__global__ __launch_bounds__(256) void k_chartnodes(const DChartEntity* __restrict__ ents, int32_t entCount, int32_t nodeCount,
    float texelsPerUnit, PimChartNode* __restrict__ nodes)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount) return;
    int32_t lo = 0, hi = entCount - 1;
    while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (ents[mid].nodeBack <= i) lo = mid; else hi = mid - 1; }
    const DChartEntity e = ents[lo];
    const int32_t v = (i - e.nodeBack) * 3;
    const v4 A = m44_mul_pt(e.M, e.positions[v]), B = m44_mul_pt(e.M, e.positions[v + 1]), C = m44_mul_pt(e.M, e.positions[v + 2]);
    const ChartGeom g = ChartNodeGeom(A, B, C, texelsPerUnit);
    PimChartNode n;
    n.plane[0] = g.plane.x; n.plane[1] = g.plane.y; n.plane[2] = g.plane.z; n.plane[3] = g.plane.w;
    n.triCoord[0] = g.a.x; n.triCoord[1] = g.a.y; n.triCoord[2] = g.b.x; n.triCoord[3] = g.b.y; n.triCoord[4] = g.c.x; n.triCoord[5] = g.c.y;
    n.area = g.area; n.drawableIndex = e.drawable; n.vertIndex = v; n._pad[0] = n._pad[1] = n._pad[2] = 0;
    nodes[i] = n;
}

__device__ __forceinline__ float wave_min_f(float v) { for (int o = 32; o > 0; o >>= 1) v = f1_min(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_max_f(float v) { for (int o = 32; o > 0; o >>= 1) v = f1_max(v, __shfl_xor(v, o, 64)); return v; }

/* bounds of a chart's triangles (chart_minmax, lightmap.c:359-376): min and max do not depend on the order */
__device__ __forceinline__ void chart_bounds(const PimChartNode* __restrict__ nodes, int32_t first, int32_t last, v2* lo, v2* hi)
{
    const float bigNum = (float)(1 << 23);
    float lx = bigNum, ly = bigNum, hx = -bigNum, hy = -bigNum;
    for (int32_t i = first + (int32_t)(threadIdx.x & 63); i < last; i += 64)
        for (int k = 0; k < 3; ++k)
        {
            const float x = nodes[i].triCoord[2 * k], y = nodes[i].triCoord[2 * k + 1];
            lx = f1_min(lx, x); ly = f1_min(ly, y); hx = f1_max(hx, x); hy = f1_max(hy, y);
        }
    *lo = f2_v(wave_min_f(lx), wave_min_f(ly)); *hi = f2_v(wave_max_f(hx), wave_max_f(hy));
}

/* ChartMaskFn up to mask_new, lightmap.c:532-552: one wave per chart */
__global__ __launch_bounds__(64) void k_chart_layout(PimChartNode* __restrict__ nodes, const int32_t* __restrict__ chartFirst, int32_t chartCount,
    float* __restrict__ areas, int32_t* __restrict__ maskSizes)
{
    const int32_t c = blockIdx.x;
    if (c >= chartCount) return;
    const int32_t first = chartFirst[c], last = chartFirst[c + 1];
    v2 lo, hi;
    chart_bounds(nodes, first, last, &lo, &hi);
    lo = f2_v(lo.x - 2.0f, lo.y - 2.0f);
    /* move to the origin and take the new bounds from the moved coordinates as they are written */
    const float bigNum = (float)(1 << 23);
    float lx = bigNum, ly = bigNum, hx = -bigNum, hy = -bigNum;
    for (int32_t i = first + (int32_t)threadIdx.x; i < last; i += 64)
        for (int k = 0; k < 3; ++k)
        {
            const float x = nodes[i].triCoord[2 * k] - lo.x, y = nodes[i].triCoord[2 * k + 1] - lo.y;
            nodes[i].triCoord[2 * k] = x; nodes[i].triCoord[2 * k + 1] = y;
            lx = f1_min(lx, x); ly = f1_min(ly, y); hx = f1_max(hx, x); hy = f1_max(hy, y);
        }
    lo = f2_v(wave_min_f(lx), wave_min_f(ly)); hi = f2_v(wave_max_f(hx), wave_max_f(hy));
    if (threadIdx.x == 0)
    {
        const v2 size = f2_sub(hi, lo);
        areas[c] = size.x * size.y;
        maskSizes[2 * c] = (int32_t)(hi.x + 2.0f); maskSizes[2 * c + 1] = (int32_t)(hi.y + 2.0f);
    }
}

struct ChartMaskArgs
{
    const PimChartNode* nodes; const int32_t* nodeChart;   /* chart of every node */
    int32_t nodeCount;
    const int32_t* maskSizes; const int64_t* maskOffsets; uint8_t* masks;
};

/* mask_tri (lightmap.c:244-259) for every node: one wave per node over the texels its padded bounds cover (the whole mask
 * for slivers, where the bounds argument of k_embed_raster does not hold); every writer stores the same 0xff */
__global__ __launch_bounds__(256) void k_chart_masks(ChartMaskArgs a)
{
    const int lane = threadIdx.x & 63;
    const int32_t wavesTotal = (int32_t)gridDim.x * 4;
    for (int32_t i = (int32_t)blockIdx.x * 4 + (int32_t)(threadIdx.x >> 6); i < a.nodeCount; i += wavesTotal)
    {
        const int32_t c = a.nodeChart[i];
        const int32_t sx = a.maskSizes[2 * c], sy = a.maskSizes[2 * c + 1];
        if (sx <= 0 || sy <= 0) continue;
        uint8_t* mask = a.masks + a.maskOffsets[c];
        const float* t = a.nodes[i].triCoord;
        const v2 A = f2_v(t[0], t[1]), B = f2_v(t[2], t[3]), C = f2_v(t[4], t[5]);
        const v2 ba = f2_sub(B, A), cb = f2_sub(C, B), ac = f2_sub(A, C);
        const float area2 = f1_abs(ba.x * ac.y - ba.y * ac.x);
        const float lmax2 = f1_max(f2_dot(ba, ba), f1_max(f2_dot(cb, cb), f2_dot(ac, ac)));
        const float lox = f1_min(A.x, f1_min(B.x, C.x)), hix = f1_max(A.x, f1_max(B.x, C.x));
        const float loy = f1_min(A.y, f1_min(B.y, C.y)), hiy = f1_max(A.y, f1_max(B.y, C.y));
        const bool bounded = (area2 > 1.0e-5f * lmax2) && lox > -1.0e5f && loy > -1.0e5f && hix < 1.0e5f && hiy < 1.0e5f;
        int32_t x0 = 0, x1 = sx - 1, y0 = 0, y1 = sy - 1;
        if (bounded)
        {
            x0 = (int32_t)f1_max(lox - 3.0f, 0.0f); y0 = (int32_t)f1_max(loy - 3.0f, 0.0f);
            x1 = (int32_t)f1_min(hix + 3.0f, (float)(sx - 1)); y1 = (int32_t)f1_min(hiy + 3.0f, (float)(sy - 1));
        }
        const int32_t w = x1 - x0 + 1, h = y1 - y0 + 1;
        if (w <= 0 || h <= 0) continue;
        const int64_t cells = (int64_t)w * h;
        for (int64_t k = lane; k < cells; k += 64)
        {
            const int32_t x = x0 + (int32_t)(k % w), y = y0 + (int32_t)(k / w);
            if (sdTriangle2D(A, B, C, f2_v(x + 0.5f, y + 0.5f)) <= PT_LM_MASK_PADDING) mask[x + (int64_t)y * sx] = 0xff;
        }
    }
}

} /* namespace pt */

using namespace pt;

namespace {


/* ---------------------------------------------------------------------------------------------- */
/* atlases_create (rendering/lightmap.c:845-874): first-fit placement of every chart's mask, in the sorted order.
 *
 * The reference hands the sorted charts to its worker threads through one counter (AtlasFn, :779-823); every worker scans an
 * atlas row-major for the first translation at which the chart's mask overlaps nothing (mask_find :261-282, mask_fits
 * :174-207), locks the atlas, re-checks and writes (atlas_search :680-721).  Which chart gets which place then depends on
 * thread timing.  The order of ONE worker is taken here: chart i is placed before chart i + 1 is looked at, with the worker's
 * three hints -- keep searching from the atlas and row of the previous chart as long as the areas shrink by less than a
 * tenth (:801-806), start over from atlas 0 when the search fails (:808-818).
 * What is parallel is the scan itself: the candidate translations of an atlas, taken row-major from the starting row, are
 * tested kAtlasThreads at a time by ONE workgroup (a chart's placement depends on all placements before it: there is nothing
 * for a second workgroup to do), the lowest candidate that fits wins -- the one the sequential scan meets first.
 * A chart that fits in none of the atlas_estimate()d atlases even after the worker's full retry makes the reference spin for
 * ever (:808-818; the estimate counts the charts' bounds without their margins, so it does run short).  Here the run goes on
 * where the reference stops making progress: one more atlas is opened, up to atlasCap, and the chart goes there; every
 * placement up to that point is the reference's.  status[0] stays -1, or receives the chart that fits nowhere (larger than an
 * atlas, or atlasCap reached); status[1] = atlases in use at the end.
 */
static constexpr int kAtlasThreads = 1024;
#define PT_ROW_RESET (-(1 << 20))              /* lightmap.c:37 */
struct AtlasArgs
{
    const uint8_t* masks; const int64_t* maskOffsets; const int32_t* maskSizes; const float* areas; const int32_t* order;
    int32_t chartCount, atlasSize, atlasCount, atlasCap;
    uint8_t* atlases; int32_t* atlasCharts; int32_t* translations; int32_t* atlasIndex; int32_t* status;
};

__global__ __launch_bounds__(kAtlasThreads) void k_atlas_place(AtlasArgs a)
{
    __shared__ int32_t sBest;
    const int32_t tid = (int32_t)threadIdx.x;
    const int32_t S = a.atlasSize;
    int32_t prevAtlas = 0, prevRow = PT_ROW_RESET;
    float prevArea = (float)(1 << 23);
    int32_t atlasCount = a.atlasCount;
    for (int32_t i = 0; i < a.chartCount; ++i)
    {
        const int32_t c = a.order[i];
        const uint8_t* __restrict__ item = a.masks + a.maskOffsets[c];
        const int32_t bw = a.maskSizes[2 * c], bh = a.maskSizes[2 * c + 1];
        const float area = a.areas[c];
        if (area < (prevArea * 0.9f)) { prevArea = area; prevAtlas = 0; prevRow = PT_ROW_RESET; }
        bool placed = false;
        for (int attempt = 0; attempt < 3 && !placed; ++attempt)
        {
            for (int32_t at = prevAtlas; at < atlasCount && !placed; ++at)
            {
                uint8_t* atlas = a.atlases + (size_t)at * (size_t)S * (size_t)S;
                const int32_t rangeX = S - bw, rangeY = S - bh;
                const int32_t y0 = (prevRow != PT_ROW_RESET) ? prevRow : 0;
                int64_t found = -1;
                if (rangeX > 0 && rangeY > y0)
                {
                    const int64_t total = (int64_t)(rangeY - y0) * rangeX;
                    for (int64_t base = 0; base < total && found < 0; base += kAtlasThreads)
                    {
                        if (tid == 0) sBest = 0x7fffffff;
                        __syncthreads();
                        const int64_t idx = base + tid;
                        if (idx < total)
                        {
                            const int32_t y = y0 + (int32_t)(idx / rangeX), x = (int32_t)(idx % rangeX);
                            bool fits = true;                                        /* mask_fits: the bounds hold by the ranges */
                            for (int32_t by = 0; by < bh && fits; ++by)
                            {
                                const uint8_t* arow = atlas + (size_t)(y + by) * (size_t)S + x;
                                const uint8_t* brow = item + (size_t)by * (size_t)bw;
                                for (int32_t bx = 0; bx < bw; ++bx) if ((brow[bx] & arow[bx]) != 0) { fits = false; break; }
                            }
                            if (fits) atomicMin(&sBest, tid);
                        }
                        __syncthreads();
                        const int32_t best = sBest;
                        __syncthreads();
                        if (best != 0x7fffffff) found = base + best;
                    }
                }
                if (found >= 0)
                {
                    const int32_t ty = y0 + (int32_t)(found / rangeX), tx = (int32_t)(found % rangeX);
                    for (int32_t k = tid; k < bw * bh; k += kAtlasThreads)              /* mask_write, lightmap.c:209-230 */
                    {
                        const uint8_t v = item[k];
                        if (v) atlas[(size_t)(ty + k / bw) * (size_t)S + (size_t)(tx + k % bw)] = v;
                    }
                    __threadfence();             /* the next chart's scan reads these bytes from other lanes' writes */
                    __syncthreads();
                    if (tid == 0) { a.translations[2 * c] = tx; a.translations[2 * c + 1] = ty; a.atlasIndex[c] = at; a.atlasCharts[at] += 1; }
                    prevAtlas = at; prevRow = ty;
                    placed = true;
                }
                else prevRow = PT_ROW_RESET;
            }
            if (!placed && attempt == 0) { prevArea = area; prevAtlas = 0; prevRow = PT_ROW_RESET; }
            if (!placed && attempt == 1)
            {
                /* the full search found nothing.  An empty atlas takes the chart iff its mask is smaller than the atlas on both axes */
                if (atlasCount >= a.atlasCap || bw >= S || bh >= S) break;
                prevArea = area; prevAtlas = atlasCount; prevRow = PT_ROW_RESET;
                ++atlasCount;
            }
        }
        if (!placed) { if (tid == 0) { a.status[0] = c; a.status[1] = atlasCount; } return; }
    }
    if (tid == 0) a.status[1] = atlasCount;
}


/*
 * The same search with the atlas as BITS in LDS.  mask_tri writes 0xff or nothing (lightmap.c:255), so a texel of an atlas is
 * one bit; an atlas of up to 1024^2 texels -- the size the reference packs (render_system.c:176) -- is 128 KiB and fits the
 * CU's 160 KiB of LDS beside the chart's own mask (as bit rows, <= kItemWordsLds 64-bit words; a larger one stays in HBM).
 * A candidate translation is then tested row by row with 64 texels per AND, on LDS words every lane of a wave shares
 * (broadcast reads), instead of byte by byte through the vector L1.  The atlas in LDS is written back to HBM (atlasBits) only
 * when the search moves to another atlas and something was placed in it.
 *
 * The chain from one chart to the next is what the kernel's time is made of, so everything that is not the search is taken off it:
 *   - the charts' descriptors and bit rows are laid out in PLACEMENT order by a parallel pass before (k_chart_bits); the loop
 *     reads the descriptor two charts ahead and the bit rows one chart ahead, into the other half of the item buffer, so no
 *     load from HBM is waited for between two charts;
 *   - a round of the scan costs ONE barrier (three result slots in rotation: the slot of round r + 1 is cleared during round r);
 *   - a lane keeps its column x and walks down rows (no division per candidate); a wave reports its first fitting lane by a
 *     ballot, one LDS atomic per wave instead of one per fitting lane;
 *   - a round that finds nothing doubles the rows of the next one (up to 16 per lane): the long scans through crowded atlases
 *     after a reset of the hints pay a sixteenth of the barriers.
 * The candidates are the reference's, in its order (row-major from the starting row), and the lowest fitting one wins.
 */
static constexpr int kAtlasBitsMaxSize = 1024;
static constexpr int kAtlasRowWordsMax = kAtlasBitsMaxSize / 64;
static constexpr int kItemWordsLds = 3584;                     /* 28 KiB: a 345 x 345 chart (a third of the atlas + margins) has 2070 */
static constexpr int kAtlasBitThreads = 256;                   /* one wave per SIMD: the chain from chart to chart is barriers and issue slots, not lanes */
static constexpr int kItemWordsSmall = kAtlasBitThreads;       /* a chart of at most a word per lane: prefetched, two of them side by side */
static constexpr int kAtlasRowsMax = 16;                       /* rows of candidates per lane and round, at most */
struct ChartDesc { int32_t bw, bh; float area; uint32_t wordOff; };      /* in placement order */
struct AtlasBitArgs
{
    AtlasArgs a;                        /* translations / atlasIndex / status[0] are in placement order here (the host maps them back) */
    unsigned long long* atlasBits;      /* [atlasCap][S * rowWords]: atlases not in LDS at the moment; zeroed by the host */
    const ChartDesc* desc;              /* [chartCount] */
    const unsigned long long* bits;     /* the charts' bit rows, chart i's at desc[i].wordOff */
};

/* a wave per chart: the mask's bytes as rows of 64-bit words */
__global__ __launch_bounds__(256) void k_chart_bits(const uint8_t* __restrict__ masks, const int64_t* __restrict__ maskOffsets, const int32_t* __restrict__ order,
    const ChartDesc* __restrict__ desc, int32_t chartCount, unsigned long long* __restrict__ bits)
{
    const int32_t i = (int32_t)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = (int32_t)(threadIdx.x & 63);
    if (i >= chartCount) return;
    const ChartDesc d = desc[i];
    const uint8_t* __restrict__ item = masks + maskOffsets[order[i]];
    const int32_t rowWords = (d.bw + 63) >> 6, words = d.bh * rowWords;
    for (int32_t w = lane; w < words; w += 64)
    {
        const int32_t by = w / rowWords, x0 = (w % rowWords) * 64;
        const uint8_t* brow = item + (size_t)by * (size_t)d.bw;
        unsigned long long v = 0ull;
        const int32_t n = min(64, d.bw - x0);
        for (int32_t b = 0; b < n; ++b) if (brow[x0 + b]) v |= 1ull << b;
        bits[(size_t)d.wordOff + (size_t)w] = v;
    }
}

/* OR over each row of 16 lanes, in every lane of the row (four rotations within the row: vector instructions, no LDS crossbar) */
__device__ __forceinline__ uint32_t row16_or(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);          /* row_ror:1 */
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);          /* row_ror:2 */
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);          /* row_ror:4 */
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);          /* row_ror:8 */
    return v;
}

/* the columns x of a word of 64 candidates at which row `iw` of a mask collides with the atlas row whose words at and after the
 * candidates' word are A0, A1: the OR over the set bits b of iw of (A1:A0) >> b, a run of L set bits in log2 L shift-ORs */
__device__ __forceinline__ unsigned long long blocked_columns(unsigned long long A0, unsigned long long A1, unsigned long long iw)
{
    unsigned long long B = 0ull, m = iw;
    while (m)
    {
        const int32_t l = __builtin_ctzll(m);
        const unsigned long long run = ~(m >> l);
        const int32_t L = run ? __builtin_ctzll(run) : 64;               /* l + L <= 64 */
        m = (l + L >= 64) ? 0ull : (m & (~0ull << (l + L)));
        unsigned long long U0 = (A0 >> l) | (l ? (A1 << (64 - l)) : 0ull), U1 = A1 >> l;
        for (int32_t k = 1; k < L; )
        {
            const int32_t sft = min(k, L - k);
            U0 |= (U0 >> sft) | (U1 << (64 - sft)); U1 |= U1 >> sft;
            k += sft;
        }
        B |= U0;
    }
    return B;
}

__global__ __launch_bounds__(kAtlasBitThreads) void k_atlas_place_bits(AtlasBitArgs g)
{
    const AtlasArgs& a = g.a;
    __shared__ unsigned long long sAtlas[kAtlasBitsMaxSize * kAtlasRowWordsMax];
    __shared__ unsigned long long sItem[kItemWordsLds];
    __shared__ int32_t sBest[4];
    const int32_t tid = (int32_t)threadIdx.x;
    const int32_t S = a.atlasSize;
    const int32_t rowWords = (S + 63) >> 6;
    const int32_t atlasWords = S * rowWords;
    const int32_t n = a.chartCount;
    int32_t prevAtlas = 0, prevRow = PT_ROW_RESET;
    float prevArea = (float)(1 << 23);
    int32_t atlasCount = a.atlasCount;
    int32_t resident = -1;                                      /* the atlas whose bits are in LDS */
    bool dirty = false;                                         /* ... and whether it differs from its copy in HBM */
    uint32_t round = 0;                                         /* rounds of the scan so far */
    uint32_t slot = 0;                                          /* the result slot in use, round % 3 (kept, not divided out) */
    int32_t swaps = 0;
    if (tid < 3) sBest[tid] = 0x7fffffff;
    int32_t opaqueZero;                                         /* 0, in a vector register the compiler knows nothing about */
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaqueZero));
    ChartDesc cur = g.desc[0], nxt = g.desc[min(1, n - 1)];
    int32_t half = 0;                                           /* which half of sItem holds a small chart's rows */
    {
        const int32_t words = cur.bh * ((cur.bw + 63) >> 6);
        if (words <= kItemWordsLds) for (int32_t w = tid; w < words; w += kAtlasBitThreads) sItem[w] = g.bits[(size_t)cur.wordOff + w];
    }
    __syncthreads();
    for (int32_t i = 0; i < n; ++i)
    {
        /* The common charts by the short road, as many in a row as come: no more than 64 rows and 64 columns, the search goes on in
         * the atlas that is in LDS from the row of the chart before, that row has room, and the chart behind it has no more than 64 words of rows --
         * ONE round of one candidate row, a group of lanes per word of its columns.  A wave issues an instruction every
         * fourth cycle at best, so what a chart costs is the instructions on its road: this loop has nothing of the general search
         * around it (attempts, atlases, rows, units, swaps) and its state stays out of the way in between.  A chart it cannot take,
         * or for which the round finds nothing, is left untouched for the general search below (a round was spent). */
        for (;;)
        {
            const int32_t bw = cur.bw, bh = cur.bh;
            const int32_t nextWords = nxt.bh * ((nxt.bw + 63) >> 6);
            if (!(bw <= 64 && bh <= 64 && i + 1 < n && nextWords <= 64 && !(cur.area < (prevArea * 0.9f)) && resident == prevAtlas
                  && prevRow != PT_ROW_RESET && S - bw > 0 && S - bh > prevRow)) break;
            const int4 aheadV = ((const int4*)g.desc)[min(i + 2, n - 1) + opaqueZero];
            unsigned long long pre = 0ull;
            const int32_t lane = tid & 63;
            if (lane < nextWords) pre = g.bits[(size_t)nxt.wordOff + lane];
            /* groups of 16, 32 or 64 lanes by the mask's height: 16, 8 or 4 words of columns per pass, the row's 16 words in 1, 2 or 4 */
            const int32_t gs = bh <= 16 ? 4 : (bh <= 32 ? 5 : 6);
            const int32_t by = tid & ((1 << gs) - 1), grp = tid >> gs, groups = kAtlasBitThreads >> gs;
            const unsigned long long rowOfLane = by < bh ? sItem[half * kItemWordsSmall + by] : 0ull;
            /* the row of the chart before, then up to eleven rows below it, a round each (the candidates in the reference's order): a
             * strip of the atlas that is full sends its next chart a row or two down, and that chart stays on this road */
            int32_t best = 0x7fffffff, row = prevRow;
            for (int32_t down = 0; down < 12 && S - bh > row; ++down, ++row)
            {
                if (tid == 0) sBest[slot == 2u ? 0u : slot + 1u] = 0x7fffffff;
                for (int32_t w = grp; w < 16; w += groups)
                {
                    const int32_t nvalid = (S - bw) - 64 * w;
                    unsigned long long B = 0ull;
                    if (w < rowWords && rowOfLane)
                    {
                        const unsigned long long* arow = sAtlas + (row + by) * rowWords;
                        B = blocked_columns(arow[w], (w + 1 < rowWords) ? arow[w + 1] : 0ull, rowOfLane);
                    }
                    const uint32_t lo = row16_or((uint32_t)B), hi = row16_or((uint32_t)(B >> 32));
                    B = ((unsigned long long)hi << 32) | lo;
                    if (gs >= 5) B |= __shfl_xor(B, 16);
                    if (gs == 6) B |= __shfl_xor(B, 32);
                    const unsigned long long validMask = nvalid >= 64 ? ~0ull : (nvalid > 0 ? (1ull << nvalid) - 1ull : 0ull);
                    const unsigned long long open = ~B & validMask;
                    const unsigned long long vote = __ballot(by == 0 && open != 0ull);
                    if (vote)                                                          /* a wave's groups and passes are in column order */
                    {
                        if (lane == __ffsll((long long)vote) - 1) atomicMin(&sBest[slot], 64 * w + __builtin_ctzll(open));
                        break;
                    }
                }
                __syncthreads();
                best = sBest[slot];
                ++round; slot = slot == 2u ? 0u : slot + 1u;
                if (best != 0x7fffffff) break;
            }
            if (best == 0x7fffffff) break;
            /* mask_write, lightmap.c:209-230 -- by EVERY wave (lanes 0 .. bh - 1 hold the rows), and so the next chart's rows below:
             * the same bits ORed and the same words stored four times over, so that a wave finds what it wrote itself, in its own
             * order, and the round's one barrier above is the only one a chart needs */
            if (lane < bh && rowOfLane)
            {
                const int32_t wi = best >> 6, sh = best & 63;
                unsigned long long* arow = sAtlas + (row + lane) * rowWords;
                atomicOr(&arow[wi], rowOfLane << sh);
                if (sh && wi + 1 < rowWords) atomicOr(&arow[wi + 1], rowOfLane >> (64 - sh));
            }
            dirty = true;
            if (tid == 0) { a.translations[2 * i] = best; a.translations[2 * i + 1] = row; a.atlasIndex[i] = prevAtlas; atomicAdd(&a.atlasCharts[prevAtlas], 1); }
            prevRow = row;
            half ^= 1;
            if (lane < nextWords) sItem[half * kItemWordsSmall + lane] = pre;
            cur = nxt;
            nxt.bw = __builtin_amdgcn_readfirstlane(aheadV.x); nxt.bh = __builtin_amdgcn_readfirstlane(aheadV.y);
            nxt.area = __int_as_float(__builtin_amdgcn_readfirstlane(aheadV.z)); nxt.wordOff = (uint32_t)__builtin_amdgcn_readfirstlane(aheadV.w);
            ++i;
        }
        /* the descriptor two charts ahead, as a VECTOR load whose result is made uniform only at the end of this chart's
         * placement: a uniform load is waited for where it is issued (its readfirstlane follows at once), and that wait is a
         * trip to L2 or HBM on the chain from chart to chart */
        const int4 aheadV = ((const int4*)g.desc)[min(i + 2, n - 1) + opaqueZero];
        const int32_t bw = cur.bw, bh = cur.bh;
        const float area = cur.area;
        const int32_t itemRowWords = (bw + 63) >> 6;
        const int32_t itemWords = bh * itemRowWords;
        const bool itemInLds = itemWords <= kItemWordsLds;
        const bool small = itemWords <= kItemWordsSmall;
        const unsigned long long* __restrict__ itemHbm = g.bits + (size_t)cur.wordOff;
        const unsigned long long* item = sItem + (small ? half * kItemWordsSmall : 0);
        /* the next chart's rows: on their way while this one is placed */
        const int32_t nextWords = nxt.bh * ((nxt.bw + 63) >> 6);
        const bool nextSmall = i + 1 < n && nextWords <= kItemWordsSmall;
        unsigned long long pre = 0ull;
        if (nextSmall && tid < nextWords) pre = g.bits[(size_t)nxt.wordOff + tid];
        if (area < (prevArea * 0.9f)) { prevArea = area; prevAtlas = 0; prevRow = PT_ROW_RESET; }
        /* the word form below (masks of up to 64 x 64): lane `by` of a group of G keeps row `by` of the mask, for the test and for the write */
        const bool wordForm = itemRowWords == 1 && bh <= 64;
        const int32_t gShift = bh <= 16 ? 4 : (bh <= 32 ? 5 : 6), G = 1 << gShift;         /* (shifts: G is not a constant, and a division is forty instructions) */
        const unsigned long long rowOfLane = (wordForm && (tid & (G - 1)) < bh) ? item[tid & (G - 1)] : 0ull;
        bool placed = false;
        for (int attempt = 0; attempt < 3 && !placed; ++attempt)
        {
            for (int32_t at = prevAtlas; at < atlasCount && !placed; ++at)
            {
                const int32_t rangeX = S - bw, rangeY = S - bh;
                const int32_t y0 = (prevRow != PT_ROW_RESET) ? prevRow : 0;
                int32_t foundX = -1, foundY = -1;
                if (rangeX > 0 && rangeY > y0)
                {
                    if (resident != at)
                    {
                        /* swap the LDS atlas: write the resident one back, bring this one in */
                        if (resident >= 0 && dirty) for (int32_t w = tid; w < atlasWords; w += kAtlasBitThreads) g.atlasBits[(size_t)resident * atlasWords + w] = sAtlas[w];
                        __threadfence();
                        __syncthreads();
                        for (int32_t w = tid; w < atlasWords; w += kAtlasBitThreads) sAtlas[w] = g.atlasBits[(size_t)at * atlasWords + w];
                        resident = at; dirty = false; ++swaps;
                        __syncthreads();
                    }
                    if (wordForm)
                    {
                        /* A mask of at most 64 x 64 texels (nearly every chart): the test runs over WORDS of candidates.  Wave w
                         * owns the columns 64 w .. 64 w + 63 of a candidate row y; its lane `by` takes row `by` of the mask and
                         * forms, for all 64 columns at once, where that row collides: the OR over the mask row's set bits b of
                         * (atlas row y + by) >> b -- run by run, a run of L bits in log2 L shift-ORs.  The OR over the lanes is
                         * the blocked columns of row y; the first clear bit is the first fit.  Masks of up to 16 (32) rows share
                         * a wave four (two) candidate rows at a time. */
                        const int32_t perWave = 64 >> gShift;
                        const int32_t lane = tid & 63, wave = tid >> 6;
                        const int32_t gi = lane >> gShift, by = lane & (G - 1);
                        /* a unit of work = (candidate row j, word w of columns), numbered u = 16 j + w in candidate order; a group of
                         * G lanes takes one unit, groups in unit order over the waves: a round of ONE candidate row (nine in ten) keeps
                         * 16 / perWave waves busy, one per SIMD when masks have <= 16 rows, and the others step aside */
                        const unsigned long long iw = rowOfLane;
                        int32_t rows = 1;                                            /* candidate rows in the next round */
                        for (int32_t yBase = y0; yBase < rangeY && foundY < 0; )
                        {
                            if (tid == 0) sBest[slot == 2u ? 0u : slot + 1u] = 0x7fffffff;
                            const int32_t units = rows * 16;
                            for (int32_t u0 = wave * perWave; u0 < units; u0 += (kAtlasBitThreads / 64) * perWave)
                            {
                                const int32_t u = u0 + gi;
                                const int32_t rowOff = u >> 4, w = u & 15;
                                const int32_t y = yBase + rowOff;
                                const bool live = rowOff < rows && y < rangeY && w < rowWords;
                                unsigned long long B = 0ull;
                                if (live && iw)
                                {
                                    const unsigned long long* arow = sAtlas + (y + by) * rowWords;
                                    const unsigned long long A0 = arow[w], A1 = (w + 1 < rowWords) ? arow[w + 1] : 0ull;
                                    B = blocked_columns(A0, A1, iw);
                                }
                                {
                                    uint32_t lo = row16_or((uint32_t)B), hi = row16_or((uint32_t)(B >> 32));
                                    B = ((unsigned long long)hi << 32) | lo;
                                    if (G >= 32) B |= __shfl_xor(B, 16);
                                    if (G == 64) B |= __shfl_xor(B, 32);
                                }
                                const int32_t nvalid = rangeX - 64 * w;
                                const unsigned long long validMask = nvalid >= 64 ? ~0ull : (nvalid > 0 ? (1ull << nvalid) - 1ull : 0ull);
                                const unsigned long long open = live ? (~B & validMask) : 0ull;
                                const unsigned long long vote = __ballot(by == 0 && open != 0ull);
                                if (vote)                                            /* groups of a wave are in candidate order */
                                {
                                    if (lane == __ffsll((long long)vote) - 1) atomicMin(&sBest[slot], rowOff * 1024 + 64 * w + __builtin_ctzll(open));
                                    break;
                                }
                            }
                            __syncthreads();
                            const int32_t best = sBest[slot];
                            ++round; slot = slot == 2u ? 0u : slot + 1u;
                            if (best != 0x7fffffff) { foundY = yBase + (best >> 10); foundX = best & 1023; }
                            else { yBase += rows; rows = min(64, rows * 2); }
                        }
                    }
                    else
                    {
                    /* lane -> (row within a pass, column): several rows per pass when the lanes outnumber the columns, several
                     * passes per row when the columns outnumber the lanes */
                    const int32_t rowsPer = max(1, kAtlasBitThreads / rangeX);
                    const int32_t lr = rowsPer > 1 ? tid / rangeX : 0, x0 = tid - lr * rangeX;
                    const bool active = lr < rowsPer;
                    int32_t rows = 1;                                                /* passes per lane in the next round */
                    for (int32_t yBase = y0; yBase < rangeY && foundY < 0; )
                    {
                        if (tid == 0) sBest[slot == 2u ? 0u : slot + 1u] = 0x7fffffff;
                        bool voted = false;
                        for (int32_t j = 0; j < rows && !voted; ++j)
                        for (int32_t xb = 0; xb < rangeX && !voted; xb += kAtlasBitThreads)
                        {
                            const int32_t rowOff = j * rowsPer + lr, x = x0 + xb;
                            const int32_t y = yBase + rowOff;
                            bool fits = active && x < rangeX && y < rangeY;
                            if (itemRowWords == 1)
                            {
                                /* a mask of at most 64 columns (nearly every chart): four rows per step, their reads independent of
                                 * each other -- row by row the test is a chain of LDS latencies */
                                const int32_t wi = x >> 6, sh = x & 63;
                                const bool two = sh != 0 && wi + 1 < rowWords;
                                const int32_t wj = two ? wi + 1 : wi;
                                const int32_t shUp = two ? 64 - sh : 0;
                                const unsigned long long keepUp = two ? ~0ull : 0ull;
                                for (int32_t by = 0; by < bh && fits; by += 4)
                                {
                                    unsigned long long hit = 0ull;
#pragma unroll
                                    for (int32_t u = 0; u < 4; ++u)
                                    {
                                        const int32_t r = min(by + u, bh - 1);
                                        const unsigned long long* arow = sAtlas + (y + r) * rowWords;
                                        const unsigned long long aw = (arow[wi] >> sh) | ((arow[wj] << shUp) & keepUp);
                                        hit |= aw & item[r];
                                    }
                                    if (hit) fits = false;
                                }
                            }
                            else
                            for (int32_t by = 0; by < bh && fits; ++by)
                            {
                                const unsigned long long* arow = sAtlas + (y + by) * rowWords;
                                for (int32_t k = 0; k < itemRowWords; ++k)
                                {
                                    const unsigned long long iw = itemInLds ? item[by * itemRowWords + k] : itemHbm[by * itemRowWords + k];
                                    if (!iw) continue;
                                    const int32_t pos = x + 64 * k, wi = pos >> 6, sh = pos & 63;
                                    unsigned long long aw = arow[wi] >> sh;
                                    if (sh && wi + 1 < rowWords) aw |= arow[wi + 1] << (64 - sh);
                                    if (aw & iw) { fits = false; break; }
                                }
                            }
                            const unsigned long long vote = __ballot(fits);
                            if (vote)                                                /* lanes of a wave are in candidate order, and pass j before pass j + 1 */
                            {
                                if ((int32_t)(threadIdx.x & 63) == __ffsll((long long)vote) - 1) atomicMin(&sBest[slot], rowOff * 1024 + x);
                                voted = true;
                            }
                        }
                        __syncthreads();
                        const int32_t best = sBest[slot];
                        ++round; slot = slot == 2u ? 0u : slot + 1u;
                        if (best != 0x7fffffff) { foundY = yBase + (best >> 10); foundX = best & 1023; }
                        else { yBase += rows * rowsPer; rows = min(kAtlasRowsMax, rows * 2); }
                    }
                    }
                }
                if (foundY >= 0)
                {
                    const int32_t ty = foundY, tx = foundX;
                    for (int32_t w = tid; w < itemWords; w += kAtlasBitThreads)                 /* mask_write, lightmap.c:209-230 */
                    {
                        const unsigned long long iw = wordForm ? rowOfLane : (itemInLds ? item[w] : itemHbm[w]);     /* word form: w = tid < bh <= G */
                        if (!iw) continue;
                        int32_t by = w, pos = tx;
                        if (itemRowWords != 1) { by = w / itemRowWords; pos = tx + 64 * (w - by * itemRowWords); }
                        const int32_t wi = pos >> 6, sh = pos & 63;
                        unsigned long long* arow = sAtlas + (ty + by) * rowWords;
                        atomicOr(&arow[wi], iw << sh);
                        if (sh && wi + 1 < rowWords) atomicOr(&arow[wi + 1], iw >> (64 - sh));
                    }
                    dirty = true;
                    if (tid == 0) { a.translations[2 * i] = tx; a.translations[2 * i + 1] = ty; a.atlasIndex[i] = at; atomicAdd(&a.atlasCharts[at], 1); }
                    prevAtlas = at; prevRow = ty;
                    placed = true;
                }
                else prevRow = PT_ROW_RESET;
            }
            if (!placed && attempt == 0) { prevArea = area; prevAtlas = 0; prevRow = PT_ROW_RESET; }
            if (!placed && attempt == 1)
            {
                if (atlasCount >= a.atlasCap || bw >= S || bh >= S) break;
                prevArea = area; prevAtlas = atlasCount; prevRow = PT_ROW_RESET;
                ++atlasCount;
            }
        }
        if (!placed) { if (tid == 0) { a.status[0] = i; a.status[1] = atlasCount; } return; }
        /* the next chart's rows into the item buffer: a small one beside a small one needs no barrier of its own */
        if (i + 1 < n)
        {
            if (small && nextSmall) { half ^= 1; if (tid < nextWords) sItem[half * kItemWordsSmall + tid] = pre; }
            else
            {
                __syncthreads();                                     /* the rows in use have been read */
                if (nextSmall) { half = 0; if (tid < nextWords) sItem[tid] = pre; }
                else if (nextWords <= kItemWordsLds) for (int32_t w = tid; w < nextWords; w += kAtlasBitThreads) sItem[w] = g.bits[(size_t)nxt.wordOff + w];
            }
        }
        __syncthreads();                                             /* the atlas with this chart in it, and the next chart's rows */
        cur = nxt;
        nxt.bw = __builtin_amdgcn_readfirstlane(aheadV.x); nxt.bh = __builtin_amdgcn_readfirstlane(aheadV.y);
        nxt.area = __int_as_float(__builtin_amdgcn_readfirstlane(aheadV.z)); nxt.wordOff = (uint32_t)__builtin_amdgcn_readfirstlane(aheadV.w);
    }
    if (tid == 0) { a.status[1] = atlasCount; a.status[2] = (int32_t)round; a.status[3] = swaps; }
}

struct DevBuf
{
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t bytes)
    {
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { p = nullptr; set_error("hipMalloc of %zu bytes failed", bytes); return PIMGPU_E_NOMEM; }
        return PIMGPU_OK;
    }
};

int flatten_validate(const PimDrawables* d, const char* who)
{
    if (!d) { set_error("%s: NULL drawables", who); return PIMGPU_E_INVALID; }
    if (d->count < 0 || d->meshCount < 0) { set_error("%s: negative count", who); return PIMGPU_E_INVALID; }
    if (d->count > 0 && (!d->meshes || !d->matrices || !d->materials)) { set_error("%s: missing entity columns", who); return PIMGPU_E_INVALID; }
    if (d->meshCount > 0 && !d->meshTable) { set_error("%s: meshCount > 0 but meshTable == NULL", who); return PIMGPU_E_INVALID; }
    for (int32_t i = 0; i < d->count; ++i)
    {
        if (d->meshes[i] >= d->meshCount) { set_error("%s: entity %d names mesh %d of %d", who, i, d->meshes[i], d->meshCount); return PIMGPU_E_INVALID; }
    }
    for (int32_t m = 0; m < d->meshCount; ++m)
    {
        const PimMesh& mesh = d->meshTable[m];
        if (mesh.length < 0 || (mesh.length > 0 && (!mesh.positions || !mesh.normals || !mesh.uvs)))
        { set_error("%s: mesh %d is missing an array", who, m); return PIMGPU_E_INVALID; }
    }
    return PIMGPU_OK;
}

} /* namespace */

extern "C" {

int pimgpu_flatten_counts(const PimDrawables* d, int32_t* vertCount, int32_t* matCount)
{
    int rc = flatten_validate(d, "flatten_counts"); if (rc) return rc;
    int64_t verts = 0; int32_t mats = 0;
    for (int32_t i = 0; i < d->count; ++i)
    {
        if (d->meshes[i] < 0) continue;                               /* Mesh_Get() == NULL, path_tracer.c:706-709 */
        verts += d->meshTable[d->meshes[i]].length;
        ++mats;
    }
    if (verts > 0x7fffffffll) { set_error("flatten_counts: %lld vertices do not fit the reference's i32 vertCount", (long long)verts); return PIMGPU_E_LIMIT; }
    if (vertCount) *vertCount = (int32_t)verts;
    if (matCount) *matCount = mats;
    return PIMGPU_OK;
}

int pimgpu_flatten(const PimDrawables* d, float* positions, float* normals, float* uvs, int32_t* matIds, PimMaterial* materials)
{
    int rc = check_ready(); if (rc) return rc;
    int32_t V = 0, M = 0;
    if ((rc = pimgpu_flatten_counts(d, &V, &M))) return rc;
    if (V > 0 && (!positions || !normals || !uvs || !matIds)) { set_error("flatten: NULL output array"); return PIMGPU_E_INVALID; }
    if (M > 0 && !materials) { set_error("flatten: NULL materials output"); return PIMGPU_E_INVALID; }

    /* every mesh goes to HBM once, however many entities draw it */
    std::vector<DevBuf> meshBufs((size_t)d->meshCount);
    std::vector<const v4*> meshPos((size_t)d->meshCount, nullptr), meshNor((size_t)d->meshCount, nullptr), meshUv((size_t)d->meshCount, nullptr);
    std::vector<char> used((size_t)d->meshCount, 0);
    for (int32_t i = 0; i < d->count; ++i) if (d->meshes[i] >= 0) used[(size_t)d->meshes[i]] = 1;
    for (int32_t m = 0; m < d->meshCount; ++m)
    {
        const PimMesh& mesh = d->meshTable[m];
        if (!used[(size_t)m] || mesh.length == 0) continue;
        const size_t bytes = sizeof(v4) * (size_t)mesh.length;
        if ((rc = meshBufs[(size_t)m].alloc(bytes * 3))) return rc;
        char* base = (char*)meshBufs[(size_t)m].p;
        PT_HIP(hipMemcpyAsync(base, mesh.positions, bytes, hipMemcpyHostToDevice, g_lib.stream));
        PT_HIP(hipMemcpyAsync(base + bytes, mesh.normals, bytes, hipMemcpyHostToDevice, g_lib.stream));
        PT_HIP(hipMemcpyAsync(base + 2 * bytes, mesh.uvs, bytes, hipMemcpyHostToDevice, g_lib.stream));
        meshPos[(size_t)m] = (const v4*)base; meshNor[(size_t)m] = (const v4*)(base + bytes); meshUv[(size_t)m] = (const v4*)(base + 2 * bytes);
    }
    std::vector<DFlatEntity> ents; ents.reserve((size_t)M);
    int32_t vertBack = 0;
    for (int32_t i = 0; i < d->count; ++i)
    {
        const int32_t m = d->meshes[i];
        if (m < 0) continue;
        const PimMesh& mesh = d->meshTable[m];
        materials[ents.size()] = d->materials[i];                    /* sceneMats[matBack] = materials[i], :743 */
        DFlatEntity e;
        const float* c = d->matrices + (size_t)i * 16;
        e.M.c0 = f4_v(c[0], c[1], c[2], c[3]); e.M.c1 = f4_v(c[4], c[5], c[6], c[7]);
        e.M.c2 = f4_v(c[8], c[9], c[10], c[11]); e.M.c3 = f4_v(c[12], c[13], c[14], c[15]);
        e.positions = meshPos[(size_t)m]; e.normals = meshNor[(size_t)m]; e.uvs = meshUv[(size_t)m];
        e.meshLen = mesh.length; e.vertBack = vertBack;
        vertBack += mesh.length;
        ents.push_back(e);
    }
    if (V == 0) { PT_HIP(hipStreamSynchronize(g_lib.stream)); return PIMGPU_OK; }

    DevBuf entBuf, outBuf;
    if ((rc = entBuf.alloc(sizeof(DFlatEntity) * ents.size()))) return rc;
    const size_t bytesOut = (size_t)V * (16 + 16 + 8 + 4);
    if ((rc = outBuf.alloc(bytesOut))) return rc;
    PT_HIP(hipMemcpyAsync(entBuf.p, ents.data(), sizeof(DFlatEntity) * ents.size(), hipMemcpyHostToDevice, g_lib.stream));
    v4* posDev = (v4*)outBuf.p; v4* norDev = posDev + V; v2* uvDev = (v2*)(norDev + V); int32_t* matDev = (int32_t*)(uvDev + V);
    k_flatten<<<(V + 255) / 256, 256, 0, g_lib.stream>>>((const DFlatEntity*)entBuf.p, (int32_t)ents.size(), V, posDev, norDev, uvDev, matDev);
    PT_HIP(hipGetLastError());
    PT_HIP(hipMemcpyAsync(positions, posDev, sizeof(v4) * (size_t)V, hipMemcpyDeviceToHost, g_lib.stream));
    PT_HIP(hipMemcpyAsync(normals, norDev, sizeof(v4) * (size_t)V, hipMemcpyDeviceToHost, g_lib.stream));
    PT_HIP(hipMemcpyAsync(uvs, uvDev, sizeof(v2) * (size_t)V, hipMemcpyDeviceToHost, g_lib.stream));
    PT_HIP(hipMemcpyAsync(matIds, matDev, sizeof(int32_t) * (size_t)V, hipMemcpyDeviceToHost, g_lib.stream));
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    return PIMGPU_OK;
}

int pimgpu_brdf_lut_bake(float* texels, int32_t width, int32_t height, int32_t prevSamples, int32_t numSamples, uint32_t seed)
{
    int rc = check_ready(); if (rc) return rc;
    if (!texels || width <= 0 || height <= 0 || (int64_t)width * height > (1 << 26)) { set_error("brdf_lut_bake: bad table"); return PIMGPU_E_INVALID; }
    if (prevSamples < 0 || numSamples < prevSamples) { set_error("brdf_lut_bake: samples [%d, %d)", prevSamples, numSamples); return PIMGPU_E_INVALID; }
    if (numSamples == prevSamples) return PIMGPU_OK;                  /* BrdfLut_Update: nothing to add, lighting.c:110 */
    const int32_t n = width * height;
    DevBuf buf;
    if ((rc = buf.alloc(sizeof(v2) * (size_t)n))) return rc;
    PT_HIP(hipMemcpyAsync(buf.p, texels, sizeof(v2) * (size_t)n, hipMemcpyHostToDevice, g_lib.stream));
    k_brdf_lut<<<(n + 255) / 256, 256, 0, g_lib.stream>>>((v2*)buf.p, width, height, prevSamples, numSamples, seed);
    PT_HIP(hipGetLastError());
    PT_HIP(hipMemcpyAsync(texels, buf.p, sizeof(v2) * (size_t)n, hipMemcpyDeviceToHost, g_lib.stream));
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    return PIMGPU_OK;
}

int pimgpu_lightmap_embed(const PimDrawables* d, const int32_t* pageTexIds, PimLightmapPage* pages, int32_t lmCount, int32_t lmSize)
{
    int rc = check_ready(); if (rc) return rc;
    if ((rc = flatten_validate(d, "lightmap_embed"))) return rc;
    if (lmCount < 0 || lmSize <= 0 || lmSize > 16384 || (lmCount > 0 && !pages)) { set_error("lightmap_embed: bad page arguments"); return PIMGPU_E_INVALID; }
    for (int32_t i = 0; i < lmCount; ++i) if (!pages[i].block) { set_error("lightmap_embed: page %d has no block", i); return PIMGPU_E_INVALID; }
    if (lmCount == 0) return PIMGPU_OK;
    hipStream_t st = g_lib.stream;
    const int64_t lmLen = (int64_t)lmSize * lmSize, texels = lmLen * lmCount;

    /* the drawables EmbedTaskFn does not skip (:976-984), in its order; every mesh goes to HBM once */
    std::vector<DevBuf> meshBufs((size_t)d->meshCount);
    std::vector<DEmbedEntity> ents;
    uint64_t triTotal = 0;
    for (int32_t i = 0; i < d->count; ++i)
    {
        const int32_t m = d->meshes[i];
        if (m < 0 || (d->materials[i].flags & (PIM_MATFLAG_SKY | PIM_MATFLAG_LAVA))) continue;
        const PimMesh& mesh = d->meshTable[m];
        if (!mesh.texIndices || mesh.length < 3) continue;
        const size_t bytes = sizeof(v4) * (size_t)mesh.length;
        if (!meshBufs[(size_t)m].p)
        {
            if ((rc = meshBufs[(size_t)m].alloc(bytes * 4))) return rc;
            char* base = (char*)meshBufs[(size_t)m].p;
            PT_HIP(hipMemcpyAsync(base, mesh.positions, bytes, hipMemcpyHostToDevice, st));
            PT_HIP(hipMemcpyAsync(base + bytes, mesh.normals, bytes, hipMemcpyHostToDevice, st));
            PT_HIP(hipMemcpyAsync(base + 2 * bytes, mesh.uvs, bytes, hipMemcpyHostToDevice, st));
            PT_HIP(hipMemcpyAsync(base + 3 * bytes, mesh.texIndices, bytes, hipMemcpyHostToDevice, st));
        }
        char* base = (char*)meshBufs[(size_t)m].p;
        DEmbedEntity e;
        e.positions = (const v4*)base; e.normals = (const v4*)(base + bytes); e.uvs = (const v4*)(base + 2 * bytes);
        e.texIndices = (const int4*)(base + 3 * bytes);
        e.triFirst = (uint32_t)triTotal; e.triCount = mesh.length / 3;
        triTotal += (uint64_t)e.triCount;
        ents.push_back(e);
    }
    if (triTotal > 0xfffffff0ull) { set_error("lightmap_embed: %llu triangles exceed the 32-bit triangle order", (unsigned long long)triTotal); return PIMGPU_E_LIMIT; }

    DevBuf entBuf, idBuf, keyBuf, outBuf;
    if ((rc = entBuf.alloc(sizeof(DEmbedEntity) * ents.size()))) return rc;
    if ((rc = keyBuf.alloc(sizeof(unsigned long long) * (size_t)texels))) return rc;
    if ((rc = outBuf.alloc(sizeof(float) * 7 * (size_t)texels))) return rc;
    if (!ents.empty()) PT_HIP(hipMemcpyAsync(entBuf.p, ents.data(), sizeof(DEmbedEntity) * ents.size(), hipMemcpyHostToDevice, st));
    if (pageTexIds)
    {
        if ((rc = idBuf.alloc(sizeof(int32_t) * (size_t)lmCount))) return rc;
        PT_HIP(hipMemcpyAsync(idBuf.p, pageTexIds, sizeof(int32_t) * (size_t)lmCount, hipMemcpyHostToDevice, st));
    }
    PT_HIP(hipMemsetAsync(keyBuf.p, 0xff, sizeof(unsigned long long) * (size_t)texels, st));
    EmbedArgs a;
    a.entities = (const DEmbedEntity*)entBuf.p; a.entityCount = (int32_t)ents.size(); a.triTotal = (uint32_t)triTotal;
    a.pageTexIds = pageTexIds ? (const int32_t*)idBuf.p : nullptr; a.lmCount = lmCount; a.lmSize = lmSize;
    a.keys = (unsigned long long*)keyBuf.p;
    a.position = (float*)outBuf.p; a.normal = a.position + 3 * texels; a.sampleCounts = a.normal + 3 * texels;
    if (triTotal > 0)
    {
        const int64_t wantBlocks = ((int64_t)triTotal + (kEmbedThreads / 64) - 1) / (kEmbedThreads / 64);
        const int blocks = (int)std::min<int64_t>(wantBlocks, (int64_t)g_lib.cuCount * 64);
        k_embed_raster<<<blocks, kEmbedThreads, 0, st>>>(a);
        PT_HIP(hipGetLastError());
    }
    k_embed_resolve<<<(unsigned)((texels + 255) / 256), 256, 0, st>>>(a);
    PT_HIP(hipGetLastError());
    for (int32_t i = 0; i < lmCount; ++i)
    {
        float* block = pages[i].block;                                 /* [5][S*S] float4 | float3 position | float3 normal | float counts */
        PT_HIP(hipMemcpyAsync(block + lmLen * 20, a.position + 3 * lmLen * i, sizeof(float) * 3 * (size_t)lmLen, hipMemcpyDeviceToHost, st));
        PT_HIP(hipMemcpyAsync(block + lmLen * 23, a.normal + 3 * lmLen * i, sizeof(float) * 3 * (size_t)lmLen, hipMemcpyDeviceToHost, st));
        PT_HIP(hipMemcpyAsync(block + lmLen * 26, a.sampleCounts + lmLen * i, sizeof(float) * (size_t)lmLen, hipMemcpyDeviceToHost, st));
    }
    PT_HIP(hipStreamSynchronize(st));
    return PIMGPU_OK;
}

int pimgpu_lightmap_chartnodes(const PimDrawables* d, float texelsPerUnit, PimChartNode* nodes, int32_t capacity, int32_t* nodeCount)
{
    int rc = flatten_validate(d, "lightmap_chartnodes"); if (rc) return rc;
    if (!nodeCount) { set_error("lightmap_chartnodes: NULL nodeCount"); return PIMGPU_E_INVALID; }
    int64_t count = 0;
    for (int32_t i = 0; i < d->count; ++i)
        if (d->meshes[i] >= 0 && !(d->materials[i].flags & (PIM_MATFLAG_SKY | PIM_MATFLAG_LAVA))) count += d->meshTable[d->meshes[i]].length / 3;
    if (count > 0x7fffffffll) { set_error("lightmap_chartnodes: %lld nodes", (long long)count); return PIMGPU_E_LIMIT; }
    *nodeCount = (int32_t)count;
    if (!nodes || count == 0) return PIMGPU_OK;                       /* the sizing pass */
    if ((rc = check_ready())) return rc;
    if (capacity < count) { set_error("lightmap_chartnodes: room for %d of %lld nodes", capacity, (long long)count); return PIMGPU_E_INVALID; }
    hipStream_t st = g_lib.stream;
    std::vector<DevBuf> meshBufs((size_t)d->meshCount);
    std::vector<DChartEntity> ents;
    int32_t nodeBack = 0;
    for (int32_t i = 0; i < d->count; ++i)
    {
        const int32_t m = d->meshes[i];
        if (m < 0 || (d->materials[i].flags & (PIM_MATFLAG_SKY | PIM_MATFLAG_LAVA))) continue;
        const PimMesh& mesh = d->meshTable[m];
        if (mesh.length < 3) continue;
        if (!meshBufs[(size_t)m].p)
        {
            if ((rc = meshBufs[(size_t)m].alloc(sizeof(v4) * (size_t)mesh.length))) return rc;
            PT_HIP(hipMemcpyAsync(meshBufs[(size_t)m].p, mesh.positions, sizeof(v4) * (size_t)mesh.length, hipMemcpyHostToDevice, st));
        }
        DChartEntity e;
        const float* c = d->matrices + (size_t)i * 16;
        e.M.c0 = f4_v(c[0], c[1], c[2], c[3]); e.M.c1 = f4_v(c[4], c[5], c[6], c[7]);
        e.M.c2 = f4_v(c[8], c[9], c[10], c[11]); e.M.c3 = f4_v(c[12], c[13], c[14], c[15]);
        e.positions = (const v4*)meshBufs[(size_t)m].p; e.drawable = i; e.nodeBack = nodeBack; e.triCount = mesh.length / 3;
        nodeBack += e.triCount;
        ents.push_back(e);
    }
    DevBuf entBuf, nodeBuf;
    if ((rc = entBuf.alloc(sizeof(DChartEntity) * ents.size()))) return rc;
    if ((rc = nodeBuf.alloc(sizeof(PimChartNode) * (size_t)count))) return rc;
    PT_HIP(hipMemcpyAsync(entBuf.p, ents.data(), sizeof(DChartEntity) * ents.size(), hipMemcpyHostToDevice, st));
    k_chartnodes<<<(unsigned)((count + 255) / 256), 256, 0, st>>>((const DChartEntity*)entBuf.p, (int32_t)ents.size(), (int32_t)count, texelsPerUnit, (PimChartNode*)nodeBuf.p);
    PT_HIP(hipGetLastError());
    PT_HIP(hipMemcpyAsync(nodes, nodeBuf.p, sizeof(PimChartNode) * (size_t)count, hipMemcpyDeviceToHost, st));
    PT_HIP(hipStreamSynchronize(st));
    return PIMGPU_OK;
}

static int chart_args_check(const char* who, const void* nodes, const int32_t* chartFirst, int32_t chartCount)
{
    if (chartCount < 0 || (chartCount > 0 && (!nodes || !chartFirst))) { set_error("%s: bad chart arguments", who); return PIMGPU_E_INVALID; }
    for (int32_t c = 0; c < chartCount; ++c)
        if (chartFirst[c] < 0 || chartFirst[c + 1] <= chartFirst[c]) { set_error("%s: chart %d is empty or out of order", who, c); return PIMGPU_E_INVALID; }
    return PIMGPU_OK;
}

int pimgpu_lightmap_chart_layout(PimChartNode* nodes, const int32_t* chartFirst, int32_t chartCount, float* areas, int32_t* maskSizes)
{
    int rc = check_ready(); if (rc) return rc;
    if ((rc = chart_args_check("lightmap_chart_layout", nodes, chartFirst, chartCount))) return rc;
    if (chartCount == 0) return PIMGPU_OK;
    if (!areas || !maskSizes) { set_error("lightmap_chart_layout: NULL output"); return PIMGPU_E_INVALID; }
    hipStream_t st = g_lib.stream;
    const size_t nodeCount = (size_t)chartFirst[chartCount];
    DevBuf nodeBuf, firstBuf, outBuf;
    if ((rc = nodeBuf.alloc(sizeof(PimChartNode) * nodeCount))) return rc;
    if ((rc = firstBuf.alloc(sizeof(int32_t) * ((size_t)chartCount + 1)))) return rc;
    if ((rc = outBuf.alloc(12 * (size_t)chartCount))) return rc;
    PT_HIP(hipMemcpyAsync(nodeBuf.p, nodes, sizeof(PimChartNode) * nodeCount, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(firstBuf.p, chartFirst, sizeof(int32_t) * ((size_t)chartCount + 1), hipMemcpyHostToDevice, st));
    float* areasDev = (float*)outBuf.p; int32_t* sizesDev = (int32_t*)(areasDev + chartCount);
    k_chart_layout<<<chartCount, 64, 0, st>>>((PimChartNode*)nodeBuf.p, (const int32_t*)firstBuf.p, chartCount, areasDev, sizesDev);
    PT_HIP(hipGetLastError());
    PT_HIP(hipMemcpyAsync(nodes, nodeBuf.p, sizeof(PimChartNode) * nodeCount, hipMemcpyDeviceToHost, st));
    PT_HIP(hipMemcpyAsync(areas, areasDev, sizeof(float) * (size_t)chartCount, hipMemcpyDeviceToHost, st));
    PT_HIP(hipMemcpyAsync(maskSizes, sizesDev, 8 * (size_t)chartCount, hipMemcpyDeviceToHost, st));
    PT_HIP(hipStreamSynchronize(st));
    return PIMGPU_OK;
}

int pimgpu_lightmap_chart_masks(const PimChartNode* nodes, const int32_t* chartFirst, int32_t chartCount, const int32_t* maskSizes,
    const int64_t* maskOffsets, uint8_t* masks)
{
    int rc = check_ready(); if (rc) return rc;
    if ((rc = chart_args_check("lightmap_chart_masks", nodes, chartFirst, chartCount))) return rc;
    if (chartCount == 0) return PIMGPU_OK;
    if (!maskSizes || !maskOffsets || !masks) { set_error("lightmap_chart_masks: NULL argument"); return PIMGPU_E_INVALID; }
    int64_t total = 0;
    for (int32_t c = 0; c < chartCount; ++c)
    {
        const int64_t sx = maskSizes[2 * c], sy = maskSizes[2 * c + 1];
        if (sx < 0 || sy < 0 || sx > 65536 || sy > 65536 || maskOffsets[c] < 0) { set_error("lightmap_chart_masks: chart %d has a bad mask size or offset", c); return PIMGPU_E_INVALID; }
        total = std::max<int64_t>(total, maskOffsets[c] + sx * sy);
    }
    if (total == 0) return PIMGPU_OK;
    hipStream_t st = g_lib.stream;
    const size_t nodeCount = (size_t)chartFirst[chartCount];
    std::vector<int32_t> nodeChart(nodeCount);
    for (int32_t c = 0; c < chartCount; ++c) for (int32_t i = chartFirst[c]; i < chartFirst[c + 1]; ++i) nodeChart[(size_t)i] = c;
    DevBuf nodeBuf, chartBuf, sizeBuf, offBuf, maskBuf;
    if ((rc = nodeBuf.alloc(sizeof(PimChartNode) * nodeCount))) return rc;
    if ((rc = chartBuf.alloc(sizeof(int32_t) * nodeCount))) return rc;
    if ((rc = sizeBuf.alloc(8 * (size_t)chartCount))) return rc;
    if ((rc = offBuf.alloc(8 * (size_t)chartCount))) return rc;
    if ((rc = maskBuf.alloc((size_t)total))) return rc;
    PT_HIP(hipMemcpyAsync(nodeBuf.p, nodes, sizeof(PimChartNode) * nodeCount, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(chartBuf.p, nodeChart.data(), sizeof(int32_t) * nodeCount, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(sizeBuf.p, maskSizes, 8 * (size_t)chartCount, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(offBuf.p, maskOffsets, 8 * (size_t)chartCount, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemsetAsync(maskBuf.p, 0, (size_t)total, st));
    ChartMaskArgs a;
    a.nodes = (const PimChartNode*)nodeBuf.p; a.nodeChart = (const int32_t*)chartBuf.p; a.nodeCount = (int32_t)nodeCount;
    a.maskSizes = (const int32_t*)sizeBuf.p; a.maskOffsets = (const int64_t*)offBuf.p; a.masks = (uint8_t*)maskBuf.p;
    const int blocks = (int)std::min<int64_t>(((int64_t)nodeCount + 3) / 4, (int64_t)g_lib.cuCount * 64);
    k_chart_masks<<<blocks, 256, 0, st>>>(a);
    PT_HIP(hipGetLastError());
    /* masks packed one after the other (the usual layout) come back in one copy; a layout with gaps chart by chart, so
     * that only the bytes the call owns are written */
    bool packed = maskOffsets[0] == 0;
    for (int32_t c = 1; c < chartCount && packed; ++c)
        packed = maskOffsets[c] == maskOffsets[c - 1] + (int64_t)maskSizes[2 * (c - 1)] * maskSizes[2 * (c - 1) + 1];
    if (packed) PT_HIP(hipMemcpyAsync(masks, maskBuf.p, (size_t)total, hipMemcpyDeviceToHost, st));
    else for (int32_t c = 0; c < chartCount; ++c)
    {
        const size_t bytes = (size_t)maskSizes[2 * c] * (size_t)maskSizes[2 * c + 1];
        if (bytes) PT_HIP(hipMemcpyAsync(masks + maskOffsets[c], (uint8_t*)maskBuf.p + maskOffsets[c], bytes, hipMemcpyDeviceToHost, st));
    }
    PT_HIP(hipStreamSynchronize(st));
    return PIMGPU_OK;
}


int pimgpu_lightmap_atlas_place(const uint8_t* masks, const int64_t* maskOffsets, const int32_t* maskSizes, const float* areas,
    const int32_t* order, int32_t chartCount, int32_t atlasSize, int32_t atlasCount, int32_t atlasCap, int32_t* translations, int32_t* atlasIndex, int32_t* usedAtlases)
{
    if (usedAtlases) *usedAtlases = 0;
    int rc = check_ready(); if (rc) return rc;
    if (atlasCap < atlasCount) atlasCap = atlasCount;
    if (chartCount < 0 || atlasSize <= 0 || atlasSize > 16384 || atlasCount <= 0 || atlasCap > (1 << 20) || (chartCount > 0 && (!masks || !maskOffsets || !maskSizes || !areas || !order || !translations || !atlasIndex)))
    { set_error("lightmap_atlas_place: bad argument"); return PIMGPU_E_INVALID; }
    if (chartCount == 0) return PIMGPU_OK;
    int64_t total = 0;
    std::vector<uint8_t> seen((size_t)chartCount, 0);
    for (int32_t c = 0; c < chartCount; ++c)
    {
        const int64_t sx = maskSizes[2 * c], sy = maskSizes[2 * c + 1];
        if (sx <= 0 || sy <= 0 || sx > 65536 || sy > 65536 || maskOffsets[c] < 0) { set_error("lightmap_atlas_place: chart %d has a bad mask size or offset", c); return PIMGPU_E_INVALID; }
        total = std::max<int64_t>(total, maskOffsets[c] + sx * sy);
        if (order[c] < 0 || order[c] >= chartCount || seen[(size_t)order[c]]) { set_error("lightmap_atlas_place: order[] is not a permutation of the charts"); return PIMGPU_E_INVALID; }
        seen[(size_t)order[c]] = 1;
    }
    hipStream_t st = g_lib.stream;
    const size_t aBytes = (size_t)atlasSize * (size_t)atlasSize;
    /* any covered texel is 0xff (mask_tri, lightmap.c:255): the bit form of the search relies on it */
    {
        bool bad = false;
        for (int64_t k = 0; k < total; ++k) { const uint8_t b = masks[k]; bad |= (b != 0) & (b != 0xff); }
        if (bad) { set_error("lightmap_atlas_place: a mask holds a byte that is neither 0 nor 0xff"); return PIMGPU_E_INVALID; }
    }
    const bool asBits = atlasSize <= kAtlasBitsMaxSize && !getenv("PIMGPU_ATLAS_BYTES");      /* the byte form: larger atlases, and A/B runs of the tests */
    const size_t rowWords = ((size_t)atlasSize + 63) / 64, atlasWords = (size_t)atlasSize * rowWords;
    /* the bit form takes the charts in placement order: descriptors and the offsets of their bit rows */
    std::vector<ChartDesc> desc;
    uint64_t bitWords = 0;
    if (asBits)
    {
        desc.resize((size_t)chartCount);
        for (int32_t i = 0; i < chartCount; ++i)
        {
            const int32_t c = order[i];
            ChartDesc& d = desc[(size_t)i];
            d.bw = maskSizes[2 * c]; d.bh = maskSizes[2 * c + 1]; d.area = areas[c]; d.wordOff = (uint32_t)bitWords;
            bitWords += (uint64_t)d.bh * (uint64_t)((d.bw + 63) >> 6);
            if (bitWords >= (1ull << 32)) { set_error("lightmap_atlas_place: the charts' masks exceed 2^32 words of 64 texels"); return PIMGPU_E_LIMIT; }
        }
    }
    DevBuf maskBuf, offBuf, sizeBuf, areaBuf, orderBuf, atlasBuf, outBuf, descBuf, bitBuf;
    if ((rc = maskBuf.alloc((size_t)total))) return rc;
    if ((rc = offBuf.alloc(8 * (size_t)chartCount))) return rc;
    if ((rc = sizeBuf.alloc(8 * (size_t)chartCount))) return rc;
    if ((rc = areaBuf.alloc(4 * (size_t)chartCount))) return rc;
    if ((rc = orderBuf.alloc(4 * (size_t)chartCount))) return rc;
    const size_t atlasBytes = asBits ? 8 * atlasWords * (size_t)atlasCap : aBytes * (size_t)atlasCap;
    if ((rc = atlasBuf.alloc(atlasBytes))) return rc;
    if (asBits && ((rc = descBuf.alloc(sizeof(ChartDesc) * (size_t)chartCount)) || (rc = bitBuf.alloc(8 * (size_t)bitWords)))) return rc;
    const size_t outInts = 3 * (size_t)chartCount + (size_t)atlasCap + 4;
    if ((rc = outBuf.alloc(4 * outInts))) return rc;
    PT_HIP(hipMemcpyAsync(maskBuf.p, masks, (size_t)total, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(offBuf.p, maskOffsets, 8 * (size_t)chartCount, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(sizeBuf.p, maskSizes, 8 * (size_t)chartCount, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(areaBuf.p, areas, 4 * (size_t)chartCount, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(orderBuf.p, order, 4 * (size_t)chartCount, hipMemcpyHostToDevice, st));
    if (asBits) PT_HIP(hipMemcpyAsync(descBuf.p, desc.data(), sizeof(ChartDesc) * (size_t)chartCount, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemsetAsync(atlasBuf.p, 0, atlasBytes, st));
    PT_HIP(hipMemsetAsync(outBuf.p, 0xff, 4 * outInts, st));            /* -1: not placed / no failure */
    AtlasArgs a;
    a.masks = (const uint8_t*)maskBuf.p; a.maskOffsets = (const int64_t*)offBuf.p; a.maskSizes = (const int32_t*)sizeBuf.p;
    a.areas = (const float*)areaBuf.p; a.order = (const int32_t*)orderBuf.p; a.chartCount = chartCount; a.atlasSize = atlasSize; a.atlasCount = atlasCount; a.atlasCap = atlasCap;
    a.atlases = (uint8_t*)atlasBuf.p;
    int32_t* out = (int32_t*)outBuf.p;
    a.translations = out; a.atlasIndex = out + 2 * (size_t)chartCount; a.atlasCharts = out + 3 * (size_t)chartCount; a.status = a.atlasCharts + atlasCap;
    PT_HIP(hipMemsetAsync(a.atlasCharts, 0, 4 * (size_t)atlasCap, st));
    if (asBits)
    {
        AtlasBitArgs g; g.a = a; g.atlasBits = (unsigned long long*)atlasBuf.p; g.desc = (const ChartDesc*)descBuf.p; g.bits = (const unsigned long long*)bitBuf.p;
        k_chart_bits<<<(uint32_t)(((size_t)chartCount * 64 + 255) / 256), 256, 0, st>>>(a.masks, a.maskOffsets, a.order, g.desc, chartCount, (unsigned long long*)bitBuf.p);
        k_atlas_place_bits<<<1, kAtlasBitThreads, 0, st>>>(g);
    }
    else k_atlas_place<<<1, kAtlasThreads, 0, st>>>(a);
    PT_HIP(hipGetLastError());
    std::vector<int32_t> host(outInts);
    PT_HIP(hipMemcpyAsync(host.data(), out, 4 * outInts, hipMemcpyDeviceToHost, st));
    PT_HIP(hipStreamSynchronize(st));
    if (asBits && getenv("PIMGPU_ATLAS_LOG")) fprintf(stderr, "atlas_place: %d charts, %d rounds of the scan, %d atlas swaps\n", chartCount, host[3 * (size_t)chartCount + (size_t)atlasCap + 2], host[3 * (size_t)chartCount + (size_t)atlasCap + 3]);
    int32_t failedChart = host[3 * (size_t)chartCount + (size_t)atlasCap];
    if (asBits && failedChart >= 0) failedChart = order[failedChart];          /* the bit form speaks in placement order */
    if (failedChart >= 0)
    {
        set_error("lightmap_atlas_place: chart %d (mask %d x %d) fits in none of the %d atlases of %d^2, %d more allowed (the reference's AtlasFn does not return here, lightmap.c:808-818)",
            failedChart, maskSizes[2 * failedChart], maskSizes[2 * failedChart + 1], host[3 * (size_t)chartCount + (size_t)atlasCap + 1], atlasSize,
            atlasCap - host[3 * (size_t)chartCount + (size_t)atlasCap + 1]);
        return PIMGPU_E_LIMIT;
    }
    if (asBits)
    {
        for (int32_t i = 0; i < chartCount; ++i)
        {
            const int32_t c = order[i];
            translations[2 * c] = host[2 * (size_t)i]; translations[2 * c + 1] = host[2 * (size_t)i + 1];
            atlasIndex[c] = host[2 * (size_t)chartCount + (size_t)i];
        }
    }
    else
    {
        memcpy(translations, host.data(), 8 * (size_t)chartCount);
        memcpy(atlasIndex, host.data() + 2 * (size_t)chartCount, 4 * (size_t)chartCount);
    }
    int32_t used = 0;
    for (int32_t i = 0; i < atlasCap; ++i) if (host[3 * (size_t)chartCount + (size_t)i] > 0) ++used;        /* atlases_create, lightmap.c:862-870 */
    if (usedAtlases) *usedAtlases = used;
    return PIMGPU_OK;
}

} /* extern "C" */
