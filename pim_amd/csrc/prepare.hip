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

} /* namespace pt */

using namespace pt;

namespace {

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

} /* extern "C" */
