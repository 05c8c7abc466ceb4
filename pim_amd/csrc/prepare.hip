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

} /* extern "C" */
