/*
 * lightmap.hip -- LmPack_Bake / BakeFn (rendering/lightmap.c:1117-1223, "lm.c" below) on the
 * wavefront integrator.
 *
 * The reference gives each texel to one host thread, which loops over spp samples: hemisphere
 * direction, jittered origin, Pt_TraceRay, SG_Accumulate.  The spherical-gaussian fit is a
 * sequential recurrence per texel (math/sphgauss.c:19-58), so samples of one texel stay in order:
 * here the host loops over the sample index and every iteration is one whole-batch pass
 *      bake_raygen (one lane per active texel) -> bounce loop (integrate.hip) -> bake_accumulate.
 * A texel owns two Prng streams, as in the reference: the bake stream (hemisphere +
 * jitter) and the path tracer's stream, which carries on from sample to sample.
 *
 * Pages are the reference's single allocation per lightmap (lm.c:81-110):
 *   [5][S*S] float4 probes | [S*S] float3 position | [S*S] float3 normal | [S*S] float sampleCounts
 */
#include "pt_host.h"
#include "pt_api.h"
#include "pt_scene_fn.h"
#include <cstring>
#include <algorithm>

namespace pt {

int integrate_paths(PimScene* scene, int32_t n, int32_t maxBounces, PimTraceStats* statsOut, float* traceMsOut);

struct BakeArgs
{
    float* const* pages;         /* device pointers of the pages' blocks, indexed by iLightmap - pageBase */
    int32_t pageBase;
    int32_t lmSize;
    int64_t workBegin;           /* first work item (iLightmap * lmSize^2 + iTexel) of this chunk */
    int32_t workCount;
    float metersPerTexel, timeSlice;
    uint32_t passIndex, seed;
    int64_t* activeList;         /* work item of every active slot */
    int32_t* activeCount;
    uint4* bakeRng;              /* per slot: the bake stream */
    float4* bakeRd;              /* per slot: direction of the current sample */
    int32_t tileFirst, tileStep; /* multi-device: this context bakes the 64x64-texel tiles t with t % tileStep == tileFirst (1 step: all) */
};

static constexpr int32_t kBakeTile = 64;     /* SURVEY 8e / BASELINE config 4: 64x64-texel tiles round-robin over the devices */

/* kGiAxii, rendering/lightmap.h:14-21 */
__device__ __forceinline__ v4 gi_axis(int i)
{
    const float k = 0.577350f;
    const float x = (i == 0) ? 0.0f : ((i == 1 || i == 3) ? k : -k);
    const float y = (i == 0) ? 0.0f : ((i == 1 || i == 2) ? k : -k);
    const float z = (i == 0) ? 1.0f : k;
    return f4_v(x, y, z, 4.999773f);
}

struct TexelRef { float* probes; const float* position; const float* normal; float* sampleCounts; int32_t iTexel; int64_t lmLen; };

__device__ __forceinline__ TexelRef texel_ref(const BakeArgs& b, int64_t iWork)
{
    TexelRef t;
    t.lmLen = (int64_t)b.lmSize * b.lmSize;
    const int32_t iLightmap = (int32_t)(iWork / t.lmLen);
    t.iTexel = (int32_t)(iWork % t.lmLen);
    float* block = b.pages[iLightmap - b.pageBase];
    t.probes = block;
    t.position = block + 5 * 4 * t.lmLen;
    t.normal = t.position + 3 * t.lmLen;
    t.sampleCounts = (float*)(t.normal + 3 * t.lmLen);
    return t;
}

/* BakeFn's prologue, lm.c:1139-1153: unmapped texels and the time-slice test */
__global__ __launch_bounds__(256) void k_bake_begin(BakeArgs b, PathState ps)
{
    const int32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = false;
    int64_t iWork = 0;
    u4 rng = { 0, 0, 0, 0 }, prng = { 0, 0, 0, 0 };
    if (w < b.workCount)
    {
        iWork = b.workBegin + w;
        const TexelRef t = texel_ref(b, iWork);
        const int32_t iLightmap = (int32_t)(iWork / t.lmLen);
        bool mine = true;
        if (b.tileStep > 1)
        {
            const int32_t perRow = (b.lmSize + kBakeTile - 1) / kBakeTile;
            const int32_t tx = (t.iTexel % b.lmSize) / kBakeTile, ty = (t.iTexel / b.lmSize) / kBakeTile;
            const int64_t tile = ((int64_t)iLightmap * perRow + ty) * perRow + tx;
            mine = (int32_t)(tile % b.tileStep) == b.tileFirst;
        }
        if (mine && t.sampleCounts[t.iTexel] != 0.0f)
        {
            rng = pcg4((uint32_t)t.iTexel, (uint32_t)iLightmap, b.passIndex, b.seed ^ PT_TAG_BAKE);
            prng = pcg4((uint32_t)t.iTexel, (uint32_t)iLightmap, b.passIndex, b.seed ^ PT_TAG_BAKE_PT);
            active = !(rng_f32(&rng) > b.timeSlice);
        }
    }
    const int lane = threadIdx.x & 63;
    const uint64_t ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint64_t m = __ballot(active);
    if (m)
    {
        const int leader = __ffsll((long long)m) - 1;
        int32_t q0 = 0;
        if (lane == leader) q0 = atomicAdd(b.activeCount, (int32_t)__popcll(m));
        q0 = __shfl(q0, leader, 64);
        if (active)
        {
            const int32_t j = q0 + (int32_t)__popcll(m & ltMask);
            b.activeList[j] = iWork;
            b.bakeRng[j] = make_uint4(rng.x, rng.y, rng.z, rng.w);
            ps.rng[j] = make_uint4(prng.x, prng.y, prng.z, prng.w);
        }
    }
}

__device__ __forceinline__ void path_start_bake(const PathState& ps, float4* extO, float4* extD, int32_t* queue, int32_t slot, v4 ro, v4 rd)
{
    Ctx ctx = ctx_make();
    { const uint4 r = ps.rng[slot]; ctx.rng.x = r.x; ctx.rng.y = r.y; ctx.rng.z = r.z; ctx.rng.w = r.w; }
    /* Pt_TraceRay's first roulette draw, pt.c:2321-2331 (attenuation = 1: always survives) */
    v4 attenuation = f4_s(1.0f);
    const float p = f1_sat(f4_avglum(attenuation));
    if (rng_f32(&ctx.rng) < p) attenuation = f4_divs(attenuation, p);
    ps.rng[slot] = make_uint4(ctx.rng.x, ctx.rng.y, ctx.rng.z, ctx.rng.w);
    ps.ro[slot] = make_float4(ro.x, ro.y, ro.z, 0.0f);
    ps.rd[slot] = make_float4(rd.x, rd.y, rd.z, 0.0f);
    ps.atten[slot] = make_float4(attenuation.x, attenuation.y, attenuation.z, 0.0f);
    ps.lum[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
    ps.albedo[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    ps.normal[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    extO[slot] = make_float4(ro.x, ro.y, ro.z, 0.0f);
    extD[slot] = make_float4(rd.x, rd.y, rd.z, PT_RCP_EPS);
    queue[slot] = slot;
}

/* one sample of BakeFn's loop up to Pt_TraceRay, lm.c:1174-1187 */
__global__ __launch_bounds__(256) void k_bake_raygen(BakeArgs b, PathState ps, float4* extO, float4* extD, int32_t* queue, int32_t n)
{
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const TexelRef t = texel_ref(b, b.activeList[j]);
    u4 rng; { const uint4 r = b.bakeRng[j]; rng.x = r.x; rng.y = r.y; rng.z = r.z; rng.w = r.w; }
    const v4 N = f4_normalize3(f4_v(t.normal[t.iTexel * 3], t.normal[t.iTexel * 3 + 1], t.normal[t.iTexel * 3 + 2], 0.0f));
    const v4 P = f4_add(f4_v(t.position[t.iTexel * 3], t.position[t.iTexel * 3 + 1], t.position[t.iTexel * 3 + 2], 1.0f), f4_muls(N, PT_MILLI));
    const m33 TBN = NormalToTBN(N);
    const v4 Lts = SampleUnitHemisphere(rng_f2(&rng));
    const v4 rd = TbnToWorld(TBN, Lts);
    const float dt = (rng_f32(&rng) - 0.5f) * b.metersPerTexel;
    const float db = (rng_f32(&rng) - 0.5f) * b.metersPerTexel;
    v4 ro = P;
    ro = f4_add(ro, f4_muls(TBN.c0, dt));
    ro = f4_add(ro, f4_muls(TBN.c1, db));
    b.bakeRng[j] = make_uint4(rng.x, rng.y, rng.z, rng.w);
    b.bakeRd[j] = make_float4(rd.x, rd.y, rd.z, rd.w);
    path_start_bake(ps, extO, extD, queue, j, ro, rd);
}

/* the rest of the sample, lm.c:1188-1192: progressive SG fit */
__global__ __launch_bounds__(256) void k_bake_accumulate(BakeArgs b, const PimResult* __restrict__ results, int32_t n)
{
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const TexelRef t = texel_ref(b, b.activeList[j]);
    const v4 N = f4_normalize3(f4_v(t.normal[t.iTexel * 3], t.normal[t.iTexel * 3 + 1], t.normal[t.iTexel * 3 + 2], 0.0f));
    const m33 TBN = NormalToTBN(N);
    v4 probes[5], axii[5];
    for (int i = 0; i < 5; ++i)
    {
        const float* pr = t.probes + ((int64_t)i * t.lmLen + t.iTexel) * 4;
        probes[i] = f4_v(pr[0], pr[1], pr[2], pr[3]);
        v4 ax = gi_axis(i);
        const float sharpness = ax.w;
        ax = TbnToWorld(TBN, ax);
        ax.w = sharpness;
        axii[i] = ax;
    }
    const float4 d4 = b.bakeRd[j];
    const v4 rd = f4_v(d4.x, d4.y, d4.z, d4.w);
    const PimResult r = results[j];
    float sampleCount = t.sampleCounts[t.iTexel];
    const float weight = 1.0f / sampleCount;
    sampleCount += 1.0f;
    SG_Accumulate(weight, rd, f4_v(r.color[0], r.color[1], r.color[2], 0.0f), axii, probes, 5);
    for (int i = 0; i < 5; ++i)
    {
        float* pr = t.probes + ((int64_t)i * t.lmLen + t.iTexel) * 4;
        pr[0] = probes[i].x; pr[1] = probes[i].y; pr[2] = probes[i].z; pr[3] = probes[i].w;
    }
    t.sampleCounts[t.iTexel] = sampleCount;
}

} /* namespace pt */

using namespace pt;

/* pages that stay in HBM between calls (pimgpu_lightmap_set_*): one block per page in the reference's layout; with several
 * contexts every context holds a replica and keeps ITS tiles of it up to date */
struct PimLightmapSet
{
    int32_t lmCount = 0, lmSize = 0;
    std::vector<float*> dev;
    PimLightmapSet* peers[kMaxDevices] = {};
};

/* one context bakes its tiles of the work items [texelBegin, texelEnd).  outPages == pages: results go straight to the caller's
 * pages; else (a peer context) into outPages, page blocks of probes | sampleCounts only, from which the caller merges this
 * context's tiles. */
static int lightmap_bake_one(PimScene* scene, PimLightmapPage* pages, int32_t lmSize, float texelsPerMeter,
    float timeSlice, int32_t spp, uint32_t passIndex, uint32_t seed, int32_t maxBounces, int64_t texelBegin, int64_t texelEnd,
    int32_t tileFirst, int32_t tileStep, float* const* outProbes, float* const* outCounts, float* const* resident = nullptr)
{
    int rc = PIMGPU_OK;
    const int64_t lmLen = (int64_t)lmSize * lmSize;
    hipStream_t st = g_lib.stream;

    const int32_t pageBase = (int32_t)(texelBegin / lmLen);
    const int32_t pageLast = (int32_t)((texelEnd - 1) / lmLen);
    const int32_t pageCount = pageLast - pageBase + 1;
    const size_t pageFloats = (size_t)lmLen * (5 * 4 + 3 + 3 + 1);
    std::vector<float*> devPages((size_t)pageCount, nullptr);
    float** devPageTable = nullptr; int64_t* activeList = nullptr; int32_t* activeCount = nullptr; uint4* bakeRng = nullptr; float4* bakeRd = nullptr;
    /* resident != nullptr: the pages already live in HBM (a PimLightmapSet, indexed by page): nothing is copied in or out */
    auto cleanup = [&]() { if (!resident) for (float* p : devPages) hipFree(p); hipFree(devPageTable); hipFree(activeList); hipFree(activeCount); hipFree(bakeRng); hipFree(bakeRd); };
    auto oom = [&]() { cleanup(); set_error("lightmap_bake: out of device memory"); return PIMGPU_E_NOMEM; };
    for (int32_t i = 0; i < pageCount; ++i)
    {
        if (resident) { devPages[(size_t)i] = resident[pageBase + i]; continue; }
        if (!pages[pageBase + i].block) { cleanup(); set_error("lightmap_bake: page %d has no block", pageBase + i); return PIMGPU_E_INVALID; }
        if (hipMalloc(&devPages[(size_t)i], sizeof(float) * pageFloats) != hipSuccess) return oom();
        hipMemcpyAsync(devPages[(size_t)i], pages[pageBase + i].block, sizeof(float) * pageFloats, hipMemcpyHostToDevice, st);
    }
    if (hipMalloc(&devPageTable, sizeof(float*) * (size_t)pageCount) != hipSuccess) return oom();
    hipMemcpyAsync(devPageTable, devPages.data(), sizeof(float*) * (size_t)pageCount, hipMemcpyHostToDevice, st);

    const int64_t kChunk = 1ll << 24;        /* work items per pass: bounds the path-state footprint (~4 GB) */
    const int64_t chunkCap = std::min<int64_t>(kChunk, texelEnd - texelBegin);
    if (hipMalloc(&activeList, sizeof(int64_t) * (size_t)chunkCap) != hipSuccess || hipMalloc(&activeCount, 64) != hipSuccess ||
        hipMalloc(&bakeRng, sizeof(uint4) * (size_t)chunkCap) != hipSuccess || hipMalloc(&bakeRd, sizeof(float4) * (size_t)chunkCap) != hipSuccess) return oom();
    WorkBuffers& w = g_lib.work;
    if ((rc = w.ensure_paths(chunkCap))) { cleanup(); return rc; }

    for (int64_t begin = texelBegin; begin < texelEnd && !rc; begin += kChunk)
    {
        BakeArgs b;
        b.pages = devPageTable; b.pageBase = pageBase; b.lmSize = lmSize; b.workBegin = begin;
        b.workCount = (int32_t)std::min<int64_t>(kChunk, texelEnd - begin);
        b.metersPerTexel = 1.0f / texelsPerMeter; b.timeSlice = timeSlice; b.passIndex = passIndex; b.seed = seed;
        b.activeList = activeList; b.activeCount = activeCount; b.bakeRng = bakeRng; b.bakeRd = bakeRd;
        b.tileFirst = tileFirst; b.tileStep = tileStep;
        hipMemsetAsync(activeCount, 0, 4, st);
        k_bake_begin<<<(b.workCount + 255) / 256, 256, 0, st>>>(b, w.ps);
        int32_t nActive = 0;
        hipMemcpyAsync(&nActive, activeCount, 4, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("lightmap_bake: device error"); rc = PIMGPU_E_DEVICE; break; }
        if (nActive == 0) continue;
        for (int32_t s = 0; s < spp && !rc; ++s)
        {
            k_bake_raygen<<<(nActive + 255) / 256, 256, 0, st>>>(b, w.ps, w.extO[0], w.extD[0], w.queue[0], nActive);
            rc = integrate_paths(scene, nActive, maxBounces, nullptr, nullptr);
            if (!rc) k_bake_accumulate<<<(nActive + 255) / 256, 256, 0, st>>>(b, w.results, nActive);
        }
    }
    if (!rc && resident) { if (hipStreamSynchronize(st) != hipSuccess) { set_error("lightmap_bake: device error"); rc = PIMGPU_E_DEVICE; } }
    else if (!rc)
    {
        /* probes and sampleCounts are the only parts BakeFn writes (lm.c:1194-1200) */
        for (int32_t i = 0; i < pageCount; ++i)
        {
            float* host = pages[pageBase + i].block; float* dev = devPages[(size_t)i];
            float* hostProbes = outProbes ? outProbes[pageBase + i] : host;
            float* hostCounts = outCounts ? outCounts[pageBase + i] : host + (size_t)lmLen * 26;
            hipMemcpyAsync(hostProbes, dev, sizeof(float) * (size_t)lmLen * 20, hipMemcpyDeviceToHost, st);
            hipMemcpyAsync(hostCounts, dev + (size_t)lmLen * 26, sizeof(float) * (size_t)lmLen, hipMemcpyDeviceToHost, st);
        }
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("lightmap_bake: device error"); rc = PIMGPU_E_DEVICE; }
    }
    cleanup();
    return rc;
}

static void merge_tiles(float* dstPage, const float* srcProbes, const float* srcCounts, int32_t pg, int32_t lmSize, int owner, int G);

extern "C" int pimgpu_lightmap_bake(PimScene* scene, PimLightmapPage* pages, int32_t lmCount, int32_t lmSize, float texelsPerMeter,
    float timeSlice, int32_t spp, uint32_t passIndex, uint32_t seed, int32_t maxBounces, int64_t texelBegin, int64_t texelEnd)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || lmCount < 0 || lmSize <= 0 || (lmCount > 0 && !pages) || !(texelsPerMeter > 0.0f))
    { set_error("lightmap_bake: bad argument"); return PIMGPU_E_INVALID; }
    const int64_t lmLen = (int64_t)lmSize * lmSize;
    const int64_t total = (int64_t)lmCount * lmLen;
    if (texelBegin < 0) texelBegin = 0;
    if (texelEnd < 0 || texelEnd > total) texelEnd = total;
    if ((rc = scene_update_dists_all(scene))) return rc;            /* PtScene_Update, lm.c:1209 */
    if (texelEnd <= texelBegin) { PT_HIP(hipStreamSynchronize(g_lib.stream)); return PIMGPU_OK; }
    if (spp < 1) spp = 1;
    const int G = g_libCount;
    if (G <= 1 || !scene->peers[1])
        return lightmap_bake_one(scene, pages, lmSize, texelsPerMeter, timeSlice, spp, passIndex, seed, maxBounces, texelBegin, texelEnd, 0, 1, nullptr, nullptr);
    /*
     * Several devices: 64x64-texel tiles round-robin, each texel's samples in order on ONE device (SG_Accumulate is
     * order dependent, sphgauss.c:19-58).  Disjoint writes, no reduction: context 0 writes the caller's pages, the others write
     * staging copies of probes | sampleCounts from which their tiles are copied in.  A texel's result depends on its own
     * streams and the scene only, so the pages equal the one-device bake bit for bit.
     */
    const int32_t pageBase = (int32_t)(texelBegin / lmLen), pageLast = (int32_t)((texelEnd - 1) / lmLen);
    std::vector<std::vector<float*>> probes((size_t)G, std::vector<float*>((size_t)lmCount, nullptr)), counts((size_t)G, std::vector<float*>((size_t)lmCount, nullptr));
    bool okAlloc = true;
    for (int i = 1; i < G && okAlloc; ++i)
        for (int32_t pg = pageBase; pg <= pageLast && okAlloc; ++pg)
        {
            probes[(size_t)i][(size_t)pg] = (float*)pimgpu_host_alloc(sizeof(float) * (size_t)lmLen * 20);
            counts[(size_t)i][(size_t)pg] = (float*)pimgpu_host_alloc(sizeof(float) * (size_t)lmLen);
            okAlloc = probes[(size_t)i][(size_t)pg] && counts[(size_t)i][(size_t)pg];
        }
    if (okAlloc)
        rc = for_each_context(G, [&](int i) {
            return lightmap_bake_one(i == 0 ? scene : scene->peers[i], pages, lmSize, texelsPerMeter, timeSlice, spp, passIndex, seed, maxBounces,
                texelBegin, texelEnd, i, G, i == 0 ? nullptr : probes[(size_t)i].data(), i == 0 ? nullptr : counts[(size_t)i].data());
        });
    else { set_error("lightmap_bake: out of page-locked host memory"); rc = PIMGPU_E_NOMEM; }
    if (!rc)
    {
        for (int32_t pg = pageBase; pg <= pageLast; ++pg)
            for (int owner = 1; owner < G; ++owner)
                merge_tiles(pages[pg].block, probes[(size_t)owner][(size_t)pg], counts[(size_t)owner][(size_t)pg], pg, lmSize, owner, G);
    }
    for (int i = 1; i < G; ++i) for (int32_t pg = 0; pg < lmCount; ++pg) { pimgpu_host_free(probes[(size_t)i][(size_t)pg]); pimgpu_host_free(counts[(size_t)i][(size_t)pg]); }
    return rc;
}


/* multi-device: the tiles of context `owner` (64x64 texels, round-robin over G) of page pg, from that context's probes | sampleCounts
 * into the caller's page */
static void merge_tiles(float* dstPage, const float* srcProbes, const float* srcCounts, int32_t pg, int32_t lmSize, int owner, int G)
{
    const int64_t lmLen = (int64_t)lmSize * lmSize;
    const int32_t perRow = (lmSize + kBakeTile - 1) / kBakeTile;
    for (int32_t ty = 0; ty < perRow; ++ty)
        for (int32_t tx = 0; tx < perRow; ++tx)
        {
            if ((int)((((int64_t)pg * perRow + ty) * perRow + tx) % G) != owner) continue;
            const int32_t x0 = tx * kBakeTile, x1 = std::min(lmSize, x0 + kBakeTile);
            for (int32_t y = ty * kBakeTile; y < std::min(lmSize, (ty + 1) * kBakeTile); ++y)
            {
                const size_t row = (size_t)y * lmSize;
                for (int k = 0; k < 5; ++k)
                    memcpy(dstPage + ((size_t)k * lmLen + row + x0) * 4, srcProbes + ((size_t)k * lmLen + row + x0) * 4, sizeof(float) * 4 * (size_t)(x1 - x0));
                memcpy(dstPage + (size_t)lmLen * 26 + row + x0, srcCounts + row + x0, sizeof(float) * (size_t)(x1 - x0));
            }
        }
}

static thread_local bool g_makingSetPeer = false;

extern "C" int pimgpu_lightmap_set_destroy(PimLightmapSet* set)
{
    if (!set) return PIMGPU_OK;
    for (int i = 1; i < kMaxDevices; ++i)
        if (set->peers[i])
        {
            PimLightmapSet* peer = set->peers[i]; set->peers[i] = nullptr;
            if (i < g_libCount) { DeviceScope scope(i); pimgpu_lightmap_set_destroy(peer); }
        }
    if (g_lib.initialised) hipStreamSynchronize(g_lib.stream);
    for (float* p : set->dev) hipFree(p);
    delete set;
    return PIMGPU_OK;
}

extern "C" int pimgpu_lightmap_set_create(int32_t lmCount, int32_t lmSize, PimLightmapSet** out)
{
    if (out) *out = nullptr;
    int rc = check_ready(); if (rc) return rc;
    if (!out || lmCount <= 0 || lmSize <= 0 || lmSize > 16384) { set_error("lightmap_set_create: bad argument"); return PIMGPU_E_INVALID; }
    PimLightmapSet* set = new PimLightmapSet();
    set->lmCount = lmCount; set->lmSize = lmSize;
    const size_t bytes = sizeof(float) * 27 * (size_t)lmSize * lmSize;
    set->dev.assign((size_t)lmCount, nullptr);
    for (int32_t i = 0; i < lmCount; ++i)
    {
        if (hipMalloc(&set->dev[(size_t)i], bytes) != hipSuccess) { pimgpu_lightmap_set_destroy(set); set_error("lightmap_set_create: out of device memory"); return PIMGPU_E_NOMEM; }
        hipMemsetAsync(set->dev[(size_t)i], 0, bytes, g_lib.stream);
    }
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    if (g_libCount > 1 && !g_makingSetPeer)
    {
        rc = for_each_context(g_libCount, [&](int i) {
            if (i == 0) return PIMGPU_OK;
            g_makingSetPeer = true;
            const int r = pimgpu_lightmap_set_create(lmCount, lmSize, &set->peers[i]);
            g_makingSetPeer = false;
            return r;
        });
        if (rc) { pimgpu_lightmap_set_destroy(set); return rc; }
    }
    *out = set;
    return PIMGPU_OK;
}

extern "C" int pimgpu_lightmap_set_upload(PimLightmapSet* set, const PimLightmapPage* pages)
{
    int rc = check_ready(); if (rc) return rc;
    if (!set || !pages) { set_error("lightmap_set_upload: NULL argument"); return PIMGPU_E_INVALID; }
    for (int32_t i = 0; i < set->lmCount; ++i) if (!pages[i].block) { set_error("lightmap_set_upload: page %d has no block", i); return PIMGPU_E_INVALID; }
    const size_t bytes = sizeof(float) * 27 * (size_t)set->lmSize * set->lmSize;
    const int G = (g_libCount > 1 && set->peers[1]) ? g_libCount : 1;
    return for_each_context(G, [&](int c) {
        PimLightmapSet* s = c == 0 ? set : set->peers[c];
        for (int32_t i = 0; i < s->lmCount; ++i) PT_HIP(hipMemcpyAsync(s->dev[(size_t)i], pages[i].block, bytes, hipMemcpyHostToDevice, g_lib.stream));
        PT_HIP(hipStreamSynchronize(g_lib.stream));
        return PIMGPU_OK;
    });
}

extern "C" int pimgpu_lightmap_set_download(const PimLightmapSet* set, PimLightmapPage* pages)
{
    int rc = check_ready(); if (rc) return rc;
    if (!set || !pages) { set_error("lightmap_set_download: NULL argument"); return PIMGPU_E_INVALID; }
    for (int32_t i = 0; i < set->lmCount; ++i) if (!pages[i].block) { set_error("lightmap_set_download: page %d has no block", i); return PIMGPU_E_INVALID; }
    const size_t lmLen = (size_t)set->lmSize * set->lmSize;
    /* probes and sampleCounts: the parts BakeFn writes (lm.c:1194-1200); position / normal are the caller's own */
    for (int32_t i = 0; i < set->lmCount; ++i)
    {
        PT_HIP(hipMemcpyAsync(pages[i].block, set->dev[(size_t)i], sizeof(float) * lmLen * 20, hipMemcpyDeviceToHost, g_lib.stream));
        PT_HIP(hipMemcpyAsync(pages[i].block + lmLen * 26, set->dev[(size_t)i] + lmLen * 26, sizeof(float) * lmLen, hipMemcpyDeviceToHost, g_lib.stream));
    }
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    const int G = (g_libCount > 1 && set->peers[1]) ? g_libCount : 1;
    if (G <= 1) return PIMGPU_OK;
    /* the other contexts' tiles: one staging page at a time */
    float* probes = (float*)pimgpu_host_alloc(sizeof(float) * lmLen * 20);
    float* counts = (float*)pimgpu_host_alloc(sizeof(float) * lmLen);
    if (!probes || !counts) { pimgpu_host_free(probes); pimgpu_host_free(counts); set_error("lightmap_set_download: out of page-locked host memory"); return PIMGPU_E_NOMEM; }
    for (int c = 1; c < G && !rc; ++c)
        for (int32_t i = 0; i < set->lmCount && !rc; ++i)
        {
            {
                DeviceScope scope(c);
                const float* dev = set->peers[c]->dev[(size_t)i];
                if (hipMemcpyAsync(probes, dev, sizeof(float) * lmLen * 20, hipMemcpyDeviceToHost, g_lib.stream) != hipSuccess ||
                    hipMemcpyAsync(counts, dev + lmLen * 26, sizeof(float) * lmLen, hipMemcpyDeviceToHost, g_lib.stream) != hipSuccess ||
                    hipStreamSynchronize(g_lib.stream) != hipSuccess) { set_error("lightmap_set_download: device error"); rc = PIMGPU_E_DEVICE; }
            }
            if (!rc) merge_tiles(pages[i].block, probes, counts, i, set->lmSize, c, G);
        }
    pimgpu_host_free(probes); pimgpu_host_free(counts);
    return rc;
}

extern "C" int pimgpu_lightmap_set_bake(PimScene* scene, PimLightmapSet* set, float texelsPerMeter, float timeSlice, int32_t spp,
    uint32_t passIndex, uint32_t seed, int32_t maxBounces, int64_t texelBegin, int64_t texelEnd)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || !set || !(texelsPerMeter > 0.0f)) { set_error("lightmap_set_bake: bad argument"); return PIMGPU_E_INVALID; }
    const int64_t total = (int64_t)set->lmCount * set->lmSize * set->lmSize;
    if (texelBegin < 0) texelBegin = 0;
    if (texelEnd < 0 || texelEnd > total) texelEnd = total;
    if ((rc = scene_update_dists_all(scene))) return rc;            /* PtScene_Update, lm.c:1209 */
    if (texelEnd <= texelBegin) { PT_HIP(hipStreamSynchronize(g_lib.stream)); return PIMGPU_OK; }
    if (spp < 1) spp = 1;
    const int G = (g_libCount > 1 && scene->peers[1] && set->peers[1]) ? g_libCount : 1;
    return for_each_context(G, [&](int c) {
        PimLightmapSet* s = c == 0 ? set : set->peers[c];
        return lightmap_bake_one(c == 0 ? scene : scene->peers[c], nullptr, s->lmSize, texelsPerMeter, timeSlice, spp, passIndex, seed, maxBounces,
            texelBegin, texelEnd, c, G, nullptr, nullptr, s->dev.data());
    });
}
