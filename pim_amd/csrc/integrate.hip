/*
 * integrate.hip -- wavefront form of Pt_TraceRay / Pt_Trace / Pt_RayGen
 * (rendering/path_tracer.c:2306-2416, 2514-2585, 2596-2644; "pt.c" below).
 *
 * The reference walks one path at a time: for b in 0..666 { russian roulette; rtcIntersect1;
 * media; surface; direct light (one more rtcIntersect1); scatter }.  Here all paths of a batch
 * advance one bounce together, and the two ray queries of a bounce are whole-batch launches of
 * the traversal kernel (trace.hip):
 *
 *   init      one lane per pixel / ray: seed the Prng stream, generate the camera ray (AA + DOF),
 *             draw the first roulette number, emit extension ray #0                  pt.c:2514-2553
 *   loop b:   trace(ext)  -> shade_a: miss/sky, media, surface, emission, light-grid feedback,
 *                            first half of EstimateDirect -> emits the NEE ray       pt.c:2333-2388
 *             trace(nee)  -> shade_b: second half of EstimateDirect, Scatter_Principled, AOVs,
 *                            roulette of bounce b+1, wave-ballot compaction of the
 *                            survivors into the next queue + their extension rays    pt.c:2389-2407,2319-2331
 *   resolve   running-mean accumulate into the PtTrace images                        pt.c:2547-2552
 *
 * Every path owns its Prng stream and the stages draw from it in the reference's order
 * (SURVEY A.2), so the result of a path does not depend on how paths are batched.  Queues hold
 * path slots; rays and hits are stored by QUEUE POSITION so the traversal kernel reads and
 * writes them coalesced.  Queue lengths live in device memory: no host round trip per bounce.
 * The rare rays in the middle of a stage (media in-scatter light ray, refraction thickness ray)
 * are traced inline by the lane that needs them.
 */
#include "pt_host.h"
#include "pt_api.h"
#include "pt_scene_fn.h"

#include <hip/hip_ext.h>
#include <cstring>
#include <cstdlib>

namespace pt {

static constexpr int kShadeThreads = 512;
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 4      /* waves per SIMD the FAST shade builds are compiled for (<= 128 VGPRs) */
#endif

/* nee0.x modes between shade_a and shade_b */
#define PT_NEE_NONE   0     /* no direct-light ray this bounce */
#define PT_NEE_LIGHT  1     /* light-sampling branch of EstimateDirect */
#define PT_NEE_BRDF   2     /* BRDF-sampling branch */
#define PT_NEE_MEDIA  3     /* the medium scattered the path: bounce already applied */
#define PT_NEE_DEAD   4     /* path finished in shade_a */

/* counters[]: [0],[1] queue lengths, [2],[3] slow-list lengths, [PT_COUNTS_BASE + b] queue length at the start of bounce b
 * (read back with the statistics: the host's tail policy of the next call) */
#define PT_PARKED      5     /* counters[]: paths parked for the tail kernel */
#define PT_PARK_COUNTED 0x40000000   /* in a parked ray's bounce word: its extension ray is already in the statistics */
#define PT_COUNTS_BASE 8
#define PT_COUNTS_MAX  56

void WorkBuffers::release()
{
    hipFree(rayO); hipFree(rayD);
    hipFree(ps.rng); hipFree(ps.ro); hipFree(ps.rd); hipFree(ps.atten); hipFree(ps.lum); hipFree(ps.albedo); hipFree(ps.normal);
    hipFree(ps.nee0); hipFree(ps.nee1);
    for (int i = 0; i < 2; ++i) { hipFree(extO[i]); hipFree(extD[i]); hipFree(queue[i]); }
    hipFree(neeO); hipFree(neeD); hipFree(extHit); hipFree(neeHit); hipFree(neeTarget); hipFree(parkSlot); hipFree(parkO); hipFree(parkD); hipFree(counters); hipFree(stats); hipFree(results); hipFree(slowA); hipFree(slowB);
    *this = WorkBuffers();
}

int WorkBuffers::ensure_rays(int64_t n)
{
    if (n <= rayCap) return PIMGPU_OK;
    hipFree(rayO); hipFree(rayD); rayO = rayD = nullptr; rayCap = 0;
    if (hipMalloc(&rayO, sizeof(float4) * n) != hipSuccess || hipMalloc(&rayD, sizeof(float4) * n) != hipSuccess)
    { set_error("out of device memory for %lld rays", (long long)n); return PIMGPU_E_NOMEM; }
    rayCap = n;
    return PIMGPU_OK;
}

int WorkBuffers::ensure_paths(int64_t n)
{
    if (!counters)
    {
        PT_HIP(hipMalloc(&counters, 256));
        PT_HIP(hipMalloc(&stats, 256));
    }
    if (n <= pathCap) return PIMGPU_OK;
    const int64_t keepRay = rayCap; float4* kO = rayO; float4* kD = rayD;
    int32_t* kc = counters; unsigned long long* ks = stats;
    rayO = rayD = nullptr; counters = nullptr; stats = nullptr;
    release();
    rayO = kO; rayD = kD; rayCap = keepRay; counters = kc; stats = ks;
    bool ok = true;
    auto A = [&](auto** p, size_t elem) { if (ok && hipMalloc((void**)p, elem * (size_t)n) != hipSuccess) ok = false; };
    A(&ps.rng, sizeof(uint4)); A(&ps.ro, sizeof(float4)); A(&ps.rd, sizeof(float4)); A(&ps.atten, sizeof(float4));
    A(&ps.lum, sizeof(float4)); A(&ps.albedo, sizeof(float4)); A(&ps.normal, sizeof(float4));
    A(&ps.nee0, sizeof(float4)); A(&ps.nee1, sizeof(float4));
    for (int i = 0; i < 2; ++i) { A(&extO[i], sizeof(float4)); A(&extD[i], sizeof(float4)); A(&queue[i], sizeof(int32_t)); }
    A(&slowA, sizeof(int32_t)); A(&slowB, sizeof(int32_t)); A(&neeO, sizeof(float4)); A(&neeD, sizeof(float4)); A(&extHit, sizeof(DHit)); A(&neeHit, sizeof(DHit)); A(&neeTarget, sizeof(int32_t)); A(&parkSlot, sizeof(int32_t)); A(&parkO, sizeof(float4)); A(&parkD, sizeof(float4)); A(&results, sizeof(PimResult));
    if (!ok) { set_error("out of device memory for %lld paths", (long long)n); return PIMGPU_E_NOMEM; }
    pathCap = n;
    return PIMGPU_OK;
}

__global__ __launch_bounds__(256) void k_pack_rays(const float4* __restrict__ ro, const float4* __restrict__ rd,
    const float* __restrict__ tNear, const float* __restrict__ tFar, int32_t n, float4* __restrict__ rayO, float4* __restrict__ rayD)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 o = ro[i], d = rd[i];
    rayO[i] = make_float4(o.x, o.y, o.z, tNear ? tNear[i] : 0.0f);
    rayD[i] = make_float4(d.x, d.y, d.z, tFar ? tFar[i] : PT_RCP_EPS);
}

int pack_rays(const float4* ro, const float4* rd, const float* tNear, const float* tFar, int32_t n, float4* rayO, float4* rayD, hipStream_t stream)
{
    k_pack_rays<<<(n + 255) / 256, 256, 0, stream>>>(ro, rd, tNear, tFar, n, rayO, rayD);
    PT_HIP(hipGetLastError());
    return PIMGPU_OK;
}

/* ---------------------------------------------------------------------------------------------- */
/* device helpers                                                                                  */

struct StageArgs
{
    PathState ps;
    const int32_t* queueIn; int32_t* queueOut;
    int32_t* counters;              /* [0], [1]: lengths of queue[0], queue[1] */
    int32_t inSel;                  /* which queue is the input of this bounce */
    const float4* extO; const float4* extD;   /* extension rays of the input queue, by position (the tail kernel starts from them) */
    const DHit* extHit; const DHit* neeHit;
    float4* neeO; float4* neeD;
    DHit* neeHitOut; int32_t* neeTarget;      /* shade_a: the light triangle's own record and id for a light-sampling ray (trace.hip, TGT) */
    float4* extOOut; float4* extDOut;
    PimResult* results;
    unsigned long long* stats;
    int32_t bounce, bounceLimit;
    int32_t* slowA; int32_t* slowB; /* queue positions the FAST builds deferred (lengths: counters[2], counters[3]) */
    int32_t tailLanes;              /* k_tail: lanes of each wave that take paths */
    unsigned long long* spill;      /* k_tail: overflow area of the traversal stacks (TraceResources::spill) */
    int32_t stackCap;               /* entries a traversal stack may hold (trace_stack_cap) */
    /* paths the FAST shade_a cannot finish (a medium collision) are PARKED: taken out of the wavefront with their extension ray and
     * bounce, and run from there by the tail kernel at the end of the call (counters[PT_PARKED]); null: listed for shade_a<SLOW> */
    int32_t* parkSlot; float4* parkO; float4* parkD;
    int32_t tailMainEmpty;          /* k_tail: only the parked paths (the wavefront finished the others) */
};

__device__ __forceinline__ v4 ld4(const float4* p, int64_t i) { const float4 v = p[i]; return f4_v(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st4(float4* p, int64_t i, v4 v) { p[i] = make_float4(v.x, v.y, v.z, v.w); }

/* the tail of Pt_TraceRay, pt.c:2409-2415 */
__device__ __forceinline__ void path_finish(const StageArgs& a, int32_t slot, const Ctx& ctx, v4 lum, float resultWeight, v4 albedoAcc, v4 normalAcc)
{
    const float sc = 1.0f / f1_max(resultWeight, PT_EPS);
    PimResult r;
    r.color[0] = lum.x; r.color[1] = lum.y; r.color[2] = lum.z;
    r.albedo[0] = albedoAcc.x * sc; r.albedo[1] = albedoAcc.y * sc; r.albedo[2] = albedoAcc.z * sc;
    r.normal[0] = normalAcc.x * sc; r.normal[1] = normalAcc.y * sc; r.normal[2] = normalAcc.z * sc;
    a.results[slot] = r;
    a.ps.rng[slot] = make_uint4(ctx.rng.x, ctx.rng.y, ctx.rng.z, ctx.rng.w);
}

/* russian roulette at the top of a bounce, pt.c:2321-2331 */
__device__ __forceinline__ bool roulette(Ctx& ctx, v4& attenuation)
{
    const float p = f1_sat(f4_avglum(attenuation));
    if (Sample1D(ctx) < p) { attenuation = f4_divs(attenuation, p); return true; }
    return false;
}

__device__ __forceinline__ void wave_add(unsigned long long* dst, uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, (unsigned long long)v);
}

/* wave-aggregated append: the lanes with `take` get consecutive positions of a device-side list (one atomic per wave) */
__device__ __forceinline__ int32_t wave_append(int32_t* counter, bool take)
{
    const int lane = threadIdx.x & 63;
    const uint64_t ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint64_t m = __ballot(take);
    int32_t q0 = 0;
    if (m)
    {
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) q0 = atomicAdd(counter, (int32_t)__popcll(m));
        q0 = __shfl(q0, leader, 64);
    }
    return q0 + (int32_t)__popcll(m & ltMask);
}

/* block-aggregated append: ONE returning atomic per workgroup (a single device-scope counter sustains only ~90
 * returning atomics per microsecond, MI355X_MICROARCH.md "dequeue": one per wave made the compaction, not the
 * shading, the cost of shade_b).  Must be called by every thread of the block; sh is 2 + blockDim/64 ints of LDS. */
__device__ __forceinline__ int32_t block_append(int32_t* counter, bool take, int32_t* sh)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint64_t ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint64_t m = __ballot(take);
    if (lane == 0) sh[2 + wave] = (int32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0)
    {
        int32_t total = 0;
        for (int w = 0; w < nw; ++w) { const int32_t c = sh[2 + w]; sh[2 + w] = total; total += c; }
        sh[0] = total ? atomicAdd(counter, total) : 0;
    }
    __syncthreads();
    const int32_t q = sh[0] + sh[2 + wave] + (int32_t)__popcll(m & ltMask);
    __syncthreads();
    return q;
}

/* block-aggregated statistics add */
__device__ __forceinline__ void block_add(unsigned long long* dst, uint32_t v, int32_t* sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) sh[2 + wave] = (int32_t)v;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        unsigned long long total = 0;
        for (int w = 0; w < nw; ++w) total += (uint32_t)sh[2 + w];
        if (total) atomicAdd(dst, total);
    }
    __syncthreads();
}

/* a fresh path in slot `slot`: Pt_TraceRay's locals (pt.c:2310-2317) + roulette of bounce 0 + extension ray 0 */
__device__ __forceinline__ void path_start(const PathState& ps, float4* extO, float4* extD, int32_t* queue, int32_t slot, Ctx& ctx, v4 ro, v4 rd)
{
    v4 attenuation = f4_s(1.0f);
    const bool alive = roulette(ctx, attenuation);   /* p == 1: always survives; the draw is consumed as in the reference */
    (void)alive;
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

/* ---------------------------------------------------------------------------------------------- */
/* init kernels                                                                                    */

struct CameraArgs
{
    v4 eye, right, up, fwd; v2 slope;
    PimDofInfo dof;
    int32_t width, height;
    uint32_t sampleIndex, seed;
    const float* focalDev;          /* non-null: dof.focalLength of this frame, left on the device by k_focus */
    int32_t pixelBegin, pixelCount; /* the band of pixels this launch renders (multi-device: rows of the image) */
    int32_t slotBegin;              /* path slot of the band's first pixel (shares of an image: pimgpu_trace_image_host) */
};

/* CalculateDof, pt.c:2418-2452 */
__device__ __forceinline__ void CalculateDof(Ctx& ctx, const PimDofInfo& dof, v4 right, v4 up, v4 fwd, v4* ro, v4* rd)
{
    v2 offset;
    {
        const uint32_t side = rng_u32(&ctx.rng);
        const v2 Xi = Sample2D(ctx);
        if (dof.bladeCount == 666) offset = SamplePentagram(Xi, side);
        else offset = SampleNGon(Xi, side, (uint32_t)dof.bladeCount, dof.bladeRot);
    }
    offset = f2_muls(offset, dof.aperture);
    const float t = f1_lerp(dof.focalLength / f4_dot3(*rd, fwd), dof.focalLength, dof.focalPlaneCurvature);
    const v4 focusPos = f4_add(*ro, f4_muls(*rd, t));
    const v4 aperturePos = f4_add(*ro, f4_add(f4_muls(right, offset.x), f4_muls(up, offset.y)));
    *ro = aperturePos;
    *rd = f4_normalize3(f4_sub(focusPos, aperturePos));
}

/* TraceFn up to Pt_TraceRay, pt.c:2514-2545 */
__global__ __launch_bounds__(256) void k_init_image(PathState ps, float4* extO, float4* extD, int32_t* queue, CameraArgs c)
{
    const int32_t local = blockIdx.x * blockDim.x + threadIdx.x;      /* pixel = pixelBegin + local (a band of the image), path slot = slotBegin + local */
    if (local >= c.pixelCount) return;
    const int32_t slot = c.slotBegin + local;
    const int32_t i = c.pixelBegin + local;
    const int32_t cx = i % c.width, cy = i / c.width;
    const float rcpX = 1.0f / (float)c.width, rcpY = 1.0f / (float)c.height;
    Ctx ctx = ctx_make();
    ctx.rng = pcg4((uint32_t)cx, (uint32_t)cy, c.sampleIndex, c.seed ^ PT_TAG_PIXEL);
    const v2 baseUv = f2_v((cx + 0.5f) * rcpX, (cy + 0.5f) * rcpY);
    const v2 aa = SampleGaussPixelFilter(Sample2D(ctx), 1.0f);
    const v2 rayUv = f2_add(baseUv, f2_mul(aa, f2_v(rcpX, rcpY)));
    v4 ro = c.eye;
    v4 rd = proj_dir(c.right, c.up, c.fwd, c.slope, f2_v(rayUv.x * 2.0f - 1.0f, rayUv.y * 2.0f - 1.0f));
    PimDofInfo dof = c.dof;
    if (c.focalDev) dof.focalLength = *c.focalDev;
    CalculateDof(ctx, dof, c.right, c.up, c.fwd, &ro, &rd);
    path_start(ps, extO, extD, queue, slot, ctx, ro, rd);
}

/* batched Pt_TraceRay: rays given (ro != null), RayGenFn's uniform sphere directions (pt.c:2596-2612), or the
 * tent-jittered cube-face directions of Cubemap_Bake's BakeFn (cubemap.c:138-160) when cubeSize > 0 */
__global__ __launch_bounds__(256) void k_init_rays(PathState ps, float4* extO, float4* extD, int32_t* queue,
    const float4* __restrict__ ro, const float4* __restrict__ rd, v4 origin, float4* __restrict__ directionsOut,
    int32_t n, uint32_t sampleIndex, uint32_t seed, int32_t cubeSize)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ctx ctx = ctx_make();
    ctx.rng = pcg4((uint32_t)i, sampleIndex, seed, cubeSize > 0 ? PT_TAG_CUBE : PT_TAG_RAY);
    v4 o, d;
    if (ro) { o = ld4(ro, i); d = ld4(rd, i); }
    else if (cubeSize > 0)
    {
        const int32_t flen = cubeSize * cubeSize;
        const int32_t face = i / flen, fi = i % flen;
        const v2 Xi = f2_tent(Sample2D(ctx));
        o = origin;
        d = Cubemap_CalcDir(cubeSize, face, fi % cubeSize, fi / cubeSize, Xi);
    }
    else
    {
        o = origin;
        d = SampleUnitSphere(Sample2D(ctx));
        directionsOut[i] = make_float4(d.x, d.y, d.z, d.w);
    }
    path_start(ps, extO, extD, queue, i, ctx, o, d);
}

/* Cubemap_Bake's blend, cubemap.c:158: color[face][fi] = f3_lerpvs(color, result.color, weight); faces are contiguous here */
__global__ __launch_bounds__(256) void k_cubemap_blend(const PimResult* __restrict__ results, float* __restrict__ faces, int32_t n, float weight)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* c = faces + (size_t)i * 3;
    const PimResult r = results[i];
    for (int k = 0; k < 3; ++k) c[k] = c[k] + (r.color[k] - c[k]) * weight;
}

/* AmbCube_Bake's fit loop, ambcube.c:22-30: a serial recurrence over the samples (order is part of the result),
 * so ONE lane walks them; its wave stages 64 samples at a time through registers + shuffles */
__global__ __launch_bounds__(64) void k_ambcube_fit(const PimResult* __restrict__ results, const float4* __restrict__ directions,
    int32_t samples, float w, float4* __restrict__ cube)
{
    AmbCube c;
    for (int k = 0; k < 6; ++k) { const float4 v = cube[k]; c.Values[k] = f4_v(v.x, v.y, v.z, v.w); }
    for (int32_t base = 0; base < samples; base += 64)
    {
        const int32_t mine = base + (int32_t)threadIdx.x;
        float4 d = make_float4(0.0f, 0.0f, 0.0f, 0.0f); float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
        if (mine < samples) { d = directions[mine]; const PimResult r = results[mine]; r0 = r.color[0]; r1 = r.color[1]; r2 = r.color[2]; }
        const int32_t cnt = min(64, samples - base);
        for (int32_t j = 0; j < cnt; ++j)
        {
            const v4 dir = f4_v(__shfl(d.x, j, 64), __shfl(d.y, j, 64), __shfl(d.z, j, 64), __shfl(d.w, j, 64));
            const v4 rad = f4_v(__shfl(r0, j, 64), __shfl(r1, j, 64), __shfl(r2, j, 64), 0.0f);     /* f3_f4(result.color, 0), pt.c:2610 */
            AmbCube_Fit(c, w, dir, rad);
        }
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k) cube[k] = make_float4(c.Values[k].x, c.Values[k].y, c.Values[k].z, c.Values[k].w);
}

/* ---------------------------------------------------------------------------------------------- */
/* shade_a: everything of a bounce between the extension hit and the direct-light ray              */

/* one path through the stage: its state is read from and written back to the path-state arrays (slot); the direct-light
 * ray comes back in the result (tFar < tNear: none).  `deferred` (FAST build only): nothing was written, the SLOW build
 * redoes the stage. */
struct StageAOut { float4 nO, nD; bool deferred; uint32_t shadow, other; bool overflow; int32_t target; DHit pre; };

template <bool SLOW, bool PRE>
__device__ __forceinline__ StageAOut stage_a(const DScene& s, const StageArgs& a, int32_t slot, const DHit extHit, int32_t bounce, const LdsStack& stack)
{
        StageAOut out; out.deferred = false; out.shadow = 0; out.other = 0; out.overflow = false;
        out.target = -1; out.pre.prim = -1; out.pre.t = 0.0f; out.pre.u = 0.0f; out.pre.v = 0.0f;
        Ctx ctx = ctx_make();
        if (SLOW) ctx.stack = stack;
        { const uint4 r = a.ps.rng[slot]; ctx.rng.x = r.x; ctx.rng.y = r.y; ctx.rng.z = r.z; ctx.rng.w = r.w; }
        const v4 ro = ld4(a.ps.ro, slot), rd = ld4(a.ps.rd, slot);
        v4 attenuation = ld4(a.ps.atten, slot);
        float resultWeight = attenuation.w;
        v4 luminance = ld4(a.ps.lum, slot);
        const uint32_t prevFlags = __float_as_uint(luminance.w);

        float4 nee0 = make_float4(__int_as_float(PT_NEE_DEAD), 0.0f, 0.0f, 0.0f);
        float4 nee1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float4 nO = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nD = make_float4(0.0f, 0.0f, 1.0f, -1.0f);   /* tFar < tNear: no ray */
        bool finished = false;

        const RayHit hit = make_rayhit(s, rd, extHit);
        if (hit.type == PIM_HIT_NOTHING)
        {
            luminance = f4_add(luminance, f4_mul(attenuation, GetSky(s, rd)));        /* pt.c:2334-2338 */
            finished = true;
        }
        else if ((hit.type == PIM_HIT_BACKFACE) && !(hit.flags & PIM_MATFLAG_REFRACTIVE))
        {
            finished = true;                                                          /* pt.c:2339-2343 */
        }
        else
        {
            const Scatter ms = ScatterRay<SLOW>(ctx, s, ro, rd, hit.wuvt.w);          /* pt.c:2346-2367 */
            if (!SLOW && ctx.defer)
            {
                /* the medium got in the way: nothing has been written yet, the SLOW build redoes this stage */
            }
            else if (ms.pdf > PT_EPS)
            {
                luminance = f4_add(luminance, f4_mul(attenuation, ms.luminance));
                attenuation = f4_mul(attenuation, f4_divs(ms.attenuation, ms.pdf));
                v4 albedoAcc = ld4(a.ps.albedo, slot), normalAcc = ld4(a.ps.normal, slot);
                {
                    const v4 al = f4_muls(attenuation, 1.0f / PT_TAU);
                    const float w = f1_sat(1.0f - f4_avglum(al));
                    resultWeight += w;
                    albedoAcc = f4_v(albedoAcc.x + al.x * w, albedoAcc.y + al.y * w, albedoAcc.z + al.z * w, 0.0f);
                    const v4 nr = f4_neg(rd);
                    normalAcc = f4_v(normalAcc.x + nr.x * w, normalAcc.y + nr.y * w, normalAcc.z + nr.z * w, 0.0f);
                }
                st4(a.ps.albedo, slot, albedoAcc); st4(a.ps.normal, slot, normalAcc);
                st4(a.ps.ro, slot, ms.pos);
                a.ps.rd[slot] = make_float4(ms.dir.x, ms.dir.y, ms.dir.z, 0.0f);
                luminance.w = __uint_as_float(0u);                                    /* prevFlags = 0 */
                nee0.x = __int_as_float(PT_NEE_MEDIA);
            }
            else
            {
                attenuation = f4_mul(attenuation, ms.attenuation);
                const SurfHit surf = GetSurface(s, rd, hit);                          /* pt.c:2369 */
                if (bounce > 0) LightOnHit(s, ro, surf.emission, hit.iVert);        /* pt.c:2370-2373 */
                if ((bounce == 0) || (prevFlags & PIM_MATFLAG_REFRACTIVE))
                {
                    const float keep = luminance.w;
                    luminance = f4_add(luminance, f4_mul(surf.emission, attenuation)); /* pt.c:2375-2378 */
                    luminance.w = keep;
                }
                if (hit.flags & PIM_MATFLAG_SKY)
                {
                    finished = true;                                                  /* pt.c:2380-2383 */
                }
                else
                {
                    /* EstimateDirect up to its ray, pt.c:1849-1870 / 1899-1906 */
                    nee0.x = __int_as_float(PT_NEE_NONE);
                    if (!(surf.flags & PIM_MATFLAG_REFRACTIVE))
                    {
                        const float pRough = f1_lerp(0.05f, 0.95f, surf.roughness);
                        const float pSmooth = 1.0f - pRough;
                        if (Sample1D(ctx) < pRough)
                        {
                            int32_t iVert; float selectPdf;
                            if (LightSelect(ctx, s, surf.P, &iVert, &selectPdf))
                            {
                                if (hit.iVert != iVert)
                                {
                                    const LightRay lr = SampleLight_ray(ctx, s, surf.P, iVert);
                                    nee0 = make_float4(__int_as_float(PT_NEE_LIGHT), __int_as_float(iVert), selectPdf, lr.distSq);
                                    nO = make_float4(surf.P.x, surf.P.y, surf.P.z, 0.0f);
                                    nD = make_float4(lr.rd.x, lr.rd.y, lr.rd.z, lr.distance + 0.01f * PT_MILLI);
                                    ++out.shadow;
                                    /* SampleLight asks one thing of this ray: is its closest hit the light's triangle (pt.c:1813-1815)?
                                     * The triangle itself is tested here, with the traversal's own test, so that the walk only has
                                     * to look for something in front of it and can stop at the first such thing (trace.hip, TGT) */
                                    if (PRE && a.neeTarget)
                                    {
                                        const RayPre rp = ray_pre(nO.x, nO.y, nO.z, nD.x, nD.y, nD.z);
                                        const DTriRec* lt = tri_of(s, iVert);
                                        const v4 A = lt->p0, B = lt->p1, C = lt->p2;
                                        float t, vn, wn, det;
                                        out.target = iVert / 3;
                                        if (tri_test_raw(rp, make_float4(A.x, A.y, A.z, 0.0f), make_float4(B.x, B.y, B.z, 0.0f), make_float4(C.x, C.y, C.z, 0.0f), 0.0f, nD.w, &t, &vn, &wn, &det))
                                        {
                                            out.pre.prim = out.target; out.pre.t = t; out.pre.u = vn / det; out.pre.v = wn / det;
                                            nD.w = t;                       /* nothing beyond the light matters */
                                        }
                                        else nD.w = -1.0f;                  /* the ray does not meet the light at all: no walk, a miss */
                                    }
                                }
                            }
                        }
                        else
                        {
                            const Scatter sample = Scatter_Principled<SLOW>(ctx, s, surf, rd);   /* never refractive here */
                            const float brdfPdf = sample.pdf * pSmooth;
                            if (brdfPdf > PT_EPS)
                            {
                                nee0.x = __int_as_float(PT_NEE_BRDF);
                                nee1 = make_float4(sample.attenuation.x, sample.attenuation.y, sample.attenuation.z, sample.pdf);
                                nO = make_float4(surf.P.x, surf.P.y, surf.P.z, 0.0f);
                                nD = make_float4(sample.dir.x, sample.dir.y, sample.dir.z, PT_RCP_EPS);
                                ++out.shadow;
                            }
                        }
                    }
                }
            }
        }
        out.deferred = !SLOW && ctx.defer;
        if (out.deferred)
        {
            /* park the lane's NEE slot (no ray) and leave the state as it was */
        }
        else
        {
        if (finished)
        {
            path_finish(a, slot, ctx, luminance, resultWeight, ld4(a.ps.albedo, slot), ld4(a.ps.normal, slot));
            nee0.x = __int_as_float(PT_NEE_DEAD);
        }
        else
        {
            a.ps.rng[slot] = make_uint4(ctx.rng.x, ctx.rng.y, ctx.rng.z, ctx.rng.w);
            a.ps.atten[slot] = make_float4(attenuation.x, attenuation.y, attenuation.z, resultWeight);
            a.ps.lum[slot] = make_float4(luminance.x, luminance.y, luminance.z, luminance.w);
        }
        a.ps.nee0[slot] = nee0; a.ps.nee1[slot] = nee1;
        out.other = ctx.otherRays; out.overflow = ctx.overflow;
        }
        out.nO = nO; out.nD = nD;
        return out;
}

/* SLOW builds: the place of a thread in its block's share of the short list.  Entry e of the list goes to block e % grid
 * and there to rank e / grid; ranks go to the block's WAVES first (rank r -> wave r % 8, lane r / 8), so that a block's few
 * heavy paths sit in eight waves side by side instead of sharing the first one */
__device__ __forceinline__ uint32_t slow_rank() { return (threadIdx.x >> 6) + (kShadeThreads / 64) * (threadIdx.x & 63); }

/* the traversal stack of a kernel's inline rays: lane-interleaved LDS words + the spill area behind them (traverse.h) */
__device__ __forceinline__ LdsStack kernel_stack(unsigned long long* shared, const StageArgs& a)
{
    LdsStack st;
    st.lds = (lds_u64*)(shared + threadIdx.x); st.ldsStride = (int)blockDim.x;
    st.spillStride = (int64_t)gridDim.x * blockDim.x;
    st.spill = (glb_u64*)(a.spill + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x));
    st.sp = 0; st.cap = a.stackCap;
    return st;
}

template <bool SLOW>
__global__ __launch_bounds__(kShadeThreads, SLOW ? 1 : PT_SHADE_WAVES) void k_shade_a(DScene s, StageArgs a)
{
    __shared__ unsigned long long sStack[SLOW ? PT_STACK_LDS * kShadeThreads : 1];
    const LdsStack stack = kernel_stack(sStack, a);
    const int32_t count = SLOW ? a.counters[2] : a.counters[a.inSel];
    if (!SLOW && blockIdx.x == 0 && threadIdx.x == 0)
    {
        a.counters[a.inSel ^ 1] = 0;                         /* output queue of shade_b starts empty */
        a.counters[3] = 0;                                   /* and so does shade_b's slow list */
        a.counters[PT_COUNTS_BASE + min(a.bounce, PT_COUNTS_MAX - 1)] = count;   /* history for the host's tail policy */
        atomicAdd(&a.stats[0], (unsigned long long)count);   /* extension rays of this bounce */
    }
    uint32_t shadow = 0, other = 0; bool overflow = false;
    /* SLOW: the short list is dealt round-robin over the blocks (entry e -> block e % grid), so that its few hundred
     * heavy paths run on every CU at once, a few lanes per wave, instead of filling the waves of two blocks */
    for (int32_t base = SLOW ? 0 : blockIdx.x * kShadeThreads; base < count; base += gridDim.x * kShadeThreads)
    {
        const int32_t idx = SLOW ? base + (int32_t)(slow_rank() * gridDim.x + blockIdx.x) : base + (int32_t)threadIdx.x;
        bool deferred = false;
        int32_t pos = 0;
        if (idx < count)
        {
            pos = SLOW ? a.slowA[idx] : idx;
            const StageAOut r = stage_a<SLOW, true>(s, a, a.queueIn[pos], a.extHit[pos], a.bounce, stack);
            a.neeO[pos] = r.nO; a.neeD[pos] = r.nD;
            if (a.neeTarget)
            {
                a.neeTarget[pos] = r.target;
                if (r.target >= 0) a.neeHitOut[pos] = r.pre;
            }
            deferred = r.deferred;
            shadow += r.shadow; other += r.other; overflow |= r.overflow;
        }
        if (!SLOW)
        {
            if (a.parkSlot)
            {
                /* out of the wavefront: no direct-light ray, nothing for shade_b to do, and the tail kernel starts the path again
                 * from this bounce's extension ray (the stage has written nothing) */
                const int32_t q = wave_append(&a.counters[PT_PARKED], deferred);
                if (deferred)
                {
                    const float4 eo = a.extO[pos];
                    a.parkSlot[q] = a.queueIn[pos];
                    a.parkO[q] = make_float4(eo.x, eo.y, eo.z, __int_as_float(a.bounce | PT_PARK_COUNTED));
                    a.parkD[q] = a.extD[pos];
                    a.ps.nee0[a.queueIn[pos]] = make_float4(__int_as_float(PT_NEE_DEAD), 0.0f, 0.0f, 0.0f);
                    a.neeD[pos] = make_float4(0.0f, 0.0f, 1.0f, -1.0f);
                    if (a.neeTarget) a.neeTarget[pos] = -1;
                }
            }
            else
            {
                const int32_t q = wave_append(&a.counters[2], deferred);
                if (deferred) a.slowA[q] = pos;
            }
        }
    }
    __shared__ int32_t sh[2 + kShadeThreads / 64];
    block_add(&a.stats[1], shadow, sh);
    block_add(&a.stats[2], other, sh);
    if (overflow) atomicOr((unsigned int*)&a.stats[4], 1u);
}

/* ---------------------------------------------------------------------------------------------- */
/* shade_b: from the direct-light hit to the next extension ray                                    */

/* one path through the stage; `alive`: the path goes on with the extension ray (ro, rd) */
struct StageBOut { bool alive, deferred; v4 ro, rd; uint32_t other; bool overflow; };

template <bool SLOW>
__device__ __forceinline__ StageBOut stage_b(const DScene& s, const StageArgs& a, int32_t slot, const DHit extHit, const float4 neeD,
    const DHit neeHit, int32_t bounce, const LdsStack& stack)
{
            StageBOut out; out.alive = false; out.deferred = false; out.ro = f4_s(0.0f); out.rd = f4_s(0.0f); out.other = 0; out.overflow = false;
            const float4 nee0 = a.ps.nee0[slot];
            const int32_t mode = __float_as_int(nee0.x);
            if (mode != PT_NEE_DEAD)
            {
                Ctx ctx = ctx_make();
                if (SLOW) ctx.stack = stack;
                { const uint4 r = a.ps.rng[slot]; ctx.rng.x = r.x; ctx.rng.y = r.y; ctx.rng.z = r.z; ctx.rng.w = r.w; }
                v4 ro = ld4(a.ps.ro, slot), rd = ld4(a.ps.rd, slot);
                v4 attenuation = ld4(a.ps.atten, slot);
                float resultWeight = attenuation.w;
                v4 luminance = ld4(a.ps.lum, slot);
                uint32_t prevFlags = __float_as_uint(luminance.w);
                v4 albedoAcc = ld4(a.ps.albedo, slot), normalAcc = ld4(a.ps.normal, slot);
                bool finished = false;
                if (mode != PT_NEE_MEDIA)
                {
                    const RayHit hit = make_rayhit(s, rd, extHit);
                    const SurfHit surf = GetSurface(s, rd, hit);
                    /* EstimateDirect after its ray, pt.c:1871-1897 / 1907-1918 */
                    v4 Li = f4_s(0.0f);
                    if (mode == PT_NEE_LIGHT || mode == PT_NEE_BRDF)
                    {
                        const float pRough = f1_lerp(0.05f, 0.95f, surf.roughness);
                        const float pSmooth = 1.0f - pRough;
                        const float4 nD = neeD;
                        const v4 nrd = f4_v(nD.x, nD.y, nD.z, 0.0f);
                        const RayHit nh = make_rayhit(s, nrd, neeHit);
                        if (mode == PT_NEE_LIGHT)
                        {
                            const int32_t iVert = __float_as_int(nee0.y);
                            const float selectPdf = nee0.z;
                            const LightSample sample = SampleLight_finish<SLOW>(ctx, s, surf.P, iVert, nrd, nee0.w, nh);
                            v4 L = sample.luminance;
                            const float lightPdf = sample.pdf * selectPdf * pRough;
                            if ((lightPdf > PT_EPS) && (f4_hmax3(L) > PT_EPS))
                            {
                                const v4 brdf = Eval_Principled(s, surf, rd, nrd);
                                L = f4_mul(L, brdf);
                                const float brdfPdf = brdf.w * pSmooth;
                                if (brdfPdf > PT_EPS)
                                {
                                    L = f4_muls(L, PowerHeuristic(lightPdf, brdfPdf) / lightPdf);
                                    Li = f4_add(Li, L);
                                }
                            }
                        }
                        else
                        {
                            const float4 nee1 = a.ps.nee1[slot];
                            const v4 sampleAtt = f4_v(nee1.x, nee1.y, nee1.z, nee1.w);
                            const float brdfPdf = nee1.w * pSmooth;
                            /* LightEvalPdf after its ray, pt.c:1838-1846 */
                            float lightPdf = 0.0f;
                            if (nh.type != PIM_HIT_NOTHING)
                            {
                                const float cosTheta = f1_abs(f4_dot3(nrd, nh.normal));
                                const float area = GetArea(s, nh.iVert);
                                const float distSq = f1_max(PT_EPS, nh.wuvt.w * nh.wuvt.w);
                                lightPdf = LightPdf(area, cosTheta, distSq);
                            }
                            lightPdf *= pRough;
                            if (lightPdf > PT_EPS)
                            {
                                lightPdf *= LightSelectPdf(s, nh.iVert, surf.P);
                                v4 L = f4_mul(GetEmission(s, nrd, nh), sampleAtt);
                                if (f4_hmax3(L) > PT_EPS)
                                {
                                    L = f4_muls(L, PowerHeuristic(brdfPdf, lightPdf) / brdfPdf);
                                    L = f4_mul(L, CalcTransmittance<SLOW>(ctx, s, surf.P, nrd, nh.wuvt.w));
                                    Li = f4_add(Li, L);
                                }
                            }
                        }
                    }
                    {
                        const float keep = luminance.w;
                        luminance = f4_add(luminance, f4_mul(Li, attenuation));      /* pt.c:2385-2388 */
                        luminance.w = keep;
                    }
                    const Scatter scatter = Scatter_Principled<SLOW>(ctx, s, surf, rd);  /* pt.c:2390 */
                    if (!(scatter.pdf > PT_EPS))
                    {
                        finished = true;                                              /* pt.c:2391-2394 */
                    }
                    else
                    {
                        ro = scatter.pos; rd = scatter.dir;
                        attenuation = f4_mul(attenuation, f4_divs(scatter.attenuation, scatter.pdf));
                        prevFlags = surf.flags;
                        const v4 al = f4_muls(attenuation, 1.0f / PT_PI);
                        const float w = f1_sat(1.0f - f4_avglum(al));
                        resultWeight += w;
                        albedoAcc = f4_v(albedoAcc.x + surf.albedo.x * w, albedoAcc.y + surf.albedo.y * w, albedoAcc.z + surf.albedo.z * w, 0.0f);
                        normalAcc = f4_v(normalAcc.x + surf.N.x * w, normalAcc.y + surf.N.y * w, normalAcc.z + surf.N.z * w, 0.0f);
                    }
                }
                /* top of bounce b+1, pt.c:2319-2331 */
                if (!finished)
                {
                    if (bounce + 1 >= a.bounceLimit) finished = true;
                    else if (!roulette(ctx, attenuation)) finished = true;
                }
                out.deferred = !SLOW && ctx.defer;
                if (out.deferred)
                {
                    /* refraction or a collision on the light ray: nothing written, the SLOW build redoes this stage */
                }
                else if (finished)
                {
                    path_finish(a, slot, ctx, luminance, resultWeight, albedoAcc, normalAcc);
                }
                else
                {
                    out.alive = true; out.ro = ro; out.rd = rd;
                    a.ps.rng[slot] = make_uint4(ctx.rng.x, ctx.rng.y, ctx.rng.z, ctx.rng.w);
                    a.ps.ro[slot] = make_float4(ro.x, ro.y, ro.z, 0.0f);
                    a.ps.rd[slot] = make_float4(rd.x, rd.y, rd.z, 0.0f);
                    a.ps.atten[slot] = make_float4(attenuation.x, attenuation.y, attenuation.z, resultWeight);
                    a.ps.lum[slot] = make_float4(luminance.x, luminance.y, luminance.z, __uint_as_float(prevFlags));
                    st4(a.ps.albedo, slot, albedoAcc); st4(a.ps.normal, slot, normalAcc);
                }
                if (!out.deferred) { out.other = ctx.otherRays; out.overflow = ctx.overflow; }
            }
            return out;
}

template <bool SLOW>
__global__ __launch_bounds__(kShadeThreads, SLOW ? 1 : PT_SHADE_WAVES) void k_shade_b(DScene s, StageArgs a)
{
    __shared__ unsigned long long sStack[SLOW ? PT_STACK_LDS * kShadeThreads : 1];
    const LdsStack stack = kernel_stack(sStack, a);
    const int32_t count = SLOW ? a.counters[3] : a.counters[a.inSel];
    int32_t* countOut = &a.counters[a.inSel ^ 1];
    if (!SLOW && blockIdx.x == 0 && threadIdx.x == 0) a.counters[2] = 0;     /* next bounce's shade_a slow list starts empty */
    __shared__ int32_t sh[2 + kShadeThreads / 64];
    uint32_t other = 0; bool overflow = false;
    for (int32_t base = SLOW ? 0 : blockIdx.x * kShadeThreads; base < count; base += gridDim.x * kShadeThreads)
    {
        const int32_t idx = SLOW ? base + (int32_t)(slow_rank() * gridDim.x + blockIdx.x) : base + (int32_t)threadIdx.x;   /* as in shade_a */
        bool alive = false, deferred = false;
        int32_t slot = 0, pos = 0;
        v4 ro = f4_s(0.0f), rd = f4_s(0.0f);
        if (idx < count)
        {
            pos = SLOW ? a.slowB[idx] : idx;
            slot = a.queueIn[pos];
            const StageBOut r = stage_b<SLOW>(s, a, slot, a.extHit[pos], a.neeD[pos], a.neeHit[pos], a.bounce, stack);
            alive = r.alive; deferred = r.deferred; ro = r.ro; rd = r.rd;
            other += r.other; overflow |= r.overflow;
        }
        /* russian-roulette compaction: survivors of this wave take consecutive queue positions */
        {
            const int32_t q = block_append(countOut, alive, sh);
            if (alive)
            {
                a.queueOut[q] = slot;
                a.extOOut[q] = make_float4(ro.x, ro.y, ro.z, 0.0f);
                a.extDOut[q] = make_float4(rd.x, rd.y, rd.z, PT_RCP_EPS);
            }
        }
        if (!SLOW)
        {
            const int32_t q = wave_append(&a.counters[3], deferred);
            if (deferred) a.slowB[q] = pos;
        }
    }
    block_add(&a.stats[2], other, sh);
    if (overflow) atomicOr((unsigned int*)&a.stats[4], 1u);
}

/* ---------------------------------------------------------------------------------------------- */
/* tail: every remaining bounce of a path in ONE launch                                            */

/*
 * Deep bounces hold few paths (russian roulette), and a wavefront bounce costs six dependent launches whatever its
 * size: two traversal launches and four stage launches at their latency floors (about 170 us per bounce on Cornell
 * 1080p for a queue that has shrunk to a few per cent of the image).  Once the queue is short the host hands it to this
 * kernel instead: one lane per path runs the rest of Pt_TraceRay's loop (pt.c:2319-2407) to the end -- the same stage
 * functions (SLOW builds: everything inline) and the inline closest-hit traversal.  Path state still goes through
 * the path-state arrays exactly as between the wavefront's launches, and every path owns its Prng stream, so where
 * the switch happens does not change a bit of the result (tested for every switch point).
 */
static constexpr int kTailThreads = 64;       /* one wave per block: the scheduler balances the uneven paths wave by wave */

/*
 * Paths die at different bounces, so a lane whose path has ended takes the next queued path at the top of the bounce
 * loop (one wave-aggregated atomic on counters[4], zeroed by the host with the other counters) instead of idling until
 * the longest path of its wave ends.  The host launches no more waves than the machine holds at this kernel's
 * occupancy; every lane leaves once the cursor has passed the queue's end, so the grid drains.
 */
template <int OCC>
__global__ __launch_bounds__(kTailThreads, OCC) void k_tail(DScene s, StageArgs a)
{
    const int32_t countPark = a.parkSlot ? a.counters[PT_PARKED] : 0;
    const int32_t count = countPark + (a.tailMainEmpty ? 0 : a.counters[a.inSel]);       /* the parked paths first (they have the most bounces left), then the queue */
    uint32_t ext = 0, shadow = 0, other = 0; bool overflow = false;
    bool have = false, counted = false;
    int32_t slot = 0, b = 0;
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d = o;
    /* the traversal stack of the two rays of a bounce: LDS (lane-interleaved) + spill area, as in k_trace -- a pop from
     * a private-memory stack is a memory round trip on the path's dependent chain */
    __shared__ unsigned long long sStack[PT_STACK_LDS * kTailThreads];
    LdsStack st;
    st.lds = (lds_u64*)(sStack + threadIdx.x); st.ldsStride = kTailThreads;
    st.spillStride = (int64_t)gridDim.x * kTailThreads;
    st.spill = (glb_u64*)(a.spill + ((int64_t)blockIdx.x * kTailThreads + threadIdx.x));
    st.cap = a.stackCap;
    /* the two rays of a bounce are walked by the whole wave (traverse_coop): lanes without a path stay in the loop as helpers */
    __shared__ unsigned long long sCoop[2 * kTailThreads];
    __shared__ int sList[kTailThreads];
    CoopArea ca;
    ca.best = (lds_u64*)sCoop; ca.uv = (lds_u64*)(sCoop + kTailThreads); ca.list = (lds_i32*)sList;
    bool drained = (int32_t)(threadIdx.x & 63) >= a.tailLanes;
    for (;;)
    {
        const bool want = !have && !drained;
        const int32_t pos = wave_append(&a.counters[4], want);
        if (want)
        {
            if (pos >= count) drained = true;
            else if (pos >= countPark) { const int32_t k = pos - countPark; slot = a.queueIn[k]; o = a.extO[k]; d = a.extD[k]; b = a.bounce; have = true; }
            else
            {
                slot = a.parkSlot[pos]; o = a.parkO[pos]; d = a.parkD[pos];
                b = __float_as_int(o.w) & ~PT_PARK_COUNTED; counted = true; o.w = 0.0f; have = true;
            }
        }
        if (__ballot(have) == 0ull) break;
#ifdef PT_TAIL_PROF
        TraverseCounts tc0 = { 0u, 0u }, tc1 = { 0u, 0u };
        const long long c0 = wall_clock64();
        const DHit eh = traverse_coop<true>(s.bvh, have, o.x, o.y, o.z, d.x, d.y, d.z, o.w, d.w, st, overflow, ca, &tc0);
        const long long c1 = wall_clock64();
#else
        const DHit eh = traverse_coop(s.bvh, have, o.x, o.y, o.z, d.x, d.y, d.z, o.w, d.w, st, overflow, ca);
#endif
        StageAOut ra; ra.nO = make_float4(0.0f, 0.0f, 0.0f, 0.0f); ra.nD = make_float4(0.0f, 0.0f, 1.0f, -1.0f);
        if (have)
        {
            if (!counted) ++ext;                /* a parked path's ray was counted by the shade_a launch that parked it */
            counted = false;
            /* the FAST build first, the SLOW one for the lanes it deferred (a FAST stage that defers has written nothing): the
             * common path then runs the code that was compiled without the rare heavy cases (Cornell 1080p 1.72 -> 1.68 ms) */
            ra = stage_a<false, false>(s, a, slot, eh, b, st);
            shadow += ra.shadow; other += ra.other; overflow |= ra.overflow;
            if (ra.deferred)
            {
                ra = stage_a<true, false>(s, a, slot, eh, b, st);
                shadow += ra.shadow; other += ra.other; overflow |= ra.overflow;
            }
        }
#ifdef PT_TAIL_PROF
        const long long c2 = wall_clock64();
        DHit nh = traverse_coop<true>(s.bvh, have, ra.nO.x, ra.nO.y, ra.nO.z, ra.nD.x, ra.nD.y, ra.nD.z, ra.nO.w, ra.nD.w, st, overflow, ca, &tc1);
        const long long c3 = wall_clock64();
#else
        DHit nh = traverse_coop(s.bvh, have, ra.nO.x, ra.nO.y, ra.nO.z, ra.nD.x, ra.nD.y, ra.nD.z, ra.nO.w, ra.nD.w, st, overflow, ca);
#endif
        if (have)
        {
            StageBOut rb = stage_b<false>(s, a, slot, eh, ra.nD, nh, b, st);
            other += rb.other; overflow |= rb.overflow;
            if (rb.deferred)
            {
                rb = stage_b<true>(s, a, slot, eh, ra.nD, nh, b, st);
                other += rb.other; overflow |= rb.overflow;
            }
            have = rb.alive;
            o = make_float4(rb.ro.x, rb.ro.y, rb.ro.z, 0.0f);
            d = make_float4(rb.rd.x, rb.rd.y, rb.rd.z, PT_RCP_EPS);
            ++b;
        }
#ifdef PT_TAIL_PROF
        {
            /* per wave-bounce: time of the four phases; trips of the wave's walks (lane 0), triangles tested */
            const long long c4 = wall_clock64();
            const uint64_t act = __ballot(true);
            atomicAdd(&a.stats[16], (unsigned long long)tc0.tris); atomicAdd(&a.stats[17], (unsigned long long)tc1.tris);
            if ((threadIdx.x & 63) == 0)
            {
                atomicAdd(&a.stats[8], (unsigned long long)(c1 - c0)); atomicAdd(&a.stats[9], (unsigned long long)(c2 - c1));
                atomicAdd(&a.stats[10], (unsigned long long)(c3 - c2)); atomicAdd(&a.stats[11], (unsigned long long)(c4 - c3));
                atomicAdd(&a.stats[12], 1ull); atomicAdd(&a.stats[13], (unsigned long long)__popcll(act));
                atomicAdd(&a.stats[14], (unsigned long long)tc0.nodes); atomicAdd(&a.stats[15], (unsigned long long)tc1.nodes);
            }
        }
#endif
    }
    __shared__ int32_t sh[2 + kTailThreads / 64];
    block_add(&a.stats[0], ext, sh);
    block_add(&a.stats[1], shadow, sh);
    block_add(&a.stats[2], other, sh);
    block_add(&a.stats[5], ext + shadow, sh);        /* rays traced here, not by the batched traversal launches */
    block_add(&a.stats[6], ext, sh);                 /* path-bounces run here */
    if (overflow) atomicOr((unsigned int*)&a.stats[4], 1u);
}

/* DofUpdate, pt.c:2485-2503: the autofocus ray and the smoothing of the focal length in one launch of one wave.  The ray is
 * walked by all 64 lanes (traverse_coop: ~12 trips instead of the ~50 dependent steps of a lone lane -- 67 -> 25 us on the
 * 1 M-triangle scene); the frame's focal length stays on the device for k_init_image and comes home with the statistics. */
__global__ __launch_bounds__(64) void k_focus(DBvh bvh, float4 ro, float4 rd, float focalLength, float tt, float* __restrict__ focalDev,
    unsigned long long* __restrict__ spill, uint32_t* __restrict__ status, int32_t stackCap)
{
    __shared__ unsigned long long sStack[PT_STACK_LDS * 64];
    __shared__ unsigned long long sCoop[2 * 64];
    __shared__ int sList[64];
    LdsStack st;
    st.lds = (lds_u64*)(sStack + threadIdx.x); st.ldsStride = 64;
    st.spillStride = 64; st.spill = (glb_u64*)(spill + threadIdx.x);
    st.sp = 0; st.cap = stackCap;
    CoopArea ca;
    ca.best = (lds_u64*)sCoop; ca.uv = (lds_u64*)(sCoop + 64); ca.list = (lds_i32*)sList;
    bool overflow = false;
    const DHit h = traverse_coop(bvh, threadIdx.x == 0, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, ro.w, rd.w, st, overflow, ca);
    if (overflow) atomicOr(status, 1u);
    if (threadIdx.x == 0)
    {
        if (h.prim >= 0 && h.t > 0.0f) focalLength = f1_lerp(focalLength, h.t, tt);
        *focalDev = focalLength;
    }
}

/* paths still queued when the host stops the loop (bounce limit reached inside shade_b never leaves any;
 * this covers maxBounces == 0 style early stops): not needed -- shade_b finishes every path at the limit. */

/* TraceFn's accumulate, pt.c:2547-2552 */
__global__ __launch_bounds__(256) void k_resolve_image(const PimResult* __restrict__ results, int32_t pixelBegin, int32_t pixelCount, float sampleWeight,
    float* __restrict__ color, float* __restrict__ albedo, float* __restrict__ normal)
{
    const int32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= pixelCount) return;
    const PimResult r = results[slot];
    const int64_t i = (int64_t)pixelBegin + slot;
    for (int k = 0; k < 3; ++k)
    {
        color[i * 3 + k] = f1_lerp(color[i * 3 + k], r.color[k], sampleWeight);
        albedo[i * 3 + k] = f1_lerp(albedo[i * 3 + k], r.albedo[k], sampleWeight);
        normal[i * 3 + k] = f1_lerp(normal[i * 3 + k], r.normal[k], sampleWeight);
    }
}

/* CalcStdDev, render_system.c:1374-1394.  The reference accumulates mean and variance serially in fp32, in
 * pixel order, and that order is part of the result.  Per-pixel terms are computed in parallel (k_stddev_terms);
 * the two running sums are then formed in pixel order by one wave (k_stddev_sum). */
struct FsState;
__device__ void fs_flag_term(FsState* st, float term);

__global__ __launch_bounds__(256) void k_stddev_terms(const float* __restrict__ color, int32_t len, const float* __restrict__ meanIn,
    float* __restrict__ terms, FsState* __restrict__ fast = nullptr)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const float lum = (color[i * 3] + color[i * 3 + 1] + color[i * 3 + 2]) * (1.0f / 3.0f);
    float term;
    if (!meanIn) term = lum * (1.0f / len);                           /* mean += lum * meanWeight */
    else
    {
        const float err = lum - meanIn[0];
        term = (1.0f / (len - 1)) * (err * err);                      /* variance += varianceWeight * (err * err) */
    }
    terms[i] = term;
    if (fast) fs_flag_term(fast, term);
}

/*
 * CalcStdDev's serial sums: sum += term[k] in pixel order, one dependent v_add_f32 per term -- the order is the result, so the
 * chain cannot be split.  EVERY lane of the one wave runs the whole chain: the terms arrive by vector loads whose address is
 * the same in all lanes (one 16-byte broadcast load brings four terms), kSumRing of them in flight -- a load is ~600 cycles
 * away, the ring holds 768 cycles of additions -- so the adder does not wait for memory: 11.5 ms per sum at 1920 x 1080
 * (5.5 ns per term).  Before: the running sum travelling across the lanes by one DPP add per term, 18.5 ms (the DPP wait
 * states sit on the chain); before that one lane fed through LDS, 22 ms.  Same order of additions, same bits in all three.
 * (Launched with 64 threads; the bound of 128 only keeps the compiler from proving the address uniform and moving the loads
 * to the scalar unit, whose loads cannot be waited for one by one.)
 */
static constexpr int kSumRing = 48;         /* float4 loads in flight */
static constexpr int kSumBody = 4 * kSumRing;   /* float4 per pass of the loop: straight-line code, nothing carried around the back edge */
struct FsState { float acc; int32_t pos, done, bad, mapsFrom, mapsE, mapsHaveAcc, rounds; };   /* k_fs_* below */

__global__ __launch_bounds__(128) void k_stddev_sum(const float* __restrict__ terms, int32_t len, float* __restrict__ out, int32_t sqrtOut,
    const FsState* __restrict__ fast = nullptr)
{
    const float4* __restrict__ t4 = (const float4*)terms + (threadIdx.x >> 6);       /* == terms: the launch has one wave */
    const int32_t n4 = len >> 2;
    float acc = 0.0f;
    int32_t i = 0;
    if (fast)
    {
        /* behind the parallel sum (k_fs_*): nothing to do when it finished; from the start when it saw a term it does not take;
         * else from where it stopped */
        if (fast->done) return;
        if (!fast->bad)
        {
            acc = fast->acc;
            int32_t j = fast->pos;
            for (; j < len && (j & 3) != 0; ++j) acc = acc + terms[j + (threadIdx.x >> 6)];
            if (j >= len) { if (threadIdx.x == 0) out[0] = sqrtOut ? pt_sqrt(acc) : acc; return; }
            i = j >> 2;
        }
    }
    /* One pass: kSumRing loads up front (their latency is paid once per pass, ~600 of ~3700 cycles), then every group of four
     * additions is followed by the reload of its registers for 48 groups later.  The ring is NOT carried around the loop: carried,
     * the register allocator rotated it with one v_mov per v_add on the chain's own issue slots (5.5 ns per term). */
    for (; i + kSumBody <= n4; i += kSumBody)
    {
        float4 ring[kSumRing];
#pragma unroll
        for (int k = 0; k < kSumRing; ++k) ring[k] = t4[i + k];
#pragma unroll
        for (int k = 0; k < kSumBody; ++k)
        {
            const float4 v = ring[k % kSumRing];
            acc = acc + v.x; acc = acc + v.y; acc = acc + v.z; acc = acc + v.w;
            if (k + kSumRing < kSumBody) ring[k % kSumRing] = t4[i + k + kSumRing];
        }
    }
    for (int32_t j = i * 4; j < len; ++j) acc = acc + terms[j + (threadIdx.x >> 6)];
    if (threadIdx.x == 0) out[0] = sqrtOut ? pt_sqrt(acc) : acc;
}

/*
 * The same serial sum, formed in parallel.  "The order is the result" holds for fp32 additions in general, but these sums add
 * terms >= 0 to a running sum that only grows, and while the sum stays inside one binade [2^e, 2^(e+1)) an fp32 addition is an
 * INTEGER addition: with q = 2^(e-23) the sum is A * q for an integer A in [2^23, 2^24), a term is tau * q for a real tau, and
 *     fl(A q + tau q) = (A + floor(tau) + [frac(tau) > 1/2] + [frac(tau) == 1/2 and A + floor(tau) odd]) * q
 * as long as the result stays below 2^24 q.  So a run of terms acts on A as "add an integer that depends on nothing but the
 * parity of A on entry" -- a pair of integers (increment for even A, increment for odd A) -- and such pairs compose
 * associatively: runs can be summarised in parallel and combined in order.  Only the additions that carry the sum into the
 * next binade (at most 254 of them in fp32, about 23 for a 4K frame) are done as real fp32 additions, one by one.
 *
 *   k_fs_maps   every 4096-term chunk not yet begun -> its pair for each of kFsCands binades starting at the current one
 *   k_fs_walk   one workgroup: combines the chunks' pairs in order (a scan), finds the first chunk in which the sum would leave
 *               its binade, goes through that chunk term by term (scan over its 4096 terms), does the crossing addition in
 *               fp32, goes on in the new binade; stops when the binade has left the range the pairs were made for
 *   ... launched kFsRounds times back to back without the host looking (a finished sum makes the rest return at once), and
 *   k_stddev_sum behind them takes over from wherever they stopped: if the rounds did not suffice, or if a term is negative,
 *   NaN, infinite or denormal (k_stddev_terms flags those; then the chain runs from the start as before).
 * Same additions in the same order with the same roundings: the bits are those of the chain (tests/test_gpu_stddev_sum.py
 * compares the two on images made of ties, zeros, single bright pixels and wide ranges), a 4K pt_stddev (two sums) 71.7 -> 0.76 ms.
 */
static constexpr int kFsChunk = 4096, kFsCands = 8, kFsRounds = 5, kFsWalkThreads = 1024, kFsDetailBudget = 8;
static constexpr int32_t kFsHuge = 1 << 25;                     /* "leaves the binade for certain" */
static constexpr long long kFsTop = 1ll << 24;
struct FsMap { long long a0, a1; };                             /* increment of A for even / odd A on entry */

__device__ void fs_flag_term(FsState* st, float term)
{
    const uint32_t u = __float_as_uint(term), ex = (u >> 23) & 0xFFu, mant = u & 0x7FFFFFu;
    const bool bad = ex == 0xFFu || (ex == 0u && mant != 0u) || ((u >> 31) != 0u && (u << 1) != 0u);
    if (bad) atomicOr(&st->bad, 1);
}

__device__ __forceinline__ FsMap fs_then(FsMap g, FsMap f)      /* g first, then f */
{
    FsMap h;
    h.a0 = g.a0 + ((g.a0 & 1) ? f.a1 : f.a0);
    h.a1 = g.a1 + (((1 + g.a1) & 1) ? f.a1 : f.a0);
    return h;
}

/* what adding the term with bits u does to A while the sum is in binade e: A += k + (tie & (A + k) & 1) */
__device__ __forceinline__ void fs_classify(uint32_t u, int e, bool haveAcc, int32_t& k, int32_t& tie)
{
    k = 0; tie = 0;
    const uint32_t ex = (u >> 23) & 0xFFu;
    if (ex == 0u) return;                                       /* zero */
    if (!haveAcc) { k = kFsHuge; return; }                      /* the sum is still 0: the first term that is not starts it */
    const uint32_t m = (u & 0x7FFFFFu) | 0x800000u;
    const int s = (int)ex - 127 - e;                            /* tau = m * 2^(s - 23) */
    if (s >= 0) { k = kFsHuge; return; }                        /* tau >= 2^23 and A >= 2^23: leaves the binade */
    const int r = -s;
    if (r > 24) return;                                         /* tau < 1/2 */
    const uint32_t rem = m & ((1u << r) - 1u), half = 1u << (r - 1);
    k = (int32_t)(m >> r) + (rem > half ? 1 : 0);
    tie = rem == half ? 1 : 0;
}

__device__ __forceinline__ void fs_apply(long long& A, int32_t k, int32_t tie) { A += k + (tie & (int32_t)((A + k) & 1)); }

__global__ __launch_bounds__(256) void k_fs_maps(const float* __restrict__ terms, int32_t len, FsState* __restrict__ st, FsMap* __restrict__ maps)
{
    if (st->done | st->bad) return;
    const int32_t pos = st->pos;
    const int32_t first = (pos + kFsChunk - 1) / kFsChunk;      /* the chunk pos stands in (if inside one) is the walk's own */
    const uint32_t accBits = __float_as_uint(st->acc);
    const bool haveAcc = accBits != 0u;
    const int eBase = (int)((accBits >> 23) & 0xFFu) - 127;
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->mapsFrom = first; st->mapsE = eBase; st->mapsHaveAcc = haveAcc ? 1 : 0; }
    const int32_t c = (int32_t)blockIdx.x;
    if (c < first || (long long)c * kFsChunk >= len) return;
    /* a thread's 16 terms in a row, and their pair for every candidate binade */
    constexpr int kPer = kFsChunk / 256;
    const int32_t base = c * kFsChunk + (int32_t)threadIdx.x * kPer;
    uint32_t u[kPer];
#pragma unroll
    for (int q = 0; q < kPer / 4; ++q)
    {
        const int32_t j = base + q * 4;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (j + 3 < len) v = *(const uint4*)(terms + j);
        else { if (j < len) v.x = __float_as_uint(terms[j]); if (j + 1 < len) v.y = __float_as_uint(terms[j + 1]); if (j + 2 < len) v.z = __float_as_uint(terms[j + 2]); }
        u[q * 4 + 0] = v.x; u[q * 4 + 1] = v.y; u[q * 4 + 2] = v.z; u[q * 4 + 3] = v.w;
    }
    __shared__ FsMap sMap[256][kFsCands];
#pragma unroll
    for (int cand = 0; cand < kFsCands; ++cand)
    {
        long long a0 = 0, a1 = 1;                               /* A itself (even / odd start), less the start at the end */
#pragma unroll
        for (int q = 0; q < kPer; ++q)
        {
            int32_t k, tie; fs_classify(u[q], eBase + cand, haveAcc, k, tie);
            fs_apply(a0, k, tie); fs_apply(a1, k, tie);
        }
        sMap[threadIdx.x][cand].a0 = a0; sMap[threadIdx.x][cand].a1 = a1 - 1;
    }
    for (int half = 1; half < 256; half <<= 1)
    {
        __syncthreads();
        const int items = (128 / half) * kFsCands;
        for (int item = threadIdx.x; item < items; item += 256)
        {
            const int cand = item % kFsCands, i = (item / kFsCands) * 2 * half;
            sMap[i][cand] = fs_then(sMap[i][cand], sMap[i + half][cand]);
        }
    }
    __syncthreads();
    if (threadIdx.x < kFsCands) maps[(size_t)c * kFsCands + threadIdx.x] = sMap[0][threadIdx.x];
}

/* exclusive scan of the workgroup's pairs in thread order (kFsWalkThreads threads); *total = all of them */
__device__ FsMap fs_scan(FsMap own, FsMap* sWave, FsMap* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    FsMap incl = own;
    for (int o = 1; o < 64; o <<= 1)
    {
        FsMap prev; prev.a0 = __shfl_up(incl.a0, o, 64); prev.a1 = __shfl_up(incl.a1, o, 64);
        if (lane >= o) incl = fs_then(prev, incl);
    }
    FsMap excl; excl.a0 = __shfl_up(incl.a0, 1, 64); excl.a1 = __shfl_up(incl.a1, 1, 64);
    if (lane == 0) { excl.a0 = 0; excl.a1 = 0; }
    __syncthreads();                                            /* the scan before this one has been read */
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    FsMap before; before.a0 = 0; before.a1 = 0;
    FsMap all; all.a0 = 0; all.a1 = 0;
    for (int w = 0; w < kFsWalkThreads / 64; ++w) { if (w == wave) before = all; all = fs_then(all, sWave[w]); }
    *total = all;
    return fs_then(before, excl);
}

/* smallest value of the workgroup (INT_MAX: none) */
__device__ int32_t fs_min(int32_t v, int32_t* sMin)
{
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sMin[threadIdx.x >> 6] = v;
    __syncthreads();
    int32_t m = 0x7FFFFFFF;
    for (int w = 0; w < kFsWalkThreads / 64; ++w) m = min(m, sMin[w]);
    return m;
}

__global__ __launch_bounds__(kFsWalkThreads) void k_fs_walk(const float* __restrict__ terms, int32_t len, FsState* __restrict__ st,
    const FsMap* __restrict__ maps, float* __restrict__ out, int32_t sqrtOut)
{
    if (st->done | st->bad) return;
    __shared__ FsMap sWave[kFsWalkThreads / 64];
    __shared__ int32_t sMin[kFsWalkThreads / 64];
    __shared__ long long sBefore;
    __shared__ uint32_t sTerm;
    /* the state, the same in every thread: the sum is A * 2^(e - 23), or 0 (haveAcc false), or has overflowed (isInf) */
    uint32_t accBits = __float_as_uint(st->acc);
    int32_t pos = st->pos;
    const int32_t mapsFrom = st->mapsFrom, mapsE = st->mapsE, mapsHaveAcc = st->mapsHaveAcc;
    bool haveAcc = accBits != 0u, isInf = false;
    int e = (int)((accBits >> 23) & 0xFFu) - 127;
    long long A = haveAcc ? (long long)((accBits & 0x7FFFFFu) | 0x800000u) : 0;
    const int32_t chunks = (len + kFsChunk - 1) / kFsChunk;
    int budget = kFsDetailBudget;
    for (;;)
    {
        if (pos >= len || isInf) break;
        int32_t c = pos / kFsChunk;
        const bool aligned = pos == c * kFsChunk;
        const int cand = e - mapsE;
        const bool valid = aligned && c >= mapsFrom && (mapsHaveAcc != 0) == haveAcc && (!haveAcc || (cand >= 0 && cand < kFsCands));
        if (valid)
        {
            /* the chunks' pairs in order: where does the sum leave this binade? */
            const int32_t todo = chunks - c, per = (todo + kFsWalkThreads - 1) / kFsWalkThreads;
            const int32_t c0 = min(c + (int32_t)threadIdx.x * per, chunks), c1 = min(c0 + per, chunks);
            const int useCand = haveAcc ? cand : 0;
            FsMap own; own.a0 = 0; own.a1 = 0;
            for (int32_t x = c0; x < c1; ++x) own = fs_then(own, maps[(size_t)x * kFsCands + useCand]);
            FsMap total;
            const FsMap excl = fs_scan(own, sWave, &total);
            long long a = A + ((A & 1) ? excl.a1 : excl.a0);
            int32_t found = 0x7FFFFFFF; long long before = 0;
            for (int32_t x = c0; x < c1; ++x)
            {
                const FsMap m = maps[(size_t)x * kFsCands + useCand];
                const long long next = a + ((a & 1) ? m.a1 : m.a0);
                if (next >= kFsTop) { found = x; before = a; break; }
                a = next;
            }
            const int32_t hit = fs_min(found, sMin);
            if (hit == 0x7FFFFFFF) { A += (A & 1) ? total.a1 : total.a0; pos = len; break; }
            if (found == hit) sBefore = before;
            __syncthreads();
            A = sBefore; c = hit; pos = hit * kFsChunk;
        }
        else if (aligned)
        {
            if (budget == 0) break;                             /* the next k_fs_maps makes pairs for this binade */
            --budget;
        }
        /* chunk c term by term, from pos to its end */
        const int32_t chunkEnd = min((c + 1) * kFsChunk, len);
        while (pos < chunkEnd && !isInf)
        {
            const int32_t j0 = c * kFsChunk + (int32_t)threadIdx.x * 4;
            uint32_t u[4];
            for (int q = 0; q < 4; ++q) { const int32_t j = j0 + q; u[q] = (j >= pos && j < len) ? __float_as_uint(terms[j]) : 0u; }
            long long a0 = 0, a1 = 1;
            for (int q = 0; q < 4; ++q) { int32_t k, tie; fs_classify(u[q], e, haveAcc, k, tie); fs_apply(a0, k, tie); fs_apply(a1, k, tie); }
            FsMap own; own.a0 = a0; own.a1 = a1 - 1;
            FsMap total;
            const FsMap excl = fs_scan(own, sWave, &total);
            long long a = A + ((A & 1) ? excl.a1 : excl.a0);
            int32_t found = 0x7FFFFFFF; long long before = 0; uint32_t term = 0u;
            for (int q = 0; q < 4; ++q)
            {
                int32_t k, tie; fs_classify(u[q], e, haveAcc, k, tie);
                long long next = a; fs_apply(next, k, tie);
                if (next >= kFsTop) { found = j0 + q; before = a; term = u[q]; break; }
                a = next;
            }
            const int32_t hit = fs_min(found, sMin);
            if (hit == 0x7FFFFFFF) { A += (A & 1) ? total.a1 : total.a0; pos = chunkEnd; break; }
            if (found == hit) { sBefore = before; sTerm = term; }
            __syncthreads();
            /* the addition that leaves the binade: in fp32, as the chain does it */
            const long long ab = sBefore;
            const float accBefore = haveAcc ? __uint_as_float(((uint32_t)(e + 127) << 23) | ((uint32_t)ab & 0x7FFFFFu)) : 0.0f;
            const float accNew = accBefore + __uint_as_float(sTerm);
            accBits = __float_as_uint(accNew);
            if (((accBits >> 23) & 0xFFu) == 0xFFu) { isInf = true; break; }
            haveAcc = true;
            e = (int)((accBits >> 23) & 0xFFu) - 127;
            A = (long long)((accBits & 0x7FFFFFu) | 0x800000u);
            pos = hit + 1;
        }
    }
    const float acc = isInf ? __builtin_inff() : (haveAcc ? __uint_as_float(((uint32_t)(e + 127) << 23) | ((uint32_t)A & 0x7FFFFFu)) : 0.0f);
    if (threadIdx.x == 0)
    {
        const bool done = pos >= len || isInf;
        st->acc = acc; st->pos = pos; st->done = done ? 1 : 0; st->rounds += 1;
        if (done) out[0] = sqrtOut ? pt_sqrt(acc) : acc;
    }
}

/* the launches of one sum: parallel rounds, then the chain for whatever they left (usually nothing).  head: FsState + the chunks' pairs */
static constexpr int32_t kFsStateFloats = 16;
static int64_t fs_head_floats(int64_t len) { return kFsStateFloats + ((len + kFsChunk - 1) / kFsChunk + 1) * kFsCands * (int64_t)(sizeof(FsMap) / sizeof(float)); }
static bool fs_serial_only() { const char* v = getenv("PIMGPU_STDDEV_SERIAL"); return v && v[0] == '1'; }     /* read per call: the tests switch it */

static void stddev_sum_launch(const float* color, int32_t len, const float* meanIn, float* terms, float* head, float* out, int32_t sqrtOut, hipStream_t stream)
{
    if (fs_serial_only())
    {
        k_stddev_terms<<<(len + 255) / 256, 256, 0, stream>>>(color, len, meanIn, terms);
        k_stddev_sum<<<1, 64, 0, stream>>>(terms, len, out, sqrtOut);
        return;
    }
    FsState* st = (FsState*)head;
    FsMap* maps = (FsMap*)(head + kFsStateFloats);
    hipMemsetAsync(st, 0, sizeof(FsState), stream);
    k_stddev_terms<<<(len + 255) / 256, 256, 0, stream>>>(color, len, meanIn, terms, st);
    const int32_t chunks = (len + kFsChunk - 1) / kFsChunk;
    for (int r = 0; r < kFsRounds; ++r)
    {
        k_fs_maps<<<chunks, 256, 0, stream>>>(terms, len, st, maps);
        k_fs_walk<<<1, kFsWalkThreads, 0, stream>>>(terms, len, st, maps, out, sqrtOut);
    }
    k_stddev_sum<<<1, 64, 0, stream>>>(terms, len, out, sqrtOut, st);
}

/* ---------------------------------------------------------------------------------------------- */
/* scratch of pimgpu_trace_stddev_noalloc: the per-pixel terms CalcStdDev's two serial sums walk over, and the two results.
 * A file-scope device array instead of the grown-on-demand work buffer of pimgpu_trace_stddev, for callers that cannot
 * call hipMalloc.  Every GPU holds its own copy (kStddevScratchFloats * 4 bytes = 31.9 MiB) until the process ends.  The
 * array's NAME is a host-side stand-in, not a device address: host code only ever uses the per-device address that
 * hipGetSymbolAddress returned at pimgpu_init (stddev_scratch_setup below). */
static constexpr int32_t kStddevNoAllocMaxPixels = 3840 * 2160;
/* floats in front of the terms: [0] mean, [1] stddev, from [64] the parallel sum's state and per-chunk pairs (fs_head_floats) */
static constexpr int32_t kStddevScratchHead = 64 + ((kFsStateFloats + (kStddevNoAllocMaxPixels / kFsChunk + 2) * kFsCands * (int32_t)(sizeof(FsMap) / sizeof(float)) + 63) / 64) * 64;
static constexpr size_t kStddevScratchFloats = (size_t)kStddevNoAllocMaxPixels + kStddevScratchHead;
static __device__ __align__(256) float scratch[kStddevScratchFloats];
static constexpr int kScratchMaxDevices = 64;
static float* g_scratchPtr[kScratchMaxDevices] = {};              /* per device: where that GPU's copy of `scratch` lives */

/* once per device, with that device current (pimgpu_init) */
int stddev_scratch_setup(int device)
{
    if (device < 0 || device >= kScratchMaxDevices) { set_error("device %d beyond the %d the library keeps scratch addresses for", device, kScratchMaxDevices); return PIMGPU_E_LIMIT; }
    if (g_scratchPtr[device]) return PIMGPU_OK;
    float* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) { set_error("hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e)); return PIMGPU_E_DEVICE; }
    g_scratchPtr[device] = ptr;
    return PIMGPU_OK;
}

/* ---------------------------------------------------------------------------------------------- */
/* host: the bounce loop                                                                           */

/*
 * Where the wavefront hands over to the tail kernel.  A batch that is small enough goes to the tail kernel whole (bounce
 * 0: the integrator as ONE launch); a larger one runs wavefront bounces until its queue is that short.  The queue lengths
 * are device-side; the host uses those of the PREVIOUS call (progressive rendering repeats the same camera and scene),
 * read back with the statistics.  The choice only moves time around (k_tail's header), so a stale history is harmless;
 * PIMGPU_TAIL_BOUNCE forces it (tests sweep it), negative = never.
 */
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
static const int kTailWavesPerCu = env_int("PIMGPU_TAIL_WAVES", 0);           /* 0: 12 or 16 with the build below */
/* Two builds of k_tail: 168 VGPRs (3 waves per SIMD, 76 B of spills) and <= 128 VGPRs with more spills (4 per SIMD).  A
 * tree that lives in L2 leaves little latency to hide and the spills cost more than the extra waves give; a tree served
 * from the Infinity Cache is the other way round (Cornell 1080p: tail 0.93 ms at 3 waves, 1.01 at 2 or 4, 1.28 at 5;
 * 1 M triangles: 3.42 / 3.10 / 3.13 ms at 3 / 4 / 5).  The host picks by the size of the tree. */
static constexpr int64_t kTailDenseTreeBytes = 8ll << 20;

/* the previous call's queue length at bounce b, for a call of n paths: the same image cut differently (the shares of
 * pimgpu_trace_image_host change size from one call to the next) keeps its proportions, so a history of another size within a
 * factor of two is scaled rather than dropped.  Sizes of launches and the switch to the tail kernel come from it, no result does. */
static bool history_fits(const PathGroup& g, int64_t n) { return g.historyBounces > 0 && g.historyN > 0 && n <= 2 * g.historyN && 2 * n >= g.historyN; }
static int64_t history_at(const PathGroup& g, int64_t n, int32_t b)
{ return g.historyN == n ? (int64_t)g.history[b] : (int64_t)((double)g.history[b] * (double)n / (double)g.historyN); }

static int32_t tail_policy(const PathGroup& g, int32_t n, int32_t bounceLimit)
{
    if (const char* e = getenv("PIMGPU_TAIL_BOUNCE")) { const int v = atoi(e); return v < 0 ? bounceLimit : std::min(v, bounceLimit); }
    /* Per path-bounce the tail kernel costs more than the batched launches at scale and less below a few million paths,
     * where the wavefront's launches sit near their latency floors and its queues cross HBM between them.  Swept with the
     * cooperative walk (traverse_coop) in place, switch at bounce 0 / 1 / 2: 1 M triangles 1080p (2.07 M paths) 4.66 / 4.95 /
     * 5.67 ms, Cornell 1080p 1.77 / 1.97 / 2.18, 1 M triangles 4K (8.3 M paths, 2.8 M at bounce 1) 15.9 / 13.85 / 13.87.
     * Knobs, not API: the threshold in paths and a divisor of the batch size (off = 1). */
    const int64_t threshold = std::min<int64_t>((int64_t)n / std::max(1, env_int("PIMGPU_TAIL_FRACTION", 1)), std::max(1, env_int("PIMGPU_TAIL_MAX_PATHS", 2500000)));
    if ((int64_t)n <= threshold) return 0;
    if (!history_fits(g, n)) return bounceLimit == 666 ? 16 : bounceLimit;     /* no history yet */
    for (int32_t b = 1; b < g.historyBounces && b < bounceLimit; ++b)
        if (history_at(g, n, b) < threshold) return b;
    return bounceLimit == 666 ? std::max(16, g.historyBounces) : bounceLimit;
}

/* the block a call leaves for the host (PathGroup::devBack -> hostBack): three separate copies from pageable memory were three
 * host round trips with the GPU idle in between (25 us each: 5 % of a Cornell 1080p sample) */
struct ReadBack { unsigned long long stats[8]; int32_t counters[64]; uint32_t traceStatus; float focal; unsigned long long prof[10]; };

__global__ void k_collect(ReadBack* __restrict__ out, const unsigned long long* __restrict__ stats, const int32_t* __restrict__ counters,
    uint32_t* __restrict__ status)
{
    const int i = threadIdx.x;
    if (i < 8) out->stats[i] = stats[i];
    if (i < 64) out->counters[i] = counters[i];
    if (i < 10) out->prof[i] = stats[8 + i];
    if (i == 0)
    {
        const uint32_t st = status[0];
        out->traceStatus = st;
        if (st & 1u) status[0] = st & ~1u;         /* reported once: cleared so that it cannot be blamed on the next call */
        out->focal = __uint_as_float(status[20]);
    }
}

/* the start of a call: queue length, counters and statistics (one launch instead of a copy from the host and a fill) */
__global__ void k_begin(int32_t* __restrict__ counters, unsigned long long* __restrict__ stats, int32_t n)
{
    const int i = threadIdx.x;
    if (i < 64) counters[i] = i == 0 ? n : 0;
    if (i < 32) stats[i] = 0ull;
}

/* Timing events are pooled, and the pool is BOUNDED: a call times its first kMaxTimedDispatches dispatches and launches the
 * rest plain (a roulette-only call without the tail kernel is 666 bounces x 7 launches: unbounded, that was ~9 300 live HIP
 * events per group, each holding a completion signal of the runtime, kept until pimgpu_shutdown).  After a call the pool
 * shrinks back to kKeepEvents (integrate_finish).  The bench workloads (8 bounces: <= 60 dispatches) are timed in full. */
static constexpr size_t kMaxTimedDispatches = 256, kKeepEvents = 128;

hipEvent_t PathGroup::take_event()
{
    if (eventsUsed >= 2 * kMaxTimedDispatches) return nullptr;
    if (eventsUsed == eventPool.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; eventPool.push_back(e); }
    return eventPool[eventsUsed++];
}

void PathGroup::trim_events()
{
    while (eventPool.size() > kKeepEvents) { hipEventDestroy(eventPool.back()); eventPool.pop_back(); }
    eventsUsed = 0;
}

int Library::ensure_group(int gi)
{
    PathGroup& g = groups[gi];
    if (gi == 0) { g.counters = work.counters; g.stats = work.stats; g.trace = &trace; }     /* the library's own, may have been reallocated */
    if (g.stream) return PIMGPU_OK;
    if (gi == 0) g.stream = stream;
    else
    {
        PT_HIP(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
        PT_HIP(hipMalloc(&g.counters, 256));
        PT_HIP(hipMalloc(&g.stats, 256));
        int rc = g.ownTrace.ensure(); if (rc) return rc;
        g.trace = &g.ownTrace;
    }
    PT_HIP(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    PT_HIP(hipEventCreateWithFlags(&g.stagger, hipEventDisableTiming));
    PT_HIP(hipMalloc(&g.devBack, sizeof(ReadBack)));
    PT_HIP(hipHostMalloc(&g.hostBack, sizeof(ReadBack), hipHostMallocDefault));
    return PIMGPU_OK;
}

void Library::release_groups()
{
    if (frameBegin) { hipEventDestroy(frameBegin); hipEventDestroy(frameEnd); if (copyBegin) hipEventDestroy(copyBegin); if (copyEnd) hipEventDestroy(copyEnd); frameBegin = frameEnd = copyBegin = copyEnd = nullptr; }
    for (int gi = 0; gi < kMaxGroups; ++gi)
    {
        PathGroup& g = groups[gi];
        if (!g.stream) continue;
        if (g.done) hipEventDestroy(g.done);
        if (g.stagger) hipEventDestroy(g.stagger);
        for (hipEvent_t e : g.eventPool) hipEventDestroy(e);
        if (g.devBack) hipFree(g.devBack);
        if (g.hostBack) hipHostFree(g.hostBack);
        if (gi > 0) { hipStreamSynchronize(g.stream); hipStreamDestroy(g.stream); hipFree(g.counters); hipFree(g.stats); g.ownTrace.release(); }
        g = PathGroup();
    }
}

/*
 * Enqueue every bounce of the paths at queue positions [first, first + n) of the work buffers (slots are global, the
 * position-indexed arrays are viewed from `first`) on the group's stream; nothing here waits for the device.
 * staggerAfter >= 0: record g.stagger once that many half-bounces (traversal + stage) have been enqueued.
 */
static int integrate_enqueue(PimScene* scene, PathGroup& g, int32_t first, int32_t n, int32_t maxBounces, bool timeTrace, int staggerAfter)
{
    WorkBuffers& w = g_lib.work;
    hipStream_t st = g.stream;
    const int32_t bounceLimit = (maxBounces > 0 && maxBounces < 666) ? maxBounces : 666;
    const int32_t tailBounce = tail_policy(g, n, bounceLimit);
    g.first = first; g.n = n; g.wavefrontBounces = 0; g.tailBounce = -1; g.events.clear(); g.eventsUsed = 0;
    { int rcT = g.trace->ensure(); if (rcT) return rcT; }      /* the spill area behind the kernels' traversal stacks: gridBlocks * 256 lanes */
    k_begin<<<1, 64, 0, st>>>(g.counters, g.stats, n);
    const int blocks = (int)std::min<int64_t>((int64_t)g_lib.cuCount * (2048 / kShadeThreads), ((int64_t)n + kShadeThreads - 1) / kShadeThreads);
    const int slowBlocks = std::max(1, blocks / 4);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    /* events attached to the dispatches themselves (no barrier packets, trace.hip): cls 0 traversal, 1 shade_a, 2 shade_b, 3 tail */
    auto timed = [&](int cls) { t0 = t1 = nullptr; if (timeTrace) { t0 = g.take_event(); t1 = t0 ? g.take_event() : nullptr; if (t0 && t1) g.events.push_back({ cls, t0, t1 }); else t0 = t1 = nullptr; } };
#define PT_LAUNCH(kernel, grid, block, ...) do { if (t0 && t1) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, st, t0, t1, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, st, __VA_ARGS__); } while (0)
    int rc = PIMGPU_OK;
    int halves = 0;
    bool staggered = staggerAfter < 0;
    auto half_done = [&]() { if (!staggered && ++halves >= staggerAfter) { hipEventRecord(g.stagger, st); staggered = true; } };
    if (!staggered && staggerAfter == 0) { hipEventRecord(g.stagger, st); staggered = true; }
    /* light-sampling rays as visibility queries of their light (stage_a, trace.hip TGT): the shorter walks pay for the extra
     * triangle test of the shading stage on trees that do not sit in the caches (measured: -4% of k_trace at 1M triangles,
     * +4% of k_shade_a on either scene) */
    const int forceTarget = env_int("PIMGPU_NEE_TARGET", -1);                   /* -1: by tree size; 0 / 1 force (tests) */
    const bool targeted = forceTarget >= 0 ? forceTarget != 0 : 64ll * scene->d.bvh.nodeCount + 48ll * scene->d.bvh.triCount > (8ll << 20);
    /* paths whose shade_a stage meets the rare heavy case (a medium collision) leave the wavefront and are finished by the tail
     * kernel at the end of the call, instead of a shade_a<SLOW> launch per bounce that costs its latency floor for a fraction of a
     * per cent of the paths (1 M triangles 4K: 0.19 ms per bounce).  Off when the caller forbids the tail kernel (tests of the SLOW
     * launches). */
    const bool park = tailBounce > 0 && !(getenv("PIMGPU_TAIL_BOUNCE") && atoi(getenv("PIMGPU_TAIL_BOUNCE")) < 0) && env_int("PIMGPU_PARK", 1) != 0;
    bool tailLaunched = false;
    auto launch_tail = [&](StageArgs& a, int32_t b, bool mainEmpty)
    {
        /* One lane per path of the expected queue (the previous call's length at this bounce), but no more waves than
         * the machine holds at k_tail's occupancy: lanes take further paths as theirs end */
        const int64_t expect = mainEmpty ? std::max<int64_t>(4096, n / 64) : ((history_fits(g, n) && b < g.historyBounces) ? std::min<int64_t>(n, history_at(g, n, b)) : (int64_t)n);
        a.tailLanes = std::max(1, std::min(64, env_int("PIMGPU_TAIL_LANES", 64)));
        a.tailMainEmpty = mainEmpty ? 1 : 0;
        const int64_t treeBytes = 64ll * scene->d.bvh.nodeCount + 48ll * scene->d.bvh.triCount;
        const int forceDense = env_int("PIMGPU_TAIL_DENSE", -1);            /* -1: by tree size; 0 / 1 force (tests) */
        const bool dense = forceDense >= 0 ? forceDense != 0 : treeBytes > kTailDenseTreeBytes;
        const int wavesPerCu = kTailWavesPerCu > 0 ? kTailWavesPerCu : (dense ? 16 : 12);
        const int64_t want = std::min<int64_t>((int64_t)g_lib.cuCount * wavesPerCu, (expect + a.tailLanes - 1) / a.tailLanes);
        const int tailBlocks = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)g.trace->gridBlocks * (256 / kTailThreads)));
        timed(3);
        if (dense) PT_LAUNCH(k_tail<4>, tailBlocks, kTailThreads, scene->d, a);
        else PT_LAUNCH(k_tail<3>, tailBlocks, kTailThreads, scene->d, a);
        if (!mainEmpty) g.tailBounce = b;
        tailLaunched = true;
        if (hipGetLastError() != hipSuccess) { set_error("integrator launch failed"); rc = PIMGPU_E_DEVICE; }
    };
    StageArgs last = {};
    for (int32_t b = 0; b < bounceLimit; ++b)
    {
        const int in = b & 1, out = in ^ 1;
        StageArgs a;
        a.ps = w.ps; a.queueIn = w.queue[in] + first; a.queueOut = w.queue[out] + first; a.counters = g.counters; a.inSel = in;
        a.extO = w.extO[in] + first; a.extD = w.extD[in] + first; a.extHit = w.extHit + first; a.neeHit = w.neeHit + first;
        a.neeO = w.neeO + first; a.neeD = w.neeD + first; a.neeHitOut = w.neeHit + first; a.neeTarget = targeted ? w.neeTarget + first : nullptr;
        a.parkSlot = park ? w.parkSlot + first : nullptr; a.parkO = w.parkO + first; a.parkD = w.parkD + first; a.tailMainEmpty = 0;
        a.extOOut = w.extO[out] + first; a.extDOut = w.extD[out] + first; a.results = w.results; a.stats = g.stats;
        a.bounce = b; a.bounceLimit = bounceLimit; a.slowA = w.slowA + first; a.slowB = w.slowB + first; a.tailLanes = 64; a.spill = g.trace->spill; a.stackCap = trace_stack_cap();
        if (b >= tailBounce)
        {
            /* the queue is short by now (or the caller forced it): one launch finishes every path */
            launch_tail(a, b, false);
            break;
        }
        ++g.wavefrontBounces;
        timed(0);
        if ((rc = trace_closest(scene->d.bvh, g.trace, a.extO, a.extD, w.extHit + first, n, g.counters + in, st, t0, t1))) break;
        timed(1); PT_LAUNCH(k_shade_a<false>, blocks, kShadeThreads, scene->d, a);
        if (!park) { timed(1); PT_LAUNCH(k_shade_a<true>, slowBlocks, kShadeThreads, scene->d, a); }
        half_done();
        timed(0);
        if ((rc = trace_closest(scene->d.bvh, g.trace, a.neeO, a.neeD, w.neeHit + first, n, g.counters + in, st, t0, t1, a.neeTarget))) break;
        timed(2); PT_LAUNCH(k_shade_b<false>, blocks, kShadeThreads, scene->d, a);
        timed(2); PT_LAUNCH(k_shade_b<true>, slowBlocks, kShadeThreads, scene->d, a);
        half_done();
        if (hipGetLastError() != hipSuccess) { set_error("integrator launch failed"); rc = PIMGPU_E_DEVICE; break; }
        last = a;
    }
    /* the wavefront ran to the bounce limit: the parked paths still have to be finished */
    if (rc == PIMGPU_OK && park && !tailLaunched && g.wavefrontBounces > 0) launch_tail(last, last.bounce, true);
    if (!staggered) hipEventRecord(g.stagger, st);
#undef PT_LAUNCH
    return rc;
}

/* wait for the group's paths; statistics, the traversal time and the queue-length history of the next call's tail policy */
static int integrate_finish(PathGroup& g, PimTraceStats* statsOut, float* traceMsOut)
{
    hipStream_t st = g.stream;
    /* the batched traversal launches report a full stack in their own status word (trace.hip), the stages and the tail
     * kernel in stats[4]: both end the call with PIMGPU_E_LIMIT */
    k_collect<<<1, 64, 0, st>>>((ReadBack*)g.devBack, g.stats, g.counters, g.trace->status);
    PT_HIP(hipMemcpyAsync(g.hostBack, g.devBack, sizeof(ReadBack), hipMemcpyDeviceToHost, st));
    PT_HIP(hipStreamSynchronize(st));
    const ReadBack& rb = *(const ReadBack*)g.hostBack;
    unsigned long long hs[8]; memcpy(hs, rb.stats, sizeof(hs));
    const int32_t* counts = rb.counters;
    if (rb.traceStatus & 1u) hs[4] |= 1ull;
#ifdef PT_TAIL_PROF
    const unsigned long long* prof = rb.prof;
    if (prof[4]) fprintf(stderr, "tailprof wave-bounces %llu | us per wave-bounce: ext %.1f stageA %.1f nee %.1f stageB %.1f | trips of the wave per walk: ext %.1f nee %.1f\n",
        prof[4], prof[0] * 0.01 / prof[4], prof[1] * 0.01 / prof[4], prof[2] * 0.01 / prof[4], prof[3] * 0.01 / prof[4],
        (double)prof[6] / prof[4], (double)prof[7] / prof[4]);
#endif
    float clsMs[4] = { 0.0f, 0.0f, 0.0f, 0.0f }; int32_t traceLaunches = 0;
    for (auto& e : g.events) { float ms = 0.0f; hipEventElapsedTime(&ms, e.t0, e.t1); clsMs[e.cls] += ms; if (e.cls == 0) ++traceLaunches; }
    const float traceMs = clsMs[0];
    g.events.clear();
    g.trim_events();
    if (hs[4] & 1ull) { set_error("path tracing: traversal stack overflow (tree deeper than %d entries)", PT_STACK_MAX); return PIMGPU_E_LIMIT; }
    /* queue lengths of the wavefront bounces of this call: the next call's tail policy.  A call that switched early keeps
     * what the history knew beyond its switch point. */
    if (g.historyN != g.n)
    {
        if (history_fits(g, g.n)) { for (int32_t b = 0; b < g.historyBounces; ++b) g.history[b] = (int32_t)history_at(g, g.n, b); }     /* another cut of the same image */
        else g.historyBounces = 0;
        g.historyN = g.n;
    }
    for (int32_t b = 0; b < g.wavefrontBounces && b < PT_COUNTS_MAX; ++b) g.history[b] = counts[PT_COUNTS_BASE + b];
    g.historyBounces = std::max(g.historyBounces, std::min<int32_t>(g.wavefrontBounces, PT_COUNTS_MAX));
    if (statsOut)
    {
        statsOut->extensionRays += hs[0]; statsOut->shadowRays += hs[1]; statsOut->otherRays += hs[2];
        statsOut->tailRays += hs[5]; statsOut->tailPathBounces += hs[6];
        statsOut->bounces += g.wavefrontBounces;
        statsOut->paths += (uint64_t)g.n;
        statsOut->shadeAMs += clsMs[1]; statsOut->shadeBMs += clsMs[2]; statsOut->tailMs += clsMs[3]; statsOut->traceLaunches += traceLaunches;
        for (int32_t b = 0; b < g.wavefrontBounces && b < PT_COUNTS_MAX; ++b) statsOut->wavefrontPathBounces += (uint64_t)counts[PT_COUNTS_BASE + b];
        if (g.tailBounce >= 0) statsOut->tailPaths += (uint64_t)counts[g.tailBounce & 1];
    }
    if (traceMsOut) *traceMsOut += traceMs;
    return PIMGPU_OK;
}

/* the whole batch as one group on the library's stream (rays, probes, lightmap texels) */
int integrate_paths(PimScene* scene, int32_t n, int32_t maxBounces, PimTraceStats* statsOut, float* traceMsOut)
{
    int rc = g_lib.ensure_group(0); if (rc) return rc;
    PathGroup& g = g_lib.groups[0];
    const bool timeTrace = traceMsOut != nullptr && !getenv("PIMGPU_NO_EVENTS");
    if (traceMsOut) *traceMsOut = 0.0f;
    if ((rc = integrate_enqueue(scene, g, 0, n, maxBounces, timeTrace, -1))) { hipStreamSynchronize(g.stream); return rc; }
    return integrate_finish(g, statsOut, traceMsOut);
}

/*
 * An image: the pixels split into contiguous shares, one group each, so that the shares' launches overlap (PathGroup).
 * Every path owns its slot and its Prng stream, and the scene-side accumulators the stages touch are integer atomics,
 * so how the image is split does not change a bit of the result (tests/test_gpu_tail.py sweeps it).
 * Share g + 1 starts once share g is `stagger` half-bounces in: the shares then sit in different phases.
 */
static constexpr int kDefaultGroups = 1, kDefaultStagger = 2, kDefaultGroupMinPaths = 1 << 18;

template <class Epilogue>
int integrate_image(PimScene* scene, int32_t n, int32_t maxBounces, PimTraceStats* statsOut, float* traceMsOut, Epilogue&& epilogue)
{
    /* read per call: the tests sweep them */
    const int kImageGroups = std::max(1, std::min(kMaxGroups, env_int("PIMGPU_GROUPS", kDefaultGroups)));
    const int kGroupStagger = env_int("PIMGPU_GROUP_STAGGER", kDefaultStagger);
    const int kGroupMinPaths = std::max(64, env_int("PIMGPU_GROUP_MIN_PATHS", kDefaultGroupMinPaths));
    int groups = (int)std::min<int64_t>(kImageGroups, std::max<int64_t>(1, n / kGroupMinPaths));
    if (g_lib.trace.counting) groups = 1;           /* the work counters live in the main stream's resources */
    int rc = PIMGPU_OK;
    for (int gi = 0; gi < groups; ++gi) if ((rc = g_lib.ensure_group(gi))) return rc;
    const bool timeTrace = traceMsOut != nullptr && !getenv("PIMGPU_NO_EVENTS");
    if (traceMsOut) *traceMsOut = 0.0f;
    hipStream_t main = g_lib.stream;
    if (groups <= 1)
    {
        /* the epilogue (the image resolve) is enqueued BEFORE the host waits for the statistics: no idle gap between the
         * last integrator launch and it (80 us of a 2.4 ms Cornell sample when it came after the wait) */
        PathGroup& g = g_lib.groups[0];
        if ((rc = integrate_enqueue(scene, g, 0, n, maxBounces, timeTrace, -1))) { hipStreamSynchronize(g.stream); return rc; }
        epilogue(main);
        return integrate_finish(g, statsOut, traceMsOut);
    }
    /* what the main stream has enqueued so far (path set-up) comes first for every share */
    hipEvent_t start = g_lib.groups[0].done;
    PT_HIP(hipEventRecord(start, main));
    const int32_t per = (int32_t)((((int64_t)n + groups - 1) / groups + 63) & ~(int64_t)63);
    int used = 0;
    for (int gi = 0; gi < groups && rc == PIMGPU_OK; ++gi)
    {
        const int32_t first = std::min<int64_t>(n, (int64_t)per * gi), count = std::min<int32_t>(per, n - first);
        if (count <= 0) break;
        PathGroup& g = g_lib.groups[gi];
        if (gi > 0)
        {
            PT_HIP(hipStreamWaitEvent(g.stream, start, 0));
            if (kGroupStagger >= 0) PT_HIP(hipStreamWaitEvent(g.stream, g_lib.groups[gi - 1].stagger, 0));
        }
        rc = integrate_enqueue(scene, g, first, count, maxBounces, timeTrace, kGroupStagger);
        if (gi > 0) { PT_HIP(hipEventRecord(g.done, g.stream)); PT_HIP(hipStreamWaitEvent(main, g.done, 0)); }
        ++used;
    }
    if (rc == PIMGPU_OK) epilogue(main);             /* after every share: the main stream has waited for the others */
    for (int gi = used - 1; gi >= 0; --gi)
    {
        const int r = integrate_finish(g_lib.groups[gi], statsOut, traceMsOut);
        if (rc == PIMGPU_OK) rc = r;
    }
    return rc;
}

} /* namespace pt */

using namespace pt;

static thread_local bool g_makingPeer = false;     /* pimgpu_trace_create is making a peer copy: no recursion into the other contexts */

extern "C" {

int pimgpu_trace_create(int32_t width, int32_t height, PimTrace** out)
{
    if (out) *out = nullptr;
    int rc = check_ready(); if (rc) return rc;
    if (!out || width <= 0 || height <= 0 || (int64_t)width * height > (1ll << 28)) { set_error("trace_create: bad size %dx%d", width, height); return PIMGPU_E_INVALID; }
    PimTrace* t = new PimTrace();
    t->width = width; t->height = height;
    const size_t bytes = sizeof(float) * 3 * (size_t)width * height;
    if (hipMalloc(&t->color, bytes) != hipSuccess || hipMalloc(&t->albedo, bytes) != hipSuccess || hipMalloc(&t->normal, bytes) != hipSuccess)
    {
        hipFree(t->color); hipFree(t->albedo); hipFree(t->normal); delete t;
        set_error("trace_create: out of device memory"); return PIMGPU_E_NOMEM;
    }
    *out = t;
    if ((rc = pimgpu_trace_clear(t))) return rc;
    if (g_libCount > 1 && !g_makingPeer)
    {
        /* one copy of the buffers per context; context i keeps its band of rows up to date */
        rc = for_each_context(g_libCount, [&](int i) {
            if (i == 0) return PIMGPU_OK;
            g_makingPeer = true;
            const int r = pimgpu_trace_create(width, height, &t->peers[i]);
            g_makingPeer = false;
            return r;
        });
        if (rc) { pimgpu_trace_destroy(t); *out = nullptr; }
    }
    return rc;
}

int pimgpu_trace_destroy(PimTrace* t)
{
    if (!t) return PIMGPU_OK;
    for (int i = 1; i < kMaxDevices; ++i)
        if (t->peers[i])
        {
            PimTrace* peer = t->peers[i]; t->peers[i] = nullptr;
            if (i < g_libCount) { DeviceScope scope(i); pimgpu_trace_destroy(peer); }
        }
    if (t->sampleAcc) { pimgpu_trace_destroy(t->sampleAcc); t->sampleAcc = nullptr; }
    if (t->sampleStage) { pimgpu_trace_destroy(t->sampleStage); t->sampleStage = nullptr; }
    if (g_lib.initialised) hipStreamSynchronize(g_lib.stream);
    hipFree(t->color); hipFree(t->albedo); hipFree(t->normal);
    delete t;
    return PIMGPU_OK;
}

int pimgpu_trace_clear(PimTrace* t)
{
    int rc = check_ready(); if (rc) return rc;
    if (!t) { set_error("trace_clear: NULL trace"); return PIMGPU_E_INVALID; }
    const size_t bytes = sizeof(float) * 3 * (size_t)t->width * t->height;
    PT_HIP(hipMemsetAsync(t->color, 0, bytes, g_lib.stream));
    PT_HIP(hipMemsetAsync(t->albedo, 0, bytes, g_lib.stream));
    PT_HIP(hipMemsetAsync(t->normal, 0, bytes, g_lib.stream));
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    memset(&t->stats, 0, sizeof(t->stats));
    t->gathered = true;
    for (int i = 1; i < g_libCount; ++i)
        if (t->peers[i]) { DeviceScope scope(i); if ((rc = pimgpu_trace_clear(t->peers[i]))) return rc; }
    return PIMGPU_OK;
}

/* one context renders the pixels [pixelBegin, pixelBegin + n) of the image into its copy of the PtTrace buffers */
/* hostColor / hostAlbedo / hostNormal (any may be null; all null: nothing is copied): the caller's float3[W*H] host images.  With
 * them the band is rendered in SHARES of pixels, one after the other, each on its own stream, and a share's part of the images
 * is copied to the host behind its resolve -- while the next share is being traced.  A path depends on nothing but its own
 * Prng stream and the scene, so the shares do not change a bit of the image (the groups of integrate_image, run in sequence). */
static int trace_image_one(PimScene* scene, PimTrace* t, const PimCamera* cam, PimDofInfo* dof, const PimTraceParams* p, int32_t pixelBegin, int32_t n,
    float* hostColor = nullptr, float* hostAlbedo = nullptr, float* hostNormal = nullptr, bool distsDone = false)
{
    int rc = check_ready(); if (rc) return rc;
    memset(&t->stats, 0, sizeof(t->stats));
    if (n <= 0) return PIMGPU_OK;                         /* more contexts than rows */
    hipStream_t st = g_lib.stream;
    WorkBuffers& w = g_lib.work;
    if ((rc = w.ensure_paths(n))) return rc;
    if (!g_lib.frameBegin) { PT_HIP(hipEventCreate(&g_lib.frameBegin)); PT_HIP(hipEventCreate(&g_lib.frameEnd)); PT_HIP(hipEventCreate(&g_lib.copyBegin)); PT_HIP(hipEventCreate(&g_lib.copyEnd)); }
    hipEvent_t e0 = g_lib.frameBegin, e1 = g_lib.frameEnd;
    PT_HIP(hipEventRecord(e0, st));
    if (p->updateDists && !distsDone) { if ((rc = update_dists(scene->d, st))) return rc; }      /* PtScene_Update, pt.c:1034 */

    const v4 rot = f4_v(cam->rotation[0], cam->rotation[1], cam->rotation[2], cam->rotation[3]);
    const v4 eye = f4_v(cam->position[0], cam->position[1], cam->position[2], cam->position[3]);
    uint64_t otherRays = 0;
    float* focalDev = nullptr;
    float focalHost = dof->focalLength;
    if (dof->autoFocus)                                                             /* DofUpdate, pt.c:2485-2503 */
    {
        const v4 fwd = quat_fwd(rot);
        const float4 ro = make_float4(eye.x, eye.y, eye.z, 0.0f), rd = make_float4(fwd.x, fwd.y, fwd.z, PT_RCP_EPS);
        if ((rc = g_lib.trace.ensure())) return rc;
        focalDev = (float*)g_lib.trace.status + 20;                               /* spare words of the status block */
        const float tt = 1.0f - pl_expf(-p->frameDt * dof->autoFocusSpeed);
        k_focus<<<1, 64, 0, st>>>(scene->d.bvh, ro, rd, dof->focalLength, tt, focalDev, g_lib.trace.spill, g_lib.trace.status, trace_stack_cap());
        otherRays = 1;
    }
    CameraArgs c;
    c.eye = eye; c.right = quat_right(rot); c.up = quat_up(rot); c.fwd = quat_fwd(rot);
    {
        const float fovy = f1_radians(cam->fovy), aspect = (float)t->width / (float)t->height;
        const float tn = pl_tanf(fovy * 0.5f);                                      /* proj_slope, frustum.h:26-30 */
        c.slope = f2_v(aspect * tn, tn);
    }
    c.dof = *dof; c.width = t->width; c.height = t->height; c.sampleIndex = p->sampleIndex; c.seed = p->seed; c.focalDev = focalDev;
    c.pixelBegin = pixelBegin; c.pixelCount = n; c.slotBegin = 0;
    PimTraceStats stats; memset(&stats, 0, sizeof(stats));
    float traceMs = 0.0f;
    int32_t copyPixels = 0;                                                       /* pixels of the share whose copies copyEnd follows */
    const bool toHost = hostColor || hostAlbedo || hostNormal;
    /* shares of at least 256 k pixels, 2 by default (PIMGPU_HOST_SHARES; measured at 1080p, 1 / 2 / 4 / 8 shares: 3.17 / 2.80 / 3.27 /
     * 4.29 ms per call -- every share's launch ends with its own slowest waves) */
    int shares = toHost ? std::max(1, std::min(kMaxGroups, env_int("PIMGPU_HOST_SHARES", 2))) : 1;
    shares = (int)std::min<int64_t>(shares, std::max<int64_t>(1, n / std::max(64, env_int("PIMGPU_HOST_SHARE_MIN", 1 << 18))));
    if (g_lib.trace.counting) shares = 1;            /* the work counters live in the main stream's resources */
    if (shares > 1)
    {
        const bool timeTrace = p->updateDists >= 0 && !getenv("PIMGPU_NO_EVENTS");
        for (int k = 0; k < shares; ++k) if ((rc = g_lib.ensure_group(k))) return rc;
        /* the first share is the larger one: the copy that is left over at the end is the LAST share's.  With t the time to trace a
         * pixel and c the time to copy one home, the first share's copy hides behind the second share's tracing up to the fraction
         * f = t / (t + c) of the pixels; both are measured on this PtTrace's previous call (c from the last share's copies, so a
         * pageable host image shows as the slower copy it is), and the smaller second launch traces a little slower per pixel, which
         * moves the measured optimum up by 0.02 ("Pt_Trace through the host layer").  PIMGPU_HOST_FIRST
         * (eighths of an equal part) sets the size by hand; without history it is 10 = 5/4 of an equal part. */
        int64_t firstWant = (int64_t)n * 10 / (8 * shares);
        if (getenv("PIMGPU_HOST_FIRST")) firstWant = (int64_t)n * std::max(1, std::min(8 * shares - 1, env_int("PIMGPU_HOST_FIRST", 10))) / (8 * shares);
        else if (shares == 2 && t->hostTraceNs > 0.0f && t->hostCopyNs > 0.0f)
        {
            firstWant = (int64_t)((double)n * std::min(0.875, std::max(0.5, (double)t->hostTraceNs / ((double)t->hostTraceNs + (double)t->hostCopyNs) + 0.02)));
            /* within 3 % of the previous call's size: that one again (the launches' sizes follow the previous call's queue lengths) */
            if (t->hostN == n && std::abs(firstWant - (int64_t)t->hostFirst) * 100 < 3 * (int64_t)n) firstWant = t->hostFirst;
        }
        const int32_t firstCount = (int32_t)std::min<int64_t>(n, (firstWant + 63) & ~(int64_t)63);
        t->hostN = n; t->hostFirst = firstCount;
        const int32_t per = shares > 1 ? (int32_t)((((int64_t)(n - firstCount) + shares - 2) / (shares - 1) + 63) & ~(int64_t)63) : n;
        int used = 0;
        for (int k = 0; k < shares && rc == PIMGPU_OK; ++k)
        {
            const int32_t first = k == 0 ? 0 : (int32_t)std::min<int64_t>(n, firstCount + (int64_t)per * (k - 1));
            const int32_t count = std::min<int32_t>(k == 0 ? firstCount : per, n - first);
            if (count <= 0) break;
            const bool last = first + count >= n;
            PathGroup& g = g_lib.groups[k];
            hipStream_t s = g.stream;                                     /* share 0: the library's stream, with the set-up above */
            /* a share starts when the one before it has been resolved (unchained, the shares' launches interleave and all end together:
             * 3.2 instead of 2.8 ms at 1080p) */
            if (k > 0 && hipStreamWaitEvent(s, g_lib.groups[k - 1].done, 0) != hipSuccess) { set_error("trace_image: hipStreamWaitEvent failed"); rc = PIMGPU_E_DEVICE; break; }
            c.pixelBegin = pixelBegin + first; c.pixelCount = count; c.slotBegin = first;
            k_init_image<<<(count + 255) / 256, 256, 0, s>>>(w.ps, w.extO[0], w.extD[0], w.queue[0], c);
            if ((rc = integrate_enqueue(scene, g, first, count, p->maxBounces, timeTrace, -1))) break;
            k_resolve_image<<<(count + 255) / 256, 256, 0, s>>>(w.results + first, pixelBegin + first, count, p->sampleWeight, t->color, t->albedo, t->normal);
            ++used;                                                       /* from here on the share's stream holds work: it is drained below */
            if (hipEventRecord(g.done, s) != hipSuccess) { set_error("trace_image: hipEventRecord failed"); rc = PIMGPU_E_DEVICE; break; }
            if (last) hipEventRecord(e1, s);
            /* this share's pixels go home while the next share is traced */
            if (k == 0 && timeTrace) hipEventRecord(g_lib.copyBegin, s);
            const size_t off = (size_t)(pixelBegin + first) * 3, bytes = sizeof(float) * 3 * (size_t)count;
            if (hostColor) hipMemcpyAsync(hostColor + off, t->color + off, bytes, hipMemcpyDeviceToHost, s);
            if (hostAlbedo) hipMemcpyAsync(hostAlbedo + off, t->albedo + off, bytes, hipMemcpyDeviceToHost, s);
            if (hostNormal) hipMemcpyAsync(hostNormal + off, t->normal + off, bytes, hipMemcpyDeviceToHost, s);
            if (k == 0 && timeTrace) { hipEventRecord(g_lib.copyEnd, s); copyPixels = count; }
        }
        for (int k = used - 1; k >= 0; --k)
        {
            float ms = 0.0f;
            const int r = integrate_finish(g_lib.groups[k], &stats, timeTrace ? &ms : nullptr);      /* waits for the share's stream, copies included */
            traceMs += ms;
            if (rc == PIMGPU_OK) rc = r;
        }
        /* on failure also the share whose enqueue failed: its stream may hold launches that use the work buffers */
        if (rc) { for (int k = 0; k <= used && k < shares; ++k) hipStreamSynchronize(g_lib.groups[k].stream); return rc; }
    }
    else
    {
    if (n > 0) k_init_image<<<(n + 255) / 256, 256, 0, st>>>(w.ps, w.extO[0], w.extD[0], w.queue[0], c);
    if ((rc = integrate_image(scene, n, p->maxBounces, &stats, p->updateDists >= 0 ? &traceMs : nullptr, [&](hipStream_t s)
        { k_resolve_image<<<(n + 255) / 256, 256, 0, s>>>(w.results, pixelBegin, n, p->sampleWeight, t->color, t->albedo, t->normal); hipEventRecord(e1, s);
          if (toHost)
          {
              const size_t off = (size_t)pixelBegin * 3, bytes = sizeof(float) * 3 * (size_t)n;
              if (hostColor) hipMemcpyAsync(hostColor + off, t->color + off, bytes, hipMemcpyDeviceToHost, s);
              if (hostAlbedo) hipMemcpyAsync(hostAlbedo + off, t->albedo + off, bytes, hipMemcpyDeviceToHost, s);
              if (hostNormal) hipMemcpyAsync(hostNormal + off, t->normal + off, bytes, hipMemcpyDeviceToHost, s);
          } }))) return rc;
    }
    PT_HIP(hipEventSynchronize(e1));
    if (focalDev) focalHost = ((const ReadBack*)g_lib.groups[0].hostBack)->focal;  /* came home with group 0's statistics (k_collect) */
    dof->focalLength = focalHost;                                                  /* DofUpdate updates the caller's struct */
    float totalMs = 0.0f; hipEventElapsedTime(&totalMs, e0, e1);
    stats.otherRays += otherRays;
    stats.traceMs = traceMs; stats.totalMs = totalMs;
    t->stats = stats;
    if (copyPixels > 0)                                                            /* what the next call sizes its first share with */
    {
        float copyMs = 0.0f;
        if (hipEventElapsedTime(&copyMs, g_lib.copyBegin, g_lib.copyEnd) == hipSuccess && copyMs > 0.0f && totalMs > 0.0f)
        {
            const float tn = totalMs * 1e6f / (float)n, cn = copyMs * 1e6f / (float)copyPixels;
            t->hostTraceNs = t->hostTraceNs > 0.0f ? 0.5f * (t->hostTraceNs + tn) : tn;
            t->hostCopyNs = t->hostCopyNs > 0.0f ? 0.5f * (t->hostCopyNs + cn) : cn;
        }
        if (getenv("PIMGPU_HOST_LOG")) fprintf(stderr, "host: total %.3f ms copy %.3f ms of %d px of %d -> t %.3f c %.3f ns\n", totalMs, copyMs, copyPixels, n, t->hostTraceNs, t->hostCopyNs);
    }
    return PIMGPU_OK;
}

int pimgpu_trace_image(PimScene* scene, PimTrace* t, const PimCamera* cam, PimDofInfo* dof, const PimTraceParams* p)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || !t || !cam || !dof || !p) { set_error("trace_image: NULL argument"); return PIMGPU_E_INVALID; }
    if (dof->bladeCount <= 0) { set_error("trace_image: dof.bladeCount must be positive"); return PIMGPU_E_INVALID; }
    const int G = g_libCount;
    if (G <= 1 || !scene->peers[1] || !t->peers[1]) return trace_image_one(scene, t, cam, dof, p, 0, t->width * t->height);
    /*
     * Several devices: the image is cut into G bands of rows, context i renders band i with its replica of the scene into its
     * copy of the buffers.  A pixel's path depends on nothing but its own Prng stream and the scene, and the light
     * histograms are summed over the replicas before every UpdateDists (scene_update_dists_all), so the image equals the
     * one-device image bit for bit.  No exchange step here: every device copies its own band to the host (pimgpu_trace_download).
     */
    if (p->updateDists) { if ((rc = scene_update_dists_all(scene))) return rc; }
    const PimTraceParams q = *p;                          /* UpdateDists ran above (distsDone); the sign of updateDists still says "no per-dispatch timing" */
    PimDofInfo dofs[kMaxDevices];
    for (int i = 0; i < G; ++i) dofs[i] = *dof;
    rc = for_each_context(G, [&](int i) {
        int32_t r0, r1; band_of(t->height, i, G, &r0, &r1);
        return trace_image_one(i == 0 ? scene : scene->peers[i], i == 0 ? t : t->peers[i], cam, &dofs[i], &q, r0 * t->width, (r1 - r0) * t->width,
            nullptr, nullptr, nullptr, true);
    });
    if (rc) return rc;
    *dof = dofs[0];                                       /* every context traced the same autofocus ray */
    PimTraceStats sum = t->stats;
    for (int i = 1; i < G; ++i)
    {
        const PimTraceStats& a = t->peers[i]->stats;
        sum.extensionRays += a.extensionRays; sum.shadowRays += a.shadowRays; sum.otherRays += a.otherRays - (dof->autoFocus ? 1 : 0);
        sum.paths += a.paths; sum.tailRays += a.tailRays; sum.wavefrontPathBounces += a.wavefrontPathBounces; sum.tailPaths += a.tailPaths;
        sum.tailPathBounces += a.tailPathBounces; sum.traceLaunches += a.traceLaunches;
        sum.bounces = std::max(sum.bounces, a.bounces);
        /* device times: the slowest context is the frame */
        sum.traceMs = std::max(sum.traceMs, a.traceMs); sum.totalMs = std::max(sum.totalMs, a.totalMs);
        sum.shadeAMs = std::max(sum.shadeAMs, a.shadeAMs); sum.shadeBMs = std::max(sum.shadeBMs, a.shadeBMs); sum.tailMs = std::max(sum.tailMs, a.tailMs);
    }
    t->stats = sum;
    t->gathered = false;
    return PIMGPU_OK;
}

/* ---- sample-index sharding over the device list ---- */

static int trace_gather(const PimTrace* t);

__global__ __launch_bounds__(256) void k_mean_scale(float* __restrict__ acc, int64_t n, float w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] = acc[i] * w;
}
__global__ __launch_bounds__(256) void k_mean_add(float* __restrict__ acc, const float* __restrict__ x, int64_t n, float w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] = acc[i] + x[i] * w;
}
__global__ __launch_bounds__(256) void k_mean_blend(float* __restrict__ image, const float* __restrict__ mean, int64_t n, float w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) image[i] = f1_lerp(image[i], mean[i], w);
}

/* a PimTrace of this context alone (no copies on the other contexts) */
static int trace_create_local(int32_t width, int32_t height, PimTrace** out)
{
    const bool was = g_makingPeer;
    g_makingPeer = true;
    const int rc = pimgpu_trace_create(width, height, out);
    g_makingPeer = was;
    return rc;
}

static void stats_add(PimTraceStats& sum, const PimTraceStats& a)
{
    sum.extensionRays += a.extensionRays; sum.shadowRays += a.shadowRays; sum.otherRays += a.otherRays;
    sum.paths += a.paths; sum.tailRays += a.tailRays; sum.wavefrontPathBounces += a.wavefrontPathBounces; sum.tailPaths += a.tailPaths;
    sum.tailPathBounces += a.tailPathBounces; sum.traceLaunches += a.traceLaunches;
    sum.bounces = std::max(sum.bounces, a.bounces);
}

/*
 * BASELINE config 3's sharding ("64 spp, sharded by sample index") inside one process.  pimgpu_trace_image cuts ONE sample's
 * pixels into bands of rows; here whole samples go to the contexts: in round r context i renders sample r*G + i of the S asked
 * for, the whole image, into its own running mean (weight 1 / (r + 1)), after the replicas' light histograms have been summed
 * and UpdateDists has run on every replica (scene_update_dists_all) -- the schedule bench.py's ranks follow, so each context's
 * mean equals the image of a process that renders those samples with the same distributions, bit for bit.  Then ONE exchange
 * step: the means travel to context 0 (device-to-device copies over xGMI) and are summed there in context order with weights
 * n_i / S (n_i samples on context i), acc = mean_0 * w_0, acc = acc + mean_i * w_i, and the sum is blended into the PtTrace image
 * as Pt_Trace blends a sample: image = lerp(image, acc, sampleWeight).  Every context's band of rows is refreshed from the result,
 * so calls of pimgpu_trace_image may follow.  The autofocus state runs per context (every context sees its own samples' frames);
 * the caller gets context 0's.  Plain peer copies and one kernel per image, not RCCL: the contexts live in one process, and a device
 * may appear twice in the list.
 */
int pimgpu_trace_image_samples(PimScene* scene, PimTrace* t, const PimCamera* cam, PimDofInfo* dof, const PimTraceParams* p, int32_t sampleCount)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || !t || !cam || !dof || !p) { set_error("trace_image_samples: NULL argument"); return PIMGPU_E_INVALID; }
    if (dof->bladeCount <= 0) { set_error("trace_image_samples: dof.bladeCount must be positive"); return PIMGPU_E_INVALID; }
    if (sampleCount <= 0 || sampleCount > (1 << 20)) { set_error("trace_image_samples: sampleCount %d out of range", sampleCount); return PIMGPU_E_INVALID; }
    const int G = (g_libCount > 1 && scene->peers[1] && t->peers[1]) ? g_libCount : 1;
    const int32_t W = t->width, H = t->height;
    const int64_t len = (int64_t)W * H * 3;
    const size_t bytes = sizeof(float) * (size_t)len;
    auto of = [&](int i) { return i == 0 ? t : t->peers[i]; };

    rc = for_each_context(G, [&](int i) {
        PimTrace* ti = of(i);
        int r = PIMGPU_OK;
        if (!ti->sampleAcc) r = trace_create_local(W, H, &ti->sampleAcc);
        if (!r && i == 0 && G > 1 && !ti->sampleStage) r = trace_create_local(W, H, &ti->sampleStage);
        if (r) return r;
        /* a running mean starts from zero: 0 + (sample - 0) * 1 is the sample exactly */
        PT_HIP(hipMemsetAsync(ti->sampleAcc->color, 0, bytes, g_lib.stream));
        PT_HIP(hipMemsetAsync(ti->sampleAcc->albedo, 0, bytes, g_lib.stream));
        PT_HIP(hipMemsetAsync(ti->sampleAcc->normal, 0, bytes, g_lib.stream));
        return PIMGPU_OK;
    });
    if (rc) return rc;

    PimDofInfo dofs[kMaxDevices];
    PimTraceStats per[kMaxDevices];
    for (int i = 0; i < G; ++i) { dofs[i] = *dof; memset(&per[i], 0, sizeof(per[i])); }
    PimTraceStats total; memset(&total, 0, sizeof(total));
    const int rounds = (sampleCount + G - 1) / G;
    for (int r = 0; r < rounds; ++r)
    {
        if (G > 1 && p->updateDists) { if ((rc = scene_update_dists_all(scene))) return rc; }
        PimTraceStats round[kMaxDevices];
        for (int i = 0; i < G; ++i) memset(&round[i], 0, sizeof(round[i]));
        rc = for_each_context(G, [&](int i) {
            const int k = r * G + i;
            if (k >= sampleCount) return PIMGPU_OK;
            PimTraceParams q = *p;
            q.sampleIndex = p->sampleIndex + (uint32_t)k;
            q.sampleWeight = 1.0f / (float)(r + 1);
            PimTrace* acc = of(i)->sampleAcc;
            const int r2 = trace_image_one(i == 0 ? scene : scene->peers[i], acc, cam, &dofs[i], &q, 0, W * H, nullptr, nullptr, nullptr, G > 1);
            if (!r2) round[i] = acc->stats;
            return r2;
        });
        if (rc) return rc;
        float traceMs = 0.0f, totalMs = 0.0f, aMs = 0.0f, bMs = 0.0f, tailMs = 0.0f;
        for (int i = 0; i < G; ++i)
        {
            stats_add(total, round[i]);
            /* device times: the slowest context is the round */
            traceMs = std::max(traceMs, round[i].traceMs); totalMs = std::max(totalMs, round[i].totalMs);
            aMs = std::max(aMs, round[i].shadeAMs); bMs = std::max(bMs, round[i].shadeBMs); tailMs = std::max(tailMs, round[i].tailMs);
        }
        total.traceMs += traceMs; total.totalMs += totalMs; total.shadeAMs += aMs; total.shadeBMs += bMs; total.tailMs += tailMs;
    }

    /* the exchange step, on context 0 (the caller's thread is on context 0's device: for_each_context's threads have ended);
     * the image it blends into is whole there first (after pimgpu_trace_image every context holds its own band only) */
    if ((rc = trace_gather(t))) return rc;
    {
        hipStream_t s0 = g_lib.stream;
        const unsigned grid = (unsigned)((len + 255) / 256);
        PimTrace* acc = t->sampleAcc;
        float* sums[3] = { acc->color, acc->albedo, acc->normal };
        float* images[3] = { t->color, t->albedo, t->normal };
        for (int c = 0; c < 3; ++c)
        {
            const int32_t n0 = (sampleCount + G - 1) / G;                         /* context 0 renders one sample in every round */
            k_mean_scale<<<grid, 256, 0, s0>>>(sums[c], len, (float)n0 / (float)sampleCount);
            for (int i = 1; i < G; ++i)
            {
                const int32_t ni = (sampleCount - i + G - 1) / G;
                if (i >= sampleCount || ni <= 0) continue;
                const PimTrace* src = t->peers[i]->sampleAcc;
                const float* from[3] = { src->color, src->albedo, src->normal };
                float* stage[3] = { t->sampleStage->color, t->sampleStage->albedo, t->sampleStage->normal };
                PT_HIP(hipMemcpyAsync(stage[c], from[c], bytes, hipMemcpyDeviceToDevice, s0));
                k_mean_add<<<grid, 256, 0, s0>>>(sums[c], stage[c], len, (float)ni / (float)sampleCount);
            }
            k_mean_blend<<<grid, 256, 0, s0>>>(images[c], sums[c], len, p->sampleWeight);
            /* every context keeps its band of rows of the image up to date (pimgpu_trace_image) */
            for (int i = 1; i < G; ++i)
            {
                int32_t r0, r1; band_of(H, i, G, &r0, &r1);
                const size_t off = (size_t)r0 * W * 3, bandBytes = sizeof(float) * 3 * (size_t)(r1 - r0) * W;
                float* to[3] = { t->peers[i]->color, t->peers[i]->albedo, t->peers[i]->normal };
                if (bandBytes) PT_HIP(hipMemcpyAsync(to[c] + off, images[c] + off, bandBytes, hipMemcpyDeviceToDevice, s0));
            }
        }
        PT_HIP(hipGetLastError());
        PT_HIP(hipStreamSynchronize(s0));
    }
    *dof = dofs[0];
    t->stats = total;
    t->gathered = true;
    return PIMGPU_OK;
}

int pimgpu_trace_download(const PimTrace* t, float* color, float* albedo, float* normal);

/* Pt_Trace as its callers use it (render_system.c:306-352 read the host images right after the call): the sample and the
 * copy of the three images to the host in one call, the copy of one share of the pixels running under the tracing of the
 * next (trace_image_one).  Same images, statistics and autofocus state as pimgpu_trace_image + pimgpu_trace_download. */
int pimgpu_trace_image_host(PimScene* scene, PimTrace* t, const PimCamera* cam, PimDofInfo* dof, const PimTraceParams* p,
    float* color, float* albedo, float* normal)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || !t || !cam || !dof || !p) { set_error("trace_image_host: NULL argument"); return PIMGPU_E_INVALID; }
    if (dof->bladeCount <= 0) { set_error("trace_image_host: dof.bladeCount must be positive"); return PIMGPU_E_INVALID; }
    if (g_libCount <= 1 || !scene->peers[1] || !t->peers[1])
        return trace_image_one(scene, t, cam, dof, p, 0, t->width * t->height, color, albedo, normal);
    /* several devices: every device copies its own band (pimgpu_trace_download) */
    if ((rc = pimgpu_trace_image(scene, t, cam, dof, p))) return rc;
    return (color || albedo || normal) ? pimgpu_trace_download(t, color, albedo, normal) : PIMGPU_OK;
}

/* multi-device: bring every band into the copy of context 0 (stddev, device-side readers) */
static int trace_gather(const PimTrace* t)
{
    const int G = g_libCount;
    if (G <= 1 || !t->peers[1] || t->gathered) return PIMGPU_OK;
    for (int i = 1; i < G; ++i)
    {
        int32_t r0, r1; band_of(t->height, i, G, &r0, &r1);
        const size_t off = (size_t)r0 * t->width * 3, bytes = sizeof(float) * 3 * (size_t)(r1 - r0) * t->width;
        if (!bytes) continue;
        { DeviceScope scope(i); PT_HIP(hipStreamSynchronize(g_lib.stream)); }
        const PimTrace* src = t->peers[i];
        PT_HIP(hipMemcpyAsync(t->color + off, src->color + off, bytes, hipMemcpyDeviceToDevice, g_lib.stream));
        PT_HIP(hipMemcpyAsync(t->albedo + off, src->albedo + off, bytes, hipMemcpyDeviceToDevice, g_lib.stream));
        PT_HIP(hipMemcpyAsync(t->normal + off, src->normal + off, bytes, hipMemcpyDeviceToDevice, g_lib.stream));
    }
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    const_cast<PimTrace*>(t)->gathered = true;
    return PIMGPU_OK;
}

int pimgpu_trace_download(const PimTrace* t, float* color, float* albedo, float* normal)
{
    if (t && g_libCount > 1 && t->peers[1] && !t->gathered)
    {
        /* every device sends its own band straight to the host images, all links at once */
        int rc0 = check_ready(); if (rc0) return rc0;
        return for_each_context(g_libCount, [&](int i) {
            int32_t r0, r1; band_of(t->height, i, g_libCount, &r0, &r1);
            const size_t off = (size_t)r0 * t->width * 3, bytes = sizeof(float) * 3 * (size_t)(r1 - r0) * t->width;
            const PimTrace* src = i == 0 ? t : t->peers[i];
            if (!bytes) return PIMGPU_OK;
            if (color) PT_HIP(hipMemcpyAsync(color + off, src->color + off, bytes, hipMemcpyDeviceToHost, g_lib.stream));
            if (albedo) PT_HIP(hipMemcpyAsync(albedo + off, src->albedo + off, bytes, hipMemcpyDeviceToHost, g_lib.stream));
            if (normal) PT_HIP(hipMemcpyAsync(normal + off, src->normal + off, bytes, hipMemcpyDeviceToHost, g_lib.stream));
            PT_HIP(hipStreamSynchronize(g_lib.stream));
            return PIMGPU_OK;
        });
    }

    int rc = check_ready(); if (rc) return rc;
    if (!t) { set_error("trace_download: NULL trace"); return PIMGPU_E_INVALID; }
    const size_t bytes = sizeof(float) * 3 * (size_t)t->width * t->height;
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    if (color) PT_HIP(hipMemcpy(color, t->color, bytes, hipMemcpyDeviceToHost));
    if (albedo) PT_HIP(hipMemcpy(albedo, t->albedo, bytes, hipMemcpyDeviceToHost));
    if (normal) PT_HIP(hipMemcpy(normal, t->normal, bytes, hipMemcpyDeviceToHost));
    return PIMGPU_OK;
}

int pimgpu_trace_copy_dev(const PimTrace* t, void* colorDst, void* albedoDst, void* normalDst)
{
    int rc = check_ready(); if (rc) return rc;
    if (!t) { set_error("trace_copy_dev: NULL trace"); return PIMGPU_E_INVALID; }
    if ((rc = trace_gather(t))) return rc;
    const size_t bytes = sizeof(float) * 3 * (size_t)t->width * t->height;
    if (colorDst) PT_HIP(hipMemcpyAsync(colorDst, t->color, bytes, hipMemcpyDeviceToDevice, g_lib.stream));
    if (albedoDst) PT_HIP(hipMemcpyAsync(albedoDst, t->albedo, bytes, hipMemcpyDeviceToDevice, g_lib.stream));
    if (normalDst) PT_HIP(hipMemcpyAsync(normalDst, t->normal, bytes, hipMemcpyDeviceToDevice, g_lib.stream));
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    return PIMGPU_OK;
}

int pimgpu_trace_buffers(const PimTrace* t, void** colorDev, void** albedoDev, void** normalDev)
{
    if (!t) { set_error("trace_buffers: NULL trace"); return PIMGPU_E_INVALID; }
    { const int rcg = trace_gather(t); if (rcg) return rcg; }
    if (colorDev) *colorDev = t->color;
    if (albedoDev) *albedoDev = t->albedo;
    if (normalDev) *normalDev = t->normal;
    return PIMGPU_OK;
}

int pimgpu_trace_stddev(const PimTrace* t, float* meanOut, float* stddevOut)
{
    int rc = check_ready(); if (rc) return rc;
    if (!t) { set_error("trace_stddev: NULL trace"); return PIMGPU_E_INVALID; }
    if ((rc = trace_gather(t))) return rc;
    float* outDev = (float*)g_lib.trace.status + 16;       /* spare words of the status block */
    const int32_t len = t->width * t->height;
    WorkBuffers& w = g_lib.work;
    const int64_t headFloats = (fs_head_floats(len) + 63) / 64 * 64;
    if ((rc = w.ensure_rays((len + headFloats + 3) / 4 + 1))) return rc;  /* head + len floats of scratch in the packed-ray buffer */
    float* head = (float*)w.rayO;
    float* terms = head + headFloats;
    stddev_sum_launch(t->color, len, nullptr, terms, head, outDev, 0, g_lib.stream);
    stddev_sum_launch(t->color, len, outDev, terms, head, outDev + 1, 1, g_lib.stream);
    float h[2];
    PT_HIP(hipMemcpyAsync(h, outDev, sizeof(h), hipMemcpyDeviceToHost, g_lib.stream));
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    if (meanOut) *meanOut = h[0];
    if (stddevOut) *stddevOut = h[1];
    return PIMGPU_OK;
}

/* pimgpu_trace_stddev without any allocation: the same two kernels per sum, their terms in this GPU's copy of the
 * file-scope `scratch` array (address taken at pimgpu_init).  No hipMalloc / hipFree / allocating routine is called; the two
 * results come back through a page-locked block made at init.  The call copies to the host and waits for its stream, so it is
 * NOT for use under graph capture.  The array belongs to the GPU, not to a context: every context on one GPU (a device listed
 * twice) and every concurrent caller share it, so calls that overlap on one GPU race; nothing needs clearing (terms [0, len)
 * are all rewritten). */
int pimgpu_trace_stddev_noalloc(const PimTrace* t, float* meanOut, float* stddevOut)
{
    int rc = check_ready(); if (rc) return rc;
    if (!t) { set_error("trace_stddev_noalloc: NULL trace"); return PIMGPU_E_INVALID; }
    if ((rc = trace_gather(t))) return rc;                  /* one context: nothing to do, nothing allocated */
    const int64_t len64 = (int64_t)t->width * t->height;
    if (len64 > kStddevNoAllocMaxPixels)
    {
        set_error("trace_stddev_noalloc: %dx%d = %lld pixels exceed the static scratch of %d pixels (3840x2160); use pimgpu_trace_stddev",
            t->width, t->height, (long long)len64, kStddevNoAllocMaxPixels);
        return PIMGPU_E_LIMIT;
    }
    float* base = (g_lib.device >= 0 && g_lib.device < kScratchMaxDevices) ? g_scratchPtr[g_lib.device] : nullptr;
    if (!base) { set_error("trace_stddev_noalloc: no scratch address for device %d (pimgpu_init)", g_lib.device); return PIMGPU_E_STATE; }
    const int32_t len = (int32_t)len64;
    float* outDev = base;
    float* terms = base + kStddevScratchHead;
    stddev_sum_launch(t->color, len, nullptr, terms, base + 64, outDev, 0, g_lib.stream);
    stddev_sum_launch(t->color, len, outDev, terms, base + 64, outDev + 1, 1, g_lib.stream);
    PT_HIP(hipGetLastError());
    float* h = g_lib.smallBack;                             /* page-locked, made at pimgpu_init: the copy is not staged */
    if (!h) { set_error("trace_stddev_noalloc: no read-back block (pimgpu_init)"); return PIMGPU_E_STATE; }
    PT_HIP(hipMemcpyAsync(h, outDev, 2 * sizeof(float), hipMemcpyDeviceToHost, g_lib.stream));
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    if (meanOut) *meanOut = h[0];
    if (stddevOut) *stddevOut = h[1];
    return PIMGPU_OK;
}

int pimgpu_trace_stats(const PimTrace* t, PimTraceStats* out)
{
    if (!t || !out) { set_error("trace_stats: NULL argument"); return PIMGPU_E_INVALID; }
    *out = t->stats;
    return PIMGPU_OK;
}

/* batched Pt_TraceRay / Pt_RayGen */
static int trace_rays_common(PimScene* scene, const float* ro, const float* rd, const float* origin, int32_t n,
    uint32_t sampleIndex, uint32_t seed, int32_t maxBounces, PimResult* out, float* colors, float* directions)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || n < 0) { set_error("trace_rays: bad argument"); return PIMGPU_E_INVALID; }
    if (n == 0) return PIMGPU_OK;
    hipStream_t st = g_lib.stream;
    WorkBuffers& w = g_lib.work;
    if ((rc = w.ensure_paths(n))) return rc;
    float4 *roDev = nullptr, *rdDev = nullptr, *dirDev = nullptr;
    auto cleanup = [&]() { hipFree(roDev); hipFree(rdDev); hipFree(dirDev); };
    v4 org = f4_s(0.0f);
    if (ro)
    {
        if (hipMalloc(&roDev, sizeof(float4) * (size_t)n) != hipSuccess || hipMalloc(&rdDev, sizeof(float4) * (size_t)n) != hipSuccess)
        { cleanup(); set_error("trace_rays: out of device memory"); return PIMGPU_E_NOMEM; }
        hipMemcpyAsync(roDev, ro, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(rdDev, rd, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, st);
    }
    else
    {
        if (hipMalloc(&dirDev, sizeof(float4) * (size_t)n) != hipSuccess) { set_error("raygen: out of device memory"); return PIMGPU_E_NOMEM; }
        org = f4_v(origin[0], origin[1], origin[2], origin[3]);
        /* Pt_RayGen starts with PtScene_Update (pt.c:2625), which reaches UpdateDists (:1034); Pt_TraceRay alone does not */
        if ((rc = scene_update_dists_all(scene))) { cleanup(); return rc; }
    }
    k_init_rays<<<(n + 255) / 256, 256, 0, st>>>(w.ps, w.extO[0], w.extD[0], w.queue[0], roDev, rdDev, org, dirDev, n, sampleIndex, seed, 0);
    rc = integrate_paths(scene, n, maxBounces, nullptr, nullptr);
    if (!rc)
    {
        if (out) hipMemcpyAsync(out, w.results, sizeof(PimResult) * (size_t)n, hipMemcpyDeviceToHost, st);
        else
        {
            std::vector<PimResult> res((size_t)n);
            hipMemcpyAsync(res.data(), w.results, sizeof(PimResult) * (size_t)n, hipMemcpyDeviceToHost, st);
            hipMemcpyAsync(directions, dirDev, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, st);
            hipStreamSynchronize(st);
            for (int32_t i = 0; i < n; ++i)
            {
                colors[i * 4 + 0] = res[(size_t)i].color[0]; colors[i * 4 + 1] = res[(size_t)i].color[1];
                colors[i * 4 + 2] = res[(size_t)i].color[2]; colors[i * 4 + 3] = 0.0f;
            }
        }
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("trace_rays: device error"); rc = PIMGPU_E_DEVICE; }
    }
    cleanup();
    return rc;
}

int pimgpu_trace_rays(PimScene* scene, const float* ro, const float* rd, int32_t n, uint32_t sampleIndex, uint32_t seed,
    int32_t maxBounces, PimResult* out)
{
    if (n > 0 && (!ro || !rd || !out)) { set_error("trace_rays: NULL argument"); return PIMGPU_E_INVALID; }
    return trace_rays_common(scene, ro, rd, nullptr, n, sampleIndex, seed, maxBounces, out, nullptr, nullptr);
}

int pimgpu_raygen(PimScene* scene, const float* origin, int32_t count, uint32_t sampleIndex, uint32_t seed, int32_t maxBounces,
    float* colors, float* directions)
{
    if (count > 0 && (!origin || !colors || !directions)) { set_error("raygen: NULL argument"); return PIMGPU_E_INVALID; }
    return trace_rays_common(scene, nullptr, nullptr, origin, count, sampleIndex, seed, maxBounces, nullptr, colors, directions);
}

int pimgpu_cubemap_bake(PimScene* scene, float* const faces[6], int32_t size, const float* origin, float weight,
    uint32_t sampleIndex, uint32_t seed, int32_t maxBounces)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || !faces || !origin || size < 0 || size > 8192 || !(weight > 0.0f)) { set_error("cubemap_bake: bad argument"); return PIMGPU_E_INVALID; }
    if (size == 0) return PIMGPU_OK;                          /* cubemap.c:172 */
    for (int f = 0; f < 6; ++f) if (!faces[f]) { set_error("cubemap_bake: NULL face"); return PIMGPU_E_INVALID; }
    if (size < 2) { set_error("cubemap_bake: size 1 divides by zero in LinearInvert (sampler.h:194)"); return PIMGPU_E_INVALID; }
    const int32_t flen = size * size, n = flen * 6;
    hipStream_t st = g_lib.stream;
    WorkBuffers& w = g_lib.work;
    if ((rc = w.ensure_paths(n))) return rc;
    float* facesDev = nullptr;
    if (hipMalloc(&facesDev, sizeof(float) * 3 * (size_t)n) != hipSuccess) { set_error("cubemap_bake: out of device memory"); return PIMGPU_E_NOMEM; }
    for (int f = 0; f < 6; ++f) hipMemcpyAsync(facesDev + (size_t)f * flen * 3, faces[f], sizeof(float) * 3 * (size_t)flen, hipMemcpyHostToDevice, st);
    k_init_rays<<<(n + 255) / 256, 256, 0, st>>>(w.ps, w.extO[0], w.extD[0], w.queue[0], nullptr, nullptr,
        f4_v(origin[0], origin[1], origin[2], origin[3]), nullptr, n, sampleIndex, seed, size);
    rc = integrate_paths(scene, n, maxBounces, nullptr, nullptr);
    if (!rc)
    {
        k_cubemap_blend<<<(n + 255) / 256, 256, 0, st>>>(w.results, facesDev, n, weight);
        for (int f = 0; f < 6; ++f) hipMemcpyAsync(faces[f], facesDev + (size_t)f * flen * 3, sizeof(float) * 3 * (size_t)flen, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { set_error("cubemap_bake: device error"); rc = PIMGPU_E_DEVICE; }
    }
    hipFree(facesDev);
    return rc;
}

int pimgpu_ambcube_bake(PimScene* scene, float* cube, const float* origin, int32_t samples, int32_t prevSampleCount,
    uint32_t sampleIndex, uint32_t seed, int32_t maxBounces, int32_t* newSampleCount)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || !cube || !origin || samples < 0 || prevSampleCount < 0) { set_error("ambcube_bake: bad argument"); return PIMGPU_E_INVALID; }
    hipStream_t st = g_lib.stream;
    WorkBuffers& w = g_lib.work;
    if (samples > 0)
    {
        if ((rc = w.ensure_paths(samples))) return rc;
        float4* dirDev = nullptr; float4* cubeDev = nullptr;
        if (hipMalloc(&dirDev, sizeof(float4) * (size_t)samples) != hipSuccess || hipMalloc(&cubeDev, sizeof(float4) * 6) != hipSuccess)
        { hipFree(dirDev); set_error("ambcube_bake: out of device memory"); return PIMGPU_E_NOMEM; }
        hipMemcpyAsync(cubeDev, cube, sizeof(float4) * 6, hipMemcpyHostToDevice, st);
        if ((rc = scene_update_dists_all(scene))) { hipFree(dirDev); hipFree(cubeDev); return rc; }     /* AmbCube_Bake -> Pt_RayGen -> PtScene_Update, pt.c:2625 */
        k_init_rays<<<(samples + 255) / 256, 256, 0, st>>>(w.ps, w.extO[0], w.extD[0], w.queue[0], nullptr, nullptr,
            f4_v(origin[0], origin[1], origin[2], origin[3]), dirDev, samples, sampleIndex, seed, 0);
        rc = integrate_paths(scene, samples, maxBounces, nullptr, nullptr);
        if (!rc)
        {
            float wgt = 6.0f / (1.0f + samples);                 /* ambcube.c:24-25 */
            wgt = wgt / (1.0f + prevSampleCount);
            k_ambcube_fit<<<1, 64, 0, st>>>(w.results, dirDev, samples, wgt, cubeDev);
            hipMemcpyAsync(cube, cubeDev, sizeof(float4) * 6, hipMemcpyDeviceToHost, st);
            if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { set_error("ambcube_bake: device error"); rc = PIMGPU_E_DEVICE; }
        }
        hipFree(dirDev); hipFree(cubeDev);
        if (rc) return rc;
    }
    if (newSampleCount) *newSampleCount = prevSampleCount + 1;
    return PIMGPU_OK;
}

} /* extern "C" */
