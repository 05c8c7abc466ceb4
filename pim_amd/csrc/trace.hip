/*
 * trace.hip -- batched ray queries: the wavefront form of rtcIntersect1 / rtcOccluded
 * (rendering/path_tracer.c:470-553) and of pt_intersect_local (:1421-1464).
 *
 * One ray per lane, persistent grid-stride blocks (the ray count may live in device memory: the
 * wavefront integrator's queues never round-trip to the host).  Rays are SoA float4 pairs
 * (origin.xyz, tNear) / (direction.xyz, tFar): two coalesced 16-B loads per lane; hits are one
 * 16-B store {primID, t, u, v}.
 */
#include "pt_host.h"
#include "traverse.h"

namespace pt {

static constexpr int kTraceThreads = 256;
static constexpr int kTraceBlocksPerCU = 8;

int TraceResources::ensure()
{
    if (spill) return PIMGPU_OK;
    int dev = 0; hipDeviceProp_t prop;
    PT_HIP(hipGetDevice(&dev));
    PT_HIP(hipGetDeviceProperties(&prop, dev));
    gridBlocks = prop.multiProcessorCount * kTraceBlocksPerCU;
    const size_t lanes = (size_t)gridBlocks * kTraceThreads;
    PT_HIP(hipMalloc(&spill, sizeof(uint2) * lanes * PT_STACK_SPILL));
    PT_HIP(hipMalloc(&status, 256));
    PT_HIP(hipMemset(status, 0, 256));
    work = (unsigned long long*)(status + 32);       /* bytes 128..: nodes, tris, rays of the counting build */
    return PIMGPU_OK;
}

void TraceResources::release()
{
    hipFree(spill); hipFree(status);
    spill = nullptr; status = nullptr; gridBlocks = 0;
}

template <bool ANY, bool COUNT>
__global__ __launch_bounds__(kTraceThreads) void k_trace(DBvh bvh, const float4* __restrict__ rayO, const float4* __restrict__ rayD,
    DHit* __restrict__ hits, int32_t* __restrict__ occluded, int32_t n, const int32_t* __restrict__ countDev,
    uint2* __restrict__ spill, uint32_t* __restrict__ status, unsigned long long* __restrict__ work)
{
    __shared__ uint2 sStack[PT_STACK_LDS * kTraceThreads];
    if (countDev) n = min(n, *countDev);
    LdsStack st;
    st.lds = sStack + threadIdx.x; st.ldsStride = kTraceThreads;
    st.spillStride = (int64_t)gridDim.x * kTraceThreads;
    st.spill = spill + ((int64_t)blockIdx.x * kTraceThreads + threadIdx.x);
    bool overflow = false;
    TraverseCounts tc; tc.nodes = 0; tc.tris = 0;
    uint32_t rays = 0, worst = 0;
    for (int32_t i = blockIdx.x * kTraceThreads + threadIdx.x; i < n; i += gridDim.x * kTraceThreads)
    {
        const float4 o = rayO[i], d = rayD[i];
        const uint32_t nodesBefore = tc.nodes;
        const DHit h = traverse<ANY, LdsStack, COUNT>(bvh, o.x, o.y, o.z, d.x, d.y, d.z, o.w, d.w, st, overflow, &tc);
        if (COUNT) worst = max(worst, tc.nodes - nodesBefore);
        if (ANY) occluded[i] = h.prim >= 0 ? 1 : 0;
        else hits[i] = h;
        if (COUNT && d.w > o.w) ++rays;
    }
    if (overflow) atomicOr(status, 1u);
    if (COUNT)
    {
        /* measurement build only: per-lane sums -> three device counters */
        atomicAdd(&work[0], (unsigned long long)tc.nodes);
        atomicAdd(&work[1], (unsigned long long)tc.tris);
        atomicAdd(&work[2], (unsigned long long)rays);
        atomicMax(&work[3], (unsigned long long)worst);           /* most nodes any single ray fetched */
    }
}

int trace_closest(const DBvh& bvh, TraceResources* res, const float4* rayO, const float4* rayD, DHit* hits,
    int32_t n, const int32_t* countDev, hipStream_t stream)
{
    if (n <= 0) return PIMGPU_OK;
    int rc = res->ensure(); if (rc) return rc;
    const int blocks = (int)std::min<int64_t>(res->gridBlocks, ((int64_t)n + kTraceThreads - 1) / kTraceThreads);
    if (res->counting) k_trace<false, true><<<blocks, kTraceThreads, 0, stream>>>(bvh, rayO, rayD, hits, nullptr, n, countDev, res->spill, res->status, res->work);
    else k_trace<false, false><<<blocks, kTraceThreads, 0, stream>>>(bvh, rayO, rayD, hits, nullptr, n, countDev, res->spill, res->status, res->work);
    PT_HIP(hipGetLastError());
    return PIMGPU_OK;
}

int trace_any(const DBvh& bvh, TraceResources* res, const float4* rayO, const float4* rayD, int32_t* occluded,
    int32_t n, hipStream_t stream)
{
    if (n <= 0) return PIMGPU_OK;
    int rc = res->ensure(); if (rc) return rc;
    const int blocks = (int)std::min<int64_t>(res->gridBlocks, ((int64_t)n + kTraceThreads - 1) / kTraceThreads);
    k_trace<true, false><<<blocks, kTraceThreads, 0, stream>>>(bvh, rayO, rayD, nullptr, occluded, n, nullptr, res->spill, res->status, res->work);
    PT_HIP(hipGetLastError());
    return PIMGPU_OK;
}

/* PtRayHit from the raw record: pt_intersect_local, path_tracer.c:1434-1463 */
__global__ __launch_bounds__(256) void k_make_rayhits(DScene s, const float4* __restrict__ rayD, const DHit* __restrict__ hits,
    PimRayHit* __restrict__ out, int32_t n)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DHit h = hits[i];
    PimRayHit hit;
    hit.wuvt[0] = hit.wuvt[1] = hit.wuvt[2] = 0.0f; hit.wuvt[3] = -1.0f;
    hit.normal[0] = hit.normal[1] = hit.normal[2] = hit.normal[3] = 0.0f;
    hit.type = PIM_HIT_NOTHING; hit.iVert = -1; hit.flags = 0; hit._pad = 0;
    if (h.prim >= 0 && h.t > 0.0f)
    {
        const float4 rd4 = rayD[i];
        const v4 rd = f4_v(rd4.x, rd4.y, rd4.z, 0.0f);
        const v4* p = s.positions + (size_t)h.prim * 3;
        v4 e1 = f4_sub(p[1], p[0]), e2 = f4_sub(p[2], p[0]);
        e1.w = e2.w = 0.0f;
        v4 Ng = f4_cross3(e1, e2);
        hit.type = PIM_HIT_TRIANGLE;
        if (f4_dot3(Ng, rd) > 0.0f) { Ng = f4_neg(Ng); hit.type = PIM_HIT_BACKFACE; }
        Ng = f4_normalize3(Ng);
        hit.normal[0] = Ng.x; hit.normal[1] = Ng.y; hit.normal[2] = Ng.z;
        const float u = f1_sat(h.u), v = f1_sat(h.v), w = f1_sat(1.0f - (u + v));
        hit.iVert = h.prim * 3;
        hit.wuvt[0] = w; hit.wuvt[1] = u; hit.wuvt[2] = v; hit.wuvt[3] = h.t;
        hit.flags = s.materials[s.matIds[hit.iVert]].flags;
    }
    out[i] = hit;
}

int make_rayhits(const DScene& scene, const float4* rayD, const DHit* hits, PimRayHit* out, int32_t n, hipStream_t stream)
{
    if (n <= 0) return PIMGPU_OK;
    k_make_rayhits<<<(n + 255) / 256, 256, 0, stream>>>(scene, rayD, hits, out, n);
    PT_HIP(hipGetLastError());
    return PIMGPU_OK;
}

} /* namespace pt */
