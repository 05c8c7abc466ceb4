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
#include <hip/hip_ext.h>
#include <cstdlib>
#include <algorithm>

namespace pt {

static constexpr int kTraceThreads = 256;
static constexpr int kTraceBlocksPerCU = 8;

/*
 * The kernel is a per-lane state machine, not a per-ray function call.  Every trip of the wave's loop a lane
 *   A  tests ONE node (4 boxes, the hit children sorted by entry distance, the far ones pushed),
 *   S  puts a leaf it has reached aside in `leaf` and pops the next stack entry, so that it has a node again
 *      for the next trip (a second leaf reached while the first is still waiting stalls the lane),
 *   B  tests ONE triangle of the leaf it has put aside -- but only on trips where PT_TRI_VOTE lanes of the wave
 *      have one waiting, or no lane has a node left.
 * Nodes are tested about four times as often as triangles; running the triangle code on every trip for the few
 * lanes that are at a leaf costs the whole wave its instructions, so leaves are postponed until enough of them
 * have gathered (Aila & Laine's speculative traversal).  Lanes whose ray has finished are REFILLED from the wave's
 * own contiguous slice of the ray array once PT_REFILL_IDLE of them are idle, instead of idling until the slowest
 * ray of the wave is done.
 * `cur`: >= 0 node index; < 0 leaf code ~((first << 3) | (count - 1)); PT_BVH_EMPTY = nothing.  `leaf`: the leaf
 * put aside, stepped by code + 7 per triangle (first + 1, count - 1), PT_BVH_EMPTY = none.
 * Which lane traces which ray, and in which order a ray meets its nodes and triangles, does not influence the
 * ray's result (traverse.h contract).
 */
#ifndef PT_REFILL_IDLE
#define PT_REFILL_IDLE 16
#endif
#ifndef PT_POP_MAX
#define PT_POP_MAX 1
#endif
#ifndef PT_TRI_VOTE
#define PT_TRI_VOTE 10
#endif

#ifndef PT_LEAF2
#define PT_LEAF2 0            /* 1: a lane may put TWO leaves aside before it stalls (the vote of B can then wait for more lanes) */
#endif
#ifndef PT_LEAF_DIRECT
#define PT_LEAF_DIRECT 0      /* 1: a nearest child that is a leaf goes aside straight from A and the next nearest child becomes `cur` */
#endif

#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 1      /* waves per SIMD the kernel is compiled for at least (1: whatever its registers give, 6 at 73 VGPRs) */
#endif
/*
 * TGT: rays may name a TARGET triangle (targets[i] >= 0; the light-sampling rays of the integrator, whose question is "is the
 * closest hit this triangle?", path_tracer.c:1813-1815).  The caller has tested the target itself with the same triangle test
 * and left its record in hits[i] and its distance in the ray's tFar, so what remains is a visibility query: the walk ends at the
 * first triangle the closest-hit rule would prefer -- nearer, or equally near with a lower id -- and the record is overwritten
 * with a miss; if there is none the record stays as it is.  Bit for bit the answer of the full closest-hit walk followed by the
 * comparison, without walking on once the outcome is known.
 */
template <bool ANY, bool COUNT, bool TGT = false>
__global__ __launch_bounds__(kTraceThreads, PT_TRACE_WAVES) void k_trace(DBvh bvh, const float4* __restrict__ rayO, const float4* __restrict__ rayD,
    DHit* __restrict__ hits, int32_t* __restrict__ occluded, int32_t n, const int32_t* __restrict__ countDev,
    unsigned long long* __restrict__ spill, uint32_t* __restrict__ status, unsigned long long* __restrict__ work,
    uint32_t* __restrict__ chunkCounter, uint32_t* __restrict__ chunkCounterNext, int32_t dynChunk, int32_t stackCap, const int32_t* __restrict__ targets)
{
    __shared__ unsigned long long sStack[PT_STACK_LDS * kTraceThreads];
    if (countDev) n = min(n, *countDev);
    LdsStack st;
    st.lds = (lds_u64*)(sStack + threadIdx.x); st.ldsStride = kTraceThreads;
    st.spillStride = (int64_t)gridDim.x * kTraceThreads;
    st.spill = (glb_u64*)(spill + ((int64_t)blockIdx.x * kTraceThreads + threadIdx.x));
    st.init(); st.cap = stackCap;
    bool overflow = false;
    uint32_t cNodes = 0, cTris = 0, cRays = 0, cWorst = 0, rayNodes = 0;
    uint32_t cTrips = 0, cExecA = 0, cExecB = 0, cPopIter = 0, cPopLanes = 0;     /* counting build: wave-level executions of each section */
    uint32_t cIdle = 0, cStall = 0, cNoNode = 0;                                  /* counting build: trips this lane spent without a ray / with a second leaf it cannot put aside / with nothing in hand */

    /* this wave's slice of the rays: contiguous, a multiple of 64 long */
    const int32_t wavesTotal = (int32_t)gridDim.x * (kTraceThreads / 64);
    const int32_t wave = (int32_t)blockIdx.x * (kTraceThreads / 64) + (int32_t)(threadIdx.x >> 6);
    /* dynChunk == 0: one static slice per wave (contiguous, a multiple of 64 long).  dynChunk > 0 (the host uses 128): rays in chunks -- the
     * first chunk of a wave is its own, further ones come from a counter (one returning atomic per chunk), so that waves
     * whose rays were short take over rays of the others.  That pays when rays are long (a tree served from the Infinity
     * Cache: 1 M-triangle scene 7.10 -> 6.90 ms per sample) and costs when they are short and the counter becomes the
     * bottleneck (Cornell 2.23 -> 2.88 ms): the host decides (trace_chunk).  The launch zeroes the counter of the NEXT
     * launch (the previous launch, which used that one, has finished: same stream). */
    if (blockIdx.x == 0 && threadIdx.x == 0) *chunkCounterNext = 0u;
    const int64_t per = dynChunk > 0 ? (int64_t)dynChunk : ((((int64_t)n + wavesTotal - 1) / wavesTotal) + 63) & ~(int64_t)63;
    int32_t cursor = (int32_t)min((int64_t)n, per * wave);
    int32_t sliceEnd = (int32_t)min((int64_t)n, per * (wave + 1));
    bool moreGlobal = dynChunk > 0 && per * wavesTotal < (int64_t)n;

    const float4* __restrict__ nodes = bvh.nodes;
    const uint4* __restrict__ cnodes = bvh.cnodes;
    const float4* __restrict__ tris = bvh.tris;

    bool active = false;
    int32_t ri = -1, cur = PT_BVH_EMPTY, leaf = PT_BVH_EMPTY;      /* ri >= 0 on an inactive lane: a finished ray whose record is not stored yet */
    int32_t leaf2 = PT_BVH_EMPTY;                                   /* PT_LEAF2: the second leaf put aside */
    float hDet = 1.0f;                                              /* h.u, h.v hold the numerators V, W until the record is stored */
    float tNear = 0.0f;
    RaySlabQ sl = {};
    RayPre r = {};
    DHit h = {};
    int32_t tgt = -1;

    for (;;)
    {
        /* refill idle lanes with the next rays of the slice */
        {
            if (cursor >= sliceEnd && moreGlobal)
            {
                uint32_t got = 0;
                if ((threadIdx.x & 63) == 0) got = atomicAdd(chunkCounter, (uint32_t)dynChunk);
                const int64_t base = (int64_t)__builtin_amdgcn_readfirstlane(got) + per * wavesTotal;
                if (base >= (int64_t)n) moreGlobal = false;
                else { cursor = (int32_t)base; sliceEnd = (int32_t)min((int64_t)n, base + per); }
            }
            const unsigned long long idle = __ballot(!active);
            if (cursor < sliceEnd && idle != 0ull)
            {
                const int32_t rank = (int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const int32_t mine = cursor + rank;
                if (!active && mine < sliceEnd)
                {
                    const float4 o = rayO[mine], d = rayD[mine];
                    ri = mine;
                    h.prim = -1; h.t = d.w; h.u = 0.0f; h.v = 0.0f;
                    if (TGT) tgt = targets[mine];
                    if (d.w > o.w)
                    {
                        r = ray_pre(o.x, o.y, o.z, d.x, d.y, d.z);
                        sl = ray_slab_q(o.x, o.y, o.z, d.x, d.y, d.z, nodes[7].x);
                        tNear = o.w;
                        cur = 0; leaf = PT_BVH_EMPTY; leaf2 = PT_BVH_EMPTY; active = true;
                        if (COUNT) { ++cRays; rayNodes = 0; }
                    }
                    else
                    {
                        if (ANY) occluded[ri] = 0; else hits[ri] = h;      /* no ray: a miss (also for a target the caller's own test missed) */
                        ri = -1;
                    }
                }
                cursor = min(sliceEnd, cursor + (int32_t)__popcll(idle));
            }
            if (__ballot(active) == 0ull)
            {
                if (cursor >= sliceEnd && !moreGlobal) break;
                continue;
            }
        }
        for (;;)
        {
            if (COUNT)
            {
                ++cTrips; if (__ballot(active && cur >= 0) != 0ull) ++cExecA;
                if (!active) ++cIdle;
                else if (cur == PT_BVH_EMPTY) ++cNoNode;
                else if (cur < 0 && leaf != PT_BVH_EMPTY) ++cStall;
            }
            /* A: one node (compressed, 4 x 16 bytes) */
            if (active && cur >= 0)
            {
                if (COUNT) { ++cNodes; ++rayNodes; }
                const uint4* q = cnodes + (size_t)cur * 4;
                const uint4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
                float d[4]; int32_t c[4];
                c[0] = (int32_t)w3.x; c[1] = (int32_t)w3.y; c[2] = (int32_t)w3.z; c[3] = (int32_t)w3.w;
                /* plane parameter = q * (step / d) + (origin plane parameter); near / far planes picked per ray octant,
                 * one select per packed dword of 4 children */
                const float ax = __uint_as_float(w0.w) * sl.ix, ay = __uint_as_float(w2.z) * sl.iy, az = __uint_as_float(w2.w) * sl.iz;
                const float bnx = __builtin_fmaf(__uint_as_float(w0.x), sl.ix, sl.nx), bfx = __builtin_fmaf(__uint_as_float(w0.x), sl.ix, sl.fx);
                const float bny = __builtin_fmaf(__uint_as_float(w0.y), sl.iy, sl.ny), bfy = __builtin_fmaf(__uint_as_float(w0.y), sl.iy, sl.fy);
                const float bnz = __builtin_fmaf(__uint_as_float(w0.z), sl.iz, sl.nz), bfz = __builtin_fmaf(__uint_as_float(w0.z), sl.iz, sl.fz);
                const bool px = sl.ix >= 0.0f, py = sl.iy >= 0.0f, pz = sl.iz >= 0.0f;
                const uint32_t qnx = px ? w1.x : w1.w, qfx = px ? w1.w : w1.x;
                const uint32_t qny = py ? w1.y : w2.x, qfy = py ? w2.x : w1.y;
                const uint32_t qnz = pz ? w1.z : w2.y, qfz = pz ? w2.y : w1.z;
                /* Empty slots are inverted on the grid (near 255, far 0) and reject themselves; should one slip through in a
                 * node flat on every axis, its PT_BVH_EMPTY is read as "nothing" when it is popped */
#pragma unroll
                for (int k = 0; k < 4; ++k)
                {
                    const float tnx = __builtin_fmaf((float)((qnx >> (8 * k)) & 255u), ax, bnx), tfx = __builtin_fmaf((float)((qfx >> (8 * k)) & 255u), ax, bfx);
                    const float tny = __builtin_fmaf((float)((qny >> (8 * k)) & 255u), ay, bny), tfy = __builtin_fmaf((float)((qfy >> (8 * k)) & 255u), ay, bfy);
                    const float tnz = __builtin_fmaf((float)((qnz >> (8 * k)) & 255u), az, bnz), tfz = __builtin_fmaf((float)((qfz >> (8 * k)) & 255u), az, bfz);
                    const float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tNear));
                    const float t1 = fminf(fminf(tfx, tfy), fminf(tfz, h.t));
                    d[k] = (t0 <= t1) ? t0 : __builtin_inff();
                }
                /* sort the 4 candidates by entry distance (5-comparator network); misses carry +inf */
                PT_SWAP_IF(d[0], c[0], d[1], c[1]);
                PT_SWAP_IF(d[2], c[2], d[3], c[3]);
                PT_SWAP_IF(d[0], c[0], d[2], c[2]);
                PT_SWAP_IF(d[1], c[1], d[3], c[3]);
                PT_SWAP_IF(d[1], c[1], d[2], c[2]);
#if PT_LEAF_DIRECT
                /* the nearest child is a leaf and there is room aside: it goes there now (it would be put aside by S anyway, and the
                 * next nearest child popped right back), the other candidates move up one place */
                {
                    const bool direct = d[0] < __builtin_inff() && c[0] < 0 && (leaf == PT_BVH_EMPTY || (PT_LEAF2 && leaf2 == PT_BVH_EMPTY));
                    if (direct)
                    {
                        if (leaf == PT_BVH_EMPTY) leaf = c[0]; else leaf2 = c[0];
                        d[0] = d[1]; c[0] = c[1]; d[1] = d[2]; c[1] = c[2]; d[2] = d[3]; c[2] = c[3]; d[3] = __builtin_inff();
                    }
                }
#endif
                overflow |= !st.push3(d[1], c[1], d[2], c[2], d[3], c[3]);
                cur = (d[0] < __builtin_inff()) ? c[0] : PT_BVH_EMPTY;
            }
            /* S: put a reached leaf aside, then pop (culling entries that start beyond the current closest hit) */
            if (active && cur < 0)
            {
                if (cur != PT_BVH_EMPTY && leaf == PT_BVH_EMPTY) { leaf = cur; cur = PT_BVH_EMPTY; }
#if PT_LEAF2
                else if (cur != PT_BVH_EMPTY && leaf2 == PT_BVH_EMPTY) { leaf2 = cur; cur = PT_BVH_EMPTY; }
#endif
                if (cur == PT_BVH_EMPTY)
                {
                    /* at most PT_POP_MAX entries per trip: an entry that starts beyond the closest hit is dropped and the
                     * lane tries again on the next trip (a loop until one survives keeps the whole wave waiting on LDS) */
                    float t; int32_t c;
#pragma unroll
                    for (int k = 0; k < PT_POP_MAX; ++k)
                    {
                        if (cur == PT_BVH_EMPTY && st.pop(&t, &c))
                        {
                            if (COUNT) { ++cPopLanes; if ((int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(__ballot(true) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__ballot(true), 0u)) == 0) ++cPopIter; }
                            if (t <= h.t) cur = c;
                        }
                    }
                }
            }
            /* B: one triangle of the leaf put aside, when enough lanes have one or none has a node */
            {
                const unsigned long long leafMask = __ballot(active && leaf != PT_BVH_EMPTY);
                const unsigned long long nodeMask = __ballot(active && cur >= 0);
                if (leafMask != 0ull && ((int32_t)__popcll(leafMask) >= PT_TRI_VOTE || nodeMask == 0ull))
                {
                    if (COUNT) ++cExecB;
                    if (active && leaf != PT_BVH_EMPTY)
                    {
                        if (COUNT) ++cTris;
                        const uint32_t code = (uint32_t)~leaf;
                        const float4* tp = tris + (size_t)(code >> 3) * 3;
                        const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                        float t, vn, wn, det;
                        if (tri_test_raw(r, p0, p1, p2, tNear, h.t, &t, &vn, &wn, &det))
                        {
                            const int32_t prim = __float_as_int(p0.w);
                            if (TGT && tgt >= 0)
                            {
                                /* t <= h.t = the target's own distance: anything but the target itself, or an equally near triangle with a higher id, is in front */
                                if (prim != tgt && (t < h.t || prim < tgt)) { h.prim = -2; cur = PT_BVH_EMPTY; leaf = PT_BVH_EMPTY; leaf2 = PT_BVH_EMPTY; st.init(); }
                            }
                            else if (t < h.t || (t == h.t && (h.prim < 0 || prim < h.prim))) { h.prim = prim; h.t = t; h.u = vn; h.v = wn; hDet = det; }
                        }
                        if (!TGT || leaf != PT_BVH_EMPTY) leaf = (code & 7u) ? (int32_t)~(code + 7u) : PT_BVH_EMPTY;
#if PT_LEAF2
                        if (leaf == PT_BVH_EMPTY) { leaf = leaf2; leaf2 = PT_BVH_EMPTY; }
#endif
                    }
                }
            }
            /* done: nothing in hand, nothing aside, nothing on the stack (any-hit: the first accepted triangle) */
            if (active && ((cur == PT_BVH_EMPTY && leaf == PT_BVH_EMPTY && st.sp == 0) || (ANY && h.prim >= 0)))
            {
                if (COUNT) cWorst = max(cWorst, rayNodes);
                st.init();
                active = false;             /* the record is stored at the next refill */
            }
            const int32_t idleCount = (int32_t)__popcll(__ballot(!active));
            if (idleCount == 64 || (idleCount >= PT_REFILL_IDLE && (cursor < sliceEnd || moreGlobal))) break;
        }
        /* store the records of the rays that finished during this round of trips: the two barycentric divisions and the store
         * run here, once per batch of finished lanes, not on every trip on which some lane finishes */
        if (!active && ri >= 0)
        {
            if (ANY) occluded[ri] = h.prim >= 0 ? 1 : 0;
            else if (TGT && tgt >= 0)
            {
                if (h.prim == -2) { h.prim = -1; h.u = 0.0f; h.v = 0.0f; hits[ri] = h; }      /* something in front: a miss; else the caller's record stands */
            }
            else
            {
                if (h.prim >= 0) { h.u = h.u / hDet; h.v = h.v / hDet; }
                hits[ri] = h;
            }
            ri = -1;
        }
    }
    if (overflow) atomicOr(status, 1u);
    if (COUNT)
    {
        /* measurement build only: per-lane sums -> device counters */
        atomicAdd(&work[0], (unsigned long long)cNodes);
        atomicAdd(&work[1], (unsigned long long)cTris);
        atomicAdd(&work[2], (unsigned long long)cRays);
        atomicMax(&work[3], (unsigned long long)cWorst);           /* most nodes any single ray fetched */
        if ((threadIdx.x & 63) == 0)
        {
            atomicAdd(&work[4], (unsigned long long)cTrips); atomicAdd(&work[5], (unsigned long long)cExecA);
            atomicAdd(&work[6], (unsigned long long)cExecB);
        }
        atomicAdd(&work[7], (unsigned long long)cPopIter); atomicAdd(&work[8], (unsigned long long)cPopLanes);
        atomicAdd(&work[9], (unsigned long long)cIdle); atomicAdd(&work[10], (unsigned long long)cStall); atomicAdd(&work[11], (unsigned long long)cNoNode);
    }
}

int TraceResources::ensure()
{
    if (spill) return PIMGPU_OK;
    int dev = 0; hipDeviceProp_t prop;
    PT_HIP(hipGetDevice(&dev));
    PT_HIP(hipGetDeviceProperties(&prop, dev));
    /* persistent grid = exactly the blocks that are resident at once (each wave owns a fixed slice of the rays) */
    int perCU = kTraceBlocksPerCU;
    PT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_trace<false, false>, kTraceThreads, 0));
    perCU = std::max(1, std::min(perCU, kTraceBlocksPerCU));
    if (const char* e = getenv("PIMGPU_TRACE_BLOCKS_PER_CU")) perCU = std::max(1, std::min(16, atoi(e)));      /* tuning knob */
    gridBlocks = prop.multiProcessorCount * perCU;
    const size_t lanes = (size_t)gridBlocks * kTraceThreads;
    PT_HIP(hipMalloc(&spill, sizeof(unsigned long long) * lanes * PT_STACK_SPILL));
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

int32_t trace_stack_cap()
{
    const char* e = getenv("PIMGPU_STACK_CAP");
    if (!e || !*e) return PT_STACK_MAX;
    return std::max(PT_STACK_LDS, std::min(PT_STACK_MAX, atoi(e)));
}

/* rays per dynamic chunk, 0 = static slices (k_trace's header): by the size of the tree, like the tail kernel's build */
static int32_t trace_chunk(const DBvh& bvh)
{
    static const int forced = [] { const char* e = getenv("PIMGPU_TRACE_CHUNK"); return (e && *e) ? atoi(e) : -1; }();
    if (forced >= 0) return forced & ~63;
    return (64ll * bvh.nodeCount + 48ll * bvh.triCount > (8ll << 20)) ? 128 : 0;
}

/* t0 / t1 (optional): events attached to the dispatch itself (hipExtLaunchKernelGGL): they time the kernel from its own
 * completion signal.  Separate hipEventRecord calls around the launch put two barrier packets into the queue, ~6 us of
 * idle GPU each -- 12 per Cornell sample. */
int trace_closest(const DBvh& bvh, TraceResources* res, const float4* rayO, const float4* rayD, DHit* hits,
    int32_t n, const int32_t* countDev, hipStream_t stream, hipEvent_t t0, hipEvent_t t1, const int32_t* targets)
{
    if (n <= 0) return PIMGPU_OK;
    int rc = res->ensure(); if (rc) return rc;
    const int blocks = (int)std::min<int64_t>(res->gridBlocks, ((int64_t)n + kTraceThreads - 1) / kTraceThreads);
    int32_t* noOccluded = nullptr;
    uint32_t* cc = res->status + 24 + res->chunkParity; uint32_t* cn = res->status + 24 + (res->chunkParity ^ 1); res->chunkParity ^= 1;
    const int32_t dc = trace_chunk(bvh), cap = trace_stack_cap();
    if (res->counting)
    {
        if (t0) hipEventRecord(t0, stream);
        if (targets) k_trace<false, true, true><<<blocks, kTraceThreads, 0, stream>>>(bvh, rayO, rayD, hits, nullptr, n, countDev, res->spill, res->status, res->work, cc, cn, dc, cap, targets);
        else k_trace<false, true><<<blocks, kTraceThreads, 0, stream>>>(bvh, rayO, rayD, hits, nullptr, n, countDev, res->spill, res->status, res->work, cc, cn, dc, cap, targets);
        if (t1) hipEventRecord(t1, stream);
    }
    else if (t0 && t1)
    {
        if (targets) hipExtLaunchKernelGGL((k_trace<false, false, true>), dim3(blocks), dim3(kTraceThreads), 0, stream, t0, t1, 0,
            bvh, rayO, rayD, hits, noOccluded, n, countDev, res->spill, res->status, res->work, cc, cn, dc, cap, targets);
        else hipExtLaunchKernelGGL((k_trace<false, false>), dim3(blocks), dim3(kTraceThreads), 0, stream, t0, t1, 0,
            bvh, rayO, rayD, hits, noOccluded, n, countDev, res->spill, res->status, res->work, cc, cn, dc, cap, targets);
    }
    else if (targets) k_trace<false, false, true><<<blocks, kTraceThreads, 0, stream>>>(bvh, rayO, rayD, hits, nullptr, n, countDev, res->spill, res->status, res->work, cc, cn, dc, cap, targets);
    else k_trace<false, false><<<blocks, kTraceThreads, 0, stream>>>(bvh, rayO, rayD, hits, nullptr, n, countDev, res->spill, res->status, res->work, cc, cn, dc, cap, targets);
    PT_HIP(hipGetLastError());
    return PIMGPU_OK;
}

int trace_any(const DBvh& bvh, TraceResources* res, const float4* rayO, const float4* rayD, int32_t* occluded,
    int32_t n, hipStream_t stream)
{
    if (n <= 0) return PIMGPU_OK;
    int rc = res->ensure(); if (rc) return rc;
    const int blocks = (int)std::min<int64_t>(res->gridBlocks, ((int64_t)n + kTraceThreads - 1) / kTraceThreads);
    uint32_t* cc = res->status + 24 + res->chunkParity; uint32_t* cn = res->status + 24 + (res->chunkParity ^ 1); res->chunkParity ^= 1;
    const int32_t dc = trace_chunk(bvh), cap = trace_stack_cap();
    k_trace<true, false><<<blocks, kTraceThreads, 0, stream>>>(bvh, rayO, rayD, nullptr, occluded, n, nullptr, res->spill, res->status, res->work, cc, cn, dc, cap, nullptr);
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
        const DTriRec* r = s.tri + h.prim;
        v4 e1 = f4_sub(r->p1, r->p0), e2 = f4_sub(r->p2, r->p0);
        e1.w = e2.w = 0.0f;
        v4 Ng = f4_cross3(e1, e2);
        hit.type = PIM_HIT_TRIANGLE;
        if (f4_dot3(Ng, rd) > 0.0f) { Ng = f4_neg(Ng); hit.type = PIM_HIT_BACKFACE; }
        Ng = f4_normalize3(Ng);
        hit.normal[0] = Ng.x; hit.normal[1] = Ng.y; hit.normal[2] = Ng.z;
        const float u = f1_sat(h.u), v = f1_sat(h.v), w = f1_sat(1.0f - (u + v));
        hit.iVert = h.prim * 3;
        hit.wuvt[0] = w; hit.wuvt[1] = u; hit.wuvt[2] = v; hit.wuvt[3] = h.t;
        hit.flags = r->flags;
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
