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
                const uint4* q = (const uint4*)((const char*)cnodes + ((uint32_t)cur << 6));       /* a 32-bit offset from a uniform base (trees below 2^26 nodes: bvh_build checks) */
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

/*
 * k_traceq: the same query with the leaves' triangles QUEUED per wave instead of put aside per lane.
 *
 * In k_trace a lane that reaches a leaf keeps it until enough lanes hold one (PT_TRI_VOTE), and the triangle code then runs for
 * those lanes only: 14-17 of 64 lanes per execution on the incoherent benchmark, a third of the kernel's instructions (the kernel
 * is bound by VALU issue, profiles/r03_*), and a lane that reaches a second leaf stalls.  Here a lane that arrives at a leaf
 * appends one entry per triangle -- (owner lane, triangle) -- to a ring in LDS and walks on at once; when the ring holds 64
 * entries (or no lane has a node left) the WHOLE wave tests 64 (ray, triangle) pairs, every lane one: the tester fetches the
 * owner's prepared ray by cross-lane reads (ds_bpermute), the ray's closest hit lives per owner in LDS as one 64-bit key
 * (t bits << 32 | primID) and an atomic minimum on it IS the contract's closest-hit rule (smaller t, then smaller id; t > 0, so
 * float bits order as integers) -- traverse_coop's scheme (traverse.h).  The lane whose key stands after the minimum leaves the
 * barycentric numerators beside it.  The owner culls nodes against the key's t.  A ray is finished when its walk is over AND the
 * ring has been served past its last entry (entries are served in order: one running ticket per wave).
 * Which lane tests which triangle, and when, changes nothing of the result (contract), so hits are those of k_trace bit for bit.
 */
#ifndef PT_QSTACK_LDS
#define PT_QSTACK_LDS 8       /* stack entries of a lane in LDS: 16 KiB + 9 KiB of hit keys and ring per block = 6 blocks per CU */
#endif
static constexpr int kQueueCap = 256;               /* ring entries per wave: < 64 left over + 64 lanes x PT_LEAF_MAX new ones per trip */
static_assert(63 + 64 * PT_LEAF_MAX < kQueueCap, "the ring must hold one trip's triangles on top of an unserved rest");
static constexpr uint32_t kNoPrim = 0x7fffffffu;
static constexpr int kQueueTriBits = 26;            /* entry = owner lane << 26 | triangle: up to 2^26 triangles (trace_queue_ok) */

#ifndef PT_TRACEQ_WAVES
#define PT_TRACEQ_WAVES 6     /* waves per SIMD the kernel is compiled for (<= 80 VGPRs); its LDS allows six blocks per CU */
#endif
template <bool ANY, bool COUNT, bool TGT = false>
__global__ __launch_bounds__(kTraceThreads, COUNT ? 1 : PT_TRACEQ_WAVES) void k_traceq(DBvh bvh, const float4* __restrict__ rayO, const float4* __restrict__ rayD,
    DHit* __restrict__ hits, int32_t* __restrict__ occluded, int32_t n, const int32_t* __restrict__ countDev,
    unsigned long long* __restrict__ spill, uint32_t* __restrict__ status, unsigned long long* __restrict__ work,
    uint32_t* __restrict__ chunkCounter, uint32_t* __restrict__ chunkCounterNext, int32_t dynChunk, int32_t stackCap, const int32_t* __restrict__ targets)
{
    __shared__ unsigned long long sStack[PT_QSTACK_LDS * kTraceThreads];
    __shared__ unsigned long long sBest[kTraceThreads];
    __shared__ float sNum[3][kTraceThreads];                     /* V, W, det of the hit in sBest */
    __shared__ uint32_t sQueue[kTraceThreads / 64][kQueueCap];
    if (countDev) n = min(n, *countDev);
    typedef LdsStackT<PT_QSTACK_LDS> Stack;
    Stack st;
    st.lds = (lds_u64*)(sStack + threadIdx.x); st.ldsStride = kTraceThreads;
    st.spillStride = (int64_t)gridDim.x * kTraceThreads;
    st.spill = (glb_u64*)(spill + ((int64_t)blockIdx.x * kTraceThreads + threadIdx.x));
    st.init(); st.cap = min(stackCap, PT_QSTACK_LDS + PT_STACK_SPILL);
    const int lane = (int)(threadIdx.x & 63), waveInBlock = (int)(threadIdx.x >> 6);
    const int laneBase = waveInBlock * 64;                       /* this wave's part of the per-thread LDS arrays */
    lds_u64* best = (lds_u64*)sBest;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* queue = (lds_u32*)sQueue[waveInBlock];
    bool overflow = false;
    uint32_t cNodes = 0, cTris = 0, cRays = 0, cWorst = 0, rayNodes = 0, cTrips = 0, cExecA = 0, cExecB = 0;

    const int32_t wavesTotal = (int32_t)gridDim.x * (kTraceThreads / 64);
    const int32_t wave = (int32_t)blockIdx.x * (kTraceThreads / 64) + waveInBlock;
    if (blockIdx.x == 0 && threadIdx.x == 0) *chunkCounterNext = 0u;
    const int64_t per = dynChunk > 0 ? (int64_t)dynChunk : ((((int64_t)n + wavesTotal - 1) / wavesTotal) + 63) & ~(int64_t)63;
    int32_t cursor = (int32_t)min((int64_t)n, per * wave);
    int32_t sliceEnd = (int32_t)min((int64_t)n, per * (wave + 1));
    bool moreGlobal = dynChunk > 0 && per * wavesTotal < (int64_t)n;

    const float4* __restrict__ nodes = bvh.nodes;
    const uint4* __restrict__ cnodes = bvh.cnodes;
    const float4* __restrict__ tris = bvh.tris;

    bool active = false;
    int32_t ri = -1, cur = PT_BVH_EMPTY;
    float tNear = 0.0f, tFar0 = 0.0f;
    RaySlabQ sl = {};
    RayPre r = {};
    int32_t kpack = 0;                                           /* r.kx | r.ky << 2 | r.kz << 4: one cross-lane read instead of three */
    int32_t tgt = -1;
    uint32_t qHead = 0, qTail = 0;                               /* wave-uniform running tickets of the ring */
    uint32_t lastTicket = 0xffffffffu;                           /* ticket of this ray's last entry; "served" = (int32)(qHead - lastTicket) > 0 */

    for (;;)
    {
        /* refill idle lanes with the next rays of the slice */
        {
            if (cursor >= sliceEnd && moreGlobal)
            {
                uint32_t got = 0;
                if (lane == 0) got = atomicAdd(chunkCounter, (uint32_t)dynChunk);
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
                    if (TGT) tgt = targets[mine];
                    if (d.w > o.w)
                    {
                        r = ray_pre(o.x, o.y, o.z, d.x, d.y, d.z);
                        kpack = r.kx | (r.ky << 2) | (r.kz << 4);
                        sl = ray_slab_q(o.x, o.y, o.z, d.x, d.y, d.z, nodes[7].x);
                        tNear = o.w; tFar0 = d.w;
                        best[laneBase + lane] = ((unsigned long long)__float_as_uint(d.w) << 32) | (unsigned long long)kNoPrim;
                        lastTicket = qHead - 1u;                 /* nothing queued: served already */
                        cur = 0; active = true;
                        if (COUNT) { ++cRays; rayNodes = 0; }
                    }
                    else
                    {
                        DHit h; h.prim = -1; h.t = d.w; h.u = 0.0f; h.v = 0.0f;
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
            if (COUNT) { ++cTrips; if (__ballot(active && cur >= 0) != 0ull) ++cExecA; }
            /* A: one node (compressed, 4 x 16 bytes), culled against the ray's closest hit so far, whoever found it */
            if (active && cur >= 0)
            {
                if (COUNT) { ++cNodes; ++rayNodes; }
                const uint4* q = (const uint4*)((const char*)cnodes + ((uint32_t)cur << 6));       /* 32-bit offset from a uniform base, as in k_trace */
                const uint4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
                const float lim = __uint_as_float((uint32_t)(best[laneBase + lane] >> 32));
                float d[4]; int32_t c[4];
                c[0] = (int32_t)w3.x; c[1] = (int32_t)w3.y; c[2] = (int32_t)w3.z; c[3] = (int32_t)w3.w;
                const float ax = __uint_as_float(w0.w) * sl.ix, ay = __uint_as_float(w2.z) * sl.iy, az = __uint_as_float(w2.w) * sl.iz;
                const float bnx = __builtin_fmaf(__uint_as_float(w0.x), sl.ix, sl.nx), bfx = __builtin_fmaf(__uint_as_float(w0.x), sl.ix, sl.fx);
                const float bny = __builtin_fmaf(__uint_as_float(w0.y), sl.iy, sl.ny), bfy = __builtin_fmaf(__uint_as_float(w0.y), sl.iy, sl.fy);
                const float bnz = __builtin_fmaf(__uint_as_float(w0.z), sl.iz, sl.nz), bfz = __builtin_fmaf(__uint_as_float(w0.z), sl.iz, sl.fz);
                const bool px = sl.ix >= 0.0f, py = sl.iy >= 0.0f, pz = sl.iz >= 0.0f;
                const uint32_t qnx = px ? w1.x : w1.w, qfx = px ? w1.w : w1.x;
                const uint32_t qny = py ? w1.y : w2.x, qfy = py ? w2.x : w1.y;
                const uint32_t qnz = pz ? w1.z : w2.y, qfz = pz ? w2.y : w1.z;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                {
                    const float tnx = __builtin_fmaf((float)((qnx >> (8 * k)) & 255u), ax, bnx), tfx = __builtin_fmaf((float)((qfx >> (8 * k)) & 255u), ax, bfx);
                    const float tny = __builtin_fmaf((float)((qny >> (8 * k)) & 255u), ay, bny), tfy = __builtin_fmaf((float)((qfy >> (8 * k)) & 255u), ay, bfy);
                    const float tnz = __builtin_fmaf((float)((qnz >> (8 * k)) & 255u), az, bnz), tfz = __builtin_fmaf((float)((qfz >> (8 * k)) & 255u), az, bfz);
                    const float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tNear));
                    const float t1 = fminf(fminf(tfx, tfy), fminf(tfz, lim));
                    d[k] = (t0 <= t1) ? t0 : __builtin_inff();
                }
                PT_SWAP_IF(d[0], c[0], d[1], c[1]);
                PT_SWAP_IF(d[2], c[2], d[3], c[3]);
                PT_SWAP_IF(d[0], c[0], d[2], c[2]);
                PT_SWAP_IF(d[1], c[1], d[3], c[3]);
                PT_SWAP_IF(d[1], c[1], d[2], c[2]);
                overflow |= !st.push3(d[1], c[1], d[2], c[2], d[3], c[3]);
                cur = (d[0] < __builtin_inff()) ? c[0] : PT_BVH_EMPTY;
            }
            /* S: a reached leaf's triangles go to the ring (one entry each, consecutive tickets per lane), then the lane pops */
            {
                const bool atLeaf = active && cur < 0 && cur != PT_BVH_EMPTY;
                const uint32_t code = (uint32_t)~cur;
                const uint32_t cnt = atLeaf ? (code & 7u) + 1u : 0u;
                uint32_t off = 0, total = 0;
#pragma unroll
                for (uint32_t k = 0; k < PT_LEAF_MAX; ++k)
                {
                    const unsigned long long m = __ballot(cnt > k);
                    off += (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    total += (uint32_t)__popcll(m);
                }
                if (atLeaf)
                {
                    const uint32_t first = code >> 3;
#pragma unroll
                    for (uint32_t k = 0; k < PT_LEAF_MAX; ++k)
                        if (k < cnt) queue[(qTail + off + k) & (kQueueCap - 1)] = ((uint32_t)lane << kQueueTriBits) | (first + k);
                    lastTicket = qTail + off + cnt - 1u;
                    cur = PT_BVH_EMPTY;
                }
                qTail += total;
                if (active && cur == PT_BVH_EMPTY)
                {
                    float t; int32_t c;
                    if (st.pop(&t, &c))
                    {
                        const float lim = __uint_as_float((uint32_t)(best[laneBase + lane] >> 32));
                        if (t <= lim) cur = c;
                    }
                }
            }
            /* B: 64 (ray, triangle) pairs at a time, one per lane, while the ring holds that many -- all of it when no lane has a node */
            {
                const bool drain = __ballot(active && cur >= 0) == 0ull;
                while ((int32_t)(qTail - qHead) >= 64 || (drain && qTail != qHead))
                {
                    if (COUNT) ++cExecB;
                    const uint32_t avail = qTail - qHead;
                    const bool mine = (uint32_t)lane < avail;
                    const uint32_t e = mine ? queue[(qHead + (uint32_t)lane) & (kQueueCap - 1)] : 0u;
                    const int owner = (int)(e >> kQueueTriBits);
                    /* the owner's ray as it has prepared it (every lane takes part in the cross-lane reads) */
                    RayPre fr;
                    const int fk = __shfl(kpack, owner, 64);
                    fr.kx = fk & 3; fr.ky = (fk >> 2) & 3; fr.kz = (fk >> 4) & 3;
                    fr.Sx = __shfl(r.Sx, owner, 64); fr.Sy = __shfl(r.Sy, owner, 64); fr.Sz = __shfl(r.Sz, owner, 64);
                    fr.ox = __shfl(r.ox, owner, 64); fr.oy = __shfl(r.oy, owner, 64); fr.oz = __shfl(r.oz, owner, 64);
                    const float fNear = __shfl(tNear, owner, 64);
                    const int32_t fTgt = TGT ? __shfl(tgt, owner, 64) : -1;
                    unsigned long long key = 0ull; bool found = false;
                    float vn = 0.0f, wn = 0.0f, det = 1.0f;
                    if (mine)
                    {
                        if (COUNT) ++cTris;
                        const float4* tp = (const float4*)((const char*)tris + (e & ((1u << kQueueTriBits) - 1u)) * 48u);      /* < 2^26 triangles here (trace_queue_ok): the byte offset fits 32 bits */
                        const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                        const float lim = __uint_as_float((uint32_t)(best[laneBase + owner] >> 32));
                        float t;
                        if (tri_test_raw(fr, p0, p1, p2, fNear, lim, &t, &vn, &wn, &det))
                        {
                            const int32_t prim = __float_as_int(p0.w);
                            if (TGT && fTgt >= 0)
                            {
                                /* t <= the target's own distance: anything but the target itself, or an equally near triangle with a higher id, is in
                                 * front: the key drops to 0 (t = 0 culls everything that is left of the walk), the owner reads it as "blocked" */
                                if (prim != fTgt && (t < lim || prim < fTgt)) __hip_atomic_fetch_min((unsigned long long*)(best + laneBase + owner), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            else
                            {
                                key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(uint32_t)prim;
                                found = true;
                                __hip_atomic_fetch_min((unsigned long long*)(best + laneBase + owner), key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                    wave_sync();
                    if (found && best[laneBase + owner] == key)
                    {
                        sNum[0][laneBase + owner] = vn; sNum[1][laneBase + owner] = wn; sNum[2][laneBase + owner] = det;
                    }
                    wave_sync();
                    qHead += avail < 64u ? avail : 64u;
                }
            }
            /* done: the walk is over and the ring has been served past the ray's last entry.  A blocked target ray and an any-hit ray that has
             * its hit stop walking at once; they too wait for their entries, which must not meet the lane's next ray */
            if (active)
            {
                const unsigned long long k = best[laneBase + lane];
                if ((TGT && tgt >= 0 && k == 0ull) || (ANY && (uint32_t)k != kNoPrim)) { cur = PT_BVH_EMPTY; st.init(); }
                if (cur == PT_BVH_EMPTY && st.sp == 0 && (int32_t)(qHead - lastTicket) > 0)
                {
                    if (COUNT) cWorst = max(cWorst, rayNodes);
                    active = false;             /* the record is stored at the next refill */
                }
            }
            const int32_t idleCount = (int32_t)__popcll(__ballot(!active));
            if (idleCount == 64 || (idleCount >= PT_REFILL_IDLE && (cursor < sliceEnd || moreGlobal))) break;
        }
        /* the records of the rays that finished during this round of trips */
        if (!active && ri >= 0)
        {
            const unsigned long long k = best[laneBase + lane];
            DHit h;
            if (ANY) occluded[ri] = (uint32_t)k != kNoPrim ? 1 : 0;
            else if (TGT && tgt >= 0)
            {
                if (k == 0ull) { h.prim = -1; h.t = tFar0; h.u = 0.0f; h.v = 0.0f; hits[ri] = h; }      /* something in front: a miss; else the caller's record stands */
            }
            else
            {
                h.prim = -1; h.t = tFar0; h.u = 0.0f; h.v = 0.0f;
                if ((uint32_t)k != kNoPrim)
                {
                    const float det = sNum[2][laneBase + lane];
                    h.prim = (int32_t)(uint32_t)k; h.t = __uint_as_float((uint32_t)(k >> 32));
                    h.u = sNum[0][laneBase + lane] / det; h.v = sNum[1][laneBase + lane] / det;
                }
                hits[ri] = h;
            }
            ri = -1;
        }
    }
    if (overflow) atomicOr(status, 1u);
    if (COUNT)
    {
        atomicAdd(&work[0], (unsigned long long)cNodes);
        atomicAdd(&work[1], (unsigned long long)cTris);
        atomicAdd(&work[2], (unsigned long long)cRays);
        atomicMax(&work[3], (unsigned long long)cWorst);
        if (lane == 0)
        {
            atomicAdd(&work[4], (unsigned long long)cTrips); atomicAdd(&work[5], (unsigned long long)cExecA);
            atomicAdd(&work[6], (unsigned long long)cExecB);
        }
    }
}

/*
 * Which of the two kernels a launch uses.  Measured on the 1 M-triangle scene, incoherent rays (gpurun_out/r3n, profiles/r03_*): the
 * queued kernel runs its triangle code at 63 of 64 lanes (14.5 in k_trace) in a quarter of the executions and takes 7 % fewer
 * trips, but its ring bookkeeping costs every trip what that saves -- 8.5 % fewer VALU instructions, equal time at 4 M rays
 * (1958 against 1939 Mrays/s), 2 % slower on the 5.4 M-ray launches of a 4K frame, where delayed culling adds 8 % triangle tests.
 * Where it wins is the launch that cannot fill the machine for long: a ray's dependent chain is shorter when it never waits for a
 * vote (512 k rays 0.48 against 0.54 ms, 64 k rays 0.34 against 0.43 ms).  So: queued below kQueueMaxRays rays, per lane above.
 * PIMGPU_TRACE_QUEUE = 0 / 1 forces one of them (tests run both); a ring entry names a triangle in 26 bits.
 */
static constexpr int64_t kQueueMaxRays = 3 << 19;       /* 1.5 M */
static bool trace_queue_ok(const DBvh& bvh, int64_t rays)
{
    const char* e = getenv("PIMGPU_TRACE_QUEUE");       /* read per launch: the tests switch it */
    const int forced = (e && *e) ? atoi(e) : -1;
    if (bvh.triCount >= (1 << kQueueTriBits)) return false;
    return forced >= 0 ? forced != 0 : rays <= kQueueMaxRays;
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
    {
        int perCUq = kTraceBlocksPerCU;                      /* the queued kernel and its target build share the grid */
        PT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCUq, k_traceq<false, false>, kTraceThreads, 0));
        perCU = std::min(perCU, perCUq);
        PT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCUq, k_traceq<false, false, true>, kTraceThreads, 0));
        perCU = std::min(perCU, perCUq);
    }
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
    const bool queued = trace_queue_ok(bvh, n);
#define PT_TRACE_ARGS bvh, rayO, rayD, hits, noOccluded, n, countDev, res->spill, res->status, res->work, cc, cn, dc, cap, targets
#define PT_TRACE_PICK(COUNTING, TARGETS) (queued ? (void (*)(DBvh, const float4*, const float4*, DHit*, int32_t*, int32_t, const int32_t*, unsigned long long*, uint32_t*, unsigned long long*, uint32_t*, uint32_t*, int32_t, int32_t, const int32_t*))k_traceq<false, COUNTING, TARGETS> \
                                                 : (void (*)(DBvh, const float4*, const float4*, DHit*, int32_t*, int32_t, const int32_t*, unsigned long long*, uint32_t*, unsigned long long*, uint32_t*, uint32_t*, int32_t, int32_t, const int32_t*))k_trace<false, COUNTING, TARGETS>)
    auto kernel = res->counting ? (targets ? PT_TRACE_PICK(true, true) : PT_TRACE_PICK(true, false)) : (targets ? PT_TRACE_PICK(false, true) : PT_TRACE_PICK(false, false));
    if (res->counting)
    {
        if (t0) hipEventRecord(t0, stream);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kTraceThreads), 0, stream, PT_TRACE_ARGS);
        if (t1) hipEventRecord(t1, stream);
    }
    else if (t0 && t1) hipExtLaunchKernelGGL(kernel, dim3(blocks), dim3(kTraceThreads), 0, stream, t0, t1, 0, PT_TRACE_ARGS);
    else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kTraceThreads), 0, stream, PT_TRACE_ARGS);
#undef PT_TRACE_PICK
#undef PT_TRACE_ARGS
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
    if (trace_queue_ok(bvh, n)) k_traceq<true, false><<<blocks, kTraceThreads, 0, stream>>>(bvh, rayO, rayD, nullptr, occluded, n, nullptr, res->spill, res->status, res->work, cc, cn, dc, cap, nullptr);
    else k_trace<true, false><<<blocks, kTraceThreads, 0, stream>>>(bvh, rayO, rayD, nullptr, occluded, n, nullptr, res->spill, res->status, res->work, cc, cn, dc, cap, nullptr);
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
