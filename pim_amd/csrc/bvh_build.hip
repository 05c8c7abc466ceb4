/*
 * bvh_build.hip -- GPU BVH build; replaces rtcCommitScene (rendering/path_tracer.c:618-690,
 * RTC_GEOMETRY_TYPE_TRIANGLE, identity index buffer, primID*3 == iVert).
 *
 * Pipeline, all on one stream (the collapse reads its level table back every 8 levels):
 *   tri_bounds  : per-triangle padded AABB + scene centroid bounds (wave reduce + ordered-uint atomics)
 *   morton      : 63-bit Morton code of the centroid (21 bits per axis)
 *   radix sort  : hand-written 8-bit LSD radix sort of (u64 key, u32 triangle) pairs,
 *                 wave64 ballot-match ranking (stable), 3 kernels per pass
 *   ploc        : the binary tree bottom-up from the sorted triangles by parallel locally-ordered clustering
 *                 (nearest neighbour by merged surface area within +-4 places, mutual pairs merge, ordered
 *                 compaction; the last 256 clusters in one workgroup), boxes made on the way, then the
 *                 triangles re-ordered by their place in the tree.  PIMGPU_BVH_BUILDER=lbvh keeps the
 *                 Karras 2012 topology + bottom-up refit (agent-scope acq_rel arrival counter) for A/B
 *   collapse    : top-down greedy SAH collapse into 4-wide nodes of 128 B (layout: pt_device.h), compact and
 *                 breadth-first; subtrees with <= PT_LEAF_MAX triangles become one leaf
 *   compress    : 64-byte copy of every wide node for the batched traversal kernel
 *   emit_tris   : triangles copied in leaf order (3 float4 each, primID in p0.w)
 *
 * The closest-hit contract (lowest primID among equal t, exact fp32 watertight test per lane)
 * makes query RESULTS independent of the tree's shape; the tree only has to be conservative,
 * which the padded triangle boxes guarantee.
 */
#include "pt_host.h"
#include <algorithm>
#include <cstdlib>

namespace pt {

static constexpr int kSortThreads = 256;
static constexpr int kSortRounds = 8;                      /* keys per thread */
static constexpr int kSortTile = kSortThreads * kSortRounds;
static constexpr int kMaxWideLevels = 120;                  /* the binary LBVH is at most 64 + 32 deep */

__device__ __forceinline__ uint32_t float_to_ordered(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ __forceinline__ float wave_min(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

/* bounds[0..2] = ordered min of centroids, bounds[3..5] = ordered max */
__global__ void k_bounds_init(uint32_t* bounds)
{
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;
}

/* grid-stride: a block folds its triangles' centroids in registers, waves, then LDS, and ONE lane per block issues the six
 * atomics (a single address sustains only ~90 atomics per microsecond: one set per wave made this kernel 4.3 ms of an
 * 11 ms build at 4 M triangles, 21 ms at 20 M) */
__global__ __launch_bounds__(256) void k_tri_bounds(const v4* __restrict__ positions, int32_t triCount,
    float4* __restrict__ triLo, float4* __restrict__ triHi, uint32_t* __restrict__ bounds)
{
    float c[3] = { 3.0e38f, 3.0e38f, 3.0e38f };
    float d[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < triCount; i += gridDim.x * blockDim.x)
    {
        const v4 p0 = positions[i * 3 + 0], p1 = positions[i * 3 + 1], p2 = positions[i * 3 + 2];
        float lo[3] = { fminf(p0.x, fminf(p1.x, p2.x)), fminf(p0.y, fminf(p1.y, p2.y)), fminf(p0.z, fminf(p1.z, p2.z)) };
        float hi[3] = { fmaxf(p0.x, fmaxf(p1.x, p2.x)), fmaxf(p0.y, fmaxf(p1.y, p2.y)), fmaxf(p0.z, fmaxf(p1.z, p2.z)) };
        float ctr[3];
        for (int a = 0; a < 3; ++a)
        {
            ctr[a] = 0.5f * (lo[a] + hi[a]);
            c[a] = fminf(c[a], ctr[a]); d[a] = fmaxf(d[a], ctr[a]);
            const float m = fmaxf(fmaxf(fabsf(lo[a]), fabsf(hi[a])), 1e-3f);
            const float pad = m * 4.0e-7f;          /* conservative against the rounding of the slab test */
            lo[a] -= pad; hi[a] += pad;
        }
        triLo[i] = make_float4(lo[0], lo[1], lo[2], ctr[0]);
        triHi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    __shared__ float sh[6][4];
    const int wave = threadIdx.x >> 6;
    for (int a = 0; a < 3; ++a)
    {
        const float mn = wave_min(c[a]), mx = wave_max(d[a]);
        if ((threadIdx.x & 63) == 0) { sh[a][wave] = mn; sh[3 + a][wave] = mx; }
    }
    __syncthreads();
    if (threadIdx.x < 3)
    {
        const int a = threadIdx.x;
        const float mn = fminf(fminf(sh[a][0], sh[a][1]), fminf(sh[a][2], sh[a][3]));
        const float mx = fmaxf(fmaxf(sh[3 + a][0], sh[3 + a][1]), fmaxf(sh[3 + a][2], sh[3 + a][3]));
        if (mn <= mx)
        {
            atomicMin(&bounds[a], float_to_ordered(mn));
            atomicMax(&bounds[3 + a], float_to_ordered(mx));
        }
    }
}

__device__ __forceinline__ uint64_t expand_bits21(uint32_t v)
{
    uint64_t x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

/* 16 bits spread to every third bit (48-bit codes) */
__device__ __forceinline__ uint64_t expand_bits16(uint32_t v)
{
    uint64_t x = v & 0xffffu;
    x = (x | (x << 16)) & 0x0000ff0000ffull;
    x = (x | (x << 8)) & 0x00f00f00f00full;
    x = (x | (x << 4)) & 0x0c30c30c30c3ull;
    x = (x | (x << 2)) & 0x249249249249ull;
    return x;
}

/*
 * Morton code of the centroid, PT_MORTON_BITS per axis.  PLOC only needs the order to be locally sensible -- its search looks
 * +-4 places, and equal codes stay in input order (the sort is stable) -- so the codes can be far narrower than the 63 bits an
 * LBVH wants, and every 8 bits less is one sort pass less.  20 M triangles, build ms | nodes + triangles per incoherent ray at
 * 1 M / at 20 M (scripts/ab_morton.sh):  21 bits (8 passes) 13.15 | 38.9 + 11.5;  16 bits (6) 9.91 | 38.9 + 11.5 / 43.8 + 18.0;
 * 13 bits (5) 9.72 | 39.3 + 11.6 / 43.9 + 18.0;  12 bits (5) 9.65 | 39.3 + 11.6 / 42.5 + 17.4;  10 bits (4) 9.29 | 39.0 + 11.5 /
 * 44.0 + 18.1.  Default 13: a cell of extent / 8192 (1.2 cm in the 100 m scenes) is still below the triangles of a 20 M-triangle
 * scene, so the order inside a cell does not have to come from the input; 10 bits measure the same HERE because the synthetic
 * scenes list an object's triangles together.
 */
#ifndef PT_MORTON_BITS
#define PT_MORTON_BITS 13
#endif
static constexpr int kMortonKeyBits = 3 * PT_MORTON_BITS;
static constexpr int kSortPasses = (kMortonKeyBits + 7) / 8;      /* the build follows the buffer the last pass wrote (kin / vin), odd or even */

/* kIndexBits: with at most 2^25 triangles and codes of at most 39 bits the triangle's index rides in the low bits of its key, and a
 * sort pass moves 8 bytes per triangle instead of 12 (vals == nullptr; the passes start at bit 25, k_unpack_vals reads the order off
 * the sorted keys).  Equal codes stay in index order either way: the sort is stable. */
static constexpr int kIndexBits = 25;
static constexpr bool kCanPackIndex = kMortonKeyBits + kIndexBits <= 64;

__global__ __launch_bounds__(256) void k_unpack_vals(const uint64_t* __restrict__ keys, int32_t n, uint32_t* __restrict__ vals)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vals[i] = (uint32_t)(keys[i] & ((1ull << kIndexBits) - 1ull));
}

__global__ __launch_bounds__(256) void k_morton(const float4* __restrict__ triLo, const float4* __restrict__ triHi,
    int32_t triCount, const uint32_t* __restrict__ bounds, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= triCount) return;
    const float4 lo = triLo[i], hi = triHi[i];
    uint32_t q[3];
    const float cc[3] = { 0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z) };
    for (int a = 0; a < 3; ++a)
    {
        const float mn = ordered_to_float(bounds[a]), mx = ordered_to_float(bounds[3 + a]);
        const float ext = mx - mn;
        float t = (ext > 0.0f) ? (cc[a] - mn) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const uint32_t cells = 1u << PT_MORTON_BITS;
        uint32_t v = (uint32_t)(t * (float)cells);
        q[a] = v > cells - 1u ? cells - 1u : v;
    }
    const uint64_t code = PT_MORTON_BITS > 16 ? (expand_bits21(q[0]) << 2) | (expand_bits21(q[1]) << 1) | expand_bits21(q[2])
                                              : (expand_bits16(q[0]) << 2) | (expand_bits16(q[1]) << 1) | expand_bits16(q[2]);
    if (vals) { keys[i] = code; vals[i] = (uint32_t)i; }
    else keys[i] = (code << kIndexBits) | (uint64_t)(uint32_t)i;
}

/* ---------------------------------------------------------------------------------------------- */
/* radix sort, 8 bits per pass.  counts layout: counts[digit * numBlocks + block]                  */

__global__ __launch_bounds__(kSortThreads) void k_sort_hist(const uint64_t* __restrict__ keys, int32_t n, int shift,
    uint32_t* __restrict__ counts, int32_t numBlocks)
{
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
    for (int r = 0; r < kSortRounds; ++r)
    {
        const int64_t i = base + r * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&hist[(uint32_t)(keys[i] >> shift) & 0xffu], 1u);
    }
    __syncthreads();
    counts[(size_t)threadIdx.x * numBlocks + blockIdx.x] = hist[threadIdx.x];
}

/* one block per digit: exclusive scan of that digit's row of per-block counts, digit total out */
__global__ __launch_bounds__(256) void k_sort_scan(uint32_t* __restrict__ counts, int32_t numBlocks, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t part[256];
    uint32_t* row = counts + (size_t)blockIdx.x * numBlocks;
    const int32_t per = (numBlocks + 255) / 256;
    const int32_t b0 = threadIdx.x * per, b1 = min(b0 + per, numBlocks);
    uint32_t s = 0;
    for (int32_t b = b0; b < b1; ++b) s += row[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t run = 0;
        for (int i = 0; i < 256; ++i) { uint32_t v = part[i]; part[i] = run; run += v; }
        totals[blockIdx.x] = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (int32_t b = b0; b < b1; ++b) { uint32_t v = row[b]; row[b] = run; run += v; }
}

__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const uint64_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn,
    uint64_t* __restrict__ keysOut, uint32_t* __restrict__ valsOut, int32_t n, int shift,
    const uint32_t* __restrict__ counts, const uint32_t* __restrict__ totals, int32_t numBlocks)
{
    constexpr int kWaves = kSortThreads / 64;
    __shared__ uint32_t waveCnt[kWaves][256];
    __shared__ uint32_t digitBase[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < kWaves; ++w) waveCnt[w][threadIdx.x] = 0;
    /* global base of each digit = exclusive scan of the 256 digit totals */
    digitBase[threadIdx.x] = totals[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t run = 0;
        for (int i = 0; i < 256; ++i) { uint32_t v = digitBase[i]; digitBase[i] = run; run += v; }
    }
    __syncthreads();

    /* each wave owns a contiguous run of kSortRounds*64 keys: stable rank = (wave, round, lane) order */
    const int64_t base = (int64_t)blockIdx.x * kSortTile + (int64_t)wave * (kSortRounds * 64);
    uint64_t key[kSortRounds]; uint32_t val[kSortRounds]; uint32_t rank[kSortRounds];
    const uint64_t ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < kSortRounds; ++r)
    {
        const int64_t i = base + r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? keysIn[i] : ~0ull;
        val[r] = (valid && valsIn) ? valsIn[i] : 0u;
        const uint32_t d = (uint32_t)(key[r] >> shift) & 0xffu;
        uint64_t match = __ballot(valid);
        for (int b = 0; b < 8; ++b)
        {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            match &= bit ? m : ~m;
        }
        const uint32_t prev = waveCnt[wave][d];
        rank[r] = prev + (uint32_t)__popcll(match & ltMask);
        __builtin_amdgcn_wave_barrier();
        if (valid && (match & ltMask) == 0) waveCnt[wave][d] = prev + (uint32_t)__popcll(match);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    /* exclusive prefix over the waves, per digit (thread = digit) */
    {
        uint32_t run = digitBase[threadIdx.x] + counts[(size_t)threadIdx.x * numBlocks + blockIdx.x];
        for (int w = 0; w < kWaves; ++w) { uint32_t v = waveCnt[w][threadIdx.x]; waveCnt[w][threadIdx.x] = run; run += v; }
    }
    __syncthreads();
    for (int r = 0; r < kSortRounds; ++r)
    {
        const int64_t i = base + r * 64 + lane;
        if (i < n)
        {
            const uint32_t d = (uint32_t)(key[r] >> shift) & 0xffu;
            const uint32_t dst = waveCnt[wave][d] + rank[r];
            keysOut[dst] = key[r];
            if (valsIn) valsOut[dst] = val[r];
        }
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* LBVH topology (Karras 2012).  Node ids: inner i in [0, n-1), leaf j -> (n-1) + j.               */

__device__ __forceinline__ int lbvh_delta(const uint64_t* __restrict__ keys, int32_t n, int32_t i, int32_t j)
{
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

__global__ __launch_bounds__(256) void k_karras(const uint64_t* __restrict__ keys, int32_t n,
    int32_t* __restrict__ child0, int32_t* __restrict__ child1, int32_t* __restrict__ parent,
    int32_t* __restrict__ rangeFirst, int32_t* __restrict__ rangeLast)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int32_t lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int32_t l = 0;
    for (int32_t t = lmax >> 1; t >= 1; t >>= 1)
    {
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    }
    const int32_t j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int32_t s = 0;
    int32_t t = l;
    do
    {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int32_t gamma = i + s * d + min(d, 0);
    const int32_t first = min(i, j), last = max(i, j);
    const int32_t c0 = (first == gamma) ? (n - 1) + gamma : gamma;
    const int32_t c1 = (last == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
    child0[i] = c0; child1[i] = c1;
    parent[c0] = i; parent[c1] = i;
    if (i == 0) parent[0] = -1;
    rangeFirst[i] = first; rangeLast[i] = last;
}

/* bottom-up refit: nodeLo/nodeHi for inner nodes; leaves read the (padded) triangle boxes */
__global__ __launch_bounds__(256) void k_refit(int32_t n, const uint32_t* __restrict__ vals,
    const float4* __restrict__ triLo, const float4* __restrict__ triHi,
    const int32_t* __restrict__ child0, const int32_t* __restrict__ child1, const int32_t* __restrict__ parent,
    float4* nodeLo, float4* nodeHi, uint32_t* flags)
{
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int32_t node = parent[(n - 1) + j];
    while (node >= 0)
    {
        /* the first arriver stops; the second sees both children complete (acq_rel at agent scope:
         * the sibling's box was written by another workgroup, possibly on another XCD) */
        const uint32_t old = __hip_atomic_fetch_add(&flags[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0u) return;
        const int32_t c0 = child0[node], c1 = child1[node];
        float4 lo0, hi0, lo1, hi1;
        if (c0 >= n - 1) { const uint32_t t = vals[c0 - (n - 1)]; lo0 = triLo[t]; hi0 = triHi[t]; }
        else { lo0 = nodeLo[c0]; hi0 = nodeHi[c0]; }
        if (c1 >= n - 1) { const uint32_t t = vals[c1 - (n - 1)]; lo1 = triLo[t]; hi1 = triHi[t]; }
        else { lo1 = nodeLo[c1]; hi1 = nodeHi[c1]; }
        nodeLo[node] = make_float4(fminf(lo0.x, lo1.x), fminf(lo0.y, lo1.y), fminf(lo0.z, lo1.z), 0.0f);
        nodeHi[node] = make_float4(fmaxf(hi0.x, hi1.x), fmaxf(hi0.y, hi1.y), fmaxf(hi0.z, hi1.z), 0.0f);
        node = parent[node];
    }
}


/* ---------------------------------------------------------------------------------------------- */
/* PLOC (parallel locally-ordered clustering, Meister & Bittner 2018): the binary tree bottom-up    */
/*                                                                                                  */
/* The Morton-sorted triangles start as one cluster each.  Every round a cluster looks kPlocRadius  */
/* places to either side for the neighbour whose union with it has the smallest surface area;       */
/* mutual choices merge into a new node, the list is compacted in order, and so on until one        */
/* cluster -- the root -- is left.  Ties go to the lower index, which makes the choice a total      */
/* order (the overall best pair is always mutual: every round merges something) and the tree        */
/* independent of scheduling up to the numbering of its nodes.  Merges follow surface area, not     */
/* the bit pattern of the codes, which is what a spatial-median LBVH lacks: fewer nodes per ray.    */
/* A merge may skip over clusters in between, so the leaves of a subtree are no longer a range of   */
/* the sorted order: the triangles are re-ordered afterwards by their place in the finished tree.   */

#ifndef PT_PLOC_RADIUS
#define PT_PLOC_RADIUS 4
#endif
static constexpr int kPlocRadius = PT_PLOC_RADIUS;
/* threads per block of the clustering rounds: 1024 = a quarter of the per-block counts the one-workgroup scan of every round has
 * to walk (56 rounds at 20 M triangles: 12.5 -> 11.7 ms per build against 256) */
#ifndef PT_PLOC_THREADS
#define PT_PLOC_THREADS 1024
#endif
static constexpr int kPlocThreads = PT_PLOC_THREADS;
static constexpr int kPlocFinish = 512;             /* at most this many clusters: one workgroup finishes in LDS (37 KB) */

struct PlocState { int32_t m, mPrev, nodeCounter, rootId; };

__device__ __forceinline__ float ploc_area(float4 alo, float4 ahi, float4 blo, float4 bhi)
{
    const float dx = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x);
    const float dy = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y);
    const float dz = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
    return dx * dy + dy * dz + dz * dx;
}

__global__ __launch_bounds__(256) void k_ploc_init(int32_t n, const uint32_t* __restrict__ vals,
    const float4* __restrict__ triLo, const float4* __restrict__ triHi,
    int32_t* __restrict__ cid, float4* __restrict__ cLo, float4* __restrict__ cHi, PlocState* __restrict__ state)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { state->m = n; state->mPrev = n; state->nodeCounter = 0; state->rootId = n - 1; }
    if (i >= n) return;
    const uint32_t t = vals[i];
    cid[i] = (n - 1) + i;
    cLo[i] = triLo[t]; cHi[i] = triHi[t];
}

/* is neighbour j (merged area a) a better choice for cluster `self` than bestJ (area best)?  Smaller area; among EQUAL areas the
 * neighbour with the smaller (self xor j) -- a rule that is the same seen from either side, so a run of clusters with equal boxes
 * (coincident triangles: duplicated geometry, points) pairs up as (0,1) (2,3) ... and halves every round.  With "the lower index
 * wins" only one pair of such a run was mutual per round: 1500 copies of a triangle took 1500 rounds and gave a chain deeper than
 * the collapse's level table (tests/test_gpu_build_edges.py). */
__device__ __forceinline__ bool ploc_better(float a, float best, int32_t self, int32_t j, int32_t bestJ)
{
    return a < best || (a == best && (uint32_t)(self ^ j) < (uint32_t)(self ^ bestJ));
}

/* nearest neighbour by merged surface area within +-kPlocRadius; the block's boxes are staged through LDS */
__global__ __launch_bounds__(kPlocThreads) void k_ploc_nn(const PlocState* __restrict__ state,
    const float4* __restrict__ cLo, const float4* __restrict__ cHi, int32_t* __restrict__ nn, int32_t stride = 1)
{
    const int32_t m = state->m;
    const int32_t base = blockIdx.x * kPlocThreads;
    if (base >= m) return;
    __shared__ float4 sLo[kPlocThreads + 2 * kPlocRadius], sHi[kPlocThreads + 2 * kPlocRadius];
    for (int k = threadIdx.x; k < kPlocThreads + 2 * kPlocRadius; k += kPlocThreads)
    {
        const int32_t g = base - kPlocRadius + k;
        if (g >= 0 && g < m) { sLo[k] = cLo[(size_t)g * stride]; sHi[k] = cHi[(size_t)g * stride]; }      /* stride 2: interleaved records (packed round 1) */
    }
    __syncthreads();
    const int32_t i = base + (int32_t)threadIdx.x;
    if (i >= m) return;
    const float4 lo = sLo[threadIdx.x + kPlocRadius], hi = sHi[threadIdx.x + kPlocRadius];
    float best = __builtin_inff(); int32_t bestJ = -1;
    for (int d = -kPlocRadius; d <= kPlocRadius; ++d)
    {
        const int32_t j = i + d;
        if (d == 0 || j < 0 || j >= m) continue;
        const float a = ploc_area(lo, hi, sLo[threadIdx.x + kPlocRadius + d], sHi[threadIdx.x + kPlocRadius + d]);
        if (ploc_better(a, best, i, j, bestJ)) { best = a; bestJ = j; }
    }
    nn[i] = bestJ;
}

__device__ __forceinline__ int32_t ploc_count_of(int32_t c, int32_t n, const int32_t* __restrict__ count) { return c >= n - 1 ? 1 : count[c]; }

/*
 * Mutual pairs merge: the lower index keeps the new cluster, the higher one leaves the list.  Three launches per round and
 * no atomics (node ids taken with one returning atomic per wave on a single counter cost 2 of the 3 ms the rounds took at
 * 4 M triangles; a single address sustains ~90 per microsecond): k_ploc_merge marks survivors and merging lanes and counts
 * both per block, k_ploc_scan turns the counts into list positions and node ids, k_ploc_scatter makes the nodes and
 * writes the next list.  Node numbering is deterministic this way.
 */
#define PLOC_KEPT   1
#define PLOC_MERGES 2

__global__ __launch_bounds__(kPlocThreads) void k_ploc_merge(const PlocState* __restrict__ state, const int32_t* __restrict__ nn,
    uint8_t* __restrict__ keep, int32_t* __restrict__ blockCount, int32_t* __restrict__ mergeCount)
{
    const int32_t m = state->m;
    const int32_t base = blockIdx.x * kPlocThreads;
    if (base >= m) { if (threadIdx.x == 0) { blockCount[blockIdx.x] = 0; mergeCount[blockIdx.x] = 0; } return; }
    const int32_t i = base + (int32_t)threadIdx.x;
    bool kept = false, merges = false;
    if (i < m)
    {
        const int32_t j = nn[i];
        const bool mutual = j >= 0 && nn[j] == i;
        kept = !(mutual && j < i);
        merges = mutual && i < j;
        keep[i] = (kept ? PLOC_KEPT : 0) | (merges ? PLOC_MERGES : 0);
    }
    __shared__ int32_t sCnt[2][kPlocThreads / 64];
    const int lane = threadIdx.x & 63;
    const uint64_t km = __ballot(kept), mm = __ballot(merges);
    if (lane == 0) { sCnt[0][threadIdx.x >> 6] = (int32_t)__popcll(km); sCnt[1][threadIdx.x >> 6] = (int32_t)__popcll(mm); }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        int32_t t = 0, u = 0;
        for (int w = 0; w < kPlocThreads / 64; ++w) { t += sCnt[0][w]; u += sCnt[1][w]; }
        blockCount[blockIdx.x] = t; mergeCount[blockIdx.x] = u;
    }
}

/*
 * k_ploc_nn + k_ploc_merge in one launch (the packed path): a block finds the nearest neighbours of its own elements AND of the
 * kPlocRadius elements to either side (boxes with a halo of 2 * kPlocRadius), so that "is the choice mutual" is answered from LDS;
 * the halo's neighbours are found twice, by two blocks, with the same result.  One launch and one pass over nn[] less per round.
 */
__global__ __launch_bounds__(kPlocThreads) void k_ploc_nn_merge(const PlocState* __restrict__ state,
    const float4* __restrict__ cLo, const float4* __restrict__ cHi, int32_t stride, int32_t* __restrict__ nn,
    uint8_t* __restrict__ keep, int32_t* __restrict__ blockCount, int32_t* __restrict__ mergeCount)
{
    constexpr int R = kPlocRadius, kBox = kPlocThreads + 4 * R, kNn = kPlocThreads + 2 * R;
    const int32_t m = state->m;
    const int32_t base = blockIdx.x * kPlocThreads;
    if (base >= m) { if (threadIdx.x == 0) { blockCount[blockIdx.x] = 0; mergeCount[blockIdx.x] = 0; } return; }
    __shared__ float4 sLo[kBox], sHi[kBox];
    __shared__ int32_t sNn[kNn];
    __shared__ int32_t sCnt[2][kPlocThreads / 64];
    for (int k = threadIdx.x; k < kBox; k += kPlocThreads)
    {
        const int32_t g = base - 2 * R + k;
        if (g >= 0 && g < m) { sLo[k] = cLo[(size_t)g * stride]; sHi[k] = cHi[(size_t)g * stride]; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kNn; k += kPlocThreads)
    {
        const int32_t e = base - R + k;                        /* element whose neighbour this is; its box sits at sLo[k + R] */
        int32_t bestJ = -1;
        if (e >= 0 && e < m)
        {
            const float4 lo = sLo[k + R], hi = sHi[k + R];
            float best = __builtin_inff();
            for (int d = -R; d <= R; ++d)
            {
                const int32_t j = e + d;
                if (d == 0 || j < 0 || j >= m) continue;
                const float a = ploc_area(lo, hi, sLo[k + R + d], sHi[k + R + d]);
                if (ploc_better(a, best, e, j, bestJ)) { best = a; bestJ = j; }
            }
        }
        sNn[k] = bestJ;
    }
    __syncthreads();
    const int32_t i = base + (int32_t)threadIdx.x;
    bool kept = false, merges = false;
    if (i < m)
    {
        const int32_t j = sNn[threadIdx.x + R];
        const bool mutual = j >= 0 && sNn[j - base + R] == i;
        kept = !(mutual && j < i);
        merges = mutual && i < j;
        nn[i] = j;
        keep[i] = (kept ? PLOC_KEPT : 0) | (merges ? PLOC_MERGES : 0);
    }
    const int lane = threadIdx.x & 63;
    const uint64_t km = __ballot(kept), mm = __ballot(merges);
    if (lane == 0) { sCnt[0][threadIdx.x >> 6] = (int32_t)__popcll(km); sCnt[1][threadIdx.x >> 6] = (int32_t)__popcll(mm); }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        int32_t t = 0, u = 0;
        for (int w = 0; w < kPlocThreads / 64; ++w) { t += sCnt[0][w]; u += sCnt[1][w]; }
        blockCount[blockIdx.x] = t; mergeCount[blockIdx.x] = u;
    }
}

/* exclusive scans of the per-block survivor and merge counts (one workgroup): new list length, first node id of each block */
static constexpr int kScanThreads = 1024;

__device__ __forceinline__ int32_t block_exclusive_scan(int32_t v, int32_t* sWave, int32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) { const int32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    int32_t base = 0, all = 0;
    for (int w = 0; w < kScanThreads / 64; ++w) { const int32_t t = sWave[w]; if (w < wave) base += t; all += t; }
    __syncthreads();
    *total = all;
    return base + incl - v;
}

__global__ __launch_bounds__(kScanThreads) void k_ploc_scan(PlocState* __restrict__ state, int32_t* __restrict__ blockCount, int32_t* __restrict__ mergeCount,
    int32_t numBlocks, int32_t* __restrict__ progress = nullptr, int32_t round = 0)
{
    __shared__ int32_t sWave[kScanThreads / 64];
    const int32_t m = state->m;
    const int32_t nodeBase = state->nodeCounter;
    const int32_t used = (m + kPlocThreads - 1) / kPlocThreads < numBlocks ? (m + kPlocThreads - 1) / kPlocThreads : numBlocks;
    const int32_t per = (used + kScanThreads - 1) / kScanThreads;
    const int32_t b0 = min((int32_t)threadIdx.x * per, used), b1 = min(b0 + per, used);
    int32_t s = 0, t = 0;
    for (int32_t b = b0; b < b1; ++b) { s += blockCount[b]; t += mergeCount[b]; }
    int32_t kept = 0, merged = 0;
    int32_t run = block_exclusive_scan(s, sWave, &kept);
    int32_t ids = nodeBase + block_exclusive_scan(t, sWave, &merged);
    for (int32_t b = b0; b < b1; ++b)
    {
        const int32_t v = blockCount[b]; blockCount[b] = run; run += v;
        const int32_t u = mergeCount[b]; mergeCount[b] = ids; ids += u;
    }
    if (threadIdx.x == 0)
    {
        state->mPrev = m; state->m = kept; state->nodeCounter = nodeBase + merged;
        if (progress)
        {
            /* the host sizes the next launches from this without draining the stream: length first, then the round that made it */
            __hip_atomic_store(progress + 0, kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(progress + 1, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

/* survivors move to the other list, order kept; a merging survivor becomes the new node */
__global__ __launch_bounds__(kPlocThreads) void k_ploc_scatter(int32_t n, const PlocState* __restrict__ state, const uint8_t* __restrict__ keep,
    const int32_t* __restrict__ nn, const int32_t* __restrict__ blockOffset, const int32_t* __restrict__ mergeOffset,
    const int32_t* __restrict__ cidIn, const float4* __restrict__ cLoIn, const float4* __restrict__ cHiIn,
    int32_t* __restrict__ cidOut, float4* __restrict__ cLoOut, float4* __restrict__ cHiOut,
    int32_t* __restrict__ child0, int32_t* __restrict__ child1, int32_t* __restrict__ parent,
    float4* __restrict__ nodeLo, float4* __restrict__ nodeHi, int32_t* __restrict__ count)
{
    const int32_t m = state->mPrev;
    const int32_t base = blockIdx.x * kPlocThreads;
    if (base >= m) return;
    const int32_t i = base + (int32_t)threadIdx.x;
    const uint8_t f = i < m ? keep[i] : 0;
    const bool kept = (f & PLOC_KEPT) != 0, merges = (f & PLOC_MERGES) != 0;
    __shared__ int32_t sCnt[2][kPlocThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t km = __ballot(kept), mm = __ballot(merges);
    if (lane == 0) { sCnt[0][wave] = (int32_t)__popcll(km); sCnt[1][wave] = (int32_t)__popcll(mm); }
    __syncthreads();
    int32_t off = blockOffset[blockIdx.x], ids = mergeOffset[blockIdx.x];
    for (int w = 0; w < wave; ++w) { off += sCnt[0][w]; ids += sCnt[1][w]; }
    if (kept)
    {
        const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int32_t dst = off + (int32_t)__popcll(km & lt);
        int32_t id = cidIn[i];
        float4 lo = cLoIn[i], hi = cHiIn[i];
        if (merges)
        {
            const int32_t j = nn[i];
            const int32_t a = id, b = cidIn[j];
            const float4 blo = cLoIn[j], bhi = cHiIn[j];
            lo = make_float4(fminf(lo.x, blo.x), fminf(lo.y, blo.y), fminf(lo.z, blo.z), 0.0f);
            hi = make_float4(fmaxf(hi.x, bhi.x), fmaxf(hi.y, bhi.y), fmaxf(hi.z, bhi.z), 0.0f);
            id = ids + (int32_t)__popcll(mm & lt);
            child0[id] = a; child1[id] = b;
            if (parent) { parent[a] = id; parent[b] = id; }      /* only the leaf-order walks of k_ploc_order read it */
            nodeLo[id] = lo; nodeHi[id] = hi;
            count[id] = ploc_count_of(a, n, count) + ploc_count_of(b, n, count);
        }
        cidOut[dst] = id; cLoOut[dst] = lo; cHiOut[dst] = hi;
    }
}

/* the last <= kPlocFinish clusters: the same rounds inside one workgroup, lists in LDS */
__global__ __launch_bounds__(kPlocFinish) void k_ploc_finish(int32_t n, PlocState* __restrict__ state,
    const int32_t* __restrict__ cid, const float4* __restrict__ cLo, const float4* __restrict__ cHi,
    int32_t* __restrict__ child0, int32_t* __restrict__ child1, int32_t* __restrict__ parent,
    float4* __restrict__ nodeLo, float4* __restrict__ nodeHi, int32_t* __restrict__ count)
{
    __shared__ int32_t sId[2][kPlocFinish];
    __shared__ float4 sLo[2][kPlocFinish], sHi[2][kPlocFinish];
    __shared__ int32_t sNn[kPlocFinish], sPos[kPlocFinish + 1], sM;
    const int t = threadIdx.x;
    int32_t m = state->m;
    if (m > kPlocFinish) return;                         /* the host only launches this once the list is short */
    if (t < m) { sId[0][t] = cid[t]; sLo[0][t] = cLo[t]; sHi[0][t] = cHi[t]; }
    __syncthreads();
    int cur = 0;
    while (m > 1)
    {
        if (t < m)
        {
            float best = __builtin_inff(); int32_t bestJ = -1;
            for (int d = -kPlocRadius; d <= kPlocRadius; ++d)
            {
                const int32_t j = t + d;
                if (d == 0 || j < 0 || j >= m) continue;
                const float a = ploc_area(sLo[cur][t], sHi[cur][t], sLo[cur][j], sHi[cur][j]);
                if (ploc_better(a, best, t, j, bestJ)) { best = a; bestJ = j; }
            }
            sNn[t] = bestJ;
        }
        __syncthreads();
        bool kept = false;
        int32_t id = -1; float4 lo = make_float4(0, 0, 0, 0), hi = lo;
        if (t < m)
        {
            const int32_t j = sNn[t];
            const bool mutual = j >= 0 && sNn[j] == t;
            kept = !(mutual && j < t);
            id = sId[cur][t]; lo = sLo[cur][t]; hi = sHi[cur][t];
            if (mutual && t < j)
            {
                const int32_t a = id, b = sId[cur][j];
                const float4 blo = sLo[cur][j], bhi = sHi[cur][j];
                lo = make_float4(fminf(lo.x, blo.x), fminf(lo.y, blo.y), fminf(lo.z, blo.z), 0.0f);
                hi = make_float4(fmaxf(hi.x, bhi.x), fmaxf(hi.y, bhi.y), fmaxf(hi.z, bhi.z), 0.0f);
                id = atomicAdd(&state->nodeCounter, 1);
                child0[id] = a; child1[id] = b;
                parent[a] = id; parent[b] = id;
                nodeLo[id] = lo; nodeHi[id] = hi;
                count[id] = ploc_count_of(a, n, count) + ploc_count_of(b, n, count);
            }
        }
        /* survivors keep their order: a ballot per wave, the waves' counts summed by everyone */
        {
            const uint64_t km = __ballot(kept);
            const int lane = t & 63, wave = t >> 6;
            if (lane == 0) sPos[wave] = (int32_t)__popcll(km);
            __syncthreads();
            int32_t before = 0, all = 0;
            for (int w = 0; w < kPlocFinish / 64; ++w) { const int32_t v = sPos[w]; if (w < wave) before += v; all += v; }
            if (t == 0) sM = all;
            if (kept)
            {
                const int32_t dst = before + (int32_t)__popcll(km & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                sId[cur ^ 1][dst] = id; sLo[cur ^ 1][dst] = lo; sHi[cur ^ 1][dst] = hi;
            }
        }
        __syncthreads();
        m = sM;
        cur ^= 1;
        __syncthreads();
    }
    if (t == 0) { state->rootId = sId[cur][0]; state->m = 1; parent[sId[cur][0]] = -1; }
}

/* place of a node's first leaf in the order of the finished tree: on the way up, every time the node hangs on the
 * right, the leaves of the left sibling come first */
__device__ __forceinline__ int32_t ploc_offset_of(int32_t c, int32_t n, const int32_t* __restrict__ child0, const int32_t* __restrict__ child1,
    const int32_t* __restrict__ parent, const int32_t* __restrict__ count)
{
    int32_t off = 0;
    for (int32_t p = parent[c]; p >= 0; c = p, p = parent[p])
        if (child1[p] == c) off += ploc_count_of(child0[p], n, count);
    return off;
}

/*
 * Place of every subtree's leaves in the tree order, in three launches instead of one walk to the root per node (2 n walks
 * of ~50 dependent loads were 3.4 ms of a 15.5 ms build at 20 M triangles): (0) the few nodes with at least kOrderBig
 * leaves walk to the root; (1) the other inner nodes walk only up to their first such ancestor; (2) a leaf takes one step
 * from its parent.
 */
static constexpr int32_t kOrderBig = 64;

__global__ __launch_bounds__(256) void k_ploc_order(int32_t phase, int32_t n, const uint32_t* __restrict__ vals,
    const int32_t* __restrict__ child0, const int32_t* __restrict__ child1, const int32_t* __restrict__ parent, const int32_t* __restrict__ count,
    int32_t* __restrict__ rangeFirst, int32_t* __restrict__ rangeLast, int32_t* __restrict__ newPos, uint32_t* __restrict__ valsOut)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (phase == 0)
    {
        if (i >= n - 1 || count[i] < kOrderBig) return;
        const int32_t off = ploc_offset_of(i, n, child0, child1, parent, count);
        rangeFirst[i] = off; rangeLast[i] = off + count[i] - 1;
    }
    else if (phase == 1)
    {
        if (i >= n - 1 || count[i] >= kOrderBig) return;
        int32_t off = 0, c = i, p = parent[c];
        for (; p >= 0; c = p, p = parent[p])
        {
            if (child1[p] == c) off += ploc_count_of(child0[p], n, count);
            if (count[p] >= kOrderBig) { off += rangeFirst[p]; break; }
        }
        rangeFirst[i] = off; rangeLast[i] = off + count[i] - 1;
    }
    else
    {
        if (i >= n) return;
        const int32_t c = (n - 1) + i, p = parent[c];
        int32_t off = 0;
        if (p >= 0) off = rangeFirst[p] + ((child1[p] == c) ? ploc_count_of(child0[p], n, count) : 0);
        newPos[i] = off; valsOut[off] = vals[i];
    }
}

/* leaf ids follow the new order */
__global__ __launch_bounds__(256) void k_ploc_remap(int32_t n, const int32_t* __restrict__ newPos, int32_t* __restrict__ child0, int32_t* __restrict__ child1)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int32_t a = child0[i], b = child1[i];
    if (a >= n - 1) child0[i] = (n - 1) + newPos[a - (n - 1)];
    if (b >= n - 1) child1[i] = (n - 1) + newPos[b - (n - 1)];
}

struct WideSlot { float lo[3], hi[3]; int32_t child; };

/*
 * ORDER = true (the PLOC tree): the place of every subtree's leaves in the order of the finished tree is handed DOWN by the
 * collapse instead of being computed beforehand by walks towards the root (k_ploc_order + k_ploc_remap: 1.1 of 11.5 ms at 20 M
 * triangles).  A wide node knows where its subtree's leaves start (wideFirst); opening a binary node gives its left child the same
 * start and its right child that start plus the left child's leaf count (count[], made by the merges); a slot that becomes a leaf
 * writes its triangles to their places in the new order (valsOut) there and then.  `vals` stays the Morton order, leaf id c is
 * place c - (n - 1) of it.  ORDER = false (the Karras tree, whose leaves are in tree order already): ranges as before.
 */

template <bool ORDER>
__device__ __forceinline__ void wide_slot_from(int32_t c, int32_t first, int32_t count, int32_t n, const uint32_t* __restrict__ vals,
    const float4* __restrict__ triLo, const float4* __restrict__ triHi,
    const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi,
    const int32_t* __restrict__ child0, const int32_t* __restrict__ child1, uint32_t* __restrict__ valsOut, WideSlot* s)
{
    float4 lo, hi;
    if (c >= n - 1)
    {
        const int32_t j = c - (n - 1);
        const uint32_t t = vals[j];
        lo = triLo[t]; hi = triHi[t];
        s->child = ~(((ORDER ? first : j) << 3) | 0);
        if (ORDER) valsOut[first] = t;
    }
    else
    {
        lo = nodeLo[c]; hi = nodeHi[c];
        s->child = (count <= PT_LEAF_MAX) ? ~((first << 3) | (count - 1)) : c;
        if (ORDER && count <= PT_LEAF_MAX)
        {
            /* the leaf's triangles, left to right: a subtree of at most PT_LEAF_MAX leaves */
            int32_t todo[2 * PT_LEAF_MAX]; int top = 0, pos = first;
            todo[top++] = c;
            while (top > 0)
            {
                const int32_t x = todo[--top];
                if (x >= n - 1) valsOut[pos++] = vals[x - (n - 1)];
                else { todo[top++] = child1[x]; todo[top++] = child0[x]; }
            }
        }
    }
    s->lo[0] = lo.x; s->lo[1] = lo.y; s->lo[2] = lo.z;
    s->hi[0] = hi.x; s->hi[1] = hi.y; s->hi[2] = hi.z;
}

/*
 * Collapse of the binary tree into 4-wide nodes, top down, one launch per level of the WIDE tree.
 * A wide node starts from the two children of its binary node and keeps opening the child with the
 * largest surface area (the greedy SAH collapse) until it has 4 slots; subtrees of <= PT_LEAF_MAX
 * triangles are leaves, and slots still free at the end are used to split such leaves into smaller ones
 * (tighter boxes for free).  Wide nodes are allocated level by level from one counter, so the array is
 * compact and breadth-first: node w of level L sits in [snap[L], snap[L+1]), wideSrc[w] is its binary node.
 */
#ifndef PT_COLLAPSE_BLOCKS
#define PT_COLLAPSE_BLOCKS 4096
#endif
static constexpr int kCollapseBlocks = PT_COLLAPSE_BLOCKS;      /* blocks of a collapse launch: 16 per CU, chunks of 256 nodes in turn */
struct Cand { int32_t id; int32_t count; int32_t first; float area; };

/* ORDER: `first` comes from the parent, the leaf count from count[] (passed as rangeLast); else both from the Karras ranges */
template <bool ORDER>
__device__ __forceinline__ Cand cand_of(int32_t c, int32_t first, int32_t n, const uint32_t* __restrict__ vals,
    const float4* __restrict__ triLo, const float4* __restrict__ triHi,
    const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi,
    const int32_t* __restrict__ rangeFirst, const int32_t* __restrict__ rangeLast)
{
    float4 lo, hi; Cand k; k.id = c; k.first = first;
    if (c >= n - 1) { const uint32_t t = vals[c - (n - 1)]; lo = triLo[t]; hi = triHi[t]; k.count = 1; if (!ORDER) k.first = c - (n - 1); }
    else
    {
        lo = nodeLo[c]; hi = nodeHi[c];
        if (ORDER) k.count = rangeLast[c];
        else { k.first = rangeFirst[c]; k.count = rangeLast[c] - k.first + 1; }
    }
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    k.area = dx * dy + dy * dz + dz * dx;
    return k;
}

__global__ void k_collapse_begin(int32_t* snap, int32_t* nodeCounter, int32_t* wideSrc, int32_t* wideFirst, const int32_t* rootId)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { snap[0] = 0; snap[1] = 1; *nodeCounter = 1; wideSrc[0] = rootId ? *rootId : 0; wideFirst[0] = 0; }
}

template <bool ORDER>
__global__ __launch_bounds__(256) void k_collapse_level(int32_t n, const uint32_t* __restrict__ vals,
    const float4* __restrict__ triLo, const float4* __restrict__ triHi,
    const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi,
    const int32_t* __restrict__ child0, const int32_t* __restrict__ child1,
    const int32_t* __restrict__ rangeFirst, const int32_t* __restrict__ rangeLast,
    int32_t* __restrict__ wideSrc, int32_t* __restrict__ wideFirst, uint32_t* __restrict__ valsOut,
    const int32_t* __restrict__ snap, int32_t* __restrict__ nodeCounter,
    float4* __restrict__ wide)
{
    const int32_t begin = snap[0], end = snap[1];
    /* The level's size is only known on the device and the host sizes the grid from a bound (4^level, at most n): at 20 M triangles
     * a deep level's launch had 78 k blocks of which a few thousand found nodes.  So: a grid of at most kCollapseBlocks blocks, each
     * taking chunks of 256 nodes in turn; a block with no chunk left is gone after one comparison. */
    __shared__ int32_t sTot[4], sBase;
    for (int32_t chunk = (int32_t)blockIdx.x; begin + chunk * (int32_t)blockDim.x < end; chunk += (int32_t)gridDim.x)
    {
    const int32_t w = begin + chunk * (int32_t)blockDim.x + (int32_t)threadIdx.x;
    const bool active = w < end;
    Cand cd[4]; int ns = 0;
    if (active)
    {
        const int32_t src = wideSrc[w];
        const int32_t at = ORDER ? wideFirst[w] : 0;
        cd[0] = cand_of<ORDER>(child0[src], at, n, vals, triLo, triHi, nodeLo, nodeHi, rangeFirst, rangeLast);
        cd[1] = cand_of<ORDER>(child1[src], at + cd[0].count, n, vals, triLo, triHi, nodeLo, nodeHi, rangeFirst, rangeLast);
        ns = 2;
        /* pass 0 opens inner subtrees (count > PT_LEAF_MAX), pass 1 spends leftover slots on splitting leaves */
        for (int pass = 0; pass < 2; ++pass)
        {
            while (ns < 4)
            {
                int best = -1; float bestArea = -1.0f;
                for (int k = 0; k < ns; ++k)
                {
                    const bool open = pass == 0 ? (cd[k].count > PT_LEAF_MAX) : (cd[k].count > 1);
                    if (open && cd[k].area > bestArea) { best = k; bestArea = cd[k].area; }
                }
                if (best < 0) break;
                const int32_t c = cd[best].id, at = cd[best].first;
                cd[best] = cand_of<ORDER>(child0[c], at, n, vals, triLo, triHi, nodeLo, nodeHi, rangeFirst, rangeLast);
                cd[ns] = cand_of<ORDER>(child1[c], at + cd[best].count, n, vals, triLo, triHi, nodeLo, nodeHi, rangeFirst, rangeLast);
                ++ns;
            }
        }
    }
    /* allocate the wide children of the whole block with one atomic (a single counter sustains ~90 per microsecond) */
    int need = 0;
    for (int k = 0; k < ns; ++k) need += (cd[k].count > PT_LEAF_MAX) ? 1 : 0;
    int incl = need;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += v; }
    const int wave = threadIdx.x >> 6;
    __syncthreads();                                        /* the chunk before this one has read sTot / sBase */
    if ((threadIdx.x & 63) == 63) sTot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const int total = sTot[0] + sTot[1] + sTot[2] + sTot[3];
        sBase = total > 0 ? atomicAdd(nodeCounter, total) : 0;
    }
    __syncthreads();
    int32_t base = sBase;
    for (int k = 0; k < wave; ++k) base += sTot[k];
    if (!active) continue;
    int32_t next = base + incl - need;
    WideSlot slot[4];
    for (int k = 0; k < 4; ++k)
    {
        if (k >= ns)
        {
            for (int a = 0; a < 3; ++a) { slot[k].lo[a] = __builtin_inff(); slot[k].hi[a] = -__builtin_inff(); }
            slot[k].child = PT_BVH_EMPTY;
            continue;
        }
        wide_slot_from<ORDER>(cd[k].id, cd[k].first, cd[k].count, n, vals, triLo, triHi, nodeLo, nodeHi, child0, child1, valsOut, &slot[k]);
        if (cd[k].count > PT_LEAF_MAX) { slot[k].child = next; wideSrc[next] = cd[k].id; if (ORDER) wideFirst[next] = cd[k].first; ++next; }
    }
    float4* q = wide + (size_t)w * 8;
    q[0] = make_float4(slot[0].lo[0], slot[1].lo[0], slot[2].lo[0], slot[3].lo[0]);
    q[1] = make_float4(slot[0].lo[1], slot[1].lo[1], slot[2].lo[1], slot[3].lo[1]);
    q[2] = make_float4(slot[0].lo[2], slot[1].lo[2], slot[2].lo[2], slot[3].lo[2]);
    q[3] = make_float4(slot[0].hi[0], slot[1].hi[0], slot[2].hi[0], slot[3].hi[0]);
    q[4] = make_float4(slot[0].hi[1], slot[1].hi[1], slot[2].hi[1], slot[3].hi[1]);
    q[5] = make_float4(slot[0].hi[2], slot[1].hi[2], slot[2].hi[2], slot[3].hi[2]);
    q[6] = make_float4(__int_as_float(slot[0].child), __int_as_float(slot[1].child), __int_as_float(slot[2].child), __int_as_float(slot[3].child));
    /* node 0's spare quad carries the largest |coordinate| of the scene: the traversal pads every box by a
     * per-ray epsilon proportional to (|origin| + this), see traverse.h */
    float maxAbs = 0.0f;
    if (w == 0)
    {
        const int32_t rootSrc = wideSrc[0];
        const float4 rl = nodeLo[rootSrc], rh = nodeHi[rootSrc];
        maxAbs = fmaxf(fmaxf(fmaxf(fabsf(rl.x), fabsf(rl.y)), fabsf(rl.z)), fmaxf(fmaxf(fabsf(rh.x), fabsf(rh.y)), fabsf(rh.z)));
    }
    q[7] = make_float4(maxAbs, 0.0f, 0.0f, 0.0f);
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* PLOC + collapse on PACKED records (round 3, the default)                                          */
/*                                                                                                    */
/* The clustering rounds and the collapse above keep a node's box, children, leaf count and a         */
/* cluster's id in arrays of their own: the collapse then fetches ~6 candidates per wide node from    */
/* five arrays each, 4 to 16 bytes at random places (3.0 of 11.5 ms at 20 M triangles), and a leaf    */
/* candidate through vals[] -> triLo / triHi, two dependent fetches.  Here                            */
/*   cluster      lo = (min, bits(node id))   hi = (max, bits(leaf count))   (the w of a box's two    */
/*                float4 was unused): no id list travels through the rounds, the merges gather nothing */
/*   binary node  ONE 64-byte line holding what the collapse wants when it OPENS the node: the boxes, */
/*                ids and leaf counts of its two CHILDREN                                              */
/*                  [0] = (min of child 0, bits(child 0))   [1] = (max of child 0, bits(child 1))      */
/*                  [2] = (min of child 1, bits(count 0))   [3] = (max of child 1, bits(count 1))      */
/* A wide node costs one line for its source and one per node it opens (at most two more) instead of  */
/* six 32-byte candidates and six counts, and leaves need no records at all: their boxes sit in their  */
/* parents' lines.  The collapse is bound by random lines from HBM (~2 TB/s of them), so lines count. */

static constexpr int kRec = 4;
__device__ __forceinline__ float4 with_w(float4 v, int32_t w) { v.w = __int_as_float(w); return v; }

__device__ __forceinline__ void node_record(float4* __restrict__ rec, int32_t id, float4 alo, float4 ahi, float4 blo, float4 bhi)
{
    /* a, b: the merging clusters' records (lo.w = id, hi.w = leaf count) */
    float4* r = rec + kRec * (size_t)id;
    r[0] = alo;                                                 /* min 0 | child 0 */
    r[1] = with_w(ahi, __float_as_int(blo.w));                  /* max 0 | child 1 */
    r[2] = with_w(blo, __float_as_int(ahi.w));                  /* min 1 | count 0 */
    r[3] = bhi;                                                 /* max 1 | count 1 */
}

__global__ __launch_bounds__(256) void k_ploc_init_p(int32_t n, const uint32_t* __restrict__ vals,
    const float4* __restrict__ triLo, const float4* __restrict__ triHi,
    float4* __restrict__ cLo, float4* __restrict__ cHi, PlocState* __restrict__ state)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { state->m = n; state->mPrev = n; state->nodeCounter = 0; state->rootId = n - 1; }
    if (i >= n) return;
    const uint32_t t = vals[i];
    /* a leaf's id names its TRIANGLE, not its Morton place (nothing asks for the place: a leaf's box travels in its parent's line),
     * so the collapse writes the triangle order without a look-up through vals[] */
    cLo[i] = with_w(triLo[t], (n - 1) + (int32_t)t); cHi[i] = with_w(triHi[t], 1);
}

__global__ __launch_bounds__(kPlocThreads) void k_ploc_scatter_p(const PlocState* __restrict__ state, const uint8_t* __restrict__ keep,
    const int32_t* __restrict__ nn, const int32_t* __restrict__ blockOffset, const int32_t* __restrict__ mergeOffset,
    const float4* __restrict__ cLoIn, const float4* __restrict__ cHiIn, float4* __restrict__ cLoOut, float4* __restrict__ cHiOut,
    float4* __restrict__ rec)
{
    const int32_t m = state->mPrev;
    const int32_t base = blockIdx.x * kPlocThreads;
    if (base >= m) return;
    const int32_t i = base + (int32_t)threadIdx.x;
    const uint8_t f = i < m ? keep[i] : 0;
    const bool kept = (f & PLOC_KEPT) != 0, merges = (f & PLOC_MERGES) != 0;
    __shared__ int32_t sCnt[2][kPlocThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t km = __ballot(kept), mm = __ballot(merges);
    if (lane == 0) { sCnt[0][wave] = (int32_t)__popcll(km); sCnt[1][wave] = (int32_t)__popcll(mm); }
    __syncthreads();
    int32_t off = blockOffset[blockIdx.x], ids = mergeOffset[blockIdx.x];
    for (int w = 0; w < wave; ++w) { off += sCnt[0][w]; ids += sCnt[1][w]; }
    if (kept)
    {
        const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int32_t dst = off + (int32_t)__popcll(km & lt);
        float4 lo = cLoIn[i], hi = cHiIn[i];
        if (merges)
        {
            const int32_t j = nn[i];
            const float4 blo = cLoIn[j], bhi = cHiIn[j];
            const int32_t leaves = __float_as_int(hi.w) + __float_as_int(bhi.w);
            const int32_t id = ids + (int32_t)__popcll(mm & lt);
            node_record(rec, id, lo, hi, blo, bhi);
            lo = make_float4(fminf(lo.x, blo.x), fminf(lo.y, blo.y), fminf(lo.z, blo.z), __int_as_float(id));
            hi = make_float4(fmaxf(hi.x, bhi.x), fmaxf(hi.y, bhi.y), fmaxf(hi.z, bhi.z), __int_as_float(leaves));
        }
        cLoOut[dst] = lo; cHiOut[dst] = hi;
    }
}

__global__ __launch_bounds__(kPlocFinish) void k_ploc_finish_p(PlocState* __restrict__ state,
    const float4* __restrict__ cLo, const float4* __restrict__ cHi, float4* __restrict__ rec)
{
    __shared__ float4 sLo[2][kPlocFinish], sHi[2][kPlocFinish];
    __shared__ int32_t sNn[kPlocFinish], sPos[kPlocFinish / 64 + 1], sM;
    const int t = threadIdx.x;
    int32_t m = state->m;
    if (m > kPlocFinish) return;                         /* the host only launches this once the list is short */
    if (t < m) { sLo[0][t] = cLo[t]; sHi[0][t] = cHi[t]; }
    __syncthreads();
    int cur = 0;
    while (m > 1)
    {
        if (t < m)
        {
            float best = __builtin_inff(); int32_t bestJ = -1;
            for (int d = -kPlocRadius; d <= kPlocRadius; ++d)
            {
                const int32_t j = t + d;
                if (d == 0 || j < 0 || j >= m) continue;
                const float a = ploc_area(sLo[cur][t], sHi[cur][t], sLo[cur][j], sHi[cur][j]);
                if (ploc_better(a, best, t, j, bestJ)) { best = a; bestJ = j; }
            }
            sNn[t] = bestJ;
        }
        __syncthreads();
        bool kept = false;
        float4 lo = make_float4(0, 0, 0, 0), hi = lo;
        if (t < m)
        {
            const int32_t j = sNn[t];
            const bool mutual = j >= 0 && sNn[j] == t;
            kept = !(mutual && j < t);
            lo = sLo[cur][t]; hi = sHi[cur][t];
            if (mutual && t < j)
            {
                const float4 blo = sLo[cur][j], bhi = sHi[cur][j];
                const int32_t leaves = __float_as_int(hi.w) + __float_as_int(bhi.w);
                const int32_t id = atomicAdd(&state->nodeCounter, 1);
                node_record(rec, id, lo, hi, blo, bhi);
                lo = make_float4(fminf(lo.x, blo.x), fminf(lo.y, blo.y), fminf(lo.z, blo.z), __int_as_float(id));
                hi = make_float4(fmaxf(hi.x, bhi.x), fmaxf(hi.y, bhi.y), fmaxf(hi.z, bhi.z), __int_as_float(leaves));
            }
        }
        {
            const uint64_t km = __ballot(kept);
            const int lane = t & 63, wave = t >> 6;
            if (lane == 0) sPos[wave] = (int32_t)__popcll(km);
            __syncthreads();
            int32_t before = 0, all = 0;
            for (int w = 0; w < kPlocFinish / 64; ++w) { const int32_t v = sPos[w]; if (w < wave) before += v; all += v; }
            if (t == 0) sM = all;
            if (kept)
            {
                const int32_t dst = before + (int32_t)__popcll(km & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                sLo[cur ^ 1][dst] = lo; sHi[cur ^ 1][dst] = hi;
            }
        }
        __syncthreads();
        m = sM;
        cur ^= 1;
        __syncthreads();
    }
    if (t == 0) { state->rootId = __float_as_int(sLo[cur][0].w); state->m = 1; }
}

struct CandP { int32_t id, count, first; float lo[3], hi[3]; float area; };

__device__ __forceinline__ void cand_box(CandP& k, float4 lo, float4 hi)
{
    k.lo[0] = lo.x; k.lo[1] = lo.y; k.lo[2] = lo.z; k.hi[0] = hi.x; k.hi[1] = hi.y; k.hi[2] = hi.z;
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    k.area = dx * dy + dy * dz + dz * dx;
}

/* opening binary node c, whose leaves start at `first`: its two children as candidates, from its one line */
__device__ __forceinline__ void open_node(const float4* __restrict__ rec, int32_t c, int32_t first, CandP& a, CandP& b)
{
    const float4* r = rec + kRec * (size_t)c;
    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    a.id = __float_as_int(r0.w); b.id = __float_as_int(r1.w);
    a.count = __float_as_int(r2.w); b.count = __float_as_int(r3.w);
    a.first = first; b.first = first + a.count;
    cand_box(a, r0, r1); cand_box(b, r2, r3);
}

/* the packed collapse counts the nodes of each level in levelCount[level] (a level's kernel appends to the next level's count), so
 * that a level's range is a prefix sum every block forms for itself and no copy sits between two levels' launches */
__global__ void k_collapse_begin_p(int32_t* levelCount, int32_t* wideSrc, int32_t* wideFirst, const int32_t* rootId)
{
    if (blockIdx.x != 0) return;
    for (int l = threadIdx.x; l < kMaxWideLevels + 4; l += blockDim.x) levelCount[l] = l == 0 ? 1 : 0;
    if (threadIdx.x == 0) { wideSrc[0] = *rootId; wideFirst[0] = 0; }
}

__global__ __launch_bounds__(256) void k_collapse_level_p(int32_t n, const uint32_t* __restrict__ vals, const float4* __restrict__ rec,
    int32_t* __restrict__ wideSrc, int32_t* __restrict__ wideFirst,
    uint32_t* __restrict__ valsOut, int32_t level, int32_t* __restrict__ levelCount, float4* __restrict__ wide)
{
    int32_t begin = 0;
    for (int l = 0; l < level; ++l) begin += levelCount[l];
    const int32_t end = begin + levelCount[level];
    int32_t* nextCount = levelCount + level + 1;
    __shared__ int32_t sTot[4], sBase;
    for (int32_t chunk = (int32_t)blockIdx.x; begin + chunk * (int32_t)blockDim.x < end; chunk += (int32_t)gridDim.x)
    {
        const int32_t w = begin + chunk * (int32_t)blockDim.x + (int32_t)threadIdx.x;
        const bool active = w < end;
        CandP cd[4]; int ns = 0;
        if (active)
        {
            open_node(rec, wideSrc[w], wideFirst[w], cd[0], cd[1]);
            ns = 2;
            /* pass 0 opens inner subtrees (count > PT_LEAF_MAX), pass 1 spends leftover slots on splitting leaves */
            for (int pass = 0; pass < 2; ++pass)
            {
                while (ns < 4)
                {
                    int best = -1; float bestArea = -1.0f;
                    for (int k = 0; k < ns; ++k)
                    {
                        const bool open = pass == 0 ? (cd[k].count > PT_LEAF_MAX) : (cd[k].count > 1);
                        if (open && cd[k].area > bestArea) { best = k; bestArea = cd[k].area; }
                    }
                    if (best < 0) break;
                    CandP a, b;
                    open_node(rec, cd[best].id, cd[best].first, a, b);
                    cd[best] = a; cd[ns] = b; ++ns;
                }
            }
        }
        /* allocate the wide children of the whole block with one atomic (a single counter sustains ~90 per microsecond) */
        int need = 0;
        for (int k = 0; k < ns; ++k) need += (cd[k].count > PT_LEAF_MAX) ? 1 : 0;
        int incl = need;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += v; }
        const int wave = threadIdx.x >> 6;
        __syncthreads();                                        /* the chunk before this one has read sTot / sBase */
        if ((threadIdx.x & 63) == 63) sTot[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0)
        {
            const int total = sTot[0] + sTot[1] + sTot[2] + sTot[3];
            sBase = end + (total > 0 ? atomicAdd(nextCount, total) : 0);
        }
        __syncthreads();
        int32_t base = sBase;
        for (int k = 0; k < wave; ++k) base += sTot[k];
        if (!active) continue;
        int32_t next = base + incl - need;
        float lo[3][4], hi[3][4]; int32_t child[4];
        float maxAbs = 0.0f;
        for (int k = 0; k < 4; ++k)
        {
            if (k >= ns)
            {
                for (int a = 0; a < 3; ++a) { lo[a][k] = __builtin_inff(); hi[a][k] = -__builtin_inff(); }
                child[k] = PT_BVH_EMPTY;
                continue;
            }
            for (int a = 0; a < 3; ++a) { lo[a][k] = cd[k].lo[a]; hi[a][k] = cd[k].hi[a]; maxAbs = fmaxf(maxAbs, fmaxf(fabsf(cd[k].lo[a]), fabsf(cd[k].hi[a]))); }
            if (cd[k].count > PT_LEAF_MAX)
            {
                child[k] = next; wideSrc[next] = cd[k].id; wideFirst[next] = cd[k].first; ++next;
            }
            else
            {
                /* a leaf: its triangles go to their places in the tree order, left to right (an inner id here has at most
                 * PT_LEAF_MAX leaves under it: their ids come from its line and its descendants') */
                child[k] = ~((cd[k].first << 3) | (cd[k].count - 1));
                int32_t todo[2 * PT_LEAF_MAX]; int top = 0, pos = cd[k].first;
                todo[top++] = cd[k].id;
                while (top > 0)
                {
                    const int32_t x = todo[--top];
                    if (x >= n - 1) valsOut[pos++] = (uint32_t)(x - (n - 1));
                    else { const float4* r = rec + kRec * (size_t)x; todo[top++] = __float_as_int(r[1].w); todo[top++] = __float_as_int(r[0].w); }
                }
            }
        }
        float4* q = wide + (size_t)w * 8;
        q[0] = make_float4(lo[0][0], lo[0][1], lo[0][2], lo[0][3]);
        q[1] = make_float4(lo[1][0], lo[1][1], lo[1][2], lo[1][3]);
        q[2] = make_float4(lo[2][0], lo[2][1], lo[2][2], lo[2][3]);
        q[3] = make_float4(hi[0][0], hi[0][1], hi[0][2], hi[0][3]);
        q[4] = make_float4(hi[1][0], hi[1][1], hi[1][2], hi[1][3]);
        q[5] = make_float4(hi[2][0], hi[2][1], hi[2][2], hi[2][3]);
        q[6] = make_float4(__int_as_float(child[0]), __int_as_float(child[1]), __int_as_float(child[2]), __int_as_float(child[3]));
        /* node 0's spare quad: the largest |coordinate| of the scene (traverse.h pads every box by an epsilon proportional to it);
         * the root's slots together are the scene's box */
        q[7] = make_float4(w == 0 ? maxAbs : 0.0f, 0.0f, 0.0f, 0.0f);
    }
}

/* scenes of at most PT_LEAF_MAX triangles: one node, one leaf */
__global__ void k_emit_single(int32_t n, const float4* __restrict__ triLo, const float4* __restrict__ triHi, float4* __restrict__ wide)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() };
    float hi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
    for (int32_t i = 0; i < n; ++i)
    {
        lo[0] = fminf(lo[0], triLo[i].x); lo[1] = fminf(lo[1], triLo[i].y); lo[2] = fminf(lo[2], triLo[i].z);
        hi[0] = fmaxf(hi[0], triHi[i].x); hi[1] = fmaxf(hi[1], triHi[i].y); hi[2] = fmaxf(hi[2], triHi[i].z);
    }
    const float inf = __builtin_inff();
    const int32_t leaf = n > 0 ? ~((0 << 3) | (n - 1)) : PT_BVH_EMPTY;
    wide[0] = make_float4(lo[0], inf, inf, inf);
    wide[1] = make_float4(lo[1], inf, inf, inf);
    wide[2] = make_float4(lo[2], inf, inf, inf);
    wide[3] = make_float4(hi[0], -inf, -inf, -inf);
    wide[4] = make_float4(hi[1], -inf, -inf, -inf);
    wide[5] = make_float4(hi[2], -inf, -inf, -inf);
    wide[6] = make_float4(__int_as_float(leaf), __int_as_float(PT_BVH_EMPTY), __int_as_float(PT_BVH_EMPTY), __int_as_float(PT_BVH_EMPTY));
    const float maxAbs = n > 0 ? fmaxf(fmaxf(fmaxf(fabsf(lo[0]), fabsf(lo[1])), fabsf(lo[2])), fmaxf(fmaxf(fabsf(hi[0]), fabsf(hi[1])), fabsf(hi[2]))) : 0.0f;
    wide[7] = make_float4(maxAbs, 0.0f, 0.0f, 0.0f);
}

__global__ __launch_bounds__(256) void k_emit_tris(const v4* __restrict__ positions, const uint32_t* __restrict__ vals,
    int32_t n, float4* __restrict__ tris)
{
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t t = vals ? vals[j] : (uint32_t)j;
    const v4 p0 = positions[t * 3 + 0], p1 = positions[t * 3 + 1], p2 = positions[t * 3 + 2];
    tris[(size_t)j * 3 + 0] = make_float4(p0.x, p0.y, p0.z, __uint_as_float(t));
    tris[(size_t)j * 3 + 1] = make_float4(p1.x, p1.y, p1.z, 0.0f);
    tris[(size_t)j * 3 + 2] = make_float4(p2.x, p2.y, p2.z, 0.0f);
}

/*
 * 128-byte fp32 node -> 64-byte compressed node (layout in pt_device.h).  Per axis the grid step is the smallest
 * power of two with 255 steps >= the extent of the union box; planes are rounded outwards in double precision
 * (differences of floats and power-of-two scalings are exact there), so the compressed box contains the fp32 one.
 */
__device__ __forceinline__ float quant_step(float extent)
{
    const float x = extent * (1.0f / 255.0f) * 1.000001f;
    if (!(x > 1e-30f)) return __uint_as_float(37u << 23);      /* flat node: a tiny step (2^-90) covers it with q = 0 or 1 */
    const uint32_t b = __float_as_uint(x);
    uint32_t e = b >> 23;                                    /* x > 0, finite */
    if (b & 0x7fffffu) ++e;
    return __uint_as_float(e << 23);
}

__global__ __launch_bounds__(256) void k_compress_nodes(const float4* __restrict__ wide, int32_t nodeCount, uint4* __restrict__ cnodes)
{
    const int32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nodeCount) return;
    const float4* q = wide + (size_t)w * 8;
    float lo[3][4], hi[3][4]; int32_t child[4];
    for (int a = 0; a < 3; ++a)
    {
        const float4 l = q[a], h = q[3 + a];
        lo[a][0] = l.x; lo[a][1] = l.y; lo[a][2] = l.z; lo[a][3] = l.w;
        hi[a][0] = h.x; hi[a][1] = h.y; hi[a][2] = h.z; hi[a][3] = h.w;
    }
    const float4 ch = q[6];
    child[0] = __float_as_int(ch.x); child[1] = __float_as_int(ch.y); child[2] = __float_as_int(ch.z); child[3] = __float_as_int(ch.w);
    float org[3], step[3]; uint32_t qlo[3], qhi[3];
    for (int a = 0; a < 3; ++a)
    {
        float mn = __builtin_inff(), mx = -__builtin_inff();
        for (int k = 0; k < 4; ++k) if (child[k] != PT_BVH_EMPTY) { mn = fminf(mn, lo[a][k]); mx = fmaxf(mx, hi[a][k]); }
        if (!(mn <= mx)) { mn = 0.0f; mx = 0.0f; }          /* no children at all (empty scene) */
        org[a] = mn;
        step[a] = quant_step(mx - mn);
        uint32_t pl = 0, ph = 0;
        for (int k = 0; k < 4; ++k)
        {
            uint32_t a0 = 255u, a1 = 0u;                      /* empty slot: inverted */
            if (child[k] != PT_BVH_EMPTY)
            {
                const double inv = 1.0 / (double)step[a];
                double f0 = floor(((double)lo[a][k] - (double)mn) * inv), f1 = ceil(((double)hi[a][k] - (double)mn) * inv);
                f0 = f0 < 0.0 ? 0.0 : (f0 > 255.0 ? 255.0 : f0);
                f1 = f1 < 0.0 ? 0.0 : (f1 > 255.0 ? 255.0 : f1);
                a0 = (uint32_t)f0; a1 = (uint32_t)f1;
            }
            pl |= a0 << (8 * k); ph |= a1 << (8 * k);
        }
        qlo[a] = pl; qhi[a] = ph;
    }
    uint4* c = cnodes + (size_t)w * 4;
    c[0] = make_uint4(__float_as_uint(org[0]), __float_as_uint(org[1]), __float_as_uint(org[2]), __float_as_uint(step[0]));
    c[1] = make_uint4(qlo[0], qlo[1], qlo[2], qhi[0]);
    c[2] = make_uint4(qhi[1], qhi[2], __float_as_uint(step[1]), __float_as_uint(step[2]));
    c[3] = make_uint4((uint32_t)child[0], (uint32_t)child[1], (uint32_t)child[2], (uint32_t)child[3]);
}

/* ---------------------------------------------------------------------------------------------- */

void BvhStorage::release()
{
    hipFree(nodes); hipFree(cnodes); hipFree(tris); hipFree(scratch);
    if (progress) hipHostFree(progress);
    nodes = nullptr; cnodes = nullptr; tris = nullptr; scratch = nullptr; progress = nullptr; scratchBytes = 0; capacityTris = 0;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

/* which builder makes the binary tree: PLOC (default) or the Karras LBVH (PIMGPU_BVH_BUILDER=lbvh: about half the build
 * time, about a fifth more nodes per ray) */
static bool bvh_use_ploc()
{
    const char* e = getenv("PIMGPU_BVH_BUILDER");
    return !(e && e[0] == 'l');
}

int bvh_build(BvhStorage* b, const v4* positionsDev, int32_t triCount, hipStream_t stream)
{
    const int32_t n = triCount;
    const int32_t cap = n > 0 ? n : 1;
    const int32_t numBlocks = (cap + kSortTile - 1) / kSortTile;
    /* scratch layout */
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    const size_t oTriLo = take(sizeof(float4) * cap), oTriHi = take(sizeof(float4) * cap);
    const size_t oKeysA = take(8 * (size_t)cap), oKeysB = take(8 * (size_t)cap);
    const size_t oValsA = take(4 * (size_t)cap), oValsB = take(4 * (size_t)cap);
    const size_t oCounts = take(4 * (size_t)256 * numBlocks), oTotals = take(4 * 256);
    const size_t oChild0 = take(4 * (size_t)cap), oChild1 = take(4 * (size_t)cap);
    const size_t oParent = take(4 * (size_t)(2 * cap)), oFirst = take(4 * (size_t)cap), oLast = take(4 * (size_t)cap);
    const size_t oNodeLo = take(sizeof(float4) * cap), oNodeHi = take(sizeof(float4) * cap);
    const size_t oFlags = take(4 * (size_t)cap), oWideSrc = take(4 * (size_t)cap), oBounds = take(64), oSnap = take(4 * (kMaxWideLevels + 8));
    /* PLOC: two cluster lists (id + box), neighbour, survivor flags, per-block counts, leaf counts, new order */
    const int32_t plocBlocks = (cap + kPlocThreads - 1) / kPlocThreads;
    const size_t oCidA = take(4 * (size_t)cap), oCidB = take(4 * (size_t)cap);
    const size_t oCLoA = take(sizeof(float4) * cap), oCHiA = take(sizeof(float4) * cap), oCLoB = take(sizeof(float4) * cap), oCHiB = take(sizeof(float4) * cap);
    const size_t oNn = take(4 * (size_t)cap), oKeep = take((size_t)cap), oBlockCount = take(4 * (size_t)plocBlocks + 4), oMergeCount = take(4 * (size_t)plocBlocks + 4);
    const size_t oCount = take(4 * (size_t)cap), oNewPos = take(4 * (size_t)cap), oVals2 = take(4 * (size_t)cap), oPloc = take(64);
    const size_t oRec = take(sizeof(float4) * 4 * (size_t)cap);        /* packed records: one 64-byte line per binary node (its children's boxes, ids, leaf counts) */
    const size_t need = off;
    if (b->capacityTris < cap || b->scratchBytes < need)
    {
        b->release();
        if (hipMalloc(&b->nodes, sizeof(float4) * 8 * (size_t)cap) != hipSuccess) return PIMGPU_E_NOMEM;
        if (hipMalloc(&b->cnodes, sizeof(uint4) * 4 * (size_t)cap) != hipSuccess) return PIMGPU_E_NOMEM;
        if (hipMalloc(&b->tris, sizeof(float4) * 3 * ((size_t)cap + 1)) != hipSuccess) return PIMGPU_E_NOMEM;     /* + 1: traverse.h reads 64 B of a one-triangle leaf */
        if (hipMalloc(&b->scratch, need) != hipSuccess) return PIMGPU_E_NOMEM;
        if (hipHostMalloc((void**)&b->progress, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return PIMGPU_E_NOMEM;
        b->capacityTris = cap; b->scratchBytes = need;
    }
    char* s = (char*)b->scratch;
    float4* triLo = (float4*)(s + oTriLo); float4* triHi = (float4*)(s + oTriHi);
    uint64_t* keysA = (uint64_t*)(s + oKeysA); uint64_t* keysB = (uint64_t*)(s + oKeysB);
    uint32_t* valsA = (uint32_t*)(s + oValsA); uint32_t* valsB = (uint32_t*)(s + oValsB);
    uint32_t* counts = (uint32_t*)(s + oCounts); uint32_t* totals = (uint32_t*)(s + oTotals);
    int32_t* child0 = (int32_t*)(s + oChild0); int32_t* child1 = (int32_t*)(s + oChild1);
    int32_t* parent = (int32_t*)(s + oParent); int32_t* rangeFirst = (int32_t*)(s + oFirst); int32_t* rangeLast = (int32_t*)(s + oLast);
    float4* nodeLo = (float4*)(s + oNodeLo); float4* nodeHi = (float4*)(s + oNodeHi);
    uint32_t* flags = (uint32_t*)(s + oFlags); int32_t* wideSrc = (int32_t*)(s + oWideSrc);
    uint32_t* bounds = (uint32_t*)(s + oBounds);
    int32_t* snap = (int32_t*)(s + oSnap); int32_t* nodeCounter = snap + kMaxWideLevels + 4;
    int32_t* cidA = (int32_t*)(s + oCidA); int32_t* cidB = (int32_t*)(s + oCidB);
    float4* cLoA = (float4*)(s + oCLoA); float4* cHiA = (float4*)(s + oCHiA); float4* cLoB = (float4*)(s + oCLoB); float4* cHiB = (float4*)(s + oCHiB);
    int32_t* nn = (int32_t*)(s + oNn); uint8_t* keep = (uint8_t*)(s + oKeep); int32_t* blockCount = (int32_t*)(s + oBlockCount); int32_t* mergeCount = (int32_t*)(s + oMergeCount);
    int32_t* count = (int32_t*)(s + oCount); int32_t* newPos = (int32_t*)(s + oNewPos); uint32_t* vals2 = (uint32_t*)(s + oVals2);
    PlocState* ploc = (PlocState*)(s + oPloc);
    float4* rec = (float4*)(s + oRec);

    b->triCount = n;
    b->nodeCount = 1;
    b->depth = 0;
    const int grid = (cap + 255) / 256;
    k_bounds_init<<<1, 64, 0, stream>>>(bounds);
    if (n > 0) k_tri_bounds<<<std::min(grid, 2048), 256, 0, stream>>>(positionsDev, n, triLo, triHi, bounds);
    if (n <= PT_LEAF_MAX)
    {
        k_emit_single<<<1, 64, 0, stream>>>(n, triLo, triHi, b->nodes);
        k_compress_nodes<<<1, 256, 0, stream>>>(b->nodes, 1, b->cnodes);
        if (n > 0) k_emit_tris<<<grid, 256, 0, stream>>>(positionsDev, nullptr, n, b->tris);
        return hipGetLastError() == hipSuccess ? PIMGPU_OK : PIMGPU_E_DEVICE;
    }
    /* PLOC: the index rides in the key (LBVH keeps separate values: k_karras reads the codes themselves) */
    const bool packIndex = kCanPackIndex && bvh_use_ploc() && (int64_t)n <= (1ll << kIndexBits) && !getenv("PIMGPU_BVH_SORT_VALUES");
    k_morton<<<grid, 256, 0, stream>>>(triLo, triHi, n, bounds, keysA, packIndex ? nullptr : valsA);
    uint64_t* kin = keysA; uint64_t* kout = keysB; uint32_t* vin = valsA; uint32_t* vout = valsB;
    const int shift0 = packIndex ? kIndexBits : 0;
    for (int shift = shift0; shift < shift0 + 8 * kSortPasses; shift += 8)
    {
        k_sort_hist<<<numBlocks, kSortThreads, 0, stream>>>(kin, n, shift, counts, numBlocks);
        k_sort_scan<<<256, 256, 0, stream>>>(counts, numBlocks, totals);
        k_sort_scatter<<<numBlocks, kSortThreads, 0, stream>>>(kin, packIndex ? nullptr : vin, kout, vout, n, shift, counts, totals, numBlocks);
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    if (packIndex) k_unpack_vals<<<grid, 256, 0, stream>>>(kin, n, vin);
    const uint32_t* order = vin;                    /* triangle of each leaf position */
    /* PLOC: the collapse hands the leaf order down (k_collapse_level<true>); PIMGPU_BVH_ORDER_WALKS=1 keeps the walks towards the root */
    const bool orderInCollapse = bvh_use_ploc() && !getenv("PIMGPU_BVH_ORDER_WALKS");
    /* ... and runs on packed records (k_*_p); PIMGPU_BVH_PACKED=0 keeps the arrays-per-field kernels */
    const char* packedEnv = getenv("PIMGPU_BVH_PACKED");
    const bool packed = orderInCollapse && !(packedEnv && packedEnv[0] == '0');
    int32_t* wideFirst = newPos;                    /* a wide node's first leaf place: newPos is free when the walks are not run (else unused) */
    const int32_t* rootId = nullptr;                /* device: root of the binary tree (LBVH: node 0) */
    if (!bvh_use_ploc())
    {
        k_karras<<<grid, 256, 0, stream>>>(kin, n, child0, child1, parent, rangeFirst, rangeLast);
        hipMemsetAsync(flags, 0, 4 * (size_t)n, stream);
        k_refit<<<grid, 256, 0, stream>>>(n, vin, triLo, triHi, child0, child1, parent, nodeLo, nodeHi, flags);
    }
    else if (packed)
    {
        k_ploc_init_p<<<grid, 256, 0, stream>>>(n, vin, triLo, triHi, cLoA, cHiA, ploc);
        const float4* loIn = cLoA; const float4* hiIn = cHiA;
        float4* loOut = cLoB; float4* hiOut = cHiB;
        /* The list length lives on the device.  Each round's scan reports it to pinned host memory, and the host stays at most
         * kPlocLag rounds ahead of the last report: grids are sized by a length a few rounds old (blocks past the end return at once),
         * the stream is never drained, and the few rounds launched after the list has become short enough for k_ploc_finish_p
         * are ordinary rounds on a short list. */
        constexpr int kPlocLag = 4;
        volatile int32_t* progress = b->progress;
        progress[0] = n; progress[1] = 0;
        int32_t mHost = n;
        int rounds = 0;
        for (;;)
        {
            const int32_t want = rounds - kPlocLag + 1;
            if (want > 0 && __atomic_load_n(&progress[1], __ATOMIC_ACQUIRE) < want)
            {
                for (uint32_t spins = 1; __atomic_load_n(&progress[1], __ATOMIC_ACQUIRE) < want; ++spins)
                {
                    if ((spins & 0xFFFF) != 0) continue;
                    const hipError_t q = hipStreamQuery(stream);        /* a stream that has failed or run dry never reports */
                    if (q == hipErrorNotReady) continue;
                    if (q != hipSuccess) return PIMGPU_E_DEVICE;
                    if (__atomic_load_n(&progress[1], __ATOMIC_ACQUIRE) < want) { set_error("BVH build: clustering round %d never reported", want); return PIMGPU_E_STATE; }
                }
            }
            if (__atomic_load_n(&progress[1], __ATOMIC_ACQUIRE) > 0) mHost = std::min(mHost, (int32_t)progress[0]);
            if (mHost <= kPlocFinish) break;
            if (++rounds > 4096) { set_error("BVH build: clustering does not converge"); return PIMGPU_E_STATE; }
            const int nb = (mHost + kPlocThreads - 1) / kPlocThreads;
            k_ploc_nn_merge<<<nb, kPlocThreads, 0, stream>>>(ploc, loIn, hiIn, 1, nn, keep, blockCount, mergeCount);
            k_ploc_scan<<<1, kScanThreads, 0, stream>>>(ploc, blockCount, mergeCount, nb, b->progress, rounds);
            k_ploc_scatter_p<<<nb, kPlocThreads, 0, stream>>>(ploc, keep, nn, blockCount, mergeCount, loIn, hiIn, loOut, hiOut, rec);
            { const float4* tl = loIn; const float4* th = hiIn; loIn = loOut; hiIn = hiOut; loOut = (float4*)tl; hiOut = (float4*)th; }
        }
        k_ploc_finish_p<<<1, kPlocFinish, 0, stream>>>(ploc, loIn, hiIn, rec);
        rootId = &ploc->rootId;
    }
    else
    {
        k_ploc_init<<<grid, 256, 0, stream>>>(n, vin, triLo, triHi, cidA, cLoA, cHiA, ploc);
        int32_t* cidIn = cidA; int32_t* cidOut = cidB;
        float4* loIn = cLoA; float4* hiIn = cHiA; float4* loOut = cLoB; float4* hiOut = cHiB;
        int32_t mHost = n;                          /* upper bound of the device-side list length */
        int rounds = 0;
        while (mHost > kPlocFinish)
        {
            if (++rounds > 4096) { set_error("BVH build: clustering does not converge"); return PIMGPU_E_STATE; }
            const int nb = (mHost + kPlocThreads - 1) / kPlocThreads;
            k_ploc_nn<<<nb, kPlocThreads, 0, stream>>>(ploc, loIn, hiIn, nn);
            k_ploc_merge<<<nb, kPlocThreads, 0, stream>>>(ploc, nn, keep, blockCount, mergeCount);
            k_ploc_scan<<<1, kScanThreads, 0, stream>>>(ploc, blockCount, mergeCount, nb);
            k_ploc_scatter<<<nb, kPlocThreads, 0, stream>>>(n, ploc, keep, nn, blockCount, mergeCount, cidIn, loIn, hiIn, cidOut, loOut, hiOut,
                child0, child1, orderInCollapse ? nullptr : parent, nodeLo, nodeHi, count);
            std::swap(cidIn, cidOut); std::swap(loIn, loOut); std::swap(hiIn, hiOut);
            if ((rounds & 3) == 0)
            {
                /* the list length lives on the device: look at it every fourth round (grids in between use the last known length) */
                PlocState hs;
                if (hipMemcpyAsync(&hs, ploc, sizeof(hs), hipMemcpyDeviceToHost, stream) != hipSuccess) return PIMGPU_E_DEVICE;
                if (hipStreamSynchronize(stream) != hipSuccess) return PIMGPU_E_DEVICE;
                mHost = hs.m;
            }
        }
        k_ploc_finish<<<1, kPlocFinish, 0, stream>>>(n, ploc, cidIn, loIn, hiIn, child0, child1, parent, nodeLo, nodeHi, count);
        if (!orderInCollapse)
        {
            for (int phase = 0; phase < 3; ++phase)
                k_ploc_order<<<grid, 256, 0, stream>>>(phase, n, vin, child0, child1, parent, count, rangeFirst, rangeLast, newPos, vals2);
            k_ploc_remap<<<grid, 256, 0, stream>>>(n, newPos, child0, child1);
            order = vals2;
        }
        rootId = &ploc->rootId;
    }
    if (!orderInCollapse) k_emit_tris<<<grid, 256, 0, stream>>>(positionsDev, order, n, b->tris);
    /* collapse, level by level; the level count is only known on the device, so levels go out in batches (the first about
     * log2 n long, then 8 at a time) and the level table is read back after each batch: one short sync per batch, usually one */
    if (packed) k_collapse_begin_p<<<1, 64, 0, stream>>>(snap, wideSrc, wideFirst, rootId);          /* snap[]: nodes per level here */
    else k_collapse_begin<<<1, 64, 0, stream>>>(snap, nodeCounter, wideSrc, wideFirst, rootId);
    int32_t hostSnap[kMaxWideLevels + 2];
    int level = 0;
    bool done = false;
    while (!done)
    {
        if (level >= kMaxWideLevels) { set_error("BVH collapse: more than %d levels", kMaxWideLevels); return PIMGPU_E_STATE; }
        /* (the first batch is as long as trees of n leaves usually are deep -- about log2 n levels: a launch for a level that
         * turns out empty costs a tenth of a drained stream) */
        int firstBatch = 8;
        while (firstBatch < 40 && ((int64_t)1 << firstBatch) < (int64_t)n) ++firstBatch;
        const int batchEnd = std::min(level + (level == 0 ? firstBatch : 8), kMaxWideLevels);
        for (; level < batchEnd; ++level)
        {
            /* level L holds at most 4^L nodes and never more than the binary tree has inner nodes */
            int64_t most = level < 15 ? ((int64_t)1 << (2 * level)) : (int64_t)n;
            most = std::min<int64_t>(most, n);
            const int collapseGrid = (int)std::min<int64_t>((most + 255) / 256, kCollapseBlocks);
            if (packed)
            {
                k_collapse_level_p<<<collapseGrid, 256, 0, stream>>>(n, order, rec, wideSrc, wideFirst, vals2, level, snap, b->nodes);
                continue;
            }
            if (orderInCollapse)
                k_collapse_level<true><<<collapseGrid, 256, 0, stream>>>(n, order, triLo, triHi, nodeLo, nodeHi, child0, child1,
                    nullptr, count, wideSrc, wideFirst, vals2, snap + level, nodeCounter, b->nodes);
            else
                k_collapse_level<false><<<collapseGrid, 256, 0, stream>>>(n, order, triLo, triHi, nodeLo, nodeHi, child0, child1,
                    rangeFirst, rangeLast, wideSrc, wideFirst, nullptr, snap + level, nodeCounter, b->nodes);
            /* (a last-block-done counter instead of this 4-byte copy: every one of the up to 78 k blocks of a deep level's launch
             * would hit ONE address -- 31.7 instead of 10.2 ms per build) */
            if (hipMemcpyAsync(snap + level + 2, nodeCounter, 4, hipMemcpyDeviceToDevice, stream) != hipSuccess) return PIMGPU_E_DEVICE;
        }
        if (hipMemcpyAsync(hostSnap, snap, sizeof(int32_t) * (size_t)(level + 2), hipMemcpyDeviceToHost, stream) != hipSuccess) return PIMGPU_E_DEVICE;
        if (hipStreamSynchronize(stream) != hipSuccess) return PIMGPU_E_DEVICE;
        if (packed)
        {
            int32_t total = 0;
            for (int l = 0; l <= level && !done; ++l)
            {
                if (hostSnap[l] == 0) { done = true; b->depth = l; b->nodeCount = total; }
                total += hostSnap[l];
            }
            continue;
        }
        for (int l = 0; l <= level; ++l)
        {
            if (hostSnap[l + 1] == hostSnap[l]) { done = true; b->depth = l; b->nodeCount = hostSnap[l]; break; }
        }
    }
    /* the batched traversal kernels address a node by a 32-bit byte offset from the array's base (one shift, the base in scalar registers) */
    if ((int64_t)b->nodeCount >= (1ll << 26)) { set_error("BVH build: %d wide nodes; the traversal kernels take fewer than 2^26", b->nodeCount); return PIMGPU_E_LIMIT; }
    if (orderInCollapse) k_emit_tris<<<grid, 256, 0, stream>>>(positionsDev, vals2, n, b->tris);     /* the collapse has written the triangles' order */
    k_compress_nodes<<<(b->nodeCount + 255) / 256, 256, 0, stream>>>(b->nodes, b->nodeCount, b->cnodes);
    return hipGetLastError() == hipSuccess ? PIMGPU_OK : PIMGPU_E_DEVICE;
}

/* debugging aid (PIMGPU_VALIDATE_BVH=1): walk the tree on the host and check that every child index,
 * leaf range and box is sane and that every triangle is referenced exactly once */
int bvh_validate(const BvhStorage* b, hipStream_t stream, std::string* why)
{
    hipStreamSynchronize(stream);
    const int32_t n = b->triCount;
    const size_t nodeCap = (size_t)(n > 1 ? n - 1 : 1);
    std::vector<float4> nodes(nodeCap * 8), tris((size_t)(n > 0 ? n : 1) * 3);
    if (hipMemcpy(nodes.data(), b->nodes, sizeof(float4) * nodes.size(), hipMemcpyDeviceToHost) != hipSuccess) { *why = "copy of nodes failed"; return PIMGPU_E_DEVICE; }
    if (n > 0 && hipMemcpy(tris.data(), b->tris, sizeof(float4) * (size_t)n * 3, hipMemcpyDeviceToHost) != hipSuccess) { *why = "copy of tris failed"; return PIMGPU_E_DEVICE; }
    std::vector<uint8_t> seenTri((size_t)(n > 0 ? n : 1), 0), seenNode(nodeCap, 0), seenPrim((size_t)(n > 0 ? n : 1), 0);
    std::vector<int32_t> stack; stack.push_back(0);
    char buf[256];
    size_t visited = 0;
    while (!stack.empty())
    {
        const int32_t cur = stack.back(); stack.pop_back();
        if (cur < 0 || (size_t)cur >= nodeCap) { snprintf(buf, sizeof(buf), "node index %d out of range [0,%zu)", cur, nodeCap); *why = buf; return PIMGPU_E_STATE; }
        if (seenNode[(size_t)cur]) { snprintf(buf, sizeof(buf), "node %d reached twice", cur); *why = buf; return PIMGPU_E_STATE; }
        seenNode[(size_t)cur] = 1; ++visited;
        const float4* q = &nodes[(size_t)cur * 8];
        const float ch[4] = { q[6].x, q[6].y, q[6].z, q[6].w };
        for (int k = 0; k < 4; ++k)
        {
            int32_t c; __builtin_memcpy(&c, &ch[k], 4);
            if (c == PT_BVH_EMPTY) continue;
            if (c >= 0) { stack.push_back(c); continue; }
            const uint32_t code = (uint32_t)~c, first = code >> 3, cnt = (code & 7u) + 1u;
            if (cnt > PT_LEAF_MAX || (int64_t)first + cnt > n) { snprintf(buf, sizeof(buf), "node %d slot %d: leaf [%u,+%u) out of range (n=%d)", cur, k, first, cnt, n); *why = buf; return PIMGPU_E_STATE; }
            for (uint32_t t = first; t < first + cnt; ++t)
            {
                if (seenTri[t]) { snprintf(buf, sizeof(buf), "triangle slot %u referenced twice", t); *why = buf; return PIMGPU_E_STATE; }
                seenTri[t] = 1;
                uint32_t prim; __builtin_memcpy(&prim, &tris[(size_t)t * 3].w, 4);
                if (prim >= (uint32_t)n || seenPrim[prim]) { snprintf(buf, sizeof(buf), "triangle slot %u holds primID %u (bad or duplicate)", t, prim); *why = buf; return PIMGPU_E_STATE; }
                seenPrim[prim] = 1;
            }
        }
    }
    /* boxes: every slot's box must contain the triangles below it (checked against the leaf triangles' own vertices) */
    {
        struct Item { int32_t node; float lo[3], hi[3]; };
        std::vector<Item> st; 
        Item root; root.node = 0; for (int a = 0; a < 3; ++a) { root.lo[a] = -3.0e38f; root.hi[a] = 3.0e38f; }
        st.push_back(root);
        while (!st.empty())
        {
            const Item it = st.back(); st.pop_back();
            const float4* q = &nodes[(size_t)it.node * 8];
            const float lx[4] = { q[0].x, q[0].y, q[0].z, q[0].w }, ly[4] = { q[1].x, q[1].y, q[1].z, q[1].w }, lz[4] = { q[2].x, q[2].y, q[2].z, q[2].w };
            const float hx[4] = { q[3].x, q[3].y, q[3].z, q[3].w }, hy[4] = { q[4].x, q[4].y, q[4].z, q[4].w }, hz[4] = { q[5].x, q[5].y, q[5].z, q[5].w };
            const float ch[4] = { q[6].x, q[6].y, q[6].z, q[6].w };
            for (int k = 0; k < 4; ++k)
            {
                int32_t c; __builtin_memcpy(&c, &ch[k], 4);
                if (c == PT_BVH_EMPTY) continue;
                Item child; child.node = c;
                child.lo[0] = lx[k]; child.lo[1] = ly[k]; child.lo[2] = lz[k]; child.hi[0] = hx[k]; child.hi[1] = hy[k]; child.hi[2] = hz[k];
                for (int a = 0; a < 3; ++a)
                {
                    if (!(child.lo[a] >= it.lo[a] && child.hi[a] <= it.hi[a] && child.lo[a] <= child.hi[a]))
                    { snprintf(buf, sizeof(buf), "node %d slot %d: box [%g,%g] on axis %d is not inside its parent's [%g,%g]", it.node, k, child.lo[a], child.hi[a], a, it.lo[a], it.hi[a]); *why = buf; return PIMGPU_E_STATE; }
                }
                if (c >= 0) { st.push_back(child); continue; }
                const uint32_t code = (uint32_t)~c, first = code >> 3, cnt = (code & 7u) + 1u;
                for (uint32_t t = first; t < first + cnt; ++t)
                    for (int v = 0; v < 3; ++v)
                    {
                        const float4 p = tris[(size_t)t * 3 + v];
                        const float pp[3] = { p.x, p.y, p.z };
                        for (int a = 0; a < 3; ++a)
                            if (!(pp[a] >= child.lo[a] && pp[a] <= child.hi[a]))
                            { snprintf(buf, sizeof(buf), "node %d slot %d: triangle slot %u vertex %d (%g) outside the leaf box [%g,%g] on axis %d", it.node, k, t, v, pp[a], child.lo[a], child.hi[a], a); *why = buf; return PIMGPU_E_STATE; }
                    }
            }
        }
    }
    /* compressed copy: same children, every decoded box contains the fp32 box */
    {
        std::vector<uint4> cn((size_t)b->nodeCount * 4);
        if (hipMemcpy(cn.data(), b->cnodes, sizeof(uint4) * cn.size(), hipMemcpyDeviceToHost) != hipSuccess) { *why = "copy of compressed nodes failed"; return PIMGPU_E_DEVICE; }
        for (int32_t w = 0; w < b->nodeCount; ++w)
        {
            if (!seenNode[(size_t)w]) continue;
            const float4* q = &nodes[(size_t)w * 8];
            const uint4* c = &cn[(size_t)w * 4];
            float org[3], step[3];
            __builtin_memcpy(&org[0], &c[0].x, 4); __builtin_memcpy(&org[1], &c[0].y, 4); __builtin_memcpy(&org[2], &c[0].z, 4);
            __builtin_memcpy(&step[0], &c[0].w, 4); __builtin_memcpy(&step[1], &c[2].z, 4); __builtin_memcpy(&step[2], &c[2].w, 4);
            const uint32_t ql[3] = { c[1].x, c[1].y, c[1].z }, qh[3] = { c[1].w, c[2].x, c[2].y };
            const uint32_t cc[4] = { c[3].x, c[3].y, c[3].z, c[3].w };
            const float ch[4] = { q[6].x, q[6].y, q[6].z, q[6].w };
            for (int k = 0; k < 4; ++k)
            {
                uint32_t cf; __builtin_memcpy(&cf, &ch[k], 4);
                if (cf != cc[k]) { snprintf(buf, sizeof(buf), "compressed node %d slot %d: child %08x != %08x", w, k, cc[k], cf); *why = buf; return PIMGPU_E_STATE; }
                if ((int32_t)cf == PT_BVH_EMPTY) continue;
                for (int a = 0; a < 3; ++a)
                {
                    const float* fl = &q[a].x; const float* fh = &q[3 + a].x;
                    const double dl = (double)org[a] + (double)((ql[a] >> (8 * k)) & 255u) * (double)step[a];
                    const double dh = (double)org[a] + (double)((qh[a] >> (8 * k)) & 255u) * (double)step[a];
                    if (!(dl <= (double)fl[k] && dh >= (double)fh[k]))
                    { snprintf(buf, sizeof(buf), "compressed node %d slot %d axis %d: [%.9g,%.9g] does not contain [%.9g,%.9g]", w, k, a, dl, dh, fl[k], fh[k]); *why = buf; return PIMGPU_E_STATE; }
                }
            }
        }
    }
    for (int32_t t = 0; t < n; ++t) if (!seenTri[(size_t)t]) { snprintf(buf, sizeof(buf), "triangle slot %d is not referenced by any leaf (%zu nodes visited)", t, visited); *why = buf; return PIMGPU_E_STATE; }
    return PIMGPU_OK;
}

} /* namespace pt */
