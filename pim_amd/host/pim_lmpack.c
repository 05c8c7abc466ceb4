/*
 * pim_lmpack.c -- LmPack_Pack (rendering/lightmap.c:1047-1102, "lm.c" below) with the reference's signature: the lightmap atlas
 * packer that produces LmPack_Bake's inputs.  The data-parallel steps run on the GPU through the C-ABI, the sequential ones
 * (sequential in the reference too) are host C here:
 *
 *   chartnodes_create  lm.c:723-767   GPU   pimgpu_lightmap_chartnodes
 *   chart_group        lm.c:563-636   host  plane grouping + splitting of wide / sparse charts (below)
 *   ChartMaskFn        lm.c:527-561   GPU   pimgpu_lightmap_chart_layout + pimgpu_lightmap_chart_masks
 *   chart_sort         lm.c:648-660   host  the quicksort of common/sort.c:14-75 (its order on equal areas is part of the result)
 *   atlases_create     lm.c:845-874   GPU   pimgpu_lightmap_atlas_place: first-fit search of every chart, in order
 *   Lightmap_New x n   lm.c:1085-1088 host  pim_host.c
 *   chartnodes_assign  lm.c:876-939   host  uvs.zw / texIndices.w of the meshes (below)
 *   EmbedAttributes    lm.c:1034-1045 GPU   pimgpu_lightmap_embed
 *
 * Same departures from the reference as the checker states for its own restatement: chart_split's means come from a per-call
 * stream Pcg4(split number, nodeCount, seed, TAG) instead of the clock-seeded global Prng; a chart that cannot be
 * divided is kept whole after kSplitAttempts tries instead of being split for ever (lm.c:609-636); the atlas search is ONE
 * worker's (with several the reference's result depends on thread timing, lm.c:793); a chart that fits in none of the estimated
 * atlases opens another one instead of spinning for ever (lm.c:808-818), and one larger than an atlas is reported.
 *
 * plane_find (lm.c:341-357) is a linear scan over every chart made so far -- O(nodes x charts), hours at a million triangles.
 * Here the charts' planes are bucketed by their distance term: |w - w'| < distThresh puts a match in the same or a
 * neighbouring bucket of width distThresh, and the LOWEST matching chart of those buckets is the one the scan would have
 * found first.  Same result, O(nodes x bucket).
 */
#include "pim_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <threads.h>

#define kTagChart 0x43485254u               /* 'CHRT' */
enum { kChartSplits = 2, kSplitAttempts = 16, kSplitRounds = 1000 };
#define kTau 6.283185307179586476925286766559f

static PtDrawablesProviderFn ms_drawables;
static void* ms_drawablesUser;
void PtSys_SetDrawablesProvider(PtDrawablesProviderFn fn, void* user) { ms_drawables = fn; ms_drawablesUser = user; }

/* ---- small math, as the reference's headers evaluate it (fp32, this order) ---- */

static float f_min(float a, float b) { return a < b ? a : b; }
static float f_max(float a, float b) { return a > b ? a : b; }
static float f_sat(float x) { x = (x > 0.0f) ? x : 0.0f; x = (x < 1.0f) ? x : 1.0f; return x; }          /* scalar.h:66-72 */
static float2 f2(float x, float y) { float2 v; v.x = x; v.y = y; return v; }
static float2 f2_sub(float2 a, float2 b) { return f2(a.x - b.x, a.y - b.y); }
static float2 f2_add(float2 a, float2 b) { return f2(a.x + b.x, a.y + b.y); }
static float2 f2_muls(float2 a, float s) { return f2(a.x * s, a.y * s); }
static float f2_dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
static float2 f2_min(float2 a, float2 b) { return f2(f_min(a.x, b.x), f_min(a.y, b.y)); }
static float2 f2_max(float2 a, float2 b) { return f2(f_max(a.x, b.x), f_max(a.y, b.y)); }

typedef struct { float2 a, b, c; } Tri2D;
static Tri2D node_tri(const PimChartNode* n) { Tri2D t; t.a = f2(n->triCoord[0], n->triCoord[1]); t.b = f2(n->triCoord[2], n->triCoord[3]); t.c = f2(n->triCoord[4], n->triCoord[5]); return t; }

/* sdTriangle2D, math/sdf.h:125-156: signed distance of p to the triangle, negative inside */
static float tri_distance(Tri2D t, float2 p)
{
    const float2 ba = f2_sub(t.b, t.a), cb = f2_sub(t.c, t.b), ac = f2_sub(t.a, t.c);
    const float2 pa = f2_sub(p, t.a), pb = f2_sub(p, t.b), pc = f2_sub(p, t.c);
    const float h0 = f_sat(f2_dot(pa, ba) / f2_dot(ba, ba));
    const float h1 = f_sat(f2_dot(pb, cb) / f2_dot(cb, cb));
    const float h2 = f_sat(f2_dot(pc, ac) / f2_dot(ac, ac));
    const float2 q0 = f2_sub(pa, f2_muls(ba, h0)), q1 = f2_sub(pb, f2_muls(cb, h1)), q2 = f2_sub(pc, f2_muls(ac, h2));
    const float cr = ba.x * ac.y - ba.y * ac.x;
    const float s = (cr > 0.0f) ? 1.0f : ((cr < 0.0f) ? -1.0f : 0.0f);                                   /* f1_sign, scalar.h */
    const float2 t0 = f2(f2_dot(q0, q0), s * (pa.x * ba.y - pa.y * ba.x));
    const float2 t1 = f2(f2_dot(q1, q1), s * (pb.x * cb.y - pb.y * cb.x));
    const float2 t2 = f2(f2_dot(q2, q2), s * (pc.x * ac.y - pc.y * ac.x));
    const float2 d = f2_min(t0, f2_min(t1, t2));
    const float dySign = (d.y >= 0.0f) ? 1.0f : -1.0f;
    return -sqrtf(f_max(d.x, 1.0f / ((1ull << 46) - 1ull))) * dySign;
}

/* math/pcg.h:16-22,60-76: the generators behind Prng (common/random.h:13-35,63-67) */
static uint32_t pcg1(uint32_t v)
{
    v = v * 747796405u + 2891336453u;
    v = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    return (v >> 22) ^ v;
}
static uint32_t pcg4_x(uint32_t x, uint32_t y, uint32_t z, uint32_t w)     /* lane x of Pcg4: the only lane Prng_i32 reads and rehashes */
{
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u; z = z * 1664525u + 1013904223u; w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w;
    return x;
}

/* ---- charts: ranges of a node pool ---- */

typedef struct { int32_t first, count, attempts; } ChartRange;
typedef struct { PimChartNode* nodes; int64_t count, cap; } NodePool;

static bool pool_reserve(NodePool* pool, int64_t extra)
{
    if (pool->count + extra <= pool->cap) return true;
    int64_t cap = pool->cap ? pool->cap : 1024;
    while (cap < pool->count + extra) cap *= 2;
    PimChartNode* p = realloc(pool->nodes, sizeof(PimChartNode) * (size_t)cap);
    if (!p) return false;
    pool->nodes = p; pool->cap = cap;
    return true;
}

static void range_bounds(const PimChartNode* nodes, ChartRange r, float2* lo, float2* hi)      /* chart_minmax, lm.c:359-376 */
{
    const float bigNum = (float)(1 << 23);
    *lo = f2(bigNum, bigNum); *hi = f2(-bigNum, -bigNum);
    for (int32_t i = 0; i < r.count; ++i)
    {
        const Tri2D t = node_tri(&nodes[r.first + i]);
        *lo = f2_min(*lo, t.a); *hi = f2_max(*hi, t.a);
        *lo = f2_min(*lo, t.b); *hi = f2_max(*hi, t.b);
        *lo = f2_min(*lo, t.c); *hi = f2_max(*hi, t.c);
    }
}

/* lm.c:614-616: wider than a third of the atlas, or triangles covering under a tenth of the bounds */
static bool range_wants_split(const PimChartNode* nodes, ChartRange r, float maxWidth)
{
    float2 lo, hi;
    range_bounds(nodes, r, &lo, &hi);
    const float2 size = f2_sub(hi, lo);
    const float width = f_max(size.x, size.y);                          /* chart_width, :386-392 */
    float fromTri = 0.0f;                                               /* chart_triarea, :394-404; TriArea2D, area.h:33-39 */
    for (int32_t i = 0; i < r.count; ++i)
    {
        const Tri2D t = node_tri(&nodes[r.first + i]);
        const float2 ab = f2_sub(t.b, t.a), ac = f2_sub(t.c, t.a);
        fromTri += 0.5f * fabsf(ab.x * ac.y - ac.x * ab.y);
    }
    const float fromBounds = size.x * size.y;
    const float density = fromTri / f_max(fromBounds, fromTri);         /* chart_density, :406-412 */
    return (width >= maxWidth) || (density < 0.1f);
}

static float2 tri_center(Tri2D t)                                       /* lm.c:414-421 */
{
    const float scale = 1.0f / 3;
    float2 c = f2_muls(t.a, scale);
    c = f2_add(c, f2_muls(t.b, scale));
    c = f2_add(c, f2_muls(t.c, scale));
    return c;
}

/* chart_split, lm.c:451-518: 2-means over the triangles' centres, distance = signed distance of the mean to the triangle.
 * The two new charts are appended to the pool (members in their old order); counts[] = their sizes. */
static bool split_range(NodePool* pool, ChartRange r, uint32_t serial, uint32_t seed, int32_t counts[kChartSplits], int32_t firsts[kChartSplits])
{
    const int32_t n = r.count;
    uint8_t sideSmall[256];                                             /* nearly every split is of a handful of triangles */
    uint8_t* side = n <= (int32_t)sizeof(sideSmall) ? sideSmall : malloc((size_t)n);
    if (!side || !pool_reserve(pool, n)) { if (side != sideSmall) free(side); return false; }
    const PimChartNode* nodes = pool->nodes + r.first;
    float2 means[kChartSplits], prev[kChartSplits];
    uint32_t state = pcg4_x(serial, (uint32_t)n, seed, kTagChart);
    for (int k = 0; k < kChartSplits; ++k)
    {
        state = pcg1(state);                                            /* Prng_i32: rehash lane x, keep 31 bits */
        means[k] = tri_center(node_tri(&nodes[(int32_t)(state & 0x7fffffffu) % n]));
    }
    int rounds = 0;
    do
    {
        for (int32_t i = 0; i < n; ++i)                                 /* cluster_nearest, lm.c:435-449 */
        {
            const Tri2D t = node_tri(&nodes[i]);
            int chosen = 0; float chosenDist = (float)(1 << 23); bool any = false;
            for (int k = 0; k < kChartSplits; ++k)
            {
                const float dist = tri_distance(t, means[k]);
                if (dist < chosenDist) { chosenDist = dist; chosen = k; any = true; }
            }
            side[i] = (uint8_t)(any ? chosen : 0);
        }
        for (int k = 0; k < kChartSplits; ++k)                          /* cluster_mean, lm.c:423-433: sum of centre / count, in list order */
        {
            int32_t cnt = 0;
            for (int32_t i = 0; i < n; ++i) cnt += side[i] == k;
            const float scale = 1.0f / cnt;
            float2 mean = f2(0.0f, 0.0f);
            for (int32_t i = 0; i < n; ++i) if (side[i] == k) mean = f2_add(mean, f2_muls(tri_center(node_tri(&nodes[i])), scale));
            prev[k] = means[k]; means[k] = mean; counts[k] = cnt;
        }
    } while (memcmp(means, prev, sizeof(means)) && ++rounds < kSplitRounds);
    for (int k = 0; k < kChartSplits; ++k)
    {
        firsts[k] = (int32_t)pool->count;
        for (int32_t i = 0; i < n; ++i) if (side[i] == k) pool->nodes[pool->count++] = pool->nodes[r.first + i];
    }
    if (side != sideSmall) free(side);
    return true;
}

/* buckets of chart planes by floor(plane.w / distThresh).  A bucket's planes lie side by side, in chart order, in one arena (a
 * million nodes make half a million charts, some fifty to a bucket: walked as a linked list through the node array every
 * comparison was a cache miss, 2.2 of the 3.3 s of chart_group; side by side the search is 0.3 s) */
typedef struct { float plane[4]; int32_t chart; } PlaneEntry;
typedef struct { int64_t* keys; int32_t* off; int32_t* cnt; int32_t* cap; int32_t slots; PlaneEntry* arena; int64_t arenaUsed, arenaCap; } PlaneBuckets;
static int32_t bucket_slot(const PlaneBuckets* b, int64_t key)
{
    uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ull;
    int32_t s = (int32_t)(h >> 40) & (b->slots - 1);
    while (b->cap[s] != 0 && b->keys[s] != key) s = (s + 1) & (b->slots - 1);
    return s;
}
static bool bucket_append(PlaneBuckets* b, int64_t key, const float* plane, int32_t chart)
{
    const int32_t s = bucket_slot(b, key);
    if (b->cnt[s] == b->cap[s])
    {
        const int32_t cap = b->cap[s] ? 2 * b->cap[s] : 4;
        if (b->arenaUsed + cap > b->arenaCap)
        {
            int64_t want = b->arenaCap ? 2 * b->arenaCap : 4096;
            while (want < b->arenaUsed + cap) want *= 2;
            PlaneEntry* p = realloc(b->arena, sizeof(PlaneEntry) * (size_t)want);
            if (!p) return false;
            b->arena = p; b->arenaCap = want;
        }
        if (b->cnt[s] > 0) memcpy(b->arena + b->arenaUsed, b->arena + b->off[s], sizeof(PlaneEntry) * (size_t)b->cnt[s]);
        b->keys[s] = key; b->off[s] = (int32_t)b->arenaUsed; b->cap[s] = cap;
        b->arenaUsed += cap;
    }
    PlaneEntry* e = b->arena + b->off[s] + b->cnt[s]++;
    memcpy(e->plane, plane, sizeof(e->plane)); e->chart = chart;
    return true;
}

/* chart_group up to the masks, lm.c:563-636.  On success *nodesIo is replaced by the nodes chart by chart and
 * (*chartFirstOut)[chartCount + 1] delimits them. */
static bool group_charts(PimChartNode** nodesIo, int32_t nodeCount, float distThresh, float degThresh, float maxWidth, uint32_t seed,
    int32_t** chartFirstOut, int32_t* chartCountOut)
{
    const PimChartNode* nodes = *nodesIo;
    const float minCosTheta = cosf(degThresh * (kTau / 360.0f));        /* plane_find, lm.c:348 */
    int32_t* chartOf = malloc(sizeof(int32_t) * (size_t)(nodeCount + 1));
    int32_t* rep = malloc(sizeof(int32_t) * (size_t)(nodeCount + 1));  /* node whose plane stands for the chart */
    PlaneBuckets b; memset(&b, 0, sizeof(b));
    b.slots = 1024; while (b.slots < 4 * (nodeCount + 1)) b.slots *= 2;
    b.keys = malloc(sizeof(int64_t) * (size_t)b.slots);
    b.off = malloc(sizeof(int32_t) * (size_t)b.slots); b.cnt = calloc((size_t)b.slots, sizeof(int32_t)); b.cap = calloc((size_t)b.slots, sizeof(int32_t));
    int32_t* loose = malloc(sizeof(int32_t) * (size_t)(nodeCount + 1)); int32_t looseCount = 0;   /* charts whose plane is not in a bucket */
    bool okAll = chartOf && rep && b.keys && b.off && b.cnt && b.cap && loose;
    int32_t chartCount = 0;
    if (okAll && 3ll * (nodeCount + 1) > 0x7fffffffll) okAll = false;      /* arena offsets are 32-bit (geometric growth: under 3 entries per chart) */
    if (okAll)
    {
        const bool bucketed = distThresh > 0.0f;
        for (int32_t i = 0; i < nodeCount && okAll; ++i)
        {
            const float* pl = nodes[i].plane;
            int32_t found = -1;
            int64_t key = 0;
            /* bucketed only while the quotient converts to int64 without overflow (a tiny distThresh or a huge distance term can push it
             * past 9.2e18); the few charts beyond that are kept in a plain list and compared with every node, as the reference's linear
             * scan compares everything with everything (a NaN distance term matches nothing there either: dist < distThresh is false) */
            const float quot = bucketed ? pl[3] / distThresh : 0.0f;
            const bool finite = bucketed && (fabsf(quot) < 9e18f);
            if (finite)
            {
                key = (int64_t)floorf(quot);
                for (int64_t k = key - 2; k <= key + 2; ++k)          /* +-1 in exact arithmetic; the fp32 quotient can be off by a fraction */
                {
                    const int32_t s = bucket_slot(&b, k);
                    const int32_t n = b.cnt[s];
                    const PlaneEntry* e = n > 0 ? b.arena + b.off[s] : NULL;
                    for (int32_t j = 0; j < n; ++j)     /* in chart order: the first match is the bucket's lowest chart */
                    {
                        const int32_t c = e[j].chart;
                        if (found != -1 && c > found) break;
                        const float* q = e[j].plane;
                        const float dist = fabsf(pl[3] - q[3]);                                   /* plane_equal, lm.c:333-339 */
                        const float cosTheta = q[0] * pl[0] + q[1] * pl[1] + q[2] * pl[2];
                        if ((dist < distThresh) && (cosTheta >= minCosTheta)) { found = c; break; }
                    }
                }
            }
            /* a node outside the buckets is compared with every chart so far, first match wins (the reference's scan, lm.c:341-352);
             * a bucketed one additionally with the charts outside the buckets */
            const int32_t scanCount = !bucketed ? 0 : (finite ? looseCount : chartCount);
            for (int32_t k = 0; k < scanCount; ++k)
            {
                const int32_t c = finite ? loose[k] : k;
                if (found != -1 && c > found) break;                          /* the list is in chart order */
                const float* q = nodes[rep[c]].plane;
                const float dist = fabsf(pl[3] - q[3]);
                const float cosTheta = q[0] * pl[0] + q[1] * pl[1] + q[2] * pl[2];
                if ((dist < distThresh) && (cosTheta >= minCosTheta)) { found = c; break; }
            }
            if (found == -1)
            {
                found = chartCount++;
                rep[found] = i;
                if (finite) { if (!bucket_append(&b, key, pl, found)) okAll = false; }
                else if (bucketed) loose[looseCount++] = found;
            }
            chartOf[i] = found;
        }
    }
    free(b.keys); free(b.off); free(b.cnt); free(b.cap); free(b.arena); free(rep); free(loose);
    if (!okAll) { free(chartOf); return false; }

    /* the nodes chart by chart, members in node order (a counting sort on the chart id) */
    NodePool pool; memset(&pool, 0, sizeof(pool));
    int32_t rangeCap = chartCount + 16, rangeCount = chartCount;
    ChartRange* ranges = calloc((size_t)rangeCap, sizeof(ChartRange));
    if (!ranges || !pool_reserve(&pool, nodeCount)) { free(ranges); free(pool.nodes); free(chartOf); return false; }
    for (int32_t i = 0; i < nodeCount; ++i) ranges[chartOf[i]].count++;
    { int32_t at = 0; for (int32_t c = 0; c < chartCount; ++c) { ranges[c].first = at; at += ranges[c].count; ranges[c].count = 0; } }
    for (int32_t i = 0; i < nodeCount; ++i) { ChartRange* r = &ranges[chartOf[i]]; pool.nodes[r->first + r->count++] = nodes[i]; }
    pool.count = nodeCount;
    free(chartOf);

    /* split big charts, lm.c:609-636: the pieces go to the end of the list, the last piece takes the split chart's place and
     * is looked at again */
    uint32_t serial = 0;
    for (int32_t iChart = 0; iChart < rangeCount; ++iChart)
    {
        const ChartRange r = ranges[iChart];
        if (r.count <= 1 || r.attempts >= kSplitAttempts || !range_wants_split(pool.nodes, r, maxWidth)) continue;
        int32_t counts[kChartSplits], firsts[kChartSplits];
        if (!split_range(&pool, r, serial++, seed, counts, firsts)) { free(ranges); free(pool.nodes); return false; }
        int32_t made = 0;
        for (int k = 0; k < kChartSplits; ++k)
        {
            if (counts[k] <= 0) continue;
            if (rangeCount == rangeCap)
            {
                rangeCap *= 2;
                ChartRange* p = realloc(ranges, sizeof(ChartRange) * (size_t)rangeCap);
                if (!p) { free(ranges); free(pool.nodes); return false; }
                ranges = p;
            }
            ranges[rangeCount].first = firsts[k]; ranges[rangeCount].count = counts[k]; ranges[rangeCount].attempts = 0;
            ++rangeCount; ++made;
        }
        if (made == 1) ranges[rangeCount - 1].attempts = r.attempts + 1;        /* undivided: tried again with other means, a bounded number of times */
        ranges[iChart] = ranges[rangeCount - 1];
        --rangeCount;
        --iChart;
    }

    PimChartNode* out = malloc(sizeof(PimChartNode) * (size_t)(nodeCount + 1));
    int32_t* chartFirst = malloc(sizeof(int32_t) * (size_t)(rangeCount + 1));
    if (!out || !chartFirst) { free(out); free(chartFirst); free(ranges); free(pool.nodes); return false; }
    int32_t at = 0;
    for (int32_t c = 0; c < rangeCount; ++c)
    {
        chartFirst[c] = at;
        memcpy(out + at, pool.nodes + ranges[c].first, sizeof(PimChartNode) * (size_t)ranges[c].count);
        at += ranges[c].count;
    }
    chartFirst[rangeCount] = at;
    free(ranges); free(pool.nodes);
    free(*nodesIo);
    *nodesIo = out; *chartFirstOut = chartFirst; *chartCountOut = rangeCount;
    return true;
}

/* chart_sort, lm.c:648-660: QuickSort (common/sort.c:14-75) with chart_cmp over (area, chart) records.  Selection sort below
 * five items, else Hoare partition around the middle item, both halves recursively: equal areas end up where exactly this
 * procedure leaves them, and the atlas search below depends on that order. */
typedef struct { float area; int32_t chart; } SortRec;
static int area_cmp(float a, float b) { return ((a > b) ? 1 : 0) - ((b > a) ? 1 : 0); }
static void sort_records(SortRec* items, int32_t count)
{
    while (count > 4)
    {
        int32_t i = 0, j = count - 1;
        const float pivot = items[count >> 1].area;
        for (;;)
        {
            while ((i < count) && (area_cmp(items[i].area, pivot) < 0)) ++i;
            while ((j > 0) && (area_cmp(items[j].area, pivot) > 0)) --j;
            if (i >= j) break;
            const SortRec t = items[i]; items[i] = items[j]; items[j] = t;
            ++i; --j;
        }
        sort_records(items, i);                  /* left part by recursion, right part by the loop: same calls, bounded stack */
        items += i; count -= i;
    }
    for (int32_t i = 0; i < count; ++i)
    {
        int32_t lo = i;
        for (int32_t j = i + 1; j < count; ++j) if (area_cmp(items[j].area, items[lo].area) < 0) lo = j;
        if (lo != i) { const SortRec t = items[i]; items[i] = items[lo]; items[lo] = t; }
    }
}

void LmPack_SortCharts(const float* areas, i32 chartCount, i32* order)
{
    SortRec* recs = malloc(sizeof(SortRec) * (size_t)(chartCount > 0 ? chartCount : 1));
    if (!recs) { for (i32 c = 0; c < chartCount; ++c) order[c] = c; return; }
    for (i32 c = 0; c < chartCount; ++c) { recs[c].area = areas[c]; recs[c].chart = c; }
    sort_records(recs, chartCount);
    for (i32 c = 0; c < chartCount; ++c) order[c] = recs[c].chart;
    free(recs);
}

i32 LmPack_GroupCharts(PimChartNode* nodes, i32 nodeCount, float distThresh, float degThresh, float maxWidth, uint32_t seed, i32* chartFirst)
{
    PimChartNode* work = malloc(sizeof(PimChartNode) * (size_t)(nodeCount + 1));
    if (!work) return -1;
    memcpy(work, nodes, sizeof(PimChartNode) * (size_t)nodeCount);
    int32_t* first = NULL; int32_t chartCount = 0;
    if (!group_charts(&work, nodeCount, distThresh, degThresh, maxWidth, seed, &first, &chartCount)) { free(work); return -1; }
    memcpy(nodes, work, sizeof(PimChartNode) * (size_t)nodeCount);
    memcpy(chartFirst, first, sizeof(int32_t) * (size_t)(chartCount + 1));
    free(work); free(first);
    return chartCount;
}

/* atlas_estimate, lm.c:825-843 */
static i32 atlas_estimate(i32 atlasSize, const float* areas, i32 chartCount)
{
    i32 areaRequired = 0;
    for (i32 i = 0; i < chartCount; ++i) areaRequired += (i32)ceilf(areas[i]);
    const i32 areaPerAtlas = atlasSize * atlasSize;
    i32 atlasCount = 0;
    while ((atlasCount * areaPerAtlas) < areaRequired) ++atlasCount;
    return atlasCount > 1 ? atlasCount : 1;
}

/* chartnodes_assign, lm.c:876-934 */
static void assign_charts(const PimDrawables* d, const PimChartNode* nodes, const i32* chartFirst, i32 chartCount, const i32* translations,
    const i32* atlasIndex, const Lightmap* lightmaps)
{
    for (i32 c = 0; c < chartCount; ++c)
    {
        const Lightmap* lightmap = &lightmaps[atlasIndex[c]];
        const i32 lmTexId = (i32)lightmap->slot.index;
        const float scale = 1.0f / lightmap->size;
        const float2 tr = f2((float)translations[2 * c], (float)translations[2 * c + 1]);
        for (i32 i = chartFirst[c]; i < chartFirst[c + 1]; ++i)
        {
            const PimChartNode* node = &nodes[i];
            if (d->meshes[node->drawableIndex] < 0) continue;
            const PimMesh* mesh = &d->meshTable[d->meshes[node->drawableIndex]];
            float* uvs = (float*)mesh->uvs; int32_t* texIndices = (int32_t*)mesh->texIndices;      /* the meshes are the engine's, written in place as in the reference */
            for (int k = 0; k < 3; ++k)
            {
                const i32 iv = node->vertIndex + k;
                if (texIndices) texIndices[iv * 4 + 3] = lmTexId;
                const float2 uv = f2_muls(f2_add(f2(node->triCoord[2 * k], node->triCoord[2 * k + 1]), tr), scale);
                uvs[iv * 4 + 2] = uv.x; uvs[iv * 4 + 3] = uv.y;
            }
        }
    }
}

static char ms_packError[256];
const char* LmPack_LastError(void) { return ms_packError; }
static LmPack pack_fail(const char* what, const char* detail)
{
    snprintf(ms_packError, sizeof(ms_packError), "%s%s%s", what, detail ? ": " : "", detail ? detail : "");
    fprintf(stderr, "[lm] %s\n", ms_packError);
    LmPack none; memset(&none, 0, sizeof(none));
    return none;
}

/* The pack's pages are page-locked blocks of 113 MB at 1024^2, and what such a block costs is the first touch of its pages (half a
 * second for the 62 pages of a million triangles).  Their number is known, give or take one, before the atlas search starts
 * (atlas_estimate), and the search keeps one GPU workgroup and no host core busy: a second thread makes the blocks meanwhile. */
typedef struct { i32 count, size; float** blocks; } PagesAhead;
static int pages_ahead_run(void* arg)
{
    PagesAhead* p = arg;
    const size_t bytes = (size_t)p->size * (size_t)p->size * 27 * sizeof(float);
    for (i32 i = 0; i < p->count; ++i) { p->blocks[i] = pimgpu_host_alloc(bytes); if (!p->blocks[i]) break; }
    return 0;
}

/* PIM_LMPACK_LOG=1: seconds per step of LmPack_Pack on stderr */
static double pack_now(void) { struct timespec ts; timespec_get(&ts, TIME_UTC); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static void pack_step(const char* what, double* t0)
{
    const double t1 = pack_now();
    if (getenv("PIM_LMPACK_LOG")) fprintf(stderr, "[lm] %-24s %.3f s\n", what, t1 - *t0);
    *t0 = t1;
}

LmPack LmPack_PackEx(const PimDrawables* drawables, i32 atlasSize, float texelsPerUnit, float distThresh, float degThresh, uint32_t seed)
{
    ms_packError[0] = 0;
    double stepT = pack_now();
    if (!drawables || atlasSize <= 0 || !(texelsPerUnit > 0.0f)) return pack_fail("LmPack_Pack: bad argument", NULL);
    const float maxWidth = atlasSize / 3.0f;                            /* lm.c:1065 */

    i32 nodeCount = 0;
    if (pimgpu_lightmap_chartnodes(drawables, texelsPerUnit, NULL, 0, &nodeCount)) return pack_fail("chartnodes_create", pimgpu_last_error());
    PimChartNode* nodes = malloc(sizeof(PimChartNode) * (size_t)(nodeCount + 1));
    if (!nodes) return pack_fail("LmPack_Pack: out of memory", NULL);
    if (nodeCount > 0 && pimgpu_lightmap_chartnodes(drawables, texelsPerUnit, nodes, nodeCount, &nodeCount)) { free(nodes); return pack_fail("chartnodes_create", pimgpu_last_error()); }

    pack_step("chartnodes_create", &stepT);
    i32* chartFirst = NULL; i32 chartCount = 0;
    if (!group_charts(&nodes, nodeCount, distThresh, degThresh, maxWidth, seed, &chartFirst, &chartCount)) { free(nodes); return pack_fail("chart_group: out of memory", NULL); }

    pack_step("chart_group", &stepT);
    LmPack pack; memset(&pack, 0, sizeof(pack));
    float* areas = malloc(sizeof(float) * (size_t)(chartCount + 1));
    i32* maskSizes = malloc(sizeof(i32) * 2 * (size_t)(chartCount + 1));
    int64_t* maskOffsets = malloc(sizeof(int64_t) * (size_t)(chartCount + 1));
    i32* order = malloc(sizeof(i32) * (size_t)(chartCount + 1));
    i32* translations = malloc(sizeof(i32) * 2 * (size_t)(chartCount + 1));
    i32* atlasIndex = malloc(sizeof(i32) * (size_t)(chartCount + 1));
    uint8_t* masks = NULL;
    const char* failed = NULL;
    if (!areas || !maskSizes || !maskOffsets || !order || !translations || !atlasIndex) failed = "LmPack_Pack: out of memory";
    i32 used = 0;
    PagesAhead pre; memset(&pre, 0, sizeof(pre));
    thrd_t preThread; bool preStarted = false;
    if (!failed && chartCount > 0)
    {
        if (pimgpu_lightmap_chart_layout(nodes, chartFirst, chartCount, areas, maskSizes)) failed = "ChartMaskFn";
        int64_t total = 0;
        if (!failed) for (i32 c = 0; c < chartCount; ++c) { maskOffsets[c] = total; total += (int64_t)maskSizes[2 * c] * maskSizes[2 * c + 1]; }
        if (!failed && !(masks = malloc((size_t)(total > 0 ? total : 1)))) failed = "LmPack_Pack: out of memory";
        if (!failed && pimgpu_lightmap_chart_masks(nodes, chartFirst, chartCount, maskSizes, maskOffsets, masks)) failed = "ChartMaskFn";
        pack_step("chart masks", &stepT);
        if (!failed)
        {
            LmPack_SortCharts(areas, chartCount, order);                /* chart_sort, lm.c:1074 */
            const i32 atlasCount = atlas_estimate(atlasSize, areas, chartCount);
            /* atlas_estimate counts the charts' bounds without their 2-texel margins and does run short; where the reference would
             * then search for ever, further atlases are opened, up to four times the estimate and sixteen more */
            const char* ahead = getenv("PIM_LMPACK_PAGES_AHEAD");                  /* "0": pages made after the search, on this thread */
            const int64_t pageBytes = (int64_t)atlasSize * atlasSize * 27 * (int64_t)sizeof(float);
            if (!(ahead && ahead[0] == '0') && (pre.blocks = calloc((size_t)atlasCount, sizeof(float*))))
            {
                pre.size = atlasSize;
                pre.count = (i32)((int64_t)atlasCount * pageBytes <= (32ll << 30) ? atlasCount : (32ll << 30) / pageBytes);
                preStarted = pre.count > 0 && thrd_create(&preThread, pages_ahead_run, &pre) == thrd_success;
            }
            if (pimgpu_lightmap_atlas_place(masks, maskOffsets, maskSizes, areas, order, chartCount, atlasSize, atlasCount, atlasCount * 4 + 16, translations, atlasIndex, &used))
                failed = "atlases_create";
            if (preStarted) thrd_join(preThread, NULL);
            else pre.count = 0;
            pack_step("sort + atlases_create", &stepT);
        }
    }
    if (!failed)
    {
        pack.lmCount = used; pack.lmSize = atlasSize; pack.texelsPerMeter = texelsPerUnit;
        static const float kAxii[kGiDirections][4] = {      /* what SG_Generate(axii, 5, SGDist_Hemi) leaves (sphgauss.c:139-171), lightmap.h:14-21 */
            { 0.000000f, 0.000000f, 1.000000f, 4.999773f }, { 0.577350f, 0.577350f, 0.577350f, 4.999773f }, { -0.577350f, 0.577350f, 0.577350f, 4.999773f },
            { 0.577350f, -0.577350f, 0.577350f, 4.999773f }, { -0.577350f, -0.577350f, 0.577350f, 4.999773f } };
        memcpy(pack.axii, kAxii, sizeof(kAxii));
        pack.lightmaps = calloc((size_t)(used > 0 ? used : 1), sizeof(Lightmap));
        if (!pack.lightmaps) failed = "LmPack_Pack: out of memory";
        for (i32 i = 0; i < used && !failed; ++i)
        {
            float* block = (pre.blocks && i < pre.count) ? pre.blocks[i] : NULL;
            if (block) pre.blocks[i] = NULL;                                    /* the page owns it now */
            Lightmap_NewIn(&pack.lightmaps[i], atlasSize, block);
            if (!pack.lightmaps[i].probes[0]) failed = "Lightmap_New";
        }
        pack_step("Lightmap_New pages", &stepT);
        if (!failed && used > 0)
        {
            assign_charts(drawables, nodes, chartFirst, chartCount, translations, atlasIndex, pack.lightmaps);
            pack_step("chartnodes_assign", &stepT);
            i32* texIds = malloc(sizeof(i32) * (size_t)used);
            if (texIds)
            {
                for (i32 i = 0; i < used; ++i) texIds[i] = (i32)pack.lightmaps[i].slot.index;
                LmPack_EmbedAttributes(drawables, pack.lightmaps, texIds, used);      /* lm.c:1092 */
                pack_step("EmbedAttributes", &stepT);
                free(texIds);
            }
            else failed = "LmPack_Pack: out of memory";
        }
    }
    char detail[200]; detail[0] = 0;
    if (failed) snprintf(detail, sizeof(detail), "%s", pimgpu_last_error());
    if (pre.blocks) { for (i32 i = 0; i < pre.count; ++i) pimgpu_host_free(pre.blocks[i]); free(pre.blocks); }      /* made ahead and not needed */
    free(nodes); free(chartFirst); free(areas); free(maskSizes); free(maskOffsets); free(order); free(translations); free(atlasIndex); free(masks);
    if (failed) { LmPack_Del(&pack); return pack_fail(failed, detail[0] ? detail : NULL); }
    return pack;
}

LmPack LmPack_Pack(i32 atlasSize, float texelsPerUnit, float distThresh, float degThresh)     /* lightmap.h:62-66 */
{
    if (!ms_drawables) return pack_fail("LmPack_Pack: no drawables provider (PtSys_SetDrawablesProvider)", NULL);
    PimDrawables d; memset(&d, 0, sizeof(d));
    if (!ms_drawables(&d, ms_drawablesUser)) return pack_fail("LmPack_Pack: the drawables provider failed", NULL);
    return LmPack_PackEx(&d, atlasSize, texelsPerUnit, distThresh, degThresh, PtSys_Seed());
}
