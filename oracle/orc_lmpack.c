/*
 * orc_lmpack.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The sequential steps of LmPack_Pack (rendering/lightmap.c:1047-1102, "lm.c" below) restated one for one, on one thread:
 * chart_group with plane_find / chart_split (:333-357, :451-518, :563-636), chart_sort over common/sort.c's QuickSort
 * (:648-660, sort.c:14-75), atlas_estimate (:825-843), AtlasFn / atlas_search / mask_find / mask_fits / mask_write as ONE
 * worker runs them (:174-230, :261-282, :680-721, :779-823), chartnodes_assign (:876-939).  The data-parallel steps around them
 * (chartnodes_create, ChartMaskFn, EmbedAttributes) are in pim_oracle.c.
 *
 * Parity: lightmap.c cannot be compiled here (Vulkan headers) and sort.c needs the engine's allocator, so these
 * are restatements PINNED ONLY through their leaves (sdTriangle2D, NormalToTBN, the Pcg generators: tests/golden/ref_math.bin);
 * the control flow is "parity unpinned" -- the reference holds no fixture for its packer.
 *
 * Where this departs from the reference (the same in the product, pim_amd/host/pim_lmpack.c):
 *   - chart_split seeds its two means from the global clock-seeded Prng (lm.c:464-467).  Here split call number s draws from
 *     the stream Pcg4(s, nodeCount, seed, TAG);
 *   - a chart that chart_split cannot divide (both means fall on one cluster) is split again for ever by the loop at
 *     lm.c:609-636; here it is kept whole after kSplitAttempts tries;
 *   - AtlasFn never returns when a chart fits in none of the atlas_estimate()d atlases (lm.c:808-818; the estimate counts bounds
 *     without margins and does run short); here one more atlas is opened for that chart, up to atlasCap, and a chart larger
 *     than an atlas (or a full cap) is reported (-1).  Every placement before that point is the reference's;
 *   - the atlas search runs as ONE worker: with several, the reference's result depends on thread timing (lm.c:793).
 */
#include "pim_oracle.h"
#include "orc_math.h"

#include <stdlib.h>
#include <string.h>

#define ORC_TAG_CHART 0x43485254u           /* 'CHRT' */
#define ORC_ROW_RESET (-(1 << 20))          /* lm.c:37 */
enum { kChartSplits = 2, kSplitAttempts = 16 };

typedef struct { v2 a, b, c; } Tri2;

static Tri2 node_tri(const PimChartNode* n) { Tri2 t; t.a = o2(n->triCoord[0], n->triCoord[1]); t.b = o2(n->triCoord[2], n->triCoord[3]); t.c = o2(n->triCoord[4], n->triCoord[5]); return t; }

typedef struct { PimChartNode* nodes; int32_t count, cap; int32_t attempts; } Chart;

static void chart_push(Chart* c, const PimChartNode* n)
{
    if (c->count == c->cap) { c->cap = c->cap ? c->cap * 2 : 4; c->nodes = realloc(c->nodes, sizeof(PimChartNode) * (size_t)c->cap); }
    c->nodes[c->count++] = *n;
}

static void chart_bounds(const Chart* c, v2* loOut, v2* hiOut)         /* chart_minmax, lm.c:359-376 */
{
    const float bigNum = (float)(1 << 23);
    v2 lo = o2(bigNum, bigNum), hi = o2(-bigNum, -bigNum);
    for (int32_t i = 0; i < c->count; ++i)
    {
        const Tri2 t = node_tri(&c->nodes[i]);
        lo = o2_min(lo, t.a); hi = o2(o_max(hi.x, t.a.x), o_max(hi.y, t.a.y));
        lo = o2_min(lo, t.b); hi = o2(o_max(hi.x, t.b.x), o_max(hi.y, t.b.y));
        lo = o2_min(lo, t.c); hi = o2(o_max(hi.x, t.c.x), o_max(hi.y, t.c.y));
    }
    *loOut = lo; *hiOut = hi;
}
static float chart_area(const Chart* c) { v2 lo, hi; chart_bounds(c, &lo, &hi); const v2 s = o2_sub(hi, lo); return s.x * s.y; }            /* :378-384 */
static float chart_width(const Chart* c) { v2 lo, hi; chart_bounds(c, &lo, &hi); const v2 s = o2_sub(hi, lo); return o_max(s.x, s.y); }     /* :386-392 */
static float tri_area2(Tri2 t) { const v2 ab = o2_sub(t.b, t.a), ac = o2_sub(t.c, t.a); const float cr = ab.x * ac.y - ac.x * ab.y; return 0.5f * o_abs(cr); }   /* area.h:33-39 */
static float chart_density(const Chart* c)                              /* :394-412 */
{
    float fromTri = 0.0f;
    for (int32_t i = 0; i < c->count; ++i) fromTri += tri_area2(node_tri(&c->nodes[i]));
    const float fromBounds = chart_area(c);
    return fromTri / o_max(fromBounds, fromTri);
}
static v2 tri_center(Tri2 t)                                            /* :414-421 */
{
    const float scale = 1.0f / 3;
    v2 center = o2_muls(t.a, scale);
    center = o2_add(center, o2_muls(t.b, scale));
    center = o2_add(center, o2_muls(t.c, scale));
    return center;
}
static v2 cluster_mean(const Tri2* tris, int32_t count)                /* :423-433 (count == 0: scale = inf, mean stays 0) */
{
    const float scale = 1.0f / count;
    v2 mean = o2(0.0f, 0.0f);
    for (int32_t i = 0; i < count; ++i) mean = o2_add(mean, o2_muls(tri_center(tris[i]), scale));
    return mean;
}
static int32_t cluster_nearest(const v2* means, int32_t k, Tri2 t)     /* :435-449 */
{
    int32_t chosen = -1;
    float chosenDist = (float)(1 << 23);
    for (int32_t i = 0; i < k; ++i)
    {
        const float dist = sdTriangle2D(t.a, t.b, t.c, means[i]);
        if (dist < chosenDist) { chosenDist = dist; chosen = i; }
    }
    return chosen;
}

/* chart_split, lm.c:451-518 */
static void chart_split(const Chart* chart, Chart* split, uint32_t serial, uint32_t seed)
{
    const int32_t nodeCount = chart->count;
    v2 means[kChartSplits], prevMeans[kChartSplits];
    int32_t counts[kChartSplits] = { 0, 0 };
    Tri2* triLists[kChartSplits]; int32_t* nodeLists[kChartSplits];
    u4 rng = pcg4(serial, (uint32_t)nodeCount, seed, ORC_TAG_CHART);
    for (int i = 0; i < kChartSplits; ++i)
    {
        const int32_t j = (int32_t)(rng_u32(&rng) & 0x7fffffffu) % nodeCount;       /* Prng_i32, random.h:63-67 */
        means[i] = tri_center(node_tri(&chart->nodes[j]));
        prevMeans[i] = o2(0.0f, 0.0f);
        triLists[i] = malloc(sizeof(Tri2) * (size_t)nodeCount);
        nodeLists[i] = malloc(sizeof(int32_t) * (size_t)nodeCount);
    }
    int guard = 0;
    do
    {
        for (int i = 0; i < kChartSplits; ++i) counts[i] = 0;
        for (int32_t i = 0; i < nodeCount; ++i)
        {
            const Tri2 t = node_tri(&chart->nodes[i]);
            int32_t cl = cluster_nearest(means, kChartSplits, t);
            if (cl < 0) cl = 0;                     /* every distance >= 2^23 or NaN: the reference indexes [-1] here */
            const int32_t back = counts[cl]++;
            triLists[cl][back] = t; nodeLists[cl][back] = i;
        }
        for (int i = 0; i < kChartSplits; ++i) { prevMeans[i] = means[i]; means[i] = cluster_mean(triLists[i], counts[i]); }
    } while (memcmp(means, prevMeans, sizeof(means)) && ++guard < 1000);
    for (int i = 0; i < kChartSplits; ++i)
    {
        Chart ch = { 0 };
        for (int32_t j = 0; j < counts[i]; ++j) chart_push(&ch, &chart->nodes[nodeLists[i][j]]);
        split[i] = ch;
        free(triLists[i]); free(nodeLists[i]);
    }
}

/* chart_group up to the masks, lm.c:563-636.  nodesOut: the nodes chart by chart; chartFirst[chartCount + 1]. */
int32_t orc_lightmap_chart_group(const PimChartNode* nodes, int32_t nodeCount, float distThresh, float degThresh, float maxWidth,
    uint32_t seed, PimChartNode* nodesOut, int32_t* chartFirst)
{
    int32_t chartCount = 0, chartCap = 0;
    Chart* charts = NULL;
    v4* planes = NULL;
    const float minCosTheta = cosf(degThresh * (ORC_TAU / 360.0f));     /* plane_find, lm.c:348 */
    for (int32_t iNode = 0; iNode < nodeCount; ++iNode)
    {
        const PimChartNode* node = &nodes[iNode];
        const v4 plane = o4(node->plane[0], node->plane[1], node->plane[2], node->plane[3]);
        int32_t iChart = -1;
        for (int32_t i = 0; i < chartCount; ++i)                       /* plane_equal, :333-339 */
        {
            const float dist = o_abs(plane.w - planes[i].w);
            const float cosTheta = o4_dot3(planes[i], plane);
            if ((dist < distThresh) && (cosTheta >= minCosTheta)) { iChart = i; break; }
        }
        if (iChart == -1)
        {
            if (chartCount == chartCap) { chartCap = chartCap ? chartCap * 2 : 64; charts = realloc(charts, sizeof(Chart) * (size_t)chartCap); planes = realloc(planes, sizeof(v4) * (size_t)chartCap); }
            iChart = chartCount++;
            memset(&charts[iChart], 0, sizeof(Chart));
            planes[iChart] = plane;
        }
        chart_push(&charts[iChart], node);
    }
    free(planes);
    uint32_t serial = 0;
    for (int32_t iChart = 0; iChart < chartCount; ++iChart)             /* split big charts, :609-636 */
    {
        Chart chart = charts[iChart];
        if (chart.count > 1 && chart.attempts < kSplitAttempts)
        {
            const float width = chart_width(&chart), density = chart_density(&chart);
            if ((width >= maxWidth) || (density < 0.1f))
            {
                Chart split[kChartSplits];
                chart_split(&chart, split, serial++, seed);
                int32_t made = 0;
                for (int j = 0; j < kChartSplits; ++j)
                    if (split[j].count > 0)
                    {
                        if (chartCount == chartCap) { chartCap *= 2; charts = realloc(charts, sizeof(Chart) * (size_t)chartCap); }
                        charts[chartCount++] = split[j];
                        ++made;
                    }
                    else free(split[j].nodes);
                if (made == 1) charts[chartCount - 1].attempts = chart.attempts + 1;     /* undivided: another try, with new means */
                free(chart.nodes);
                charts[iChart] = charts[chartCount - 1];
                --chartCount;
                --iChart;
            }
        }
    }
    int32_t at = 0;
    for (int32_t c = 0; c < chartCount; ++c)
    {
        chartFirst[c] = at;
        memcpy(nodesOut + at, charts[c].nodes, sizeof(PimChartNode) * (size_t)charts[c].count);
        at += charts[c].count;
        free(charts[c].nodes);
    }
    chartFirst[chartCount] = at;
    free(charts);
    return chartCount;
}

/* chart_sort: QuickSort (common/sort.c:14-75) with chart_cmp (lm.c:648-655) over records {area, chart}; order[i] = the chart
 * that ends up in place i.  Equal areas keep whatever order this particular quicksort leaves them in. */
typedef struct { float area; int32_t chart; } SortRec;
static int chart_cmp(const SortRec* lhs, const SortRec* rhs) { const float a = lhs->area, b = rhs->area; return ((a > b) ? 1 : 0) - ((b > a) ? 1 : 0); }
static void quick_sort(SortRec* items, int32_t count)
{
    if (count <= 4)
    {
        for (int32_t i = 0; i < count; ++i)
        {
            int32_t lo = i;
            for (int32_t j = i + 1; j < count; ++j) if (chart_cmp(&items[j], &items[lo]) < 0) lo = j;
            if (lo != i) { const SortRec t = items[i]; items[i] = items[lo]; items[lo] = t; }
        }
        return;
    }
    int32_t i = 0;
    {
        int32_t j = count - 1;
        const SortRec pivot = items[count >> 1];
        for (;;)
        {
            while ((i < count) && (chart_cmp(&items[i], &pivot) < 0)) ++i;
            while ((j > 0) && (chart_cmp(&items[j], &pivot) > 0)) --j;
            if (i >= j) break;
            const SortRec t = items[i]; items[i] = items[j]; items[j] = t;
            ++i; --j;
        }
    }
    quick_sort(items, i);
    quick_sort(items + i, count - i);
}
void orc_lightmap_chart_sort(const float* areas, int32_t chartCount, int32_t* order)
{
    SortRec* recs = malloc(sizeof(SortRec) * (size_t)(chartCount > 0 ? chartCount : 1));
    for (int32_t c = 0; c < chartCount; ++c) { recs[c].area = areas[c]; recs[c].chart = c; }
    quick_sort(recs, chartCount);
    for (int32_t c = 0; c < chartCount; ++c) order[c] = recs[c].chart;
    free(recs);
}

/* atlas_estimate, lm.c:825-843 (chart_area of the moved chart == the area ChartMaskFn stored) */
int32_t orc_lightmap_atlas_estimate(const float* areas, int32_t chartCount, int32_t atlasSize)
{
    int32_t areaRequired = 0;
    for (int32_t i = 0; i < chartCount; ++i) areaRequired += (int32_t)ceilf(areas[i]);
    const int32_t areaPerAtlas = atlasSize * atlasSize;
    int32_t atlasCount = 0;
    while ((atlasCount * areaPerAtlas) < areaRequired) ++atlasCount;
    return atlasCount > 1 ? atlasCount : 1;
}

static bool mask_fits(const uint8_t* a, int32_t aSize, const uint8_t* b, int32_t bw, int32_t bh, int32_t tx, int32_t ty)      /* lm.c:174-207 */
{
    if ((tx < 0) || (ty < 0)) return false;
    if ((tx + bw > aSize) || (ty + bh > aSize)) return false;
    for (int32_t ay = ty; ay < ty + bh; ++ay)
        for (int32_t ax = tx; ax < tx + bw; ++ax)
            if ((b[(ax - tx) + (ay - ty) * bw] & a[ax + ay * aSize]) != 0) return false;
    return true;
}
static bool mask_find(const uint8_t* atlas, int32_t aSize, const uint8_t* item, int32_t bw, int32_t bh, int32_t* txOut, int32_t* tyOut, int32_t prevRow)   /* :261-282 */
{
    const int32_t rangeX = aSize - bw, rangeY = aSize - bh;
    int32_t y = 0;
    if (prevRow != ORC_ROW_RESET) y = prevRow;
    for (; y < rangeY; ++y)
        for (int32_t x = 0; x < rangeX; ++x)
            if (mask_fits(atlas, aSize, item, bw, bh, x, y)) { *txOut = x; *tyOut = y; return true; }
    return false;
}

/* atlases_create's AtlasFn as one worker, lm.c:779-823 with atlas_search :680-721.  Returns the number of atlases that took a
 * chart (:862-870), or -1 when a chart fits nowhere. */
int32_t orc_lightmap_atlas_place(const uint8_t* masks, const int64_t* maskOffsets, const int32_t* maskSizes, const float* areas,
    const int32_t* order, int32_t chartCount, int32_t atlasSize, int32_t atlasCount, int32_t atlasCap, int32_t* translations, int32_t* atlasIndex)
{
    const size_t aBytes = (size_t)atlasSize * (size_t)atlasSize;
    if (atlasCap < atlasCount) atlasCap = atlasCount;
    uint8_t* atlases = calloc((size_t)atlasCap, aBytes);
    int32_t* counts = calloc((size_t)atlasCap, sizeof(int32_t));
    int32_t prevAtlas = 0, prevRow = ORC_ROW_RESET;
    float prevArea = (float)(1 << 23);
    bool failed = false;
    for (int32_t i = 0; i < chartCount && !failed; ++i)
    {
        const int32_t c = order[i];
        const uint8_t* item = masks + maskOffsets[c];
        const int32_t bw = maskSizes[2 * c], bh = maskSizes[2 * c + 1];
        atlasIndex[c] = -1;
        if (areas[c] < (prevArea * 0.9f)) { prevArea = areas[c]; prevAtlas = 0; prevRow = ORC_ROW_RESET; }
        for (int attempt = 0; ; ++attempt)
        {
            bool placed = false;
            for (int32_t a = prevAtlas; a < atlasCount && !placed; ++a)          /* atlas_search */
            {
                uint8_t* atlas = atlases + aBytes * (size_t)a;
                int32_t tx, ty;
                if (mask_find(atlas, atlasSize, item, bw, bh, &tx, &ty, prevRow))
                {
                    prevAtlas = a; prevRow = ty;
                    for (int32_t by = 0; by < bh; ++by)                           /* mask_write, :209-230 */
                        for (int32_t bx = 0; bx < bw; ++bx)
                            if (item[bx + by * bw]) atlas[(bx + tx) + (by + ty) * atlasSize] = item[bx + by * bw];
                    translations[2 * c] = tx; translations[2 * c + 1] = ty;
                    atlasIndex[c] = a;
                    counts[a]++;
                    placed = true;
                }
                else prevRow = ORC_ROW_RESET;
            }
            if (placed) break;
            if (attempt == 1)
            {
                /* a full search from atlas 0, row 0 found nothing: the reference spins here.  One more atlas, if the chart can fit one. */
                if (atlasCount >= atlasCap || bw >= atlasSize || bh >= atlasSize) { failed = true; break; }
                prevArea = areas[c]; prevAtlas = atlasCount; prevRow = ORC_ROW_RESET;
                ++atlasCount;
                continue;
            }
            if (attempt == 2) { failed = true; break; }
            prevArea = areas[c]; prevAtlas = 0; prevRow = ORC_ROW_RESET;
        }
    }
    int32_t used = 0;
    for (int32_t a = 0; a < atlasCount; ++a) if (counts[a] > 0) ++used;
    free(atlases); free(counts);
    return failed ? -1 : used;
}

/* chartnodes_assign, lm.c:876-934: lightmap uvs into uvs.zw, the page's texture id into texIndices.w of the three vertices */
void orc_lightmap_chart_assign(const PimDrawables* d, const PimChartNode* nodes, const int32_t* chartFirst, int32_t chartCount,
    const int32_t* translations, const int32_t* atlasIndex, const int32_t* pageTexIds, int32_t lmSize)
{
    for (int32_t c = 0; c < chartCount; ++c)
    {
        const int32_t lmTexId = pageTexIds ? pageTexIds[atlasIndex[c]] : atlasIndex[c];
        const float scale = 1.0f / lmSize;
        const v2 tr = o2((float)translations[2 * c], (float)translations[2 * c + 1]);
        for (int32_t i = chartFirst[c]; i < chartFirst[c + 1]; ++i)
        {
            const PimChartNode* node = &nodes[i];
            if (d->meshes[node->drawableIndex] < 0) continue;
            const PimMesh* mesh = &d->meshTable[d->meshes[node->drawableIndex]];
            float* uvs = (float*)mesh->uvs; int32_t* texIndices = (int32_t*)mesh->texIndices;
            for (int k = 0; k < 3; ++k)
            {
                const int32_t iv = node->vertIndex + k;
                if (texIndices) texIndices[iv * 4 + 3] = lmTexId;
                const v2 uv = o2_muls(o2_add(o2(node->triCoord[2 * k], node->triCoord[2 * k + 1]), tr), scale);
                uvs[iv * 4 + 2] = uv.x; uvs[iv * 4 + 3] = uv.y;
            }
        }
    }
}
