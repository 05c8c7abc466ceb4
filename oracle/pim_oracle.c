/*
 * pim_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.  See pim_oracle.h.
 *
 * Follows /root/reference/src/rendering/path_tracer.c (cited as pt.c:LINE),
 * rendering/lightmap.c (lm.c:LINE), math/dist1d.c, math/grid.h and
 * rendering/render_system.c:1374-1394, with these BUILD-DEFINED parts where the
 * reference has none or cannot be reproduced:
 *   D1  RNG streams are per pixel / ray / texel / cell, seeded by Pcg4 of the
 *       work-item coordinates (the reference seeds per host thread from the clock);
 *   D2  an optional maxBounces clamp on the 666-iteration bounce loop;
 *   D7  rtcIntersect1 is replaced by a watertight fp32 closest hit (Woop et al.
 *       2013) with lowest-primID tie-break and (tNear, tFar] acceptance.
 * Build: gcc -std=c11 -O2 -ffp-contract=off (no -ffast-math), FTZ/DAZ set on every
 * worker as threading/task.c:73-74 does.
 */
#include "pim_oracle.h"
#include "orc_math.h"

#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <xmmintrin.h>
#include <pmmintrin.h>

/* ------------------------------------------------------------------------------------------ */
/* build-defined stream tags (shared with the HIP side: pim_amd/csrc/pt_common.h)              */
#define TAG_PIXEL    0x9E3779B9u
#define TAG_RAY      0x7261795Fu
#define TAG_BAKE     0x62616B65u
#define TAG_BAKE_PT  0xA511E9B3u
#define TAG_EMIT     0x454D4954u
#define TAG_GRID     0x47524944u
#define TAG_CUBE     0x43554245u

typedef struct { float* pdf; float* cdf; uint32_t* live; } OrcDistArrays;

typedef struct
{
    v4 constantColor, noiseColor;
    float constantMfp, noiseMfp;
    v4 constantMu, noiseMu;
    float rcpMajorant, absorption;
    int32_t noiseOctaves;
    float noiseGain, noiseLacunarity, noiseFreq, noiseScale, noiseHeight, noiseRange;
    float phaseDirA, phaseDirB, phaseBlend;
} OrcMedia;

typedef struct { v4 lo, hi; int32_t sx, sy, sz; float cellsPerMeter; } OrcGrid;

typedef struct { float lo[3], hi[3]; int32_t left, right; int32_t first, count; } OrcNode;

struct OrcScene
{
    int32_t vertCount, matCount, texCount, emissiveCount;
    v4* positions; v4* normals; v2* uvs; int32_t* matIds;
    int32_t* vertToEmit; int32_t* emitToVert;
    PimMaterial* materials;
    PimTexture* textures;      /* texels copied */
    int32_t skySize; v3* sky[6];
    int32_t lutSize; v2* lut;
    OrcMedia media;
    /* light grid */
    OrcGrid grid; int32_t gridLen;
    int32_t* cellLength;       /* Dist1D.length per cell (0 or emissiveCount) */
    float* pdf; float* cdf; uint32_t* live; uint32_t* cellSum; float* cellIntegral;
    /* accel */
    int accel;
    OrcNode* nodes; int32_t nodeCount; int32_t* triIndex;
    int32_t bvhDepth;
    uint32_t seed;
    float metersPerCell;
};

typedef struct { u4 rng; uint64_t extRays, shadowRays, otherRays; } OrcCtx;

/* ------------------------------------------------------------------------------------------ */
/* parallel-for (threading/task.c:35-63: dynamic chunks of worksize / nthreads^2)              */

typedef void (*OrcRangeFn)(void* user, int64_t begin, int64_t end, int tid);
typedef struct { OrcRangeFn fn; void* user; int64_t work, gran; int64_t head; int tid; int nthreads; } OrcTask;
typedef struct { OrcTask* task; int tid; } OrcWorker;

static void* orc_worker(void* p)
{
    OrcWorker* w = (OrcWorker*)p;
    OrcTask* t = w->task;
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);          /* task.c:73 */
    _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);  /* task.c:74 */
    for (;;)
    {
        int64_t a = __atomic_fetch_add(&t->head, t->gran, __ATOMIC_RELAXED);
        if (a >= t->work) break;
        int64_t b = a + t->gran; if (b > t->work) b = t->work;
        t->fn(t->user, a, b, w->tid);
    }
    return NULL;
}

static void orc_parallel_for(OrcRangeFn fn, void* user, int64_t work, int threads)
{
    if (work <= 0) return;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    OrcTask task; task.fn = fn; task.user = user; task.work = work; task.head = 0; task.nthreads = threads;
    int64_t gran = work / ((int64_t)threads * threads);   /* task.c:40 */
    task.gran = gran < 1 ? 1 : gran;
    pthread_t th[256]; OrcWorker wk[256];
    for (int i = 1; i < threads; ++i) { wk[i].task = &task; wk[i].tid = i; pthread_create(&th[i], NULL, orc_worker, &wk[i]); }
    unsigned int csr = _mm_getcsr();
    wk[0].task = &task; wk[0].tid = 0; orc_worker(&wk[0]);
    _mm_setcsr(csr);
    for (int i = 1; i < threads; ++i) pthread_join(th[i], NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* watertight closest hit (replaces rtcIntersect1, pt.c:470-486)                   */

typedef struct { int kx, ky, kz; float Sx, Sy, Sz; float o[3]; } OrcRayPre;

static inline void ray_pre(OrcRayPre* r, v4 ro, v4 rd)
{
    float d[3] = { rd.x, rd.y, rd.z };
    int kz = 0;
    if (fabsf(d[1]) > fabsf(d[kz])) kz = 1;
    if (fabsf(d[2]) > fabsf(d[kz])) kz = 2;
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (d[kz] < 0.0f) { int t = kx; kx = ky; ky = t; }
    r->kx = kx; r->ky = ky; r->kz = kz;
    r->Sx = d[kx] / d[kz];
    r->Sy = d[ky] / d[kz];
    r->Sz = 1.0f / d[kz];
    r->o[0] = ro.x; r->o[1] = ro.y; r->o[2] = ro.z;
}

/* returns true and writes t,u,v when the triangle is hit inside (tNear, tFar] */
static inline bool tri_test(const OrcRayPre* r, const v4* p, float tNear, float tFar, float* tOut, float* uOut, float* vOut)
{
    const float a[3] = { p[0].x - r->o[0], p[0].y - r->o[1], p[0].z - r->o[2] };
    const float b[3] = { p[1].x - r->o[0], p[1].y - r->o[1], p[1].z - r->o[2] };
    const float c[3] = { p[2].x - r->o[0], p[2].y - r->o[1], p[2].z - r->o[2] };
    const float Ax = a[r->kx] - r->Sx * a[r->kz], Ay = a[r->ky] - r->Sy * a[r->kz];
    const float Bx = b[r->kx] - r->Sx * b[r->kz], By = b[r->ky] - r->Sy * b[r->kz];
    const float Cx = c[r->kx] - r->Sx * c[r->kz], Cy = c[r->ky] - r->Sy * c[r->kz];
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f)
    {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = U + V + W;
    if (det == 0.0f) return false;
    const float Az = r->Sz * a[r->kz], Bz = r->Sz * b[r->kz], Cz = r->Sz * c[r->kz];
    const float T = U * Az + V * Bz + W * Cz;
    const float t = T / det;
    if (!(t > tNear && t <= tFar)) return false;
    *tOut = t;
    *uOut = V / det;      /* weight of p1 (Embree's u) */
    *vOut = W / det;      /* weight of p2 (Embree's v) */
    return true;
}

typedef struct { int32_t prim; float t, u, v; } OrcHit;

static inline void hit_consider(OrcHit* h, int32_t prim, float t, float u, float v)
{
    if (t < h->t || (t == h->t && (h->prim < 0 || prim < h->prim)))
    {
        h->prim = prim; h->t = t; h->u = u; h->v = v;
    }
}

static OrcHit closest_brute(const OrcScene* s, v4 ro, v4 rd, float tNear, float tFar)
{
    OrcRayPre r; ray_pre(&r, ro, rd);
    OrcHit h; h.prim = -1; h.t = tFar; h.u = h.v = 0.0f;
    const int32_t triCount = s->vertCount / 3;
    for (int32_t i = 0; i < triCount; ++i)
    {
        float t, u, v;
        if (tri_test(&r, s->positions + i * 3, tNear, h.t, &t, &u, &v)) hit_consider(&h, i, t, u, v);
    }
    return h;
}

/* The fp32 triangle test accepts the triangle perturbed by a few ulp of |p - o| (grazing hits, hits a hair
 * behind the origin that round to a positive t), and the contract is "the test over ALL triangles"
 * (closest_brute): a box may only cull what the test cannot accept, so every box is grown per ray by
 * e = 2^-18 * (max|o| + max|scene coordinate|): lo planes are tested from o + e, hi planes from o - e. */
static inline bool box_test(const OrcNode* n, const float op[3], const float om[3], const float inv[3], float tNear, float tFar)
{
    float t0 = tNear, t1 = tFar;
    for (int a = 0; a < 3; ++a)
    {
        float ta = (n->lo[a] - op[a]) * inv[a];
        float tb = (n->hi[a] - om[a]) * inv[a];
        float lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
        hi *= 1.0000004f;                       /* conservative slab (Ize 2013) */
        if (!(lo != lo)) t0 = lo > t0 ? lo : t0;
        if (!(hi != hi)) t1 = hi < t1 ? hi : t1;
    }
    return t0 <= t1;
}

static OrcHit closest_bvh(const OrcScene* s, v4 ro, v4 rd, float tNear, float tFar, bool anyHit)
{
    OrcRayPre r; ray_pre(&r, ro, rd);
    OrcHit h; h.prim = -1; h.t = tFar; h.u = h.v = 0.0f;
    if (s->vertCount < 3) return h;            /* no triangle: the root is a box turned inside out with no children to follow */
    const float inv[3] = { 1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z };
    float sceneMaxAbs = 0.0f;
    for (int a = 0; a < 3; ++a) sceneMaxAbs = fmaxf(sceneMaxAbs, fmaxf(fabsf(s->nodes[0].lo[a]), fabsf(s->nodes[0].hi[a])));
    const float e = 0x1p-18f * (fmaxf(fmaxf(fabsf(ro.x), fabsf(ro.y)), fabsf(ro.z)) + sceneMaxAbs);
    const float op[3] = { ro.x + e, ro.y + e, ro.z + e };
    const float om[3] = { ro.x - e, ro.y - e, ro.z - e };
    int32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    while (sp)
    {
        const OrcNode* n = &s->nodes[stack[--sp]];
        if (!box_test(n, op, om, inv, tNear, h.t)) continue;
        if (n->count)
        {
            for (int32_t k = 0; k < n->count; ++k)
            {
                int32_t prim = s->triIndex[n->first + k];
                float t, u, v;
                if (tri_test(&r, s->positions + prim * 3, tNear, h.t, &t, &u, &v))
                {
                    hit_consider(&h, prim, t, u, v);
                    if (anyHit) return h;
                }
            }
        }
        else
        {
            stack[sp++] = n->left;
            stack[sp++] = n->right;
        }
    }
    return h;
}

static inline OrcHit closest(const OrcScene* s, v4 ro, v4 rd, float tNear, float tFar)
{
    return s->accel ? closest_bvh(s, ro, rd, tNear, tFar, false) : closest_brute(s, ro, rd, tNear, tFar);
}

static inline bool occluded(const OrcScene* s, v4 ro, v4 rd, float tNear, float tFar)
{
    if (s->accel) return closest_bvh(s, ro, rd, tNear, tFar, true).prim >= 0;
    return closest_brute(s, ro, rd, tNear, tFar).prim >= 0;
}

/* ---- binned SAH BVH2 build (the timed CPU baseline accel) ---- */
typedef struct { float lo[3], hi[3], c[3]; } TriBox;
typedef struct { OrcScene* s; TriBox* tb; int32_t depthMax; } BvhBuild;

static void box_reset(float lo[3], float hi[3]) { for (int a = 0; a < 3; ++a) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; } }
static void box_grow(float lo[3], float hi[3], const float blo[3], const float bhi[3]) { for (int a = 0; a < 3; ++a) { if (blo[a] < lo[a]) lo[a] = blo[a]; if (bhi[a] > hi[a]) hi[a] = bhi[a]; } }
static float box_area(const float lo[3], const float hi[3]) { float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return dx * dy + dy * dz + dz * dx; }

static int32_t bvh_build_rec(BvhBuild* b, int32_t first, int32_t count, int32_t depth)
{
    OrcScene* s = b->s;
    int32_t idx = s->nodeCount++;
    OrcNode* n = &s->nodes[idx];
    box_reset(n->lo, n->hi);
    float clo[3], chi[3]; box_reset(clo, chi);
    for (int32_t i = 0; i < count; ++i)
    {
        const TriBox* t = &b->tb[s->triIndex[first + i]];
        box_grow(n->lo, n->hi, t->lo, t->hi);
        box_grow(clo, chi, t->c, t->c);
    }
    if (depth > b->depthMax) b->depthMax = depth;
    n->left = n->right = -1; n->first = first; n->count = count;
    if (count <= 2 || depth >= 120) return idx;
    int axis = 0; float ext = chi[0] - clo[0];
    if (chi[1] - clo[1] > ext) { axis = 1; ext = chi[1] - clo[1]; }
    if (chi[2] - clo[2] > ext) { axis = 2; ext = chi[2] - clo[2]; }
    if (!(ext > 0.0f)) { if (count <= 8) return idx; }
    enum { NB = 16 };
    int32_t mid = first + count / 2;
    if (ext > 0.0f)
    {
        float blo[NB][3], bhi[NB][3]; int32_t bc[NB];
        for (int k = 0; k < NB; ++k) { box_reset(blo[k], bhi[k]); bc[k] = 0; }
        const float scale = (float)NB * 0.9999f / ext;
        for (int32_t i = 0; i < count; ++i)
        {
            const TriBox* t = &b->tb[s->triIndex[first + i]];
            int k = (int)((t->c[axis] - clo[axis]) * scale); if (k < 0) k = 0; if (k >= NB) k = NB - 1;
            bc[k]++; box_grow(blo[k], bhi[k], t->lo, t->hi);
        }
        float rarea[NB]; int32_t rcount[NB];
        float lo[3], hi[3]; box_reset(lo, hi); int32_t c = 0;
        for (int k = NB - 1; k > 0; --k) { box_grow(lo, hi, blo[k], bhi[k]); c += bc[k]; rarea[k] = c ? box_area(lo, hi) : 0.0f; rcount[k] = c; }
        box_reset(lo, hi); c = 0;
        float best = 3.0e38f; int bestK = -1;
        for (int k = 0; k < NB - 1; ++k)
        {
            box_grow(lo, hi, blo[k], bhi[k]); c += bc[k];
            if (!c || !rcount[k + 1]) continue;
            float cost = box_area(lo, hi) * (float)c + rarea[k + 1] * (float)rcount[k + 1];
            if (cost < best) { best = cost; bestK = k; }
        }
        float leafCost = box_area(n->lo, n->hi) * (float)count;
        if (bestK < 0 || (count <= 4 && best >= leafCost)) { if (count <= 8) return idx; }
        else
        {
            int32_t i = first, j = first + count - 1;
            while (i <= j)
            {
                const TriBox* t = &b->tb[s->triIndex[i]];
                int k = (int)((t->c[axis] - clo[axis]) * scale); if (k < 0) k = 0; if (k >= NB) k = NB - 1;
                if (k <= bestK) ++i; else { int32_t tmp = s->triIndex[i]; s->triIndex[i] = s->triIndex[j]; s->triIndex[j] = tmp; --j; }
            }
            mid = i;
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
    }
    int32_t l = bvh_build_rec(b, first, mid - first, depth + 1);
    int32_t r = bvh_build_rec(b, mid, first + count - mid, depth + 1);
    n = &s->nodes[idx];
    n->left = l; n->right = r; n->count = 0;
    return idx;
}

static void bvh_build(OrcScene* s)
{
    const int32_t triCount = s->vertCount / 3;
    s->nodes = (OrcNode*)malloc(sizeof(OrcNode) * (size_t)(2 * triCount + 1));
    s->triIndex = (int32_t*)malloc(sizeof(int32_t) * (size_t)(triCount + 1));
    s->nodeCount = 0;
    if (!triCount) { s->nodeCount = 1; memset(&s->nodes[0], 0, sizeof(OrcNode)); s->nodes[0].left = s->nodes[0].right = -1; box_reset(s->nodes[0].lo, s->nodes[0].hi); return; }
    TriBox* tb = (TriBox*)malloc(sizeof(TriBox) * (size_t)triCount);
    for (int32_t i = 0; i < triCount; ++i)
    {
        box_reset(tb[i].lo, tb[i].hi);
        for (int k = 0; k < 3; ++k)
        {
            v4 p = s->positions[i * 3 + k];
            float q[3] = { p.x, p.y, p.z };
            box_grow(tb[i].lo, tb[i].hi, q, q);
        }
        for (int a = 0; a < 3; ++a)
        {
            float m = fmaxf(fmaxf(fabsf(tb[i].lo[a]), fabsf(tb[i].hi[a])), 1e-3f);
            float pad = m * 4.0e-7f;
            tb[i].c[a] = 0.5f * (tb[i].lo[a] + tb[i].hi[a]);
            tb[i].lo[a] -= pad; tb[i].hi[a] += pad;
        }
        s->triIndex[i] = i;
    }
    BvhBuild b; b.s = s; b.tb = tb; b.depthMax = 0;
    bvh_build_rec(&b, 0, triCount, 0);
    s->bvhDepth = b.depthMax;
    free(tb);
}

/* ------------------------------------------------------------------------------------------ */
/* pt_intersect_local, pt.c:1421-1464                                                          */

static PimRayHit make_rayhit(const OrcScene* s, v4 rd, OrcHit h)
{
    PimRayHit hit; memset(&hit, 0, sizeof(hit));
    hit.wuvt[3] = -1.0f;
    hit.iVert = -1;
    bool hitNothing = (h.prim < 0) || (h.t <= 0.0f);
    if (hitNothing)
    {
        hit.type = PIM_HIT_NOTHING;
        return hit;
    }
    const v4* p = s->positions + h.prim * 3;
    v4 e1 = o4_sub(p[1], p[0]), e2 = o4_sub(p[2], p[0]);
    e1.w = e2.w = 0.0f;
    v4 Ng = o4_cross3(e1, e2);                 /* Embree Ng, unnormalised, CCW front */
    hit.type = PIM_HIT_TRIANGLE;
    if (o4_dot3(Ng, rd) > 0.0f)
    {
        Ng = o4_neg(Ng);
        hit.type = PIM_HIT_BACKFACE;
    }
    Ng = o4_normalize3(Ng);
    hit.normal[0] = Ng.x; hit.normal[1] = Ng.y; hit.normal[2] = Ng.z; hit.normal[3] = 0.0f;
    float u = o_sat(h.u);
    float v = o_sat(h.v);
    float w = o_sat(1.0f - (u + v));
    hit.iVert = h.prim * 3;
    hit.wuvt[0] = w; hit.wuvt[1] = u; hit.wuvt[2] = v; hit.wuvt[3] = h.t;
    hit.flags = s->materials[s->matIds[hit.iVert]].flags;
    return hit;
}

static inline v4 hit_wuvt(const PimRayHit* h) { return o4(h->wuvt[0], h->wuvt[1], h->wuvt[2], h->wuvt[3]); }
static inline v4 hit_normal(const PimRayHit* h) { return o4(h->normal[0], h->normal[1], h->normal[2], h->normal[3]); }

static PimRayHit pt_intersect(const OrcScene* s, v4 ro, v4 rd, float tNear, float tFar)
{
    return make_rayhit(s, rd, closest(s, ro, rd, tNear, tFar));
}

/* ------------------------------------------------------------------------------------------ */
/* Dist1D, math/dist1d.c:33-163 on raw arrays                                                  */

static float dist_bake(float* pdf, float* cdf, int32_t pdfLen)                 /* :33-73 */
{
    const int32_t cdfLen = pdfLen + 1;
    if (pdfLen <= 0) return 0.0f;
    const float rcpLen = 1.0f / pdfLen;
    cdf[0] = 0.0f;
    for (int32_t i = 1; i < cdfLen; ++i) cdf[i] = cdf[i - 1] + pdf[i - 1] * rcpLen;
    float integral = cdf[cdfLen - 1];
    if (integral == 0.0f)
    {
        for (int32_t i = 1; i < cdfLen; ++i) cdf[i] = i * rcpLen;
    }
    else
    {
        float rcpIntegral = 1.0f / integral;
        for (int32_t i = 1; i < cdfLen; ++i) cdf[i] = cdf[i] * rcpIntegral;
        for (int32_t i = 0; i < pdfLen; ++i) pdf[i] = pdf[i] * rcpIntegral;
    }
    return integral;
}

static int32_t dist_find_interval(const float* cdf, int32_t size, float u)     /* :75-94 */
{
    int32_t first = 0, len = size;
    while (len > 0)
    {
        int32_t half = len >> 1;
        int32_t middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len = len - (half + 1); }
        else len = half;
    }
    return oi_clamp(first - 1, 0, size - 2);
}

static void dist_update(float* pdf, float* cdf, uint32_t* live, int32_t pdfLen, uint32_t* sumIo, float* integralIo)   /* :128-163 */
{
    if (pdfLen <= 0) return;
    uint32_t sum = 0;
    for (int32_t i = 0; i < pdfLen; ++i) sum += live[i];
    if (sum < 30) return;
    const float scale = 1.0f / sum;
    const uint32_t prevSum = *sumIo;
    *sumIo = sum;
    float alpha = 0.5f;
    if (prevSum > 0)
    {
        double ratio = (double)sum / (double)prevSum;
        alpha = o_sat((float)ratio) * 0.9f;
        alpha = alpha * alpha;
    }
    for (int32_t i = 0; i < pdfLen; ++i)
    {
        uint32_t ct = live[i];
        pdf[i] = o_lerp(pdf[i], ct * scale, alpha);
        live[i] = (ct >> 1);
    }
    *integralIo = dist_bake(pdf, cdf, pdfLen);
}

/* Grid, math/grid.h:8-49 */
static void grid_new(OrcGrid* g, v4 lo, v4 hi, float cellsPerMeter)
{
    v4 range = o4_sub(hi, lo);
    v4 sizef = o4_ceil(o4_muls(range, cellsPerMeter));
    g->lo = lo; g->hi = hi;
    g->sx = (int32_t)sizef.x; g->sy = (int32_t)sizef.y; g->sz = (int32_t)sizef.z;
    /* BUILD-DEFINED: a scene flat along an axis would get a zero-sized grid, which the reference indexes out of
     * bounds; keep at least one cell per axis (same rule in the HIP path) */
    if (g->sx < 1) g->sx = 1;
    if (g->sy < 1) g->sy = 1;
    if (g->sz < 1) g->sz = 1;
    g->cellsPerMeter = cellsPerMeter;
}
static v4 grid_position(const OrcGrid* g, int32_t index)
{
    const float metersPerCell = 1.0f / g->cellsPerMeter;
    int32_t ix = index % g->sx;
    int32_t iy = (index / g->sx) % g->sy;
    int32_t iz = index / (g->sx * g->sy);
    v4 offset = o4((ix + 0.5f) * metersPerCell, (iy + 0.5f) * metersPerCell, (iz + 0.5f) * metersPerCell, 0.0f);
    return o4_add(g->lo, offset);
}
static int32_t grid_index(const OrcGrid* g, v4 position)
{
    v4 offset = o4_muls(o4_sub(position, g->lo), g->cellsPerMeter);
    int32_t x = oi_clamp((int32_t)offset.x, 0, g->sx - 1);
    int32_t y = oi_clamp((int32_t)offset.y, 0, g->sy - 1);
    int32_t z = oi_clamp((int32_t)offset.z, 0, g->sz - 1);
    return x + y * g->sx + z * g->sx * g->sy;
}

/* ------------------------------------------------------------------------------------------ */
/* media desc, pt.c:1944-1987, 2183-2196                                                        */

static v4 CalcMajorant(const OrcMedia* d)
{
    float a = 1.0f + d->absorption;
    v4 ca = o4_muls(d->constantMu, a);
    v4 na = o4_muls(d->noiseMu, a);
    return o4_muls(o4_add(ca, na), 2.0f);
}

static void media_from_desc(OrcMedia* m, const PimMediaDesc* d)
{
    m->constantColor = o4(d->constantColor[0], d->constantColor[1], d->constantColor[2], d->constantColor[3]);
    m->noiseColor = o4(d->noiseColor[0], d->noiseColor[1], d->noiseColor[2], d->noiseColor[3]);
    m->constantMfp = d->constantMfp; m->noiseMfp = d->noiseMfp; m->absorption = d->absorption;
    m->noiseOctaves = d->noiseOctaves; m->noiseGain = d->noiseGain; m->noiseLacunarity = d->noiseLacunarity;
    m->noiseFreq = d->noiseFreq; m->noiseScale = d->noiseScale; m->noiseHeight = d->noiseHeight;
    m->phaseDirA = d->phaseDirA; m->phaseDirB = d->phaseDirB; m->phaseBlend = d->phaseBlend;
    /* media_desc_update, pt.c:1963-1987 */
    v4 constantMfp = o4_lerpsv(m->constantMfp * 0.5f, m->constantMfp * 2.0f, m->constantColor);
    v4 noiseMfp = o4_lerpsv(m->noiseMfp * 0.5f, m->noiseMfp * 2.0f, m->noiseColor);
    m->constantMu = o4_rcp(constantMfp);
    m->noiseMu = o4_rcp(noiseMfp);
    m->phaseDirA = o_clamp(m->phaseDirA, -0.99f, 0.99f);
    m->phaseDirB = o_clamp(m->phaseDirB, -0.99f, 0.99f);
    m->phaseBlend = o_sat(m->phaseBlend);
    float sum = 0.0f, amplitude = 1.0f;
    for (int32_t i = 0; i < m->noiseOctaves; ++i) { sum += amplitude; amplitude *= m->noiseGain; }
    m->noiseRange = sum * m->noiseScale * 1.5f;
    m->rcpMajorant = 1.0f / o4_hmax3(CalcMajorant(m));
}

/* ------------------------------------------------------------------------------------------ */
/* scene set-up                                                                                */

/* EmissionPdf, pt.c:784-822 */
static float EmissionPdf(const OrcScene* s, int32_t iVert, int32_t attempts, u4* rng)
{
    const PimMaterial* mat = &s->materials[s->matIds[iVert]];
    if (mat->flags & PIM_MATFLAG_SKY) return 1.0f;
    if (mat->rome < 0) return 0.0f;
    const PimTexture* rome = &s->textures[mat->rome];
    const uint32_t* texels = (const uint32_t*)rome->texels;
    const v2 UA = s->uvs[iVert + 0], UB = s->uvs[iVert + 1], UC = s->uvs[iVert + 2];
    int32_t hits = 0;
    for (int32_t i = 0; i < attempts; ++i)
    {
        v4 wuv = SampleBaryCoord(rng_f2(rng));
        v2 uv = o2_blend(UA, UB, UC, wuv);
        int32_t cx = PointWrap1D(rome->width, uv.x), cy = PointWrap1D(rome->height, uv.y);
        uint32_t sample = texels[rome->width * cy + cx] >> 24;
        hits += sample ? 1 : 0;
    }
    return (float)hits / (float)attempts;
}

typedef struct { OrcScene* s; float* pdfs; int32_t attempts; } EmitTask;
static void emit_fn(void* user, int64_t begin, int64_t end, int tid)
{
    (void)tid;
    EmitTask* t = (EmitTask*)user;
    for (int64_t i = begin; i < end; ++i)
    {
        u4 rng = pcg4((uint32_t)i, 0u, t->s->seed, TAG_EMIT);
        t->pdfs[i] = EmissionPdf(t->s, (int32_t)i * 3, t->attempts, &rng);
    }
}

/* SetupEmissives, pt.c:845-883 */
static void SetupEmissives(OrcScene* s, int32_t attempts, int threads)
{
    const int32_t triCount = s->vertCount / 3;
    float* pdfs = (float*)malloc(sizeof(float) * (size_t)(triCount + 1));
    EmitTask t = { s, pdfs, attempts };
    orc_parallel_for(emit_fn, &t, triCount, threads);
    s->vertToEmit = (int32_t*)malloc(sizeof(int32_t) * (size_t)(s->vertCount + 1));
    s->emitToVert = (int32_t*)malloc(sizeof(int32_t) * (size_t)(triCount + 1));
    int32_t emissiveCount = 0;
    for (int32_t iTri = 0; iTri < triCount; ++iTri)
    {
        int32_t iVert = iTri * 3;
        s->vertToEmit[iVert + 0] = s->vertToEmit[iVert + 1] = s->vertToEmit[iVert + 2] = -1;
        if (pdfs[iTri] > 0.01f)
        {
            s->vertToEmit[iVert + 0] = s->vertToEmit[iVert + 1] = s->vertToEmit[iVert + 2] = emissiveCount;
            s->emitToVert[emissiveCount++] = iVert;
        }
    }
    s->emissiveCount = emissiveCount;
    free(pdfs);
}

/* RtcPointQuery + RtcPointQueryFn, pt.c:555-616: nearest triangle inside the query radius.
 * The answer is min{ d(tri) : d(tri) < radius }, which does not depend on the visiting order, so the
 * walk over the BVH (accel = 1: boxes farther than a slightly enlarged radius are skipped) and the
 * loop over all triangles (accel = 0) return the same float. */
static inline float point_tri_distance(const OrcScene* s, int32_t i, v4 pt)
{
    const v4* p = s->positions + i * 3;
    return o_abs(sdTriangle3D(p[0], p[1], p[2], pt));
}

static float point_query_distance(const OrcScene* s, v4 pt)
{
    float best = ORC_RCP_EPS;
    const float radius = pt.w;
    const int32_t triCount = s->vertCount / 3;
    if (s->accel && s->nodes)
    {
        const float prune = radius * 1.001f + 1e-6f;
        const float prune2 = prune * prune;
        const float q[3] = { pt.x, pt.y, pt.z };
        int32_t stack[128]; int sp = 0;
        stack[sp++] = 0;
        while (sp)
        {
            const OrcNode* n = &s->nodes[stack[--sp]];
            float d2 = 0.0f;
            for (int a = 0; a < 3; ++a)
            {
                float e = fmaxf(fmaxf(n->lo[a] - q[a], q[a] - n->hi[a]), 0.0f);
                d2 += e * e;
            }
            if (!(d2 <= prune2)) continue;
            if (n->count)
            {
                for (int32_t k = 0; k < n->count; ++k)
                {
                    float d = point_tri_distance(s, s->triIndex[n->first + k], pt);
                    if (d < radius && d < best) best = d;
                }
            }
            else if (n->left >= 0) { stack[sp++] = n->left; stack[sp++] = n->right; }
        }
        return best;
    }
    for (int32_t i = 0; i < triCount; ++i)
    {
        float d = point_tri_distance(s, i, pt);
        if (d < radius && d < best) best = d;
    }
    return best;
}

typedef struct { OrcScene* s; } GridTask;
/* SetupLightGridFn, pt.c:891-990 */
static void grid_fn(void* user, int64_t begin, int64_t end, int tid)
{
    (void)tid;
    OrcScene* s = ((GridTask*)user)->s;
    const int32_t E = s->emissiveCount;
    const float radius = s->metersPerCell * 0.666f;
    v4 hamm[16];
    for (int i = 0; i < 16; ++i) hamm[i] = SampleUnitSphere(Hammersley2D((uint32_t)i, 16u));
    for (int64_t iCell = begin; iCell < end; ++iCell)
    {
        u4 rng = pcg4((uint32_t)iCell, 1u, s->seed, TAG_GRID);
        s->cellLength[iCell] = 0;
        v4 position = grid_position(&s->grid, (int32_t)iCell);
        position.w = radius + 0.01f * ORC_MILLI;
        float qd = point_query_distance(s, position);
        if (qd > radius)
        {
            float hitcount = 0.0f;
            for (int j = 0; j < 16; ++j)
            {
                PimRayHit hit = pt_intersect(s, position, hamm[j], 0.0f, ORC_RCP_EPS);
                if (hit.type == PIM_HIT_TRIANGLE) ++hitcount;
            }
            float hitratio = hitcount / 16;
            if (hitratio < 0.5f) continue;
        }
        if (E <= 0) continue;            /* Dist1D_New(length 0) leaves length 0, dist1d.c:7-20 */
        float* pdf = s->pdf + (size_t)iCell * E;
        float* cdf = s->cdf + (size_t)iCell * (E + 1);
        for (int32_t iEmit = 0; iEmit < E; ++iEmit)
        {
            int32_t iVert = s->emitToVert[iEmit];
            v4 A = s->positions[iVert + 0], B = s->positions[iVert + 1], C = s->positions[iVert + 2];
            int32_t hits = 0;
            for (int k = 0; k < 16; ++k)
            {
                v4 t = o4_muls(o4_lerpsv(-1.5f, 1.5f, rng_f4(&rng)), radius);
                v4 ro = o4_add(position, t);
                ro.w = 0.0f;
                v4 at = o4_blend(A, B, C, SampleBaryCoord(rng_f2(&rng)));
                v4 rd = o4_sub(at, ro);
                float dist = o4_length3(rd);
                rd = o4_divs(rd, dist);
                float tFar = dist - 0.01f * ORC_MILLI;
                bool visible = !occluded(s, ro, rd, 0.0f, tFar);
                hits += visible ? 1 : 0;
            }
            pdf[iEmit] = (float)hits / 16.0f;
        }
        s->cellIntegral[iCell] = dist_bake(pdf, cdf, E);
        s->cellLength[iCell] = E;
    }
}

/* SetupLightGrid, pt.c:992-1009; box_from_pts math/box.h:53-68 */
static void SetupLightGrid(OrcScene* s, int threads)
{
    if (s->vertCount <= 0) { s->gridLen = 0; return; }
    v4 lo = o4_s((float)(1 << 23)), hi = o4_s(-(float)(1 << 23));
    for (int32_t i = 0; i < s->vertCount; ++i) { lo = o4_min(lo, s->positions[i]); hi = o4_max(hi, s->positions[i]); }
    grid_new(&s->grid, lo, hi, 1.0f / s->metersPerCell);
    const int32_t len = s->grid.sx * s->grid.sy * s->grid.sz;
    const int32_t E = s->emissiveCount;
    s->gridLen = len;
    s->cellLength = (int32_t*)calloc((size_t)len + 1, sizeof(int32_t));
    s->pdf = (float*)calloc((size_t)len * (E > 0 ? E : 1) + 1, sizeof(float));
    s->cdf = (float*)calloc((size_t)len * (E + 1) + 1, sizeof(float));
    s->live = (uint32_t*)calloc((size_t)len * (E > 0 ? E : 1) + 1, sizeof(uint32_t));
    s->cellSum = (uint32_t*)calloc((size_t)len + 1, sizeof(uint32_t));
    s->cellIntegral = (float*)calloc((size_t)len + 1, sizeof(float));
    GridTask t = { s };
    orc_parallel_for(grid_fn, &t, len, threads);
}

OrcScene* orc_scene_create(const PimSceneDesc* d, int accel, int threads)
{
    if (!d || d->vertCount < 0 || (d->vertCount % 3) != 0) return NULL;
    OrcScene* s = (OrcScene*)calloc(1, sizeof(OrcScene));
    const size_t V = (size_t)d->vertCount;
    s->vertCount = d->vertCount; s->matCount = d->matCount; s->texCount = d->texCount;
    s->positions = (v4*)malloc(sizeof(v4) * (V + 1)); memcpy(s->positions, d->positions, sizeof(v4) * V);
    s->normals = (v4*)malloc(sizeof(v4) * (V + 1)); memcpy(s->normals, d->normals, sizeof(v4) * V);
    s->uvs = (v2*)malloc(sizeof(v2) * (V + 1)); memcpy(s->uvs, d->uvs, sizeof(v2) * V);
    s->matIds = (int32_t*)malloc(sizeof(int32_t) * (V + 1)); memcpy(s->matIds, d->matIds, sizeof(int32_t) * V);
    s->materials = (PimMaterial*)malloc(sizeof(PimMaterial) * (size_t)(d->matCount + 1)); memcpy(s->materials, d->materials, sizeof(PimMaterial) * (size_t)d->matCount);
    s->textures = (PimTexture*)calloc((size_t)d->texCount + 1, sizeof(PimTexture));
    for (int32_t i = 0; i < d->texCount; ++i)
    {
        s->textures[i] = d->textures[i];
        size_t bytes = (size_t)d->textures[i].width * d->textures[i].height * 4;
        void* p = malloc(bytes); memcpy(p, d->textures[i].texels, bytes);
        s->textures[i].texels = p;
    }
    s->skySize = d->skySize;
    if (d->skySize > 0)
    {
        size_t n = (size_t)d->skySize * d->skySize;
        for (int f = 0; f < 6; ++f) { s->sky[f] = (v3*)malloc(sizeof(v3) * n); memcpy(s->sky[f], d->skyFaces[f], sizeof(v3) * n); }
    }
    s->lutSize = d->brdfLutSize;
    if (d->brdfLutSize > 0)
    {
        size_t n = (size_t)d->brdfLutSize * d->brdfLutSize;
        s->lut = (v2*)malloc(sizeof(v2) * n); memcpy(s->lut, d->brdfLut, sizeof(v2) * n);
    }
    s->seed = d->seed;
    s->metersPerCell = d->metersPerCell > 0.0f ? d->metersPerCell : 1.5f;
    media_from_desc(&s->media, &d->media);
    /* PtScene_Init order, pt.c:1039-1049 */
    SetupEmissives(s, d->emissiveAttempts > 0 ? d->emissiveAttempts : 1000, threads);
    s->accel = accel;
    if (accel) bvh_build(s);
    SetupLightGrid(s, threads);
    return s;
}

void orc_scene_destroy(OrcScene* s)
{
    if (!s) return;
    free(s->positions); free(s->normals); free(s->uvs); free(s->matIds);
    free(s->vertToEmit); free(s->emitToVert); free(s->materials);
    for (int32_t i = 0; i < s->texCount; ++i) free((void*)s->textures[i].texels);
    free(s->textures);
    for (int f = 0; f < 6; ++f) free(s->sky[f]);
    free(s->lut);
    free(s->cellLength); free(s->pdf); free(s->cdf); free(s->live); free(s->cellSum); free(s->cellIntegral);
    free(s->nodes); free(s->triIndex);
    free(s);
}

void orc_scene_info(const OrcScene* s, PimSceneInfo* out)
{
    memset(out, 0, sizeof(*out));
    out->vertCount = s->vertCount; out->triCount = s->vertCount / 3; out->matCount = s->matCount;
    out->emissiveCount = s->emissiveCount;
    out->gridSize[0] = s->grid.sx; out->gridSize[1] = s->grid.sy; out->gridSize[2] = s->grid.sz;
    out->gridLen = s->gridLen;
    for (int32_t i = 0; i < s->gridLen; ++i) out->activeCells += s->cellLength[i] ? 1 : 0;
    out->bvhNodeCount = s->nodeCount; out->bvhMaxDepth = s->bvhDepth;
    out->boundsLo[0] = s->grid.lo.x; out->boundsLo[1] = s->grid.lo.y; out->boundsLo[2] = s->grid.lo.z;
    out->boundsHi[0] = s->grid.hi.x; out->boundsHi[1] = s->grid.hi.y; out->boundsHi[2] = s->grid.hi.z;
}

void orc_scene_download_lights(const OrcScene* s, int32_t* vertToEmit, int32_t* emitToVert,
    int32_t* cellLength, float* pdf, float* cdf, uint32_t* live)
{
    const size_t E = (size_t)s->emissiveCount;
    if (vertToEmit) memcpy(vertToEmit, s->vertToEmit, sizeof(int32_t) * (size_t)s->vertCount);
    if (emitToVert) memcpy(emitToVert, s->emitToVert, sizeof(int32_t) * E);
    if (cellLength) memcpy(cellLength, s->cellLength, sizeof(int32_t) * (size_t)s->gridLen);
    if (pdf) memcpy(pdf, s->pdf, sizeof(float) * (size_t)s->gridLen * E);
    if (cdf) memcpy(cdf, s->cdf, sizeof(float) * (size_t)s->gridLen * (E + 1));
    if (live) memcpy(live, s->live, sizeof(uint32_t) * (size_t)s->gridLen * E);
}

/* test plumbing of the one-process-per-GPU form: the light histograms as a rank's all-reduce leaves them */
void orc_scene_set_live(OrcScene* s, const uint32_t* live)
{
    memcpy(s->live, live, sizeof(uint32_t) * (size_t)s->gridLen * (size_t)s->emissiveCount);
}

/* UpdateDists, pt.c:2460-2483 */
void orc_scene_update_dists(OrcScene* s)
{
    const int32_t E = s->emissiveCount;
    for (int32_t i = 0; i < s->gridLen; ++i)
    {
        if (!s->cellLength[i]) continue;
        dist_update(s->pdf + (size_t)i * E, s->cdf + (size_t)i * (E + 1), s->live + (size_t)i * E, E, &s->cellSum[i], &s->cellIntegral[i]);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* surface fetch, pt.c:1180-1419                                                                */

typedef struct
{
    v4 P, M, N, albedo, emission, meanFreePath;
    float roughness, occlusion, metallic, ior;
    uint32_t flags;
    int32_t type;
} SurfHit;

typedef struct { v4 pos, dir, attenuation, luminance; float pdf; } Scatter;
typedef struct { v4 direction, luminance, wuvt; float pdf; } LightSample;
typedef struct { v4 scattering, extinction; float g0, g1, gBlend; } Media;

static inline float Sample1D(OrcCtx* c) { return rng_f32(&c->rng); }     /* pt.c:377 */
static inline v2 Sample2D(OrcCtx* c) { return rng_f2(&c->rng); }          /* pt.c:382 */

static v4 GetSky(const OrcScene* s, v4 rd)                                 /* pt.c:1280-1291, cubemap.h:117-129 */
{
    if (s->skySize > 0)
    {
        v2 uv; int face = Cubemap_CalcUv(rd, &uv);
        v3 c = UvBilinearClamp_f3(s->sky[face], s->skySize, s->skySize, uv);
        return o4(c.x, c.y, c.z, 1.0f);
    }
    return o4_s(0.0f);
}

static v2 GetUV(const OrcScene* s, const PimRayHit* hit) { return o2_blend(s->uvs[hit->iVert + 0], s->uvs[hit->iVert + 1], s->uvs[hit->iVert + 2], hit_wuvt(hit)); }   /* :1206 */
static v4 GetPosition(const OrcScene* s, const PimRayHit* hit) { return o4_blend(s->positions[hit->iVert + 0], s->positions[hit->iVert + 1], s->positions[hit->iVert + 2], hit_wuvt(hit)); }   /* :1180 */
static v4 GetNormal(const OrcScene* s, const PimRayHit* hit)                /* :1192-1204 */
{
    v4 N = o4_blend(s->normals[hit->iVert + 0], s->normals[hit->iVert + 1], s->normals[hit->iVert + 2], hit_wuvt(hit));
    N = (o4_dot3(hit_normal(hit), N) > 0.0f) ? N : o4_neg(N);
    return o4_normalize3(N);
}
static float GetArea(const OrcScene* s, int32_t iVert) { return TriArea3D(s->positions[iVert + 0], s->positions[iVert + 1], s->positions[iVert + 2]); }   /* :1218 */

static v4 tex_c32(const OrcScene* s, int32_t id, v2 uv, v4 dflt)
{
    if (id < 0) return dflt;
    const PimTexture* t = &s->textures[id];
    return UvBilinearWrap_c32((const uint32_t*)t->texels, t->width, t->height, uv);
}

static v4 GetEmission(const OrcScene* s, v4 rd, const PimRayHit* hit)       /* pt.c:1293-1326 */
{
    if (hit->flags & PIM_MATFLAG_SKY) return GetSky(s, rd);
    const PimMaterial* mat = &s->materials[s->matIds[hit->iVert]];
    v2 uv = GetUV(s, hit);
    v4 albedo = tex_c32(s, mat->albedo, uv, o4_s(1.0f));
    float e = 0.0f;
    if (mat->rome >= 0) e = tex_c32(s, mat->rome, uv, o4_s(0.0f)).w;
    return UnpackEmission(albedo, e);
}

static v4 FixShadingNormal(v4 M, v4 N) { return (o4_dot3(M, N) > 0.0f) ? N : o4_reflect3(N, M); }   /* :1356 */

static SurfHit GetSurface(const OrcScene* s, v4 rd, const PimRayHit* hit)   /* pt.c:1377-1419 */
{
    SurfHit surf; memset(&surf, 0, sizeof(surf));
    surf.type = hit->type;
    const PimMaterial* mat = &s->materials[s->matIds[hit->iVert]];
    surf.flags = mat->flags;
    surf.ior = mat->ior;
    v2 uv = GetUV(s, hit);
    surf.M = GetNormal(s, hit);
    surf.N = surf.M;
    surf.P = GetPosition(s, hit);
    surf.P = o4_add(surf.P, o4_muls(surf.M, 0.01f * ORC_MILLI));
    surf.meanFreePath = o4(mat->meanFreePath[0], mat->meanFreePath[1], mat->meanFreePath[2], mat->meanFreePath[3]);
    if (hit->flags & PIM_MATFLAG_SKY)
    {
        surf.albedo = o4_s(0.0f);
        surf.roughness = 1.0f; surf.occlusion = 0.0f; surf.metallic = 0.0f;
        surf.emission = GetSky(s, rd);
        surf.M = o4_neg(rd);
        surf.N = surf.M;
        surf.ior = 1.0f;
    }
    else
    {
        if (mat->normal >= 0)        /* SampleNormal :1363-1375 */
        {
            const PimTexture* t = &s->textures[mat->normal];
            v4 Nts = UvBilinearWrap_xy16((const uint32_t*)t->texels, t->width, t->height, uv);
            surf.N = FixShadingNormal(surf.N, TanToWorld(surf.N, Nts));
        }
        surf.albedo = tex_c32(s, mat->albedo, uv, o4_s(1.0f));                       /* SampleAlbedo :1328 */
        v4 rome = tex_c32(s, mat->rome, uv, o4(0.5f, 1.0f, 0.0f, 0.0f));             /* SampleRome :1341 */
        surf.emission = UnpackEmission(surf.albedo, rome.w);
        surf.roughness = rome.x; surf.occlusion = rome.y; surf.metallic = rome.z;
    }
    return surf;
}

/* ------------------------------------------------------------------------------------------ */
/* BRDF evaluation, pt.c:1476-1707                                                              */

static v4 GGXEnergyCompensation(const OrcScene* s, v4 f0, float NoV, float alpha)   /* lighting.h:294-303 */
{
    return GGXEnergyCompensationLut(s->lut, s->lutSize, s->lutSize, f0, NoV, alpha);
}

static v4 Eval_Diffuse(const SurfHit* surf, v4 I, v4 L)                     /* :1476-1497 */
{
    v4 result = o4_s(0.0f);
    v4 N = surf->N;
    float NoL = o4_dot3(N, L);
    float pdf = LambertPdf(NoL);
    if (pdf > ORC_EPS)
    {
        v4 V = o4_neg(I);
        v4 H = o4_normalize3(o4_add(V, L));
        float HoV = o4_dotsat(H, V);
        float NoV = o4_dotsat(N, V);
        v4 Fd = o4_muls(surf->albedo, Fd_Burley(NoL, NoV, HoV, surf->roughness));
        result = o4_muls(Fd, NoL);
        result.w = pdf;
    }
    return result;
}

static v4 Eval_Specular(const OrcScene* s, const SurfHit* surf, v4 I, v4 L)   /* :1516-1548 */
{
    v4 result = o4_s(0.0f);
    v4 N = surf->N;
    float NoL = o4_dot3(N, L);
    if (NoL > ORC_EPS)
    {
        float alpha = BrdfAlpha(surf->roughness);
        v4 V = o4_neg(I);
        v4 H = o4_normalize3(o4_add(V, L));
        float NoH = o4_dot3(N, H);
        float HoV = o4_dot3(H, V);
        float pdf = GGXPdf(NoH, HoV, alpha);
        if (pdf > ORC_EPS)
        {
            float NoV = o4_dotsat(N, V);
            float f = F_Dielectric(HoV, 1.0f, 1.5f);
            v4 f0 = F_0(surf->albedo, surf->metallic);
            v4 F = o4_lerps(f0, o4_s(F_90(f0)), f);
            float D = D_GTR(NoH, alpha);
            float G = V_SmithCorrelated(NoL, NoV, alpha);
            v4 Fr = o4_muls(F, D * G);
            Fr = o4_mul(Fr, GGXEnergyCompensation(s, f0, NoV, alpha));
            result = o4_muls(Fr, NoL);
            result.w = pdf;
        }
    }
    return result;
}

static Scatter Scatter_Diffuse(OrcCtx* ctx, const SurfHit* surf, v4 I)     /* :1499-1514 */
{
    Scatter r; memset(&r, 0, sizeof(r));
    v4 L = TanToWorld(surf->N, SampleCosineHemisphere(Sample2D(ctx)));
    v4 attenPdf = Eval_Diffuse(surf, I, L);
    r.dir = L; r.attenuation = attenPdf; r.pdf = attenPdf.w; r.pos = surf->P;
    return r;
}

static Scatter Scatter_Specular(OrcCtx* ctx, const OrcScene* s, const SurfHit* surf, v4 I)   /* :1550-1565 */
{
    Scatter r; memset(&r, 0, sizeof(r));
    v4 m = SampleGGXMicrofacet(Sample2D(ctx), BrdfAlpha(surf->roughness));
    m = TanToWorld(surf->N, m);
    m = FixShadingNormal(surf->M, m);
    r.dir = o4_reflect3(I, m);
    r.attenuation = Eval_Specular(s, surf, I, r.dir);
    r.pdf = r.attenuation.w;
    r.pos = surf->P;
    return r;
}

static Scatter Scatter_Refractive(OrcCtx* ctx, const OrcScene* s, const SurfHit* surf, v4 I)   /* :1576-1638 */
{
    const float etaI = 1.000277f;
    const float etaT = o_max(1.0f, surf->ior);
    const float alpha = BrdfAlpha(surf->roughness);
    v4 V = o4_neg(I);
    v4 M = surf->M;
    v4 m = TanToWorld(surf->N, SampleGGXMicrofacet(Sample2D(ctx), alpha));
    m = FixShadingNormal(M, m);
    bool entering = surf->type != PIM_HIT_BACKFACE;
    float cosThetaI = o_sat(o_abs(o4_dot3(V, m)));
    float pdf = F_Dielectric(entering ? cosThetaI : -cosThetaI, etaI, etaT);
    v4 L;
    bool refract = false;
    if (Sample1D(ctx) < pdf)
    {
        L = o4_reflect3(I, m);
    }
    else
    {
        float k = entering ? etaI / etaT : etaT / etaI;
        L = o4_refract3(I, m, k);
        pdf = 1.0f - pdf;
        refract = true;
    }
    v4 P = surf->P;
    if (o4_dot3(L, M) < 0.0f) P = o4_sub(P, o4_muls(M, ORC_MILLI * 0.1f));
    Scatter r; memset(&r, 0, sizeof(r));
    r.pos = P; r.dir = L; r.pdf = pdf;
    if (refract && entering)
    {
        ctx->otherRays++;
        PimRayHit hit = pt_intersect(s, P, L, 0.0f, ORC_RCP_EPS);
        float thickness = o_max(hit.wuvt[3], ORC_EPS);
        if (hit.type == PIM_HIT_NOTHING) thickness = ORC_RCP_EPS;
        v4 tr = AlbedoToTransmittance(surf->albedo, surf->roughness, thickness);
        r.attenuation = o4_muls(tr, pdf);
    }
    else
    {
        r.attenuation = o4_s(pdf);
    }
    return r;
}

static v4 Eval_Principled(const OrcScene* s, const SurfHit* surf, v4 I, v4 L)   /* :1641-1668 */
{
    if (surf->flags & PIM_MATFLAG_REFRACTIVE) return o4_s(0.0f);
    if (!(o4_dot3(surf->N, L) > ORC_EPS)) return o4_s(0.0f);
    float amtA = o_lerp(0.5f, 1.0f, surf->metallic);
    float amtB = 1.0f - amtA;
    v4 evalA = Eval_Specular(s, surf, I, L);
    if (amtB > ORC_EPS)
    {
        v4 evalB = Eval_Diffuse(surf, I, L);
        evalA = o4_lerps(evalA, evalB, amtB);
    }
    return evalA;
}

static Scatter Scatter_Principled(OrcCtx* ctx, const OrcScene* s, const SurfHit* surf, v4 I)   /* :1670-1707 */
{
    if (surf->flags & PIM_MATFLAG_REFRACTIVE) return Scatter_Refractive(ctx, s, surf, I);
    Scatter scatter;
    float amtSpec = o_lerp(0.5f, 1.0f, surf->metallic);
    float amtDiffuse = 1.0f - amtSpec;
    if (Sample1D(ctx) < amtSpec)
    {
        scatter = Scatter_Specular(ctx, s, surf, I);
        if (amtDiffuse > ORC_EPS)
        {
            v4 eval = Eval_Diffuse(surf, I, scatter.dir);
            scatter.attenuation = o4_lerps(scatter.attenuation, eval, amtDiffuse);
            scatter.pdf = o_lerp(scatter.pdf, eval.w, amtDiffuse);
        }
    }
    else
    {
        scatter = Scatter_Diffuse(ctx, surf, I);
        v4 eval = Eval_Specular(s, surf, I, scatter.dir);
        scatter.attenuation = o4_lerps(scatter.attenuation, eval, amtSpec);
        scatter.pdf = o_lerp(scatter.pdf, eval.w, amtSpec);
    }
    return scatter;
}

/* ------------------------------------------------------------------------------------------ */
/* light sampling, pt.c:1709-1942                                                                */

static void LightOnHit(OrcScene* s, v4 ro, v4 lum4, int32_t iVert)          /* :1709-1733 */
{
    float lum = o4_avglum(lum4);
    if (lum <= ORC_EPS) return;
    float loglum = o_log2f(lum) - ORC_LOG2_EPS;
    loglum = o_clamp(loglum, 0.0f, 46.0f);
    uint32_t amt = (uint32_t)(loglum * (0xff / 46.0f) + 0.5f);
    int32_t iGrid = grid_index(&s->grid, ro);
    int32_t iEmit = s->vertToEmit[iVert];
    if (iEmit >= 0 && s->cellLength[iGrid])
    {
        __atomic_fetch_add(&s->live[(size_t)iGrid * s->emissiveCount + iEmit], amt, __ATOMIC_RELAXED);
    }
}

static float Dist_PdfD(const OrcScene* s, int32_t iCell, int32_t iEmit) { return s->pdf[(size_t)iCell * s->emissiveCount + iEmit] / s->emissiveCount; }   /* dist1d.c:118 */

static bool LightSelect(OrcCtx* ctx, const OrcScene* s, v4 position, int32_t* iVertOut, float* pdfOut)   /* :1735-1764 */
{
    if (s->emissiveCount == 0) { *iVertOut = -1; *pdfOut = 0.0f; return false; }
    int32_t iCell = grid_index(&s->grid, position);
    if (!s->cellLength[iCell]) return false;
    const int32_t E = s->emissiveCount;
    int32_t iEmit = dist_find_interval(s->cdf + (size_t)iCell * (E + 1), E + 1, Sample1D(ctx));
    float pdf = Dist_PdfD(s, iCell, iEmit);
    *iVertOut = s->emitToVert[iEmit];
    *pdfOut = pdf;
    return pdf > ORC_EPS;
}

static float LightSelectPdf(const OrcScene* s, int32_t iVert, v4 ro)        /* :1766-1783 */
{
    float selectPdf = 1.0f;
    int32_t iGrid = grid_index(&s->grid, ro);
    int32_t iEmit = s->vertToEmit[iVert];
    if (iEmit >= 0 && s->cellLength[iGrid]) selectPdf = Dist_PdfD(s, iGrid, iEmit);
    return selectPdf;
}

static Media Media_Sample(const OrcMedia* desc, v4 P)                       /* :2146-2181 */
{
    Media hit;
    hit.scattering = desc->constantMu;
    hit.g0 = desc->phaseDirA; hit.g1 = desc->phaseDirB; hit.gBlend = desc->phaseBlend;
    if (o_distance(P.y, desc->noiseHeight) <= desc->noiseRange)
    {
        float noise = FbmGradientNoise3(o4_muls(P, desc->noiseFreq), desc->noiseLacunarity, desc->noiseGain, desc->noiseOctaves, 1);
        float noiseScale = desc->noiseScale;
        float height = desc->noiseHeight + noiseScale * noise;
        float dist = o_distance(P.y, height) / noiseScale;
        float heightDensity = o_sat(1.0f - dist);
        hit.scattering = o4_add(hit.scattering, o4_muls(desc->noiseMu, heightDensity));
    }
    hit.extinction = o4_muls(hit.scattering, 1.0f + desc->absorption);
    return hit;
}

static float CalcPhase(Media m, float cosTheta) { return o_lerp(MiePhase(cosTheta, m.g0), MiePhase(cosTheta, m.g1), m.gBlend); }   /* :2198 */

static v4 SamplePhaseDir(OrcCtx* ctx, Media media, v4 rd)                   /* :2208-2221 */
{
    v4 L;
    do
    {
        L = SampleUnitSphere(Sample2D(ctx));
        L.w = CalcPhase(media, o4_dot3(rd, L));
    } while (Sample1D(ctx) > L.w);
    return L;
}

static v4 CalcTransmittance(OrcCtx* ctx, const OrcScene* s, v4 ro, v4 rd, float rayLen)   /* :2223-2249 */
{
    const OrcMedia* desc = &s->media;
    const float rcpMaj = desc->rcpMajorant;
    float t = 0.0f;
    v4 attenuation = o4_s(1.0f);
    while (true)
    {
        float dt = SampleFreePath(Sample1D(ctx), rcpMaj);
        if ((t + dt) >= rayLen) break;
        Media media = Media_Sample(desc, o4_add(ro, o4_muls(rd, t)));
        v4 ratio = o4_inv(o4_muls(media.extinction, rcpMaj));
        attenuation = o4_mul(attenuation, ratio);
        t += dt;
    }
    return attenuation;
}

static LightSample SampleLight(OrcCtx* ctx, const OrcScene* s, v4 ro, int32_t iVert)   /* :1785-1827 */
{
    LightSample sample; memset(&sample, 0, sizeof(sample));
    v4 wuv = SampleBaryCoord(Sample2D(ctx));
    v4 A = s->positions[iVert + 0], B = s->positions[iVert + 1], C = s->positions[iVert + 2];
    v4 pt = o4_blend(A, B, C, wuv);
    float area = TriArea3D(A, B, C);
    v4 rd = o4_sub(pt, ro);
    float distSq = o4_dot3(rd, rd);
    float distance = sqrtf(o_max(distSq, ORC_EPS_SQ));
    wuv.w = distance;
    rd = o4_divs(rd, distance);
    sample.direction = rd;
    sample.wuvt = wuv;
    ctx->shadowRays++;
    PimRayHit hit = pt_intersect(s, ro, rd, 0.0f, distance + 0.01f * ORC_MILLI);
    if ((hit.type != PIM_HIT_NOTHING) && (hit.iVert == iVert))
    {
        float cosTheta = o_abs(o4_dot3(rd, hit_normal(&hit)));
        sample.pdf = LightPdf(area, cosTheta, distSq);
        sample.luminance = GetEmission(s, rd, &hit);
        if (o4_sum3(sample.luminance) > ORC_EPS)
        {
            v4 Tr = CalcTransmittance(ctx, s, ro, rd, hit.wuvt[3]);
            sample.luminance = o4_mul(sample.luminance, Tr);
        }
    }
    return sample;
}

static float LightEvalPdf(OrcCtx* ctx, const OrcScene* s, v4 ro, v4 rd, PimRayHit* hitOut)   /* :1829-1847 */
{
    ctx->shadowRays++;
    PimRayHit hit = pt_intersect(s, ro, rd, 0.0f, ORC_RCP_EPS);
    float pdf = 0.0f;
    if (hit.type != PIM_HIT_NOTHING)
    {
        float cosTheta = o_abs(o4_dot3(rd, hit_normal(&hit)));
        float area = GetArea(s, hit.iVert);
        float distSq = o_max(ORC_EPS, hit.wuvt[3] * hit.wuvt[3]);
        pdf = LightPdf(area, cosTheta, distSq);
    }
    *hitOut = hit;
    return pdf;
}

static v4 EstimateDirect(OrcCtx* ctx, const OrcScene* s, const SurfHit* surf, const PimRayHit* srcHit, v4 I)   /* :1849-1919 */
{
    v4 result = o4_s(0.0f);
    if (surf->flags & PIM_MATFLAG_REFRACTIVE) return result;
    const v4 ro = surf->P;
    const float pRough = o_lerp(0.05f, 0.95f, surf->roughness);
    const float pSmooth = 1.0f - pRough;
    if (Sample1D(ctx) < pRough)
    {
        int32_t iVert; float selectPdf;
        if (LightSelect(ctx, s, surf->P, &iVert, &selectPdf))
        {
            if (srcHit->iVert != iVert)
            {
                LightSample sample = SampleLight(ctx, s, ro, iVert);
                v4 rd = sample.direction;
                v4 Li = sample.luminance;
                float lightPdf = sample.pdf * selectPdf * pRough;
                if ((lightPdf > ORC_EPS) && (o4_hmax3(Li) > ORC_EPS))
                {
                    v4 brdf = Eval_Principled(s, surf, I, rd);
                    Li = o4_mul(Li, brdf);
                    float brdfPdf = brdf.w * pSmooth;
                    if (brdfPdf > ORC_EPS)
                    {
                        Li = o4_muls(Li, PowerHeuristic(lightPdf, brdfPdf) / lightPdf);
                        result = o4_add(result, Li);
                    }
                }
            }
        }
    }
    else
    {
        Scatter sample = Scatter_Principled(ctx, s, surf, I);
        v4 rd = sample.dir;
        float brdfPdf = sample.pdf * pSmooth;
        if (brdfPdf > ORC_EPS)
        {
            PimRayHit hit;
            float lightPdf = LightEvalPdf(ctx, s, ro, rd, &hit) * pRough;
            if (lightPdf > ORC_EPS)
            {
                lightPdf *= LightSelectPdf(s, hit.iVert, ro);
                v4 Li = o4_mul(GetEmission(s, rd, &hit), sample.attenuation);
                if (o4_hmax3(Li) > ORC_EPS)
                {
                    Li = o4_muls(Li, PowerHeuristic(brdfPdf, lightPdf) / brdfPdf);
                    Li = o4_mul(Li, CalcTransmittance(ctx, s, ro, rd, hit.wuvt[3]));
                    result = o4_add(result, Li);
                }
            }
        }
    }
    return result;
}

static bool EvaluateLight(OrcCtx* ctx, const OrcScene* s, v4 P, v4* lightOut, v4* dirOut)   /* :1921-1942 */
{
    int32_t iVert; float selectPdf;
    if (LightSelect(ctx, s, P, &iVert, &selectPdf))
    {
        LightSample sample = SampleLight(ctx, s, P, iVert);
        ctx->shadowRays--; ctx->otherRays++;   /* bookkeeping: media NEE counts as 'other' */
        if (sample.pdf > ORC_EPS)
        {
            *lightOut = o4_divs(sample.luminance, sample.pdf * selectPdf);
            *dirOut = sample.direction;
            return true;
        }
    }
    return false;
}

static Scatter ScatterRay(OrcCtx* ctx, const OrcScene* s, v4 ro, v4 rd, float rayLen)   /* :2251-2304 */
{
    Scatter result; memset(&result, 0, sizeof(result));
    result.pdf = 0.0f;
    const OrcMedia* desc = &s->media;
    const float rcpMaj = desc->rcpMajorant;
    float t = 0.0f;
    result.attenuation = o4_s(1.0f);
    while (true)
    {
        float dt = SampleFreePath(Sample1D(ctx), rcpMaj);
        t += dt;
        if (t >= rayLen) break;
        v4 P = o4_add(ro, o4_muls(rd, t));
        Media media = Media_Sample(desc, P);
        v4 ratio = o4_inv(o4_muls(media.extinction, rcpMaj));
        result.attenuation = o4_mul(result.attenuation, ratio);
        float scatterProb = o4_hmax3(media.scattering) * rcpMaj;
        if (Sample1D(ctx) < scatterProb)
        {
            v4 lum, L;
            if (EvaluateLight(ctx, s, P, &lum, &L))
            {
                float ph = CalcPhase(media, o4_dot3(rd, L));
                lum = o4_muls(lum, ph * dt);
                lum = o4_mul(result.attenuation, lum);
                result.luminance = lum;
            }
            result.pos = P;
            result.dir = SamplePhaseDir(ctx, media, rd);
            result.pdf = result.dir.w;
            float ph = CalcPhase(media, o4_dot3(rd, result.dir));
            result.attenuation = o4_muls(result.attenuation, ph);
            break;
        }
    }
    return result;
}

/* Pt_TraceRay, pt.c:2306-2416 */
static PimResult Pt_TraceRay(OrcCtx* ctx, OrcScene* s, v4 ro, v4 rd, int32_t maxBounces)
{
    PimResult result; memset(&result, 0, sizeof(result));
    float resultWeight = 0.0f;
    v4 luminance = o4_s(0.0f);
    v4 attenuation = o4_s(1.0f);
    v3 albedoAcc = o3(0, 0, 0), normalAcc = o3(0, 0, 0);
    uint32_t prevFlags = 0;
    const int32_t bounceLimit = (maxBounces > 0 && maxBounces < 666) ? maxBounces : 666;

    for (int32_t b = 0; b < bounceLimit; ++b)
    {
        {
            float p = o_sat(o4_avglum(attenuation));
            if (Sample1D(ctx) < p) attenuation = o4_divs(attenuation, p);
            else break;
        }
        ctx->extRays++;
        PimRayHit hit = pt_intersect(s, ro, rd, 0.0f, ORC_RCP_EPS);
        if (hit.type == PIM_HIT_NOTHING)
        {
            luminance = o4_add(luminance, o4_mul(attenuation, GetSky(s, rd)));
            break;
        }
        if ((hit.type == PIM_HIT_BACKFACE) && !(hit.flags & PIM_MATFLAG_REFRACTIVE)) break;
        {
            Scatter scatter = ScatterRay(ctx, s, ro, rd, hit.wuvt[3]);
            if (scatter.pdf > ORC_EPS)
            {
                luminance = o4_add(luminance, o4_mul(attenuation, scatter.luminance));
                attenuation = o4_mul(attenuation, o4_divs(scatter.attenuation, scatter.pdf));
                {
                    v4 a = o4_muls(attenuation, 1.0f / ORC_TAU);
                    float w = o_sat(1.0f - o4_avglum(a));
                    resultWeight += w;
                    albedoAcc = o3(albedoAcc.x + a.x * w, albedoAcc.y + a.y * w, albedoAcc.z + a.z * w);
                    v4 nr = o4_neg(rd);
                    normalAcc = o3(normalAcc.x + nr.x * w, normalAcc.y + nr.y * w, normalAcc.z + nr.z * w);
                }
                ro = scatter.pos;
                rd = scatter.dir;
                prevFlags = 0;
                continue;
            }
            else
            {
                attenuation = o4_mul(attenuation, scatter.attenuation);
            }
        }
        SurfHit surf = GetSurface(s, rd, &hit);
        if (b > 0) LightOnHit(s, ro, surf.emission, hit.iVert);
        if ((b == 0) || (prevFlags & PIM_MATFLAG_REFRACTIVE))
        {
            luminance = o4_add(luminance, o4_mul(surf.emission, attenuation));
        }
        if (hit.flags & PIM_MATFLAG_SKY) break;
        {
            v4 Li = EstimateDirect(ctx, s, &surf, &hit, rd);
            luminance = o4_add(luminance, o4_mul(Li, attenuation));
        }
        Scatter scatter = Scatter_Principled(ctx, s, &surf, rd);
        if (!(scatter.pdf > ORC_EPS)) break;
        ro = scatter.pos;
        rd = scatter.dir;
        attenuation = o4_mul(attenuation, o4_divs(scatter.attenuation, scatter.pdf));
        prevFlags = surf.flags;
        {
            v4 a = o4_muls(attenuation, 1.0f / ORC_PI);
            float w = o_sat(1.0f - o4_avglum(a));
            resultWeight += w;
            albedoAcc = o3(albedoAcc.x + surf.albedo.x * w, albedoAcc.y + surf.albedo.y * w, albedoAcc.z + surf.albedo.z * w);
            normalAcc = o3(normalAcc.x + surf.N.x * w, normalAcc.y + surf.N.y * w, normalAcc.z + surf.N.z * w);
        }
    }
    {
        float sc = 1.0f / o_max(resultWeight, ORC_EPS);
        result.color[0] = luminance.x; result.color[1] = luminance.y; result.color[2] = luminance.z;
        result.albedo[0] = albedoAcc.x * sc; result.albedo[1] = albedoAcc.y * sc; result.albedo[2] = albedoAcc.z * sc;
        result.normal[0] = normalAcc.x * sc; result.normal[1] = normalAcc.y * sc; result.normal[2] = normalAcc.z * sc;
    }
    return result;
}

/* ------------------------------------------------------------------------------------------ */
/* public batched entry points                                                                  */

typedef struct
{
    const OrcScene* s; const float* ro; const float* rd; const float* tNear; const float* tFar;
    PimRayHit* out; int32_t* prim; float* tuv; double* t64; int brute;
} IsectTask;

static inline v4 ld4(const float* p, int64_t i) { return o4(p[i * 4 + 0], p[i * 4 + 1], p[i * 4 + 2], p[i * 4 + 3]); }

static void isect_fn(void* user, int64_t begin, int64_t end, int tid)
{
    (void)tid;
    IsectTask* t = (IsectTask*)user;
    for (int64_t i = begin; i < end; ++i)
    {
        v4 ro = ld4(t->ro, i), rd = ld4(t->rd, i);
        if (t->out) t->out[i] = pt_intersect(t->s, ro, rd, t->tNear[i], t->tFar[i]);
        else
        {
            OrcHit h = (t->brute || !t->s->accel) ? closest_brute(t->s, ro, rd, t->tNear[i], t->tFar[i]) : closest_bvh(t->s, ro, rd, t->tNear[i], t->tFar[i], false);
            t->prim[i] = h.prim;
            t->tuv[i * 3 + 0] = h.prim >= 0 ? h.t : 0.0f; t->tuv[i * 3 + 1] = h.u; t->tuv[i * 3 + 2] = h.v;
        }
    }
}

void orc_intersect(const OrcScene* s, const float* ro, const float* rd, const float* tNear, const float* tFar, int32_t n, PimRayHit* out, int threads)
{
    IsectTask t; memset(&t, 0, sizeof(t));
    t.s = s; t.ro = ro; t.rd = rd; t.tNear = tNear; t.tFar = tFar; t.out = out;
    orc_parallel_for(isect_fn, &t, n, threads);
}

void orc_intersect_raw(const OrcScene* s, const float* ro, const float* rd, const float* tNear, const float* tFar, int32_t n, int32_t* prim, float* tuv, int bruteForce, int threads)
{
    IsectTask t; memset(&t, 0, sizeof(t));
    t.s = s; t.ro = ro; t.rd = rd; t.tNear = tNear; t.tFar = tFar; t.prim = prim; t.tuv = tuv; t.brute = bruteForce;
    orc_parallel_for(isect_fn, &t, n, threads);
}

/* fp64 Moeller-Trumbore ground truth, two sided, (tNear, tFar], lowest primID on ties */
static void isect64_fn(void* user, int64_t begin, int64_t end, int tid)
{
    (void)tid;
    IsectTask* t = (IsectTask*)user;
    const OrcScene* s = t->s;
    const int32_t triCount = s->vertCount / 3;
    for (int64_t i = begin; i < end; ++i)
    {
        double o[3] = { t->ro[i * 4], t->ro[i * 4 + 1], t->ro[i * 4 + 2] };
        double d[3] = { t->rd[i * 4], t->rd[i * 4 + 1], t->rd[i * 4 + 2] };
        double best = (double)t->tFar[i]; int32_t bestPrim = -1;
        for (int32_t k = 0; k < triCount; ++k)
        {
            const v4* p = s->positions + k * 3;
            double e1[3] = { (double)p[1].x - p[0].x, (double)p[1].y - p[0].y, (double)p[1].z - p[0].z };
            double e2[3] = { (double)p[2].x - p[0].x, (double)p[2].y - p[0].y, (double)p[2].z - p[0].z };
            double pv[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
            double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
            if (det == 0.0) continue;
            double inv = 1.0 / det;
            double tv[3] = { o[0] - p[0].x, o[1] - p[0].y, o[2] - p[0].z };
            double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
            if (u < 0.0 || u > 1.0) continue;
            double qv[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
            double v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * inv;
            if (v < 0.0 || u + v > 1.0) continue;
            double tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
            if (!(tt > (double)t->tNear[i] && tt <= (double)t->tFar[i])) continue;
            if (tt < best || (tt == best && bestPrim < 0)) { best = tt; bestPrim = k; }
        }
        t->prim[i] = bestPrim;
        t->t64[i] = bestPrim >= 0 ? best : 0.0;
    }
}

void orc_intersect_f64(const OrcScene* s, const float* ro, const float* rd, const float* tNear, const float* tFar, int32_t n, int32_t* prim, double* tOut, int threads)
{
    IsectTask t; memset(&t, 0, sizeof(t));
    t.s = s; t.ro = ro; t.rd = rd; t.tNear = tNear; t.tFar = tFar; t.prim = prim; t.t64 = tOut;
    orc_parallel_for(isect64_fn, &t, n, threads);
}

/* The two ground truths on a bare triangle soup (no scene set-up: BASELINE config 5's 20 M triangles would spend minutes in
 * SetupEmissives / SetupLightGrid, and brute force needs neither): the contract's fp32 watertight test over ALL triangles
 * (prim, tuv; what the GPU must equal bit for bit) and, when t64 != NULL, the fp64 ground truth (prim64, t64).
 * positions: float4[vertCount], borrowed. */
void orc_brute_positions(const float* positions, int32_t vertCount, const float* ro, const float* rd, const float* tNear, const float* tFar,
    int32_t n, int32_t* prim, float* tuv, int32_t* prim64, double* t64, int threads)
{
    OrcScene s; memset(&s, 0, sizeof(s));
    s.positions = (v4*)positions; s.vertCount = vertCount; s.accel = 0;
    if (prim && tuv) orc_intersect_raw(&s, ro, rd, tNear, tFar, n, prim, tuv, 1, threads);
    if (prim64 && t64) orc_intersect_f64(&s, ro, rd, tNear, tFar, n, prim64, t64, threads);
}

typedef struct
{
    OrcScene* s; const float* ro; const float* rd; PimResult* out; uint32_t sampleIndex, seed; int32_t maxBounces;
    v4 origin; float* colors; float* directions;
} RaysTask;

static void rays_fn(void* user, int64_t begin, int64_t end, int tid)
{
    (void)tid;
    RaysTask* t = (RaysTask*)user;
    for (int64_t i = begin; i < end; ++i)
    {
        OrcCtx ctx; memset(&ctx, 0, sizeof(ctx));
        ctx.rng = pcg4((uint32_t)i, t->sampleIndex, t->seed, TAG_RAY);
        if (t->out)
        {
            t->out[i] = Pt_TraceRay(&ctx, t->s, ld4(t->ro, i), ld4(t->rd, i), t->maxBounces);
        }
        else
        {
            /* RayGenFn, pt.c:2596-2612 */
            v4 rd = SampleUnitSphere(Sample2D(&ctx));
            t->directions[i * 4 + 0] = rd.x; t->directions[i * 4 + 1] = rd.y; t->directions[i * 4 + 2] = rd.z; t->directions[i * 4 + 3] = rd.w;
            PimResult r = Pt_TraceRay(&ctx, t->s, t->origin, rd, t->maxBounces);
            t->colors[i * 4 + 0] = r.color[0]; t->colors[i * 4 + 1] = r.color[1]; t->colors[i * 4 + 2] = r.color[2]; t->colors[i * 4 + 3] = 0.0f;
        }
    }
}

void orc_trace_rays(OrcScene* s, const float* ro, const float* rd, int32_t n, uint32_t sampleIndex, uint32_t seed, int32_t maxBounces, PimResult* out, int threads)
{
    RaysTask t; memset(&t, 0, sizeof(t));
    t.s = s; t.ro = ro; t.rd = rd; t.out = out; t.sampleIndex = sampleIndex; t.seed = seed; t.maxBounces = maxBounces;
    orc_parallel_for(rays_fn, &t, n, threads);
}

void orc_raygen(OrcScene* s, const float* origin, int32_t count, uint32_t sampleIndex, uint32_t seed, int32_t maxBounces, float* colors, float* directions, int threads)
{
    RaysTask t; memset(&t, 0, sizeof(t));
    t.s = s; t.sampleIndex = sampleIndex; t.seed = seed; t.maxBounces = maxBounces;
    t.origin = o4(origin[0], origin[1], origin[2], origin[3]); t.colors = colors; t.directions = directions;
    orc_scene_update_dists(s);          /* Pt_RayGen: PtScene_Update first (pt.c:2625 -> :1034 UpdateDists); AmbCube_Bake comes through here */
    orc_parallel_for(rays_fn, &t, count, threads);
}

/* Cubemap_Bake + BakeFn, rendering/cubemap.c:130-190: one path per cube-face texel through a tent-jittered
 * direction, blended into color[face][texel] with f3_lerpvs(old, new, weight).  D1: one stream per texel. */
typedef struct { OrcScene* s; float* const* faces; int32_t size; v4 origin; float weight; uint32_t sampleIndex, seed; int32_t maxBounces; } CubeTask;
static void cube_fn(void* user, int64_t begin, int64_t end, int tid)
{
    (void)tid;
    CubeTask* t = (CubeTask*)user;
    const int32_t size = t->size, flen = size * size;
    for (int64_t i = begin; i < end; ++i)
    {
        OrcCtx ctx; memset(&ctx, 0, sizeof(ctx));
        ctx.rng = pcg4((uint32_t)i, t->sampleIndex, t->seed, TAG_CUBE);
        int32_t face = (int32_t)(i / flen), fi = (int32_t)(i % flen);
        v2 Xi = f2_tent(rng_f2(&ctx.rng));
        v4 dir = Cubemap_CalcDir(size, face, fi % size, fi / size, Xi);
        PimResult r = Pt_TraceRay(&ctx, t->s, t->origin, dir, t->maxBounces);
        float* c = t->faces[face] + (size_t)fi * 3;
        for (int k = 0; k < 3; ++k) c[k] = c[k] + (r.color[k] - c[k]) * t->weight;        /* f3_lerpvs, float3_funcs.h:373 */
    }
}

void orc_cubemap_bake(OrcScene* s, float* const faces[6], int32_t size, const float* origin, float weight,
    uint32_t sampleIndex, uint32_t seed, int32_t maxBounces, int threads)
{
    if (size <= 0) return;                                                                /* cubemap.c:172 */
    CubeTask t; memset(&t, 0, sizeof(t));
    t.s = s; t.faces = faces; t.size = size; t.origin = o4(origin[0], origin[1], origin[2], origin[3]); t.weight = weight;
    t.sampleIndex = sampleIndex; t.seed = seed; t.maxBounces = maxBounces;
    orc_parallel_for(cube_fn, &t, (int64_t)size * size * 6, threads);
}

/* Cubemap_Convolve + PrefilterFn + PrefilterEnvMap, rendering/cubemap.c:192-303: every texel of every mip of the convolved chain
 * gets the split-sum (N = V) GGX prefilter of the colour faces at the mip's roughness, blended in with `weight`.  D1: the reference
 * draws from the worker thread's generator (Prng_Get, cubemap.c:247), so its result depends on which thread ran which texel; here
 * texel i of mip m owns the stream Pcg4(m * 6 * size^2 + i, sampleIndex, seed, TAG_CONV).  The sums run in sample order. */
#define TAG_CONV     0x434F4E56u
typedef struct { const v3* color[6]; v4* convolved[6]; int32_t size, mip, msize; uint32_t sampleCount; float weight; uint32_t sampleIndex, seed; } ConvTask;
static v4 PrefilterEnvMap(u4* rng, const ConvTask* t, v4 N, float roughness)                     /* cubemap.c:192-223 */
{
    const v4 I = o4_neg(N);                                                               /* TBN.c2 = N */
    const float alpha = BrdfAlpha(roughness);
    float basisIntegral = 0.0f;
    v4 lightIntegral = o4_s(0.0f);
    for (uint32_t i = 0; i < t->sampleCount; ++i)
    {
        const v4 H = TanToWorld(N, SampleGGXMicrofacet(rng_f2(rng), alpha));              /* TbnToWorld(NormalToTBN(N), .) */
        const v4 L = o4_reflect3(I, H);
        const float NoL = o4_dot3(L, N);
        if (NoL > 0.0f)
        {
            const v3 c = Cubemap_ReadColor(t->color, t->size, L);
            v4 sample = o4(c.x, c.y, c.z, 1.0f);
            sample = o4_muls(sample, NoL);
            lightIntegral = o4_add(lightIntegral, sample);
            basisIntegral += NoL;
        }
    }
    if (basisIntegral > 0.0f) lightIntegral = o4_divs(lightIntegral, basisIntegral);
    return lightIntegral;
}
static void conv_fn(void* user, int64_t begin, int64_t end, int tid)                            /* PrefilterFn, cubemap.c:235-262 */
{
    (void)tid;
    const ConvTask* t = (const ConvTask*)user;
    const int32_t msize = t->msize, len = msize * msize;
    const float roughness = MipToRoughness((float)t->mip);
    for (int64_t i = begin; i < end; ++i)
    {
        u4 rng = pcg4((uint32_t)((int64_t)t->mip * 6 * t->size * t->size + i), t->sampleIndex, t->seed, TAG_CONV);
        const int32_t face = (int32_t)(i / len), fi = (int32_t)(i % len);
        const v2 Xi = f2_tent(rng_f2(&rng));
        const v4 N = Cubemap_CalcDir(msize, face, fi % msize, fi / msize, Xi);
        const v4 light = PrefilterEnvMap(&rng, t, N, roughness);
        Cubemap_BlendMip((v4* const*)t->convolved, t->size, face, fi % msize, fi / msize, t->mip, light, t->weight);
    }
}
void orc_cubemap_convolve(const float* const faces[6], float* const convolved[6], int32_t size, uint32_t sampleCount, float weight,
    uint32_t sampleIndex, uint32_t seed, int threads)
{
    if (size <= 0) return;
    ConvTask t; memset(&t, 0, sizeof(t));
    for (int f = 0; f < 6; ++f) { t.color[f] = (const v3*)faces[f]; t.convolved[f] = (v4*)convolved[f]; }
    t.size = size; t.sampleCount = sampleCount; t.weight = weight; t.sampleIndex = sampleIndex; t.seed = seed;
    const int32_t mipCount = CalcMipCount(size);
    for (int32_t m = 0; m < mipCount; ++m)                                                /* cubemap.c:279-293: one task per mip */
    {
        const int32_t msize = size >> m;
        const int64_t len = (int64_t)msize * msize * 6;
        if (len <= 0) continue;
        t.mip = m; t.msize = msize;
        orc_parallel_for(conv_fn, &t, len, threads);
    }
}

int32_t orc_cubemap_elem_count(int32_t size) { return size > 0 ? CubemapElemCount(size) : 0; }

/* AmbCube_Bake, math/ambcube.c:5-32: Pt_RayGen + the serial fit over the samples in order */
int32_t orc_ambcube_bake(OrcScene* s, float* cube, const float* origin, int32_t samples, int32_t prevSampleCount,
    uint32_t sampleIndex, uint32_t seed, int32_t maxBounces, int threads)
{
    float* colors = (float*)malloc(sizeof(float) * 4 * (size_t)(samples + 1));
    float* directions = (float*)malloc(sizeof(float) * 4 * (size_t)(samples + 1));
    orc_raygen(s, origin, samples, sampleIndex, seed, maxBounces, colors, directions, threads);
    OrcAmbCube c; memcpy(&c, cube, sizeof(c));
    float w = 6.0f / (1.0f + samples);
    w = w / (1.0f + prevSampleCount);
    for (int32_t i = 0; i < samples; ++i) AmbCube_Fit(&c, w, ld4(directions, i), ld4(colors, i));
    memcpy(cube, &c, sizeof(c));
    free(colors); free(directions);
    return prevSampleCount + 1;
}

/* CalculateDof, pt.c:2418-2452 */
static void CalculateDof(OrcCtx* ctx, const PimDofInfo* dof, v4 right, v4 up, v4 fwd, v4* ro, v4* rd)
{
    v2 offset;
    {
        uint32_t side = rng_u32(&ctx->rng);
        v2 Xi = Sample2D(ctx);
        if (dof->bladeCount == 666) offset = SamplePentagram(Xi, side);
        else offset = SampleNGon(Xi, side, (uint32_t)dof->bladeCount, dof->bladeRot);
    }
    offset = o2_muls(offset, dof->aperture);
    float t = o_lerp(dof->focalLength / o4_dot3(*rd, fwd), dof->focalLength, dof->focalPlaneCurvature);
    v4 focusPos = o4_add(*ro, o4_muls(*rd, t));
    v4 aperturePos = o4_add(*ro, o4_add(o4_muls(right, offset.x), o4_muls(up, offset.y)));
    *ro = aperturePos;
    *rd = o4_normalize3(o4_sub(focusPos, aperturePos));
}

typedef struct
{
    OrcScene* s; float* color; float* albedo; float* normal; int32_t w, h;
    v4 eye, right, up, fwd; v2 slope; const PimDofInfo* dof; const PimTraceParams* p;
    uint64_t ext[256], shadow[256], other[256];
} ImageTask;

/* TraceFn, pt.c:2514-2554 */
static void image_fn(void* user, int64_t begin, int64_t end, int tid)
{
    ImageTask* t = (ImageTask*)user;
    const float rcpX = 1.0f / (float)t->w, rcpY = 1.0f / (float)t->h;
    const float sampleWeight = t->p->sampleWeight;
    for (int64_t i = begin; i < end; ++i)
    {
        const int32_t cx = (int32_t)(i % t->w), cy = (int32_t)(i / t->w);
        OrcCtx ctx; memset(&ctx, 0, sizeof(ctx));
        ctx.rng = pcg4((uint32_t)cx, (uint32_t)cy, t->p->sampleIndex, t->p->seed ^ TAG_PIXEL);
        const v2 baseUv = o2((cx + 0.5f) * rcpX, (cy + 0.5f) * rcpY);
        const v2 aa = SampleGaussPixelFilter(Sample2D(&ctx), 1.0f);
        const v2 rayUv = o2_add(baseUv, o2_mul(aa, o2(rcpX, rcpY)));
        v4 ro = t->eye;
        v4 rd = proj_dir(t->right, t->up, t->fwd, t->slope, o2(rayUv.x * 2.0f - 1.0f, rayUv.y * 2.0f - 1.0f));
        CalculateDof(&ctx, t->dof, t->right, t->up, t->fwd, &ro, &rd);
        PimResult r = Pt_TraceRay(&ctx, t->s, ro, rd, t->p->maxBounces);
        for (int k = 0; k < 3; ++k)
        {
            t->color[i * 3 + k] = o_lerp(t->color[i * 3 + k], r.color[k], sampleWeight);
            t->albedo[i * 3 + k] = o_lerp(t->albedo[i * 3 + k], r.albedo[k], sampleWeight);
            t->normal[i * 3 + k] = o_lerp(t->normal[i * 3 + k], r.normal[k], sampleWeight);
        }
        t->ext[tid] += ctx.extRays; t->shadow[tid] += ctx.shadowRays; t->other[tid] += ctx.otherRays;
    }
}

/* Pt_Trace, pt.c:2557-2585 (+ DofUpdate :2485-2503) */
void orc_trace_image(OrcScene* s, float* color, float* albedo, float* normal, int32_t width, int32_t height,
    const PimCamera* cam, PimDofInfo* dof, const PimTraceParams* p, int threads, PimTraceStats* stats)
{
    if (p->updateDists) orc_scene_update_dists(s);
    v4 rot = o4(cam->rotation[0], cam->rotation[1], cam->rotation[2], cam->rotation[3]);
    v4 eye = o4(cam->position[0], cam->position[1], cam->position[2], cam->position[3]);
    uint64_t other = 0;
    if (dof->autoFocus)
    {
        unsigned int csr = _mm_getcsr();
        _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON); _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
        PimRayHit hit = pt_intersect(s, eye, quat_fwd(rot), 0.0f, ORC_RCP_EPS);
        other = 1;
        if (hit.type != PIM_HIT_NOTHING)
        {
            float t = 1.0f - o_expf(-p->frameDt * dof->autoFocusSpeed);
            dof->focalLength = o_lerp(dof->focalLength, hit.wuvt[3], t);
        }
        _mm_setcsr(csr);
    }
    ImageTask* t = (ImageTask*)calloc(1, sizeof(ImageTask));
    t->s = s; t->color = color; t->albedo = albedo; t->normal = normal; t->w = width; t->h = height;
    t->eye = eye; t->right = quat_right(rot); t->up = quat_up(rot); t->fwd = quat_fwd(rot);
    t->slope = proj_slope(o_radians(cam->fovy), (float)width / (float)height);
    t->dof = dof; t->p = p;
    if (threads > 256) threads = 256;
    orc_parallel_for(image_fn, t, (int64_t)width * height, threads);
    if (stats)
    {
        memset(stats, 0, sizeof(*stats));
        for (int i = 0; i < 256; ++i) { stats->extensionRays += t->ext[i]; stats->shadowRays += t->shadow[i]; stats->otherRays += t->other[i]; }
        stats->otherRays += other;
        stats->paths = (uint64_t)width * height;
    }
    free(t);
}

/* CalcStdDev, render_system.c:1374-1394 */
void orc_stddev(const float* color, int32_t width, int32_t height, float* meanOut, float* stddevOut)
{
    const int32_t len = width * height;
    const float meanWeight = 1.0f / len;
    const float varianceWeight = 1.0f / (len - 1);
    float mean = 0.0f;
    for (int32_t i = 0; i < len; ++i)
    {
        float lum = (color[i * 3] + color[i * 3 + 1] + color[i * 3 + 2]) * (1.0f / 3.0f);
        mean += lum * meanWeight;
    }
    float variance = 0.0f;
    for (int32_t i = 0; i < len; ++i)
    {
        float lum = (color[i * 3] + color[i * 3 + 1] + color[i * 3 + 2]) * (1.0f / 3.0f);
        float err = lum - mean;
        variance += varianceWeight * (err * err);
    }
    if (meanOut) *meanOut = mean;
    if (stddevOut) *stddevOut = sqrtf(variance);
}

/* ------------------------------------------------------------------------------------------ */
/* lightmap bake, lm.c:1117-1223, lightmap.h:14-21                                               */

static const float kGiAxii[5][4] =
{
    { 0.000000f, 0.000000f, 1.000000f, 4.999773f },
    { 0.577350f, 0.577350f, 0.577350f, 4.999773f },
    { -0.577350f, 0.577350f, 0.577350f, 4.999773f },
    { 0.577350f, -0.577350f, 0.577350f, 4.999773f },
    { -0.577350f, -0.577350f, 0.577350f, 4.999773f },
};

typedef struct
{
    OrcScene* s; PimLightmapPage* pages; int32_t lmSize; float metersPerTexel, timeSlice; int32_t spp;
    uint32_t passIndex, seed; int32_t maxBounces; int64_t texelBegin;
} BakeTask;

static void bake_fn(void* user, int64_t begin, int64_t end, int tid)        /* BakeFn lm.c:1125-1201 */
{
    (void)tid;
    BakeTask* t = (BakeTask*)user;
    const int64_t lmLen = (int64_t)t->lmSize * t->lmSize;
    for (int64_t w = begin; w < end; ++w)
    {
        const int64_t iWork = t->texelBegin + w;
        const int32_t iLightmap = (int32_t)(iWork / lmLen);
        const int32_t iTexel = (int32_t)(iWork % lmLen);
        float* block = t->pages[iLightmap].block;
        float* probesBase = block;                                 /* [5][lmLen] float4, lm.c:97-101 */
        const float* position = block + 5 * 4 * lmLen;             /* float3 */
        const float* normal = position + 3 * lmLen;
        float* sampleCounts = (float*)(normal + 3 * lmLen);

        float sampleCount = sampleCounts[iTexel];
        if (sampleCount == 0.0f) continue;
        u4 rng = pcg4((uint32_t)iTexel, (uint32_t)iLightmap, t->passIndex, t->seed ^ TAG_BAKE);
        OrcCtx ctx; memset(&ctx, 0, sizeof(ctx));
        ctx.rng = pcg4((uint32_t)iTexel, (uint32_t)iLightmap, t->passIndex, t->seed ^ TAG_BAKE_PT);
        if (rng_f32(&rng) > t->timeSlice) continue;

        const v4 N = o4_normalize3(o4(normal[iTexel * 3], normal[iTexel * 3 + 1], normal[iTexel * 3 + 2], 0.0f));
        const v4 P = o4_add(o4(position[iTexel * 3], position[iTexel * 3 + 1], position[iTexel * 3 + 2], 1.0f), o4_muls(N, ORC_MILLI));
        const m33 TBN = NormalToTBN(N);
        v4 probes[5], axii[5];
        for (int i = 0; i < 5; ++i)
        {
            const float* pr = probesBase + ((int64_t)i * lmLen + iTexel) * 4;
            probes[i] = o4(pr[0], pr[1], pr[2], pr[3]);
            v4 ax = o4(kGiAxii[i][0], kGiAxii[i][1], kGiAxii[i][2], kGiAxii[i][3]);
            float sharpness = ax.w;
            ax = TbnToWorld(TBN, ax);
            ax.w = sharpness;
            axii[i] = ax;
        }
        for (int32_t i = 0; i < t->spp; ++i)
        {
            v4 Lts = SampleUnitHemisphere(rng_f2(&rng));
            v4 rd = TbnToWorld(TBN, Lts);
            float dt = (rng_f32(&rng) - 0.5f) * t->metersPerTexel;
            float db = (rng_f32(&rng) - 0.5f) * t->metersPerTexel;
            v4 ro = P;
            ro = o4_add(ro, o4_muls(TBN.c0, dt));
            ro = o4_add(ro, o4_muls(TBN.c1, db));
            PimResult result = Pt_TraceRay(&ctx, t->s, ro, rd, t->maxBounces);
            float weight = 1.0f / sampleCount;
            sampleCount += 1.0f;
            SG_Accumulate(weight, rd, o4(result.color[0], result.color[1], result.color[2], 0.0f), axii, probes, 5);
        }
        for (int i = 0; i < 5; ++i)
        {
            float* pr = probesBase + ((int64_t)i * lmLen + iTexel) * 4;
            pr[0] = probes[i].x; pr[1] = probes[i].y; pr[2] = probes[i].z; pr[3] = probes[i].w;
        }
        sampleCounts[iTexel] = sampleCount;
    }
}

/* LmPack_Bake, lm.c:1204-1223 */
void orc_lightmap_bake(OrcScene* s, PimLightmapPage* pages, int32_t lmCount, int32_t lmSize, float texelsPerMeter,
    float timeSlice, int32_t spp, uint32_t passIndex, uint32_t seed, int32_t maxBounces,
    int64_t texelBegin, int64_t texelEnd, int threads)
{
    orc_scene_update_dists(s);                  /* PtScene_Update, lm.c:1209 */
    const int64_t total = (int64_t)lmCount * lmSize * lmSize;
    if (texelBegin < 0) texelBegin = 0;
    if (texelEnd < 0 || texelEnd > total) texelEnd = total;
    BakeTask t; t.s = s; t.pages = pages; t.lmSize = lmSize; t.metersPerTexel = 1.0f / texelsPerMeter; t.timeSlice = timeSlice;
    t.spp = spp > 1 ? spp : 1; t.passIndex = passIndex; t.seed = seed; t.maxBounces = maxBounces; t.texelBegin = texelBegin;
    orc_parallel_for(bake_fn, &t, texelEnd - texelBegin, threads);
}

/* FlattenDrawables for one entity, pt.c:741-771 */
void orc_flatten_entity(const float* translation, const float* rotation, const float* scale,
    const float* meshPositions, const float* meshNormals, const float* meshUvs, int32_t meshLen,
    float* outPositions, float* outNormals, float* outUvs)
{
    m44 M = m44_trs(o4(translation[0], translation[1], translation[2], 0.0f),
        o4(rotation[0], rotation[1], rotation[2], rotation[3]), o4(scale[0], scale[1], scale[2], 0.0f));
    m33 IM = m44_IM(M);
    for (int32_t j = 0; j < meshLen; ++j)
    {
        v4 p = m44_mul_pt(M, ld4(meshPositions, j));
        v4 n = o4_normalize3(m33_mul_col(IM, ld4(meshNormals, j)));
        outPositions[j * 4 + 0] = p.x; outPositions[j * 4 + 1] = p.y; outPositions[j * 4 + 2] = p.z; outPositions[j * 4 + 3] = p.w;
        outNormals[j * 4 + 0] = n.x; outNormals[j * 4 + 1] = n.y; outNormals[j * 4 + 2] = n.z; outNormals[j * 4 + 3] = n.w;
        outUvs[j * 2 + 0] = meshUvs[j * 4 + 0]; outUvs[j * 2 + 1] = meshUvs[j * 4 + 1];
    }
}

/* FlattenDrawables, pt.c:692-782 */
void orc_flatten(const PimDrawables* d, float* positions, float* normals, float* uvs, int32_t* matIds, PimMaterial* materials)
{
    int32_t vertCount = 0, matCount = 0;
    for (int32_t i = 0; i < d->count; ++i)
    {
        if (d->meshes[i] < 0) continue;                                 /* Mesh_Get() == NULL, :726-729 */
        const PimMesh* mesh = &d->meshTable[d->meshes[i]];
        const int32_t meshLen = mesh->length;
        const int32_t vertBack = vertCount, matBack = matCount;
        vertCount += meshLen; matCount += 1;
        const float* c = d->matrices + (size_t)i * 16;
        m44 M;
        M.c0 = o4(c[0], c[1], c[2], c[3]); M.c1 = o4(c[4], c[5], c[6], c[7]);
        M.c2 = o4(c[8], c[9], c[10], c[11]); M.c3 = o4(c[12], c[13], c[14], c[15]);
        const m33 IM = m44_IM(M);
        materials[matBack] = d->materials[i];
        for (int32_t j = 0; j < meshLen; ++j)
        {
            const v4 p = m44_mul_pt(M, ld4(mesh->positions, j));
            const v4 n = o4_normalize3(m33_mul_col(IM, ld4(mesh->normals, j)));
            float* P = positions + (size_t)(vertBack + j) * 4; float* N = normals + (size_t)(vertBack + j) * 4;
            P[0] = p.x; P[1] = p.y; P[2] = p.z; P[3] = p.w;
            N[0] = n.x; N[1] = n.y; N[2] = n.z; N[3] = n.w;
            uvs[(size_t)(vertBack + j) * 2 + 0] = 0.0f; uvs[(size_t)(vertBack + j) * 2 + 1] = 0.0f;    /* Perm_Calloc */
            matIds[vertBack + j] = matBack;
        }
        for (int32_t j = 0; (j + 3) <= meshLen; j += 3)               /* whole triangles only, :755-766 */
        {
            for (int32_t k = 0; k < 3; ++k)
            {
                uvs[(size_t)(vertBack + j + k) * 2 + 0] = mesh->uvs[(size_t)(j + k) * 4 + 0];
                uvs[(size_t)(vertBack + j + k) * 2 + 1] = mesh->uvs[(size_t)(j + k) * 4 + 1];
            }
        }
    }
}

/* BakeBrdfFn, math/lighting.c:31-84; the texel's stream is build-defined, tag shared with prepare.hip */
#define TAG_LUT 0x4C555421u
void orc_brdf_lut_bake(float* texels, int32_t sx, int32_t sy, int32_t prevSamples, int32_t numSamples, uint32_t seed)
{
    if (numSamples <= prevSamples) return;                              /* BrdfLut_Update, :110 */
    const int32_t deltaSamples = numSamples - prevSamples;
    const float dx = 1.0f / (float)deltaSamples;
    const float dt = (float)deltaSamples / (float)numSamples;
    const float rcpWidth = 1.0f / (float)sx, rcpHeight = 1.0f / (float)sy;
    for (int32_t iTexel = 0; iTexel < sx * sy; ++iTexel)
    {
        const int32_t cx = iTexel % sx, cy = iTexel / sx;
        u4 rng = pcg4((uint32_t)iTexel, (uint32_t)prevSamples, seed, TAG_LUT);
        v2 result = o2(0.0f, 0.0f);
        for (int32_t iSample = prevSamples; iSample < numSamples; ++iSample)
        {
            const v2 XiAA = rng_f2(&rng);
            const float XiPhi = rng_f32(&rng);
            const v2 XiH = rng_f2(&rng);
            const v2 c = BrdfLutSample(cx, cy, rcpWidth, rcpHeight, dx, XiAA, XiPhi, XiH);
            result.x += c.x;
            result.y += c.y;
        }
        v2 old = o2(texels[iTexel * 2 + 0], texels[iTexel * 2 + 1]);
        v2 r = o2_lerps(old, result, dt);
        texels[iTexel * 2 + 0] = r.x; texels[iTexel * 2 + 1] = r.y;
    }
}

/* shims for oracle/orc_hooks.c (golden-vector pinning of the Dist1D restatement) */
float orc_test_dist_bake(float* pdf, float* cdf, int32_t len) { return dist_bake(pdf, cdf, len); }
int32_t orc_test_dist_find(const float* cdf, int32_t size, float u) { return dist_find_interval(cdf, size, u); }
void orc_test_dist_update(float* pdf, float* cdf, uint32_t* live, int32_t len, uint32_t* sum, float* integral) { dist_update(pdf, cdf, live, len, sum, integral); }

/* ------------------------------------------------------------------------------------------ */
/* EmbedAttributes, rendering/lightmap.c:941-1045                                              */

typedef struct { const PimDrawables* d; const int32_t* pageTexIds; PimLightmapPage* pages; int32_t lmSize; const int64_t* items; } EmbedTask;

static void embed_fn(void* user, int64_t begin, int64_t end, int tid)
{
    (void)tid;
    const EmbedTask* t = (const EmbedTask*)user;
    const PimDrawables* d = t->d;
    const int32_t lmSize = t->lmSize;
    const int64_t lmLen = (int64_t)lmSize * lmSize;
    const float texelSize = (float)lmSize;
    for (int64_t iItem = begin; iItem < end; ++iItem)                       /* :960-1031 */
    {
        const int64_t iWork = t->items ? t->items[iItem] : iItem;
        const int32_t iLightmap = (int32_t)(iWork / lmLen);
        const int32_t iTexel = (int32_t)(iWork % lmLen);
        const int32_t x = iTexel % lmSize, y = iTexel / lmSize;
        const v2 pxCenter = o2(x + 0.5f, y + 0.5f);
        const int32_t lmTxId = t->pageTexIds ? t->pageTexIds[iLightmap] : iLightmap;
        v4 lmPos = o4_s(0.0f), lmNor = o4_s(0.0f);
        float lmDist = (float)(1 << 23);
        for (int32_t iDraw = 0; iDraw < d->count; ++iDraw)
        {
            if (d->materials[iDraw].flags & (PIM_MATFLAG_SKY | PIM_MATFLAG_LAVA)) continue;      /* kUnmappedMaterials, :38 */
            if (d->meshes[iDraw] < 0) continue;
            const PimMesh* mesh = &d->meshTable[d->meshes[iDraw]];
            if (!mesh->texIndices) continue;
            for (int32_t iVert = 0; (iVert + 3) <= mesh->length; iVert += 3)
            {
                if (mesh->texIndices[(size_t)iVert * 4 + 3] != lmTxId) continue;
                const float* ua = mesh->uvs + (size_t)iVert * 4;
                const v2 A = o2_muls(o2(ua[2], ua[3]), texelSize), B = o2_muls(o2(ua[6], ua[7]), texelSize), C = o2_muls(o2(ua[10], ua[11]), texelSize);
                /* Not in the reference: a texel more than 3 texels outside the triangle's bounds cannot pass the test below,
                 * and skipping it changes no result, only the oracle's running time -- PROVIDED the triangle is not a
                 * sliver.  sdTriangle2D calls a point inside when none of its three edge cross products comes out negative;
                 * each is exact to a relative 5e-7 of |p - v| |edge|, a point that far outside is on the outer side of an
                 * edge line by an angle of at least half the sharpest corner, so the sign holds while every corner is wider
                 * than ~1e-6 rad.  Below 1e-5 rad (zero area included: then EVERY texel gets a negative distance, the
                 * reference's behaviour), or with coordinates that are not finite or huge, nothing is skipped. */
                {
                    const v2 ba = o2_sub(B, A), cb = o2_sub(C, B), ac = o2_sub(A, C);
                    const float area2 = o_abs(ba.x * ac.y - ba.y * ac.x);
                    const float lmax2 = o_max(o2_dot(ba, ba), o_max(o2_dot(cb, cb), o2_dot(ac, ac)));
                    const float lox = o_min(A.x, o_min(B.x, C.x)), hix = o_max(A.x, o_max(B.x, C.x));
                    const float loy = o_min(A.y, o_min(B.y, C.y)), hiy = o_max(A.y, o_max(B.y, C.y));
                    const int bounded = (area2 > 1.0e-5f * lmax2) && lox > -1.0e5f && loy > -1.0e5f && hix < 1.0e5f && hiy < 1.0e5f;
                    if (bounded && (pxCenter.x < lox - 3.0f || pxCenter.x > hix + 3.0f || pxCenter.y < loy - 3.0f || pxCenter.y > hiy + 3.0f)) continue;
                }
                const float dist = sdTriangle2D(A, B, C, pxCenter);
                if (dist < ORC_LM_FILL_PADDING && dist < lmDist)
                {
                    v4 pos[3] = { ld4(mesh->positions, iVert), ld4(mesh->positions, iVert + 1), ld4(mesh->positions, iVert + 2) };
                    v4 nor[3] = { ld4(mesh->normals, iVert), ld4(mesh->normals, iVert + 1), ld4(mesh->normals, iVert + 2) };
                    LmEmbedFragment(A, B, C, pxCenter, pos, nor, &lmPos, &lmNor);
                    lmDist = dist;
                }
            }
        }
        if (lmDist < ORC_LM_FILL_PADDING) lmNor = o4_normalize3(lmNor);
        float* block = t->pages[iLightmap].block;
        float* position = block + (size_t)lmLen * 20; float* normal = block + (size_t)lmLen * 23; float* sampleCounts = block + (size_t)lmLen * 26;
        sampleCounts[iTexel] = lmDist < ORC_LM_FILL_PADDING ? 1.0f : 0.0f;
        position[(size_t)iTexel * 3 + 0] = lmPos.x; position[(size_t)iTexel * 3 + 1] = lmPos.y; position[(size_t)iTexel * 3 + 2] = lmPos.z;
        normal[(size_t)iTexel * 3 + 0] = lmNor.x; normal[(size_t)iTexel * 3 + 1] = lmNor.y; normal[(size_t)iTexel * 3 + 2] = lmNor.z;
    }
}

void orc_lightmap_embed(const PimDrawables* drawables, const int32_t* pageTexIds, PimLightmapPage* pages,
    int32_t lmCount, int32_t lmSize, int threads)
{
    EmbedTask t = { drawables, pageTexIds, pages, lmSize, NULL };
    orc_parallel_for(embed_fn, &t, (int64_t)lmCount * lmSize * lmSize, threads);
}

/* the same loop body for a LIST of work items (iLightmap * lmSize^2 + iTexel) only: spot checks at sizes where the
 * whole O(texels x triangles) loop would take hours */
void orc_lightmap_embed_items(const PimDrawables* drawables, const int32_t* pageTexIds, PimLightmapPage* pages,
    int32_t lmSize, const int64_t* items, int64_t itemCount, int threads)
{
    EmbedTask t = { drawables, pageTexIds, pages, lmSize, items };
    orc_parallel_for(embed_fn, &t, itemCount, threads);
}

/* ------------------------------------------------------------------------------------------ */
/* LmPack_Pack's data-parallel steps, rendering/lightmap.c                                     */

/* chartnodes_create, :723-767 (nodes == NULL: counts only) */
int32_t orc_lightmap_chartnodes(const PimDrawables* d, float texelsPerUnit, PimChartNode* nodes)
{
    int32_t nodeCount = 0;
    for (int32_t di = 0; di < d->count; ++di)
    {
        if (d->materials[di].flags & (PIM_MATFLAG_SKY | PIM_MATFLAG_LAVA)) continue;
        if (d->meshes[di] < 0) continue;
        const PimMesh* mesh = &d->meshTable[d->meshes[di]];
        const float* c = d->matrices + (size_t)di * 16;
        m44 M;
        M.c0 = o4(c[0], c[1], c[2], c[3]); M.c1 = o4(c[4], c[5], c[6], c[7]);
        M.c2 = o4(c[8], c[9], c[10], c[11]); M.c3 = o4(c[12], c[13], c[14], c[15]);
        int32_t nodeBack = nodeCount;
        nodeCount += mesh->length / 3;
        if (!nodes) continue;
        for (int32_t v = 0; (v + 3) <= mesh->length; v += 3)
        {
            const v4 A = m44_mul_pt(M, ld4(mesh->positions, v)), B = m44_mul_pt(M, ld4(mesh->positions, v + 1)), C = m44_mul_pt(M, ld4(mesh->positions, v + 2));
            const OrcChartGeom g = ChartNodeGeom(A, B, C, texelsPerUnit);
            PimChartNode* n = &nodes[nodeBack++];
            n->plane[0] = g.plane.x; n->plane[1] = g.plane.y; n->plane[2] = g.plane.z; n->plane[3] = g.plane.w;
            n->triCoord[0] = g.a.x; n->triCoord[1] = g.a.y; n->triCoord[2] = g.b.x; n->triCoord[3] = g.b.y; n->triCoord[4] = g.c.x; n->triCoord[5] = g.c.y;
            n->area = g.area; n->drawableIndex = di; n->vertIndex = v; n->_pad[0] = n->_pad[1] = n->_pad[2] = 0;
        }
    }
    return nodeCount;
}

static void chart_minmax(const PimChartNode* nodes, int32_t first, int32_t last, v2* loOut, v2* hiOut)   /* :359-376 */
{
    const float bigNum = (float)(1 << 23);
    v2 lo = o2(bigNum, bigNum), hi = o2(-bigNum, -bigNum);
    for (int32_t i = first; i < last; ++i)
        for (int k = 0; k < 3; ++k)
        {
            const v2 p = o2(nodes[i].triCoord[2 * k], nodes[i].triCoord[2 * k + 1]);
            lo = o2_min(lo, p); hi = o2(o_max(hi.x, p.x), o_max(hi.y, p.y));
        }
    *loOut = lo; *hiOut = hi;
}

/* ChartMaskFn up to mask_new, :532-552 */
void orc_lightmap_chart_layout(PimChartNode* nodes, const int32_t* chartFirst, int32_t chartCount, float* areas, int32_t* maskSizes)
{
    for (int32_t c = 0; c < chartCount; ++c)
    {
        const int32_t first = chartFirst[c], last = chartFirst[c + 1];
        v2 lo, hi;
        chart_minmax(nodes, first, last, &lo, &hi);
        lo = o2(lo.x - 2.0f, lo.y - 2.0f);
        for (int32_t i = first; i < last; ++i)
            for (int k = 0; k < 3; ++k) { nodes[i].triCoord[2 * k] -= lo.x; nodes[i].triCoord[2 * k + 1] -= lo.y; }
        chart_minmax(nodes, first, last, &lo, &hi);
        const v2 size = o2_sub(hi, lo);
        areas[c] = size.x * size.y;
        hi = o2(hi.x + 2.0f, hi.y + 2.0f);
        maskSizes[2 * c] = (int32_t)hi.x; maskSizes[2 * c + 1] = (int32_t)hi.y;
    }
}

/* mask_tri over every node of every chart, :244-259 and :553-557 */
void orc_lightmap_chart_masks(const PimChartNode* nodes, const int32_t* chartFirst, int32_t chartCount, const int32_t* maskSizes,
    const int64_t* maskOffsets, uint8_t* masks)
{
    for (int32_t c = 0; c < chartCount; ++c)
    {
        const int32_t sx = maskSizes[2 * c], sy = maskSizes[2 * c + 1];
        uint8_t* mask = masks + maskOffsets[c];
        memset(mask, 0, (size_t)sx * (size_t)sy);                          /* mask_new: Perm_Calloc */
        for (int32_t i = chartFirst[c]; i < chartFirst[c + 1]; ++i)
        {
            const float* t = nodes[i].triCoord;
            const v2 a = o2(t[0], t[1]), b = o2(t[2], t[3]), cc = o2(t[4], t[5]);
            for (int32_t y = 0; y < sy; ++y)
                for (int32_t x = 0; x < sx; ++x)
                    if (sdTriangle2D(a, b, cc, o2(x + 0.5f, y + 0.5f)) <= ORC_LM_MASK_PADDING) mask[x + (size_t)y * sx] = 0xff;
        }
    }
}
