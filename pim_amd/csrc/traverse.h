/*
 * traverse.h -- per-lane BVH4 traversal with a watertight ray/triangle test; replaces
 * rtcIntersect1 / rtcOccluded (rendering/path_tracer.c:470-553).
 *
 * Contract (the same one the CPU checker states):
 *   - closest hit over t in (tNear, tFar]; two sided; among equal t the LOWEST primID wins, so
 *     the answer does not depend on traversal order or on the shape of the tree;
 *   - the triangle test is Woop/Benthin/Wald 2013 "watertight" in fp32 with the fp64 edge-function
 *     fallback, written without FMA contraction so the host evaluates the same bits;
 *   - u = weight of p1, v = weight of p2 (Embree's convention, RTC_GEOMETRY_TYPE_TRIANGLE).
 *
 * Boxes and the fp32 triangle test: the test works on vertices translated by the ray origin, so what it
 * accepts is the triangle perturbed by a few ulp of |p - o| (laterally and in t; grazing and just-behind-the-
 * origin configurations give accepted "hits" whose exact-arithmetic t is slightly outside (tNear, tFar]).
 * The contract defines the answer as the fp32 test over ALL triangles, so a box may only cull what the
 * test cannot accept: every box is grown, per ray, by  e = 2^-18 * (max|o| + max|scene coordinate|)
 * (>= 8x the test's error bound) by testing it against the origins o + e / o - e (ray_pad below).
 *
 * Stack: entries are (entry distance, child).  The first PT_STACK_LDS entries of a lane live in LDS
 * (lane-interleaved, conflict free), deeper ones in a global spill area; TrivialStack keeps all of
 * them in per-lane scratch for the rare rays that shading code traces inline.
 */
#ifndef PT_TRAVERSE_H
#define PT_TRAVERSE_H

#include "pt_device.h"

namespace pt {

#ifndef PT_STACK_LDS
#define PT_STACK_LDS    12
#endif
#define PT_STACK_SPILL  52
#define PT_STACK_MAX    (PT_STACK_LDS + PT_STACK_SPILL)

typedef float f2 __attribute__((ext_vector_type(2)));      /* packed fp32 math: v_pk_add_f32 / v_pk_mul_f32 */

struct RayPre
{
    int kx, ky, kz;
    float Sx, Sy, Sz;
    float ox, oy, oz;
};

__device__ __forceinline__ float sel3(float x, float y, float z, int k) { return k == 0 ? x : (k == 1 ? y : z); }

__device__ __forceinline__ RayPre ray_pre(float ox, float oy, float oz, float dx, float dy, float dz)
{
    RayPre r;
    int kz = 0;
    float dmax = dx;
    if (fabsf(dy) > fabsf(dmax)) { kz = 1; dmax = dy; }
    if (fabsf(dz) > fabsf(dmax)) { kz = 2; dmax = dz; }
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (dmax < 0.0f) { int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.Sx = sel3(dx, dy, dz, kx) / dmax;
    r.Sy = sel3(dx, dy, dz, ky) / dmax;
    r.Sz = 1.0f / dmax;
    r.ox = ox; r.oy = oy; r.oz = oz;
    return r;
}

/* per-ray box padding (see header): lo planes are tested from o + e, hi planes from o - e */
struct RayPad { float xp, yp, zp, xm, ym, zm; };
__device__ __forceinline__ RayPad ray_pad(float ox, float oy, float oz, float sceneMaxAbs)
{
    const float e = 0x1p-18f * (fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) + sceneMaxAbs);
    RayPad p;
    p.xp = ox + e; p.yp = oy + e; p.zp = oz + e;
    p.xm = ox - e; p.ym = oy - e; p.zm = oz - e;
    return p;
}

/*
 * Slab test in fused form: the entry/exit parameter of a plane is fma(plane, 1/d, c) with c = -(o +- e) / d computed
 * once per ray, one v_pk_fma_f32 for two boxes.  The rounding of c moves the plane by <= 2^-24 |o| and the single
 * rounding of the fma by <= 2^-24 |plane - o|, both far inside the padding e, so the test stays conservative without
 * a widening factor on the exit distance.  1/d is clamped to +-1e30 (a zero or denormal component would give
 * inf - inf = NaN in the fused form); such a ray is parallel to the slab and the test then reads "inside or outside".
 */
struct RaySlab { float ix, iy, iz, lx, ly, lz, hx, hy, hz; };
/* the same for compressed nodes (pt_device.h): per axis the NEAR plane of a box is its lo plane when 1/d >= 0 and
 * its hi plane otherwise, so the constants are kept per near / far side: plane parameter = fma(plane, i, n or f) */
struct RaySlabQ { float ix, iy, iz, nx, ny, nz, fx, fy, fz; };
__device__ __forceinline__ float slab_rcp(float d)
{
    const float i = __builtin_amdgcn_rcpf(d);               /* 1 ulp is plenty for a padded box test */
    return fabsf(i) <= 1e30f ? i : __builtin_copysignf(1e30f, d);
}
__device__ __forceinline__ RaySlab ray_slab(float ox, float oy, float oz, float dx, float dy, float dz, float sceneMaxAbs)
{
    const RayPad p = ray_pad(ox, oy, oz, sceneMaxAbs);
    RaySlab s;
    s.ix = slab_rcp(dx); s.iy = slab_rcp(dy); s.iz = slab_rcp(dz);
    s.lx = -(p.xp * s.ix); s.ly = -(p.yp * s.iy); s.lz = -(p.zp * s.iz);
    s.hx = -(p.xm * s.ix); s.hy = -(p.ym * s.iy); s.hz = -(p.zm * s.iz);
    return s;
}

__device__ __forceinline__ RaySlabQ ray_slab_q(float ox, float oy, float oz, float dx, float dy, float dz, float sceneMaxAbs)
{
    const RaySlab s = ray_slab(ox, oy, oz, dx, dy, dz, sceneMaxAbs);
    RaySlabQ q;
    q.ix = s.ix; q.iy = s.iy; q.iz = s.iz;
    q.nx = s.ix >= 0.0f ? s.lx : s.hx; q.fx = s.ix >= 0.0f ? s.hx : s.lx;
    q.ny = s.iy >= 0.0f ? s.ly : s.hy; q.fy = s.iy >= 0.0f ? s.hy : s.ly;
    q.nz = s.iz >= 0.0f ? s.lz : s.hz; q.fz = s.iz >= 0.0f ? s.hz : s.lz;
    return q;
}

/* true and t,u,v written when the triangle is hit inside (tNear, tFar] */
__device__ __forceinline__ bool tri_test(const RayPre& r, float4 p0, float4 p1, float4 p2, float tNear, float tFar,
    float* tOut, float* uOut, float* vOut)
{
    const float a0 = p0.x - r.ox, a1 = p0.y - r.oy, a2 = p0.z - r.oz;
    const float b0 = p1.x - r.ox, b1 = p1.y - r.oy, b2 = p1.z - r.oz;
    const float c0 = p2.x - r.ox, c1 = p2.y - r.oy, c2 = p2.z - r.oz;
    const float akx = sel3(a0, a1, a2, r.kx), aky = sel3(a0, a1, a2, r.ky), akz = sel3(a0, a1, a2, r.kz);
    const float bkx = sel3(b0, b1, b2, r.kx), bky = sel3(b0, b1, b2, r.ky), bkz = sel3(b0, b1, b2, r.kz);
    const float ckx = sel3(c0, c1, c2, r.kx), cky = sel3(c0, c1, c2, r.ky), ckz = sel3(c0, c1, c2, r.kz);
    const float Ax = akx - r.Sx * akz, Ay = aky - r.Sy * akz;
    const float Bx = bkx - r.Sx * bkz, By = bky - r.Sy * bkz;
    const float Cx = ckx - r.Sx * ckz, Cy = cky - r.Sy * ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f)
    {
        const double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        const double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        const double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = U + V + W;
    if (det == 0.0f) return false;
    const float Az = r.Sz * akz, Bz = r.Sz * bkz, Cz = r.Sz * ckz;
    const float T = U * Az + V * Bz + W * Cz;
    const float t = T / det;
    if (!(t > tNear && t <= tFar)) return false;
    *tOut = t;
    *uOut = V / det;
    *vOut = W / det;
    return true;
}

/* the same test with the two barycentric divisions left to the caller: u = V / det, v = W / det (the batched kernel
 * divides once per ray, for the hit that won, instead of once per accepted triangle) */
__device__ __forceinline__ bool tri_test_raw(const RayPre& r, float4 p0, float4 p1, float4 p2, float tNear, float tFar,
    float* tOut, float* vNum, float* wNum, float* detOut)
{
    const float a0 = p0.x - r.ox, a1 = p0.y - r.oy, a2 = p0.z - r.oz;
    const float b0 = p1.x - r.ox, b1 = p1.y - r.oy, b2 = p1.z - r.oz;
    const float c0 = p2.x - r.ox, c1 = p2.y - r.oy, c2 = p2.z - r.oz;
    const float akx = sel3(a0, a1, a2, r.kx), aky = sel3(a0, a1, a2, r.ky), akz = sel3(a0, a1, a2, r.kz);
    const float bkx = sel3(b0, b1, b2, r.kx), bky = sel3(b0, b1, b2, r.ky), bkz = sel3(b0, b1, b2, r.kz);
    const float ckx = sel3(c0, c1, c2, r.kx), cky = sel3(c0, c1, c2, r.ky), ckz = sel3(c0, c1, c2, r.kz);
    const float Ax = akx - r.Sx * akz, Ay = aky - r.Sy * akz;
    const float Bx = bkx - r.Sx * bkz, By = bky - r.Sy * bkz;
    const float Cx = ckx - r.Sx * ckz, Cy = cky - r.Sy * ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f)
    {
        const double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        const double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        const double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = U + V + W;
    if (det == 0.0f) return false;
    const float Az = r.Sz * akz, Bz = r.Sz * bkz, Cz = r.Sz * ckz;
    const float T = U * Az + V * Bz + W * Cz;
    const float t = T / det;
    if (!(t > tNear && t <= tFar)) return false;
    *tOut = t; *vNum = V; *wNum = W; *detOut = det;
    return true;
}

__device__ __forceinline__ float4 as_f4(uint4 w) { return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)); }

__device__ __forceinline__ void hit_consider(DHit& h, int32_t prim, float t, float u, float v)
{
    if (t < h.t || (t == h.t && (h.prim < 0 || prim < h.prim)))
    {
        h.prim = prim; h.t = t; h.u = u; h.v = v;
    }
}

/* ---- stacks ---- */

struct TrivialStack
{
    uint2 e[PT_STACK_MAX];
    int sp;
    __device__ __forceinline__ void init() { sp = 0; }
    __device__ __forceinline__ bool push(float t, int32_t c)
    {
        if (sp >= PT_STACK_MAX) return false;
        e[sp++] = make_uint2(__float_as_uint(t), (uint32_t)c);
        return true;
    }
    __device__ __forceinline__ bool pop(float* t, int32_t* c)
    {
        if (sp == 0) return false;
        const uint2 v = e[--sp];
        *t = __uint_as_float(v.x); *c = (int32_t)v.y;
        return true;
    }
};

/* entries are one 64-bit word: child in the high half, entry distance (float bits) in the low half */
/* typed address spaces: the LDS part is read and written with ds_* instructions, the spill part with global_*,
 * and the compiler cannot fold the two into one generic-address access */
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(1))) unsigned long long glb_u64;
/* LDSN: entries of a lane kept in LDS (the kernels' LdsStack has PT_STACK_LDS; the queued batched kernel of trace.hip fewer) */
template <int LDSN>
struct LdsStackT
{
    lds_u64* lds;               /* &shared[threadIdx.x], stride ldsStride */
    glb_u64* spill;             /* &spill[global lane slot], stride spillStride */
    int ldsStride;
    int64_t spillStride;
    int sp;
    int cap;                    /* entries the stack may hold: PT_STACK_MAX, or fewer (>= LDSN) when a test lowers it
                                   (PIMGPU_STACK_CAP) to exercise the overflow report without a 64-level tree */
    __device__ __forceinline__ static unsigned long long pack(float t, int32_t c) { return ((unsigned long long)(uint32_t)c << 32) | (unsigned long long)__float_as_uint(t); }
    __device__ __forceinline__ void init() { sp = 0; }
    __device__ __forceinline__ bool push(float t, int32_t c)
    {
        if (sp >= cap) return false;
        if (sp < LDSN) lds[sp * ldsStride] = pack(t, c);
        else spill[(int64_t)(sp - LDSN) * spillStride] = pack(t, c);
        ++sp;
        return true;
    }
    __device__ __forceinline__ bool pop(float* t, int32_t* c)
    {
        if (sp == 0) return false;
        --sp;
        /* the LDS read is unconditional (clamped slot) and the rare spill read sits under its own branch, so the
         * common case is one ds_read_b64 and never a generic-address load */
        unsigned long long v = lds[(sp < LDSN ? sp : LDSN - 1) * ldsStride];
        if (sp >= LDSN) v = spill[(int64_t)(sp - LDSN) * spillStride];
        *t = __uint_as_float((uint32_t)v); *c = (int32_t)(v >> 32);
        return true;
    }
    /* pop that stops at `base`: the entries below it have been handed to other lanes (traverse_coop) */
    __device__ __forceinline__ bool popb(int base, float* t, int32_t* c)
    {
        if (sp <= base) return false;
        return pop(t, c);
    }
    /* up to three entries, farthest first, d1 <= d2 <= d3 with misses (+inf) last; the common case (all of them fit
     * the LDS part) is three predicated 64-bit LDS stores without a branch per entry */
    __device__ __forceinline__ bool push3(float d1, int32_t c1, float d2, int32_t c2, float d3, int32_t c3)
    {
        const float inf = __builtin_inff();
        const int n = (d1 < inf ? 1 : 0) + (d2 < inf ? 1 : 0) + (d3 < inf ? 1 : 0);
        if (sp + n <= LDSN)
        {
            if (n >= 3) lds[sp * ldsStride] = pack(d3, c3);
            if (n >= 2) lds[(sp + n - 2) * ldsStride] = pack(d2, c2);
            if (n >= 1) lds[(sp + n - 1) * ldsStride] = pack(d1, c1);
            sp += n;
            return true;
        }
        bool ok = true;
        if (d3 < inf) ok &= push(d3, c3);
        if (d2 < inf) ok &= push(d2, c2);
        if (d1 < inf) ok &= push(d1, c1);
        return ok;
    }
};

typedef LdsStackT<PT_STACK_LDS> LdsStack;

#define PT_SWAP_IF(da, ca, db, cb) { const bool s_ = (db) < (da); const float td_ = s_ ? (db) : (da); const float tD_ = s_ ? (da) : (db); \
    const int32_t tc_ = s_ ? (cb) : (ca); const int32_t tC_ = s_ ? (ca) : (cb); (da) = td_; (db) = tD_; (ca) = tc_; (cb) = tC_; }

/*
 * ANY = false: closest hit (see contract above).  ANY = true: returns at the first accepted hit.
 * overflow: set to true when the stack was full and a subtree had to be dropped (the caller reports it).
 */
/* per-lane work counters of the measurement build (COUNT = true): nodes fetched (64 B each, the compressed copy), triangles tested (48 B each) */
struct TraverseCounts { uint32_t nodes, tris; };

/*
 * One memory round trip per step, whatever the lane is at.  The callers of this function (the tail kernel, the stages'
 * inline rays, set-up) are latency chains: a wave's lanes sit at inner nodes and at leaves at the same time, and a loop
 * with a node branch and a leaf branch pays the two branches' load latencies one after the other on every trip (measured
 * in k_tail: 2.5 us per step on the 1 M-triangle scene).  Here a step loads 64 bytes from ONE address -- the compressed node,
 * or the leaf's first triangle and the first vertex of its second -- plus the rest of a second triangle under a
 * predicate, all issued before any of it is used; the node arithmetic and the triangle tests then run on registers.
 * (A leaf's last triangle may be the last of the array: the builder pads tris[] by one triangle.)
 */
template <bool ANY, class Stack, bool COUNT = false>
__device__ __forceinline__ DHit traverse(const DBvh& bvh, float ox, float oy, float oz, float dx, float dy, float dz,
    float tNear, float tFar, Stack& st, bool& overflow, TraverseCounts* counts = nullptr)
{
    DHit h; h.prim = -1; h.t = tFar; h.u = 0.0f; h.v = 0.0f;
    if (!(tFar > tNear)) return h;
    const RayPre r = ray_pre(ox, oy, oz, dx, dy, dz);
    st.init();
    int32_t cur = 0;
    const uint4* __restrict__ cnodes = bvh.cnodes;
    const RaySlabQ sl = ray_slab_q(ox, oy, oz, dx, dy, dz, bvh.nodes[7].x);
    const bool px = sl.ix >= 0.0f, py = sl.iy >= 0.0f, pz = sl.iz >= 0.0f;
    const uint4* __restrict__ tris = (const uint4*)bvh.tris;
    for (;;)
    {
        const bool inner = cur >= 0;
        const uint32_t code = (uint32_t)~cur;
        const uint32_t first = code >> 3, cnt = inner ? 0u : (code & 7u) + 1u;
        const uint4* q = inner ? cnodes + (size_t)cur * 4 : tris + (size_t)first * 3;
        const uint4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
        uint4 w4 = make_uint4(0u, 0u, 0u, 0u), w5 = w4;
        if (cnt > 1u) { w4 = q[4]; w5 = q[5]; }
        int32_t next = PT_BVH_EMPTY;
        if (inner)
        {
            if (COUNT) counts->nodes++;
            /* the 64-byte compressed node (pt_device.h), decoded as in the batched kernel (trace.hip) */
            float d[4]; int32_t c[4];
            c[0] = (int32_t)w3.x; c[1] = (int32_t)w3.y; c[2] = (int32_t)w3.z; c[3] = (int32_t)w3.w;
            const float ax = __uint_as_float(w0.w) * sl.ix, ay = __uint_as_float(w2.z) * sl.iy, az = __uint_as_float(w2.w) * sl.iz;
            const float bnx = __builtin_fmaf(__uint_as_float(w0.x), sl.ix, sl.nx), bfx = __builtin_fmaf(__uint_as_float(w0.x), sl.ix, sl.fx);
            const float bny = __builtin_fmaf(__uint_as_float(w0.y), sl.iy, sl.ny), bfy = __builtin_fmaf(__uint_as_float(w0.y), sl.iy, sl.fy);
            const float bnz = __builtin_fmaf(__uint_as_float(w0.z), sl.iz, sl.nz), bfz = __builtin_fmaf(__uint_as_float(w0.z), sl.iz, sl.fz);
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
                const float t1 = fminf(fminf(tfx, tfy), fminf(tfz, h.t));
                d[k] = (t0 <= t1 && c[k] != PT_BVH_EMPTY) ? t0 : __builtin_inff();
            }
            /* sort the 4 candidates by entry distance (5-comparator network); misses carry +inf */
            PT_SWAP_IF(d[0], c[0], d[1], c[1]);
            PT_SWAP_IF(d[2], c[2], d[3], c[3]);
            PT_SWAP_IF(d[0], c[0], d[2], c[2]);
            PT_SWAP_IF(d[1], c[1], d[3], c[3]);
            PT_SWAP_IF(d[1], c[1], d[2], c[2]);
            if (d[3] < __builtin_inff()) overflow |= !st.push(d[3], c[3]);
            if (d[2] < __builtin_inff()) overflow |= !st.push(d[2], c[2]);
            if (d[1] < __builtin_inff()) overflow |= !st.push(d[1], c[1]);
            if (d[0] < __builtin_inff()) next = c[0];
        }
        else
        {
            if (COUNT) counts->tris += cnt;
            float t, u, v;
            if (tri_test(r, as_f4(w0), as_f4(w1), as_f4(w2), tNear, h.t, &t, &u, &v))
            {
                hit_consider(h, (int32_t)w0.w, t, u, v);
                if (ANY) return h;
            }
            if (cnt > 1u && tri_test(r, as_f4(w3), as_f4(w4), as_f4(w5), tNear, h.t, &t, &u, &v))
            {
                hit_consider(h, (int32_t)w3.w, t, u, v);
                if (ANY) return h;
            }
#if PT_LEAF_MAX > 2
            for (uint32_t k = 2; k < cnt; ++k)
            {
                const uint4* tp = tris + (size_t)(first + k) * 3;
                const uint4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                if (tri_test(r, as_f4(p0), as_f4(p1), as_f4(p2), tNear, h.t, &t, &u, &v))
                {
                    hit_consider(h, (int32_t)p0.w, t, u, v);
                    if (ANY) return h;
                }
            }
#endif
        }
        if (next == PT_BVH_EMPTY)
        {
            /* pop, culling entries that start beyond the current closest hit */
            float t; int32_t c;
            bool got = false;
            while (st.pop(&t, &c))
            {
                if (t <= h.t) { got = true; break; }
            }
            if (!got) break;
            next = c;
        }
        cur = next;
    }
    return h;
}


/*
 * The same walk, shared by the lanes of a wave.  Called by every lane of the (converged) wave, with or without a ray of its
 * own (valid = false: a helper).  A wave that runs one ray per lane in lock-step takes as many steps as its LONGEST ray
 * (measured in k_tail: 85 against a mean of 43, with 35 of 64 lanes holding a ray at all), and every step costs the whole
 * wave its instructions.  Here a lane that has run out of work takes the BOTTOM entry of a busy lane's stack -- the ray
 * travels by cross-lane reads, the entry through LDS -- and walks that subtree for the ray's owner.  The closest hit of a ray
 * is kept per OWNER in LDS as one 64-bit key (t bits, prim): an atomic minimum on it IS the closest-hit rule of the contract
 * (smaller t, then smaller prim; t > 0, so the float bits order as integers), every lane that works for the owner culls
 * against it, and the lane whose key stands after the minimum writes its barycentrics next to it.  Which lane meets which
 * triangle changes nothing of the result (contract), so this returns the bits of traverse<false>.
 */
typedef __attribute__((address_space(3))) int lds_i32;
struct CoopArea
{
    lds_u64* best;      /* [64] per owner lane: (t bits << 32) | prim, prim = 0x7fffffff: no hit yet, t = tFar */
    lds_u64* uv;        /* [64] per owner lane: (v bits << 32) | u bits of the hit in `best` */
    lds_i32* list;      /* [64] scratch: lanes that can give work away */
};
#ifndef PT_COOP_LEAF_VOTE
#define PT_COOP_LEAF_VOTE 16
#endif
#ifndef PT_STEAL_MIN
#define PT_STEAL_MIN 8      /* lanes out of work before the wave spends a trip on handing work over */
#endif

/* lanes of one wave hand data to each other through LDS: the hardware runs a wave's LDS instructions in order, this keeps the
 * compiler from moving them across the hand-over point (a wavefront-scope fence emits no cache operation) */
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__device__ __forceinline__ int wave_rank(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

template <bool COUNT = false>
__device__ __forceinline__ DHit traverse_coop(const DBvh& bvh, bool valid, float ox, float oy, float oz, float dx, float dy, float dz,
    float tNear, float tFar, LdsStack& st, bool& overflow, const CoopArea& ca, TraverseCounts* counts = nullptr)
{
    const int lane = (int)(threadIdx.x & 63);
    valid = valid && tFar > tNear;
    RayPre r = {}; RaySlabQ sl = {};
    if (valid) { r = ray_pre(ox, oy, oz, dx, dy, dz); sl = ray_slab_q(ox, oy, oz, dx, dy, dz, bvh.nodes[7].x); }
    int owner = lane, base = 0;
    st.init();
    ca.best[lane] = ((unsigned long long)__float_as_uint(tFar) << 32) | 0x7fffffffull;
    int32_t cur = valid ? 0 : PT_BVH_EMPTY;
    const uint4* __restrict__ cnodes = bvh.cnodes;
    const uint4* __restrict__ tris = (const uint4*)bvh.tris;
    for (;;)
    {
        const unsigned long long all = __ballot(true);
        const unsigned long long idle = __ballot(cur == PT_BVH_EMPTY);
        if (idle == all) break;
        if (COUNT && lane == 0) counts->nodes++;          /* measurement build: trips of the wave (lane 0 is always in the call) */
        if ((int)__popcll(idle) >= PT_STEAL_MIN)
        {
            const bool giver = cur != PT_BVH_EMPTY && base < (st.sp < PT_STACK_LDS ? st.sp : PT_STACK_LDS);
            const unsigned long long givers = __ballot(giver);
            if (givers != 0ull)
            {
                /* taker i takes entry (i / givers) from the bottom of giver (i mod givers), while that giver has one */
                const int nGivers = (int)__popcll(givers), nIdle = (int)__popcll(idle);
                const int rankG = wave_rank(givers), rankI = wave_rank(idle);
                if (giver) ca.list[rankG] = lane;
                wave_sync();
                const bool asks = cur == PT_BVH_EMPTY;
                const int from = asks ? ca.list[rankI % nGivers] : lane;
                const int depth = rankI / nGivers;
                const int avail = giver ? (st.sp < PT_STACK_LDS ? st.sp : PT_STACK_LDS) - base : 0;
                /* the giver's ray, as it has prepared it, its owner and the bottom of its stack */
                const int fkx = __shfl(r.kx, from, 64), fky = __shfl(r.ky, from, 64), fkz = __shfl(r.kz, from, 64);
                const float fSx = __shfl(r.Sx, from, 64), fSy = __shfl(r.Sy, from, 64), fSz = __shfl(r.Sz, from, 64);
                const float fox = __shfl(r.ox, from, 64), foy = __shfl(r.oy, from, 64), foz = __shfl(r.oz, from, 64);
                const float fix = __shfl(sl.ix, from, 64), fiy = __shfl(sl.iy, from, 64), fiz = __shfl(sl.iz, from, 64);
                const float fnx = __shfl(sl.nx, from, 64), fny = __shfl(sl.ny, from, 64), fnz = __shfl(sl.nz, from, 64);
                const float ffx = __shfl(sl.fx, from, 64), ffy = __shfl(sl.fy, from, 64), ffz = __shfl(sl.fz, from, 64);
                const float fNear = __shfl(tNear, from, 64);
                const int fOwner = __shfl(owner, from, 64), fBase = __shfl(base, from, 64), fAvail = __shfl(avail, from, 64);
                if (asks && depth < fAvail)
                {
                    const unsigned long long e = st.lds[(fBase + depth) * st.ldsStride + (from - lane)];
                    r.kx = fkx; r.ky = fky; r.kz = fkz; r.Sx = fSx; r.Sy = fSy; r.Sz = fSz; r.ox = fox; r.oy = foy; r.oz = foz;
                    sl.ix = fix; sl.iy = fiy; sl.iz = fiz; sl.nx = fnx; sl.ny = fny; sl.nz = fnz; sl.fx = ffx; sl.fy = ffy; sl.fz = ffz;
                    tNear = fNear; owner = fOwner;
                    cur = (int32_t)(e >> 32);
                }
                if (giver)
                {
                    const int askers = rankG < nIdle ? (nIdle - rankG + nGivers - 1) / nGivers : 0;      /* idle lanes i with i mod givers == rankG */
                    base += askers < avail ? askers : avail;
                }
                wave_sync();
            }
        }
#if PT_COOP_LEAF_VOTE > 0
        /* leaves wait until enough lanes stand at one (or no lane has a node): the triangle code costs the whole wave its
         * instructions on every trip it runs */
        const unsigned long long atLeaf = __ballot(cur < 0 && cur != PT_BVH_EMPTY);
        const bool leafTrip = (int)__popcll(atLeaf) >= PT_COOP_LEAF_VOTE || __ballot(cur >= 0) == 0ull;
        if (cur != PT_BVH_EMPTY && (cur >= 0 || leafTrip))
#else
        if (cur != PT_BVH_EMPTY)
#endif
        {
            const bool px = sl.ix >= 0.0f, py = sl.iy >= 0.0f, pz = sl.iz >= 0.0f;
            const bool inner = cur >= 0;
            const uint32_t code = (uint32_t)~cur;
            const uint32_t first = code >> 3, cnt = inner ? 0u : (code & 7u) + 1u;
            const uint4* q = inner ? cnodes + (size_t)cur * 4 : tris + (size_t)first * 3;
            const uint4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
            uint4 w4 = make_uint4(0u, 0u, 0u, 0u), w5 = w4;
            if (cnt > 1u) { w4 = q[4]; w5 = q[5]; }
            const float lim = __uint_as_float((uint32_t)(ca.best[owner] >> 32));        /* the owner's closest hit so far, whoever found it */
            int32_t next = PT_BVH_EMPTY;
            if (inner)
            {
                float d[4]; int32_t c[4];
                c[0] = (int32_t)w3.x; c[1] = (int32_t)w3.y; c[2] = (int32_t)w3.z; c[3] = (int32_t)w3.w;
                const float ax = __uint_as_float(w0.w) * sl.ix, ay = __uint_as_float(w2.z) * sl.iy, az = __uint_as_float(w2.w) * sl.iz;
                const float bnx = __builtin_fmaf(__uint_as_float(w0.x), sl.ix, sl.nx), bfx = __builtin_fmaf(__uint_as_float(w0.x), sl.ix, sl.fx);
                const float bny = __builtin_fmaf(__uint_as_float(w0.y), sl.iy, sl.ny), bfy = __builtin_fmaf(__uint_as_float(w0.y), sl.iy, sl.fy);
                const float bnz = __builtin_fmaf(__uint_as_float(w0.z), sl.iz, sl.nz), bfz = __builtin_fmaf(__uint_as_float(w0.z), sl.iz, sl.fz);
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
                    d[k] = (t0 <= t1 && c[k] != PT_BVH_EMPTY) ? t0 : __builtin_inff();
                }
                PT_SWAP_IF(d[0], c[0], d[1], c[1]);
                PT_SWAP_IF(d[2], c[2], d[3], c[3]);
                PT_SWAP_IF(d[0], c[0], d[2], c[2]);
                PT_SWAP_IF(d[1], c[1], d[3], c[3]);
                PT_SWAP_IF(d[1], c[1], d[2], c[2]);
                if (d[3] < __builtin_inff()) overflow |= !st.push(d[3], c[3]);
                if (d[2] < __builtin_inff()) overflow |= !st.push(d[2], c[2]);
                if (d[1] < __builtin_inff()) overflow |= !st.push(d[1], c[1]);
                if (d[0] < __builtin_inff()) next = c[0];
            }
            else
            {
                if (COUNT) counts->tris += cnt;
                DHit h; h.prim = -1; h.t = lim; h.u = 0.0f; h.v = 0.0f;
                float t, u, v;
                if (tri_test(r, as_f4(w0), as_f4(w1), as_f4(w2), tNear, h.t, &t, &u, &v)) hit_consider(h, (int32_t)w0.w, t, u, v);
                if (cnt > 1u && tri_test(r, as_f4(w3), as_f4(w4), as_f4(w5), tNear, h.t, &t, &u, &v)) hit_consider(h, (int32_t)w3.w, t, u, v);
#if PT_LEAF_MAX > 2
                for (uint32_t k = 2; k < cnt; ++k)
                {
                    const uint4* tp = tris + (size_t)(first + k) * 3;
                    const uint4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                    if (tri_test(r, as_f4(p0), as_f4(p1), as_f4(p2), tNear, h.t, &t, &u, &v)) hit_consider(h, (int32_t)p0.w, t, u, v);
                }
#endif
                /* publish: the minimum over everything found for this owner; the lane whose key stands writes u, v */
                const unsigned long long key = ((unsigned long long)__float_as_uint(h.t) << 32) | (unsigned long long)(uint32_t)h.prim;
                if (h.prim >= 0) __hip_atomic_fetch_min((unsigned long long*)(ca.best + owner), key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                wave_sync();
                if (h.prim >= 0 && ca.best[owner] == key) ca.uv[owner] = ((unsigned long long)__float_as_uint(h.v) << 32) | (unsigned long long)__float_as_uint(h.u);
            }
            if (next == PT_BVH_EMPTY)
            {
                const float lim2 = __uint_as_float((uint32_t)(ca.best[owner] >> 32));
                float t; int32_t c;
                while (st.popb(base, &t, &c))
                {
                    if (t <= lim2) { next = c; break; }
                }
                if (next == PT_BVH_EMPTY) { st.sp = 0; base = 0; }
            }
            cur = next;
        }
    }
    wave_sync();
    const unsigned long long k = ca.best[lane];
    DHit h; h.prim = -1; h.t = tFar; h.u = 0.0f; h.v = 0.0f;
    if ((uint32_t)k != 0x7fffffffu)
    {
        const unsigned long long b = ca.uv[lane];
        h.prim = (int32_t)(uint32_t)k; h.t = __uint_as_float((uint32_t)(k >> 32)); h.u = __uint_as_float((uint32_t)b); h.v = __uint_as_float((uint32_t)(b >> 32));
    }
    return h;
}

} /* namespace pt */

#endif
