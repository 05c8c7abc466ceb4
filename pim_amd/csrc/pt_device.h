/*
 * pt_device.h -- device-side scene description shared by every kernel of libpimgpu.
 *
 * DScene is the HBM-resident PtScene (rendering/path_tracer.c:122-166): the six flattened
 * vertex arrays, the emissive maps, the light grid with its per-cell Dist1D arrays, materials,
 * textures, sky cubemap, BRDF LUT, media description, plus the GPU acceleration structure that
 * replaces Embree's RTCScene (path_tracer.c:618-690).  It is passed to kernels BY VALUE
 * (a few hundred bytes of kernarg), so a kernel dereferences no descriptor in memory.
 */
#ifndef PT_DEVICE_H
#define PT_DEVICE_H

#include "pt_math.h"
#include "../../include/pimgpu.h"

namespace pt {

/* build-defined RNG stream tags (the reference seeds per host thread from the clock;
 * here every work item owns a stream seeded by Pcg4 of its coordinates and one of these tags) */
#define PT_TAG_PIXEL    0x9E3779B9u
#define PT_TAG_RAY      0x7261795Fu
#define PT_TAG_BAKE     0x62616B65u
#define PT_TAG_BAKE_PT  0xA511E9B3u
#define PT_TAG_EMIT     0x454D4954u
#define PT_TAG_GRID     0x47524944u
#define PT_TAG_CUBE     0x43554245u

struct DTexture { int32_t width, height, format, _pad; const uint32_t* texels; };

/* PtMediaDesc with its derived fields, path_tracer.c:91-111,1963-1987 */
struct DMedia
{
    v4 constantColor, noiseColor;
    float constantMfp, noiseMfp;
    v4 constantMu, noiseMu;
    float rcpMajorant, absorption;
    int32_t noiseOctaves;
    float noiseGain, noiseLacunarity, noiseFreq, noiseScale, noiseHeight, noiseRange;
    float phaseDirA, phaseDirB, phaseBlend;
};

/* Grid, math/grid.h:8-15 */
struct DGrid { v4 lo, hi; int32_t sx, sy, sz; float cellsPerMeter; };

/*
 * Acceleration structure.  Wide BVH with 4 children per node ("BVH4"), 128 bytes per node so a
 * node is exactly one 128-B cache line / one 8-lane x 16-B coalesced group:
 *
 *   float4 q[0] = (lo.x[0..3])   q[1] = (lo.y[0..3])   q[2] = (lo.z[0..3])
 *   float4 q[3] = (hi.x[0..3])   q[4] = (hi.y[0..3])   q[5] = (hi.z[0..3])
 *   int4   q[6] = child[0..3]    q[7] = unused
 *
 * child >= 0: inner node index; child < 0 and != PT_BVH_EMPTY: leaf, ~child = (first << 3) | (count - 1),
 * first = index of the leaf's first triangle in tris[] (3 float4 per triangle, leaf order:
 * p0.xyz + primID bits in w, p1.xyz, p2.xyz); empty slots have an inverted box and PT_BVH_EMPTY.
 *
 * The batched traversal kernel reads a COMPRESSED copy of the same tree, 64 bytes per node (4 x 16-B loads per
 * visit instead of 7: the kernel is bound by the vector L1's one-lookup-per-lane-per-cycle rate, not by bytes):
 *
 *   uint4 c[0] = (lo.x, lo.y, lo.z, scale.x)   floats: minimum corner of the union of the child boxes, grid step
 *   uint4 c[1] = (qlo.x, qlo.y, qlo.z, qhi.x)  each dword = one plane of the 4 children, one byte per child
 *   uint4 c[2] = (qhi.y, qhi.z, scale.y, scale.z)
 *   uint4 c[3] = child[0..3]                   as above
 *
 * child plane = lo + q * scale with q in 0..255 and scale a power of two (so q * scale is exact), rounded
 * outwards (floor / ceil), i.e. every compressed box contains its fp32 box; empty slots hold qlo = 255, qhi = 0.
 */
#define PT_BVH_EMPTY  ((int32_t)0x80000000)
#ifndef PT_LEAF_MAX
#define PT_LEAF_MAX   2
#endif

struct DBvh
{
    const float4* nodes;     /* 8 float4 per node */
    const uint4* cnodes;     /* 4 uint4 per node: the compressed copy */
    const float4* tris;      /* 3 float4 per triangle, leaf order */
    int32_t nodeCount;
    int32_t triCount;
};

/*
 * One triangle as the shading stages read it: the three vertices' positions, normals and uvs, the material id and the
 * material's flags, copied from the PtScene arrays (same bits) into ONE 128-byte, 128-byte-aligned record = one cache line.
 * GetSurface needs 3 x (16 + 16 + 8) + 4 bytes of a hit triangle; from the reference's four separate arrays (48-, 48-, 24- and
 * 12-byte strides per triangle) an incoherent hit touches 6-7 memory sectors in four lines, from the record two sectors of one
 * line -- the stages are bound by HBM-side traffic at 1.6-1.8 x their algorithmic bytes (profiles/r03_synth1m_4k_*).
 * The arrays themselves stay: they are the upload format and what the set-up kernels read.
 */
struct __attribute__((aligned(128))) DTriRec
{
    v4 p0, p1, p2;
    v4 n0, n1, n2;
    v2 uv0, uv1, uv2;
    int32_t matId;          /* matIds[3 * tri] */
    uint32_t flags;         /* materials[matId].flags */
};
static_assert(sizeof(DTriRec) == 128, "one cache line per triangle");

struct DScene
{
    int32_t vertCount, triCount, matCount, texCount, emissiveCount;
    const v4* positions; const v4* normals; const v2* uvs; const int32_t* matIds;
    const int32_t* vertToEmit; const int32_t* emitToVert;
    const DTriRec* tri;         /* [triCount], built from the four arrays above at scene creation */
    const PimMaterial* materials;
    const DTexture* textures;
    int32_t skySize; const v3* sky[6];
    int32_t lutSize; const v2* lut;
    DMedia media;
    DGrid grid; int32_t gridLen;
    const int32_t* cellLength; float* pdf; float* cdf; uint32_t* live; uint32_t* cellSum; float* cellIntegral;
    DBvh bvh;
};

/* raw closest-hit record, 16 bytes: primID (-1 = miss), t, u, v (u = weight of p1, v = weight of p2) */
struct DHit { int32_t prim; float t, u, v; };

} /* namespace pt */

#endif
