/*
 * pim_host.h -- plain C host layer with the names, argument meaning and error behaviour of Pim's own
 * path-tracer API (rendering/path_tracer.h:67-103, rendering/lightmap.h:56-72), implemented over the
 * C-ABI of libpimgpu.so (include/pimgpu.h).  This is the "C host code calling HIP through a thin
 * C-ABI layer": a Pim build replaces rendering/path_tracer.c and the BakeFn part of
 * rendering/lightmap.c with this file's implementation (pim_host.c) and links libpimgpu.so.
 *
 * Differences from the reference signatures, all forced by the device boundary:
 *   - PtScene_New takes the scene SNAPSHOT (PimSceneDesc) instead of pulling Entities_Get(),
 *     Texture_Get(), Cubemaps_Get(), g_BrdfLut and cv_pt_dist_meters from global singletons;
 *     INTEGRATION.md shows the few lines that fill the snapshot from those singletons;
 *   - the Prng streams are per pixel / ray / texel and take an explicit seed and sample index
 *     (SURVEY D1), carried in PtTrace / the extra arguments;
 *   - errors: NULL / false returns + a message (PtSys_LastError), like the reference's
 *     NULL scene + Con_Logf (path_tracer.c:389-396, 645-651, 1085-1089).
 */
#ifndef PIM_HOST_H
#define PIM_HOST_H

#include <stdbool.h>
#include <stdint.h>
#include "../../include/pimgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z, w; } float4;
typedef struct { float x, y, z; } float3;
typedef struct { int32_t x, y; } int2;
typedef int32_t i32;

typedef PimScene PtScene;                 /* opaque, path_tracer.h:13 */
typedef PimDofInfo PtDofInfo;             /* path_tracer.h:33-42 */
typedef PimCamera Camera;                 /* rendering/camera.h:8-15 */

/* PtRayHit, path_tracer.h:24-31 */
typedef struct PtRayHit { float4 wuvt; float4 normal; i32 type; i32 iVert; uint32_t flags; } PtRayHit;
/* PtResult, path_tracer.h:54-59 */
typedef struct PtResult { float3 color; float3 albedo; float3 normal; } PtResult;
/* PtResults, path_tracer.h:61-65 (caller frees with PtResults_Del: there is no Temp_ frame arena here) */
typedef struct PtResults { float4* colors; float4* directions; } PtResults;

/* PtTrace, path_tracer.h:44-52: host float3 images the caller reads after Pt_Trace (running mean) */
typedef struct PtTrace
{
    float3* color;
    float3* albedo;
    float3* normal;
    float3* denoised;
    int2 imageSize;
    float sampleWeight;
    /* build-side state */
    PimTrace* gpu;            /* the HBM-resident copies */
    uint32_t sampleIndex;     /* index of the next sample; seeds the per-pixel streams */
    uint32_t seed;
    i32 maxBounces;           /* <= 0: the reference's roulette-only termination */
    float frameDt;            /* Time_Deltaf() for DofUpdate */
} PtTrace;

/* Lightmap / LmPack, rendering/lightmap.h:27-44 (the fields BakeFn touches) */
typedef struct Lightmap { float4* probes[5]; float3* position; float3* normal; float* sampleCounts; i32 size; } Lightmap;
/* Cubemap (rendering/cubemap.h:28-34; the convolved mips are not on this path) and AmbCube_t (math/types.h:513-517) */
typedef struct Cubemap { i32 size; float3* color[6]; } Cubemap;
typedef struct AmbCube_t { float4 Values[6]; } AmbCube_t;
typedef struct LmPack { Lightmap* lightmaps; i32 lmCount; i32 lmSize; float texelsPerMeter; } LmPack;

bool PtSys_Init(void);                                   /* path_tracer.c:421-434 */
void PtSys_Shutdown(void);                               /* :439-446 */
const char* PtSys_LastError(void);

PtScene* PtScene_New(const PimSceneDesc* snapshot);      /* :1083-1093 */
/* FlattenDrawables (path_tracer.c:692-782) + PtScene_Init: the first five fields of `rest` (vertCount, matCount, positions,
 * normals, uvs, matIds, materials) are ignored and filled from the GPU-flattened draw table */
PtScene* PtScene_NewFromDrawables(const PimDrawables* drawables, const PimSceneDesc* rest);
void PtScene_Update(PtScene* scene);                     /* :1023-1037 (UpdateDists; a changed scene is a new snapshot) */
void PtScene_Del(PtScene* scene);                        /* :1095-1102 */

/* BrdfLut, math/lighting.h:45-55; BrdfLut_New / BrdfLut_Update / BrdfLut_Del, math/lighting.c:86-118.  seed: the per-texel
 * streams */
typedef struct { float x, y; } float2;
typedef struct BrdfLut { int2 size; float2* texels; } BrdfLut;
BrdfLut BrdfLut_New(int2 size, i32 numSamples, uint32_t seed);
void BrdfLut_Update(BrdfLut* lut, i32 prevSamples, i32 newSamples, uint32_t seed);
void BrdfLut_Del(BrdfLut* lut);

void PtTrace_New(PtTrace* trace, int2 imageSize);        /* :1117-1131 */
void PtTrace_Del(PtTrace* trace);                        /* :1133-1139 */
void DofInfo_New(PtDofInfo* dof);                        /* :1141-1153 */

PtRayHit Pt_Intersect(PtScene* scene, float4 ro, float4 rd, float tNear, float tFar);      /* :1466-1474 */
PtResult Pt_TraceRay(PtScene* scene, float4 ro, float4 rd, uint32_t sampleIndex, uint32_t seed, i32 maxBounces);   /* :2306-2416 */
void Pt_Trace(PtTrace* trace, PtDofInfo* dof, PtScene* scene, const Camera* camera);       /* :2557-2585 */
PtResults Pt_RayGen(PtScene* scene, float4 origin, i32 count, uint32_t sampleIndex, uint32_t seed, i32 maxBounces);   /* :2614-2644 */
void PtResults_Del(PtResults* results);
/* Cubemap_Bake, rendering/cubemap.h:209-213 / cubemap.c:162-190; AmbCube_Bake, math/ambcube.h:69-74 / ambcube.c:5-32.
 * sampleIndex / seed / maxBounces: the per-work-item streams and the optional bounce clamp */
void Cubemap_Bake(Cubemap* cm, PtScene* scene, float4 origin, float weight, uint32_t sampleIndex, uint32_t seed, i32 maxBounces);
i32 AmbCube_Bake(PtScene* scene, AmbCube_t* pCube, float4 origin, i32 samples, i32 prevSampleCount, uint32_t sampleIndex, uint32_t seed, i32 maxBounces);

/* Lightmap_New / Lightmap_Del, rendering/lightmap.c:81-133: the single zeroed allocation per page (the Vulkan texture
 * slot and upload are the renderer's); texId stands for Lightmap.slot.index (lightmap.h:30) */
void Lightmap_New(Lightmap* lm, i32 size);
void Lightmap_Del(Lightmap* lm);
/* EmbedAttributes, rendering/lightmap.c:1034-1045, the last step of LmPack_Pack (:1092): fills position / normal /
 * sampleCounts of every page from the draw table's uvs.zw and texIndices.w; texIds[lmCount] = the pages' texture ids
 * (NULL: page i has id i) */
void LmPack_EmbedAttributes(const PimDrawables* drawables, Lightmap* lightmaps, const i32* texIds, i32 lmCount);
/* rendering/lightmap.c:1204-1223; pages must be the reference's single allocation per lightmap (probes[0] = base) */
void LmPack_Bake(PtScene* scene, LmPack* pack, float timeSlice, i32 spp, uint32_t passIndex, uint32_t seed, i32 maxBounces);

#ifdef __cplusplus
}
#endif
#endif
