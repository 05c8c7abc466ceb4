/*
 * pim_host.h -- plain C host layer with the names, SIGNATURES, struct layouts and error behaviour of Pim's own
 * path-tracer API (rendering/path_tracer.h:13-103, rendering/lightmap.h:27-72, rendering/cubemap.h:28-34,209-213,
 * math/ambcube.h:69-74), implemented over the C-ABI of libpimgpu.so (include/pimgpu.h).  This is the "C host code
 * calling HIP through a thin C-ABI layer": a Pim build drops rendering/path_tracer.c and the BakeFn part of
 * rendering/lightmap.c, compiles pim_host.c instead and links libpimgpu.so; the callers in render_system.c, cubemap.c and
 * ambcube.c compile unchanged against their own headers (tests/host/ref_caller.c is such a caller).
 *
 * PART 1 below is the reference's interface, prototype for prototype.  What the reference pulls from global singletons or
 * keeps in per-thread state has no parameter there, so it has none here either; PART 2 (extensions, `PtSys_*Ex`, `*Ex`)
 * is how the embedding code supplies it:
 *   - the scene: PtScene_New(void) / PtScene_Update read Entities_Get(), Texture_Get(), Cubemaps_Get(), g_BrdfLut,
 *     cv_pt_dist_meters (path_tracer.c:692-782,1011-1049).  Here they call the SNAPSHOT PROVIDER registered with
 *     PtSys_SetSceneProvider, which fills a PimSceneDesc from those singletons (INTEGRATION.md shows it) and reports
 *     Entities.modtime; a changed modtime rebuilds the scene in place, as PtScene_Update does (:1028-1032);
 *   - the Prng streams: per host thread and clock-seeded in the reference (path_tracer.c:168-176,410-414); per work item here
 *     (SURVEY D1), derived from the seed of PtSys_Configure and from call counters the library keeps, as the reference keeps
 *     its streams: Pt_Trace call k renders sample index k, Pt_TraceRay call k is stream k, and so on;
 *   - the bounce clamp and Time_Deltaf(): PtSys_Configure (0 = the reference's roulette-only termination; 1/60 s).
 * Errors: the reference has no error codes (NULL scene, Con_Logf + ASSERT); here a failed call logs to stderr, leaves its
 * outputs untouched / returns NULL, and PtSys_LastError() returns the message.
 */
#ifndef PIM_HOST_H
#define PIM_HOST_H

#include <stdbool.h>
#include <stdint.h>
#include "../../include/pimgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the reference's plain types (math/types.h:52-73,152-157; same size and alignment) ---- */
typedef int32_t i32;
typedef uint32_t u32;
typedef uint64_t u64;
typedef struct float2_s { _Alignas(8) float x; float y; } float2;
typedef struct float3_s { float x, y, z; } float3;
typedef struct float4_s { _Alignas(16) float x; float y, z, w; } float4;
typedef struct int2_s { _Alignas(8) i32 x; i32 y; } int2;
typedef struct quat_s { float4 v; } quat;

/* ======================================================================================================================= */
/* PART 1: the reference's interface                                                                                        */

typedef struct PtScene_s PtScene;         /* opaque, path_tracer.h:13 */

/* Camera, rendering/camera.h:8-15 */
typedef struct Camera_s { float4 position; quat rotation; float zNear; float zFar; float fovy; } Camera;

typedef enum { PtHit_Nothing = 0, PtHit_Backface, PtHit_Triangle, PtHit_COUNT } PtHitType;     /* path_tracer.h:15-22 */
typedef struct PtRayHit_s { float4 wuvt; float4 normal; PtHitType type; i32 iVert; u32 flags; } PtRayHit;     /* :24-31 */
typedef struct PtDofInfo_s      /* :33-42 */
{
    float aperture; float focalLength; i32 bladeCount; float bladeRot; float focalPlaneCurvature; float autoFocusSpeed; bool autoFocus;
} PtDofInfo;
typedef struct PtTrace_s        /* :44-52: host float3 images the caller reads right after Pt_Trace (running mean) */
{
    float3* color; float3* albedo; float3* normal; float3* denoised; int2 imageSize; float sampleWeight;
} PtTrace;
typedef struct PtResult_s { float3 color; float3 albedo; float3 normal; } PtResult;           /* :54-59 */
typedef struct PtResults_s { float4* colors; float4* directions; } PtResults;                 /* :61-65 */

void PtSys_Init(void);                                   /* path_tracer.h:67, path_tracer.c:421-434 */
void PtSys_Update(void);                                 /* :68 (frees what the frame's Pt_RayGen calls returned: MemSys_Update's role, main.c:90) */
void PtSys_Shutdown(void);                               /* :69 */

PtScene* PtScene_New(void);                              /* :71, path_tracer.c:1083-1093 */
void PtScene_Update(PtScene* scene);                     /* :72, path_tracer.c:1023-1037 */
void PtScene_Del(PtScene* scene);                        /* :73 */
void PtScene_Gui(PtScene* scene);                        /* :74: the igText lines of path_tracer.c:1104-1115, printed to stdout */

void PtTrace_New(PtTrace* trace, int2 imageSize);        /* :76, path_tracer.c:1117-1131 */
void PtTrace_Del(PtTrace* trace);                        /* :77 */

void DofInfo_New(PtDofInfo* dof);                        /* :79, path_tracer.c:1141-1153 */
void DofInfo_Gui(PtDofInfo* dof);                        /* :80: sliders in the reference; the values, printed to stdout, here */

PtRayHit Pt_Intersect(const PtScene* scene, float4 ro, float4 rd, float tNear, float tFar);   /* :82-87 */
PtResult Pt_TraceRay(PtScene* scene, float4 ro, float4 rd);                                    /* :89-92 */
void Pt_Trace(PtTrace* trace, PtDofInfo* dof, PtScene* scene, const Camera* camera);           /* :94-98 */
PtResults Pt_RayGen(PtScene* scene, float4 origin, i32 count);                                 /* :100-103; valid until PtSys_Update */

/* rendering/lightmap.h:11-72 (vkrTextureId: rendering/vulkan/vkr.h:125-131) */
#define kGiDirections 5
typedef struct vkrTextureId_s { u32 version : 8; u32 type : 3; u32 index : 21; } vkrTextureId;
typedef struct Lightmap_s
{
    float4* probes[kGiDirections]; float3* position; float3* normal; float* sampleCounts; i32 size; vkrTextureId slot;
} Lightmap;
typedef struct LmPack_s { float4 axii[kGiDirections]; Lightmap* lightmaps; i32 lmCount; i32 lmSize; float texelsPerMeter; } LmPack;
void Lightmap_New(Lightmap* lm, i32 size);               /* lightmap.h:56, lightmap.c:81-133 (the Vulkan slot stays zero: the renderer's) */
void Lightmap_Del(Lightmap* lm);                         /* :57 */
void Lightmap_NewIn(Lightmap* lm, i32 size, float* block);   /* not in the reference: Lightmap_New over a page-locked, zeroed block made ahead (NULL: as Lightmap_New) */
LmPack* LmPack_Get(void);                                /* :61: the pack LmPack_Bake works on (ms_pack, lightmap.c:79) */
void LmPack_Del(LmPack* pack);                           /* :67 */
void LmPack_Bake(PtScene* scene, float timeSlice, i32 spp);   /* :69, lightmap.c:1204-1223 */
LmPack LmPack_Pack(i32 atlasSize, float texelsPerUnit, float distThresh, float degThresh);   /* :62-66, lightmap.c:1047-1102 (pim_lmpack.c) */
void Lightmap_Upload(Lightmap* lm);                      /* :59, lightmap.c:135-144: the kGiDirections probe layers go to the uploader below */

/* the lightmap checkpoint: common/guid.h:7-12,24-26, io/types.h:36-39, assets/crate.h:9-33, lightmap.h:11,46-54,71-72
 * (pim_crate.c; the file it writes is the reference's, byte for byte) */
typedef struct Guid_s { _Alignas(16) u64 a; u64 b; } Guid;
bool Guid_IsNull(Guid x);
bool Guid_Equal(Guid lhs, Guid rhs);
Guid Guid_FromStr(const char* str);
Guid Guid_HashStr(const char* str);
typedef struct FStream_s { void* handle; } FStream;
enum { kCrateLen = 1024 };
typedef struct Crate_s { FStream file; Guid ids[kCrateLen]; i32 offsets[kCrateLen]; i32 sizes[kCrateLen]; } Crate;
bool Crate_Open(Crate* crate, const char* path);
bool Crate_Close(Crate* crate);
bool Crate_Get(Crate* crate, Guid id, void* dst, i32 size);
bool Crate_Set(Crate* crate, Guid id, const void* src, i32 size);
bool Crate_Rm(Crate* crate, Guid id);
bool Crate_Stat(const Crate* crate, Guid id, i32* offsetOut, i32* sizeOut);
#define kLmPackVersion 2
typedef struct DiskLmPack_s { i32 version; i32 directions; i32 lmCount; i32 lmSize; i32 bytesPerLightmap; float texelsPerMeter; } DiskLmPack;
bool LmPack_Save(Crate* crate, const LmPack* src);
bool LmPack_Load(Crate* crate, LmPack* dst);

/* rendering/cubemap.h:28-34,209-213 and math/types.h:513-517, math/ambcube.h:69-74 */
typedef struct Cubemap_s { i32 size; i32 mipCount; float3* color[6]; float4* convolved[6]; } Cubemap;
typedef struct AmbCube_s { float4 Values[6]; } AmbCube_t;
void Cubemap_New(Cubemap* cm, i32 size);                                    /* cubemap.c:96-115 */
void Cubemap_Del(Cubemap* cm);                                              /* cubemap.c:117-128 */
void Cubemap_Bake(Cubemap* cm, PtScene* scene, float4 origin, float weight);
void Cubemap_Convolve(Cubemap* cm, u32 sampleCount, float weight);          /* cubemap.c:264-303 */
i32 AmbCube_Bake(PtScene* scene, AmbCube_t* pCube, float4 origin, i32 samples, i32 prevSampleCount);

/* math/lighting.h:45-55, math/types.h:503-509; BrdfLut_New / BrdfLut_Update / BrdfLut_Del, math/lighting.c:86-118 */
typedef struct BrdfLut_s { int2 size; float2* texels; } BrdfLut;
BrdfLut BrdfLut_New(int2 size, i32 numSamples);
void BrdfLut_Del(BrdfLut* lut);
void BrdfLut_Update(BrdfLut* lut, i32 prevSamples, i32 newSamples);

/* ======================================================================================================================= */
/* PART 2: extensions -- what the reference takes from global singletons or per-thread state                                */

const char* PtSys_LastError(void);
bool PtSys_Ok(void);                                     /* PtSys_Init succeeded (the reference asserts) */

/* The snapshot provider: fills *out from the engine's singletons and returns true; *modtime = Entities_Get()->modtime.
 * The arrays it points to must stay valid until the provider is called again (PtScene_New / a PtScene_Update that finds a
 * new modtime copy them to HBM before returning).  peekModtime != 0: only *modtime is wanted. */
typedef bool (*PtSceneProviderFn)(PimSceneDesc* out, uint64_t* modtime, int peekModtime, void* user);
void PtSys_SetSceneProvider(PtSceneProviderFn fn, void* user);

typedef struct PtSysConfig
{
    uint32_t seed;            /* of every per-work-item Prng stream */
    i32 maxBounces;           /* <= 0: the reference's termination, russian roulette only */
    float frameDt;            /* Time_Deltaf() as DofUpdate sees it (path_tracer.c:2497); <= 0: 1/60 s */
    i32 downloadAovs;         /* 0: Pt_Trace brings back `color` only; albedo / normal (read by the denoiser alone,
                                 render_system.c:306-324) are fetched by PtTrace_DownloadAovs on demand.  Default 1. */
    i32 device;               /* HIP device; < 0: environment PIM_GPU_DEVICE, else 0 */
} PtSysConfig;
void PtSys_Configure(const PtSysConfig* config);         /* before or after PtSys_Init; also resets the call counters below */
/* the counters that stand in for the reference's ever-advancing per-thread streams: index of the next Pt_Trace sample,
 * Pt_TraceRay stream, Pt_RayGen / AmbCube_Bake batch, Cubemap_Bake pass, LmPack_Bake pass */
typedef struct PtSysCounters { uint32_t trace, traceRay, rayGen, cubemap, lmBake, convolve; } PtSysCounters;
void PtSys_GetCounters(PtSysCounters* out);
void PtTrace_DownloadAovs(PtTrace* trace);               /* with downloadAovs == 0: bring albedo / normal up to date now */

/* LmPack_Pack reads Entities_Get() / Mesh_Get() (lightmap.c:725-741) and writes the lightmap uvs and texture ids it assigns INTO
 * the meshes (uvs.zw, texIndices.w, :919-930).  The provider fills *out with the draw table; its mesh arrays are the engine's
 * own, writable, float4 uvs[length] and int4 texIndices[length]. */
typedef bool (*PtDrawablesProviderFn)(PimDrawables* out, void* user);
void PtSys_SetDrawablesProvider(PtDrawablesProviderFn fn, void* user);
uint32_t PtSys_Seed(void);                               /* PtSysConfig.seed as configured */
/* the same with the draw table and the seed of chart_split's streams passed in; a zeroed pack and LmPack_LastError() on failure */
LmPack LmPack_PackEx(const PimDrawables* drawables, i32 atlasSize, float texelsPerUnit, float distThresh, float degThresh, uint32_t seed);
const char* LmPack_LastError(void);
/* the packer's sequential steps on their own (tests): chart_group with chart_split (lightmap.c:563-636; nodes are reordered chart
 * by chart, chartFirst[count + 1], returns the chart count or -1) and chart_sort (:648-660; order[i] = chart in place i) */
i32 LmPack_GroupCharts(PimChartNode* nodes, i32 nodeCount, float distThresh, float degThresh, float maxWidth, uint32_t seed, i32* chartFirst);
void LmPack_SortCharts(const float* areas, i32 chartCount, i32* order);

/* vkrTexTable_Upload's stand-in (rendering/vulkan/vkr_textable.h; lightmap.c:142): Lightmap_Upload hands it layer `layer` of the
 * page's 2D-array texture, size x size RGBA32F = `bytes` bytes at `data`.  None registered: Lightmap_Upload does nothing. */
typedef void (*LmUploadFn)(vkrTextureId slot, i32 layer, const void* data, i32 bytes, void* user);
void PtSys_SetLightmapUploader(LmUploadFn fn, void* user);

/* LmPack_Bake with the pack's pages kept in HBM between calls (pimgpu_lightmap_set_*).  The reference bakes in place in host
 * memory; its callers read the pages only in Lightmap_Upload (every 10 s, render_system.c:200-210) and LmPack_Save, so:
 *   0  every call copies the pages to the device and probes | sampleCounts back (default: host pages are ALWAYS current, and a
 *      caller may edit them between calls);
 *   1  the pages go up once, every call copies probes | sampleCounts back (host pages current after every call);
 *   2  nothing crosses PCIe in LmPack_Bake; Lightmap_Upload, LmPack_Save and LmPack_Sync bring the results to the host.
 * In modes 1 and 2 a caller that writes the pages itself (not through LmPack_EmbedAttributes / LmPack_Load, which know)
 * calls LmPack_Invalidate afterwards. */
void LmPack_SetResident(i32 mode);
void LmPack_Sync(void);
void LmPack_Invalidate(void);

/* explicit-argument forms (tests, tools): the same work with the counters' values passed in */
PtScene* PtScene_NewFromSnapshot(const PimSceneDesc* snapshot);
/* FlattenDrawables (path_tracer.c:692-782) + PtScene_Init: the vertex / material fields of `rest` are ignored and filled
 * from the GPU-flattened draw table */
PtScene* PtScene_NewFromDrawables(const PimDrawables* drawables, const PimSceneDesc* rest);
PimScene* PtScene_Gpu(const PtScene* scene);             /* the C-ABI handle behind the scene */
PtResult Pt_TraceRayEx(PtScene* scene, float4 ro, float4 rd, uint32_t sampleIndex, uint32_t seed, i32 maxBounces);
/* EmbedAttributes, rendering/lightmap.c:1034-1045, the last step of LmPack_Pack (:1092): fills position / normal /
 * sampleCounts of every page from the draw table's uvs.zw and texIndices.w; texIds[lmCount] = the pages' texture ids
 * (NULL: page i has id i) */
void LmPack_EmbedAttributes(const PimDrawables* drawables, Lightmap* lightmaps, const i32* texIds, i32 lmCount);

#ifdef __cplusplus
}
#endif
#endif
