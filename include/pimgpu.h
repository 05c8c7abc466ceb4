/*
 * pimgpu.h -- C-ABI of the MI355X path-tracing hot path (libpimgpu.so).
 *
 * This is the ONE process/device boundary of the build: plain C, plain
 * pointers and sizes, int status codes.  Everything above it is host C that
 * mirrors Pim's own path-tracer API (pim_amd/host/pim_host.h); everything
 * below it is hand-written HIP for gfx950.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to the Pim source tree, src/...).
 *
 * Conventions
 *   - every function returns PIMGPU_OK (0) or a negative PIMGPU_E_* code;
 *     pimgpu_last_error() returns a human readable message for the calling
 *     thread's last failure (the reference has no error codes: NULL scene /
 *     Con_Logf, path_tracer.c:389-396,645-651,1085-1089);
 *   - all pointers are caller owned; "host" pointers are ordinary host memory,
 *     "dev" pointers are device (HBM) addresses obtained by the caller
 *     (hipMalloc / torch);
 *   - calls are synchronous from one host thread, like the reference's
 *     main-thread entry points (path_tracer.h:67-103);
 *   - float4 arrays are 4 packed floats per element, float3 3, float2 2.
 */
#ifndef PIMGPU_H
#define PIMGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIMGPU_OK            0
#define PIMGPU_E_INVALID    -1   /* bad argument (NULL, negative size, shape mismatch) */
#define PIMGPU_E_DEVICE     -2   /* HIP runtime error / no gfx950 device */
#define PIMGPU_E_NOMEM      -3   /* hipMalloc failed */
#define PIMGPU_E_STATE      -4   /* call order violated (not initialised ...) */
#define PIMGPU_E_LIMIT      -5   /* size beyond a documented maximum */

/* MatFlag bits, rendering/material.h:8-20 */
#define PIM_MATFLAG_EMISSIVE   (1u << 0)
#define PIM_MATFLAG_SKY        (1u << 1)
#define PIM_MATFLAG_LAVA       (1u << 4)
#define PIM_MATFLAG_REFRACTIVE (1u << 5)

/* PtHitType, rendering/path_tracer.h:15-22 */
#define PIM_HIT_NOTHING  0
#define PIM_HIT_BACKFACE 1
#define PIM_HIT_TRIANGLE 2

/* texture formats the path reads (rendering/sampler.h:370-392) */
#define PIM_TEX_RGBA8 0   /* R8G8B8A8 with the cubic sRGB fit decode */
#define PIM_TEX_XY16  1   /* short2 tangent-space normal */

/* Material, rendering/material.h:22-31; TextureId -> index into textures[], -1 = none */
typedef struct PimMaterial
{
    int32_t  albedo;
    int32_t  rome;
    int32_t  normal;
    uint32_t flags;
    float    meanFreePath[4];
    float    ior;
    float    bumpiness;
    float    _pad[2];
} PimMaterial;                       /* 48 bytes */

/* Texture, rendering/texture.h (size, texels, format) */
typedef struct PimTexture
{
    int32_t     width;
    int32_t     height;
    int32_t     format;              /* PIM_TEX_* */
    int32_t     _pad;
    const void* texels;              /* host, width*height*4 bytes */
} PimTexture;

/* user-facing part of PtMediaDesc, path_tracer.c:91-111; derived fields
 * (constantMu, noiseMu, noiseRange, rcpMajorant) are recomputed by the library
 * as media_desc_update does (path_tracer.c:1963-1987). */
typedef struct PimMediaDesc
{
    float   constantColor[4];
    float   noiseColor[4];
    float   constantMfp;
    float   noiseMfp;
    float   absorption;
    int32_t noiseOctaves;
    float   noiseGain;
    float   noiseLacunarity;
    float   noiseFreq;
    float   noiseScale;
    float   noiseHeight;
    float   phaseDirA;
    float   phaseDirB;
    float   phaseBlend;
} PimMediaDesc;

/* The flattened PtScene (path_tracer.c:122-166) + everything the reference pulls
 * from global singletons at PtScene_Update time: texture table,
 * sky cubemap, g_BrdfLut, cv_pt_dist_meters. */
typedef struct PimSceneDesc
{
    int32_t            vertCount;        /* multiple of 3; primID*3 == iVert */
    int32_t            matCount;
    int32_t            texCount;
    int32_t            skySize;          /* 0 = no sky cubemap */
    const float*       positions;        /* float4[vertCount], world space, w=1 */
    const float*       normals;          /* float4[vertCount] */
    const float*       uvs;              /* float2[vertCount] */
    const int32_t*     matIds;           /* i32[vertCount] */
    const PimMaterial* materials;        /* [matCount] */
    const PimTexture*  textures;         /* [texCount] */
    const float*       skyFaces[6];      /* float3[skySize^2] per Cubeface (cubemap.h:16-26) */
    const float*       brdfLut;          /* float2[brdfLutSize^2] (lighting.h:45-55) */
    int32_t            brdfLutSize;
    int32_t            emissiveAttempts; /* 1000 in the reference (path_tracer.c:853) */
    float              metersPerCell;    /* cv_pt_dist_meters, 1.5 (cvars.c:208-216) */
    uint32_t           seed;             /* build-defined seed of the set-up RNG streams */
    PimMediaDesc       media;
} PimSceneDesc;

typedef struct PimSceneInfo
{
    int32_t vertCount;
    int32_t triCount;
    int32_t matCount;
    int32_t emissiveCount;
    int32_t gridSize[3];
    int32_t gridLen;
    int32_t activeCells;          /* cells with a Dist1D (length != 0) */
    int32_t bvhNodeCount;
    int32_t bvhMaxDepth;
    float   boundsLo[3];
    float   boundsHi[3];
    float   bvhBuildMs;           /* device time of the last BVH build */
    float   sceneSetupMs;         /* emissives + light grid */
} PimSceneInfo;

/* PtRayHit, rendering/path_tracer.h:24-31 (48 bytes) */
typedef struct PimRayHit
{
    float    wuvt[4];
    float    normal[4];
    int32_t  type;                /* PIM_HIT_* */
    int32_t  iVert;
    uint32_t flags;
    int32_t  _pad;
} PimRayHit;

/* PtResult, rendering/path_tracer.h:54-59 (36 bytes) */
typedef struct PimResult
{
    float color[3];
    float albedo[3];
    float normal[3];
} PimResult;

/* Camera, rendering/camera.h:8-15 */
typedef struct PimCamera
{
    float position[4];
    float rotation[4];            /* quaternion xyzw */
    float zNear;
    float zFar;
    float fovy;                   /* degrees */
    float _pad;
} PimCamera;

/* PtDofInfo, rendering/path_tracer.h:33-42 */
typedef struct PimDofInfo
{
    float   aperture;
    float   focalLength;
    int32_t bladeCount;
    float   bladeRot;
    float   focalPlaneCurvature;
    float   autoFocusSpeed;
    uint8_t autoFocus;            /* bool at offset 24, as the reference's; the padding bytes are not read */
    uint8_t _pad[3];
} PimDofInfo;

/* build-side extensions of one Pt_Trace call (D1, D2) */
typedef struct PimTraceParams
{
    float    sampleWeight;        /* PtTrace.sampleWeight = 1/sampleCount, path_tracer.h:51 */
    float    frameDt;             /* Time_Deltaf() fed to DofUpdate, path_tracer.c:2497 */
    uint32_t sampleIndex;         /* index of this sample (frame); seeds the per-pixel stream */
    uint32_t seed;
    int32_t  maxBounces;          /* <=0: reference behaviour (666, RR only) */
    int32_t  updateDists;         /* !=0: run UpdateDists first as PtScene_Update does (:1034) */
} PimTraceParams;

/* one lightmap page: the single allocation of Lightmap_New (lightmap.c:81-110):
 * [5][S*S] float4 probes | [S*S] float3 position | [S*S] float3 normal | [S*S] float sampleCounts */
typedef struct PimLightmapPage
{
    float* block;                 /* host pointer to the page's allocation */
} PimLightmapPage;

typedef struct PimScene PimScene;  /* opaque, like PtScene (path_tracer.h:13) */
typedef struct PimTrace PimTrace;  /* device-side PtTrace (path_tracer.h:44-52) */

/* ---- library ------------------------------------------------------------- */

/* PtSys_Init / PtSys_Shutdown (path_tracer.c:421-446): select the HIP device. */
int pimgpu_init(int device);
/* The same with a device LIST: the reference fans every entry point out over all host threads inside one process
 * (Task_Run over Thread_HardwareCount() threads, threading/task.c:213-223); this fans out over the listed GPUs.  Every scene
 * is replicated on each device; pimgpu_trace_image renders one band of rows per device, pimgpu_lightmap_bake 64x64-texel tiles
 * round-robin; results land in the caller's host buffers (each device copies its own share), the light histograms of the
 * replicas are summed before every UpdateDists.  Sharding does not change a bit of any result.  The other entry points run
 * on the first device.  A device may be listed more than once (two contexts on one GPU).  pimgpu_init(d) == a list of one. */
int pimgpu_init_devices(const int* devices, int deviceCount);
int pimgpu_device_count(void);           /* entries of the list in use, 0 before init */
int pimgpu_shutdown(void);
const char* pimgpu_last_error(void);
/* Host memory for the buffers a caller reads right after a call (PtTrace's images, Tex_Calloc in PtTrace_New,
 * path_tracer.c:1126-1129): zeroed and page-locked, so that the device-to-host copies of pimgpu_trace_download run at
 * the link's rate instead of through a pageable staging copy.  Any host pointer works everywhere; this one is faster. */
void* pimgpu_host_alloc(size_t bytes);
void pimgpu_host_free(void* p);
/* the hipStream_t every call of the library runs on (for HIP events around a call: bench.py's roofline leg) */
void* pimgpu_stream(void);
/* measurement aid (no counterpart in the reference): with enable != 0 every closest-hit launch runs the COUNTING
 * build of the traversal kernel, which adds up the BVH nodes it fetches (64 B each: the compressed nodes), the triangles it tests (48 B
 * each) and the rays it traces.  out (may be NULL) receives {nodes, triangles, rays, most nodes of one ray} since the last call;
 * the counters are then cleared.  These are the per-ray figures of the roofline's algorithmic bytes. */
int pimgpu_work_counters(int enable, uint64_t* out);

/* ---- scene inputs --------------------------------------------------------- */

/* Mesh, rendering/mesh.h:25-33: the three arrays FlattenDrawables reads, float4[length] each, host; uvs.zw are the
 * lightmap coordinates and texIndices[].w the lightmap's texture id (int4[length]; written by chartnodes_assign,
 * lightmap.c:919-930; only pimgpu_lightmap_embed reads them, NULL = no triangle of the mesh is mapped) */
typedef struct PimMesh
{
    int32_t        length;
    int32_t        _pad;
    const float*   positions;
    const float*   normals;
    const float*   uvs;
    const int32_t* texIndices;
} PimMesh;

/* the columns of the Entities table FlattenDrawables reads (rendering/drawable.h:14-27) and the
 * mesh table behind Mesh_Get (rendering/mesh.h:55) */
typedef struct PimDrawables
{
    int32_t            count;
    int32_t            meshCount;
    const int32_t*     meshes;     /* [count] index into meshTable; < 0 = Mesh_Get() returns NULL, entity skipped */
    const float*       matrices;   /* float4x4[count], columns c0 c1 c2 c3 (math/types.h:392-398) */
    const PimMaterial* materials;  /* [count] */
    const PimMesh*     meshTable;  /* [meshCount] */
} PimDrawables;

/* FlattenDrawables (rendering/path_tracer.c:692-782).  flatten_counts is its sizing pass (:701-712);
 * flatten fills caller-owned host arrays of those sizes: float4 positions/normals[vertCount],
 * float2 uvs[vertCount], i32 matIds[vertCount], materials[matCount] -- the first five fields of PimSceneDesc. */
int pimgpu_flatten_counts(const PimDrawables* drawables, int32_t* vertCount, int32_t* matCount);
int pimgpu_flatten(const PimDrawables* drawables, float* positions, float* normals, float* uvs,
    int32_t* matIds, PimMaterial* materials);

/* BakeBrdfFn (math/lighting.c:31-84) as driven by BrdfLut_New (:86-99, prevSamples = 0 on a zeroed table)
 * and BrdfLut_Update (:107-118): adds samples [prevSamples, numSamples) to every texel of the host table
 * float2[width*height] (g_BrdfLut, lighting.h:45-55).  Texel i draws from the stream Pcg4(i, prevSamples, seed, k). */
int pimgpu_brdf_lut_bake(float* texels, int32_t width, int32_t height, int32_t prevSamples, int32_t numSamples, uint32_t seed);

/* ---- scene --------------------------------------------------------------- */

/* PtScene_New -> PtScene_Init (path_tracer.c:1039-1049,1083-1093):
 * upload, SetupEmissives (:845-883), RtcNewScene/rtcCommitScene (:618-690, replaced by the
 * GPU LBVH build), SetupLightGrid (:885-1009). */
int pimgpu_scene_create(const PimSceneDesc* desc, PimScene** out);
/* PtScene_Del (path_tracer.c:1095-1102) */
int pimgpu_scene_destroy(PimScene* scene);
/* rtcCommitScene alone (path_tracer.c:687): full BVH rebuild from the resident positions */
int pimgpu_scene_rebuild_bvh(PimScene* scene);
/* UpdateDists (path_tracer.c:2471-2483): Dist1D_Update on every grid cell */
int pimgpu_scene_update_dists(PimScene* scene);
/* igText lines of PtScene_Gui (path_tracer.c:1104-1115) as data */
int pimgpu_scene_info(const PimScene* scene, PimSceneInfo* out);
/* parity probes of the scene set-up: any pointer may be NULL.
 *   vertToEmit i32[vertCount], emitToVert i32[emissiveCount],
 *   cellLength i32[gridLen] (Dist1D.length, 0 = skipped cell),
 *   pdf float[gridLen*E], cdf float[gridLen*(E+1)], live u32[gridLen*E] */
int pimgpu_scene_download_lights(const PimScene* scene,
    int32_t* vertToEmit, int32_t* emitToVert, int32_t* cellLength,
    float* pdf, float* cdf, uint32_t* live);

/* The light histograms alone -- Dist1D.live of every grid cell, u32[gridLen * emissiveCount], the counters LightOnHit adds to
 * with relaxed atomics (path_tracer.c:1709-1733) and UpdateDists folds into the pdfs (:2460-2483, math/dist1d.c:128-163) --
 * copied device-to-device to / from caller-owned HBM.  The reference has one PtScene per process; with one PROCESS per GPU
 *  every replica collects its own increments, and summing them over the ranks (RCCL all-reduce of live - base)
 * before pimgpu_scene_update_dists keeps every replica's sampling distributions equal to the ones a single process tracing all
 * the paths would hold (pim_amd/dist.py, bench.py --gpus N).  A device LIST inside one process does this itself. */
int pimgpu_scene_live_get_dev(const PimScene* scene, void* dstDev);
int pimgpu_scene_live_set_dev(PimScene* scene, const void* srcDev);

/* ---- rays ---------------------------------------------------------------- */

/* batched Pt_Intersect (path_tracer.c:1466-1474, pt_intersect_local :1421-1464);
 * ro/rd float4[n] (w ignored), tNear/tFar float[n]; host pointers. */
int pimgpu_intersect(const PimScene* scene, const float* ro, const float* rd,
    const float* tNear, const float* tFar, int32_t n, PimRayHit* out);
/* same with device pointers (rays and hits stay in HBM); hits are the raw closest-hit
 * records int4-sized {primID, t, u, v} (primID -1 = miss), the layout the wavefront
 * integrator consumes. */
int pimgpu_intersect_dev(const PimScene* scene, const void* roDev, const void* rdDev,
    const void* tNearDev, const void* tFarDev, int32_t n, void* hitsDev);

/* the traversal kernel alone on rays already packed the way it reads them: rayO = (origin.xyz, tNear),
 * rayD = (direction.xyz, tFar), float4[n] each in HBM; hits as above.  This is the batched rtcIntersect1
 * (path_tracer.c:470-486) the wavefront integrator issues twice per bounce, and the unit of the
 * incoherent-ray benchmark. */
int pimgpu_intersect_packed_dev(const PimScene* scene, const void* rayODev, const void* rayDDev, int32_t n, void* hitsDev);

/* batched Pt_TraceRay (path_tracer.c:2306-2416); stream i is seeded Pcg4(i, sampleIndex, seed, k) */
int pimgpu_trace_rays(PimScene* scene, const float* ro, const float* rd, int32_t n,
    uint32_t sampleIndex, uint32_t seed, int32_t maxBounces, PimResult* out);
/* Pt_RayGen (path_tracer.c:2596-2644): count uniform-sphere paths from origin;
 * colors/directions float4[count], host. */
int pimgpu_raygen(PimScene* scene, const float* origin, int32_t count,
    uint32_t sampleIndex, uint32_t seed, int32_t maxBounces,
    float* colors, float* directions);

/* Cubemap_Bake (rendering/cubemap.c:130-190, BakeFn :138-160): one path per cube-face texel from `origin`
 * through Cubemap_CalcDir(size, face, coord, f2_tent(Xi)) (cubemap.h:190-207), blended into the six host faces
 * (float3[size*size] each, Cubemap.color[face], cubemap.h:28-34) as lerp(old, result.color, weight).
 * Texel i = face*size^2 + y*size + x owns the stream Pcg4(i, sampleIndex, seed, k). */
int pimgpu_cubemap_bake(PimScene* scene, float* const faces[6], int32_t size, const float* origin, float weight,
    uint32_t sampleIndex, uint32_t seed, int32_t maxBounces);
/* Cubemap_Convolve (rendering/cubemap.c:192-303, PrefilterFn :235-262, PrefilterEnvMap :192-223; called right after Cubemap_Bake,
 * render_system.c:239-241): every texel of every mip m of the convolved chain gets the split-sum GGX prefilter of the colour faces
 * at roughness m / 6 (MipToRoughness, cubemap.h:66-69) from `sampleCount` samples taken in order, blended in as
 * lerp(old, new, weight) (Cubemap_BlendMip, cubemap.h:169-188).  faces: the six host colour faces, float3[size*size]
 * (Cubemap.color); convolved: the six host chains, float4[pimgpu_cubemap_elem_count(size)] each (Cubemap.convolved as Cubemap_New
 * sizes it, cubemap.c:103-114: mip m at CalcMipOffset(size, m), sampler.h:42-51), read and written.  Texel i of mip m owns the
 * stream Pcg4(m*6*size^2 + i, sampleIndex, seed, k) (the reference draws from its worker thread's generator).  The 1x1 mip is
 * blended towards zero, as the reference's arithmetic gives it (LinearInvert divides by size - 1, sampler.h:192-195). */
int pimgpu_cubemap_convolve(const float* const faces[6], float* const convolved[6], int32_t size, uint32_t sampleCount, float weight,
    uint32_t sampleIndex, uint32_t seed);
int32_t pimgpu_cubemap_elem_count(int32_t size);
/* AmbCube_Bake (math/ambcube.c:5-32): Pt_RayGen(origin, samples) then AmbCube_Fit (ambcube.h:45-67) over the
 * samples IN ORDER with w = 6/(1+samples)/(1+prevSampleCount).  cube: AmbCube_t, 6 float4 (math/types.h:513-517),
 * host, updated in place; *newSampleCount = prevSampleCount + 1. */
int pimgpu_ambcube_bake(PimScene* scene, float* cube, const float* origin, int32_t samples, int32_t prevSampleCount,
    uint32_t sampleIndex, uint32_t seed, int32_t maxBounces, int32_t* newSampleCount);

/* ---- image --------------------------------------------------------------- */

/* PtTrace_New / PtTrace_Del (path_tracer.c:1117-1139): zeroed device buffers */
int pimgpu_trace_create(int32_t width, int32_t height, PimTrace** out);
int pimgpu_trace_destroy(PimTrace* trace);
int pimgpu_trace_clear(PimTrace* trace);
/* Pt_Trace (path_tracer.c:2557-2585): [UpdateDists] + DofUpdate (dof is updated in place,
 * :2485-2503) + TraceFn over W*H pixels (:2514-2554), one sample per pixel, running mean. */
int pimgpu_trace_image(PimScene* scene, PimTrace* trace, const PimCamera* camera,
    PimDofInfo* dof, const PimTraceParams* params);
/* Pt_Trace for `sampleCount` samples at once, sample indices params->sampleIndex + 0 .. sampleCount-1 -- the caller's loop
 * around Pt_Trace (render_system.c:306-352 calls it once per frame; BASELINE config 3 runs 64 of them), sharded by SAMPLE INDEX
 * over the device list: in round r context i renders sample r*G + i, the whole image, into its own running mean, after the
 * replicas' light histograms have been summed and UpdateDists has run on each ([UpdateDists] once per round); at the end the
 * contexts' means are summed on context 0 with weights n_i / sampleCount (context order, multiply then add in fp32) and the sum
 * is blended into the image as one sample: image = lerp(image, sum, params->sampleWeight).  One context: the same with G = 1.
 * dof: every context advances its own copy over its frames; context 0's comes back.  Statistics: summed over all frames. */
int pimgpu_trace_image_samples(PimScene* scene, PimTrace* trace, const PimCamera* camera,
    PimDofInfo* dof, const PimTraceParams* params, int32_t sampleCount);
/* copy the running-mean buffers to host float3[W*H] arrays (any may be NULL);
 * the reference's callers read PtTrace.color/albedo/normal on the host (render_system.c:306-352) */
int pimgpu_trace_download(const PimTrace* trace, float* color, float* albedo, float* normal);
/* Pt_Trace as the reference's callers use it (path_tracer.h:94, render_system.c:306-352: the host images are read right
 * after the call): pimgpu_trace_image and pimgpu_trace_download in one call (any image pointer may be NULL).  The pixels are
 * traced in shares, one after the other, and a share's part of the images travels to the host while the next share is
 * traced; the result is the same bit for bit.  Page-locked images (pimgpu_host_alloc) copy at the link's rate.  Two shares by
 * default; the first one's size follows the tracing and copy times measured on this PimTrace's previous call (INTEGRATION.md,
 * PIMGPU_HOST_FIRST). */
int pimgpu_trace_image_host(PimScene* scene, PimTrace* trace, const PimCamera* camera, PimDofInfo* dof, const PimTraceParams* params,
    float* color, float* albedo, float* normal);
/* device addresses of the float3 buffers (for torch.distributed reduce without a host copy) */
int pimgpu_trace_buffers(const PimTrace* trace, void** colorDev, void** albedoDev, void** normalDev);
/* the same buffers copied device-to-device into caller-owned HBM (any pointer may be NULL): what a multi-GPU
 * caller hands to an RCCL reduce */
int pimgpu_trace_copy_dev(const PimTrace* trace, void* colorDst, void* albedoDst, void* normalDst);
/* CalcStdDev (render_system.c:1374-1394) on the device colour buffer: the reference's two serial fp32 sums in pixel order, bit for
 * bit (formed in parallel where the terms allow it, by the one-wave chain otherwise; PIMGPU_STDDEV_SERIAL=1 forces the chain) */
int pimgpu_trace_stddev(const PimTrace* trace, float* meanOut, float* stddevOut);
/* NO COUNTERPART IN THE REFERENCE.  pimgpu_trace_stddev for callers that cannot allocate (tight serving loops, harnesses that
 * forbid hipMalloc after set-up): same result bit for bit, but the per-pixel terms live in a file-scope device array of the
 * library (31.9 MiB per GPU, held until the process ends) instead of a work buffer grown on demand, and the two floats come
 * back through a page-locked block made at pimgpu_init.  The call is synchronous (a device-to-host copy and a stream wait), so
 * it cannot run under HIP graph capture.  Limit: width*height <= 3840*2160, else PIMGPU_E_LIMIT.  The array belongs to the GPU:
 * all contexts on one GPU (a device listed twice in pimgpu_init_devices) and all concurrent callers share it, so calls that
 * overlap on one GPU race. */
int pimgpu_trace_stddev_noalloc(const PimTrace* trace, float* meanOut, float* stddevOut);
/* statistics of the last pimgpu_trace_image call: rays traced (extension, shadow, other), bounces */
typedef struct PimTraceStats
{
    uint64_t extensionRays;
    uint64_t shadowRays;
    uint64_t otherRays;         /* media NEE + refraction thickness + autofocus */
    uint64_t paths;
    int32_t  bounces;           /* wavefront bounce iterations launched (two traversal launches each) before the tail kernel */
    float    traceMs;           /* device time inside the traversal kernels.  The per-kernel times of this struct come from events attached
                                   to the dispatches; a call times its first 256 dispatches (36 wavefront bounces) and launches the rest
                                   plain, so for deeper roulette-only calls without the tail kernel they are lower bounds */
    float    totalMs;           /* device time of the whole call */
    uint64_t tailRays;          /* of the rays above: traced inside the tail kernel, which finishes the short queues of deep
                                   bounces in one launch (not by the batched traversal launches traceMs covers) */
    /* device time of the other kernels of the call, from events attached to their dispatches (measurement: bench.py's
     * per-kernel rooflines), and the units they processed */
    float    shadeAMs;          /* k_shade_a, FAST + SLOW builds */
    float    shadeBMs;          /* k_shade_b, FAST + SLOW builds */
    float    tailMs;            /* k_tail */
    int32_t  traceLaunches;     /* batched traversal launches traceMs covers */
    uint64_t wavefrontPathBounces;  /* sum of the queue lengths at the start of every wavefront bounce: paths through shade_a / shade_b */
    uint64_t tailPaths;         /* paths handed to the tail kernel */
    uint64_t tailPathBounces;   /* bounces the tail kernel ran (one extension ray each) */
} PimTraceStats;
int pimgpu_trace_stats(const PimTrace* trace, PimTraceStats* out);

/* ---- lightmap ------------------------------------------------------------ */

/* LmPack_Bake (lightmap.c:1204-1223, BakeFn :1125-1201) on host pages of the exact
 * reference layout; pages are updated in place. texelBegin/texelEnd select a
 * sub-range of the lmCount*lmSize^2 work items (texel-tile sharding). */
int pimgpu_lightmap_bake(PimScene* scene, PimLightmapPage* pages, int32_t lmCount,
    int32_t lmSize, float texelsPerMeter, float timeSlice, int32_t spp,
    uint32_t passIndex, uint32_t seed, int32_t maxBounces,
    int64_t texelBegin, int64_t texelEnd);

/* The same bake on pages that STAY in HBM between calls.  The reference bakes in place in host memory, spp samples per
 * frame (render_system.c:196-198), and reads the pages only when it uploads them to the renderer (every 10 s, :200-210)
 * or saves them (lightmap.c:1225): a caller in that position creates a set, uploads its pages once (after
 * EmbedAttributes or LmPack_Load), bakes as often as it likes with no PCIe traffic at all, and downloads probes |
 * sampleCounts -- the parts BakeFn writes (lightmap.c:1194-1200) -- when it needs them.  Same kernels, same bits as
 * pimgpu_lightmap_bake.  With a device list every context keeps a replica and bakes its 64x64 tiles of it. */
typedef struct PimLightmapSet PimLightmapSet;
int pimgpu_lightmap_set_create(int32_t lmCount, int32_t lmSize, PimLightmapSet** out);      /* zeroed pages */
int pimgpu_lightmap_set_destroy(PimLightmapSet* set);
int pimgpu_lightmap_set_upload(PimLightmapSet* set, const PimLightmapPage* pages);          /* whole pages, host -> HBM */
int pimgpu_lightmap_set_download(const PimLightmapSet* set, PimLightmapPage* pages);        /* probes | sampleCounts, HBM -> host */
int pimgpu_lightmap_set_bake(PimScene* scene, PimLightmapSet* set, float texelsPerMeter, float timeSlice, int32_t spp,
    uint32_t passIndex, uint32_t seed, int32_t maxBounces, int64_t texelBegin, int64_t texelEnd);

/* chartnode_t (lightmap.c:48-55), field for field at the same offsets: 64 bytes (the reference's float4 is 16-aligned) */
typedef struct PimChartNode
{
    float   plane[4];           /* triToPlane: unit normal, distance */
    float   triCoord[6];        /* the triangle on its own plane's tangent frame, in texels: a.xy, b.xy, c.xy */
    float   area;               /* TriArea3D, world units */
    int32_t drawableIndex;
    int32_t vertIndex;
    int32_t _pad[3];
} PimChartNode;

/* The data-parallel steps of LmPack_Pack (lightmap.c:1047-1102) around its sequential chart grouping and atlas search,
 * which stay on the host:
 *  chartnodes  = chartnodes_create (:723-767): one node per triangle of every mapped drawable (not MatFlag_Sky | MatFlag_Lava,
 *                mesh present), world space through the entity's matrix.  nodes == NULL: only counts.
 *  chart_layout = the first half of ChartMaskFn (:527-552): chart c owns nodes [chartFirst[c], chartFirst[c+1]); its
 *                triangles move to the origin with a 2-texel margin (nodes updated in place), areas[c] = bounds area,
 *                maskSizes[2c..] = mask width, height.
 *  chart_masks = mask_tri over every node (:244-259, :553-557): texel = 0xff where sdTriangle2D <= kMaskPadding (1 texel)
 *                of any of the chart's triangles, else 0; chart c's mask is maskSizes[2c] x maskSizes[2c+1] bytes at
 *                masks + maskOffsets[c]. */
int pimgpu_lightmap_chartnodes(const PimDrawables* drawables, float texelsPerUnit, PimChartNode* nodes, int32_t capacity, int32_t* nodeCount);
int pimgpu_lightmap_chart_layout(PimChartNode* nodes, const int32_t* chartFirst, int32_t chartCount, float* areas, int32_t* maskSizes);
int pimgpu_lightmap_chart_masks(const PimChartNode* nodes, const int32_t* chartFirst, int32_t chartCount, const int32_t* maskSizes,
    const int64_t* maskOffsets, uint8_t* masks);

/* atlases_create (lightmap.c:845-874) as ONE worker of AtlasFn (:779-823) runs it: the charts, taken in `order` (chart_sort's
 * result, :648-660), are placed one after the other at the first translation -- row-major from the worker's row hint, atlas by
 * atlas from its atlas hint (atlas_search :680-721, mask_find :261-282) -- where their mask overlaps nothing already placed
 * (mask_fits :174-207), and written there (mask_write :209-230).  With several workers the reference's result depends on
 * thread timing; this is the result of one.  masks / maskOffsets / maskSizes / areas as pimgpu_lightmap_chart_layout and
 * _chart_masks leave them, atlasCount from atlas_estimate (:825-843).  Out, by chart: translations[2c], [2c+1] and
 * atlasIndex[c]; *usedAtlases = atlases that took a chart (:862-870) = the pages LmPack_Pack makes.
 * A chart that fits in none of the atlasCount atlases after the worker's full retry makes the reference's AtlasFn spin for ever
 * (:808-818; atlas_estimate counts bounds without margins and does run short): here one more atlas is opened for it, up to
 * atlasCap in all (>= atlasCount; everything placed before that point is where the reference put it).  PIMGPU_E_LIMIT when a
 * chart is larger than an atlas or atlasCap is reached. */
int pimgpu_lightmap_atlas_place(const uint8_t* masks, const int64_t* maskOffsets, const int32_t* maskSizes, const float* areas,
    const int32_t* order, int32_t chartCount, int32_t atlasSize, int32_t atlasCount, int32_t atlasCap,
    int32_t* translations, int32_t* atlasIndex, int32_t* usedAtlases);

/* EmbedAttributes / EmbedTaskFn (lightmap.c:941-1045), the last step of LmPack_Pack (:1092) and the producer of
 * LmPack_Bake's inputs: every texel of every page takes the mapped triangle (texIndices[iVert].w == the page's texture
 * id; drawables with MatFlag_Sky | MatFlag_Lava and NULL meshes skipped) whose signed 2D distance (sdTriangle2D on
 * uvs.zw * lmSize) to the texel centre is smallest and below kFillPadding = 2 texels -- the first in drawable, vertex
 * order on equal distance -- and keeps its barycentric position and normal (MESH space, as the reference);
 * sampleCounts = 1 where a triangle was found, else 0, position = normal = 0.  pageTexIds[lmCount] are the pages'
 * texture ids (Lightmap.slot.index, lightmap.h:30); NULL = page i has id i.  Writes the position | normal |
 * sampleCounts parts of the host pages; the probes are left alone. */
int pimgpu_lightmap_embed(const PimDrawables* drawables, const int32_t* pageTexIds, PimLightmapPage* pages,
    int32_t lmCount, int32_t lmSize);

#ifdef __cplusplus
}
#endif

#endif /* PIMGPU_H */
