/*
 * pim_host.c -- Pim's path-tracer entry points (rendering/path_tracer.h:67-103, lightmap.h:56-69, cubemap.h:209,
 * ambcube.h:69) with their own signatures, as thin C wrappers over libpimgpu.so.  No arithmetic of the path happens
 * here: every function marshals its arguments, makes C-ABI calls and hands the result back in the reference's host-side
 * shape.  State the reference keeps in globals / per-thread contexts (the scene source, the Prng streams, the lightmap
 * pack, the frame arena of Pt_RayGen) is kept here in file-scope variables, as path_tracer.c keeps ms_contexts and
 * lightmap.c keeps ms_pack.
 */
#include "pim_host.h"

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

_Static_assert(sizeof(float4) == 16 && _Alignof(float4) == 16, "float4 (math/types.h:66-73)");
_Static_assert(sizeof(PtRayHit) == 48 && sizeof(PimRayHit) == 48, "PtRayHit (path_tracer.h:24-31)");
_Static_assert(sizeof(PtDofInfo) == 28 && sizeof(PimDofInfo) == 28 && offsetof(PtDofInfo, autoFocus) == 24 && offsetof(PimDofInfo, autoFocus) == 24, "PtDofInfo (path_tracer.h:33-42)");
_Static_assert(sizeof(PtResult) == 36 && sizeof(PimResult) == 36, "PtResult (path_tracer.h:54-59)");
_Static_assert(sizeof(Camera) == 48 && sizeof(PimCamera) == 48, "Camera (rendering/camera.h:8-15)");
_Static_assert(sizeof(PtTrace) == 48, "PtTrace (path_tracer.h:44-52)");
_Static_assert(sizeof(vkrTextureId) == 4, "vkrTextureId (vkr.h:125-131)");

struct PtScene_s
{
    PimScene* gpu;
    uint64_t modtime;         /* PtScene.modtime, path_tracer.c:164 */
    bool fromProvider;
};

static char ms_error[512];
static bool ms_ok;
static PtSysConfig ms_config = { 0x9E3779B9u, 0, 1.0f / 60.0f, 1, -1 };
static PtSysCounters ms_counters;
static PtSceneProviderFn ms_provider;
static void* ms_providerUser;
static LmPack ms_pack;                       /* lightmap.c:79 */

/* PtTrace is the caller's struct and has no room for the device handle: a side table keyed by the color pointer */
#define kMaxTraces 64
static struct { float3* key; PimTrace* gpu; } ms_traces[kMaxTraces];
/* Pt_RayGen's results live in the reference's Temp_ frame arena (path_tracer.c:2630-2631): freed by PtSys_Update */
#define kMaxFrameAllocs 256
static void* ms_frameAllocs[kMaxFrameAllocs];
static int ms_frameAllocCount;

static bool ok(int rc, const char* what)
{
    if (rc == PIMGPU_OK) return true;
    snprintf(ms_error, sizeof(ms_error), "%s: %s", what, pimgpu_last_error());
    fprintf(stderr, "[pt] %s\n", ms_error);          /* Con_Logf(LogSev_Error, "pt", ...) in a Pim build */
    return false;
}

static void fail(const char* what)
{
    snprintf(ms_error, sizeof(ms_error), "%s", what);
    fprintf(stderr, "[pt] %s\n", ms_error);
}

const char* PtSys_LastError(void) { return ms_error; }
bool PtSys_Ok(void) { return ms_ok; }

void PtSys_Configure(const PtSysConfig* config)
{
    if (config) ms_config = *config;
    if (!(ms_config.frameDt > 0.0f)) ms_config.frameDt = 1.0f / 60.0f;
    memset(&ms_counters, 0, sizeof(ms_counters));
}

void PtSys_GetCounters(PtSysCounters* out) { if (out) *out = ms_counters; }
uint32_t PtSys_Seed(void) { return ms_config.seed; }
void PtSys_SetSceneProvider(PtSceneProviderFn fn, void* user) { ms_provider = fn; ms_providerUser = user; }

void PtSys_Init(void)
{
    /* the reference fans out over every hardware thread of the machine (task.c:213-223); PIM_GPU_DEVICES = "0,1,2,..." fans
     * out over those GPUs (pimgpu_init_devices), PIM_GPU_DEVICE / PtSysConfig.device pick one */
    const char* list = ms_config.device < 0 ? getenv("PIM_GPU_DEVICES") : NULL;
    if (list && *list)
    {
        int devices[16], n = 0;
        for (const char* p = list; *p && n < 16; )
        {
            devices[n++] = atoi(p);
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
        ms_ok = ok(pimgpu_init_devices(devices, n), "PtSys_Init");
        return;
    }
    int device = ms_config.device;
    if (device < 0) { const char* dev = getenv("PIM_GPU_DEVICE"); device = dev ? atoi(dev) : 0; }
    ms_ok = ok(pimgpu_init(device), "PtSys_Init");
}

void PtSys_Update(void)
{
    for (int i = 0; i < ms_frameAllocCount; ++i) free(ms_frameAllocs[i]);
    ms_frameAllocCount = 0;
}

void PtSys_Shutdown(void)
{
    PtSys_Update();
    pimgpu_shutdown();
    ms_ok = false;
}

/* ---- scene ---- */

PimScene* PtScene_Gpu(const PtScene* scene) { return scene ? scene->gpu : NULL; }

PtScene* PtScene_NewFromSnapshot(const PimSceneDesc* snapshot)
{
    PimScene* gpu = NULL;
    if (!ok(pimgpu_scene_create(snapshot, &gpu), "PtScene_New")) return NULL;
    PtScene* scene = calloc(1, sizeof(*scene));              /* Perm_Calloc, path_tracer.c:1087 */
    if (!scene) { pimgpu_scene_destroy(gpu); return NULL; }
    scene->gpu = gpu;
    return scene;
}

PtScene* PtScene_New(void)
{
    if (!ms_provider) { fail("PtScene_New: no scene provider (PtSys_SetSceneProvider)"); return NULL; }
    PimSceneDesc desc; memset(&desc, 0, sizeof(desc));
    uint64_t modtime = 0;
    if (!ms_provider(&desc, &modtime, 0, ms_providerUser)) { fail("PtScene_New: the scene provider failed"); return NULL; }
    PtScene* scene = PtScene_NewFromSnapshot(&desc);
    if (scene) { scene->modtime = modtime; scene->fromProvider = true; }
    return scene;
}

/* the first half of PtScene_Update (path_tracer.c:1028-1032): Entities_Get()->modtime != scene->modtime -> PtScene_Clear +
 * PtScene_Init, in place.  Returns true when the scene was rebuilt. */
static bool scene_refresh(PtScene* scene)
{
    if (!scene->fromProvider || !ms_provider) return false;
    uint64_t modtime = scene->modtime;
    PimSceneDesc desc; memset(&desc, 0, sizeof(desc));
    if (!ms_provider(&desc, &modtime, 1, ms_providerUser) || modtime == scene->modtime) return false;
    memset(&desc, 0, sizeof(desc));
    if (!ms_provider(&desc, &modtime, 0, ms_providerUser)) { fail("PtScene_Update: the scene provider failed"); return false; }
    PimScene* gpu = NULL;
    if (!ok(pimgpu_scene_create(&desc, &gpu), "PtScene_Update")) return false;
    pimgpu_scene_destroy(scene->gpu);
    scene->gpu = gpu; scene->modtime = modtime;
    return true;
}

void PtScene_Update(PtScene* scene)
{
    if (!scene) return;
    scene_refresh(scene);
    ok(pimgpu_scene_update_dists(scene->gpu), "PtScene_Update");      /* UpdateDists, :1034 */
}

void PtScene_Del(PtScene* scene)
{
    if (!scene) return;
    pimgpu_scene_destroy(scene->gpu);
    free(scene);
}

void PtScene_Gui(PtScene* scene)
{
    if (!scene) return;
    PimSceneInfo info;
    if (!ok(pimgpu_scene_info(scene->gpu, &info), "PtScene_Gui")) return;
    /* path_tracer.c:1104-1115 */
    printf("Vertex Count: %d\nMaterial Count: %d\nEmissive Count: %d\n", info.vertCount, info.matCount, info.emissiveCount);
    printf("Light Grid: %d x %d x %d, %d of %d cells lit; BVH: %d nodes, %d levels, built in %.3f ms\n",
        info.gridSize[0], info.gridSize[1], info.gridSize[2], info.activeCells, info.gridLen, info.bvhNodeCount, info.bvhMaxDepth, info.bvhBuildMs);
}

PtScene* PtScene_NewFromDrawables(const PimDrawables* drawables, const PimSceneDesc* rest)
{
    if (!rest) { fail("PtScene_NewFromDrawables: NULL snapshot"); return NULL; }
    int32_t vertCount = 0, matCount = 0;
    if (!ok(pimgpu_flatten_counts(drawables, &vertCount, &matCount), "FlattenDrawables")) return NULL;
    /* Perm_Calloc in the reference (path_tracer.c:715-721) */
    float* positions = calloc((size_t)vertCount + 1, 4 * sizeof(float));
    float* normals = calloc((size_t)vertCount + 1, 4 * sizeof(float));
    float* uvs = calloc((size_t)vertCount + 1, 2 * sizeof(float));
    int32_t* matIds = calloc((size_t)vertCount + 1, sizeof(int32_t));
    PimMaterial* materials = calloc((size_t)matCount + 1, sizeof(PimMaterial));
    PtScene* scene = NULL;
    if (positions && normals && uvs && matIds && materials &&
        ok(pimgpu_flatten(drawables, positions, normals, uvs, matIds, materials), "FlattenDrawables"))
    {
        PimSceneDesc desc = *rest;
        desc.vertCount = vertCount; desc.matCount = matCount;
        desc.positions = positions; desc.normals = normals; desc.uvs = uvs; desc.matIds = matIds; desc.materials = materials;
        scene = PtScene_NewFromSnapshot(&desc);
    }
    free(positions); free(normals); free(uvs); free(matIds); free(materials);      /* the scene keeps its own copies in HBM */
    return scene;
}

/* ---- BRDF table ---- */

BrdfLut BrdfLut_New(int2 size, i32 numSamples)
{
    BrdfLut lut; lut.size = size; lut.texels = NULL;
    if (size.x <= 0 || size.y <= 0) { lut.size.x = lut.size.y = 0; return lut; }
    lut.texels = calloc((size_t)size.x * (size_t)size.y, sizeof(float2));           /* Tex_Calloc, lighting.c:92 */
    if (lut.texels) BrdfLut_Update(&lut, 0, numSamples);
    return lut;
}

void BrdfLut_Update(BrdfLut* lut, i32 prevSamples, i32 newSamples)
{
    if (lut && lut->texels && (newSamples > prevSamples))                           /* lighting.c:110 */
        ok(pimgpu_brdf_lut_bake(&lut->texels[0].x, lut->size.x, lut->size.y, prevSamples, newSamples, ms_config.seed), "BrdfLut_Update");
}

void BrdfLut_Del(BrdfLut* lut) { if (lut) { free(lut->texels); memset(lut, 0, sizeof(*lut)); } }

/* ---- images ---- */

static int trace_find(const float3* key)
{
    if (!key) return -1;
    for (int i = 0; i < kMaxTraces; ++i) if (ms_traces[i].key == key) return i;
    return -1;
}

void PtTrace_New(PtTrace* trace, int2 imageSize)
{
    if (!trace) return;
    memset(trace, 0, sizeof(*trace));
    const size_t texelCount = (size_t)imageSize.x * (size_t)imageSize.y;
    if (!texelCount) return;
    int slot = -1;
    for (int i = 0; i < kMaxTraces && slot < 0; ++i) if (!ms_traces[i].key) slot = i;
    if (slot < 0) { fail("PtTrace_New: too many live PtTrace objects"); return; }
    PimTrace* gpu = NULL;
    if (!ok(pimgpu_trace_create(imageSize.x, imageSize.y, &gpu), "PtTrace_New")) return;
    trace->imageSize = imageSize;
    /* Tex_Calloc in the reference; page-locked here: Pt_Trace copies the images back on every call */
    trace->color = pimgpu_host_alloc(texelCount * sizeof(float3));
    trace->albedo = pimgpu_host_alloc(texelCount * sizeof(float3));
    trace->normal = pimgpu_host_alloc(texelCount * sizeof(float3));
    trace->denoised = pimgpu_host_alloc(texelCount * sizeof(float3));
    trace->sampleWeight = 1.0f;
    if (!trace->color || !trace->albedo || !trace->normal || !trace->denoised)
    {
        pimgpu_trace_destroy(gpu);
        pimgpu_host_free(trace->color); pimgpu_host_free(trace->albedo); pimgpu_host_free(trace->normal); pimgpu_host_free(trace->denoised);
        memset(trace, 0, sizeof(*trace));
        fail("PtTrace_New: out of page-locked host memory");
        return;
    }
    ms_traces[slot].key = trace->color; ms_traces[slot].gpu = gpu;
}

void PtTrace_Del(PtTrace* trace)
{
    if (!trace) return;
    const int slot = trace_find(trace->color);
    if (slot >= 0) { pimgpu_trace_destroy(ms_traces[slot].gpu); ms_traces[slot].key = NULL; ms_traces[slot].gpu = NULL; }
    pimgpu_host_free(trace->color); pimgpu_host_free(trace->albedo); pimgpu_host_free(trace->normal); pimgpu_host_free(trace->denoised);
    memset(trace, 0, sizeof(*trace));
}

void DofInfo_New(PtDofInfo* dof)
{
    if (!dof) return;
    dof->aperture = 5.0f * 1e-3f;
    dof->focalLength = 6.0f;
    dof->bladeCount = 5;
    dof->bladeRot = 3.1415926535897932384626433832795f / 10.0f;
    dof->focalPlaneCurvature = 0.05f;
    dof->autoFocus = true;
    dof->autoFocusSpeed = 3.0f;
}

void DofInfo_Gui(PtDofInfo* dof)
{
    if (!dof) return;
    /* path_tracer.c:1155-1178: the values its sliders show */
    printf("Aperture: %g mm\nFocal Length: %g m\nBlade Count: %d\nBlade Rotation: %g\nFocal Plane Curvature: %g\nAutofocus: %s (speed %g)\n",
        dof->aperture * 1e3f, dof->focalLength, dof->bladeCount, dof->bladeRot, dof->focalPlaneCurvature, dof->autoFocus ? "on" : "off", dof->autoFocusSpeed);
}

static PimDofInfo dof_to_gpu(const PtDofInfo* d)
{
    PimDofInfo g; memset(&g, 0, sizeof(g));
    g.aperture = d->aperture; g.focalLength = d->focalLength; g.bladeCount = d->bladeCount; g.bladeRot = d->bladeRot;
    g.focalPlaneCurvature = d->focalPlaneCurvature; g.autoFocusSpeed = d->autoFocusSpeed; g.autoFocus = d->autoFocus ? 1 : 0;
    return g;
}

PtRayHit Pt_Intersect(const PtScene* scene, float4 ro, float4 rd, float tNear, float tFar)
{
    PimRayHit h; memset(&h, 0, sizeof(h)); h.wuvt[3] = -1.0f; h.iVert = -1;
    if (scene) ok(pimgpu_intersect(scene->gpu, &ro.x, &rd.x, &tNear, &tFar, 1, &h), "Pt_Intersect");
    PtRayHit out;
    memcpy(&out.wuvt, h.wuvt, sizeof(float4)); memcpy(&out.normal, h.normal, sizeof(float4));
    out.type = (PtHitType)h.type; out.iVert = h.iVert; out.flags = h.flags;
    return out;
}

PtResult Pt_TraceRayEx(PtScene* scene, float4 ro, float4 rd, uint32_t sampleIndex, uint32_t seed, i32 maxBounces)
{
    PimResult r; memset(&r, 0, sizeof(r));
    if (scene) ok(pimgpu_trace_rays(scene->gpu, &ro.x, &rd.x, 1, sampleIndex, seed, maxBounces, &r), "Pt_TraceRay");
    PtResult out; memcpy(&out, &r, sizeof(out));
    return out;
}

PtResult Pt_TraceRay(PtScene* scene, float4 ro, float4 rd)
{
    return Pt_TraceRayEx(scene, ro, rd, ms_counters.traceRay++, ms_config.seed, ms_config.maxBounces);
}

void Pt_Trace(PtTrace* trace, PtDofInfo* dof, PtScene* scene, const Camera* camera)
{
    if (!trace || !dof || !scene || !camera) return;
    const int slot = trace_find(trace->color);
    if (slot < 0) { fail("Pt_Trace: this PtTrace was not made by PtTrace_New"); return; }
    scene_refresh(scene);                                /* PtScene_Update, path_tracer.c:2565: rebuild on a new modtime ... */
    PimTraceParams p;
    p.sampleWeight = trace->sampleWeight;
    p.frameDt = ms_config.frameDt;
    p.sampleIndex = ms_counters.trace;
    p.seed = ms_config.seed;
    p.maxBounces = ms_config.maxBounces;
    p.updateDists = 1;                                   /* ... and UpdateDists (:1034), enqueued with the frame */
    PimDofInfo gdof = dof_to_gpu(dof);
    PimCamera cam; memcpy(&cam, camera, sizeof(cam));
    /* the reference's callers read the host images right after Pt_Trace (render_system.c:306-352): the copy of one share of
     * the pixels runs under the tracing of the next */
    const bool aovs = ms_config.downloadAovs != 0;
    if (!ok(pimgpu_trace_image_host(scene->gpu, ms_traces[slot].gpu, &cam, &gdof, &p, &trace->color->x, aovs ? &trace->albedo->x : NULL, aovs ? &trace->normal->x : NULL), "Pt_Trace")) return;
    dof->focalLength = gdof.focalLength;                 /* DofUpdate, path_tracer.c:2485-2503 */
    ms_counters.trace++;
}

void PtTrace_DownloadAovs(PtTrace* trace)
{
    if (!trace) return;
    const int slot = trace_find(trace->color);
    if (slot >= 0) ok(pimgpu_trace_download(ms_traces[slot].gpu, NULL, &trace->albedo->x, &trace->normal->x), "PtTrace_DownloadAovs");
}

PtResults Pt_RayGen(PtScene* scene, float4 origin, i32 count)
{
    PtResults r = { NULL, NULL };
    if (!scene || count <= 0) return r;
    if (ms_frameAllocCount + 2 > kMaxFrameAllocs) PtSys_Update();       /* a caller that never ends its frames */
    r.colors = calloc((size_t)count, sizeof(float4));                   /* Temp_Alloc, path_tracer.c:2630-2631 */
    r.directions = calloc((size_t)count, sizeof(float4));
    if (!r.colors || !r.directions ||
        !ok(pimgpu_raygen(scene->gpu, &origin.x, count, ms_counters.rayGen, ms_config.seed, ms_config.maxBounces, &r.colors->x, &r.directions->x), "Pt_RayGen"))
    {
        free(r.colors); free(r.directions);
        r.colors = NULL; r.directions = NULL;
        return r;
    }
    ms_counters.rayGen++;
    ms_frameAllocs[ms_frameAllocCount++] = r.colors;
    ms_frameAllocs[ms_frameAllocCount++] = r.directions;
    return r;
}

void Cubemap_Bake(Cubemap* cm, PtScene* scene, float4 origin, float weight)
{
    if (!cm || !scene || cm->size <= 0) return;                      /* cubemap.c:168-173 */
    float* faces[6];
    for (int f = 0; f < 6; ++f) faces[f] = &cm->color[f]->x;
    if (ok(pimgpu_cubemap_bake(scene->gpu, faces, cm->size, &origin.x, weight, ms_counters.cubemap, ms_config.seed, ms_config.maxBounces), "Cubemap_Bake"))
        ms_counters.cubemap++;
}

/* Cubemap_New / Cubemap_Del, cubemap.c:96-128: size^2 colour texels and the whole mip chain of float4 per face, zeroed */
void Cubemap_New(Cubemap* cm, i32 size)
{
    if (!cm) return;
    memset(cm, 0, sizeof(*cm));
    if (size < 0) return;
    const i32 elems = pimgpu_cubemap_elem_count(size);
    cm->size = size;
    for (cm->mipCount = 0; (size >> cm->mipCount) > 0; ++cm->mipCount) { }            /* 1 + floor(log2 size), sampler.h:23-26 */
    for (int f = 0; f < 6; ++f)
    {
        cm->color[f] = calloc((size_t)size * (size_t)size + 1, sizeof(float3));
        cm->convolved[f] = calloc((size_t)elems + 1, sizeof(float4));
    }
}

void Cubemap_Del(Cubemap* cm)
{
    if (!cm) return;
    for (int f = 0; f < 6; ++f) { free(cm->color[f]); free(cm->convolved[f]); }
    memset(cm, 0, sizeof(*cm));
}

/* Cubemap_Convolve, cubemap.c:264-303: call k of the process is sample index k of the convolution streams */
void Cubemap_Convolve(Cubemap* cm, u32 sampleCount, float weight)
{
    if (!cm || cm->size <= 0) return;
    const float* faces[6]; float* conv[6];
    for (int f = 0; f < 6; ++f) { faces[f] = &cm->color[f]->x; conv[f] = &cm->convolved[f]->x; }
    if (ok(pimgpu_cubemap_convolve(faces, conv, cm->size, sampleCount, weight, ms_counters.convolve, ms_config.seed), "Cubemap_Convolve"))
        ms_counters.convolve++;
}

i32 AmbCube_Bake(PtScene* scene, AmbCube_t* pCube, float4 origin, i32 samples, i32 prevSampleCount)
{
    i32 count = prevSampleCount;
    if (!scene || !pCube) return count;
    if (ok(pimgpu_ambcube_bake(scene->gpu, &pCube->Values[0].x, &origin.x, samples, prevSampleCount, ms_counters.rayGen, ms_config.seed, ms_config.maxBounces, &count), "AmbCube_Bake"))
        ms_counters.rayGen++;                                        /* AmbCube_Bake is a Pt_RayGen call (ambcube.c:17) */
    return count;
}

/* ---- lightmaps ---- */

void Lightmap_New(Lightmap* lm, i32 size) { Lightmap_NewIn(lm, size, NULL); }

/* block: what Lightmap_New would allocate, made ahead by the caller (pimgpu_host_alloc of size * size * 27 floats, still zeroed:
 * LmPack_Pack page-locks its pages on a second thread while the atlas search runs); NULL: allocated here.  The page owns it. */
void Lightmap_NewIn(Lightmap* lm, i32 size, float* block)
{
    memset(lm, 0, sizeof(*lm));
    if (size <= 0) { pimgpu_host_free(block); return; }
    lm->size = size;
    const size_t texelcount = (size_t)size * (size_t)size;
    /* [5] float4 | float3 | float3 | float, lightmap.c:90-110; Tex_Calloc there, page-locked here: LmPack_Bake copies a
     * page (113 MB at 1024^2) to the device and back on every call */
    float* allocation = block ? block : pimgpu_host_alloc(texelcount * 27 * sizeof(float));
    if (!allocation) { lm->size = 0; return; }
    for (i32 i = 0; i < kGiDirections; ++i) lm->probes[i] = (float4*)(allocation + texelcount * 4 * (size_t)i);
    lm->position = (float3*)(allocation + texelcount * 20);
    lm->normal = (float3*)(allocation + texelcount * 23);
    lm->sampleCounts = allocation + texelcount * 26;
    /* vkrTexTable_Alloc (lightmap.c:112-120) gives every page a texture slot of its own; chartnodes_assign stamps slot.index into
     * the meshes and EmbedAttributes finds the page's triangles by it (:919-921, :962).  The renderer is not here: ids count up. */
    static u32 nextSlot = 1;
    lm->slot.version = 1; lm->slot.type = 0; lm->slot.index = nextSlot++ & 0x1fffffu;
    Lightmap_Upload(lm);                                                 /* lightmap.c:122 */
}

void Lightmap_Del(Lightmap* lm)
{
    if (lm) { pimgpu_host_free(lm->probes[0]); memset(lm, 0, sizeof(*lm)); }
}

LmPack* LmPack_Get(void) { return &ms_pack; }

/* ---- the pack's pages kept in HBM between LmPack_Bake calls (LmPack_SetResident) ---- */
static i32 ms_lmMode;                        /* 0: every call copies the pages in and out (the reference's contract, default) */
static PimLightmapSet* ms_lmSet;
static i32 ms_lmSetCount, ms_lmSetSize;
static const void* ms_lmSetKey;              /* probes[0] of page 0 when the set was filled */
static bool ms_lmDeviceNewer;                /* mode 2: the device copy holds bake results the host pages do not have yet */

static void lm_set_drop(void)
{
    if (ms_lmSet) pimgpu_lightmap_set_destroy(ms_lmSet);
    ms_lmSet = NULL; ms_lmSetCount = ms_lmSetSize = 0; ms_lmSetKey = NULL; ms_lmDeviceNewer = false;
}

static PimLightmapPage* lm_page_table(const LmPack* pack)
{
    PimLightmapPage* pages = calloc((size_t)pack->lmCount, sizeof(PimLightmapPage));
    if (pages) for (i32 i = 0; i < pack->lmCount; ++i) pages[i].block = &pack->lightmaps[i].probes[0]->x;   /* lightmap.c:95-110 */
    return pages;
}

void LmPack_Sync(void)
{
    /* only into the pack the set was filled from: same page count, page size and first page (a caller may have replaced
     * *LmPack_Get() without LmPack_Del; downloading the old set's lmSize^2 * 20 floats into other pages would write stale
     * results and overrun smaller ones) */
    if (!ms_lmSet || !ms_lmDeviceNewer || !ms_pack.lightmaps || ms_pack.lmCount != ms_lmSetCount || ms_pack.lmSize != ms_lmSetSize ||
        ms_lmSetKey != (const void*)ms_pack.lightmaps[0].probes[0]) return;
    PimLightmapPage* pages = lm_page_table(&ms_pack);
    if (!pages) return;
    if (ok(pimgpu_lightmap_set_download(ms_lmSet, pages), "LmPack_Sync")) ms_lmDeviceNewer = false;
    free(pages);
}

void LmPack_Invalidate(void) { LmPack_Sync(); lm_set_drop(); }

void LmPack_SetResident(i32 mode)
{
    if (mode < 0 || mode > 2) mode = 0;
    if (mode != ms_lmMode) LmPack_Invalidate();
    ms_lmMode = mode;
}

void LmPack_Del(LmPack* pack)
{
    if (!pack) return;
    if (pack == &ms_pack) lm_set_drop();
    for (i32 i = 0; i < pack->lmCount; ++i) Lightmap_Del(pack->lightmaps + i);      /* lightmap.c:1104-1115 */
    free(pack->lightmaps);
    memset(pack, 0, sizeof(*pack));
}

void LmPack_EmbedAttributes(const PimDrawables* drawables, Lightmap* lightmaps, const i32* texIds, i32 lmCount)
{
    if (!drawables || !lightmaps || lmCount <= 0) return;
    PimLightmapPage* pages = calloc((size_t)lmCount, sizeof(PimLightmapPage));
    if (!pages) return;
    for (i32 i = 0; i < lmCount; ++i) pages[i].block = &lightmaps[i].probes[0]->x;
    if (ms_pack.lightmaps == lightmaps) LmPack_Invalidate();            /* position / normal / sampleCounts change under a resident copy */
    ok(pimgpu_lightmap_embed(drawables, texIds, pages, lmCount, lightmaps[0].size), "LmPack_EmbedAttributes");
    free(pages);
}

void LmPack_Bake(PtScene* scene, float timeSlice, i32 spp)
{
    const LmPack* pack = LmPack_Get();                                   /* lightmap.c:1211 */
    if (!scene || pack->lmCount <= 0 || !pack->lightmaps) return;
    PimLightmapPage* pages = lm_page_table(pack);
    if (!pages) return;
    scene_refresh(scene);                                                /* PtScene_Update, lightmap.c:1209; its UpdateDists runs inside the bake call */
    if (ms_lmMode == 0)
    {
        if (ok(pimgpu_lightmap_bake(scene->gpu, pages, pack->lmCount, pack->lmSize, pack->texelsPerMeter, timeSlice, spp > 1 ? spp : 1,
            ms_counters.lmBake, ms_config.seed, ms_config.maxBounces, 0, -1), "LmPack_Bake"))
            ms_counters.lmBake++;
        free(pages);
        return;
    }
    /* resident: the pages go up once (again whenever the pack is another one), every call bakes in HBM */
    if (ms_lmSet && (ms_lmSetCount != pack->lmCount || ms_lmSetSize != pack->lmSize || ms_lmSetKey != (const void*)pack->lightmaps[0].probes[0]))
        lm_set_drop();                                                     /* another pack: the old set has nowhere to go */
    if (!ms_lmSet)
    {
        if (!ok(pimgpu_lightmap_set_create(pack->lmCount, pack->lmSize, &ms_lmSet), "LmPack_Bake") ||
            !ok(pimgpu_lightmap_set_upload(ms_lmSet, pages), "LmPack_Bake")) { lm_set_drop(); free(pages); return; }
        ms_lmSetCount = pack->lmCount; ms_lmSetSize = pack->lmSize; ms_lmSetKey = pack->lightmaps[0].probes[0];
    }
    if (ok(pimgpu_lightmap_set_bake(scene->gpu, ms_lmSet, pack->texelsPerMeter, timeSlice, spp > 1 ? spp : 1,
        ms_counters.lmBake, ms_config.seed, ms_config.maxBounces, 0, -1), "LmPack_Bake"))
    {
        ms_counters.lmBake++;
        ms_lmDeviceNewer = true;
        if (ms_lmMode == 1) LmPack_Sync();                               /* host pages up to date after every call, as in the reference */
    }
    free(pages);
}
