/*
 * pim_host.c -- Pim's path-tracer entry points (rendering/path_tracer.h:67-103, lightmap.h:69) as thin C
 * wrappers over libpimgpu.so.  No arithmetic of the path happens here: every function marshals its
 * arguments, makes one C-ABI call and hands the result back in the reference's host-side shape.
 */
#include "pim_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char ms_error[512];

static bool ok(int rc, const char* what)
{
    if (rc == PIMGPU_OK) return true;
    snprintf(ms_error, sizeof(ms_error), "%s: %s", what, pimgpu_last_error());
    fprintf(stderr, "[pt] %s\n", ms_error);          /* Con_Logf(LogSev_Error, "pt", ...) in a Pim build */
    return false;
}

const char* PtSys_LastError(void) { return ms_error; }

bool PtSys_Init(void)
{
    const char* dev = getenv("PIM_GPU_DEVICE");
    return ok(pimgpu_init(dev ? atoi(dev) : 0), "PtSys_Init");
}

void PtSys_Shutdown(void) { pimgpu_shutdown(); }

PtScene* PtScene_New(const PimSceneDesc* snapshot)
{
    PtScene* scene = NULL;
    if (!ok(pimgpu_scene_create(snapshot, &scene), "PtScene_New")) return NULL;
    return scene;
}

PtScene* PtScene_NewFromDrawables(const PimDrawables* drawables, const PimSceneDesc* rest)
{
    if (!rest) { snprintf(ms_error, sizeof(ms_error), "PtScene_NewFromDrawables: NULL snapshot"); return NULL; }
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
        scene = PtScene_New(&desc);
    }
    free(positions); free(normals); free(uvs); free(matIds); free(materials);      /* the scene keeps its own copies in HBM */
    return scene;
}

BrdfLut BrdfLut_New(int2 size, i32 numSamples, uint32_t seed)
{
    BrdfLut lut = { size, NULL };
    if (size.x <= 0 || size.y <= 0) { lut.size.x = lut.size.y = 0; return lut; }
    lut.texels = calloc((size_t)size.x * (size_t)size.y, sizeof(float2));           /* Tex_Calloc, lighting.c:92 */
    if (lut.texels) BrdfLut_Update(&lut, 0, numSamples, seed);
    return lut;
}

void BrdfLut_Update(BrdfLut* lut, i32 prevSamples, i32 newSamples, uint32_t seed)
{
    if (lut && lut->texels && (newSamples > prevSamples))                           /* lighting.c:110 */
        ok(pimgpu_brdf_lut_bake(&lut->texels[0].x, lut->size.x, lut->size.y, prevSamples, newSamples, seed), "BrdfLut_Update");
}

void BrdfLut_Del(BrdfLut* lut) { if (lut) { free(lut->texels); memset(lut, 0, sizeof(*lut)); } }

void PtScene_Update(PtScene* scene) { if (scene) ok(pimgpu_scene_update_dists(scene), "PtScene_Update"); }
void PtScene_Del(PtScene* scene) { if (scene) pimgpu_scene_destroy(scene); }

void PtTrace_New(PtTrace* trace, int2 imageSize)
{
    if (!trace) return;
    memset(trace, 0, sizeof(*trace));
    const size_t texelCount = (size_t)imageSize.x * (size_t)imageSize.y;
    if (!texelCount) return;
    if (!ok(pimgpu_trace_create(imageSize.x, imageSize.y, &trace->gpu), "PtTrace_New")) return;
    trace->imageSize = imageSize;
    /* Tex_Calloc in the reference; page-locked here: Pt_Trace copies the three images back on every call */
    trace->color = pimgpu_host_alloc(texelCount * sizeof(float3));
    trace->albedo = pimgpu_host_alloc(texelCount * sizeof(float3));
    trace->normal = pimgpu_host_alloc(texelCount * sizeof(float3));
    trace->denoised = pimgpu_host_alloc(texelCount * sizeof(float3));
    trace->sampleWeight = 1.0f;
    trace->frameDt = 1.0f / 60.0f;
}

void PtTrace_Del(PtTrace* trace)
{
    if (!trace) return;
    pimgpu_trace_destroy(trace->gpu);
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
    dof->autoFocus = 1;
    dof->autoFocusSpeed = 3.0f;
}

PtRayHit Pt_Intersect(PtScene* scene, float4 ro, float4 rd, float tNear, float tFar)
{
    PimRayHit h; memset(&h, 0, sizeof(h)); h.wuvt[3] = -1.0f; h.iVert = -1;
    ok(pimgpu_intersect(scene, &ro.x, &rd.x, &tNear, &tFar, 1, &h), "Pt_Intersect");
    PtRayHit out;
    memcpy(&out.wuvt, h.wuvt, sizeof(float4)); memcpy(&out.normal, h.normal, sizeof(float4));
    out.type = h.type; out.iVert = h.iVert; out.flags = h.flags;
    return out;
}

PtResult Pt_TraceRay(PtScene* scene, float4 ro, float4 rd, uint32_t sampleIndex, uint32_t seed, i32 maxBounces)
{
    PimResult r; memset(&r, 0, sizeof(r));
    ok(pimgpu_trace_rays(scene, &ro.x, &rd.x, 1, sampleIndex, seed, maxBounces, &r), "Pt_TraceRay");
    PtResult out; memcpy(&out, &r, sizeof(out));
    return out;
}

void Pt_Trace(PtTrace* trace, PtDofInfo* dof, PtScene* scene, const Camera* camera)
{
    if (!trace || !trace->gpu || !dof || !scene || !camera) return;
    PimTraceParams p;
    p.sampleWeight = trace->sampleWeight;
    p.frameDt = trace->frameDt;
    p.sampleIndex = trace->sampleIndex;
    p.seed = trace->seed;
    p.maxBounces = trace->maxBounces;
    p.updateDists = 1;                                   /* PtScene_Update, path_tracer.c:2565 */
    if (!ok(pimgpu_trace_image(scene, trace->gpu, camera, dof, &p), "Pt_Trace")) return;
    trace->sampleIndex++;
    /* the reference's callers read the host images right after Pt_Trace (render_system.c:306-352) */
    ok(pimgpu_trace_download(trace->gpu, &trace->color->x, &trace->albedo->x, &trace->normal->x), "Pt_Trace");
}

PtResults Pt_RayGen(PtScene* scene, float4 origin, i32 count, uint32_t sampleIndex, uint32_t seed, i32 maxBounces)
{
    PtResults r = { NULL, NULL };
    if (count <= 0) return r;
    r.colors = calloc((size_t)count, sizeof(float4));
    r.directions = calloc((size_t)count, sizeof(float4));
    if (!ok(pimgpu_raygen(scene, &origin.x, count, sampleIndex, seed, maxBounces, &r.colors->x, &r.directions->x), "Pt_RayGen"))
        PtResults_Del(&r);
    return r;
}

void PtResults_Del(PtResults* results)
{
    if (!results) return;
    free(results->colors); free(results->directions);
    results->colors = NULL; results->directions = NULL;
}

void Cubemap_Bake(Cubemap* cm, PtScene* scene, float4 origin, float weight, uint32_t sampleIndex, uint32_t seed, i32 maxBounces)
{
    if (!cm || !scene || cm->size <= 0) return;                      /* cubemap.c:168-173 */
    float* faces[6];
    for (int f = 0; f < 6; ++f) faces[f] = &cm->color[f]->x;
    ok(pimgpu_cubemap_bake(scene, faces, cm->size, &origin.x, weight, sampleIndex, seed, maxBounces), "Cubemap_Bake");
}

i32 AmbCube_Bake(PtScene* scene, AmbCube_t* pCube, float4 origin, i32 samples, i32 prevSampleCount, uint32_t sampleIndex, uint32_t seed, i32 maxBounces)
{
    i32 count = prevSampleCount;
    if (!scene || !pCube) return count;
    ok(pimgpu_ambcube_bake(scene, &pCube->Values[0].x, &origin.x, samples, prevSampleCount, sampleIndex, seed, maxBounces, &count), "AmbCube_Bake");
    return count;
}

void Lightmap_New(Lightmap* lm, i32 size)
{
    memset(lm, 0, sizeof(*lm));
    if (size <= 0) return;
    lm->size = size;
    const size_t texelcount = (size_t)size * (size_t)size;
    /* [5] float4 | float3 | float3 | float, lightmap.c:90-110; Tex_Calloc there, page-locked here: LmPack_Bake copies a
     * page (113 MB at 1024^2) to the device and back on every call */
    float* allocation = pimgpu_host_alloc(texelcount * 27 * sizeof(float));
    if (!allocation) { lm->size = 0; return; }
    for (i32 i = 0; i < 5; ++i) lm->probes[i] = (float4*)(allocation + texelcount * 4 * (size_t)i);
    lm->position = (float3*)(allocation + texelcount * 20);
    lm->normal = (float3*)(allocation + texelcount * 23);
    lm->sampleCounts = allocation + texelcount * 26;
}

void Lightmap_Del(Lightmap* lm)
{
    if (lm) { pimgpu_host_free(lm->probes[0]); memset(lm, 0, sizeof(*lm)); }
}

void LmPack_EmbedAttributes(const PimDrawables* drawables, Lightmap* lightmaps, const i32* texIds, i32 lmCount)
{
    if (!drawables || !lightmaps || lmCount <= 0) return;
    PimLightmapPage* pages = calloc((size_t)lmCount, sizeof(PimLightmapPage));
    for (i32 i = 0; i < lmCount; ++i) pages[i].block = &lightmaps[i].probes[0]->x;
    ok(pimgpu_lightmap_embed(drawables, texIds, pages, lmCount, lightmaps[0].size), "LmPack_EmbedAttributes");
    free(pages);
}

void LmPack_Bake(PtScene* scene, LmPack* pack, float timeSlice, i32 spp, uint32_t passIndex, uint32_t seed, i32 maxBounces)
{
    if (!scene || !pack || pack->lmCount <= 0) return;
    PimLightmapPage* pages = calloc((size_t)pack->lmCount, sizeof(PimLightmapPage));
    for (i32 i = 0; i < pack->lmCount; ++i) pages[i].block = &pack->lightmaps[i].probes[0]->x;   /* lightmap.c:95-110 */
    ok(pimgpu_lightmap_bake(scene, pages, pack->lmCount, pack->lmSize, pack->texelsPerMeter, timeSlice, spp,
        passIndex, seed, maxBounces, 0, -1), "LmPack_Bake");
    free(pages);
}
