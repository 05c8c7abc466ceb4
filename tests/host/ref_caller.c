/*
 * ref_caller.c -- TEST PROGRAM: a caller written against the REFERENCE's own interface.
 *
 * The declarations in the first block are the prototypes and structs of Pim's headers (rendering/path_tracer.h:13-103,
 * rendering/lightmap.h:27-69, rendering/cubemap.h:28-34,209-213, math/ambcube.h:69-74, rendering/camera.h:8-15,
 * math/types.h) as a translation unit of Pim sees them -- this file does NOT include pim_amd/host/pim_host.h.  It is
 * linked against libpimhost.so; tests/test_gpu_host_api.py runs it on the GPU box and compares everything it writes
 * with the oracle, bit for bit.  If a signature or struct layout of the host layer differed from the reference's, the
 * arguments would arrive scrambled and the comparison would fail.
 *
 * The second block is the adapter a Pim build adds: the scene-snapshot provider, which in Pim reads
 * Entities_Get() / Texture_Get() / g_BrdfLut and here reads the scene file the test wrote.
 *
 *   ref_caller <scene.bin> <out.bin>
 */
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- block 1: the reference's interface ------------------------------------------------------------------------------ */
typedef int32_t i32;
typedef uint32_t u32;
typedef struct float2_s { _Alignas(8) float x; float y; } float2;
typedef struct float3_s { float x; float y; float z; } float3;
typedef struct float4_s { _Alignas(16) float x; float y; float z; float w; } float4;
typedef struct int2_s { _Alignas(8) i32 x; i32 y; } int2;
typedef struct quat_s { float4 v; } quat;

typedef struct Camera_s { float4 position; quat rotation; float zNear; float zFar; float fovy; } Camera;
typedef struct PtScene_s PtScene;
typedef enum { PtHit_Nothing = 0, PtHit_Backface, PtHit_Triangle, PtHit_COUNT } PtHitType;
typedef struct PtRayHit_s { float4 wuvt; float4 normal; PtHitType type; i32 iVert; u32 flags; } PtRayHit;
typedef struct PtDofInfo_s { float aperture; float focalLength; i32 bladeCount; float bladeRot; float focalPlaneCurvature; float autoFocusSpeed; bool autoFocus; } PtDofInfo;
typedef struct PtTrace_s { float3* color; float3* albedo; float3* normal; float3* denoised; int2 imageSize; float sampleWeight; } PtTrace;
typedef struct PtResult_s { float3 color; float3 albedo; float3 normal; } PtResult;
typedef struct PtResults_s { float4* colors; float4* directions; } PtResults;

void PtSys_Init(void);
void PtSys_Update(void);
void PtSys_Shutdown(void);
PtScene* PtScene_New(void);
void PtScene_Update(PtScene* scene);
void PtScene_Del(PtScene* scene);
void PtScene_Gui(PtScene* scene);
void PtTrace_New(PtTrace* trace, int2 imageSize);
void PtTrace_Del(PtTrace* trace);
void DofInfo_New(PtDofInfo* dof);
void DofInfo_Gui(PtDofInfo* dof);
PtRayHit Pt_Intersect(const PtScene* scene, float4 ro, float4 rd, float tNear, float tFar);
PtResult Pt_TraceRay(PtScene* scene, float4 ro, float4 rd);
void Pt_Trace(PtTrace* trace, PtDofInfo* dof, PtScene* scene, const Camera* camera);
PtResults Pt_RayGen(PtScene* scene, float4 origin, i32 count);

#define kGiDirections 5
typedef struct vkrTextureId_s { u32 version : 8; u32 type : 3; u32 index : 21; } vkrTextureId;
typedef struct Lightmap_s { float4* probes[kGiDirections]; float3* position; float3* normal; float* sampleCounts; i32 size; vkrTextureId slot; } Lightmap;
typedef struct LmPack_s { float4 axii[kGiDirections]; Lightmap* lightmaps; i32 lmCount; i32 lmSize; float texelsPerMeter; } LmPack;
void Lightmap_New(Lightmap* lm, i32 size);
void Lightmap_Del(Lightmap* lm);
LmPack* LmPack_Get(void);
void LmPack_Del(LmPack* pack);
void LmPack_Bake(PtScene* scene, float timeSlice, i32 spp);

typedef struct Cubemap_s { i32 size; i32 mipCount; float3* color[6]; float4* convolved[6]; } Cubemap;
typedef struct AmbCube_s { float4 Values[6]; } AmbCube_t;
void Cubemap_New(Cubemap* cm, i32 size);
void Cubemap_Del(Cubemap* cm);
void Cubemap_Bake(Cubemap* cm, PtScene* scene, float4 origin, float weight);
void Cubemap_Convolve(Cubemap* cm, u32 sampleCount, float weight);
i32 AmbCube_Bake(PtScene* scene, AmbCube_t* pCube, float4 origin, i32 samples, i32 prevSampleCount);

/* ---- block 2: the adapter (what INTEGRATION.md adds to a Pim build) ------------------------------------------------- */
#include "../../include/pimgpu.h"
typedef bool (*PtSceneProviderFn)(PimSceneDesc* out, uint64_t* modtime, int peekModtime, void* user);
void PtSys_SetSceneProvider(PtSceneProviderFn fn, void* user);
typedef struct PtSysConfig { uint32_t seed; i32 maxBounces; float frameDt; i32 downloadAovs; i32 device; } PtSysConfig;
void PtSys_Configure(const PtSysConfig* config);
bool PtSys_Ok(void);

static PimSceneDesc g_desc;
static uint64_t g_modtime = 1;
static int g_providerFull;

static void* rd(FILE* f, size_t bytes)
{
    void* p = malloc(bytes ? bytes : 1);
    if (bytes && fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "ref_caller: short scene file\n"); exit(2); }
    return p;
}

/* the scene file: int32 header, PimMediaDesc, Camera, then the arrays in PimSceneDesc order (tests/test_gpu_host_api.py writes it) */
static Camera g_camera;
static void load_scene(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    int32_t h[8];
    if (fread(h, 4, 8, f) != 8 || h[0] != 0x53434E45) { fprintf(stderr, "ref_caller: bad scene file\n"); exit(2); }
    memset(&g_desc, 0, sizeof(g_desc));
    g_desc.vertCount = h[1]; g_desc.matCount = h[2]; g_desc.texCount = h[3]; g_desc.skySize = h[4]; g_desc.brdfLutSize = h[5];
    g_desc.emissiveAttempts = h[6]; g_desc.seed = (uint32_t)h[7];
    if (fread(&g_desc.metersPerCell, 4, 1, f) != 1 || fread(&g_desc.media, sizeof(PimMediaDesc), 1, f) != 1 || fread(&g_camera, sizeof(Camera), 1, f) != 1) exit(2);
    const size_t V = (size_t)g_desc.vertCount;
    g_desc.positions = rd(f, V * 16); g_desc.normals = rd(f, V * 16); g_desc.uvs = rd(f, V * 8); g_desc.matIds = rd(f, V * 4);
    g_desc.materials = rd(f, (size_t)g_desc.matCount * sizeof(PimMaterial));
    PimTexture* tex = calloc((size_t)g_desc.texCount + 1, sizeof(PimTexture));
    for (int32_t i = 0; i < g_desc.texCount; ++i)
    {
        int32_t t[3];
        if (fread(t, 4, 3, f) != 3) exit(2);
        tex[i].width = t[0]; tex[i].height = t[1]; tex[i].format = t[2];
        tex[i].texels = rd(f, (size_t)t[0] * (size_t)t[1] * 4);
    }
    g_desc.textures = tex;
    for (int k = 0; k < 6 && g_desc.skySize > 0; ++k) g_desc.skyFaces[k] = rd(f, (size_t)g_desc.skySize * g_desc.skySize * 12);
    g_desc.brdfLut = rd(f, (size_t)g_desc.brdfLutSize * g_desc.brdfLutSize * 8);
    fclose(f);
}

static bool provider(PimSceneDesc* out, uint64_t* modtime, int peekModtime, void* user)
{
    (void)user;
    *modtime = g_modtime;
    if (!peekModtime) { *out = g_desc; ++g_providerFull; }
    return true;
}

static void put(FILE* f, const void* p, size_t bytes) { fwrite(p, 1, bytes, f); }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_caller scene.bin out.bin\n"); return 2; }
    load_scene(argv[1]);
    FILE* out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 2; }

    PtSysConfig cfg = { 17u, 4, 1.0f / 60.0f, 1, -1 };
    PtSys_Configure(&cfg);
    PtSys_SetSceneProvider(provider, NULL);
    PtSys_Init();
    if (!PtSys_Ok()) return 3;
    PtScene* scene = PtScene_New();
    if (!scene) return 3;
    PtScene_Gui(scene);

    /* PathTrace(), render_system.c:277-358: progressive frames, the host images read right after every call */
    const int2 size = { 80, 45 };
    PtTrace trace;
    PtTrace_New(&trace, size);
    PtDofInfo dof;
    DofInfo_New(&dof);
    DofInfo_Gui(&dof);
    double checksum = 0.0;
    for (int s = 0; s < 3; ++s)
    {
        trace.sampleWeight = 1.0f / (float)(s + 1);
        Pt_Trace(&trace, &dof, scene, &g_camera);
        checksum += trace.color[size.x * 20 + 40].x;      /* read like a caller would */
    }
    const size_t px = (size_t)size.x * size.y;
    put(out, trace.color, px * 12); put(out, trace.albedo, px * 12); put(out, trace.normal, px * 12);
    put(out, &dof, sizeof(dof));

    /* single rays */
    const float4 ro = { 0.5f, -1.0f, 0.25f, 0.0f }, rdir = { 0.0f, 0.6f, 0.8f, 0.0f };
    PtRayHit hit = Pt_Intersect(scene, ro, rdir, 0.0f, 8388608.0f);
    put(out, &hit, sizeof(hit));
    for (int k = 0; k < 3; ++k) { PtResult r = Pt_TraceRay(scene, ro, rdir); put(out, &r, sizeof(r)); }

    /* Pt_RayGen: results stay valid until the end of the frame (PtSys_Update) */
    const float4 origin = { 0.25f, -0.5f, 0.5f, 0.0f };
    PtResults gen = Pt_RayGen(scene, origin, 256);
    put(out, gen.colors, 256 * 16); put(out, gen.directions, 256 * 16);
    PtSys_Update();

    /* Cubemap_Bake (cubemap.c:162-190) and AmbCube_Bake (ambcube.c:5-32) */
    Cubemap cm;
    Cubemap_New(&cm, 8);
    Cubemap_Bake(&cm, scene, origin, 1.0f);
    for (int f = 0; f < 6; ++f) put(out, cm.color[f], 64 * 12);
    /* ... and the convolution that follows it (render_system.c:239-241), twice: the second call blends into the first's result */
    Cubemap_Convolve(&cm, 64, 1.0f);
    Cubemap_Convolve(&cm, 64, 0.5f);
    const i32 mips = cm.mipCount;
    put(out, &mips, 4);
    for (int f = 0; f < 6; ++f) put(out, cm.convolved[f], (64 + 16 + 4 + 1) * 16);
    Cubemap_Del(&cm);
    AmbCube_t cube; memset(&cube, 0, sizeof(cube));
    i32 n = AmbCube_Bake(scene, &cube, origin, 200, 0);
    put(out, &cube, sizeof(cube)); put(out, &n, 4);

    /* LmPack_Bake (lightmap.c:1204-1223) on the global pack: one 32^2 page, a patch of the floor mapped */
    LmPack* pack = LmPack_Get();
    pack->lmCount = 1; pack->lmSize = 32; pack->texelsPerMeter = 4.0f;
    pack->lightmaps = calloc(1, sizeof(Lightmap));
    Lightmap_New(&pack->lightmaps[0], 32);
    Lightmap* lm = &pack->lightmaps[0];
    for (int y = 4; y < 28; ++y) for (int x = 4; x < 28; ++x)
    {
        const int i = y * 32 + x;
        lm->position[i].x = -3.0f + 0.25f * (float)x; lm->position[i].y = -4.95f; lm->position[i].z = -3.0f + 0.25f * (float)y;
        lm->normal[i].x = 0.0f; lm->normal[i].y = 1.0f; lm->normal[i].z = 0.0f;
        lm->sampleCounts[i] = 1.0f;
    }
    LmPack_Bake(scene, 1.0f, 3);
    LmPack_Bake(scene, 0.5f, 2);
    put(out, lm->probes[0], (size_t)32 * 32 * 27 * 4);

    /* a changed Entities.modtime rebuilds the scene inside PtScene_Update (path_tracer.c:1028-1032) */
    const int before = g_providerFull;
    PtScene_Update(scene);                       /* same modtime: UpdateDists only */
    const int same = g_providerFull - before;
    g_modtime = 2;
    PtScene_Update(scene);
    const int rebuilt = g_providerFull - before;
    PtRayHit hit2 = Pt_Intersect(scene, ro, rdir, 0.0f, 8388608.0f);
    put(out, &hit2, sizeof(hit2));
    int32_t tailw[4] = { same, rebuilt, (int32_t)sizeof(PtTrace), (int32_t)sizeof(Lightmap) };
    put(out, tailw, sizeof(tailw));
    fclose(out);

    LmPack_Del(pack);
    PtTrace_Del(&trace);
    PtScene_Del(scene);
    PtSys_Shutdown();
    printf("ref_caller ok %f\n", checksum);
    return 0;
}
