/*
 * pt_api.h -- host objects behind the opaque handles of include/pimgpu.h.
 */
#ifndef PT_API_H
#define PT_API_H

#include "pt_host.h"
#include <vector>
#include <utility>
#include <functional>

/* PtScene (rendering/path_tracer.c:122-166) resident in HBM */
struct PimScene
{
    pt::DScene d = {};
    pt::DScene* sceneDev = nullptr;      /* copy of d in HBM (kernels take d by value; out-of-line device functions take this) */
    /* owned device memory */
    pt::v4* positions = nullptr; pt::v4* normals = nullptr; pt::v2* uvs = nullptr; int32_t* matIds = nullptr;
    int32_t* vertToEmit = nullptr; int32_t* emitToVert = nullptr;
    pt::DTriRec* tri = nullptr;          /* per-triangle records of the shading stages (pt_device.h) */
    PimMaterial* materials = nullptr; pt::DTexture* textures = nullptr;
    std::vector<void*> texels;
    pt::v3* sky[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    pt::v2* lut = nullptr;
    int32_t* cellLength = nullptr; float* pdf = nullptr; float* cdf = nullptr; uint32_t* live = nullptr;
    uint32_t* cellSum = nullptr; float* cellIntegral = nullptr;
    pt::BvhStorage bvh;
    uint32_t seed = 0;
    float metersPerCell = 1.5f;
    float bvhBuildMs = 0.0f, setupMs = 0.0f;
    int32_t bvhDepth = 0;
    /* multi-device (pimgpu_init_devices with more than one entry): this is the replica of context 0, peers[i] the one of
     * context i (peers[0] unused); liveBase = the light histograms as they were after the last UpdateDists, on context 0,
     * against which every replica's own increments are measured when the histograms are summed (api.hip, scene_update_dists_all) */
    PimScene* peers[16] = {};
    uint32_t* liveBase = nullptr; uint32_t* liveTmp = nullptr;
};

/* PtTrace (rendering/path_tracer.h:44-52): running-mean float3 images in HBM */
struct PimTrace
{
    int32_t width = 0, height = 0;
    float* color = nullptr; float* albedo = nullptr; float* normal = nullptr;
    PimTraceStats stats = {};
    PimTrace* peers[16] = {};            /* multi-device: context i keeps rows [H*i/G, H*(i+1)/G) of ITS copy up to date */
    bool gathered = true;                /* multi-device: the copy of context 0 holds every band (stddev / buffers gather on demand) */
    /* pimgpu_trace_image_samples: this context's running mean of ITS samples (made on first use; every context's copy has its own),
     * and on context 0 the landing place of the other contexts' means on their way into the sum */
    PimTrace* sampleAcc = nullptr;
    PimTrace* sampleStage = nullptr;
    /* pimgpu_trace_image_host: ns per pixel of tracing and of the copy home on the previous call (0: no call yet) */
    float hostTraceNs = 0.0f, hostCopyNs = 0.0f;
    int32_t hostN = 0, hostFirst = 0;    /* ... and that call's pixels and first share */
};

namespace pt {

/* per-path state of the wavefront integrator, SoA over path slots (DESIGN.md "path state") */
struct PathState
{
    uint4* rng;          /* Prng stream */
    float4* ro;          /* ray origin (w unused) */
    float4* rd;          /* ray direction (w unused) */
    float4* atten;       /* attenuation.xyz, resultWeight */
    float4* lum;         /* luminance.xyz, w = bits: bounce | prevFlags<<16 */
    float4* albedo;      /* albedo accumulator.xyz */
    float4* normal;      /* normal accumulator.xyz */
    float4* nee0;        /* between shade_a and shade_b: mode, iVert, selectPdf, distSq */
    float4* nee1;        /* BRDF-branch sample: attenuation.xyz, pdf */
};

/* library-owned work buffers, grown on demand and reused by every call */
struct WorkBuffers
{
    int64_t rayCap = 0, pathCap = 0;
    float4* rayO = nullptr; float4* rayD = nullptr;           /* packed rays for the public intersect call */
    /* integrator */
    PathState ps = {};
    float4* extO[2] = { nullptr, nullptr }; float4* extD[2] = { nullptr, nullptr };   /* extension rays, by queue position */
    float4* neeO = nullptr; float4* neeD = nullptr;
    DHit* extHit = nullptr; DHit* neeHit = nullptr;
    int32_t* parkSlot = nullptr; float4* parkO = nullptr; float4* parkD = nullptr;   /* paths parked for the tail kernel (integrate.hip) */
    int32_t* neeTarget = nullptr;       /* per NEE ray: the light triangle it is aimed at (light-sampling branch), -1 = a plain closest-hit query */
    int32_t* queue[2] = { nullptr, nullptr };
    int32_t* slowA = nullptr; int32_t* slowB = nullptr;     /* queue positions deferred to the SLOW builds of the stages */
    int32_t* counters = nullptr;        /* [0],[1]: queue lengths; [2],[3]: slow-list lengths; device side */
    unsigned long long* stats = nullptr;/* ext, shadow, other rays */
    PimResult* results = nullptr;       /* per path slot */
    int ensure_rays(int64_t n);
    int ensure_paths(int64_t n);
    void release();
};

/*
 * One share of a batch of paths, run on its own stream with its own device-side counters and traversal resources so that
 * several shares of one image are in flight at once: the deep bounces of a share are latency-bound (few paths, long
 * dependent launches), and another share's early, throughput-bound bounces fill the machine meanwhile.
 * Group 0 is the library's main stream with WorkBuffers::counters/stats and Library::trace.
 */
struct PathGroup
{
    hipStream_t stream = nullptr;
    int32_t* counters = nullptr; unsigned long long* stats = nullptr;      /* device, 256 bytes each */
    TraceResources* trace = nullptr; TraceResources ownTrace;
    hipEvent_t done = nullptr, stagger = nullptr;
    /* what the host reads after a call -- statistics, queue lengths, the traversal status word, the frame's focal length -- is
     * gathered by one small kernel (k_collect) and comes home in ONE copy into page-locked memory: integrate.hip, ReadBack */
    void* devBack = nullptr; void* hostBack = nullptr;
    /* queue length at the start of each wavefront bounce of the previous integrate call (host copy): the tail policy */
    int32_t history[64] = {}; int32_t historyBounces = 0; int64_t historyN = -1;
    /* the call in flight */
    int32_t first = 0, n = 0, wavefrontBounces = 0, tailBounce = -1;     /* tailBounce: bounce at which k_tail took over, -1 = never */
    struct Timed { int cls; hipEvent_t t0, t1; };             /* cls: 0 traversal, 1 shade_a, 2 shade_b, 3 tail */
    std::vector<Timed> events;
    std::vector<hipEvent_t> eventPool; size_t eventsUsed = 0;   /* timing events are made once and reused: creating and
                                                                    destroying 14 per frame was host time with the GPU idle */
    hipEvent_t take_event();                                    /* nullptr once the call has timed kMaxTimedDispatches dispatches */
    void trim_events();                                         /* after a call: the pool shrinks back to a few frames' worth */
};
static constexpr int kMaxGroups = 8;

struct Library
{
    bool initialised = false;
    int device = 0;
    int cuCount = 0;
    hipStream_t stream = nullptr;
    TraceResources trace;
    WorkBuffers work;
    hipEvent_t frameBegin = nullptr, frameEnd = nullptr;     /* pimgpu_trace_image's totalMs */
    hipEvent_t copyBegin = nullptr, copyEnd = nullptr;       /* around the first share's copies home (pimgpu_trace_image_host) */
    float* smallBack = nullptr;                              /* 64 page-locked bytes made at init: where pimgpu_trace_stddev_noalloc's two floats land (a copy
                                                                into pageable memory is staged by the runtime, which may allocate) */
    PathGroup groups[kMaxGroups];
    int ensure_group(int g);
    void release_groups();
};
/*
 * One Library per entry of the device list of pimgpu_init_devices (a HIP device may appear more than once: two contexts on
 * one GPU, which is how the one-GPU test box exercises the multi-device paths).  Every call runs against the CURRENT
 * context of its host thread -- context 0 unless a DeviceScope says otherwise -- and the code below the entry points
 * reaches it through the name it always used, g_lib.
 */
static constexpr int kMaxDevices = 16;
extern Library g_libs[kMaxDevices];
extern int g_libCount;                       /* contexts initialised by pimgpu_init_devices */
Library& cur_lib();
void set_cur_lib(Library* lib);
#define g_lib (pt::cur_lib())
struct DeviceScope
{
    Library* prev;
    explicit DeviceScope(int index) : prev(&cur_lib()) { set_cur_lib(&g_libs[index]); hipSetDevice(g_libs[index].device); }
    ~DeviceScope() { set_cur_lib(prev); hipSetDevice(prev->device); }
};

int check_ready();

/* api.hip: UpdateDists on every replica, after summing the replicas' light-histogram increments (one replica: just UpdateDists) */
int scene_update_dists_all(PimScene* scene);
/* the rows of an image context i of G renders */
inline void band_of(int32_t height, int i, int G, int32_t* rowBegin, int32_t* rowEnd) { *rowBegin = (int32_t)((int64_t)height * i / G); *rowEnd = (int32_t)((int64_t)height * (i + 1) / G); }
/* run fn(i) for i in [0, count) on one host thread per context, each inside its DeviceScope; returns the first failure and keeps its message */
int for_each_context(int count, const std::function<int(int)>& fn);

/* setup.hip */
int pack_trirecs(const DScene& s, DTriRec* out, hipStream_t stream);
int setup_emission_pdfs(const DScene& s, int32_t attempts, uint32_t seed, float* pdfsDev, hipStream_t stream);
int setup_light_grid(const DScene& s, TraceResources* res, float metersPerCell, uint32_t seed, int32_t* cellLengthDev, hipStream_t stream);
int update_dists(const DScene& s, hipStream_t stream);

/* integrate.hip */
int stddev_scratch_setup(int device);      /* address of this device's copy of the no-alloc scratch array */
int pack_rays(const float4* ro, const float4* rd, const float* tNear, const float* tFar, int32_t n, float4* rayO, float4* rayD, hipStream_t stream);

} /* namespace pt */

#endif
