/*
 * pt_host.h -- host-side declarations internal to libpimgpu (not part of the C-ABI).
 */
#ifndef PT_HOST_H
#define PT_HOST_H

#include "pt_device.h"
#include <string>
#include <vector>

namespace pt {

/* device memory of the acceleration structure + the build's scratch (kept for rebuilds) */
struct BvhStorage
{
    float4* nodes = nullptr;
    uint4* cnodes = nullptr;            /* 64-byte compressed nodes (pt_device.h) */
    float4* tris = nullptr;
    void* scratch = nullptr;
    size_t scratchBytes = 0;
    int32_t capacityTris = 0;
    int32_t triCount = 0;
    int32_t nodeCount = 0;
    int32_t depth = 0;                  /* levels of the 4-wide tree of the last build */
    int32_t* progress = nullptr;        /* pinned host memory the clustering rounds report to: [0] list length, [1] rounds done */
    void release();
    DBvh view() const { DBvh b; b.nodes = nodes; b.cnodes = cnodes; b.tris = tris; b.nodeCount = nodeCount; b.triCount = triCount; return b; }
};

/* bvh_build.hip: full (re)build from device positions (float4[3*triCount]) on stream; returns with the stream drained */
int bvh_build(BvhStorage* b, const v4* positionsDev, int32_t triCount, hipStream_t stream);

int bvh_validate(const BvhStorage* b, hipStream_t stream, std::string* why);

/* per-process traversal resources: overflow area of the per-lane traversal stacks */
struct TraceResources
{
    unsigned long long* spill = nullptr;
    uint32_t* status = nullptr;          /* device word: bit 0 = traversal stack overflow */
    unsigned long long* work = nullptr;  /* device: nodes fetched, triangles tested, rays (counting build of the kernel) */
    bool counting = false;               /* pimgpu_work_counters(1, ..): launch the counting build */
    int32_t gridBlocks = 0;
    int chunkParity = 0;                 /* which of the two ray-chunk counters (status words 24, 25) the next launch uses */
    int ensure();
    void release();
};
/* entries a traversal stack may hold before the kernels report overflow: PT_STACK_MAX; PIMGPU_STACK_CAP (tests only)
 * lowers it, never below the LDS part.  Read per launch. */
int32_t trace_stack_cap();

/* trace.hip */
/* targets (optional, per ray): >= 0 makes the ray a VISIBILITY query of that triangle -- hits[i] already holds the triangle's own
 * record, the ray's tFar is its distance, and the launch only overwrites the record with a miss when something lies in front
 * (trace.hip) */
int trace_closest(const DBvh& bvh, TraceResources* res, const float4* rayO, const float4* rayD, DHit* hits,
    int32_t n, const int32_t* countDev, hipStream_t stream, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr, const int32_t* targets = nullptr);
int trace_any(const DBvh& bvh, TraceResources* res, const float4* rayO, const float4* rayD, int32_t* occluded,
    int32_t n, hipStream_t stream);
int make_rayhits(const DScene& scene, const float4* rayD, const DHit* hits, PimRayHit* out, int32_t n, hipStream_t stream);

void set_error(const char* fmt, ...);

#define PT_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    pt::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return PIMGPU_E_DEVICE; } } while (0)

} /* namespace pt */

#endif
