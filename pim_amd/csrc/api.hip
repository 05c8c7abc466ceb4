/*
 * api.hip -- the C-ABI of libpimgpu.so (include/pimgpu.h): library state, scene upload and
 * set-up, batched ray queries.  Image, ray-batch and lightmap entry points are in integrate.hip.
 */
#include "pt_host.h"
#include "pt_api.h"
#include "traverse.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <csignal>
#include <execinfo.h>
#include <unistd.h>

namespace pt {

static thread_local char g_error[512] = "";
Library g_libs[kMaxDevices];
int g_libCount = 0;
static thread_local Library* tl_lib = nullptr;
Library& cur_lib() { return tl_lib ? *tl_lib : g_libs[0]; }
void set_cur_lib(Library* lib) { tl_lib = lib; }

void set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int check_ready()
{
    if (!g_lib.initialised) { set_error("pimgpu_init has not been called"); return PIMGPU_E_STATE; }
    return PIMGPU_OK;
}

template <class T>
static int upload(T** dst, const void* src, size_t count)
{
    *dst = nullptr;
    const size_t bytes = sizeof(T) * (count ? count : 1);
    if (hipMalloc((void**)dst, bytes) != hipSuccess) { set_error("hipMalloc of %zu bytes failed", bytes); return PIMGPU_E_NOMEM; }
    if (count && src) PT_HIP(hipMemcpy(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice));
    else PT_HIP(hipMemsetAsync(*dst, 0, bytes, g_lib.stream));   /* ordered with the kernels (the stream is non-blocking) */
    return PIMGPU_OK;
}

/* media_desc_update, path_tracer.c:1963-1987 */
static void media_from_desc(DMedia* m, const PimMediaDesc* d)
{
    m->constantColor = f4_v(d->constantColor[0], d->constantColor[1], d->constantColor[2], d->constantColor[3]);
    m->noiseColor = f4_v(d->noiseColor[0], d->noiseColor[1], d->noiseColor[2], d->noiseColor[3]);
    m->constantMfp = d->constantMfp; m->noiseMfp = d->noiseMfp; m->absorption = d->absorption;
    m->noiseOctaves = d->noiseOctaves; m->noiseGain = d->noiseGain; m->noiseLacunarity = d->noiseLacunarity;
    m->noiseFreq = d->noiseFreq; m->noiseScale = d->noiseScale; m->noiseHeight = d->noiseHeight;
    m->phaseDirA = d->phaseDirA; m->phaseDirB = d->phaseDirB; m->phaseBlend = d->phaseBlend;
    const v4 constantMfp = f4_lerpsv(m->constantMfp * 0.5f, m->constantMfp * 2.0f, m->constantColor);
    const v4 noiseMfp = f4_lerpsv(m->noiseMfp * 0.5f, m->noiseMfp * 2.0f, m->noiseColor);
    m->constantMu = f4_rcp(constantMfp);
    m->noiseMu = f4_rcp(noiseMfp);
    m->phaseDirA = f1_clamp(m->phaseDirA, -0.99f, 0.99f);
    m->phaseDirB = f1_clamp(m->phaseDirB, -0.99f, 0.99f);
    m->phaseBlend = f1_sat(m->phaseBlend);
    float sum = 0.0f, amplitude = 1.0f;
    for (int32_t i = 0; i < m->noiseOctaves; ++i) { sum += amplitude; amplitude *= m->noiseGain; }
    m->noiseRange = sum * m->noiseScale * 1.5f;
    /* CalcMajorant, path_tracer.c:2183-2196 */
    const float a = 1.0f + m->absorption;
    const v4 maj = f4_muls(f4_add(f4_muls(m->constantMu, a), f4_muls(m->noiseMu, a)), 2.0f);
    m->rcpMajorant = 1.0f / f4_hmax3(maj);
}

static void scene_free(PimScene* s)
{
    if (!s) return;
    hipFree(s->positions); hipFree(s->normals); hipFree(s->uvs); hipFree(s->matIds);
    hipFree(s->vertToEmit); hipFree(s->emitToVert); hipFree(s->tri); hipFree(s->materials); hipFree(s->textures);
    for (void* p : s->texels) hipFree(p);
    for (int f = 0; f < 6; ++f) hipFree(s->sky[f]);
    hipFree(s->lut);
    hipFree(s->sceneDev);
    hipFree(s->cellLength); hipFree(s->pdf); hipFree(s->cdf); hipFree(s->live); hipFree(s->cellSum); hipFree(s->cellIntegral);
    s->bvh.release();
    delete s;
}

static int check_status_word(const char* what)
{
    uint32_t st = 0;
    PT_HIP(hipMemcpy(&st, g_lib.trace.status, 4, hipMemcpyDeviceToHost));
    if (st & 1u)
    {
        PT_HIP(hipMemset(g_lib.trace.status, 0, 4));
        set_error("%s: traversal stack overflow (tree deeper than %d entries)", what, PT_STACK_MAX);
        return PIMGPU_E_LIMIT;
    }
    return PIMGPU_OK;
}

static int scene_sync_device_copy(PimScene* s)
{
    if (!s->sceneDev && hipMalloc(&s->sceneDev, sizeof(DScene)) != hipSuccess) { set_error("out of device memory"); return PIMGPU_E_NOMEM; }
    PT_HIP(hipMemcpyAsync(s->sceneDev, &s->d, sizeof(DScene), hipMemcpyHostToDevice, g_lib.stream));
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    return PIMGPU_OK;
}

static int scene_build_bvh(PimScene* s)
{
    hipEvent_t e0, e1;
    PT_HIP(hipEventCreate(&e0)); PT_HIP(hipEventCreate(&e1));
    PT_HIP(hipEventRecord(e0, g_lib.stream));
    int rc = bvh_build(&s->bvh, s->positions, s->d.triCount, g_lib.stream);
    PT_HIP(hipEventRecord(e1, g_lib.stream));
    PT_HIP(hipEventSynchronize(e1));
    hipEventElapsedTime(&s->bvhBuildMs, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (rc) { if (rc == PIMGPU_E_NOMEM) set_error("out of device memory building the BVH"); return rc; }
    s->d.bvh = s->bvh.view();
    if ((rc = scene_sync_device_copy(s))) return rc;
    if (getenv("PIMGPU_VALIDATE_BVH"))
    {
        std::string why;
        if ((rc = bvh_validate(&s->bvh, g_lib.stream, &why))) { set_error("BVH validation failed: %s", why.c_str()); return rc; }
    }
    s->bvhDepth = s->bvh.depth;
    return PIMGPU_OK;
}

int for_each_context(int count, const std::function<int(int)>& fn)
{
    if (count <= 1) return fn(0);
    std::vector<int> rcs((size_t)count, PIMGPU_OK);
    std::vector<std::string> msgs((size_t)count);
    std::vector<std::thread> threads;
    for (int i = 0; i < count; ++i)
        threads.emplace_back([&, i]() {
            DeviceScope scope(i);
            rcs[(size_t)i] = fn(i);
            if (rcs[(size_t)i]) msgs[(size_t)i] = g_error;       /* the worker's thread-local message */
        });
    for (auto& t : threads) t.join();
    for (int i = 0; i < count; ++i)
        if (rcs[(size_t)i]) { set_error("context %d (device %d): %s", i, g_libs[i].device, msgs[(size_t)i].c_str()); return rcs[(size_t)i]; }
    return PIMGPU_OK;
}

/* live0 += liveD - base: the increments replica D made since the last UpdateDists */
__global__ __launch_bounds__(256) void k_live_merge(uint32_t* __restrict__ live0, const uint32_t* __restrict__ liveD, const uint32_t* __restrict__ base, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) live0[i] += liveD[i] - base[i];
}

/*
 * UpdateDists (path_tracer.c:2471-2483) with replicas.  LightOnHit's feedback (integer atomics into live[]) lands in the replica
 * of the device that traced the path, so before Dist1D_Update runs the replicas' increments are summed: replica 0 adds
 * (live_d - base) of every other replica, base being live[] as the last UpdateDists left it (identical everywhere), and
 * the sum is copied back to all.  Every replica then runs the same update on the same numbers, so pdf / cdf stay
 * identical on all devices AND identical to what ONE device tracing all the paths would hold: sharding does not change a
 * bit of the result.  One replica: UpdateDists on the current stream, nothing else.
 */
int scene_update_dists_all(PimScene* scene)
{
    const int G = g_libCount;
    if (G <= 1 || !scene->peers[1]) return update_dists(scene->d, g_lib.stream);
    const int64_t n = (int64_t)scene->d.gridLen * scene->d.emissiveCount;
    if (n <= 0) return PIMGPU_OK;
    const size_t bytes = sizeof(uint32_t) * (size_t)n;
    Library& L0 = g_libs[0];
    for (int i = 1; i < G; ++i) { DeviceScope scope(i); PT_HIP(hipStreamSynchronize(g_libs[i].stream)); }
    {
        DeviceScope scope(0);
        for (int i = 1; i < G; ++i)
        {
            PT_HIP(hipMemcpyAsync(scene->liveTmp, scene->peers[i]->live, bytes, hipMemcpyDeviceToDevice, L0.stream));
            k_live_merge<<<(unsigned)((n + 255) / 256), 256, 0, L0.stream>>>(scene->live, scene->liveTmp, scene->liveBase, n);
        }
        for (int i = 1; i < G; ++i) PT_HIP(hipMemcpyAsync(scene->peers[i]->live, scene->live, bytes, hipMemcpyDeviceToDevice, L0.stream));
        PT_HIP(hipStreamSynchronize(L0.stream));
    }
    int rc = for_each_context(G, [&](int i) {
        PimScene* r = i == 0 ? scene : scene->peers[i];
        int rc2 = update_dists(r->d, g_lib.stream);
        if (!rc2 && i == 0 && hipMemcpyAsync(scene->liveBase, scene->live, bytes, hipMemcpyDeviceToDevice, g_lib.stream) != hipSuccess) rc2 = PIMGPU_E_DEVICE;
        if (!rc2 && hipStreamSynchronize(g_lib.stream) != hipSuccess) rc2 = PIMGPU_E_DEVICE;
        return rc2;
    });
    return rc;
}

} /* namespace pt */

using namespace pt;

/*
 * Diagnostics of the library's first calls (host code only).  PIMGPU_SEGV_BACKTRACE=1: a process that dies of SIGSEGV /
 * SIGBUS / SIGABRT / SIGFPE inside the library or the HIP runtime under it prints its NATIVE frames to stderr before the
 * default action runs (Python's faulthandler only names the ctypes call); PIMGPU_INIT_TRACE=1: one stderr line before every
 * HIP call of pimgpu_init_devices, so that a hang or a fault names the call it happened in.
 */
static void fault_backtrace(int sig)
{
    static const char head[] = "[pimgpu] fatal signal, native frames:\n";
    (void)!write(2, head, sizeof(head) - 1);
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

static void install_fault_backtrace()
{
    static bool done = false;
    if (done) return;
    done = true;
    const char* e = getenv("PIMGPU_SEGV_BACKTRACE");
    if (!e || !*e || *e == '0') return;
    void* warm[4]; (void)backtrace(warm, 4);          /* loads libgcc's unwinder now: not from inside the handler */
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_handler = fault_backtrace; sa.sa_flags = SA_NODEFER | SA_RESETHAND;
    sigemptyset(&sa.sa_mask);
    const int sigs[] = { SIGSEGV, SIGBUS, SIGABRT, SIGFPE };
    for (int sgn : sigs) sigaction(sgn, &sa, nullptr);
}

static bool init_trace_on() { static const bool on = [] { const char* e = getenv("PIMGPU_INIT_TRACE"); return e && *e && *e != '0'; }(); return on; }
#define INIT_TRACE(...) do { if (init_trace_on()) { fprintf(stderr, "[pimgpu init] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

extern "C" {

/* one context: stream, traversal resources, scratch address; the device is current */
static int init_context(Library& L, int device)
{
    hipDeviceProp_t prop;
    INIT_TRACE("hipGetDeviceProperties(%d)", device);
    PT_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    {
        set_error("device %d is %s; libpimgpu is built for gfx950 only", device, prop.gcnArchName);
        return PIMGPU_E_DEVICE;
    }
    INIT_TRACE("hipStreamCreateWithFlags");
    PT_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    L.device = device;
    L.cuCount = prop.multiProcessorCount;
    INIT_TRACE("traversal resources (occupancy query, spill hipMalloc)");
    int rc = L.trace.ensure();
    if (rc) return rc;
    PT_HIP(hipHostMalloc((void**)&L.smallBack, 64, hipHostMallocDefault));
    INIT_TRACE("hipGetSymbolAddress(scratch)");
    if ((rc = stddev_scratch_setup(device))) return rc;
    L.initialised = true;
    INIT_TRACE("context on device %d ready", device);
    return PIMGPU_OK;
}

int pimgpu_init_devices(const int* devices, int deviceCount)
{
    if (g_libs[0].initialised)
    {
        /* PtSys_Init twice is harmless in the reference too; a different list needs pimgpu_shutdown first */
        bool same = deviceCount == g_libCount;
        for (int i = 0; same && i < deviceCount; ++i) same = devices && devices[i] == g_libs[i].device;
        if (same) return PIMGPU_OK;
        set_error("init_devices: already initialised with another device list (pimgpu_shutdown first)");
        return PIMGPU_E_STATE;
    }
    if (!devices || deviceCount < 1 || deviceCount > kMaxDevices) { set_error("init_devices: need 1..%d devices", kMaxDevices); return PIMGPU_E_INVALID; }
    install_fault_backtrace();
    int count = 0;
    INIT_TRACE("hipGetDeviceCount");
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device visible"); return PIMGPU_E_DEVICE; }
    for (int i = 0; i < deviceCount; ++i)
        if (devices[i] < 0 || devices[i] >= count) { set_error("device %d out of range (have %d)", devices[i], count); return PIMGPU_E_INVALID; }
    int rc = PIMGPU_OK;
    for (int i = 0; i < deviceCount && !rc; ++i)
    {
        INIT_TRACE("hipSetDevice(%d)", devices[i]);
        PT_HIP(hipSetDevice(devices[i]));
        rc = init_context(g_libs[i], devices[i]);
        /* replicas read each other's light histograms and image bands over xGMI */
        for (int k = 0; k < i && !rc; ++k)
            if (devices[k] != devices[i])
            {
                INIT_TRACE("peer access %d <-> %d", devices[i], devices[k]);
                hipError_t e = hipDeviceEnablePeerAccess(devices[k], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { set_error("no peer access from device %d to %d", devices[i], devices[k]); rc = PIMGPU_E_DEVICE; }
                (void)hipGetLastError();
                hipSetDevice(devices[k]);
                e = hipDeviceEnablePeerAccess(devices[i], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled && !rc) { set_error("no peer access from device %d to %d", devices[k], devices[i]); rc = PIMGPU_E_DEVICE; }
                (void)hipGetLastError();
                hipSetDevice(devices[i]);
            }
    }
    hipSetDevice(devices[0]);
    if (rc)
    {
        for (int i = 0; i < deviceCount; ++i)
            if (g_libs[i].stream)
            {
                hipSetDevice(devices[i]);
                g_libs[i].trace.release();
                if (g_libs[i].smallBack) hipHostFree(g_libs[i].smallBack);
                hipStreamDestroy(g_libs[i].stream);
                g_libs[i] = Library();
            }
        hipSetDevice(devices[0]);
        return rc;
    }
    g_libCount = deviceCount;
    set_cur_lib(nullptr);
    return PIMGPU_OK;
}

int pimgpu_init(int device) { return pimgpu_init_devices(&device, 1); }

int pimgpu_device_count(void) { return g_libs[0].initialised ? g_libCount : 0; }

void* pimgpu_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { set_error("host_alloc: %zu bytes of page-locked memory", bytes); return nullptr; }
    /* zeroed like the Tex_Calloc it stands for; a lightmap page is 113 MB and a pack has dozens: several threads clear a large block
     * (the first touch of page-locked memory is cheap, the pages are there; one core's store rate is the limit) */
    const size_t kPart = (size_t)4 << 20;
    const char* e = getenv("PIMGPU_HOST_CLEAR_THREADS");                     /* tuning knob */
    const size_t clearThreads = (size_t)std::max(1, std::min(64, (e && *e) ? atoi(e) : 16));
    const unsigned workers = (unsigned)std::min<size_t>(std::min<size_t>(clearThreads, std::max(1u, std::thread::hardware_concurrency())), bytes / kPart);
    if (workers <= 1) memset(p, 0, bytes);
    else
    {
        std::vector<std::thread> pool;
        const size_t per = ((bytes + workers - 1) / workers + 4095) & ~(size_t)4095;
        size_t done = 0;                                   /* bytes handed to a thread so far */
        try
        {
            for (unsigned k = 0; k + 1 < workers; ++k)
            {
                const size_t lo = std::min(bytes, per * k), hi = std::min(bytes, per * (k + 1));
                if (hi > lo) { pool.emplace_back([=] { memset((char*)p + lo, 0, hi - lo); }); done = hi; }
            }
        }
        catch (...) { }                                    /* no more threads to be had: this one clears the rest */
        memset((char*)p + done, 0, bytes - done);
        for (auto& t : pool) t.join();
    }
    return p;
}

void pimgpu_host_free(void* p)
{
    if (p) hipHostFree(p);
}

int pimgpu_shutdown(void)
{
    if (!g_libs[0].initialised) return PIMGPU_OK;
    set_cur_lib(nullptr);
    for (int i = 0; i < g_libCount; ++i)
    {
        Library& L = g_libs[i];
        if (!L.initialised) continue;
        hipSetDevice(L.device);
        DeviceScope scope(i);
        hipStreamSynchronize(L.stream);
        L.release_groups();
        L.work.release();
        L.trace.release();
        if (L.smallBack) hipHostFree(L.smallBack);
        hipStreamDestroy(L.stream);
        L = Library();
    }
    g_libCount = 0;
    return PIMGPU_OK;
}

const char* pimgpu_last_error(void) { return g_error; }

void* pimgpu_stream(void) { return (void*)g_lib.stream; }

int pimgpu_work_counters(int enable, uint64_t* out)
{
    int rc = check_ready(); if (rc) return rc;
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    if (out) PT_HIP(hipMemcpy(out, g_lib.trace.work, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (getenv("PIMGPU_TRACE_DIAG"))
    {
        uint64_t w[12] = {};
        PT_HIP(hipMemcpy(w, g_lib.trace.work, sizeof(w), hipMemcpyDeviceToHost));
        if (w[2]) fprintf(stderr, "[trace diag] rays %llu  trips(wave) %llu  A-exec %llu  B-exec %llu  pop-iter(wave) %llu  pop-lanes %llu  | per 64 rays: trips %.1f A %.1f B %.1f popIter %.1f"
            " | lanes per A-exec %.1f, per B-exec %.1f | lane-trips: at a node %.3f, no ray %.3f, second leaf waiting %.3f, nothing in hand %.3f, first leaf only %.3f\n",
            (unsigned long long)w[2], (unsigned long long)w[4], (unsigned long long)w[5], (unsigned long long)w[6], (unsigned long long)w[7], (unsigned long long)w[8],
            64.0 * w[4] / w[2], 64.0 * w[5] / w[2], 64.0 * w[6] / w[2], 64.0 * w[7] / w[2],
            (double)w[0] / (double)(w[5] ? w[5] : 1), (double)w[1] / (double)(w[6] ? w[6] : 1),
            w[0] / (64.0 * w[4]), w[9] / (64.0 * w[4]), w[10] / (64.0 * w[4]), w[11] / (64.0 * w[4]),
            1.0 - (double)(w[0] + w[9] + w[10] + w[11]) / (64.0 * w[4]));
    }
    PT_HIP(hipMemset(g_lib.trace.work, 0, 12 * sizeof(uint64_t)));
    g_lib.trace.counting = enable != 0;
    return PIMGPU_OK;
}

/* one replica, in the current context */
static int scene_create_one(const PimSceneDesc* d, PimScene** out)
{
    if (out) *out = nullptr;
    int rc = check_ready(); if (rc) return rc;
    if (!d || !out) { set_error("scene_create: NULL argument"); return PIMGPU_E_INVALID; }
    if (d->vertCount < 0 || (d->vertCount % 3) != 0) { set_error("scene_create: vertCount %d is not a multiple of 3", d->vertCount); return PIMGPU_E_INVALID; }
    if (d->vertCount / 3 >= (1 << 28)) { set_error("scene_create: more than 2^28 triangles"); return PIMGPU_E_LIMIT; }
    if (d->vertCount > 0 && (!d->positions || !d->normals || !d->uvs || !d->matIds || !d->materials || d->matCount <= 0))
    { set_error("scene_create: missing vertex or material arrays"); return PIMGPU_E_INVALID; }
    if (d->texCount > 0 && !d->textures) { set_error("scene_create: texCount > 0 but textures == NULL"); return PIMGPU_E_INVALID; }
    if (!d->brdfLut || d->brdfLutSize <= 0) { set_error("scene_create: the BRDF LUT (lighting.h g_BrdfLut) is required"); return PIMGPU_E_INVALID; }
    for (int32_t i = 0; i < d->vertCount; ++i)
    {
        if (d->matIds[i] < 0 || d->matIds[i] >= d->matCount) { set_error("scene_create: matIds[%d] = %d out of range", i, d->matIds[i]); return PIMGPU_E_INVALID; }
    }
    for (int32_t i = 0; i < d->matCount; ++i)
    {
        const PimMaterial& m = d->materials[i];
        if (m.albedo >= d->texCount || m.rome >= d->texCount || m.normal >= d->texCount) { set_error("scene_create: material %d names a texture out of range", i); return PIMGPU_E_INVALID; }
    }
    for (int32_t i = 0; i < d->texCount; ++i)
    {
        if (d->textures[i].width <= 0 || d->textures[i].height <= 0 || !d->textures[i].texels) { set_error("scene_create: texture %d is empty", i); return PIMGPU_E_INVALID; }
    }

    PimScene* s = new PimScene();
    auto fail = [&](int code) { scene_free(s); return code; };
    const size_t V = (size_t)d->vertCount;
    s->d.vertCount = d->vertCount; s->d.triCount = d->vertCount / 3; s->d.matCount = d->matCount; s->d.texCount = d->texCount;
    s->seed = d->seed;
    s->metersPerCell = d->metersPerCell > 0.0f ? d->metersPerCell : 1.5f;
    if ((rc = upload(&s->positions, d->positions, V))) return fail(rc);
    if ((rc = upload(&s->normals, d->normals, V))) return fail(rc);
    if ((rc = upload(&s->uvs, d->uvs, V))) return fail(rc);
    if ((rc = upload(&s->matIds, d->matIds, V))) return fail(rc);
    if ((rc = upload(&s->materials, d->materials, (size_t)d->matCount))) return fail(rc);
    {
        std::vector<DTexture> tex((size_t)d->texCount);
        for (int32_t i = 0; i < d->texCount; ++i)
        {
            uint32_t* texels = nullptr;
            if ((rc = upload(&texels, d->textures[i].texels, (size_t)d->textures[i].width * d->textures[i].height))) return fail(rc);
            s->texels.push_back(texels);
            tex[i].width = d->textures[i].width; tex[i].height = d->textures[i].height; tex[i].format = d->textures[i].format; tex[i]._pad = 0;
            tex[i].texels = texels;
        }
        if ((rc = upload(&s->textures, tex.data(), tex.size()))) return fail(rc);
    }
    s->d.skySize = d->skySize;
    if (d->skySize > 0)
    {
        for (int f = 0; f < 6; ++f)
        {
            if (!d->skyFaces[f]) { set_error("scene_create: sky face %d is NULL", f); return fail(PIMGPU_E_INVALID); }
            if ((rc = upload(&s->sky[f], d->skyFaces[f], (size_t)d->skySize * d->skySize))) return fail(rc);
            s->d.sky[f] = s->sky[f];
        }
    }
    s->d.lutSize = d->brdfLutSize;
    if ((rc = upload(&s->lut, d->brdfLut, (size_t)d->brdfLutSize * d->brdfLutSize))) return fail(rc);
    s->d.lut = s->lut;
    media_from_desc(&s->d.media, &d->media);
    s->d.positions = s->positions; s->d.normals = s->normals; s->d.uvs = s->uvs; s->d.matIds = s->matIds;
    s->d.materials = s->materials; s->d.textures = s->textures;
    /* the shading stages' per-triangle records, before anything that classifies a hit (the light-grid set-up does) */
    if ((rc = upload(&s->tri, nullptr, (size_t)s->d.triCount))) return fail(rc);
    if ((rc = pack_trirecs(s->d, s->tri, g_lib.stream))) return fail(rc);
    s->d.tri = s->tri;

    hipEvent_t e0, e1;
    PT_HIP(hipEventCreate(&e0)); PT_HIP(hipEventCreate(&e1));
    float msEmit = 0.0f, msGrid = 0.0f;

    /* PtScene_Init order, path_tracer.c:1039-1049: SetupEmissives, RtcNewScene, SetupLightGrid */
    const int32_t triCount = s->d.triCount;
    {
        float* pdfsDev = nullptr;
        if ((rc = upload(&pdfsDev, nullptr, (size_t)triCount))) return fail(rc);
        PT_HIP(hipEventRecord(e0, g_lib.stream));
        rc = setup_emission_pdfs(s->d, d->emissiveAttempts > 0 ? d->emissiveAttempts : 1000, s->seed, pdfsDev, g_lib.stream);
        PT_HIP(hipEventRecord(e1, g_lib.stream));
        if (rc) { hipFree(pdfsDev); return fail(rc); }
        std::vector<float> pdfs((size_t)triCount);
        PT_HIP(hipMemcpyAsync(pdfs.data(), pdfsDev, sizeof(float) * (size_t)triCount, hipMemcpyDeviceToHost, g_lib.stream));
        PT_HIP(hipStreamSynchronize(g_lib.stream));
        hipEventElapsedTime(&msEmit, e0, e1);
        hipFree(pdfsDev);
        /* ordered compaction of the emissive triangles, path_tracer.c:861-880 */
        std::vector<int32_t> vertToEmit(V), emitToVert;
        for (int32_t iTri = 0; iTri < triCount; ++iTri)
        {
            const int32_t iVert = iTri * 3;
            int32_t e = -1;
            if (pdfs[(size_t)iTri] > 0.01f) { e = (int32_t)emitToVert.size(); emitToVert.push_back(iVert); }
            vertToEmit[(size_t)iVert + 0] = vertToEmit[(size_t)iVert + 1] = vertToEmit[(size_t)iVert + 2] = e;
        }
        s->d.emissiveCount = (int32_t)emitToVert.size();
        if ((rc = upload(&s->vertToEmit, vertToEmit.data(), V))) return fail(rc);
        if ((rc = upload(&s->emitToVert, emitToVert.data(), emitToVert.size()))) return fail(rc);
        s->d.vertToEmit = s->vertToEmit; s->d.emitToVert = s->emitToVert;
    }
    if ((rc = scene_build_bvh(s))) return fail(rc);
    {
        /* SetupLightGrid, path_tracer.c:992-1009; box_from_pts math/box.h:53-68 */
        if (V > 0)
        {
            v4 lo = f4_s((float)(1 << 23)), hi = f4_s(-(float)(1 << 23));
            const v4* pos = (const v4*)d->positions;
            for (size_t i = 0; i < V; ++i) { lo = f4_min(lo, pos[i]); hi = f4_max(hi, pos[i]); }
            DGrid g;
            const float cellsPerMeter = 1.0f / s->metersPerCell;
            const v4 sizef = f4_ceil(f4_muls(f4_sub(hi, lo), cellsPerMeter));
            g.lo = lo; g.hi = hi; g.sx = (int32_t)sizef.x; g.sy = (int32_t)sizef.y; g.sz = (int32_t)sizef.z; g.cellsPerMeter = cellsPerMeter;
            /* a scene flat along an axis gives Grid_New a zero extent (math/grid.h:17-25), which the reference then
             * indexes out of bounds; keep at least one cell per axis */
            g.sx = g.sx < 1 ? 1 : g.sx; g.sy = g.sy < 1 ? 1 : g.sy; g.sz = g.sz < 1 ? 1 : g.sz;
            const int64_t len = (int64_t)g.sx * g.sy * g.sz;
            if (len <= 0 || len > (1ll << 30)) { set_error("scene_create: light grid of %lld cells", (long long)len); return fail(PIMGPU_E_LIMIT); }
            s->d.grid = g; s->d.gridLen = (int32_t)len;
        }
        const size_t len = (size_t)s->d.gridLen, E = (size_t)s->d.emissiveCount;
        if ((rc = upload(&s->cellLength, nullptr, len))) return fail(rc);
        if ((rc = upload(&s->pdf, nullptr, len * E))) return fail(rc);
        if ((rc = upload(&s->cdf, nullptr, len * (E + 1)))) return fail(rc);
        if ((rc = upload(&s->live, nullptr, len * E))) return fail(rc);
        if ((rc = upload(&s->cellSum, nullptr, len))) return fail(rc);
        if ((rc = upload(&s->cellIntegral, nullptr, len))) return fail(rc);
        s->d.cellLength = s->cellLength; s->d.pdf = s->pdf; s->d.cdf = s->cdf; s->d.live = s->live;
        s->d.cellSum = s->cellSum; s->d.cellIntegral = s->cellIntegral;
        PT_HIP(hipEventRecord(e0, g_lib.stream));
        rc = setup_light_grid(s->d, &g_lib.trace, s->metersPerCell, s->seed, s->cellLength, g_lib.stream);
        PT_HIP(hipEventRecord(e1, g_lib.stream));
        if (rc) return fail(rc);
        PT_HIP(hipStreamSynchronize(g_lib.stream));
        hipEventElapsedTime(&msGrid, e0, e1);
        if ((rc = check_status_word("scene_create"))) return fail(rc);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    s->setupMs = msEmit + msGrid;
    if ((rc = scene_sync_device_copy(s))) return fail(rc);
    *out = s;
    return PIMGPU_OK;
}

int pimgpu_scene_create(const PimSceneDesc* d, PimScene** out)
{
    int rc = scene_create_one(d, out);
    if (rc || g_libCount <= 1) return rc;
    /* replicas: every context runs the same deterministic set-up on its own copy (scenes are small against 288 GB) */
    PimScene* primary = *out;
    rc = for_each_context(g_libCount, [&](int i) { return i == 0 ? PIMGPU_OK : scene_create_one(d, &primary->peers[i]); });
    const size_t n = (size_t)primary->d.gridLen * (size_t)primary->d.emissiveCount;
    if (!rc && (hipMalloc(&primary->liveBase, sizeof(uint32_t) * (n ? n : 1)) != hipSuccess || hipMalloc(&primary->liveTmp, sizeof(uint32_t) * (n ? n : 1)) != hipSuccess))
    { set_error("scene_create: out of device memory"); rc = PIMGPU_E_NOMEM; }
    if (!rc && n) { hipMemcpyAsync(primary->liveBase, primary->live, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, g_lib.stream); hipStreamSynchronize(g_lib.stream); }
    if (rc) { pimgpu_scene_destroy(primary); *out = nullptr; }
    return rc;
}

int pimgpu_scene_destroy(PimScene* scene)
{
    if (!scene) return PIMGPU_OK;
    for (int i = 1; i < kMaxDevices; ++i)
        if (scene->peers[i])
        {
            if (i < g_libCount) { DeviceScope scope(i); hipStreamSynchronize(g_lib.stream); scene_free(scene->peers[i]); }
            scene->peers[i] = nullptr;
        }
    if (g_lib.initialised) hipStreamSynchronize(g_lib.stream);
    hipFree(scene->liveBase); hipFree(scene->liveTmp);
    scene_free(scene);
    return PIMGPU_OK;
}

int pimgpu_scene_rebuild_bvh(PimScene* scene)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene) { set_error("scene_rebuild_bvh: NULL scene"); return PIMGPU_E_INVALID; }
    if (g_libCount > 1 && scene->peers[1]) return for_each_context(g_libCount, [&](int i) { return scene_build_bvh(i == 0 ? scene : scene->peers[i]); });
    return scene_build_bvh(scene);
}

int pimgpu_scene_update_dists(PimScene* scene)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene) { set_error("scene_update_dists: NULL scene"); return PIMGPU_E_INVALID; }
    if ((rc = scene_update_dists_all(scene))) return rc;
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    return PIMGPU_OK;
}

/* the light histograms (LightOnHit's feedback, path_tracer.c:1709-1733) to and from caller-owned HBM: what one process per GPU
 * exchanges before UpdateDists -- u32[gridLen * emissiveCount], device-to-device on the library's stream */
int pimgpu_scene_live_get_dev(const PimScene* s, void* dstDev)
{
    int rc = check_ready(); if (rc) return rc;
    if (!s || !dstDev) { set_error("scene_live_get_dev: NULL argument"); return PIMGPU_E_INVALID; }
    if (g_libCount > 1 && s->peers[1]) { set_error("scene_live_get_dev: a device list sums its replicas' histograms itself"); return PIMGPU_E_STATE; }
    const size_t bytes = sizeof(uint32_t) * (size_t)s->d.gridLen * (size_t)s->d.emissiveCount;
    if (bytes) PT_HIP(hipMemcpyAsync(dstDev, s->live, bytes, hipMemcpyDeviceToDevice, g_lib.stream));
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    return PIMGPU_OK;
}

int pimgpu_scene_live_set_dev(PimScene* s, const void* srcDev)
{
    int rc = check_ready(); if (rc) return rc;
    if (!s || !srcDev) { set_error("scene_live_set_dev: NULL argument"); return PIMGPU_E_INVALID; }
    if (g_libCount > 1 && s->peers[1]) { set_error("scene_live_set_dev: a device list sums its replicas' histograms itself"); return PIMGPU_E_STATE; }
    const size_t bytes = sizeof(uint32_t) * (size_t)s->d.gridLen * (size_t)s->d.emissiveCount;
    if (bytes) PT_HIP(hipMemcpyAsync(s->live, srcDev, bytes, hipMemcpyDeviceToDevice, g_lib.stream));
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    return PIMGPU_OK;
}

int pimgpu_scene_info(const PimScene* s, PimSceneInfo* out)
{
    if (!s || !out) { set_error("scene_info: NULL argument"); return PIMGPU_E_INVALID; }
    memset(out, 0, sizeof(*out));
    out->vertCount = s->d.vertCount; out->triCount = s->d.triCount; out->matCount = s->d.matCount;
    out->emissiveCount = s->d.emissiveCount;
    out->gridSize[0] = s->d.grid.sx; out->gridSize[1] = s->d.grid.sy; out->gridSize[2] = s->d.grid.sz;
    out->gridLen = s->d.gridLen;
    if (s->d.gridLen > 0)
    {
        std::vector<int32_t> len((size_t)s->d.gridLen);
        PT_HIP(hipMemcpy(len.data(), s->cellLength, sizeof(int32_t) * len.size(), hipMemcpyDeviceToHost));
        for (int32_t v : len) out->activeCells += v ? 1 : 0;
    }
    out->bvhNodeCount = s->bvh.nodeCount; out->bvhMaxDepth = s->bvhDepth;
    out->boundsLo[0] = s->d.grid.lo.x; out->boundsLo[1] = s->d.grid.lo.y; out->boundsLo[2] = s->d.grid.lo.z;
    out->boundsHi[0] = s->d.grid.hi.x; out->boundsHi[1] = s->d.grid.hi.y; out->boundsHi[2] = s->d.grid.hi.z;
    out->bvhBuildMs = s->bvhBuildMs; out->sceneSetupMs = s->setupMs;
    return PIMGPU_OK;
}

int pimgpu_scene_download_lights(const PimScene* s, int32_t* vertToEmit, int32_t* emitToVert, int32_t* cellLength,
    float* pdf, float* cdf, uint32_t* live)
{
    int rc = check_ready(); if (rc) return rc;
    if (!s) { set_error("scene_download_lights: NULL scene"); return PIMGPU_E_INVALID; }
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    const size_t E = (size_t)s->d.emissiveCount, L = (size_t)s->d.gridLen, V = (size_t)s->d.vertCount;
    if (vertToEmit && V) PT_HIP(hipMemcpy(vertToEmit, s->vertToEmit, 4 * V, hipMemcpyDeviceToHost));
    if (emitToVert && E) PT_HIP(hipMemcpy(emitToVert, s->emitToVert, 4 * E, hipMemcpyDeviceToHost));
    if (cellLength && L) PT_HIP(hipMemcpy(cellLength, s->cellLength, 4 * L, hipMemcpyDeviceToHost));
    if (pdf && L * E) PT_HIP(hipMemcpy(pdf, s->pdf, 4 * L * E, hipMemcpyDeviceToHost));
    if (cdf && L) PT_HIP(hipMemcpy(cdf, s->cdf, 4 * L * (E + 1), hipMemcpyDeviceToHost));
    if (live && L * E) PT_HIP(hipMemcpy(live, s->live, 4 * L * E, hipMemcpyDeviceToHost));
    if (live && L * E && g_libCount > 1 && s->peers[1])
    {
        /* the histogram as ONE device would hold it: replica 0's plus the increments the other replicas made since the last UpdateDists */
        std::vector<uint32_t> other(L * E), base(L * E);
        PT_HIP(hipMemcpy(base.data(), s->liveBase, 4 * L * E, hipMemcpyDeviceToHost));
        for (int i = 1; i < g_libCount; ++i)
        {
            { DeviceScope scope(i); PT_HIP(hipStreamSynchronize(g_lib.stream)); }
            PT_HIP(hipMemcpy(other.data(), s->peers[i]->live, 4 * L * E, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < L * E; ++k) live[k] += other[k] - base[k];
        }
    }
    return PIMGPU_OK;
}

/* ---- rays ---------------------------------------------------------------- */

int pimgpu_intersect_dev(const PimScene* scene, const void* roDev, const void* rdDev, const void* tNearDev, const void* tFarDev,
    int32_t n, void* hitsDev)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || n < 0 || (n > 0 && (!roDev || !rdDev || !hitsDev))) { set_error("intersect_dev: bad argument"); return PIMGPU_E_INVALID; }
    if (n == 0) return PIMGPU_OK;
    if ((rc = g_lib.work.ensure_rays(n))) return rc;
    /* pack (ro.xyz, tNear) / (rd.xyz, tFar); NULL tNear/tFar mean 0 and kRcpEpsilon */
    if ((rc = pack_rays((const float4*)roDev, (const float4*)rdDev, (const float*)tNearDev, (const float*)tFarDev, n,
        g_lib.work.rayO, g_lib.work.rayD, g_lib.stream))) return rc;
    if ((rc = trace_closest(scene->d.bvh, &g_lib.trace, g_lib.work.rayO, g_lib.work.rayD, (DHit*)hitsDev, n, nullptr, g_lib.stream))) return rc;
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    return check_status_word("intersect");
}

int pimgpu_intersect_packed_dev(const PimScene* scene, const void* rayODev, const void* rayDDev, int32_t n, void* hitsDev)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || n < 0 || (n > 0 && (!rayODev || !rayDDev || !hitsDev))) { set_error("intersect_packed_dev: bad argument"); return PIMGPU_E_INVALID; }
    if (n == 0) return PIMGPU_OK;
    if ((rc = trace_closest(scene->d.bvh, &g_lib.trace, (const float4*)rayODev, (const float4*)rayDDev, (DHit*)hitsDev, n, nullptr, g_lib.stream))) return rc;
    PT_HIP(hipStreamSynchronize(g_lib.stream));
    return check_status_word("intersect");
}

int pimgpu_intersect(const PimScene* scene, const float* ro, const float* rd, const float* tNear, const float* tFar,
    int32_t n, PimRayHit* out)
{
    int rc = check_ready(); if (rc) return rc;
    if (!scene || n < 0 || (n > 0 && (!ro || !rd || !out))) { set_error("intersect: bad argument"); return PIMGPU_E_INVALID; }
    if (n == 0) return PIMGPU_OK;
    float4 *roDev = nullptr, *rdDev = nullptr; float *tnDev = nullptr, *tfDev = nullptr; DHit* hitsDev = nullptr; PimRayHit* outDev = nullptr;
    auto cleanup = [&]() { hipFree(roDev); hipFree(rdDev); hipFree(tnDev); hipFree(tfDev); hipFree(hitsDev); hipFree(outDev); };
    if ((rc = upload(&roDev, ro, (size_t)n)) || (rc = upload(&rdDev, rd, (size_t)n)) ||
        (tNear && (rc = upload(&tnDev, tNear, (size_t)n))) || (tFar && (rc = upload(&tfDev, tFar, (size_t)n))) ||
        (rc = upload(&hitsDev, nullptr, (size_t)n)) || (rc = upload(&outDev, nullptr, (size_t)n)))
    { cleanup(); return rc; }
    rc = pimgpu_intersect_dev(scene, roDev, rdDev, tnDev, tfDev, n, hitsDev);
    if (!rc) rc = make_rayhits(scene->d, rdDev, hitsDev, outDev, n, g_lib.stream);
    if (!rc && hipMemcpyAsync(out, outDev, sizeof(PimRayHit) * (size_t)n, hipMemcpyDeviceToHost, g_lib.stream) != hipSuccess) { set_error("intersect: copy back failed"); rc = PIMGPU_E_DEVICE; }
    hipStreamSynchronize(g_lib.stream);
    cleanup();
    return rc;
}

} /* extern "C" */
