/*
 * Cubemap_Convolve (rendering/cubemap.c:192-303): the GGX prefilter of a baked cubemap into its mip chain -- what the rasteriser's
 * image-based lighting reads (Cubemap_ReadConvolved, cubemap.h:102-115) and the step render_system.c:239-241 runs right after
 * Cubemap_Bake.  One thread per texel of the chain (every mip in ONE launch: 6 * size^2 * 4/3 texels), its `sampleCount` samples in
 * order -- the two sums of PrefilterEnvMap are sequential fp32 adds under a data-dependent condition, and their order is the result.
 * The colour faces are small (64^2 texels by default) and every sample is a bilinear read of them: they live in L2.
 * D1 as everywhere: the reference draws from its worker thread's generator (cubemap.c:247); texel i of mip m owns the stream
 * Pcg4(m * 6 * size^2 + i, sampleIndex, seed, PT_TAG_CONV).
 */
#include <hip/hip_runtime.h>
#include "pt_api.h"
#include "pt_math.h"

namespace pt {

#define PT_TAG_CONV 0x434F4E56u

struct ConvArgs
{
    const v3* color[6];          /* float3[size * size] */
    v4* convolved[6];            /* float4[CubemapElemCount(size)] */
    int32_t size, total;
    uint32_t sampleCount;
    float weight;
    uint32_t sampleIndex, seed;
};

__global__ __launch_bounds__(256) void k_cubemap_convolve(ConvArgs a)
{
    const int32_t g = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (g >= a.total) return;
    /* which mip: mip m holds 6 * (size >> m)^2 texels (cubemap.c:279-293; the last one has a single texel per face) */
    int32_t mip = 0, base = 0, msize = a.size;
    for (;;)
    {
        const int32_t len = 6 * msize * msize;
        if (g < base + len) break;
        base += len; ++mip; msize = a.size >> mip;
    }
    const int32_t i = g - base, flen = msize * msize;
    const int32_t face = i / flen, fi = i % flen;
    const int32_t cx = fi % msize, cy = fi / msize;
    u4 rng = pcg4((uint32_t)(mip * 6 * a.size * a.size + i), a.sampleIndex, a.seed, PT_TAG_CONV);

    /* PrefilterFn, cubemap.c:235-262.  A 1 x 1 mip: LinearInvert (sampler.h:192-195) divides by size - 1 = 0, the direction is NaN,
     * no sample passes NoL > 0 and the texel is blended towards zero -- as the reference's arithmetic gives it */
    const float roughness = MipToRoughness((float)mip);
    const v2 Xi = f2_tent(rng_f2(&rng));
    const v4 N = Cubemap_CalcDir(msize, face, cx, cy, Xi);

    /* PrefilterEnvMap, cubemap.c:192-223: split sum, N = V */
    const v4 I = f4_neg(N);
    const float alpha = BrdfAlpha(roughness);
    float basisIntegral = 0.0f;
    v4 lightIntegral = f4_s(0.0f);
    for (uint32_t k = 0; k < a.sampleCount; ++k)
    {
        const v4 H = TanToWorld(N, SampleGGXMicrofacet(rng_f2(&rng), alpha));
        const v4 L = f4_reflect3(I, H);
        const float NoL = f4_dot3(L, N);
        if (NoL > 0.0f)
        {
            const v3 c = Cubemap_ReadColor(a.color, a.size, L);
            v4 sample = f4_v(c.x, c.y, c.z, 1.0f);
            sample = f4_muls(sample, NoL);
            lightIntegral = f4_add(lightIntegral, sample);
            basisIntegral += NoL;
        }
    }
    if (basisIntegral > 0.0f) lightIntegral = f4_divs(lightIntegral, basisIntegral);

    /* Cubemap_BlendMip, cubemap.h:169-188 */
    v4* dst = a.convolved[face] + CalcMipOffset(a.size, mip) + msize * cy + cx;
    *dst = f4_lerps(*dst, lightIntegral, a.weight);
}

} /* namespace pt */

using namespace pt;

extern "C" {

int32_t pimgpu_cubemap_elem_count(int32_t size) { return (size > 0 && size <= 8192) ? CubemapElemCount(size) : 0; }

int pimgpu_cubemap_convolve(const float* const faces[6], float* const convolved[6], int32_t size, uint32_t sampleCount, float weight,
    uint32_t sampleIndex, uint32_t seed)
{
    int rc = check_ready(); if (rc) return rc;
    if (!faces || !convolved || size < 0 || size > 8192) { set_error("cubemap_convolve: bad argument"); return PIMGPU_E_INVALID; }
    if (size == 0) return PIMGPU_OK;                           /* mipCount 0: no task is submitted (cubemap.c:279-293) */
    for (int f = 0; f < 6; ++f) if (!faces[f] || !convolved[f]) { set_error("cubemap_convolve: NULL face"); return PIMGPU_E_INVALID; }
    const int32_t flen = size * size, elems = CubemapElemCount(size);
    int64_t total = 0;
    for (int32_t m = 0; m < CalcMipCount(size); ++m) { const int64_t ms = size >> m; total += 6 * ms * ms; }
    hipStream_t st = g_lib.stream;
    float* colorDev = nullptr; float* convDev = nullptr;
    if (hipMalloc(&colorDev, sizeof(float) * 3 * 6 * (size_t)flen) != hipSuccess || hipMalloc(&convDev, sizeof(float) * 4 * 6 * (size_t)elems) != hipSuccess)
    { hipFree(colorDev); set_error("cubemap_convolve: out of device memory"); return PIMGPU_E_NOMEM; }
    ConvArgs a;
    for (int f = 0; f < 6; ++f)
    {
        a.color[f] = (const v3*)(colorDev + (size_t)f * flen * 3);
        a.convolved[f] = (v4*)(convDev + (size_t)f * elems * 4);
        hipMemcpyAsync((void*)a.color[f], faces[f], sizeof(float) * 3 * (size_t)flen, hipMemcpyHostToDevice, st);
        hipMemcpyAsync((void*)a.convolved[f], convolved[f], sizeof(float) * 4 * (size_t)elems, hipMemcpyHostToDevice, st);
    }
    a.size = size; a.total = (int32_t)total; a.sampleCount = sampleCount; a.weight = weight; a.sampleIndex = sampleIndex; a.seed = seed;
    k_cubemap_convolve<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a);
    for (int f = 0; f < 6; ++f) hipMemcpyAsync(convolved[f], a.convolved[f], sizeof(float) * 4 * (size_t)elems, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { set_error("cubemap_convolve: device error"); rc = PIMGPU_E_DEVICE; }
    hipFree(colorDev); hipFree(convDev);
    return rc;
}

}
