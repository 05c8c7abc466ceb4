// VALU issue peak of the MI355X as this repository's rooflines use it: wave64 fp32 instructions per second with every SIMD
// holding W waves that issue independent v_fma_f32 / v_cndmask / v_max3 streams (no memory).  Build: hipcc --offload-arch=gfx950 -O3
// Prints wave-instructions per second and the cycles per instruction per SIMD at the clock the run held.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, float a, float b)
{
    float r[16];
    for (int k = 0; k < 16; ++k) r[k] = (float)(threadIdx.x + k) * 0.001f;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int k = 0; k < 16; ++k)
        {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
            else if (KIND == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
            else asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(r[k]) : "v"(r[(k + 1) & 15]));
        }
    }
    float s = 0.0f;
    for (int k = 0; k < 16; ++k) s += r[k];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
static void run(const char* name, int wavesPerSimd, int cus, float* out)
{
    const int blocks = cus * wavesPerSimd;          // 256 threads = 4 waves = one per SIMD of a CU
    const int iters = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_valu<KIND><<<blocks, 256>>>(out, 1000, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_valu<KIND><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * (double)iters * 16;
    const double rate = insts / (ms * 1e-3);
    printf("%-18s waves/SIMD %d  %8.3f ms  %8.1f G wave-instr/s  = %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, wavesPerSimd, ms, rate * 1e-9,
        (double)cus * 4 * 2.4e9 / rate);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    float* out; hipMalloc(&out, 256);
    printf("%s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int w : { 1, 2, 4, 6, 8 })
    {
        run<0>("v_fma_f32", w, p.multiProcessorCount, out);
        run<1>("v_max3_f32", w, p.multiProcessorCount, out);
        run<2>("v_cvt_f32_ubyte0", w, p.multiProcessorCount, out);
    }
    return 0;
}
