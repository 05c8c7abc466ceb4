// Issue cost of short VALU sequences as the compiler emits them in the traversal kernels' node step (wave64, gfx950): compare +
// selects through VCC or an SGPR pair, with and without the wait states the hazard rules ask for.  8 waves per SIMD, no memory.
// Values are meaningless (hazards are not honoured inside the blocks on purpose); only the time is read.
// Build: hipcc --offload-arch=gfx950 -O3 valu_seq.hip -o valu_seq
#include <hip/hip_runtime.h>
#include <cstdio>

#define SEQS(X) \
    X(0, "4x cndmask vcc",                          4, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc") \
    X(1, "4x cndmask s[20:21]",                     4, "v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]") \
    X(2, "cmp vcc, s_nop 1, 4x cndmask vcc",        5, "v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc") \
    X(3, "cmp vcc, 4x cndmask vcc (no nop)",        5, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc") \
    X(4, "cmp s[20:21], s_nop 1, 4x cndmask sgpr",  5, "v_cmp_lt_f32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]") \
    X(5, "cmp s[20:21], 4x cndmask sgpr (no nop)",  5, "v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]") \
    X(6, "4x (mov, s_nop 0)",                       4, "v_mov_b32 %0, %4\n s_nop 0\n v_mov_b32 %1, %4\n s_nop 0\n v_mov_b32 %2, %4\n s_nop 0\n v_mov_b32 %3, %4\n s_nop 0") \
    X(7, "4x mov, one s_nop 1",                     4, "v_mov_b32 %0, %4\n s_nop 1\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4") \
    X(8, "2x (min_f32, max_f32) = 2 key swaps",     4, "v_min_f32 %0, %0, %1\n v_max_f32 %1, %0, %1\n v_min_f32 %2, %2, %3\n v_max_f32 %3, %2, %3") \
    X(9, "4x (cmp vcc, cndmask vcc) dependent",     8, "v_cmp_le_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %4, %0, vcc\n v_cmp_le_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %1, %4, %1, vcc\n v_cmp_le_f32 vcc, %2, %3\n s_nop 1\n v_cndmask_b32 %2, %4, %2, vcc\n v_cmp_le_f32 vcc, %3, %0\n s_nop 1\n v_cndmask_b32 %3, %4, %3, vcc") \
    X(10, "4x (cmp vcc, cndmask vcc) no nop",       8, "v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %4, %0, vcc\n v_cmp_le_f32 vcc, %1, %2\n v_cndmask_b32 %1, %4, %1, vcc\n v_cmp_le_f32 vcc, %2, %3\n v_cndmask_b32 %2, %4, %2, vcc\n v_cmp_le_f32 vcc, %3, %0\n v_cndmask_b32 %3, %4, %3, vcc") \
    X(11, "4x fma (reference)",                     4, "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4") \
    X(12, "2 fma + 2 max interleaved",              4, "v_fma_f32 %0, %0, %4, %4\n v_max_f32 %1, %1, %4\n v_fma_f32 %2, %2, %4, %4\n v_max_f32 %3, %3, %4") \
    X(13, "2 fma + 2 cvt_ubyte interleaved",        4, "v_fma_f32 %0, %0, %4, %4\n v_cvt_f32_ubyte1 %1, %1\n v_fma_f32 %2, %2, %4, %4\n v_cvt_f32_ubyte2 %3, %3") \
    X(14, "cmp vcc + s_nop 1 + cndmask vcc x2",     3, "v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %2, %0, %1, vcc\n v_cndmask_b32 %3, %1, %0, vcc") \
    X(15, "max3 + min3 + cmp_le vcc + cndmask vcc", 4, "v_max3_f32 %0, %0, %1, %4\n v_min3_f32 %1, %2, %3, %4\n v_cmp_le_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %2, %4, %0, vcc")

template <int KIND>
__global__ __launch_bounds__(256) void k_seq(float* out, int iters, float a)
{
    float r[16];
    for (int k = 0; k < 16; ++k) r[k] = (float)(threadIdx.x + k) * 0.001f + 1.0f;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int k = 0; k < 16; k += 4)
        {
#define X(ID, NAME, N, TEXT) if (KIND == ID) asm volatile(TEXT : "+v"(r[k]), "+v"(r[k + 1]), "+v"(r[k + 2]), "+v"(r[k + 3]) : "v"(a) : "vcc", "s20", "s21");
            SEQS(X)
#undef X
        }
    }
    float s = 0.0f;
    for (int k = 0; k < 16; ++k) s += r[k];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
static void run(const char* name, int perBlock, int cus, float* out)
{
    const int blocks = cus * 8, iters = 40000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_seq<KIND><<<blocks, 256>>>(out, 1000, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_seq<KIND><<<blocks, 256>>>(out, iters, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double seqs = (double)blocks * 4 * (double)iters * 4;
    printf("%-44s %d VALU  %8.3f ms  %6.2f cycles per sequence per SIMD at 2.4 GHz = %5.2f per VALU instruction\n", name, perBlock, ms,
        (double)cus * 4 * 2.4e9 * (ms * 1e-3) / seqs, (double)cus * 4 * 2.4e9 * (ms * 1e-3) / seqs / perBlock);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float* out; (void)hipMalloc(&out, 256);
    printf("%s, %d CUs, 8 waves per SIMD\n", p.gcnArchName, p.multiProcessorCount);
#define X(ID, NAME, N, TEXT) run<ID>(NAME, N, p.multiProcessorCount, out);
    SEQS(X)
#undef X
    return 0;
}
