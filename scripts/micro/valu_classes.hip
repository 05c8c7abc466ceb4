// Issue cost of the VALU instruction classes the traversal kernels are made of (wave64, gfx950): every SIMD holds W waves that
// issue 16 independent chains of ONE opcode, no memory.  Prints wave-instructions per second and cycles per instruction per SIMD
// at 2.4 GHz.  Build: hipcc --offload-arch=gfx950 -O3 valu_classes.hip -o valu_classes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define KINDS(X) \
    X(0,  "v_fma_f32",          "v_fma_f32 %0, %0, %2, %3",              1) \
    X(1,  "v_mul_f32",          "v_mul_f32 %0, %0, %2",                  1) \
    X(2,  "v_add_f32",          "v_add_f32 %0, %0, %2",                  1) \
    X(3,  "v_max_f32",          "v_max_f32 %0, %0, %2",                  1) \
    X(4,  "v_max3_f32",         "v_max3_f32 %0, %0, %2, %3",             1) \
    X(5,  "v_min3_f32",         "v_min3_f32 %0, %0, %2, %3",             1) \
    X(6,  "v_cvt_f32_ubyte1",   "v_cvt_f32_ubyte1 %0, %1",               1) \
    X(7,  "v_cvt_f32_u32",      "v_cvt_f32_u32 %0, %1",                  1) \
    X(8,  "v_perm_b32",         "v_perm_b32 %0, %1, %2, %3",             1) \
    X(9,  "v_cndmask_b32",      "v_cndmask_b32 %0, %1, %2, vcc",         1) \
    X(10, "v_cmp_lt_f32",       "v_cmp_lt_f32 vcc, %1, %2",              1) \
    X(11, "v_bfe_u32",          "v_bfe_u32 %0, %1, 8, 8",                1) \
    X(12, "v_and_b32",          "v_and_b32 %0, %1, %2",                  1) \
    X(13, "v_lshl_or_b32",      "v_lshl_or_b32 %0, %1, 3, %2",           1) \
    X(14, "v_add_u32",          "v_add_u32 %0, %1, %2",                  1) \
    X(15, "v_mov_b32",          "v_mov_b32 %0, %1",                      1) \
    X(16, "v_pk_fma_f32",       "v_pk_fma_f32 %4, %4, %5, %5",           2) \
    X(17, "v_pk_mul_f32",       "v_pk_mul_f32 %4, %4, %5",               2) \
    X(18, "v_and_or_b32",       "v_and_or_b32 %0, %1, %2, %3",           1) \
    X(19, "v_min_u32",          "v_min_u32 %0, %1, %2",                  1) \
    X(20, "v_med3_f32",         "v_med3_f32 %0, %0, %2, %3",             1) \
    X(21, "v_fma_f64",          "v_fma_f64 %4, %4, %5, %5",              1) \
    X(22, "v_mul_f64",          "v_mul_f64 %4, %4, %5",                  1) \
    X(23, "v_mad_u32_u24",      "v_mad_u32_u24 %0, %1, %2, %3",          1) \
    X(24, "v_sub_f32",          "v_sub_f32 %0, %0, %2",                  1) \
    X(25, "v_cmp_le_f32_sgpr",  "v_cmp_le_f32 s[20:21], %1, %2",         1) \
    X(26, "v_cndmask_b32_sgpr", "v_cndmask_b32 %0, %1, %2, s[20:21]",    1) \
    X(27, "v_pk_add_f32",       "v_pk_add_f32 %4, %4, %5",               2) \
    X(28, "v_cvt_f32_ubyte0_sdwa", "v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2", 1) \
    X(29, "v_mul_f32_dpp",      "v_mul_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 1) \
    X(30, "v_fmac_f32",         "v_fmac_f32 %0, %1, %2",                 1) \
    X(31, "v_min_f32",          "v_min_f32 %0, %0, %2",                  1) \
    X(32, "v_or_b32",           "v_or_b32 %0, %1, %2",                   1) \
    X(33, "v_xor_b32",          "v_xor_b32 %0, %1, %2",                  1) \
    X(34, "v_not_b32",          "v_not_b32 %0, %1",                      1) \
    X(35, "v_lshlrev_b32",      "v_lshlrev_b32 %0, 3, %1",               1) \
    X(36, "v_lshrrev_b32",      "v_lshrrev_b32 %0, 3, %1",               1) \
    X(37, "v_ashrrev_i32",      "v_ashrrev_i32 %0, 3, %1",               1) \
    X(38, "v_sub_u32",          "v_sub_u32 %0, %1, %2",                  1) \
    X(39, "v_add3_u32",         "v_add3_u32 %0, %1, %2, %3",             1) \
    X(40, "v_lshl_add_u32",     "v_lshl_add_u32 %0, %1, 3, %2",          1) \
    X(41, "v_xad_u32",          "v_xad_u32 %0, %1, %2, %3",              1) \
    X(42, "v_bfi_b32",          "v_bfi_b32 %0, %1, %2, %3",              1) \
    X(43, "v_mul_lo_u32",       "v_mul_lo_u32 %0, %1, %2",               1) \
    X(44, "v_mul_hi_u32",       "v_mul_hi_u32 %0, %1, %2",               1) \
    X(45, "v_mul_u32_u24",      "v_mul_u32_u24 %0, %1, %2",              1) \
    X(46, "v_mad_u64_u32",      "v_mad_u64_u32 %4, s[20:21], %1, %2, %5", 1) \
    X(47, "v_lshl_add_u64",     "v_lshl_add_u64 %4, %5, 3, %4",          1) \
    X(48, "v_mov_b64",          "v_mov_b64 %4, %5",                      1) \
    X(49, "v_pk_mov_b32",       "v_pk_mov_b32 %4, %5, %4",               1) \
    X(50, "v_rcp_f32",          "v_rcp_f32 %0, %1",                      1) \
    X(51, "v_rsq_f32",          "v_rsq_f32 %0, %1",                      1) \
    X(52, "v_sqrt_f32",         "v_sqrt_f32 %0, %1",                     1) \
    X(53, "v_div_scale_f32",    "v_div_scale_f32 %0, vcc, %1, %2, %3",   1) \
    X(54, "v_div_fmas_f32",     "v_div_fmas_f32 %0, %1, %2, %3",         1) \
    X(55, "v_div_fixup_f32",    "v_div_fixup_f32 %0, %1, %2, %3",        1) \
    X(56, "v_cvt_f32_i32",      "v_cvt_f32_i32 %0, %1",                  1) \
    X(57, "v_cvt_i32_f32",      "v_cvt_i32_f32 %0, %1",                  1) \
    X(58, "v_cvt_f64_f32",      "v_cvt_f64_f32 %4, %1",                  1) \
    X(59, "v_cvt_f32_f64",      "v_cvt_f32_f64 %0, %5",                  1) \
    X(60, "v_add_f64",          "v_add_f64 %4, %4, %5",                  1) \
    X(61, "v_min_i32",          "v_min_i32 %0, %1, %2",                  1) \
    X(62, "v_readlane_b32",     "v_readlane_b32 s20, %1, 5",             1) \
    X(63, "v_readfirstlane",    "v_readfirstlane_b32 s20, %1",           1) \
    X(64, "v_writelane_b32",    "v_writelane_b32 %0, s4, 5",             1) \
    X(65, "v_mbcnt_lo",         "v_mbcnt_lo_u32_b32 %0, %1, %2",         1) \
    X(66, "v_addc_co_u32",      "v_addc_co_u32 %0, vcc, %1, %2, vcc",    1) \
    X(67, "v_add_co_u32",       "v_add_co_u32 %0, vcc, %1, %2",          1) \
    X(68, "v_cmp_class_f32",    "v_cmp_class_f32 vcc, %1, %2",           1) \
    X(69, "v_fmamk_f32",        "v_fmamk_f32 %0, %1, 0x40490fdb, %2",    1) \
    X(70, "v_mul_f32_literal",  "v_mul_f32 %0, 0x40490fdb, %1",          1) \
    X(71, "v_bcnt_u32_b32",     "v_bcnt_u32_b32 %0, %1, %2",             1) \
    X(72, "v_trunc_f32",        "v_trunc_f32 %0, %1",                    1) \
    X(73, "v_exp_f32",          "v_exp_f32 %0, %1",                      1) \
    X(74, "v_log_f32",          "v_log_f32 %0, %1",                      1) \
    X(75, "v_sin_f32",          "v_sin_f32 %0, %1",                      1) \
    X(76, "v_ldexp_f32",        "v_ldexp_f32 %0, %1, %2",                1) \
    X(77, "v_cvt_f32_f16",      "v_cvt_f32_f16 %0, %1",                  1) \
    X(78, "v_pk_fma_f16",       "v_pk_fma_f16 %0, %1, %2, %3",           1) \
    X(79, "v_pk_max_f16",       "v_pk_max_f16 %0, %1, %2",               1) \
    X(80, "v_dot2_f32_f16",     "v_dot2_f32_f16 %0, %1, %2, %0",         1) \
    X(81, "v_sad_u32",          "v_sad_u32 %0, %1, %2, %3",              1) \
    X(82, "v_max_u32",          "v_max_u32 %0, %1, %2",                  1) \
    X(83, "v_cndmask_b32_vcc",  "v_cndmask_b32 %0, %1, %2, vcc",         1)

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, float a, float b)
{
    float r[16];
    double d2[8];
    for (int k = 0; k < 16; ++k) r[k] = (float)(threadIdx.x + k) * 0.001f + 1.0f;
    for (int k = 0; k < 8; ++k) d2[k] = (double)(threadIdx.x + k) * 0.001 + 1.0;
    double pb = (double)a;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int k = 0; k < 16; ++k)
        {
#define X(ID, NAME, TEXT, W) if (KIND == ID) asm volatile(TEXT : "+v"(r[k]) : "v"(r[(k + 1) & 15]), "v"(a), "v"(b), "v"(d2[k & 7]), "v"(pb) : "vcc", "s20", "s21");
            KINDS(X)
#undef X
        }
    }
    float s = 0.0f;
    for (int k = 0; k < 16; ++k) s += r[k];
    for (int k = 0; k < 8; ++k) s += (float)d2[k];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
static void run(const char* name, int wavesPerSimd, int cus, float* out)
{
    const int blocks = cus * wavesPerSimd;          // 256 threads = 4 waves = one per SIMD of a CU
    const int iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_valu<KIND><<<blocks, 256>>>(out, 1000, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_valu<KIND><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * (double)iters * 16;
    const double rate = insts / (ms * 1e-3);
    printf("%-22s waves/SIMD %d  %8.3f ms  %8.1f G wave-instr/s  = %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, wavesPerSimd, ms, rate * 1e-9,
        (double)cus * 4 * 2.4e9 / rate);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    float* out; hipMalloc(&out, 256);
    printf("%s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int w : { 8 })
    {
#define X(ID, NAME, TEXT, W) run<ID>(NAME, w, p.multiProcessorCount, out);
        KINDS(X)
#undef X
    }
    return 0;
}
