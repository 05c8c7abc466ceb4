// What the memory system gives a traversal-like access pattern on gfx950, and what FETCH_SIZE says about it: every lane reads
// whole records at random places of a table far larger than the 256 MB Infinity Cache (default 4 GiB), several independent
// records in flight per lane.  Record sizes: 64 B at 64-B alignment (a compressed BVH4 node), 128 B at 128-B alignment,
// 48 B at 48-B multiples (a triangle), 2 x 64 B = one 128-B line read as two nodes -- and a coalesced 16 B/lane stream as the
// calibration the MI355X guide gives for FETCH_SIZE (x2).  Prints bytes asked for and time per kernel; run it under
// `rocprofv3 --pmc FETCH_SIZE` (and `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`) to put the counters beside the known bytes.
// Build: hipcc --offload-arch=gfx950 -O3 gather.hip -o gather      usage: ./gather [table MiB = 4096] [lanes log2 = 24]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

constexpr int kInFlight = 8;       // independent records per lane per round
constexpr int kRounds = 4;

// REC = bytes per record (multiple of 16), STRIDE = distance between record slots
template <int REC, int STRIDE>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ table, uint64_t slots, uint32_t* __restrict__ out)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < kRounds; ++r)
    {
        uint4 v[kInFlight][REC / 16];
#pragma unroll
        for (int k = 0; k < kInFlight; ++k)
        {
            const uint64_t h = ((uint64_t)hash32(lane * 977u + r * 131u + k) << 16) ^ hash32(lane + 0x9e3779b9u * (uint32_t)(r * kInFlight + k + 1));
            const uint64_t slot = h % slots;
            const uint4* p = (const uint4*)((const char*)table + slot * STRIDE);
#pragma unroll
            for (int q = 0; q < REC / 16; ++q) v[k][q] = p[q];
        }
#pragma unroll
        for (int k = 0; k < kInFlight; ++k)
#pragma unroll
            for (int q = 0; q < REC / 16; ++q) acc += v[k][q].x ^ v[k][q].y ^ v[k][q].z ^ v[k][q].w;
    }
    if (acc == 0x12345678u) out[lane] = acc;      // keeps the loads alive; (almost) never true
}

// the two 64-byte halves of a 128-byte line one after the other: the second load is issued when the first has come back, from
// few enough lanes (one record in flight each) that the lines of all of them fit the L2s many times over.
// One fabric request per record = the miss filled the whole line; two = the L2 fills 64-byte sectors.
template <int SECOND_OFFSET>
__global__ __launch_bounds__(256) void k_halves(const uint4* __restrict__ table, uint64_t slots, uint32_t* __restrict__ out, int rounds)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r)
    {
        const uint64_t h = ((uint64_t)hash32(lane * 977u + r * 131u) << 16) ^ hash32(lane + 0x9e3779b9u * (uint32_t)(r + 1));
        const uint4* p = (const uint4*)((const char*)table + (h % slots) * 128);
        const uint4 a = p[0];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += a.x;
        const uint4 b = p[SECOND_OFFSET / 16 + (a.x == 0x12345u ? 1 : 0)];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += b.x ^ b.w;
    }
    if (acc == 0x12345678u) out[lane] = acc;
}

__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ table, uint64_t count, uint32_t* __restrict__ out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += stride) { const uint4 v = table[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int REC, int STRIDE>
static int run(const char* name, const uint4* table, uint64_t bytes, uint32_t lanes, uint32_t* out)
{
    const uint64_t slots = bytes / STRIDE;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_gather<REC, STRIDE><<<lanes / 256, 256>>>(table, slots, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_gather<REC, STRIDE><<<lanes / 256, 256>>>(table, slots, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double records = (double)lanes * kInFlight * kRounds, asked = records * REC;
    printf("%-34s %8.1f M records  %7.2f GB asked  %8.3f ms  %7.1f G records/s  %6.2f TB/s asked\n", name, records * 1e-6, asked * 1e-9, ms, records / ms * 1e-6, asked / ms * 1e-9);
    return 0;
}

int main(int argc, char** argv)
{
    const uint64_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096;
    const int lg = argc > 2 ? atoi(argv[2]) : 24;
    const uint64_t bytes = mib << 20;
    const uint32_t lanes = 1u << lg;
    uint4* table = nullptr; uint32_t* out = nullptr;
    CK(hipMalloc(&table, bytes)); CK(hipMalloc(&out, sizeof(uint32_t) * (size_t)lanes));
    CK(hipMemset(table, 0x5a, bytes));
    printf("table %llu MiB, %u lanes, %d records in flight per lane, %d rounds\n", (unsigned long long)mib, lanes, kInFlight, kRounds);
    if (run<64, 64>("64 B at 64-B slots (node)", table, bytes, lanes, out)) return 1;
    if (run<128, 128>("128 B at 128-B slots", table, bytes, lanes, out)) return 1;
    if (run<48, 48>("48 B at 48-B slots (triangle)", table, bytes, lanes, out)) return 1;
    if (run<16, 64>("16 B of a 64-B slot", table, bytes, lanes, out)) return 1;
    if (run<32, 128>("32 B of a 128-B slot", table, bytes, lanes, out)) return 1;
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const uint64_t slots = bytes / 128; const int rounds = 512, blocks = 64; const double records = (double)blocks * 256 * rounds;
        float ms = 0;
        k_halves<64><<<blocks, 256>>>(table, slots, out, rounds); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_halves<64><<<blocks, 256>>>(table, slots, out, rounds); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8.1f M records  %27s %8.3f ms  (16 K lanes, one record in flight each)\n", "16 B, then 16 B of the other half", records * 1e-6, "", ms);
        k_halves<16><<<blocks, 256>>>(table, slots, out, rounds); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_halves<16><<<blocks, 256>>>(table, slots, out, rounds); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8.1f M records  %27s %8.3f ms\n", "16 B, then 16 B of the same half", records * 1e-6, "", ms);
    }
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const uint64_t count = bytes / 16;
        k_stream<<<256 * 16, 256>>>(table, count, out); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k_stream<<<256 * 16, 256>>>(table, count, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8s            %7.2f GB asked  %8.3f ms  %17s  %6.2f TB/s asked\n", "coalesced stream, 16 B per lane", "", bytes * 1e-9, ms, "", bytes / ms * 1e-9);
    }
    CK(hipFree(table)); CK(hipFree(out));
    return 0;
}
