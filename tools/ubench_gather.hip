// Micro-benchmark (not part of the product): rate of random record gathers from a table far
// larger than the Infinity Cache, for 32/64/128-byte records, each record read by
// REC/16 adjacent lanes with 16-byte loads (the access shape of the rank kernel).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int LANES>  // lanes per record; record = LANES*16 bytes
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ table, uint64_t nrec, int iters, int chain,
                                              uint32_t *__restrict__ sink) {
    const uint32_t sub = threadIdx.x % LANES;
    const uint64_t grp = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) / LANES;
    uint32_t acc = 0;
    uint64_t state = grp * 0x9E3779B97F4A7C15ull + 12345;
    for (int i = 0; i < iters; ++i) {
        state = mix(state + (chain ? (acc & 1u) : 0u));  // chain=1: next address depends on the data
        const uint64_t r = state % nrec;
        const uint4 v = table[r * LANES + sub];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        // make the group agree on acc's low bit so the chain stays group-uniform
        if (chain) acc = __shfl(acc, (threadIdx.x & 63) / LANES * LANES);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int LANES>
void run(const uint4 *table, uint64_t bytes, int iters, int chain, uint32_t *sink) {
    const uint64_t nrec = bytes / (LANES * 16);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int blocks = 2048;
    hipLaunchKernelGGL(gather<LANES>, dim3(blocks), dim3(256), 0, 0, table, nrec, 4, chain, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(gather<LANES>, dim3(blocks), dim3(256), 0, 0, table, nrec, iters, chain, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double recs = double(blocks) * 256 / LANES * iters;
    printf("rec %3d B  chain %d  table %.1f GB: %.2f Grec/s  %.2f TB/s (record bytes)  %.3f ms\n", LANES * 16, chain,
           bytes / 1e9, recs / ms / 1e6, recs * LANES * 16 / ms / 1e9, ms);
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 8.0;
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    const uint64_t bytes = uint64_t(gb * 1e9) / 4096 * 4096;
    uint4 *table;
    uint32_t *sink;
    CK(hipMalloc(&table, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(table, 1, bytes));
    for (int chain = 0; chain < 2; ++chain) {
        run<2>(table, bytes, iters, chain, sink);
        run<4>(table, bytes, iters, chain, sink);
        run<8>(table, bytes, iters, chain, sink);
        run<16>(table, bytes, iters, chain, sink);
    }
    return 0;
}
