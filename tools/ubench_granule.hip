// Micro-benchmark (not part of the product): does the memory system serve random 64-byte (or 32-byte)
// granules faster than random 128-byte lines?  Groups of G lanes read G x 16 contiguous bytes of a random,
// granule-aligned record of a table far larger than the Infinity Cache; 8 independent loads per lane in
// flight, nothing dependent.  Reports records/s and bytes/s for G = 8 (128 B), 4 (64 B), 2 (32 B).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_granule tools/ubench_granule.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int G>
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ table, uint64_t nrec, int iters, uint32_t *__restrict__ sink) {
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t group = tid / G;
    const uint32_t piece = uint32_t(tid % G);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t rec = __umul64hi(mix(group * 0x9E3779B97F4A7C15ull + uint64_t(it) * 8 + u), nrec);
            v[u] = table[rec * G + piece];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int G>
void run(const uint4 *table, uint64_t bytes, uint32_t *sink, int blocks, int iters) {
    const uint64_t nrec = bytes / (16 * G);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    gather<G><<<blocks, 256>>>(table, nrec, 2, sink);
    CK(hipEventRecord(a));
    gather<G><<<blocks, 256>>>(table, nrec, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double recs = double(blocks) * 256 / G * iters * 8;
    printf("granule %3d B: %.3e records/s  %.2f TB/s  (%.1f ms)\n", 16 * G, recs / (ms * 1e-3), recs * 16 * G / (ms * 1e-3) / 1e12, ms);
}

int main(int argc, char **argv) {
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 32ull) << 30;
    const int blocks = argc > 2 ? atoi(argv[2]) : 256 * 8 * 4;
    const int iters = argc > 3 ? atoi(argv[3]) : 256;
    uint4 *table;
    uint32_t *sink;
    CK(hipMalloc(&table, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(table, 1, bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        run<8>(table, bytes, sink, blocks, iters);
        run<4>(table, bytes, sink, blocks, iters);
        run<2>(table, bytes, sink, blocks, iters);
        run<1>(table, bytes, sink, blocks, iters);
    }
    return 0;
}
