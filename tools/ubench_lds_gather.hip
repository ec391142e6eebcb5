// Micro-benchmark (not part of the product): dependent random 128-byte line gathers with one
// QUERY PER LANE.  A wave keeps 64 (or 128) lines in flight: 8 lanes fetch one line with a
// coalesced LDS-DMA load (global_load_lds_dwordx4, per-lane source address), every lane then
// reads its own line back from LDS and derives its next address from the data.  Compared with
// the 8-lanes-per-query register gather (tools/ubench_gather.hip) this multiplies the lines in
// flight per wave by 8 without spending VGPRs on them.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_lds_gather tools/ubench_lds_gather.hip
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

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_void;


// LINES = lines per lane per iteration (1 or 2); READS = 16-byte pieces each lane reads back per line
template <int LINES, int READS>
__global__ __launch_bounds__(64 * (4 / LINES)) void gather_lds(const uint4 *__restrict__ table, uint64_t nrec, int iters,
                                                  uint32_t *__restrict__ sink) {
    // per wave and line set: 8 regions of 1 KiB (one per DMA instruction), every odd region pushed
    // 128 B further so that the per-lane read-back is bank-conflict free
    constexpr int kWaves = 4 / LINES;  // 33 KiB of LDS per block either way
    __shared__ uint4 lds[kWaves][LINES][4 * 136];  // region i at 136 (i / 2) + 72 (i % 2): 64 pieces + padding
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t g = lane >> 3, c = (lane & 7u) ^ g;  // DMA role: line of query 8i+g, chunk c
    uint64_t state[LINES];
    uint32_t acc = 0;
#pragma unroll
    for (int s = 0; s < LINES; ++s) state[s] = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 977 * s + 12345;
    for (int it = 0; it < iters; ++it) {
        uint64_t r[LINES];
#pragma unroll
        for (int s = 0; s < LINES; ++s) {
            state[s] = mix(state[s] + (acc & 1u));
            r[s] = state[s] % nrec;
        }
#pragma unroll
        for (int s = 0; s < LINES; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = 8 * i;  // lane q + g owns the query whose line this 8-lane group fetches
                const uint32_t lo = uint32_t(__shfl(int(uint32_t(r[s])), q + int(g)));
                const uint32_t hi = uint32_t(__shfl(int(uint32_t(r[s] >> 32)), q + int(g)));
                const uint64_t rec = (uint64_t(hi) << 32) | lo;
                const uint4 *src = table + rec * 8 + c;
                __builtin_amdgcn_global_load_lds((global_void *)src, (lds_void *)&lds[wave][s][136 * (i >> 1) + 72 * (i & 1)], 16, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) lgkmcnt(0): every line has landed
        __builtin_amdgcn_wave_barrier();
        // lane L = 8i' + g' reads chunk j of its own line: slot 8g' + (j ^ g') of region i'
        const uint32_t myi = lane >> 3, myg = lane & 7u;
#pragma unroll
        for (int s = 0; s < LINES; ++s) {
#pragma unroll
            for (int j = 0; j < READS; ++j) {
                const uint4 v = lds[wave][s][136 * (myi >> 1) + 72 * (myi & 1u) + 8 * myg + (uint32_t(j) ^ myg)];
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int LINES, int READS>
void run(const uint4 *table, uint64_t bytes, int iters, int blocks, uint32_t *sink) {
    const uint64_t nrec = bytes / 128;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_lds<LINES, READS>), dim3(blocks * LINES), dim3(256 / LINES), 0, 0, table, nrec, 4, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_lds<LINES, READS>), dim3(blocks * LINES), dim3(256 / LINES), 0, 0, table, nrec, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double recs = double(blocks) * 256 * LINES * iters;  // blocks*LINES blocks of 256/LINES lanes, LINES lines each
    printf("lane-per-query LDS-DMA: %d line(s)/lane, %d reads/line, %4d blocks, table %.1f GB: %.2f Glines/s  %.2f TB/s  %.3f ms\n",
           LINES, READS, blocks, bytes / 1e9, recs / ms / 1e6, recs * 128 / ms / 1e9, ms);
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
    for (int blocks : {256, 512, 768, 1024, 1536, 2048}) {
        run<1, 5>(table, bytes, iters, blocks, sink);
        run<2, 5>(table, bytes, iters, blocks, sink);
    }
    run<1, 8>(table, bytes, iters, 1024, sink);
    run<2, 8>(table, bytes, iters, 1024, sink);
    return 0;
}
