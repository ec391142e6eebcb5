// Micro-benchmark (not part of the product): does the rate of random 128-byte line fetches from a large table depend on WHERE
// the table sits in HBM?  (Round 4: the c4r line -- 73 GB packed table, random lines -- ran 18.0 or 20.5 ms per launch, per
// process and per placement of the table on one box; tools/placement_probe.py.)  One process allocates the table behind
// spacers of several sizes, either with hipMalloc or from 1 GiB physical chunks mapped into a 1 GiB-aligned range
// (hipMemCreate / hipMemMap), and times the same gather on each.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_placement tools/ubench_placement.hip
//   usage: ubench_placement [table GiB = 68] [spacer GiB ...]
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

__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ table, uint64_t nrec, int iters, uint32_t *__restrict__ sink) {
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t group = tid / 8;
    const uint32_t piece = uint32_t(tid % 8);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t rec = __umul64hi(mix(group * 0x9E3779B97F4A7C15ull + uint64_t(it) * 8 + u), nrec);
            v[u] = table[rec * 8 + piece];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

double lines_per_s(const uint4 *table, uint64_t bytes, uint32_t *sink) {
    const int blocks = 256 * 8 * 4, iters = 256;
    const uint64_t nrec = bytes / 128;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    gather<<<blocks, 256>>>(table, nrec, 2, sink);
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        gather<<<blocks, 256>>>(table, nrec, iters, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        const double rate = double(blocks) * 256 / 8 * iters * 8 / (ms * 1e-3);
        if (rate > best) best = rate;
    }
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return best;
}

struct Mapped {
    void *base = nullptr;
    size_t size = 0, chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

bool map_chunks(Mapped *m, size_t bytes, size_t chunk, int device) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return false;
    if (chunk % gran) chunk = (chunk / gran + 1) * gran;
    m->chunk = chunk;
    m->size = (bytes + chunk - 1) / chunk * chunk;
    if (hipMemAddressReserve(&m->base, m->size, chunk, nullptr, 0) != hipSuccess) return false;
    for (size_t off = 0; off < m->size; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) return false;
        m->handles.push_back(h);
        if (hipMemMap(static_cast<char *>(m->base) + off, chunk, 0, h, 0) != hipSuccess) return false;
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    return hipMemSetAccess(m->base, m->size, &acc, 1) == hipSuccess;
}

void unmap_chunks(Mapped *m) {
    if (m->base) {
        (void)hipMemUnmap(m->base, m->size);
        for (auto h : m->handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(m->base, m->size);
    }
    *m = Mapped{};
}

int main(int argc, char **argv) {
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 68ull) << 30;
    std::vector<uint64_t> spacers;
    for (int i = 2; i < argc; ++i) spacers.push_back(strtoull(argv[i], nullptr, 10));
    if (spacers.empty()) spacers = {0, 0, 20, 40, 60, 80, 100, 130, 160, 0};
    uint32_t *sink;
    CK(hipMalloc(&sink, 4));
    for (int mode = 0; mode < 3; ++mode) {  // 0: hipMalloc, 1: 1 GiB chunks, 2: 2 MiB-granule chunks of 64 MiB
        for (uint64_t gib : spacers) {
            void *spacer = nullptr;
            if (gib) CK(hipMalloc(&spacer, gib << 30));
            uint4 *table = nullptr;
            Mapped m;
            if (mode == 0) {
                CK(hipMalloc(reinterpret_cast<void **>(&table), bytes));
            } else {
                if (!map_chunks(&m, bytes, mode == 1 ? (1ull << 30) : (64ull << 20), 0)) {
                    printf("mode %d: mapping failed (%s)\n", mode, hipGetErrorString(hipGetLastError()));
                    unmap_chunks(&m);
                    if (spacer) CK(hipFree(spacer));
                    break;
                }
                table = static_cast<uint4 *>(m.base);
            }
            CK(hipMemset(table, 1, bytes));
            CK(hipDeviceSynchronize());
            const double rate = lines_per_s(table, bytes, sink);
            printf("%s spacer %3llu GiB  table %p  %.3e lines/s\n", mode == 0 ? "hipMalloc      " : mode == 1 ? "1 GiB chunks   " : "64 MiB chunks  ",
                   (unsigned long long)gib, static_cast<void *>(table), rate);
            fflush(stdout);
            if (mode == 0) CK(hipFree(table));
            else unmap_chunks(&m);
            if (spacer) CK(hipFree(spacer));
        }
    }
    return 0;
}
