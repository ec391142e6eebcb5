// Batch order keys: in which order a batch of k-mers walks the index most cheaply.
//
// A backward search consumes a k-mer from its LAST symbol.  After j symbols the ranges of different queries are laid out in
// the BWT in lexicographic order of their j-symbol suffixes, and the suffix table holds its entries in exactly that order
// for j = table depth.  A batch whose queries are ordered by  (the last 17 symbols as a string, then the symbols before them
// going leftwards)  therefore reads table lines in ascending order, starts from ascending ranges, and keeps that order
// within each symbol class through the following steps: neighbouring lanes and waves touch neighbouring lines, which the
// L2 / Infinity Cache and the DRAM pages reward -- measured 2.0x on dense batches (C4 read-derived, C3), +15 % on the
// human-scale default batch (profiles/r03_lab/sorted_batch_*.json).  The library does not reorder batches
// itself (a sort of 3 x 10^8 keys costs about what it saves there); it hands out the key, and a caller that can afford or
// already has the order -- a sorted k-mer list, a batch that is counted more than once -- sorts by it.
//
// key = sum over t < min(k, 17) of code(kmer[k-1-t]) << (28 + 2 t)  (A C G T -> 0..3: the table index, most significant)
//     | the next up to 14 symbols to the left, kmer[k-18] most significant, in the low 28 bits;
// a '$' / 'N' / invalid symbol among the symbols used gives UINT64_MAX (such queries sort last).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "order.hpp"

namespace msbwt {

__host__ __device__ inline uint64_t order_key_of(const uint8_t *kmer, uint32_t k) {
    const uint32_t depth = k < 17u ? k : 17u, more = k - depth < 14u ? k - depth : 14u;
    uint64_t key = 0;
    for (uint32_t t = 0; t < depth; ++t) {
        const uint32_t s = kmer[k - 1u - t];
        if (s != 1u && s != 2u && s != 3u && s != 5u) return ~0ull;
        key |= uint64_t(s - 1u - (s >> 2)) << (28u + 2u * t);
    }
    for (uint32_t j = 0; j < more; ++j) {
        const uint32_t s = kmer[k - 1u - depth - j];
        if (s != 1u && s != 2u && s != 3u && s != 5u) return ~0ull;
        key |= uint64_t(s - 1u - (s >> 2)) << (26u - 2u * j);
    }
    return key;
}

namespace {
__global__ __launch_bounds__(256) void k_order_keys(const uint8_t *__restrict__ kmers, uint32_t k, uint64_t n, uint64_t *__restrict__ keys) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t q = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < n; q += stride) keys[q] = order_key_of(kmers + q * k, k);
}
}  // namespace

void order_keys_host(const uint8_t *kmers, uint32_t k, uint64_t n, uint64_t *keys) {
    for (uint64_t q = 0; q < n; ++q) keys[q] = order_key_of(kmers + q * k, k);
}

hipError_t launch_order_keys(const uint8_t *d_kmers, uint32_t k, uint64_t n, uint64_t *d_keys, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_order_keys, dim3(uint32_t(blocks > 256 * 32 ? 256 * 32 : blocks)), dim3(256), 0, stream, d_kmers, k, n, d_keys);
    return hipGetLastError();
}

}  // namespace msbwt
