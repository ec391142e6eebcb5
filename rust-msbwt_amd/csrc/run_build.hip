// Device-side build of the run-block index (run_index.hpp) from finished plane blocks in HBM (round 4; until then the
// memory-lean format was built on the host: 32 s at human scale).  The load path expands the RLE bytes into plane blocks on
// the device as for the default format (device_build.hip); here every 512-position run block is made from its two plane
// blocks by ONE thread -- run boundaries are the positions whose symbol differs from the one before (three XORs per 32
// positions), runs are cut into pieces of at most 31 -- and the plane blocks are freed afterwards.  A block that needs more
// than 96 pieces keeps its two plane blocks as they are, headers included, in the side array: that IS the overflow format.
//
//   k_run_blocks<false>  counts the pieces of every block and the blocks that overflow
//   k_run_blocks<true>   writes the blocks (runs staged in LDS, stored as whole lines) and the overflow pairs
#include <hip/hip_runtime.h>

#include "plane_index.hpp"
#include "run_build.hpp"
#include "run_index.hpp"

namespace msbwt {
namespace {

constexpr int kThreads = 256;

// Walks the runs of the (at most) 512 positions of run block b: fn(sym, len) for every maximal run of equal symbols, in order.
template <class Fn>
__device__ __forceinline__ void for_each_run(const uint4 *__restrict__ planes, uint64_t nplane_blocks, uint64_t b, uint32_t valid, Fn &&fn) {
    uint32_t cur_sym = 8u, cur_len = 0u;  // (no symbol yet)
    for (uint32_t half = 0; half < 2u && half * 256u < valid; ++half) {
        const uint64_t pb = 2u * b + half;
        if (pb >= nplane_blocks) break;
        for (uint32_t j = 0; j < 8u && half * 256u + j * 32u < valid; ++j) {
            const uint4 c = planes[pb * 8u + j];
            const uint32_t here = min(32u, valid - (half * 256u + j * 32u));  // positions of this chunk that exist
            // bit i: the symbol at i differs from the one at i - 1 (bit 0: from the running symbol)
            const uint32_t first = (c.x & 1u) | ((c.y & 1u) << 1) | ((c.z & 1u) << 2);
            uint32_t edge = ((c.x ^ (c.x << 1)) | (c.y ^ (c.y << 1)) | (c.z ^ (c.z << 1))) & ~1u;
            edge |= first != cur_sym ? 1u : 0u;
            edge &= here >= 32u ? ~0u : ((1u << here) - 1u);
            uint32_t at = 0;  // positions of this chunk already given to the running run
            while (edge != 0u) {
                const uint32_t i = uint32_t(__ffs(int(edge))) - 1u;
                cur_len += i - at;
                if (cur_len != 0u) fn(cur_sym, cur_len);
                cur_sym = ((c.x >> i) & 1u) | (((c.y >> i) & 1u) << 1) | (((c.z >> i) & 1u) << 2);
                cur_len = 0u;
                at = i;
                edge &= edge - 1u;
            }
            cur_len += here - at;
        }
    }
    if (cur_len != 0u) fn(cur_sym, cur_len);
}

template <bool kWrite>
__global__ __launch_bounds__(256) void k_run_blocks(const uint4 *__restrict__ planes, uint64_t nplane_blocks, uint64_t total, uint64_t nblocks,
                                                    unsigned long long *__restrict__ counters /* [0] overflow blocks, [1] cursor */,
                                                    uint4 *__restrict__ out, uint4 *__restrict__ overflow) {
    __shared__ uint32_t lds_runs[kWrite ? kThreads * 24 : 1];  // 96 bytes of runs per thread
    const uint64_t b = uint64_t(blockIdx.x) * kThreads + threadIdx.x;
    uint8_t *mine = reinterpret_cast<uint8_t *>(lds_runs) + threadIdx.x * 96u;
    if (kWrite) {
#pragma unroll
        for (int i = 0; i < 24; ++i) lds_runs[threadIdx.x * 24 + i] = 0u;
    }
    bool over = false;
    if (b < nblocks) {
        const uint64_t first = b << kRunShift;
        const uint32_t valid = first >= total ? 0u : uint32_t(min(uint64_t(512), total - first));
        uint32_t np = 0;
        for_each_run(planes, nplane_blocks, b, valid, [&](uint32_t sym, uint32_t len) {
            while (len != 0u) {
                const uint32_t piece = min(len, 31u);
                if (kWrite && np < uint32_t(kRunsPerBlock)) mine[np] = uint8_t(sym | (piece << 3));
                ++np;
                len -= piece;
            }
        });
        over = np > uint32_t(kRunsPerBlock);
        if (!kWrite) {
            if (over) atomicAdd(counters, 1ull);
            return;
        }
        // header = the first plane block's (A[s] at 512 b); the block beyond the last position has none of its own
        uint32_t meta[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) meta[j] = 2u * b < nplane_blocks ? planes[2u * b * 8u + j].w : 0u;
        uint4 *dst = out + b * 8u;
        uint32_t word8 = 0;
        if (over) {  // the two plane blocks move to the side array as they are: each with its own header
            const unsigned long long idx = atomicAdd(counters + 1, 1ull);
            word8 = uint32_t(idx);
            for (uint32_t half = 0; half < 2u; ++half)
                for (uint32_t j = 0; j < 8u; ++j)
                    overflow[idx * 16u + half * 8u + j] = 2u * b + half < nplane_blocks ? planes[(2u * b + half) * 8u + j] : make_uint4(0u, 0u, 0u, 0u);
            meta[7] |= kRunOverflowBit;
        }
        dst[0] = make_uint4(meta[0], meta[1], meta[2], meta[3]);
        dst[1] = make_uint4(meta[4], meta[5], meta[6], meta[7]);
        const uint32_t *r = lds_runs + threadIdx.x * 24;
        if (over) {
            dst[2] = make_uint4(word8, 0u, 0u, 0u);
#pragma unroll
            for (int j = 3; j < 8; ++j) dst[j] = make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) dst[2 + j] = make_uint4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]);
        }
    }
}

}  // namespace

hipError_t launch_run_block_count(const void *d_planes, uint64_t nplane_blocks, uint64_t total, unsigned long long *d_counters, hipStream_t stream) {
    const uint64_t nblocks = run_block_count(total);
    hipError_t e = hipMemsetAsync(d_counters, 0, 16, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_run_blocks<false>), dim3(uint32_t((nblocks + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, static_cast<const uint4 *>(d_planes),
                       nplane_blocks, total, nblocks, d_counters, static_cast<uint4 *>(nullptr), static_cast<uint4 *>(nullptr));
    return hipGetLastError();
}

hipError_t launch_run_block_write(const void *d_planes, uint64_t nplane_blocks, uint64_t total, unsigned long long *d_counters, void *d_run_blocks,
                                  void *d_overflow, hipStream_t stream) {
    const uint64_t nblocks = run_block_count(total);
    hipLaunchKernelGGL((k_run_blocks<true>), dim3(uint32_t((nblocks + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, static_cast<const uint4 *>(d_planes),
                       nplane_blocks, total, nblocks, d_counters, static_cast<uint4 *>(d_run_blocks), static_cast<uint4 *>(d_overflow));
    return hipGetLastError();
}

}  // namespace msbwt
