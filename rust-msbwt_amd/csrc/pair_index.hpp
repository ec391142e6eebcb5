// Pair-index build (pair_index.hip): plane blocks in HBM -> pair blocks + superblock table.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace msbwt {

struct PairIndexSizes {
    uint64_t pair_blocks = 0, tiles = 0, supers = 0;
    size_t pair_block_bytes = 0, super_bytes = 0, scratch_bytes = 0;
};

// Sizes for an index of `nblocks` plane blocks; stride = 128 (disjoint pair blocks) or 96 (overlapping).
PairIndexSizes pair_index_sizes(uint64_t nblocks, int stride = 128);

// Enqueues the whole build on `stream`.  d_scratch may be freed once the stream has drained.
hipError_t build_pair_index(const void *d_blocks, uint64_t nblocks, const uint64_t start_index[6], void *d_pair_blocks,
                            void *d_super, void *d_scratch, hipStream_t stream, int stride = 128);

}  // namespace msbwt
