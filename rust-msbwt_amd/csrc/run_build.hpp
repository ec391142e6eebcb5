// Device-side build of the run-block index (run_build.hip): plane blocks in HBM -> run blocks + overflow pairs.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace msbwt {
// d_counters: two u64 on the device; after the count pass (synchronise the stream) [0] = blocks that overflow.
hipError_t launch_run_block_count(const void *d_planes, uint64_t nplane_blocks, uint64_t total, unsigned long long *d_counters, hipStream_t stream);
// d_run_blocks: run_block_count(total) x 128 bytes; d_overflow: 256 bytes per overflowing block (may be nullptr when there is none)
hipError_t launch_run_block_write(const void *d_planes, uint64_t nplane_blocks, uint64_t total, unsigned long long *d_counters, void *d_run_blocks,
                                  void *d_overflow, hipStream_t stream);
}  // namespace msbwt
