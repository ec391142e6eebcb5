// Device-side index build (device_build.hip): RLE bytes already in HBM -> plane blocks.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace msbwt {

constexpr uint32_t kBuildBadSymbol = 1u;  // a byte carries symbol code 6 or 7
constexpr uint32_t kBuildTooLarge = 2u;   // a run has a non-zero digit beyond 32^7: T >= 2^40

struct DeviceBuildState {
    uint64_t *d_totals = nullptr;       // [0..5] symbol counts, [6] total symbols
    uint64_t *d_start_index = nullptr;  // 6 x u64, written by the host between the passes
    uint32_t *d_flags = nullptr;
    unsigned long long *d_long_count = nullptr;   // sub-runs of >= 2048 symbols (exact)
    unsigned long long *d_long_cursor = nullptr;
    void *d_tiles = nullptr;
    uint64_t ntiles = 0;
};

// Scratch for n RLE bytes (56 bytes per 4 KiB tile).
size_t device_build_scratch_bytes(size_t n);
size_t device_build_long_run_bytes(uint64_t nlong);

// Pass 1: tile sums + scan.  Afterwards st->d_totals / d_flags / d_long_count are valid
// (read them back after synchronising the stream).
hipError_t device_build_pass1(const uint8_t *d_rle, size_t n, void *d_scratch, DeviceBuildState *st, hipStream_t stream);
// Pass 2: paint.  d_blocks must be zero-filled, st.d_start_index filled in.
hipError_t device_build_pass2(const uint8_t *d_rle, size_t n, const DeviceBuildState &st, void *d_long_runs,
                              uint64_t nlong, void *d_blocks, hipStream_t stream);

}  // namespace msbwt
