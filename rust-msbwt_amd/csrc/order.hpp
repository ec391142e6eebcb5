// Batch order (order.hip): the order in which a batch of k-mers walks the index most cheaply -- its key, and the in-library
// bucket pass that puts a dense batch into that order on the device.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace msbwt {
void order_keys_host(const uint8_t *kmers, uint32_t k, uint64_t n, uint64_t *keys);
hipError_t launch_order_keys(const uint8_t *d_kmers, uint32_t k, uint64_t n, uint64_t *d_keys, hipStream_t stream);

// How a batch of n k-symbol queries (k <= 64, n < 2^32) is ordered: `bits` key bits -- the top bits of the table index
// of `reach` symbols -- in two bucket passes (level 0: one global pass of bits0 <= 11 bits over chunks of the batch; level 1:
// bits1 <= 12 more inside each level-0 bucket), and where everything lives inside ONE scratch allocation.
struct OrderPlan {
    uint64_t n;
    uint32_t k, words, reach, bits0, bits1, chunk, nchunks, wg0, wg1;
    bool from_rows;  // the batch arrives as rows of symbol codes (packed here) or as 2-bit words already
    uint64_t off_packed_rows, off_elems0, off_elems1, off_counts, off_hist, off_totals, off_starts, off_exceptions, off_nexceptions;
    uint64_t scratch_bytes;
};
OrderPlan plan_order(uint64_t n, uint32_t k, uint32_t reach, uint32_t bits, bool from_rows);
// Enqueues the passes.  *ordered: n elements of (words + 1) u64 -- the query in the layout of QuerySource::packed, then one
// word that (when *place_inline) names where the search is to write the query's count inside *counts (else: at its own
// position); all inside d_scratch.  Rows that two bits cannot say are listed for launch_count_exceptions.
hipError_t launch_order_batch(const OrderPlan &p, const uint8_t *d_rows, const uint64_t *d_packed, void *d_scratch, hipStream_t stream,
                              const uint64_t **ordered, bool *place_inline, uint64_t **counts);
// After the search: the counts go from the scratch to d_out in the caller's order.
hipError_t launch_order_finish(const OrderPlan &p, void *d_scratch, uint64_t *d_out, hipStream_t stream);
// Counts the listed rows from the caller's matrix (any symbols) into d_counts[their index]; enqueue AFTER the search kernel.
hipError_t launch_count_exceptions(const OrderPlan &p, const void *d_blocks, uint64_t total, const uint8_t *d_rows, void *d_scratch, uint64_t *d_counts,
                                   uint32_t *d_flags, hipStream_t stream);
// rows[q * k + i] = symbol code of query q's i-th symbol, from its 2-bit words (QuerySource::packed layout)
hipError_t launch_unpack_rows(const uint64_t *d_packed, uint32_t k, uint64_t n, uint8_t *d_rows, hipStream_t stream);
}  // namespace msbwt
