// Builder of the sparse suffix table (sparse_table.hpp; sparse_table.hip): frontier expansion on the device with the
// index's own rank code -- each present d-mer -> its <= 16 present (d+2)-mers by one pair step -- so the table is
// bit-exact by construction.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "kernels.hpp"
#include "sparse_table.hpp"

namespace msbwt {

// What the builder learned about the data (msbwt_rle_sparse_table_info): how many DISTINCT suffixes of each length occur.
struct SparseBuildReport {
    uint64_t distinct[kSparseMaxDepth + 1] = {};  // [d] = non-empty ranges at depth d (0 where the build did not pass)
    uint64_t escapes[kSparseMaxDepth + 1] = {};   // [d] = of which 255 or more wide
    uint64_t singles[kSparseMaxDepth + 1] = {};   // [d] = of which exactly 1 wide: suffixes that occur once (on reads with errors, mostly error k-mers)
    uint64_t entries = 0, nescapes = 0, displaced = 0, nbuckets = 0;  // of the table that was filled
    uint64_t filtered = 0;                        // two-tier form: suffixes that went into the filters instead of taking an entry
    bool tier = false;
    int depth = 0, parent_depth = 0;
};

// Scratch both passes work in (two frontier buffers + cursors), sized once.
size_t sparse_work_bytes(uint64_t free_bytes);

// Sizing pass: expands the non-empty entries of the flat table of `flat_depth` levels (flat_entries == nullptr: from
// [0, total), depth 0) level by level up to max_depth and reports the distinct counts on the way.  Synchronises the stream.
hipError_t sparse_count_levels(const IndexView &ix, const void *flat_entries, int flat_depth, int max_depth, void *d_work, size_t work_bytes,
                               SparseBuildReport *report, hipStream_t stream);
// Fill pass at `depth` (any depth the sizing pass reached or passed): `lines` (nbuckets x 128 bytes, zeroed here) and `side`
// (nside = report->escapes[depth] entries of 16 bytes, may be nullptr when 0) are written; d_counts: nbuckets x u32 of scratch.
// hipErrorInvalidValue when some entry found no slot within `probe` buckets (the caller retries with more buckets).
// `report` is IN and OUT: it must be the sizing pass's report (distinct[depth] is read to derive `entries`); depth, nbuckets, nescapes,
// displaced, entries, tier and filtered are filled in.
// tier: the two-tier form (sparse_table.hpp) -- ranges 1 wide set filter bits of their own bucket instead of taking an entry.
hipError_t sparse_fill(const IndexView &ix, const void *flat_entries, int flat_depth, int depth, bool tier, void *lines, uint64_t nbuckets, uint32_t probe,
                       void *side, uint64_t nside, void *d_counts, void *d_work, size_t work_bytes, SparseBuildReport *report, hipStream_t stream);

}  // namespace msbwt
