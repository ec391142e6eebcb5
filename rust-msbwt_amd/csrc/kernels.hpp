// Launch wrappers of the gfx950 kernels (kernels.hip).  All pointers are device pointers.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "sparse_table.hpp"

namespace msbwt {

// bits of the device status word
constexpr uint32_t kFlagInvalidSymbol = 1u;
constexpr uint32_t kFlagInvalidRange = 2u;
constexpr uint32_t kFlagInternal = 4u;  // a device-side consistency check failed (never on a well-formed index)
constexpr uint32_t kFlagNarrowOverflow = 8u;  // a count did not fit the wire width of msbwt_rle_allgather_counts

// One entry of the suffix table: the range after the last `depth` symbols of a k-mer.
struct TableView {
    const void *entries;  // 4^depth entries, or nullptr: flat = {l, h} u64 pairs (16 B each);
                          // packed = 128-byte lines of 30 entries (search_common.hpp, kPackedPerLine)
    int depth;            // 0 = no table
    bool packed = false;
    // presence filter: bit i = "some table entry whose low 2*filter_depth index bits equal i is
    // non-empty"; small enough (<= 2 MiB) to live in L2, it decides absent k-mers without the
    // random table line.  nullptr = none.
    const uint32_t *filter = nullptr;
    int filter_depth = 0;
    // packed tables: the ranges of the lines whose deltas do not fit 16 bits (high-copy repeats), 32 flat 16-byte
    // {l, h} entries (30 used) per escaped line; the line's base word names its group.  nullptr = queries of an
    // escaped line search from scratch.
    const void *side = nullptr;
};

// which count_kmers kernel serves 1 <= k <= 64 (kernels.hip / lanes.hip)
constexpr int kSearchAuto = 0, kSearchGroups = 1, kSearchLanes = 2;

constexpr int kBlocksPlanes = 0, kBlocksRuns = 1;  // IndexView::block_format

// The lanes kernel deals tiles out by atomic tickets.  One address takes only ~7 x 10^7 atomics/s, so a
// launch uses kTicketCounters counters on separate 128-byte lines (waves are spread over them; counter c
// hands out tickets c, c + 16, c + 32, ...).
constexpr int kTicketCounters = 16;
constexpr size_t kTicketBytes = size_t(kTicketCounters) * 128;

struct IndexView {
    const void *blocks;  // plane blocks (256 positions) or run blocks (512 positions, run_index.hpp), 128 B each
    int block_format = kBlocksPlanes;
    const void *overflow = nullptr;  // run blocks: plane-shaped lines of the overflowing blocks
    uint64_t nblocks;
    uint64_t total;
    TableView table;
    const void *pair_blocks = nullptr;      // optional: two symbols per step (rank_ops.hpp)
    const uint64_t *pair_super = nullptr;
    bool pair_stride96 = false;             // pair blocks overlap (start every 96 positions)
    int search_kernel = kSearchAuto;
    uint64_t *debug = nullptr;  // 8 words: [0] != 0 once a consistency check has recorded its values in [1..]
    void *tile_counter = nullptr;  // kTicketBytes of ticket counters for the lanes kernel's tile dealing (zeroed per launch)
    // one-wave launches of the lanes kernel (n <= 64): *done = done_seq (system scope) once every count is out, so
    // that a host may poll instead of synchronising the stream; nullptr = not wanted
    uint64_t *done = nullptr;
    uint64_t done_seq = 0;
    // optional search counters of the lanes kernel (kSearchCounters u64, added to by every wave): what a batch did to
    // the index -- steps, second lines, escape lines ... (msbwt_rle_search_counters); nullptr = not wanted
    uint64_t *counters = nullptr;
    // The index's random-access arrays are far larger than L2 and Infinity Cache: the lanes kernel fetches their lines with the
    // non-temporal hint, so that lines used once do not evict what is reused (capi.cpp decides; lanes.hip has the measurements).
    bool stream_lines = false;
    // optional sparse suffix table (sparse_table.hpp): the ranges of the suffixes that occur, deeper than the direct table
    // reaches; the lanes kernel looks up queries of at least its depth there (beside a pair index), shorter ones in `table`
    SparseView sparse;
    // optional SECOND, shallower sparse table (round 6): serves the queries shorter than the first one's entries -- with k undeclared the
    // first table is 23 deep, and k = 17..22 would otherwise fall to the direct table, which stays shallow beside a sparse table
    SparseView sparse2;
};

// the sparse table that serves k-symbol queries: the deepest one whose entries are no longer than k (nullptr: the direct table)
inline const SparseView *sparse_for(const IndexView &ix, uint32_t k) {
    // (beside a pair index, or -- round 6 -- on run blocks, whose table was built from pair blocks that are gone again)
    if (ix.pair_blocks == nullptr && ix.block_format != kBlocksRuns) return nullptr;
    if (ix.sparse.lines != nullptr && k >= ix.sparse.depth) return &ix.sparse;
    if (ix.sparse2.lines != nullptr && k >= ix.sparse2.depth) return &ix.sparse2;
    return nullptr;
}

inline bool sparse_serves(const IndexView &ix, uint32_t k) { return sparse_for(ix, k) != nullptr; }

// indices into IndexView::counters
enum SearchCounter {
    kCntWaveSteps = 0,     // search steps of a wave (each costs the same whether 5 or 64 lanes take it)
    kCntLaneSteps,         // steps taken by a lane's query (pair or single-symbol)
    kCntPairSteps,         // ... of which consumed two symbols
    kCntSecondLines,       // steps whose range needed a second line (l and h in different blocks)
    kCntSatOut,            // lane-steps lost because the step's second-line slots were taken
    kCntEscapeQueries,     // queries whose packed table line is an escape line
    kCntEscapeRestarts,    // ... that searched from [0, total) because the table has no side array
    kCntTableDecided,      // queries decided by the table or the presence filter alone
    kCntSearched,          // queries that entered the search
    kCntFirstLines,        // first-bound lines fetched (one per lane-step) + side-array entries fetched for escape-line queries
    kCntTableSteps,        // sparse table: bucket lines fetched (lane-steps that were lookups; included in kCntLaneSteps)
    kCntTableDisplaced,    // ... of which did not find the key in a bucket that had displaced entries (the lookup went on)
    kCntTableRides,        // ... of which rode along with the search of the tile before (no search step of their own)
    kCntWavesWorked,       // waves of the persistent kernel that were dealt at least one tile (a workgroup that became resident late finds none)
    kCntTierFallbacks,     // two-tier sparse table: lookups that ended in the filter and went on through the direct table
    kSearchCounters = 16
};

// counts[q] = count_kmer(kmers[q*k .. q*k+k)) for q < n.  Sets kFlagInvalidSymbol in *flags
// (and writes UINT64_MAX) for a query holding a code >= 6.
// Which kernel a count_kmers / count_read_kmers launch for k-symbol queries runs on this index:
// kSearchLanes (lanes.hip), kSearchGroups (the tiled kernel of kernels.hip) or 0 (k > 64: generic kernel)
int search_kernel_for(const IndexView &ix, uint32_t k);
// inline_kmer (optional, HOST pointer, n == 1, k <= 64): the query travels inside the kernel arguments instead of being
// read from `kmers` (lanes kernel only: ask lanes_serves(ix, k) first).
hipError_t launch_count_kmers(const IndexView &ix, const uint8_t *kmers, uint32_t k, uint64_t n,
                              uint64_t *counts, uint32_t *flags, hipStream_t stream, const uint8_t *inline_kmer = nullptr);
// The same for queries handed over as 2-bit words (n x ceil(k / 32) u64: the k-mer as a base-4 number, first symbol most
// significant, A C G T -> 0..3; search_common.hpp, QuerySource::packed), 1 <= k <= 64, plane blocks: always the lanes
// kernel.  out_index (optional, n < 2^32): query v's count goes to counts[out_index[v]] -- an ordered batch (order.hip).
// stride_words (0 = ceil(k / 32)): u64 words from one query to the next; place_inline: the word after a query's own holds
// the place of its count (low 32 bits) instead of out_index[] -- the elements the ordering passes produce.
hipError_t launch_count_packed(const IndexView &ix, const uint64_t *packed, uint32_t k, uint64_t n, uint64_t *counts,
                               const uint32_t *out_index, uint32_t *flags, hipStream_t stream, uint32_t stride_words = 0, bool place_inline = false);
// true when a count_kmers launch for k-symbol queries runs the lanes kernel (the only one that knows ix.done / inline queries)
bool lanes_serves(const IndexView &ix, uint32_t k);

// Every k-mer window of every read (n_reads x read_len bytes, symbol codes or ASCII), forward
// and/or reverse-complemented; out_*[r * (read_len-k+1) + w].  1 <= k <= 32, k <= read_len.
hipError_t launch_count_read_kmers(const IndexView &ix, const uint8_t *reads, uint32_t read_len, uint64_t n_reads,
                                   uint32_t k, bool ascii, uint64_t *out_fwd, uint64_t *out_rc, uint32_t *flags,
                                   hipStream_t stream);

// Same for reads of different lengths: read r = reads[read_off[r] .. read_off[r+1]), its
// windows are the global windows [win_off[r], win_off[r+1]) (win_off = prefix sum of
// max(0, len - k + 1)); all arrays on the device; out_*[global window].
hipError_t launch_count_ragged_read_kmers(const IndexView &ix, const uint8_t *reads, const uint64_t *read_off,
                                          const uint64_t *win_off, uint64_t n_reads, uint64_t n_windows, uint32_t k,
                                          bool ascii, uint64_t *out_fwd, uint64_t *out_rc, uint32_t *flags,
                                          hipStream_t stream);

// (out_l[i], out_h[i]) = constrain_range(syms[i], [l[i], h[i])).
hipError_t launch_constrain_ranges(const IndexView &ix, const uint8_t *syms, const uint64_t *l,
                                   const uint64_t *h, uint64_t n, uint64_t *out_l, uint64_t *out_h,
                                   uint32_t *flags, hipStream_t stream);

// Fills the suffix table of `depth` symbols (entries: 4^depth x {l,h}) by backward search
// on the device.
hipError_t launch_build_table(const IndexView &ix, int depth, void *entries, hipStream_t stream);
// Packs a finished FLAT table of `flat_depth` levels into a PACKED table two levels deeper
// (`packed_entries`: ceil(4^(flat_depth+2) / 30) lines of 128 bytes): every entry is extended by
// one two-symbol step of the pair index (required).  Returns the packed size through the helper.
// escape_count (device, zeroed by the caller): receives the number of escape lines.  A second call with `side` (that many
// groups of 512 bytes) and `side_cursor` (device, zeroed) gives every escape line its group of flat entries.
hipError_t launch_pack_table(const IndexView &ix, int flat_depth, const void *flat_entries, void *packed_entries,
                             unsigned long long *escape_count, void *side, unsigned long long *side_cursor, hipStream_t stream);
// d_out[g] (g < nsamples) = number of occurrences of a `steps`-mer that is PRESENT in the index (an LF walk from a
// pseudo-random row, searched as it is read off; 0 = the walk met '$' / 'N').  What the pair-stride policy reads
// (table_policy.hpp).  Plane blocks only.
hipError_t launch_probe_widths(const IndexView &ix, uint32_t nsamples, uint32_t steps, uint64_t seed, uint64_t *d_out, hipStream_t stream);
// Random 128-byte lines of [base, base + bytes) gathered for `iters` rounds (2^19 lines per round, nothing written): timed by the
// caller with events, lines per second say how well THIS allocation is served (msbwt_rle_probe_line_rate).  sink: 4 device bytes or nullptr.
hipError_t launch_probe_lines(const void *base, uint64_t bytes, uint32_t iters, uint64_t *lines_touched, uint32_t *sink, hipStream_t stream);
inline uint64_t packed_table_bytes(int depth) { return ((uint64_t(1) << (2 * depth)) + 29) / 30 * 128; }
// filter (zero-filled, 4^filter_depth bits) from a finished table of `depth` levels
hipError_t launch_build_filter(const void *entries, int depth, int filter_depth, uint32_t *filter, hipStream_t stream);

}  // namespace msbwt
