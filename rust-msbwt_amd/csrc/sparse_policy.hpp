// Which depth does the sparse suffix table (sparse_table.hpp) get?  Host only, no HIP: a pure function of what the sizing pass counted
// and of the bytes that are free, pinned by a CPU test through msbwt_auto_sparse_depth.
//
// The table's size follows the DATA -- 14.2 bytes per distinct d-symbol suffix that occurs -- not the depth: on an error-free read
// set the distinct count saturates at the genome's size (human scale: 2.74e9 17-mers, 2.98e9 23-mers), on reads with errors every
// error adds up to d novel d-mers (C4: 2.4e8 distinct 23-mers for a 6.4e7-bp genome).  So the automatic choice is the DEEPEST depth
// the sizing pass reached (it advances two symbols at a time from the direct table's depth, at most to 23) whose table fits the
// bytes available, skipping depths whose smallest permissible table (the tags need 2^(2d - 21) buckets) would be more than 16 x
// larger than its entries need (toy indexes stay small); none fits -> no sparse table (the loader then builds the deep direct table).
// Round 6: where the complete table of a depth does not fit, its TWO-TIER form (sparse_table.hpp) is tried before the next shallower
// depth -- entries only for the suffixes that occur at least twice, filter bits for the rest.
#pragma once
#include <algorithm>
#include <cstdint>

#include "sparse_table.hpp"

namespace msbwt {

struct SparseChoice {
    int depth = 0;            // 0 = none
    uint64_t nbuckets = 0;    // of the table at that depth
    uint64_t bytes = 0;       // bucket lines + side array
    uint64_t build_bytes = 0; // ... + the slot counters the fill pass needs beside them
    bool tier = false;        // the two-tier form (sparse_table.hpp): entries for the suffixes at least 2 wide, filter bits for the rest
};

// How deep the AUTOMATIC table may go, given the k the index will mostly be asked about (msbwt_rle_set_query_length; 0 = unknown).
// A hashed table of d-mers serves k >= d only, and every two symbols of depth save a present k-mer one index line: unknown -> 23 (serves
// every k >= 23; what round 5 shipped), a declared k -> that k, within what the format reaches (16..31: a present 31-mer behind depth 27
// needs 3 lines instead of 5, behind depth 29 two -- 1.34e10 and 1.92e10 against 8.4e9 q/s at human scale --, behind depth 31 the one bucket).  Whether the deepest depths
// are WORTH their memory is choose_sparse_depth's call: depth 29 needs 2^29 buckets (69 GB) whatever the index holds.
inline int sparse_auto_max_depth(int query_length) {
    if (query_length <= 0) return kSparseAutoDepth;
    return std::max(kSparseMinDepth, std::min(query_length, kSparseMaxDepth));
}

// distinct[d] / wide[d]: non-empty ranges at depth d and how many of them are 255 or more wide (0 for depths the pass did not reach);
// parent_depth: the direct table the pass started from; avail: bytes the table (and its build scratch) may take;
// explicit_depth: 0 = automatic, else exactly that depth or nothing.
// Fewest buckets of a TWO-TIER table: a probe limit of 3 suffices there (4 W <= 2^bits instead of the complete table's 8 W) -- its buckets hold
// only the solid suffixes, 6.4 of 10 slots on average at the most and far fewer wherever this bound binds (a chr20-sized read set at depth 23:
// 4.0 per bucket, 0.3 % of the buckets over-subscribed, three in a row never), so what the complete table needs seven further buckets for the
// two-tier one does with three; should an entry find no slot after all, the fill fails and the loader retries with a quarter more buckets.
// Half the complete table's least size: 2^24 buckets (2.1 GB) at depth 23.
inline uint64_t sparse_tier_min_buckets(int depth) {
    const int bits = int(sparse_tag_bits(uint32_t(depth)));
    const int shift = 2 * depth - 32 + 2;  // 4 W <= 2^bits  <=>  ceil(2^32 / nb) <= 2^(bits - 2 - (n - 32))
    if (shift >= bits) return ~uint64_t(0);
    if (bits - shift >= 32) return 1;
    const uint64_t per_top = uint64_t(1) << (bits - shift);
    return ((uint64_t(1) << 32) + per_top - 1) / per_top;
}

// Entries per bucket the two-tier form aims for: 5.8 of its 10 slots (5.0 of 9 with 32-bit tags), NOT the complete table's 64 % -- a bucket of 10
// overflows relatively more often than one of 14, and at 6.4 per bucket the chains of over-subscribed buckets grow past the probe limit of 15
// somewhere in a table of 5e8 buckets (round 6: the human-scale fill failed once and was retried with a quarter more buckets, 74 GB instead of
// 60; a simulation of the cascade gives runs of 12 in 3e6 buckets at 6.43, of 8 at 5.8, of 7 for the complete table's 9 of 14).
inline double sparse_tier_load(int depth) { return sparse_wide(uint32_t(depth)) ? 5.0 : 5.8; }

// buckets of the two-tier form: `solid` entries at its load, room in the filters for `singles` suffixes that occur once, what its tags allow
inline uint64_t sparse_tier_buckets(int depth, uint64_t solid, uint64_t singles) {
    const uint64_t by_entries = uint64_t(double(solid) / sparse_tier_load(depth)) + 1;
    const uint64_t by_filter = uint64_t(double(singles) / kTierMaxSinglesPerBucket) + 1;
    return std::max(std::max(by_entries, by_filter), sparse_tier_min_buckets(depth));
}

// singles[d]: of distinct[d], the suffixes that occur exactly once (nullptr: not counted -- no two-tier form); tiers: -1 = the complete
// table where it fits, else the two-tier form of the SAME depth where that fits (reads with errors: the complete table follows the error
// k-mers, the two-tier one the genome), else the next shallower depth; 0 = complete tables only; 1 = two-tier only.
inline SparseChoice choose_sparse_depth(const uint64_t *distinct, const uint64_t *wide, int parent_depth, int max_depth, uint64_t avail, int explicit_depth,
                                        const uint64_t *singles = nullptr, int tiers = 0) {
    SparseChoice none;
    for (int d = std::min(max_depth, kSparseMaxDepth); d >= kSparseMinDepth && d > parent_depth; --d) {
        if (explicit_depth ? d != explicit_depth : distinct[d] == 0) continue;  // not a level of the pass (the other parity), or nothing occurs
        for (int tier = 0; tier <= 1; ++tier) {
            // (tiers = 1 asks for the two-tier form wherever it exists: depths 30..31 have none and stay complete)
            if (tier ? (tiers == 0 || singles == nullptr || d > kTierMaxDepth) : (tiers == 1 && singles != nullptr && d <= kTierMaxDepth)) continue;
            const uint64_t single = tier ? std::min(singles[d], distinct[d]) : 0, entries = distinct[d] - single;
            const uint64_t needed = uint64_t(double(entries) / (tier ? sparse_tier_load(d) : sparse_load(d))) + 1;
            const uint64_t nb = tier ? sparse_tier_buckets(d, entries, single) : sparse_buckets_for(d, distinct[d]);
            const uint64_t least = tier ? sparse_tier_min_buckets(d) : sparse_min_buckets(d);
            const uint64_t lines = nb + kSparseMaxProbe;
            if (lines > 0xFFFFFFFFull) continue;
            // a table that the tags force to be far larger than its entries need is not worth its depth: 16 x for the cheap depths (a toy index
            // gets a toy table), 4 x for depth 29, whose least size is 69 GB (a human-scale index fills it to 72 % of the aimed load; a
            // chr20-sized one would fill 7 % and takes depth 27 in 4.3 GB instead)
            const uint64_t slack = d == 29 ? 4 : 16;
            const uint64_t data_needs = std::max<uint64_t>(needed, tier ? uint64_t(double(single) / kTierMaxSinglesPerBucket) + 1 : 0);
            if (!explicit_depth && least > std::max<uint64_t>(slack * data_needs, 65536)) continue;
            SparseChoice c;
            c.depth = d;
            c.nbuckets = nb;
            c.tier = tier != 0;
            c.bytes = lines * 128 + wide[d] * 16;
            c.build_bytes = c.bytes + lines * sizeof(uint32_t);
            if (c.build_bytes > avail) continue;
            return c;
        }
        if (explicit_depth) return none;
    }
    return none;
}

}  // namespace msbwt
