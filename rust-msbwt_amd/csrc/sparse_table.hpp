// Sparse suffix table: the ranges of the d-symbol suffixes that OCCUR, d up to 31 -- the reference's stubbed kmer_cache
// (src/msbwt_core.rs:133-146, src/rle_bwt.rs:332-346) taken past what a direct-address table can hold.
//
// The direct table (kernels.hpp, TableView) has 4^d entries whatever the data: at d = 17 that is 73 GB of which at most
// 17 % (a 30x human read set), 0.4 % (chr20-sized) or 0.02 % (C2) can be non-empty.  The same bytes spent on the suffixes
// that are PRESENT reach d = 23: three pair steps of seven fewer for every present 31-mer, and lines per query are the only
// lever the search has left in the HBM regime.  One lookup = ONE random 128-byte line, like the direct
// table's; a miss in this table -- it is complete: every d-mer that occurs has an entry -- is count 0, exactly the early
// exit of src/msbwt_core.rs:151-153.
//
// Layout: `nbuckets` lines of 128 bytes, 14 entries each, structure-of-arrays inside the line so that the lane that owns
// the query finds its entry with dword compares:
//     words  0..13   tag[i]  = low 24 bits of the mixed key | width << 24   (width 1..254; 255 = ESCAPE; 0 = empty slot)
//     words 14..27   l_lo[i] = low 32 bits of the range's l (ESCAPE: of the entry's index in the side array)
//     bytes 112..125 l_hi[i] = bits 32..39 of l
//     bytes 126..127 header  = how many entries wanted this bucket (saturating): > 14 = some were displaced to the next
//                              bucket(s), so a lookup that does not find its key here goes on (linear probing over buckets)
// Key = the table index of the direct table (A C G T -> 0..3, step t at bits [2t, 2t+2), step 0 = the k-mer's LAST symbol),
// n = 2 d bits.  It is mixed by a bijection of n-bit words (two odd multiplications and xor-shifts): bucket = the top 32
// bits of the mixed key scaled to [0, nbuckets) -- contiguous windows of the mixed key per bucket, at most
// W = 2^(n-32) * ceil(2^32 / nbuckets) wide -- and the low 24 bits are kept as the tag: within (probe + 1) * W <= 2^24
// consecutive values no two share their low 24 bits, so a tag match IS a key match (no false positives, nothing to verify).
// An entry whose range is 255 or more wide (a suffix of a high-copy repeat) names a flat {l, h} entry of 16 bytes in a side
// array: one more line for that query, like the escape lines of the packed direct table.
//
// Depths 25..29 (round 6: a present 31-mer behind a depth-27 table needs 3 lines instead of 5 -- measured with 27-mers on the
// depth-23 table: 1.46e10 q/s at human scale) -- a 50..56-bit key leaves 24-bit tags no room ((probe + 1) * W <= 2^24 with W =
// 2^(n-32) * 13 fails from n = 50 on), so these depths keep the low 32 bits of the mixed key as the tag: 10 bytes per entry, 12 per
// bucket ("wide" layout):
//     words  0..11   tag[i]   = low 32 bits of the mixed key
//     words 12..23   l_lo[i]
//     bytes  96..107 l_hi[i]
//     bytes 108..119 width[i]  (1..254; 255 = ESCAPE; 0 = empty slot)
//     bytes 126..127 header    (as above; > 12 = entries were displaced)
// Uniqueness: (probe + 1) * W <= 2^32.  Depth 29 (a 58-bit key: one pair step left of a 31-mer) is the last this tag width reaches:
// W <= 2^6 / 8 means 2^29 buckets at least -- 69 GB, which a human-scale index still has room for (5.5 entries per bucket) and a
// small one has no use for.
//
// Depths 30..31 ("xwide" layout: 40-bit tags, 11 entries of 11 bytes) -- at depth 31 the table's range IS the count of a 31-mer: one
// line per query (23-mers on the depth-23 table: 3.4e10 q/s at human scale):
//     words  0..10   tag_lo[i] = bits 0..31 of the mixed key
//     words 11..21   l_lo[i]
//     bytes  88..98  tag_hi[i] = bits 32..39 of the mixed key
//     bytes  99..109 l_hi[i]
//     bytes 110..120 width[i]
//     bytes 126..127 header    (> 11 = entries were displaced)
// Uniqueness: (probe + 1) * W <= 2^40 -- 2^25 buckets suffice again, and 7 entries per bucket (the same 64 %) make 18.3 bytes per
// distinct suffix: 54.5 GB for the 2.98e9 distinct 31-mers of the error-free human-scale index.
//
// TWO-TIER form (round 6; depths up to 29): for read sets WITH errors.  The complete table's size follows the distinct suffixes, and
// on reads with substitutions most of those are error k-mers that occur ONCE (30x human reads, 0.5 % errors: 1.3e10 distinct 23-mers
// of which about 3e9 -- the genome's -- occur more than once: 185 GB complete).  The two-tier table keeps an ENTRY only for the
// suffixes whose range is at least 2 wide ("solid") and, for the ones that occur once, sets 4 bits of one of the 8 FILTER words of
// their own bucket line (a blocked Bloom filter inside the line the lookup fetches anyway: no false negatives).  A lookup that finds
// its tag is served as before; one that finds neither its tag nor its filter bits is count 0, exactly as in the complete table (every
// suffix that occurs is either an entry or in the filter); one whose filter bits are set (a suffix that occurs once -- or a false
// positive, about 1 % at 21 such suffixes per bucket) continues through the DIRECT table and the search: the reference's own path
// (src/rle_bwt.rs:202-287), so the count stays exact whatever the filter says.  Layout (24-bit tags, depths up to 24): 10 entries
//     words  0..9    tag[i] | width << 24      (width 2..254; 255 = ESCAPE; 0 = empty slot)
//     words 10..19   l_lo[i]
//     bytes 80..89   l_hi[i]
//     bytes 90..91   header (entries that wanted this bucket; > 10 = some were displaced)
//     words 23..30   the filter: key -> one word (3 hash bits) and 4 bits in it (4 x 5 hash bits)
// and with 32-bit tags (depths 25..29): 9 entries -- tags in words 0..8, l_lo in words 9..17, l_hi in bytes 72..80, widths in bytes
// 81..89, header and filter as above.  Size: solid / 5.8 (5.0) buckets, at least singles / 32 (the filter's load): the 30x human
// read set with errors keeps depth 23 in about 66 GB (5.8 entries per bucket: sparse_policy.hpp, sparse_tier_load).
#pragma once
#include <cstdint>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define MSBWT_HD __host__ __device__ __forceinline__
#else
#define MSBWT_HD inline
#endif

namespace msbwt {

constexpr uint32_t kSparseSlots = 14;        // entries per 128-byte bucket (depths up to 24)
constexpr uint32_t kSparseTagBits = 24;
constexpr uint32_t kSparseWideSlots = 12;    // ... and of the wide layout (depths 25..29): 32-bit tags
constexpr uint32_t kSparseWideL0Word = 12, kSparseWideHiByte = 96, kSparseWideWidthByte = 108;
constexpr int kSparseWideFrom = 25;
constexpr uint32_t kSparseXSlots = 11;       // ... and of the xwide layout (depths 30..31): 40-bit tags
constexpr uint32_t kSparseXL0Word = 11, kSparseXTagHiByte = 88, kSparseXHiByte = 99, kSparseXWidthByte = 110;
constexpr int kSparseXFrom = 30;
// two-tier form: entries per bucket (24-bit / 32-bit tags), where its fields sit, the filter
constexpr uint32_t kTierSlots = 10, kTierL0Word = 10, kTierHiByte = 80;
constexpr uint32_t kTierWideSlots = 9, kTierWideL0Word = 9, kTierWideHiByte = 72, kTierWideWidthByte = 81;
constexpr uint32_t kTierHeaderByte = 90, kTierFilterWord = 23, kTierFilterWords = 8;
constexpr int kTierMaxDepth = 29;                 // (the 40-bit-tag layout of depths 30..31 has no two-tier form)
constexpr double kTierMaxSinglesPerBucket = 32.0; // filter load the builder accepts: 4 keys per 32-bit word, 2.5 % false positives
constexpr uint32_t kSparseEscapeWidth = 255;  // width field of an entry whose range lives in the side array
constexpr uint32_t kSparseMaxProbe = 15;      // a key lives at most this many buckets behind its own
constexpr int kSparseMinDepth = 16, kSparseMaxDepth = 31;
constexpr int kSparseAutoDepth = 23;          // the automatic choice never goes deeper (msbwt_rle_set_sparse_table takes 16..31)
constexpr double kSparseLoad = 9.0;           // entries per bucket the builder aims for (64 % of the slots: 0.9 % of the entries displaced)

MSBWT_HD bool sparse_wide(uint32_t depth) { return depth >= uint32_t(kSparseWideFrom); }   // (xwide included: the tag's low word is whole)
MSBWT_HD bool sparse_xwide(uint32_t depth) { return depth >= uint32_t(kSparseXFrom); }
MSBWT_HD uint32_t sparse_slots(uint32_t depth, bool tier = false) {
    if (tier) return sparse_wide(depth) ? kTierWideSlots : kTierSlots;
    return sparse_xwide(depth) ? kSparseXSlots : sparse_wide(depth) ? kSparseWideSlots : kSparseSlots;
}
// two-tier filter: the word (0..7) and the four bits of a key, from its tag (unique within a bucket's probe window, so two keys of a
// bucket never share all of it)
MSBWT_HD uint32_t sparse_filter_hash(uint32_t tag) {
    uint32_t f = tag * 0x9E3779B1u;
    f ^= f >> 15;
    f *= 0x85EBCA77u;
    f ^= f >> 13;
    return f;
}
MSBWT_HD uint32_t sparse_filter_word(uint32_t f) { return f >> 29; }
MSBWT_HD uint32_t sparse_filter_mask(uint32_t f) { return (1u << (f & 31u)) | (1u << ((f >> 5) & 31u)) | (1u << ((f >> 10) & 31u)) | (1u << ((f >> 15) & 31u)); }
MSBWT_HD uint32_t sparse_tag_bits(uint32_t depth) { return sparse_xwide(depth) ? 40u : sparse_wide(depth) ? 32u : kSparseTagBits; }
inline double sparse_load(int depth, bool tier = false) { return kSparseLoad * double(sparse_slots(uint32_t(depth), tier)) / double(kSparseSlots); }  // the same 64 % of the slots
constexpr uint32_t kSparseL0Word = 14, kSparseHiByte = 112, kSparseHeaderByte = 126;

struct SparseView {
    const void *lines = nullptr;   // nbuckets x 128 bytes, or nullptr: no sparse table
    uint32_t nbuckets = 0;
    uint32_t depth = 0;            // symbols an entry stands for
    uint32_t probe = 0;            // buckets a lookup may go beyond its own
    const void *side = nullptr;    // 16-byte {l, h} entries of the ESCAPE entries
    uint32_t tier = 0;             // 1 = two-tier form: entries for the suffixes at least 2 wide, filter bits for the ones that occur once
};

// the bijection of n-bit words (n = 2 depth, 32 <= n <= 62)
MSBWT_HD uint64_t sparse_mix(uint64_t key, uint32_t n) {
    const uint64_t mask = (uint64_t(1) << n) - 1u;
    uint64_t x = key & mask;
    x = (x * 0x9E3779B97F4A7C15ull) & mask;
    x ^= x >> (n >> 1);
    x = (x * 0xD6E8FEB86659FD93ull) & mask;
    x ^= x >> (n >> 1);
    return x;
}

// bucket of a mixed key: its top 32 bits scaled to [0, nbuckets)
MSBWT_HD uint32_t sparse_bucket(uint64_t mixed, uint32_t n, uint32_t nbuckets) {
    const uint32_t top = uint32_t(mixed >> (n - 32u));
    return uint32_t((uint64_t(top) * nbuckets) >> 32);
}

MSBWT_HD uint32_t sparse_tag(uint64_t mixed, uint32_t depth) { return sparse_wide(depth) ? uint32_t(mixed) : uint32_t(mixed) & ((1u << kSparseTagBits) - 1u); }
// bits 32..39 of the tag (xwide layout; 0 otherwise)
MSBWT_HD uint32_t sparse_tag_hi(uint64_t mixed, uint32_t depth) { return sparse_xwide(depth) ? uint32_t(mixed >> 32) & 0xFFu : 0u; }

// How far a lookup may probe with `nbuckets` buckets at depth d so that tags stay unambiguous: (probe + 1) * W <= 2^tagbits,
// W = 2^(n-32) * ceil(2^32 / nbuckets).  Negative: this many buckets are too few for the depth.
inline int sparse_probe_limit(int depth, uint64_t nbuckets) {
    if (depth < kSparseMinDepth || depth > kSparseMaxDepth || nbuckets == 0 || nbuckets > 0xFFFFFFFFull) return -1;
    const uint64_t per_top = ((uint64_t(1) << 32) + nbuckets - 1) / nbuckets;
    const uint64_t window = per_top << (2 * depth - 32);
    const uint64_t fit = (uint64_t(1) << sparse_tag_bits(uint32_t(depth))) / window;  // windows that fit the tag space
    if (fit < 2) return -1;
    return int(fit - 1 < kSparseMaxProbe ? fit - 1 : kSparseMaxProbe);
}

// fewest buckets a table of this depth may have: a probe limit of at least 7, so that at the builder's load practically no
// entry is displaced beyond it (with 4 -- what 9 entries per bucket leave a chr20-sized index at depth 23 -- about one entry in
// 10^5 found no slot and the whole table was filled a second time with more buckets: round 5's first C4 builds)
inline uint64_t sparse_min_buckets(int depth) {
    const int bits = int(sparse_tag_bits(uint32_t(depth)));
    const int shift = 2 * depth - 32 + 3;  // 8 W <= 2^bits  <=>  ceil(2^32 / nb) <= 2^(bits - 3 - (n - 32))
    if (shift >= bits) return ~uint64_t(0);
    if (bits - shift >= 32) return 1;     // (shallow depths with wide tags: any bucket count will do)
    const uint64_t per_top = uint64_t(1) << (bits - shift);  // allowed ceil(2^32 / nb)
    return ((uint64_t(1) << 32) + per_top - 1) / per_top;
}

// buckets for `entries` entries at depth d (the load the builder aims for, within what the tags allow)
inline uint64_t sparse_buckets_for(int depth, uint64_t entries, double load = 0.0) {
    const uint64_t want = uint64_t(double(entries) / (load > 0.0 ? load : sparse_load(depth))) + 1;
    const uint64_t least = sparse_min_buckets(depth);
    return want > least ? want : least;
}
// ... of the two-tier form: `solid` entries at its load, and room in the filters for `singles` suffixes that occur once
inline uint64_t sparse_tier_buckets_for(int depth, uint64_t solid, uint64_t singles) {
    const uint64_t by_entries = sparse_buckets_for(depth, solid, sparse_load(depth, true));
    const uint64_t by_filter = uint64_t(double(singles) / kTierMaxSinglesPerBucket) + 1;
    return by_entries > by_filter ? by_entries : by_filter;
}

}  // namespace msbwt
