// (lanes_kernel.hpp: the kernel template and its launch helpers, compiled into four translation units -- lanes.hip: direct table and the
// 24-bit-tag sparse table; lanes_wide.hip, lanes_xwide.hip: the 32- and 40-bit-tag layouts; lanes_tier.hip: the two-tier form -- so that
// every kernel carries ONE table scan and the 80-odd instantiations compile side by side)
// gfx950 (MI355X, CDNA4): count_kmers with ONE QUERY PER LANE and the index lines staged through
// LDS -- the search kernel for 6 <= k <= 64 on an index with pair blocks (kernels.hip, long_search);
// throughput is set by how many random 128-byte lines are in flight.
//
// The 8-lanes-per-query kernel (kernels.hip) keeps 8 queries x 1-2 lines in flight per wave and
// pays the whole instruction stream of a step once per 8 queries.  Here a wave is persistent,
// owns 64 queries at a time and runs them in lock step; one search step of the wave is
//
//   1. every busy lane names the 128-byte block(s) of its two range bounds: the block of l, and
//      the block of h when that is a different one (narrow ranges mostly share a block); the
//      second blocks are compacted with a wave ballot + prefix count into a line list in LDS;
//   2. the wave fetches every listed line with coalesced LDS-DMA loads (global_load_lds_dwordx4:
//      8 lanes x 16 B = one line, eight lines per instruction, no VGPRs spent on data in flight)
//      -- 64..80 lines in flight per wave (64..112 for k > 32) instead of 8..16;
//   3. after s_waitcnt vmcnt(0) every lane reads its own line(s) back from LDS (bank-conflict
//      free, see line_base) and ranks both bounds itself: XOR / AND / popcount on the bit planes,
//      no cross-lane reduction at all.
//
// Lanes whose query is finished take the next undecided query from a ring in LDS that the setup
// code (stage the tile's bytes, validate, pack, suffix-table lookup -- the pieces of
// search_common.hpp) keeps topped up, so all 64 lanes stay busy whatever the mix of early exits.
// Setup never waits for memory by itself: a tile's bytes are fetched an iteration early and its
// table entries ride along with the next search step's lines; while lanes are idle and a tile is on
// its way, its survivors are fetched before a search step is spent.  Tiles are dealt out by atomic
// tickets (sharded counters), so slow waves simply take fewer.  Counts go straight to the caller's
// buffer.  Block layouts: plane_index.hpp, rank_ops.hpp.  One wave per workgroup, 12.1 KiB of LDS
// (k <= 32) or 12.9 KiB (k <= 64) each: 12 waves per CU either way (kRegionsFor, resident_waves).
// Fused query preparation (kReads): the bytes a tile of consecutive read windows spans are converted
// to symbol codes ONCE per read symbol and staged in LDS, forward and reverse-complemented, so that a
// window is packed exactly like a row of a query matrix (stage_read_span).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"
#include "rank_ops.hpp"
#include "search_common.hpp"
#include "sparse_table.hpp"

namespace msbwt {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_void;

constexpr int kRing = 64;        // undecided queries waiting for a lane: one tile's worth (the next tile waits in registers)
// LDS-DMA regions of 8 lines per wave: 8 for the 64 first-bound lines + the rest for second-bound
// lines.  Fewer regions = less LDS = more resident waves, and lines in flight per CU are what the
// throughput follows: measured at human scale (tools/sweep_variants.sh), 14 regions / 8 waves per CU
// 5.05, 12 / 10 4.88 (uneven SIMDs), 10 / 8 4.77, 10 / 12 5.32 x 10^9 q/s -- sixteen second-line slots
// cost 7 % (the queries of a tile take their first, widest step together), four more waves bring 13 %.
// Round 3: ring entries shrank to 24 / 36 bytes (RingItemT) and the line list moved INTO the line area,
// so searches of k > 32 fit the same 10 regions and 12 waves (they had 14 regions and 8 waves).
// Tried in round 3 and dropped: no second-line slots at all -- a range that straddles two lines takes TWO
// iterations (the line of l, then the line of h): 8 regions, 170 instead of 260 VALU per wave step, 120 VGPRs.
// Human scale 4.65 against 5.16 x 10^9 q/s, C4 read-derived 5.57 against 5.89, C3 fused 9.16 against 9.02: the
// lanes of a tile fall out of step (the refill code then runs every iteration instead of once per tile) and a
// straddling lane holds its slot for two memory round trips.
#ifdef MSBWT_LANES_REGIONS  // experiments
template <int kWords> constexpr int kRegionsFor = MSBWT_LANES_REGIONS;
#else
template <int kWords> constexpr int kRegionsFor = 10;
#endif
#ifndef MSBWT_LANES_WAVE_CAP
#define MSBWT_LANES_WAVE_CAP 12
#endif
// experiments (tools/build_variant.sh): what the side-array fetch and the optional counters cost the kernels that never see either
#ifdef MSBWT_LANES_NO_SIDE_FETCH
constexpr bool kSideFetch = false;
#else
constexpr bool kSideFetch = true;
#endif
#ifdef MSBWT_LANES_NO_COUNTERS
constexpr bool kCounting = false;
#else
constexpr bool kCounting = true;
#endif

// One undecided query waiting in the ring: 24 bytes (k <= 32) or 36 (k <= 64).  The ring only ever holds
// queries of ONE tile (it is refilled when empty), so the tile index is a wave-uniform register, not a field.
// kPlaced (the instantiations for packed queries, which may come with a place for every count): 4 bytes more.
// kTier (the instantiations for the two-tier sparse table): 4 bytes more -- the query's index into the DIRECT table, should its lookup
// end in the filter.
template <int kWords, bool kPlaced, bool kTier>
struct RingItemT {
    uint32_t l_lo, h_lo;
    uint32_t meta;        // l >> 32 (8 bits) | h >> 32 (8 bits) << 8 | remaining steps << 16 | lane in the tile << 24
    uint32_t w[kWords];   // remaining symbols, 3 bits each, next step in the low bits
    uint32_t extra[(kPlaced ? 1 : 0) + (kTier ? 1 : 0)];  // [0] kPlaced: QuerySource::out_index / place_inline, where the count goes; [last] kTier: direct-table index
    static constexpr int kOutAt = 0, kDkeyAt = kPlaced ? 1 : 0;
};

template <int kWords, bool kPlaced, bool kTier = false>
struct LaneScratchT {
    static constexpr int kMaxK = kWords * 32 / 3;  // 32 or 64
    static constexpr int kRegions = kRegionsFor<kWords>;
    static constexpr int kLineSlots = kRegions * 8;
    static constexpr uint32_t kMaxSecond = uint32_t(kLineSlots) - 64u;  // lanes beyond that with a second line sit the step out
    static_assert(kRegions >= 10 && kRegions % 2 == 0, "first-bound lines take 8 regions; regions come in padded pairs");
    // Region i (one LDS-DMA instruction: lane j writes 16 bytes at 16 j) starts at uint4 index
    // region_base(i): every odd region is pushed 128 bytes further, so that the 64 lanes'
    // read-back of "chunk j of my line" touches every bank exactly once per 16 lanes (lanes 16 m ..
    // 16 m + 15 own the lines of regions 2 m and 2 m + 1, whose bank phases differ by 128 bytes).
    // Between two steps the same memory stages the tile's query bytes (2 or 4 KiB; reads mode: 512 B of
    // symbol codes), and at the start of a step its first 640 bytes hold the step's line addresses: they
    // are in registers (s_waitcnt lgkmcnt(0)) before the first LDS-DMA load is issued.
    uint4 lines[(kRegions / 2) * 136];   // 10.6 KiB
    RingItemT<kWords, kPlaced, kTier> ring[kRing];  // 1.5 or 2.25 KiB (+ 256 bytes when kPlaced, + 256 when kTier)
    uint32_t cnt[kSearchCounters];       // optional search counters of this wave (kernels.hpp): in LDS, so that they cost no registers
};
static_assert(sizeof(LaneScratchT<6, true>) <= 160 * 1024 / 12 && sizeof(LaneScratchT<6, false, true>) <= 160 * 1024 / 12 && sizeof(LaneScratchT<3, true, true>) <= 160 * 1024 / 12,
              "12 one-wave workgroups per CU (but for packed queries of k > 32 on a two-tier table: 8)");

// uint4 index of the first of region i's 64 sixteen-byte pieces: pairs of regions take 136 pieces,
// the odd one starting 72 in (64 + 8 of padding)
__host__ __device__ constexpr uint32_t region_base(uint32_t i) { return 136u * (i >> 1) + 72u * (i & 1u); }

// uint4 index of chunk 0 of the line in slot `s` (chunk j sits at line_base(s) + (j ^ (s & 7)))
__device__ __forceinline__ uint32_t line_base(uint32_t s) { return region_base(s >> 3) + 8u * (s & 7u); }

// (a ^ b) & c in one v_bitop3_b32 (truth-table index = 4 a + 2 b + c)
__device__ __forceinline__ uint32_t xor_and(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x28); }

// ---- rank of one bound from a staged plane-block line (single-symbol step) ----------------------
struct PlaneLine {
    uint32_t m[8];     // per 32 positions: 1 where the symbol matches
    uint32_t meta[8];  // header words
};

__device__ __forceinline__ void read_plane_line(const uint4 *lines, uint32_t slot, uint32_t s, PlaneLine &L) {
    const uint32_t base = line_base(slot), g = slot & 7u;
    const uint32_t x0 = (s & 1u) ? 0u : ~0u, x1 = (s & 2u) ? 0u : ~0u, x2 = (s & 4u) ? 0u : ~0u;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const uint4 c = lines[base + (j ^ g)];
        L.m[j] = xor_and(c.z, x2, xor_and(c.y, x1, c.x ^ x0));
        L.meta[j] = c.w;
    }
}

// start_index[s] + rank(s, pos) from the line of pos's block (plane_index.hpp layout)
__device__ __forceinline__ uint64_t plane_line_bound(const PlaneLine &L, uint32_t s, uint64_t pos) {
    const uint32_t r = uint32_t(pos) & 255u, idx = r >> 6;
    const uint64_t t = (1ull << (r & 63u)) - 1ull;  // the 64-position word that holds r keeps its low r % 64 bits
    uint32_t cnt = 0, lo = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint64_t mask = j < idx ? ~0ull : (j == idx ? t : 0ull);
        cnt += uint32_t(__popc(L.m[2 * j] & uint32_t(mask))) + uint32_t(__popc(L.m[2 * j + 1] & uint32_t(mask >> 32)));
    }
#pragma unroll
    for (uint32_t j = 0; j < 6; ++j) lo = (s == j) ? L.meta[j] : lo;
    const uint32_t hi = (((s >> 2) ? L.meta[7] : L.meta[6]) >> ((s & 3u) * 8u)) & 0xFFu;
    return ((uint64_t(hi) << 32) | lo) + cnt;
}

// ---- ranks from a staged RUN-block line (run_index.hpp: header + 96 one-byte runs for 512 positions) ----------------------------
// The lane decodes its own line.  FIRST bound (run_line_first): per dword of four runs, the lengths (x >> 3 & 31 per byte), a byte
// mask of the runs whose symbol is s, and two sums of absolute differences -- all four lengths, and the matching ones; a dword that
// ends at or before the position adds its matching lengths whole, the first dword that reaches beyond it is kept and finished run
// by run afterwards.  It also notes where it stood (RunCont: that dword, its first position, the matches before it).
// SECOND bound (run_line_continue, round 5): a range is narrow, so h lies a dword or two behind l -- its rank CONTINUES from where l's
// decode stood (same run block: 94 % of the steps) or, in another block, starts at that block's first dword; either way a short
// lane-varying loop over the few dwords up to h instead of a second decode of all 24 (round 4 decoded twice: 1.82e9 q/s at human
// scale; one decode for both bounds inside one block had been tried and lost to the lane-varying select, 1.59e9).
// An OVERFLOW block (more than 96 pieces) holds no runs: *need receives the 1-based number of the plane block, in the side array,
// that holds the position (the caller fetches that line in its next iteration).
struct RunCont {
    uint32_t dword, at, acc;  // first dword (0..23; 24: none) that reaches beyond the first bound, its first position, matches before it
};

__device__ __forceinline__ void run_dword(uint32_t x, uint32_t sx, uint32_t &all, uint32_t &mine) {
    const uint32_t lens = (x >> 3) & 0x1F1F1F1Fu, t = (x & 0x07070707u) ^ sx;
    const uint32_t other = ((t + 0x7F7F7F7Fu) | t) & 0x80808080u;        // bit 7 of a byte: its run's symbol is NOT s
    const uint32_t match = ((other ^ 0x80808080u) >> 7) * 0xFFu;          // 0xFF in the bytes of the runs of s
    all = __builtin_amdgcn_sad_u8(lens, 0u, 0u);
    mine = __builtin_amdgcn_sad_u8(lens & match, 0u, 0u);
}

// matches among the positions [at, r) of the four runs of dword x that starts at position `at`
__device__ __forceinline__ uint32_t run_dword_clip(uint32_t x, uint32_t s, uint32_t at, uint32_t r) {
    uint32_t cnt = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t run = (x >> (8 * b)) & 0xFFu, len = run >> 3;
        cnt += (run & 7u) == s ? uint32_t(min(max(int(r) - int(at), 0), int(len))) : 0u;
        at += len;
    }
    return cnt;
}

__device__ __forceinline__ void run_line_first(const uint4 *lines, uint32_t slot, uint32_t s, uint64_t pos, uint64_t &out, uint32_t &need, RunCont &cont) {
    const uint32_t base = line_base(slot), g = slot & 7u;
    const uint4 c0 = lines[base + (0u ^ g)], c1 = lines[base + (1u ^ g)];
    cont = RunCont{24u, 0u, 0u};
    if ((c1.w & 0x80000000u) != 0u) {
        need = lines[base + (2u ^ g)].x * 2u + 1u + ((uint32_t(pos) & 511u) >> 8);
        return;
    }
    const uint32_t lo = s == 0u ? c0.x : s == 1u ? c0.y : s == 2u ? c0.z : s == 3u ? c0.w : s == 4u ? c1.x : c1.y;
    const uint32_t hi = (((s >> 2) ? c1.w : c1.z) >> ((s & 3u) * 8u)) & 0xFFu;
    const uint32_t r0 = uint32_t(pos) & 511u, sx = s * 0x01010101u;
    uint32_t cnt = 0, cur = 0, str = 0;
#pragma unroll
    for (uint32_t j = 2; j < 8; ++j) {
        const uint4 c = lines[base + (j ^ g)];
        const uint32_t word[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            uint32_t all, mine;
            run_dword(word[i], sx, all, mine);
            const uint32_t end = cur + all;
            cnt += end <= r0 ? mine : 0u;
            const bool here = end > r0 && cont.dword == 24u;  // the first dword that reaches beyond the position
            str = here ? word[i] : str;
            cont.at = here ? cur : cont.at;
            cont.dword = here ? 4u * (j - 2u) + i : cont.dword;
            cur = end;
        }
    }
    cont.acc = cnt;
    out = ((uint64_t(hi) << 32) | lo) + cnt + run_dword_clip(str, s, cont.at, r0);  // (no such dword: four empty runs, nothing added)
}

// same_line: `slot` is the line run_line_first decoded and pos lies at or behind its position (cont says where it stood);
// otherwise the rank starts at the line's first dword
__device__ __forceinline__ void run_line_continue(const uint4 *lines, uint32_t slot, uint32_t s, uint64_t pos, bool same_line, const RunCont &cont, uint64_t &out,
                                                  uint32_t &need) {
    const uint32_t base = line_base(slot), g = slot & 7u;
    const uint4 c0 = lines[base + (0u ^ g)], c1 = lines[base + (1u ^ g)];
    if ((c1.w & 0x80000000u) != 0u) {
        need = lines[base + (2u ^ g)].x * 2u + 1u + ((uint32_t(pos) & 511u) >> 8);
        return;
    }
    const uint32_t lo = s == 0u ? c0.x : s == 1u ? c0.y : s == 2u ? c0.z : s == 3u ? c0.w : s == 4u ? c1.x : c1.y;
    const uint32_t hi = (((s >> 2) ? c1.w : c1.z) >> ((s & 3u) * 8u)) & 0xFFu;
    const uint32_t r1 = uint32_t(pos) & 511u, sx = s * 0x01010101u;
    uint32_t d = same_line ? cont.dword : 0u, cur = same_line ? cont.at : 0u, acc = same_line ? cont.acc : 0u;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(lines);
    while (d < 24u) {  // lane-varying, a dword or two for a narrow range
        const uint32_t x = words[(base + ((2u + (d >> 2)) ^ g)) * 4u + (d & 3u)];
        uint32_t all, mine;
        run_dword(x, sx, all, mine);
        if (cur + all <= r1) {
            acc += mine;
            cur += all;
            ++d;
        } else {
            acc += run_dword_clip(x, s, cur, r1);
            break;
        }
    }
    out = ((uint64_t(hi) << 32) | lo) + acc;
}

// ---- rank of one bound from a staged pair-block line (two-symbol step, rank_ops.hpp layout) -----
struct PairLine {
    uint32_t m[4];   // per 32 positions: 1 where (S, S2) == (a, b)
    uint32_t field;  // the block's 24-bit header count of the pair
};

__device__ __forceinline__ void read_pair_line(const uint4 *lines, uint32_t slot, uint32_t a2, uint32_t b2, PairLine &L) {
    const uint32_t base = line_base(slot), g = slot & 7u, p = a2 * 4u + b2;
    const uint4 a0 = lines[base + (0u ^ g)], a1 = lines[base + (1u ^ g)], b0 = lines[base + (2u ^ g)], b1 = lines[base + (3u ^ g)];
    const uint4 v = lines[base + (uint32_t(kPairValidChunk) ^ g)];
    // a plane word matches where it equals the wanted code bit: XOR with all-ones when that bit is 0
    const uint32_t na0 = (a2 & 1u) - 1u, na1 = ((a2 >> 1) & 1u) - 1u, nb0 = (b2 & 1u) - 1u, nb1 = ((b2 >> 1) & 1u) - 1u;
    L.m[0] = xor_and(b1.x, nb1, xor_and(b0.x, nb0, xor_and(a1.x, na1, xor_and(a0.x, na0, v.x))));
    L.m[1] = xor_and(b1.y, nb1, xor_and(b0.y, nb0, xor_and(a1.y, na1, xor_and(a0.y, na0, v.y))));
    L.m[2] = xor_and(b1.z, nb1, xor_and(b0.z, nb0, xor_and(a1.z, na1, xor_and(a0.z, na0, v.z))));
    L.m[3] = xor_and(b1.w, nb1, xor_and(b0.w, nb0, xor_and(a1.w, na1, xor_and(a0.w, na0, v.w))));
    // header: u16 low half in chunk 5/6, u8 high byte in chunk 7
    const uint16_t *halves = reinterpret_cast<const uint16_t *>(lines + base + ((uint32_t(kPairLoChunk) + (p >> 3)) ^ g));
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(lines + base + (uint32_t(kPairHiChunk) ^ g));
    L.field = uint32_t(halves[p & 7u]) | (uint32_t(bytes[p]) << 16);
}

// matches among the first r (0..127) positions of the block (branch-free: a 128-bit low mask)
__device__ __forceinline__ uint32_t pair_line_count(const PairLine &L, uint32_t r) {
    const uint64_t t = (1ull << (r & 63u)) - 1ull;
    const bool upper = r >= 64u;
    const uint64_t lo = upper ? ~0ull : t, hi = upper ? t : 0ull;
    return uint32_t(__popc(L.m[0] & uint32_t(lo))) + uint32_t(__popc(L.m[1] & uint32_t(lo >> 32))) +
           uint32_t(__popc(L.m[2] & uint32_t(hi))) + uint32_t(__popc(L.m[3] & uint32_t(hi >> 32)));
}

__device__ __forceinline__ uint64_t pair_line_bound(const PairLine &L, uint64_t super_base, uint32_t r) {  // r = position - block start
    return super_base + L.field + pair_line_count(L, r);
}

// ---- sparse suffix table (sparse_table.hpp): this lane's key among the 14 entries of a staged bucket line -------------------------
// The line: 14 tags in words 0..13 (chunks 0-3), l in words 14..27 and bytes 112..125, the bucket's header in bytes 126..127.  A
// key sits in at most one slot of at most one bucket, so a tag match IS the entry.  -> true: l and width (255 = the range lives
// in the side array, l = its index there); false: not in this bucket -- header > 14 says that entries of it were displaced.
__device__ __forceinline__ bool sparse_scan(const uint4 *lines, uint32_t slot, uint32_t want, uint64_t &l, uint32_t &width, uint32_t &header) {
    const uint32_t base = line_base(slot), g = slot & 7u;
    const uint4 c0 = lines[base + (0u ^ g)], c1 = lines[base + (1u ^ g)], c2 = lines[base + (2u ^ g)], c3 = lines[base + (3u ^ g)];
    const uint32_t tags[kSparseSlots] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y};
    // Three instructions per slot (and, compare, select), from the highest slot down so that the LOWEST matching slot wins: a
    // bucket fills from slot 0 up, so an entry always beats an empty slot (tag 0, width 0), which a key whose tag is 0 matches too.
    uint32_t hit = kSparseSlots;
#pragma unroll
    for (int i = int(kSparseSlots) - 1; i >= 0; --i) hit = (tags[i] & ((1u << kSparseTagBits) - 1u)) == want ? uint32_t(i) : hit;
    header = lines[base + (7u ^ g)].w >> 16;
    width = 0;
    if (hit >= kSparseSlots) return false;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(lines);
    width = words[(base + ((hit >> 2) ^ g)) * 4u + (hit & 3u)] >> kSparseTagBits;
    if (width == 0u) return false;  // an empty slot: the key is not in this bucket (and none was displaced from a bucket with room)
    const uint32_t word = kSparseL0Word + hit;
    const uint32_t lo = words[(base + ((word >> 2) ^ g)) * 4u + (word & 3u)];
    const uint32_t hi = reinterpret_cast<const uint8_t *>(lines)[(base + (7u ^ g)) * 16u + hit];
    l = (uint64_t(hi) << 32) | lo;
    return true;
}

// The WIDE layout of depths 25..28 (sparse_table.hpp): 12 entries, the tag is the whole word, the width a byte of its own.
__device__ __forceinline__ bool sparse_scan_wide(const uint4 *lines, uint32_t slot, uint32_t want, uint64_t &l, uint32_t &width, uint32_t &header) {
    const uint32_t base = line_base(slot), g = slot & 7u;
    const uint4 c0 = lines[base + (0u ^ g)], c1 = lines[base + (1u ^ g)], c2 = lines[base + (2u ^ g)];
    const uint32_t tags[kSparseWideSlots] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w};
    uint32_t hit = kSparseWideSlots;
#pragma unroll
    for (int i = int(kSparseWideSlots) - 1; i >= 0; --i) hit = tags[i] == want ? uint32_t(i) : hit;  // the lowest matching slot: an entry beats an empty slot
    header = lines[base + (7u ^ g)].w >> 16;
    width = 0;
    if (hit >= kSparseWideSlots) return false;
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(lines);
    const uint32_t wb = kSparseWideWidthByte + hit, hb = kSparseWideHiByte + hit;
    width = bytes[(base + ((wb >> 4) ^ g)) * 16u + (wb & 15u)];
    if (width == 0u) return false;  // an empty slot (a key whose tag is 0 matches it)
    const uint32_t *words = reinterpret_cast<const uint32_t *>(lines);
    const uint32_t word = kSparseWideL0Word + hit;
    const uint32_t lo = words[(base + ((word >> 2) ^ g)) * 4u + (word & 3u)];
    const uint32_t hi = bytes[(base + ((hb >> 4) ^ g)) * 16u + (hb & 15u)];
    l = (uint64_t(hi) << 32) | lo;
    return true;
}

// The XWIDE layout of depths 30..31: 11 entries, 40-bit tags (a low word and a byte).  The low words alone nearly always decide; the
// candidate's high byte is checked, and should it differ the scan goes on behind it (two keys of one probe window may share their low
// 32 bits -- never all 40).
__device__ __forceinline__ bool sparse_scan_xwide(const uint4 *lines, uint32_t slot, uint64_t want, uint64_t &l, uint32_t &width, uint32_t &header) {
    const uint32_t base = line_base(slot), g = slot & 7u;
    const uint4 c0 = lines[base + (0u ^ g)], c1 = lines[base + (1u ^ g)], c2 = lines[base + (2u ^ g)];
    const uint32_t tags[kSparseXSlots] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z};
    const uint32_t want_lo = uint32_t(want), want_hi = uint32_t(want >> 32) & 0xFFu;
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(lines);
    const uint32_t *words = reinterpret_cast<const uint32_t *>(lines);
    auto byte_at = [&](uint32_t b) -> uint32_t { return bytes[(base + ((b >> 4) ^ g)) * 16u + (b & 15u)]; };
    header = lines[base + (7u ^ g)].w >> 16;
    width = 0;
    uint32_t cand = 0;  // bit i: slot i's low word matches (the words need not stay in registers beyond this)
#pragma unroll
    for (uint32_t i = 0; i < kSparseXSlots; ++i) cand |= (tags[i] == want_lo ? 1u : 0u) << i;
    while (cand != 0u) {  // lowest candidate first: one round, but for low words shared by chance
        const uint32_t hit = uint32_t(__builtin_ctz(cand));
        const uint32_t w = byte_at(kSparseXWidthByte + hit);
        if (w == 0u) return false;  // an empty slot whose zero low word matched: a bucket fills from slot 0 up, nothing lies behind it
        if (byte_at(kSparseXTagHiByte + hit) == want_hi) {
            width = w;
            const uint32_t word = kSparseXL0Word + hit;
            const uint32_t lo = words[(base + ((word >> 2) ^ g)) * 4u + (word & 3u)];
            l = (uint64_t(byte_at(kSparseXHiByte + hit)) << 32) | lo;
            return true;
        }
        cand &= cand - 1u;  // an entry that shares the low word only
    }
    return false;
}

// The TWO-TIER forms (sparse_table.hpp): 10 entries with 24-bit tags or 9 with 32-bit tags, the bucket's header in bytes 90..91 and eight
// filter words behind it.  -> also `maybe`: the four filter bits of this key are all set in THIS bucket's filter (meaningful in the key's
// own bucket: a suffix that occurs once has no entry anywhere, only those bits).
__device__ __forceinline__ bool sparse_scan_tier(const uint4 *lines, uint32_t slot, uint32_t want, bool wide_layout, uint64_t &l, uint32_t &width, uint32_t &header, bool &maybe) {
    const uint32_t base = line_base(slot), g = slot & 7u;
    const uint4 c0 = lines[base + (0u ^ g)], c1 = lines[base + (1u ^ g)], c2 = lines[base + (2u ^ g)];
    const uint32_t tags[kTierSlots] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y};
    const uint32_t *words = reinterpret_cast<const uint32_t *>(lines);
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(lines);
    auto word_at = [&](uint32_t w) -> uint32_t { return words[(base + ((w >> 2) ^ g)) * 4u + (w & 3u)]; };
    auto byte_at = [&](uint32_t b) -> uint32_t { return bytes[(base + ((b >> 4) ^ g)) * 16u + (b & 15u)]; };
    const uint32_t nslots = wide_layout ? kTierWideSlots : kTierSlots, tag_mask = wide_layout ? ~0u : ((1u << kSparseTagBits) - 1u);
    uint32_t hit = kTierSlots;
#pragma unroll
    for (int i = int(kTierSlots) - 1; i >= 0; --i) hit = ((tags[i] & tag_mask) == want && uint32_t(i) < nslots) ? uint32_t(i) : hit;  // the lowest matching slot
    header = word_at(kTierHeaderByte / 4u) >> 16;
    const uint32_t f = sparse_filter_hash(want), m = sparse_filter_mask(f);
    maybe = (word_at(kTierFilterWord + sparse_filter_word(f)) & m) == m;
    width = 0;
    if (hit >= nslots) return false;
    width = wide_layout ? byte_at(kTierWideWidthByte + hit) : (word_at(hit) >> kSparseTagBits);
    if (width == 0u) return false;  // an empty slot (a key whose tag is 0 matches it)
    const uint32_t lo = word_at((wide_layout ? kTierWideL0Word : kTierL0Word) + hit), hi = byte_at((wide_layout ? kTierWideHiByte : kTierHiByte) + hit);
    l = (uint64_t(hi) << 32) | lo;
    return true;
}

// (the k > 32 instantiations sit at their register limit: they take the two-tier scan as a call too)
struct TierScanOut {
    uint64_t l;
    uint32_t width, header, hit, maybe;
};
[[maybe_unused]] __device__ __attribute__((noinline)) TierScanOut sparse_scan_tier_call(const uint4 *lines, uint32_t slot, uint32_t want, bool wide_layout) {  // (results by value: no stack)
    TierScanOut o{0, 0, 0, 0, 0};
    bool maybe = false;
    o.hit = sparse_scan_tier(lines, slot, want, wide_layout, o.l, o.width, o.header, maybe) ? 1u : 0u;
    o.maybe = maybe ? 1u : 0u;
    return o;
}

// drops a WAVE-UNIFORM number of bits (0..95) of the packed symbols: whole words by selects, the rest by alignbit
template <int kWords>
__device__ __forceinline__ void consume_symbols_uniform(uint32_t (&w)[kWords], uint32_t bits) {
    const uint32_t whole = bits >> 5, rest = bits & 31u;
    uint32_t t[kWords + 1];
#pragma unroll
    for (int i = 0; i < kWords; ++i) {
        t[i] = 0u;
#pragma unroll
        for (int j = i; j < kWords; ++j) t[i] = (uint32_t(i) + whole == uint32_t(j)) ? w[j] : t[i];
    }
    t[kWords] = 0u;
#pragma unroll
    for (int i = 0; i < kWords; ++i) w[i] = __builtin_amdgcn_alignbit(t[i + 1], t[i], rest);
}

// (the k > 32 instantiations sit at their register limit: they take the xwide scan as a call, not inlined)
struct SparseScanOut {
    uint64_t l;
    uint32_t width, header, hit;
};
[[maybe_unused]] __device__ __attribute__((noinline)) SparseScanOut sparse_scan_xwide_call(const uint4 *lines, uint32_t slot, uint64_t want) {  // (results by value: no stack)
    SparseScanOut o{0, 0, 0, 0};
    o.hit = sparse_scan_xwide(lines, slot, want, o.l, o.width, o.header) ? 1u : 0u;
    return o;
}

// kPacked (matrix mode only): the queries come as 2-bit words (QuerySource::packed), possibly with a place for each count.
// A compile-time switch, not a launch-uniform branch: with both ways of fetching a tile in one kernel the compiler merged
// their results through register copies, i.e. WAITED for the tile's bytes right after asking for them -- setup no longer
// ran ahead of memory (same box, C2 random 21-mers: 0.253 ms per 10^7 against 0.225 ms before packed queries existed).
// kSparse (kPair only): the launch looks its queries up in the SPARSE suffix table (sparse_table.hpp) instead of the direct one --
// `table` = its bucket lines, `depth` = its depth, `table_side` = its side array.  A lookup is a search step of its own kind: the
// lane's line is its key's bucket, fetched with the other lanes' lines, and the lane finds its entry among the line's 14 tags.
// A compile-time switch for the same reason as kPacked -- and so that the direct-table kernels carry none of it.
// kSparse = 2: the table is of the TWO-TIER form (entries for the suffixes at least 2 wide, filter bits for the ones that occur once):
// a lookup that ends in the filter goes on through the DIRECT table (`dtable`: its line is the query's next step, decoded by the lane)
// and searches from there -- the complete-table kernels (kSparse = 1) carry none of that.
// kPair = false with kSparse (round 6): run blocks behind a sparse table -- the post-lookup steps are single-symbol steps.
template <bool kReads, bool kPair, int kWords, bool kStride96, bool kPacked, int kSparse>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_count_kmers_lanes(const uint4 *__restrict__ blocks, uint64_t total,
                                                          const uint4 *__restrict__ table, uint32_t depth, uint32_t table_packed,
                                                          const uint32_t *__restrict__ filter, uint32_t filter_mask,
                                                          const uint4 *__restrict__ table_side, const uint4 *__restrict__ pair_blocks,
                                                          const uint64_t *__restrict__ pair_super, const QuerySource src,
                                                          uint32_t *__restrict__ flags, uint64_t *__restrict__ debug,
                                                          unsigned long long *__restrict__ tile_counter, uint32_t grain,
                                                          uint64_t *__restrict__ done, uint64_t done_seq, uint64_t *__restrict__ counters,
                                                          uint32_t format, const uint4 *__restrict__ run_overflow, uint32_t sparse_nbuckets,
                                                          uint32_t sparse_probe, const uint4 *__restrict__ dtable, uint32_t dinfo,
                                                          const uint4 *__restrict__ dside) {
    static_assert(!(kReads && kPacked), "packed queries are a matrix-mode input");
    static_assert(kSparse >= 0 && kSparse <= 4, "0 = direct table; sparse table: 1 = 24-bit tags (depths up to 24), 2 = two-tier form, 3 = 32-bit tags (25..29), 4 = 40-bit tags (30..31)");
    constexpr bool kTier = kSparse == 2, kWideTags = kSparse == 3, kXwide = kSparse == 4;
    constexpr int kReach = kSparse >= 2 ? 32 : 24;  // how deep this kernel's table index may reach (search_common.hpp, pack_row_swar)
    using Scratch = LaneScratchT<kWords, kPacked, kTier>;
    // run blocks (run_index.hpp; launch-uniform): `blocks` are 128-byte lines of 512 positions with 96 one-byte runs, decoded
    // by the lane that owns the query; single-symbol steps only (the kPair instantiations never see them)
    const bool runs = !kPair && !kPacked && format != 0u;  // (round 5: k <= 64 -- the long instantiation fits the decode after all: 163 / 168 VGPRs, no spill)
    using RingItem = RingItemT<kWords, kPacked, kTier>;
    constexpr int kPieces = Scratch::kMaxK / 16;  // 16-byte pieces of a tile per lane: 2 or 4
    constexpr int kRegions = Scratch::kRegions, kLineSlots = Scratch::kLineSlots;
    constexpr uint32_t kMaxSecond = Scratch::kMaxSecond;
    __shared__ Scratch ws;
    const uint8_t *__restrict__ kmers = src.data;
    const uint32_t k = src.k;
    const uint64_t n = src.n;
    const uint32_t lane = threadIdx.x;
    const uint8_t *stage_bytes = reinterpret_cast<const uint8_t *>(ws.lines) + kStageLead;  // pack_query reads in front of a query
    const uint64_t ntiles = (n + kTile - 1) / kTile;
    const bool use_table = table != nullptr && depth > 0 && k >= depth;
    const TableEnv env{table, depth, use_table, (table_packed & 1u) != 0u, filter, filter_mask, total, table_side};
    // bit 1 of the same argument: the index is far larger than the caches (IndexView::stream_lines), so its lines are fetched with the
    // non-temporal hint -- a line is used once, and left to the default policy it evicts what IS reused (the superblock table, the query
    // stream).  Round 5, human scale, alternating on one box: 38.4 ms (36.7-39.8) by default, 36.1 ms (35.6-36.6) streaming; C3 fused,
    // whose pair blocks half live in the Infinity Cache, 14.9 -> 19.8 ms -- hence a launch-uniform switch, not a constant.
    const bool stream_lines = (table_packed & 2u) != 0u;
    // the sparse table's layout follows its depth (launch-uniform): 14 entries with 24-bit tags up to depth 24, 12 with 32-bit tags beyond
    // (the table's layout is the kernel's: one scan per instantiation; only the two-tier form still tells its two tag widths apart at run time)
    const bool sparse_wide_layout = kTier && sparse_wide(depth);
    const uint32_t sparse_nslots = kTier ? sparse_slots(depth, true) : (kXwide ? kSparseXSlots : kWideTags ? kSparseWideSlots : kSparseSlots);
    // two-tier: the direct table a lookup falls back to -- dd symbols deep (0: none, such a query searches from [0, total)), packed or flat
    const uint32_t dd = kTier ? (dinfo & 0xFFu) : 0u;
    const bool dpacked = kTier && ((dinfo >> 8) & 1u) != 0u;
    const uint64_t dkey_mask = (1ull << (2u * dd)) - 1ull;  // (dd <= 18: the direct table's index takes up to 36 bits -- a query carries its LINE there (32 bits) and its slot in the line (5 bits))
    auto scan_bucket = [&](uint32_t slot, uint64_t want, uint64_t &tl, uint32_t &tw, uint32_t &header, bool &maybe) -> bool {  // want: the tag, up to 40 bits
        maybe = false;
        if constexpr (kTier && kWords == 6) {
            const TierScanOut o = sparse_scan_tier_call(ws.lines, slot, uint32_t(want), sparse_wide_layout);
            tl = o.hit ? o.l : tl;
            tw = o.width;
            header = o.header;
            maybe = o.maybe != 0u;
            return o.hit != 0u;
        } else if constexpr (kTier) {
            return sparse_scan_tier(ws.lines, slot, uint32_t(want), sparse_wide_layout, tl, tw, header, maybe);
        }
        if constexpr (kXwide) {
            if constexpr (kWords == 6) {
                const SparseScanOut o = sparse_scan_xwide_call(ws.lines, slot, want);
                tl = o.hit ? o.l : tl;
                tw = o.width;
                header = o.header;
                return o.hit != 0u;
            } else {
                return sparse_scan_xwide(ws.lines, slot, want, tl, tw, header);
            }
        } else if constexpr (kWideTags) {
            return sparse_scan_wide(ws.lines, slot, uint32_t(want), tl, tw, header);
        } else if constexpr (kSparse == 1) {
            return sparse_scan(ws.lines, slot, uint32_t(want), tl, tw, header);
        } else {
            return false;
        }
    };
    // optional search counters (kernels.hpp, SearchCounter): wave sums kept in LDS, added to the caller's block at the end
    const bool counting = kCounting && counters != nullptr;
    if (counting && lane < uint32_t(kSearchCounters)) ws.cnt[lane] = 0u;
    auto count = [&](int which, uint64_t ballot) {
        if (lane == 0u) ws.cnt[which] += uint32_t(__popcll(ballot));
    };
    // this lane's part in the line fetches: 16 bytes (one chunk) of the line in list slot 8 i + dma_group
    const uint32_t dma_group = lane >> 3, dma_chunk_bytes = ((lane & 7u) ^ dma_group) * 16u;

    // Tiles are dealt out dynamically, `grain` consecutive tiles per ticket, so that waves that run slower
    // -- the third wave of a SIMD, more survivors, a CU shared with an RCCL kernel, a workgroup that became
    // resident late -- simply take fewer (owning a fixed share instead cost 30 % at 12 waves per CU).  A
    // wave's first ticket is its own index; further ones come from one of kTicketCounters counters
    // (counter c of n hands out tickets W + c, W + c + n, ...: one address takes only ~7 x 10^7 atomics/s,
    // which would be most of a SHORT launch's time), taken one segment ahead so that nobody waits for them.
    // Neighbouring tiles thus run at the same time on ALL eight XCDs.  Round 4 tried the opposite for ordered batches --
    // eight contiguous spans of the launch, one per XCD (workgroups b and b + 8 share an L2), so that an XCD's waves work
    // through ONE run of consecutive queries: C4, batch ordered by 24 key bits, 9.2 ms against 8.4 ms this way (12: 14.1
    // against 14.4; unordered: the same) -- with interleaved tiles one XCD's miss is the other seven's Infinity Cache hit.
    const uint32_t ncounters = min(uint32_t(kTicketCounters), max(1u, gridDim.x >> 3));  // every counter in use has waves drawing from it
    const uint32_t my_counter = (blockIdx.x >> 3) % ncounters;  // consecutive workgroups sit on different XCDs
    uint64_t static_next = (uint64_t(blockIdx.x) + gridDim.x) * grain;  // tile_counter == nullptr (small launches: no memset, no atomics): static striding
    auto take_ticket = [&]() -> uint64_t {  // first tile of this wave's next `grain` tiles (>= ntiles: there are none)
        if (tile_counter == nullptr) {
            const uint64_t t = static_next;
            static_next += uint64_t(gridDim.x) * grain;
            return t;
        }
        unsigned long long t = 0;
        if (lane == 0) t = atomicAdd(tile_counter + my_counter * 16u, 1ull);
        const uint64_t drawn = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(t >> 32))))) << 32) |
                               uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(t))));
        return (uint64_t(gridDim.x) + my_counter + drawn * uint64_t(ncounters)) * grain;
    };
    uint64_t next_tile = uint64_t(blockIdx.x) * grain;  // next tile to set up (its bytes are being fetched)
    uint32_t seg_left = grain;                          // tiles left in the current run of consecutive tiles
    uint64_t seg_after = take_ticket();
    auto advance_tile = [&]() {
        ++next_tile;
        if (--seg_left == 0u) {
            next_tile = seg_after;
            seg_left = grain;
            seg_after = take_ticket();
        }
    };
    uint32_t ring_head = 0, ring_count = 0;  // wave-uniform
    uint64_t ring_tile = 0;                  // the tile whose queries are in the ring
    uint32_t filter_pause = 0;           // as in the tiled kernel: the filter rests while nearly everything passes

    // the lane's running query
    bool have = false;
    bool tmode = false;   // kSparse: the query is still to be looked up -- l = its bucket, h = its tag, the next step fetches the bucket line
    uint32_t tdist = 0;   // ... and how many buckets beyond its own the lookup has gone
    bool tmaybe = false;  // kTier: ... and its own bucket's filter holds its bits (l >> 32 = its line of the direct table meanwhile, tdist >> 8 its slot there)
    bool dmode = false;   // kTier: the lookup ended in the filter -- l = the query's line of the DIRECT table, h = its slot there, fetched by the next step
    uint32_t ovf_l = 0, ovf_h = 0;  // run blocks: 1 + the overflow plane block this bound is to be ranked from (0: its run block)
    uint64_t l = 0, h = 0;
    uint32_t w[kWords], rem = 0;
    uint64_t qid = 0;  // where the count of the lane's query goes (its global index, or its place by QuerySource::out_index)
#pragma unroll
    for (int i = 0; i < kWords; ++i) w[i] = 0;

    // Setup runs one step AHEAD of the search and never waits for memory on its own: a tile's
    // query bytes are fetched one iteration before they are packed, and the suffix-table entries
    // a tile asks for stay in flight across the next search step (whose s_waitcnt covers them).
    //   staged_next[]: bytes of tile `next_tile` (loads issued, not waited for)
    //   prepared:      a tile has been packed and its table loads issued; per lane: prep_kind
    //                  (0 nothing to do, 1 range comes from prep_entry, 2 range is [0, total)),
    //                  prep_q (the packed symbols); prep_tile (wave-uniform) is the tile they belong to
    uint4 staged_next[kPieces];
#pragma unroll
    for (int i = 0; i < kPieces; ++i) staged_next[i] = make_uint4(0, 0, 0, 0);
    // Reads mode, fixed read length, at least a tile's worth of windows per read (launch-uniform): a tile of
    // consecutive windows lies in at most two reads, so the bytes it spans -- its windows + k - 1, twice
    // that tail across a read border: at most 64 + 2 x 63 = 190 -- are ONE dword per lane.  staged_n0 =
    // how many of the tile's windows still belong to the first of the two reads.
    const uint32_t wshift = (kReads && src.strands == 3u) ? 1u : 0u;  // both strands: two queries per window
    const uint32_t tile_windows = uint32_t(kTile) >> wshift;
    const bool reads_fast = kReads && src.win_off == nullptr && src.windows >= tile_windows;
    const double inv_windows = kReads ? 1.0 / double(src.windows) : 0.0;
    uint32_t staged_n0 = 0;  // wave-uniform
    uint32_t staged_out = 0, prep_out = 0;  // QuerySource::out_index: the output place of this lane's query of the fetched / the prepared tile
    const bool placed = kPacked && (src.out_index != nullptr || src.place_inline != 0u);  // launch-uniform
    auto place_of = [&](uint64_t v, uint32_t out) -> uint64_t { return placed ? uint64_t(out) : v; };
    // (`first`: the call in front of the loop -- the only one that can meet the single query that came with the kernel
    // arguments; the call inside the loop is compiled without that case, so that it has ONE way of loading a tile and
    // nothing to merge after the loads)
    auto fetch_tile_bytes = [&](uint64_t tile, auto first) {
        if (!kPacked && tile >= ntiles) return;  // (packed queries: loaded unconditionally, see below; n > 0)
        if (!kReads) {
            if (decltype(first)::value && !kPacked && src.inline_n != 0u) {  // pieces 0..3 belong to lanes 0..3
                staged_next[0] = lane == 0u ? src.inline_kmer[0] : lane == 1u ? src.inline_kmer[1] : lane == 2u ? src.inline_kmer[2] : lane == 3u ? src.inline_kmer[3] : make_uint4(0, 0, 0, 0);
                return;
            }
            const uint64_t q0 = tile * kTile;
            if constexpr (kPacked) {  // this lane's own query: one or two u64 words, a tile is one coalesced load
                // One shape of loads whatever the element layout and no branch around them: every result lands in ITS register
                // and nothing is merged afterwards (a merge is a register copy, and a copy waits for the load).  Lanes beyond
                // the batch -- or a tile beyond it, once per wave -- read its last query again (never used); a {query, place} element is read as
                // 8 + 4 bytes of the same 16; without places the 4 bytes are the query's own low word (not used either).
                constexpr uint32_t kOwn = kWords == 3 ? 1u : 2u;
                const uint32_t stride = src.packed_stride ? src.packed_stride : kOwn;
                const uint64_t qi = min(q0 + lane, n - 1u);
                const uint64_t *words = reinterpret_cast<const uint64_t *>(kmers) + qi * stride;
                const uint2 a = *reinterpret_cast<const uint2 *>(words);
                staged_next[0].x = a.x;
                staged_next[0].y = a.y;
                if constexpr (kWords == 6) {
                    const uint2 b = *reinterpret_cast<const uint2 *>(words + 1);
                    staged_next[0].z = b.x;
                    staged_next[0].w = b.y;
                }
                const uint32_t *place = src.place_inline != 0u ? reinterpret_cast<const uint32_t *>(words + kOwn)
                                        : src.out_index != nullptr ? src.out_index + qi : reinterpret_cast<const uint32_t *>(words);
                staged_out = *place;
                return;
            }
            const uint32_t nbytes = uint32_t(min(uint64_t(kTile), n - q0)) * k;
#pragma unroll
            for (int i = 0; i < kPieces; ++i) staged_next[i] = load_piece(kmers + q0 * k, nbytes, lane + 64u * i);
        } else if (reads_fast) {
            // first window of the tile -> (read, offset): one division per TILE, by way of a double
            // reciprocal (window indices stay below 2^38) with an exact fix-up
            const uint64_t g0 = (tile * kTile) >> wshift;
            uint64_t r0 = uint64_t(double(g0) * inv_windows);
            int64_t w0 = int64_t(g0 - r0 * src.windows);
            if (w0 < 0) { --r0; w0 += src.windows; }
            if (w0 >= int64_t(src.windows)) { ++r0; w0 -= src.windows; }
            staged_n0 = min(tile_windows, src.windows - uint32_t(w0));
            const uint64_t at = r0 * src.read_len + uint64_t(w0) + 4u * lane, data_len = src.n_reads * src.read_len;
            uint32_t d = 0;
            if (at + 4u <= data_len) {
                __builtin_memcpy(&d, kmers + at, 4);  // unaligned dword load
            } else {  // the batch's last bytes: nothing beyond the caller's buffer is touched
                for (uint32_t b = 0; b < 4u; ++b)
                    if (at + b < data_len) d |= uint32_t(kmers[at + b]) << (8u * b);
            }
            staged_next[0].x = d;
        }
    };
    fetch_tile_bytes(next_tile, std::true_type{});
    bool prepared = false;  // wave-uniform
    uint32_t prep_kind = 0;
    uint64_t prep_tile = 0;
    PackedQuery<kWords> prep_q;
#pragma unroll
    for (int i = 0; i < PackedQuery<kWords>::kBits; ++i) prep_q.bits[i] = 0;
    uint4 prep_entry = make_uint4(0, 0, 0, 0);

    for (;;) {
        // ---- A: the ring is empty: the prepared tile (its table entries arrived a step ago) moves in ----
        if (prepared && ring_count == 0u) {
            uint64_t pl = 0, ph = total;
            uint32_t skip = 0;
            // An entry of an escape line (the suffixes of a high-copy repeat; none at all on most indexes) names its flat {l, h}
            // entry in the side array: fetched HERE and waited for, by the few tiles that hold such a query.  Round 4 first
            // carried these queries into the search with a flag and fetched the entry as their first step, beside the other
            // lanes' lines -- nothing waits, but the flag's bookkeeping in every step cost C3 fused, an index without a single
            // escape line, 2.3 % (18.55 against 18.12 ms, same box); the wait costs only the tiles that meet one.
            bool escaped = false, restart = false;
            bool lookup = false;  // kSparse: enters the ring as a table lookup
            if (kSparse && prep_kind == 3u) {  // not looked up on the way (no free slot, a short search before it): its bucket is its first step
                pl = prep_entry.x;  // bucket
                ph = prep_entry.y;  // tag
                skip = kTier ? dd : depth;  // (two-tier: the symbols beyond the DIRECT table's stay until the lookup has hit)
                lookup = true;
            }
            if (kTier && prep_kind == 5u) {  // its lookup rode along and ended in the filter: the direct table's line is its first step
                pl = prep_entry.w;
                ph = (prep_entry.z >> 10) & 31u;
                skip = dd;
                lookup = true;
            }
            if (kSparse && prep_kind == 4u) {  // looked up while the tile before it was searched (step D): l, width
                pl = (uint64_t(prep_entry.y & 0xFFu) << 32) | prep_entry.x;
                ph = pl + (prep_entry.y >> 8);
                skip = depth;
                escaped = (prep_entry.y >> 8) == kSparseEscapeWidth;  // (pl = the entry's index in the side array)
            }
            if (!kSparse && prep_kind == 1u) {
                if (table_decode(env, prep_entry, pl, ph)) skip = depth;
                else escaped = true;
            }
            if (__ballot(escaped) != 0ull) {  // wave-uniform, rare
                if (kSideFetch && table_side != nullptr) {
                    if (escaped) {
                        const uint4 e = table_side[pl];
                        pl = (uint64_t(e.y) << 32) | e.x;
                        ph = (uint64_t(e.w) << 32) | e.z;
                        skip = depth;
                    }
                } else if (escaped) {  // no side array: from scratch
                    restart = true;
                    pl = 0;
                    ph = total;
                    skip = 0;
                }
            }
            const uint32_t prep_rem = k - skip;
            uint32_t prep_w[kWords];
            unpack_words<kWords>(prep_q, skip, prep_w);
            bool pending = prep_kind != 0u;
            if (pending && !lookup && (prep_rem == 0u || pl == ph)) {  // decided by the table (or an empty index)
                store_count<kReads>(src, place_of(prep_tile * kTile + lane, prep_out), ph - pl);
                pending = false;
            }
            const uint64_t pend_mask = __ballot(pending);
            if (counting) {
                count(kCntEscapeQueries, __ballot(escaped));
                count(kCntFirstLines, __ballot(escaped && !restart));  // (side-array entries fetched: lines like any other)
                count(kCntEscapeRestarts, __ballot(restart));
                count(kCntTableDecided, __ballot(prep_kind != 0u && !pending));
                count(kCntSearched, pend_mask);
            }
            if (pending) {
                const uint32_t at = __builtin_amdgcn_mbcnt_hi(uint32_t(pend_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(pend_mask), 0u));
                RingItem it;
                it.l_lo = uint32_t(pl);
                it.h_lo = uint32_t(ph);
                it.meta = uint32_t(pl >> 32) | (uint32_t(ph >> 32) << 8) | (prep_rem << 16) | (lane << 24);  // l, h < 2^40; rem <= 64
                if (kSparse && lookup) it.meta = prep_entry.z | (prep_rem << 16) | (lane << 24) | (1u << 23);  // (z: buckets gone beyond its own so far, and the tag's high byte -- two-tier: bit 8 = direct-table lookup, bit 9 = in the filter, bits 10..14 = slot in the direct table's line)
#pragma unroll
                for (int i = 0; i < kWords; ++i) it.w[i] = prep_w[i];
                if constexpr (kPacked) it.extra[RingItem::kOutAt] = prep_out;
                if constexpr (kTier) it.extra[RingItem::kDkeyAt] = prep_entry.w;
                ws.ring[(ring_head + ring_count + at) & (kRing - 1)] = it;
            }
            ring_tile = prep_tile;  // the ring was empty: everything in it belongs to this tile
            ring_count += uint32_t(__popcll(pend_mask));
            prepared = false;
            wave_lds_sync();
        }
        // ---- B: idle lanes take the waiting queries, in lane order ----
        uint64_t busy = __ballot(have);
        if (busy != ~0ull && ring_count > 0u) {
            const uint64_t idle = ~busy;
            const uint32_t my = __builtin_amdgcn_mbcnt_hi(uint32_t(idle >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle), 0u));
            if (!have && my < ring_count) {
                const RingItem it = ws.ring[(ring_head + my) & (kRing - 1)];
                l = (uint64_t(it.meta & 0xFFu) << 32) | it.l_lo;
                h = (uint64_t((it.meta >> 8) & 0xFFu) << 32) | it.h_lo;
#pragma unroll
                for (int i = 0; i < kWords; ++i) w[i] = it.w[i];
                rem = (it.meta >> 16) & (kSparse ? 0x7Fu : 0xFFu);
                tmode = kSparse && ((it.meta >> 23) & 1u) != 0u;
                tdist = 0u;
                if (kSparse && tmode) {
                    tdist = it.meta & 0xFFu;
                    l = it.l_lo;
                    h = (uint64_t((it.meta >> 8) & 0xFFu) << 32) | it.h_lo;  // the tag (40 bits in the xwide layout)
                }
                if constexpr (kTier) {
                    dmode = tmode && ((it.meta >> 8) & 1u) != 0u;
                    tmaybe = tmode && ((it.meta >> 9) & 1u) != 0u;
                    if (tmode) {
                        h = it.h_lo;  // the tag (at most 32 bits), or the slot in the direct table's line
                        l = dmode ? uint64_t(it.l_lo) : (uint64_t(it.extra[RingItem::kDkeyAt]) << 32) | it.l_lo;
                        tdist |= ((it.meta >> 10) & 31u) << 8;  // the slot travels beside the distance
                        tmode = !dmode;
                    }
                }
                qid = ring_tile * kTile + (it.meta >> 24);
                if constexpr (kPacked) qid = place_of(qid, it.extra[RingItem::kOutAt]);
                ovf_l = ovf_h = 0u;
                have = true;
            }
            const uint32_t taken = min(ring_count, uint32_t(__popcll(idle)));
            ring_head = (ring_head + taken) & (kRing - 1);
            ring_count -= taken;
        }
        // a range outside the index would turn into a wild line address: end such a query with
        // u64::MAX and a status flag instead (never seen on a well-formed index; cheap insurance)
        const bool broken = have && !(kSparse && tmode) && !(kTier && dmode) && (h > total || l > h);
        if (broken) {
            atomicOr(flags, kFlagInternal);
            if (debug != nullptr && atomicCAS(reinterpret_cast<unsigned long long *>(debug), 0ull, 1ull) == 0ull) {
                debug[1] = l;
                debug[2] = h;
                debug[3] = (uint64_t(rem) << 56) | qid;
                debug[4] = (uint64_t(w[1]) << 32) | w[0];
                debug[5] = (uint64_t(blockIdx.x) << 32) | lane;
            }
            store_count<kReads>(src, qid, ~0ull);
            have = false;
        }
        busy = __ballot(have);

        // ---- C: nothing prepared: pack the next tile and ask the table for its ranges ----
        if (!prepared && next_tile < ntiles) {
            if (counting && lane == 0u) ws.cnt[kCntWavesWorked] = 1u;
            const uint64_t tile = next_tile;
            advance_tile();
            const uint64_t q0 = tile * kTile;
            const uint32_t in_tile = uint32_t(min(uint64_t(kTile), n - q0));
            const bool filter_now = filter != nullptr && filter_pause == 0;
            bool looked_up = false, passed = false, filtered = false;
            const uint32_t tile_n0 = staged_n0;  // (fetch_tile_bytes below replaces it with the next tile's)
            const uint4 packed_words = staged_next[0];
            // (an explicit copy HERE, where the fetched tile has arrived anyway: left to the compiler, the copy moved behind the
            // next tile's load -- into a scratch register, waited for and copied back: setup stood still for a memory round trip)
            if constexpr (kPacked) asm volatile("v_mov_b32 %0, %1" : "=v"(prep_out) : "v"(staged_out));
            if (kPacked) {
                // (packed queries sit in this lane's registers already)
            } else if (!kReads) {  // the tile's bytes go through LDS (the line area is free between two steps)
#pragma unroll
                for (int i = 0; i < kPieces; ++i) ws.lines[kStageLead / 16 + lane + 64u * i] = staged_next[i];
            } else if (reads_fast) {
                // every read symbol becomes a code ONCE: forward codes at stage_bytes[0..256), the reverse
                // complement of the whole span at stage_bytes[256..512) (byte j = comp(span byte 255 - j))
                const uint32_t d = staged_next[0].x;
                uint32_t f = 0, c = 0;
#pragma unroll
                for (uint32_t b = 0; b < 4u; ++b) {
                    uint32_t sym = (d >> (8u * b)) & 0xFFu;
                    if (src.ascii) sym = ascii_to_code(sym);
                    f |= sym << (8u * b);
                    c |= complement_code(sym) << (8u * (3u - b));
                }
                uint32_t *codes = reinterpret_cast<uint32_t *>(ws.lines) + kStageLead / 4;
                codes[lane] = f;
                codes[64u + (63u - lane)] = c;
            }
            wave_lds_sync();
            prep_kind = 0;
            prep_tile = tile;
            if (lane < in_tile) {
                PackedQuery<kWords> pq;
                if (kPacked) {
                    pack_two_bit<kWords>(k, depth, (uint64_t(packed_words.y) << 32) | packed_words.x, (uint64_t(packed_words.w) << 32) | packed_words.z, pq);
                } else if (!kReads) {
                    pack_query<false, kWords, kReach>(src, depth, stage_bytes + lane * k, q0 + lane, pq);
                } else if (reads_fast) {
                    // window j of the tile starts o bytes into the span (k - 1 more once the read border is
                    // crossed); forward: the row at F + o; reverse complement: the row at C + 256 - o - k --
                    // either way a k-byte row packed exactly like a row of a query matrix
                    const uint32_t j = lane >> wshift, o = j + (j >= tile_n0 ? k - 1u : 0u);
                    const bool rc = src.strands == 3u ? (lane & 1u) != 0u : src.strands == 2u;
                    pack_query<false, kWords, kReach>(src, depth, stage_bytes + (rc ? 512u - o - k : o), q0 + lane, pq);
                } else {
                    pack_query<true, kWords, kReach>(src, depth, stage_bytes, q0 + lane, pq);
                }
                if (pq.bad) {  // the reference asserts (msbwt_core.rs:127)
                    store_count<kReads>(src, place_of(q0 + lane, prep_out), ~0ull);
                    atomicOr(flags, kFlagInvalidSymbol);
                } else if (use_table && pq.acgt) {
                    bool maybe = true;
                    if (filter_now) {  // L2-resident presence bit first: an absent suffix never touches the table line
                        const uint32_t fi = uint32_t(pq.tidx) & filter_mask;
                        maybe = ((filter[fi >> 5] >> (fi & 31u)) & 1u) != 0u;
                        looked_up = true;
                        passed = maybe;
                    }
                    if (maybe) {
                        if constexpr (kSparse) {  // nothing to fetch here: the bucket line is the query's first search step
                            const uint64_t x = sparse_mix(pq.tidx, 2u * depth);
                            prep_entry.x = sparse_bucket(x, 2u * depth, sparse_nbuckets);
                            prep_entry.y = (kWideTags || kXwide || sparse_wide_layout) ? uint32_t(x) : uint32_t(x) & ((1u << kSparseTagBits) - 1u);  // sparse_tag
                            prep_entry.z = kXwide ? (uint32_t(x >> 32) & 0xFFu) << 8 : 0u;  // sparse_tag_hi  // low byte: buckets gone beyond its own; next byte: bits 32..39 of the tag (xwide layout; else 0)
                            if constexpr (kTier) {  // where the direct table keeps this query's suffix: line (w) and slot in it (z, bits 10..14)
                                const uint64_t dk = pq.tidx & dkey_mask, dline = dpacked ? dk / kPackedPerLine : dk >> 3;
                                prep_entry.w = uint32_t(dline);
                                prep_entry.z |= uint32_t(dk - dline * (dpacked ? uint64_t(kPackedPerLine) : 8ull)) << 10;
                            }
                            prep_kind = 3;
                        } else {
                            prep_entry = table_fetch(env, pq.tidx);  // stays in flight: consumed in step A of a later iteration
                            prep_kind = 1;
                        }
                        prep_q = pq;
                    } else {
                        store_count<kReads>(src, place_of(q0 + lane, prep_out), 0ull);
                        filtered = true;
                    }
                } else {
                    prep_kind = 2;
                    prep_q = pq;
                }
            }
            if (filter != nullptr) {
                if (filter_now) {
                    const uint32_t nlook = uint32_t(__popcll(__ballot(looked_up))), npass = uint32_t(__popcll(__ballot(passed)));
                    if (nlook > 0 && npass * 10u >= nlook * 9u) filter_pause = 7;
                } else {
                    --filter_pause;
                }
            }
            if (counting) count(kCntTableDecided, __ballot(filtered));
            prepared = true;
            wave_lds_sync();         // every lane has read its staged bytes: the line area may be overwritten
            fetch_tile_bytes(next_tile, std::false_type{});  // the following tile's bytes start their trip now
        }
        // A search step costs the same whether 5 or 64 lanes take it: while lanes are idle and another
        // tile is on its way (prepared; the ring is empty then), fetch that tile's survivors first.  When
        // most queries end in the filter or the table (random k-mers) search steps become rare and full.
#ifndef MSBWT_LANES_MIN_BUSY
#define MSBWT_LANES_MIN_BUSY 64
#endif
        if (prepared && uint32_t(__popcll(busy)) < uint32_t(MSBWT_LANES_MIN_BUSY)) continue;  // (waits for the tile's table entries)
        if (busy == 0ull) break;     // nothing in flight, nothing waiting, no tiles left

        // ---- D: one search step of every busy lane ----
        const uint32_t s1 = w[0] & 7u, s2 = (w[0] >> 3) & 7u;
        const bool looking = kSparse && have && tmode;  // this step fetches the query's bucket of the sparse table
        const bool dlooking = kTier && have && dmode;   // ... its line of the direct table (two-tier: the lookup ended in the filter)
        const bool pair = kPair && have && !looking && !dlooking && rem >= 2u && (acgt_bit(s1) & acgt_bit(s2)) != 0u;
        const uint32_t a2 = acgt_code(s1) & 3u, b2 = acgt_code(s2) & 3u;
        constexpr bool s96 = kStride96;  // compile-time: the stride-128 kernel carries no division
        const uint64_t base = pair ? reinterpret_cast<uint64_t>(pair_blocks) : reinterpret_cast<uint64_t>(blocks);
        // an idle slot names the index's first block (an L2 hit) instead of masking its eight DMA lanes
        // off: one branch-free load instruction per region is cheaper than the exec-mask dance
        const uint64_t dummy = reinterpret_cast<uint64_t>(blocks);
        uint64_t *list = reinterpret_cast<uint64_t *>(ws.lines);  // this step's line addresses: read back before the first line lands
        // the block of l -- and h's own block only when h does not fit the same line (overlapping pair
        // blocks hold 32 positions beyond their own 96)
        const uint64_t bl = pair ? pair_block_of(l, s96) : l >> 8;
        const uint64_t start_l = pair ? pair_block_start(bl, s96) : bl << 8;
        const bool same = pair ? (h - start_l) < 128u : (h >> 8) == bl;
        const uint64_t bh = same ? bl : (pair ? pair_block_of(h, s96) : h >> 8);
        const uint32_t r_l = uint32_t(l - start_l), r_h = uint32_t(h - (same ? start_l : (pair ? pair_block_start(bh, s96) : bh << 8)));
        uint64_t line_l = base + bl * 128u, line_h = base + bh * 128u;
        bool one_line = same;
        if (!kPair && !kPacked && runs) {  // a bound's line: its run block, or -- found out in the iteration before -- its overflow plane block
            line_l = ovf_l != 0u ? reinterpret_cast<uint64_t>(run_overflow) + uint64_t(ovf_l - 1u) * 128u : reinterpret_cast<uint64_t>(blocks) + (l >> 9) * 128u;
            line_h = ovf_h != 0u ? reinterpret_cast<uint64_t>(run_overflow) + uint64_t(ovf_h - 1u) * 128u : reinterpret_cast<uint64_t>(blocks) + (h >> 9) * 128u;
            one_line = line_l == line_h;
        }
        if (kSparse && looking) {
            line_l = reinterpret_cast<uint64_t>(table) + uint64_t(kTier ? uint32_t(l) : l) * 128u;
            one_line = true;
        }
        if (kTier && dlooking) {
            line_l = reinterpret_cast<uint64_t>(dtable) + l * 128u;
            one_line = true;
        }
        const bool second = have && !one_line;
        const uint64_t second_mask = __ballot(second);
        const uint32_t second_rank = __builtin_amdgcn_mbcnt_hi(uint32_t(second_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(second_mask), 0u));
        const uint32_t nsecond = min(uint32_t(__popcll(second_mask)), kMaxSecond);
        // room for 16 second lines per step: a lane beyond that sits this step out (its query simply
        // takes the step in the next iteration)
        const bool act = have && !(second && second_rank >= kMaxSecond);
        const uint32_t slot_l = lane, slot_h = second ? 64u + second_rank : lane;
        list[lane] = act ? line_l : dummy;
        if (second && act) list[slot_h] = line_h;
        // kSparse: the second-line slots this step leaves free carry lookups of the PREPARED tile -- its queries' bucket lines ride
        // along with the search of the tile before it (16 a step: the four pair steps of a 31-mer behind a depth-23 table bring in
        // a whole tile), so that a tile whose turn comes mostly knows its ranges and a lookup costs no search step of its own.
        bool riding = false;
        uint32_t slot_ride = 0, nextra = nsecond;
        if (kSparse) {
            const bool wants = prepared && prep_kind == 3u;
            const uint64_t wants_mask = __ballot(wants);
            if (wants_mask != 0ull) {  // wave-uniform
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(wants_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(wants_mask), 0u));
                const uint32_t room = kMaxSecond - nsecond;
                riding = wants && rank < room;
                slot_ride = 64u + nsecond + rank;
                if (riding) list[slot_ride] = reinterpret_cast<uint64_t>(table) + uint64_t(prep_entry.x) * 128u;
                nextra += min(uint32_t(__popcll(wants_mask)), room);
            }
        }
        if (lane < 8u && 64u + nextra + lane < uint32_t(kLineSlots)) list[64u + nextra + lane] = dummy;  // the ragged end of the last second-bound region
        // pair steps: K[a][b] + occ2 at the superblock start (L2-resident table).  One 8-byte load per lane
        // is a separate L2 request each (64 per wave): the second bound's base is fetched only in the rare
        // case that it lies in another superblock -- into a register of its own, so that nothing here waits
        // for the first load (a copy of super_l would: a whole L2 round trip before the lines are even asked for)
        uint64_t super_l = 0, super_far = 0;
        uint32_t far = 0;
        if (pair) {
            const uint32_t p = a2 * 4u + b2;
            const uint64_t sbl = bl >> kPairSuperBlocks, sbh = bh >> kPairSuperBlocks;
            super_l = pair_super[sbl * 16u + p];
            far = sbh != sbl ? 1u : 0u;
            if (far != 0u) super_far = pair_super[sbh * 16u + p];
        }
        wave_lds_sync();
        {   // all line addresses first (one LDS round trip), then the LDS-DMA loads back to back
            uint64_t addr[kRegions];
#pragma unroll
            for (int i = 0; i < kRegions; ++i) addr[i] = list[8u * i + dma_group] + dma_chunk_bytes;
            // the list lives in the line area: every address must be in registers before a line may land on it
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < kRegions; ++i) {
                const bool wanted = i < 8 ? ((busy >> (8 * i)) & 0xFFull) != 0ull : nextra > uint32_t(8 * (i - 8));  // wave-uniform
                if (wanted) {  // (the cache policy is an immediate of the instruction: two copies of the load)
                    if (stream_lines) __builtin_amdgcn_global_load_lds((global_void *)addr[i], (lds_void *)&ws.lines[region_base(i)], 16, 0, 2);
                    else __builtin_amdgcn_global_load_lds((global_void *)addr[i], (lds_void *)&ws.lines[region_base(i)], 16, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): every line has landed in LDS (and the table entries of step C are here)
        wave_lds_sync();
        if (counting) {
            count(kCntWaveSteps, 1ull);
            count(kCntLaneSteps, __ballot(act));
            count(kCntPairSteps, __ballot(act && pair));
            count(kCntSecondLines, __ballot(act && second));
            count(kCntSatOut, __ballot(have && !act));
            if (kSparse) {
                count(kCntTableSteps, __ballot(act && (looking || dlooking)));
                count(kCntTableSteps, __ballot(riding));
                count(kCntFirstLines, __ballot(riding));
                count(kCntTableRides, __ballot(riding));
            }
        }
        if (kSparse && __ballot(riding) != 0ull) {  // the prepared tile's lookups that rode along
            if (riding) {
                uint64_t tl = 0;
                uint32_t tw = 0, header = 0;
                bool maybe = false;
                const bool hit = scan_bucket(slot_ride, kTier ? uint64_t(prep_entry.y) : ((uint64_t((prep_entry.z >> 8) & 0xFFu) << 32) | prep_entry.y), tl, tw, header, maybe);
                if (kTier && !hit && (prep_entry.z & 0xFFu) == 0u && maybe) prep_entry.z |= 1u << 9;  // its own bucket's filter holds its bits
                if (hit) {
                    prep_entry.x = uint32_t(tl);
                    prep_entry.y = uint32_t(tl >> 32) | (tw << 8);
                    prep_kind = 4;
                } else if (header > sparse_nslots && (prep_entry.z & 0xFFu) < sparse_probe) {  // entries of this bucket were displaced: the next one, next time
                    ++prep_entry.x;
                    ++prep_entry.z;
                } else if (kTier && ((prep_entry.z >> 9) & 1u) != 0u) {  // no entry, but the filter knows it (occurs once, or a false positive): the direct table's path
                    prep_entry.z |= 1u << 8;
                    prep_kind = dd != 0u ? 5u : 2u;  // (no direct table: from [0, total))
                } else {  // a miss in a complete table: the suffix does not occur (msbwt_core.rs:151-153)
                    store_count<kReads>(src, place_of(prep_tile * kTile + lane, prep_out), 0ull);
                    prep_kind = 0;
                }
            }
            if (counting) {
                count(kCntTableDisplaced, __ballot(riding && prep_kind == 3u));
                count(kCntTableDecided, __ballot(riding && prep_kind == 0u));
                if (kTier) count(kCntTierFallbacks, __ballot(riding && (prep_kind == 5u || prep_kind == 2u)));
            }
        }
        if (act) {
            uint64_t nl, nh;
            bool step_done = true;
            if (kSparse && looking) {  // this step fetched the query's own bucket
                uint64_t tl = 0;
                uint32_t width = 0, header = 0;
                bool maybe = false;
                const bool hit = scan_bucket(slot_l, h, tl, width, header, maybe);
                if (kTier && !hit && (tdist & 0xFFu) == 0u && maybe) tmaybe = true;  // its own bucket's filter holds its bits
                bool fell = false;
                nl = nh = 0;
                step_done = false;
                if (hit) {
                    if constexpr (kTier) {  // the symbols between the direct table's depth and this table's are answered now
                        consume_symbols_uniform<kWords>(w, 3u * (depth - dd));
                        rem -= depth - dd;
                    }
                    l = tl;
                    h = tl + width;
                    if (width == kSparseEscapeWidth) {  // a high-copy suffix: its range is a flat entry of the side array, one more line
                        uint4 e = table_side[tl];
                        // The wait for this rare load stays HERE (the empty asm is a use of its registers).  Round 6 found what happens
                        // otherwise: the loaded registers simply BECOME l and h, their first use is the next iteration's address arithmetic,
                        // and the compiler puts s_waitcnt vmcnt(0) in front of it -- in EVERY iteration, for every lane: the tile bytes and
                        // table lines fetched a step ahead are waited for at once, setup no longer runs ahead of memory (C3 fused 15.5 ->
                        // 17.2 ms on the same box; eleven vmcnt(0) more in the kernel than round 5's had).
                        asm volatile("" : "+v"(e.x), "+v"(e.y), "+v"(e.z), "+v"(e.w));
                        l = (uint64_t(e.y) << 32) | e.x;
                        h = (uint64_t(e.w) << 32) | e.z;
                    }
                    tmode = false;
                    if (rem == 0u) {  // k == depth: the table's range is the answer
                        store_count<kReads>(src, qid, h - l);
                        have = false;
                    }
                } else if (header > sparse_nslots && (kTier ? (tdist & 0xFFu) : tdist) < sparse_probe) {  // entries of this bucket were displaced: the next one
                    ++l;
                    ++tdist;
                } else if (kTier && tmaybe) {  // no entry, but the filter knows it (occurs once, or a false positive): the direct table's path
                    tmode = false;
                    fell = true;
                    if (dd != 0u) {
                        dmode = true;
                        h = tdist >> 8;
                        l = l >> 32;
                    } else {  // no direct table: from [0, total) (w and rem were never cut)
                        l = 0;
                        h = total;
                    }
                } else {  // a miss in a complete table: the suffix does not occur (msbwt_core.rs:151-153)
                    store_count<kReads>(src, qid, 0ull);
                    have = false;
                }
                if (counting) {
                    count(kCntEscapeQueries, __ballot(hit && width == kSparseEscapeWidth));
                    count(kCntFirstLines, __ballot(hit && width == kSparseEscapeWidth));
                    count(kCntTableDisplaced, __ballot(!hit && have && !fell));
                    count(kCntTableDecided, __ballot(!hit && !have));
                    if (kTier) count(kCntTierFallbacks, __ballot(fell));
                }
            } else if (kTier && dlooking) {  // this step fetched the query's line of the direct table: its range after dd symbols
                const uint32_t base_d = line_base(slot_l), g_d = slot_l & 7u, dslot = uint32_t(h);
                const uint32_t *lw = reinterpret_cast<const uint32_t *>(ws.lines);
                bool esc = false;
                nl = nh = 0;
                step_done = false;
                if (dpacked) {  // u64 base | 30 x { l - base : 16, h - l : 16 } (search_common.hpp)
                    const uint4 c0 = ws.lines[base_d + (0u ^ g_d)];
                    const uint64_t b = (uint64_t(c0.y) << 32) | c0.x;
                    const uint32_t word = 2u + dslot, e = lw[(base_d + ((word >> 2) ^ g_d)) * 4u + (word & 3u)];
                    esc = (b & kPackedEscape) != 0ull;
                    l = esc ? (b & ~kPackedEscape) * kSidePerLine + dslot : b + (e & 0xFFFFu);
                    h = l + (e >> 16);
                } else {  // flat: eight {l, h} entries per line
                    const uint4 e = ws.lines[base_d + (dslot ^ g_d)];
                    l = (uint64_t(e.y) << 32) | e.x;
                    h = (uint64_t(e.w) << 32) | e.z;
                }
                if (esc) {  // an escape line (the suffixes of a high-copy repeat): its flat entry in the direct table's side array, one more line
                    uint4 e = dside[l];
                    asm volatile("" : "+v"(e.x), "+v"(e.y), "+v"(e.z), "+v"(e.w));  // (waited for here, not by the next iteration: see the sparse table's side array above)
                    l = (uint64_t(e.y) << 32) | e.x;
                    h = (uint64_t(e.w) << 32) | e.z;
                }
                dmode = false;
                if (l == h) {  // the filter's false positive (or an absent k-mer that shares its bits): nothing occurs
                    store_count<kReads>(src, qid, 0ull);
                    have = false;
                }
                if (counting) {
                    count(kCntEscapeQueries, __ballot(esc));
                    count(kCntFirstLines, __ballot(esc));
                    count(kCntTableDecided, __ballot(!have));
                }
            } else if (pair) {
                PairLine L;
                read_pair_line(ws.lines, slot_l, a2, b2, L);
                nl = pair_line_bound(L, super_l, r_l);
                if (second) read_pair_line(ws.lines, slot_h, a2, b2, L);
                asm volatile("" : "+v"(far));  // opaque here: the compiler must not fold this select back into the branch above
                nh = pair_line_bound(L, far != 0u ? super_far : super_l, r_h);
                consume_symbols<kWords>(w, 6);
                rem -= 2u;
            } else if (!kPair && !kPacked && runs) {
                uint32_t need_l = 0, need_h = 0;
                PlaneLine L;
                nl = nh = 0;
                RunCont cont{24u, 0u, 0u};
                if (ovf_l != 0u) {
                    read_plane_line(ws.lines, slot_l, s1, L);
                    nl = plane_line_bound(L, s1, l);
                } else {
                    run_line_first(ws.lines, slot_l, s1, l, nl, need_l, cont);
                }
                if (ovf_h != 0u) {
                    read_plane_line(ws.lines, slot_h, s1, L);
                    nh = plane_line_bound(L, s1, h);
                } else {  // h's rank continues l's decode inside the same run block, or starts at its own block's first dword
                    run_line_continue(ws.lines, slot_h, s1, h, one_line && ovf_l == 0u, cont, nh, need_h);
                }
                if ((need_l | need_h) != 0u) {  // an overflow block: the step is taken again with that bound's plane block fetched
                    ovf_l = ovf_l != 0u ? ovf_l : need_l;
                    ovf_h = ovf_h != 0u ? ovf_h : need_h;
                    step_done = false;
                } else {
                    ovf_l = ovf_h = 0u;
                    consume_symbols<kWords>(w, 3);
                    --rem;
                }
            } else {
                PlaneLine L;
                read_plane_line(ws.lines, slot_l, s1, L);
                nl = plane_line_bound(L, s1, l);
                if (second) read_plane_line(ws.lines, slot_h, s1, L);
                nh = plane_line_bound(L, s1, h);
                consume_symbols<kWords>(w, 3);
                --rem;
            }
            if (step_done) {
                l = nl;
                h = nh;
                if (rem == 0u || l == h) {
                    store_count<kReads>(src, qid, h - l);
                    have = false;
                }
            }
        }
        wave_lds_sync();  // the next iteration overwrites the lines
    }
    // Small host batches run as ONE wave and announce their completion in host-visible memory, so that the caller
    // can poll a word instead of paying for a stream synchronisation: every count of this wave is out (system
    // scope) before the word changes.
    if (counting) {
        wave_lds_sync();
        if (lane < uint32_t(kSearchCounters) && lane != uint32_t(kCntFirstLines))
            atomicAdd(reinterpret_cast<unsigned long long *>(counters + lane), uint64_t(ws.cnt[lane]));
        if (lane == uint32_t(kCntFirstLines)) atomicAdd(reinterpret_cast<unsigned long long *>(counters + lane), uint64_t(ws.cnt[kCntLaneSteps]) + ws.cnt[kCntFirstLines]);
    }
    if (done != nullptr) {
        __threadfence_system();
        if (lane == 0u) __hip_atomic_store(done, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// A work item carries its tile index in 32 bits: at most 2^32 tiles (2^38 queries) per launch
constexpr uint64_t kMaxTiles = 1ull << 32;

// The kernel is persistent: the grid is what the device keeps resident -- workgroups per CU
// (occupancy API: LDS- and VGPR-bound, capped below) x CUs; tiles are dealt out by atomic tickets.
template <bool kReads, bool kPair, int kWords, bool kStride96, bool kPacked, int kSparse>
uint32_t resident_waves() {
    static const uint32_t cached = [] {
        int device = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&device) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_count_kmers_lanes<kReads, kPair, kWords, kStride96, kPacked, kSparse>, 64, 0) != hipSuccess ||
            cus <= 0 || per_cu <= 0)
            return 7u * 256u;
        // LDS decides (12.1 / 12.9 KiB -> 12 waves); whole multiples of the four SIMDs only: with
        // 10 waves the two three-wave SIMDs run slower and throughput FELL (tools/sweep_variants.sh)
        per_cu = std::min(per_cu, MSBWT_LANES_WAVE_CAP);
        if (per_cu > 4) per_cu -= per_cu % 4;
        if (const char *env = std::getenv("MSBWT_LANES_WAVES_PER_CU")) {  // experiments
            const int want = std::atoi(env);
            if (want > 0) per_cu = want;
        }
        if (std::getenv("MSBWT_VERBOSE")) std::fprintf(stderr, "[msbwt] lanes kernel <%d,%d,%d,%d,%d,%d>: %d workgroups per CU x %d CUs\n", int(kReads), int(kPair), kWords, int(kStride96), int(kPacked), int(kSparse), per_cu, cus);
        return uint32_t(cus) * uint32_t(per_cu);
    }();
    return cached;
}

template <bool kReads, bool kPair, int kWords, bool kStride96, bool kPacked, int kSparse>
hipError_t launch_variant(hipStream_t stream, const IndexView &ix, const QuerySource &src, uint32_t *flags) {
    // (kSparse: the sparse table's lines, depth and side array travel in the direct table's arguments; `sp`: the one of the two that serves this k)
    const SparseView none{};
    const SparseView &sp = kSparse && sparse_for(ix, src.k) ? *sparse_for(ix, src.k) : none;
    const uint4 *table = static_cast<const uint4 *>(kSparse ? sp.lines : ix.table.entries);
    const uint32_t *filter = ix.table.entries ? ix.table.filter : nullptr;
    const uint32_t filter_mask = filter ? uint32_t((1ull << (2 * ix.table.filter_depth)) - 1ull) : 0u;
    const uint64_t tiles = (src.n + kTile - 1) / kTile;
    const uint64_t waves = std::min<uint64_t>(tiles, resident_waves<kReads, kPair, kWords, kStride96, kPacked, kSparse>());
    if (tiles > kMaxTiles) return hipErrorInvalidValue;
    // Ticket counters are only needed when there are more tiles than waves; without them (small batches, the
    // single-query path: no memset, no atomics) the kernel strides statically.
    const bool tickets = ix.tile_counter != nullptr && tiles > waves;
    if (tickets) {
        const hipError_t zeroed = hipMemsetAsync(ix.tile_counter, 0, kTicketBytes, stream);
        if (zeroed != hipSuccess) return zeroed;
    }
    // tiles per ticket: about eight tickets per wave at least, sixteen tiles at most
    const uint32_t grain = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(16, tiles / (waves * 8))));
    const uint4 *side = static_cast<const uint4 *>(kSparse ? sp.side : (table ? ix.table.side : nullptr));
    hipLaunchKernelGGL((k_count_kmers_lanes<kReads, kPair, kWords, kStride96, kPacked, kSparse>), dim3(uint32_t(waves)), dim3(64), 0, stream,
                       static_cast<const uint4 *>(ix.blocks), ix.total, table, kSparse ? sp.depth : uint32_t(ix.table.depth), ((!kSparse && ix.table.packed) ? 1u : 0u) | (ix.stream_lines ? 2u : 0u),
                       filter, filter_mask, side, static_cast<const uint4 *>(ix.pair_blocks), ix.pair_super, src, flags, ix.debug,
                       tickets ? static_cast<unsigned long long *>(ix.tile_counter) : nullptr, grain, waves == 1 ? ix.done : nullptr, ix.done_seq, ix.counters,
                       uint32_t(ix.block_format), static_cast<const uint4 *>(ix.overflow), sp.nbuckets, sp.probe,
                       // two-tier: the direct table its filter sends queries to (depth | packed << 8), and that table's side array
                       static_cast<const uint4 *>(kSparse == 2 ? ix.table.entries : nullptr),
                       kSparse == 2 && ix.table.entries ? (uint32_t(ix.table.depth) & 0xFFu) | (ix.table.packed ? 0x100u : 0u) : 0u,
                       static_cast<const uint4 *>(kSparse == 2 && ix.table.entries ? ix.table.side : nullptr));
    return hipGetLastError();
}

template <bool kReads, bool kPacked, int kSparse>
hipError_t launch_sparse_shape(bool pair, bool longk, hipStream_t stream, const IndexView &ix, const QuerySource &src, uint32_t *flags) {
    if (!pair) {  // run blocks behind a sparse table: single-symbol steps after the lookup (packed queries never reach run blocks)
        if constexpr (kPacked) return hipErrorInvalidValue;
        else return longk ? launch_variant<kReads, false, 6, false, false, kSparse>(stream, ix, src, flags) : launch_variant<kReads, false, 3, false, false, kSparse>(stream, ix, src, flags);
    }
    if (ix.pair_stride96) return longk ? launch_variant<kReads, true, 6, true, kPacked, kSparse>(stream, ix, src, flags) : launch_variant<kReads, true, 3, true, kPacked, kSparse>(stream, ix, src, flags);
    return longk ? launch_variant<kReads, true, 6, false, kPacked, kSparse>(stream, ix, src, flags) : launch_variant<kReads, true, 3, false, kPacked, kSparse>(stream, ix, src, flags);
}

}  // namespace

// the sparse-table launches of the other translation units (kSparse: 2 = two-tier, 3 = 32-bit tags, 4 = 40-bit tags)
hipError_t launch_lanes_sparse_tier(bool reads, bool packed, bool pair, bool longk, hipStream_t stream, const IndexView &ix, const QuerySource &src, uint32_t *flags);
hipError_t launch_lanes_sparse_wide(bool reads, bool packed, bool pair, bool longk, hipStream_t stream, const IndexView &ix, const QuerySource &src, uint32_t *flags);
hipError_t launch_lanes_sparse_xwide(bool reads, bool packed, bool pair, bool longk, hipStream_t stream, const IndexView &ix, const QuerySource &src, uint32_t *flags);

#define MSBWT_DEFINE_SPARSE_LAUNCH(NAME, LAYOUT)                                                                                                          \
    hipError_t NAME(bool reads, bool packed, bool pair, bool longk, hipStream_t stream, const IndexView &ix, const QuerySource &src, uint32_t *flags) { \
        if (reads) return packed ? hipErrorInvalidValue : launch_sparse_shape<true, false, LAYOUT>(pair, longk, stream, ix, src, flags);                \
        return packed ? launch_sparse_shape<false, true, LAYOUT>(pair, longk, stream, ix, src, flags)                                                   \
                      : launch_sparse_shape<false, false, LAYOUT>(pair, longk, stream, ix, src, flags);                                                 \
    }

}  // namespace msbwt
