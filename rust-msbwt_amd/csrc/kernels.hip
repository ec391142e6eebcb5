// gfx950 (MI355X, CDNA4) kernels for batched FM-index backward search over plane blocks.
// Integer / bit work bound by random 128-byte fetches: no MFMA anywhere.
//
// Rank primitives (rank_ops.hpp): an 8-lane group owns one query.  In the search loop lanes
// 0-3 handle bound l and lanes 4-7 bound h; a lane loads two 16-byte chunks of its bound's
// 128-byte block (two coalesced global_load_dwordx4 per step), popcounts its symbols, and the
// quad sums with two DPP steps plus one cross-quad exchange -- no LDS traffic, no barriers.
// Block layout: plane_index.hpp.  This is the kernel for SHORT searches (few symbols left after
// the suffix table: setup dominates and 32 waves per CU hide its latency); long searches go to
// the one-query-per-lane kernel of lanes.hip, which also takes two symbols per step.
//
// count_kmers (k <= 64) works on tiles of 64 queries per wave, in two phases:
//   1. lane-per-query setup: the tile's query bytes are staged through LDS with coalesced
//      16-byte loads; every lane validates and bit-packs one query and looks its last
//      `depth` symbols up in the suffix table (64 independent loads in flight per wave).
//      Queries that are already decided are done; the others are compacted into an LDS
//      work list with a wave ballot + prefix count.
//   2. group-per-query search: the wave's 8 groups pull queries off the work list and run
//      one backward-search step per loop iteration each, refilling as they finish, so lanes
//      stay busy although queries need different numbers of steps.
// Results go to LDS and leave as one coalesced 512-byte store per tile.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.hpp"
#include "rank_ops.hpp"
#include "search_common.hpp"

namespace msbwt {
namespace {

constexpr int kWavesPerBlock = 4;

// ---- count_kmers, any k: one query per group at a time, symbols read as needed ----------
__global__ __launch_bounds__(256) void k_count_kmers_generic(const uint4 *__restrict__ blocks, uint32_t format,
                                                             const uint4 *__restrict__ overflow, uint64_t total,
                                                             const uint8_t *__restrict__ kmers, uint32_t k,
                                                             uint64_t n, uint64_t *__restrict__ counts,
                                                             uint32_t *__restrict__ flags) {
    const uint32_t sub = threadIdx.x & (kGroup - 1);
    const uint64_t ngroups = (uint64_t(gridDim.x) * blockDim.x) / kGroup;
    for (uint64_t q = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) / kGroup; q < n; q += ngroups) {
        const uint8_t *kmer = kmers + q * k;
        // the reference asserts every symbol < 6 before searching (msbwt_core.rs:127)
        uint32_t bad = 0;
        for (uint32_t i = sub; i < k; i += kGroup) bad |= (kmer[i] >= 6u) ? 1u : 0u;
        bad = group_sum(bad);
        uint64_t result;
        if (bad) {
            result = ~0ull;
            if (sub == 0) atomicOr(flags, kFlagInvalidSymbol);
        } else {
            Range r{0, total};
            bool broken = false;  // a range outside the index (never seen on a well-formed one) must not become a block address
            for (uint32_t i = k; i-- > 0 && r.l != r.h && !(broken = r.h > total || r.l > r.h);)
                r = constrain_any(format, blocks, overflow, kmer[i], r.l, r.h, sub);
            result = broken ? ~0ull : r.h - r.l;
            if (broken && sub == 0) atomicOr(flags, kFlagInternal);
        }
        if (sub == 0) counts[q] = result;
    }
}

// ---- count_kmers, 1 <= k <= 64: tiled two-phase kernel -----------------------------------
template <int kWords>
struct WaveScratchT {
    static constexpr int kMaxK = kWords * 32 / 3;  // 32 or 64
    uint4 stage[kStageLead / 16 + kTile * kMaxK / 16];  // kStageLead free bytes (search_common.hpp), then the tile's query bytes (2 or 4 KiB)
    WorkItemT<kWords> work[kTile];                 // 2 or 3 KiB
    uint64_t result[kTile];                        // 512 B
};

template <bool kReads, int kWords>
__global__ __launch_bounds__(256, kWords == 6 ? 4 : 6) void k_count_kmers_tiled(
    const uint4 *__restrict__ blocks, uint32_t format, const uint4 *__restrict__ overflow, uint64_t total,
    const uint4 *__restrict__ table, uint32_t depth, uint32_t table_packed, const uint32_t *__restrict__ filter,
    uint32_t filter_mask, const uint4 *__restrict__ table_side, const QuerySource src, uint32_t *__restrict__ flags) {
    constexpr int kLanes = kGroup;
    using Scratch = WaveScratchT<kWords>;
    using WorkItem = WorkItemT<kWords>;
    constexpr int kPieces = Scratch::kMaxK / 16;  // 16-byte pieces of a tile per lane: 2 or 4
    __shared__ Scratch scratch[kWavesPerBlock];
    const uint8_t *__restrict__ kmers = src.data;
    const uint32_t k = src.k;
    const uint64_t n = src.n;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sub = lane & (kLanes - 1);
    const uint32_t group_first_lane = lane & ~uint32_t(kLanes - 1);
    constexpr uint64_t kGroupLeaders = 0x0101010101010101ull;
    Scratch &ws = scratch[threadIdx.x >> 6];
    const uint8_t *stage_bytes = reinterpret_cast<const uint8_t *>(ws.stage) + kStageLead;

    const uint64_t ntiles = (n + kTile - 1) / kTile;
    const uint64_t wave_id = uint64_t(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    const uint64_t nwaves = uint64_t(gridDim.x) * kWavesPerBlock;
    const bool use_table = table != nullptr && depth > 0 && k >= depth;

    const TableEnv env{table, depth, use_table, table_packed != 0u, filter, filter_mask, total, table_side};
    // A tile is at most 64 x kMaxK bytes = kPieces pieces per lane, all loads in flight at once.
    // For k <= 32 the loads of the NEXT tile are issued before the current tile is searched, hiding
    // their latency; the k <= 64 variant has no registers to spare for that and loads at the top
    // of the tile.
    constexpr bool kPrefetch = kWords == 3;
    uint4 staged[kPieces];
#pragma unroll
    for (int i = 0; i < kPieces; ++i) staged[i] = make_uint4(0, 0, 0, 0);
    if (!kReads && kPrefetch && wave_id < ntiles) {
        const uint64_t q0 = wave_id * kTile;
        const uint32_t nbytes = uint32_t(min(uint64_t(kTile), n - q0)) * k;
#pragma unroll
        for (int i = 0; i < kPieces; ++i) staged[i] = load_piece(kmers + q0 * k, nbytes, lane + 64u * i);
    }

    // The presence filter only pays when it rejects queries.  Each wave watches its own pass
    // rate: a tile in which >= 90 % of the looked-up queries passed switches the filter off for
    // the next 7 tiles, then it is probed again (all wave-uniform).
    uint32_t filter_pause = 0;

    for (uint64_t tile = wave_id; tile < ntiles; tile += nwaves) {
        const uint64_t q0 = tile * kTile;
        const uint32_t in_tile = uint32_t(min(uint64_t(kTile), n - q0));
        const bool filter_now = filter != nullptr && filter_pause == 0;
        bool looked_up = false, passed = false;
        // ---- phase 1a: the tile's bytes (contiguous, 16-byte aligned) go through LDS ----
        if (!kReads) {
            if (!kPrefetch) {
#pragma unroll
                for (int i = 0; i < kPieces; ++i) staged[i] = load_piece(kmers + q0 * k, in_tile * k, lane + 64u * i);
            }
#pragma unroll
            for (int i = 0; i < kPieces; ++i) ws.stage[kStageLead / 16 + lane + 64u * i] = staged[i];
            const uint64_t next_tile = tile + nwaves;
            if (kPrefetch && next_tile < ntiles) {
                const uint64_t nq0 = next_tile * kTile;
                const uint32_t nbytes = uint32_t(min(uint64_t(kTile), n - nq0)) * k;
#pragma unroll
                for (int i = 0; i < kPieces; ++i) staged[i] = load_piece(kmers + nq0 * k, nbytes, lane + 64u * i);
            }
        }
        wave_lds_sync();
        // ---- phase 1b: one lane = one query: validate, pack, table lookup ----
        bool pending = false;
        uint64_t l = 0, h = total;
        uint32_t w[kWords], rem = k;
#pragma unroll
        for (int i = 0; i < kWords; ++i) w[i] = 0;
        if (lane < in_tile) {
            uint64_t result = 0;
            pending = prepare_query<kReads, kWords>(src, env, stage_bytes + lane * k, q0 + lane, filter_now, flags, l, h, w, rem,
                                                    result, looked_up, passed);
            if (!pending) ws.result[lane] = result;
        }
        if (filter != nullptr) {
            if (filter_now) {
                const uint32_t nlook = uint32_t(__popcll(__ballot(looked_up))), npass = uint32_t(__popcll(__ballot(passed)));
                if (nlook > 0 && npass * 10u >= nlook * 9u) filter_pause = 7;
            } else {
                --filter_pause;
            }
        }
        // compact the undecided queries into the work list: ballot + prefix count
        const uint64_t pend_mask = __ballot(pending);
        const uint32_t nwork = uint32_t(__popcll(pend_mask));
        if (pending) {
            const uint32_t at = __builtin_amdgcn_mbcnt_hi(uint32_t(pend_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(pend_mask), 0u));
            WorkItem it;
            it.l_lo = uint32_t(l); it.l_hi = uint32_t(l >> 32);
            it.h_lo = uint32_t(h); it.h_hi = uint32_t(h >> 32);
#pragma unroll
            for (int i = 0; i < kWords; ++i) it.w[i] = w[i];
            it.rem_slot = rem | (lane << 8);
            ws.work[at] = it;
        }
        wave_lds_sync();
        // ---- phase 2: groups pull work items; one backward-search step per iteration ----
        {
            uint32_t next = 0;  // wave-uniform: first unassigned work item
            bool have = false;
            uint32_t slot = 0;
            for (;;) {
                // hand the next items to the idle groups, in group order -- only when there is
                // an idle group and work left (wave-uniform test on the busy mask)
                uint64_t busy = __ballot(have);
                if (busy != ~0ull && next < nwork) {
                    const uint64_t idle = ~busy & kGroupLeaders;  // one bit per idle group (its first lane)
                    const uint32_t idle_before = uint32_t(__popcll(idle & ((1ull << group_first_lane) - 1ull)));
                    if (!have && next + idle_before < nwork) {
                        const uint4 *it = reinterpret_cast<const uint4 *>(&ws.work[next + idle_before]);
                        const uint4 a = it[0], b = it[1];
                        l = (uint64_t(a.y) << 32) | a.x;
                        h = (uint64_t(a.w) << 32) | a.z;
                        uint32_t rem_slot;
                        if constexpr (kWords == 3) {
                            w[0] = b.x; w[1] = b.y; w[2] = b.z;
                            rem_slot = b.w;
                        } else {
                            const uint4 c = it[2];
                            w[0] = b.x; w[1] = b.y; w[2] = b.z; w[3] = b.w; w[4] = c.x; w[5] = c.y;
                            rem_slot = c.z;
                        }
                        rem = rem_slot & 0xFFu;
                        slot = rem_slot >> 8;
                        have = true;
                    }
                    next = min(nwork, next + uint32_t(__popcll(idle)));
                    busy = __ballot(have);
                }
                if (busy == 0ull) break;
                if (have && (h > total || l > h)) {
                    // a range outside the index would turn into a wild block address: end such a query with u64::MAX and
                    // MSBWT_ERR_INTERNAL instead (never seen on a well-formed index; the lanes kernel has the same guard)
                    if (sub == 0u) {
                        atomicOr(flags, kFlagInternal);
                        ws.result[slot] = ~0ull;
                    }
                    have = false;
                }
                if (have) {
                    const Range r = format == 0u ? constrain_split(blocks, w[0] & 7u, l, h, sub)
                                                 : constrain_any(format, blocks, overflow, w[0] & 7u, l, h, sub);
                    l = r.l;
                    h = r.h;
                    consume_symbols<kWords>(w, 3);
                    --rem;
                    if (rem == 0u || l == h) {
                        if (sub == 0u) ws.result[slot] = h - l;
                        have = false;
                    }
                }
            }
        }
        wave_lds_sync();
        if (lane < in_tile) store_count<kReads>(src, q0 + lane, ws.result[lane]);
        wave_lds_sync();  // the next tile must not overwrite result[] / stage[] before this
    }
}

// ---- suffix table, built level by level in place ------------------------------------------
// Level j holds the range of every ACGT string of length j, indexed by sum(two(x_t) << 2t)
// with x_0 the LAST symbol of the k-mer (the first one searched).  A group reads one parent
// (level j-1, index p) and writes its four children p + c * 4^(j-1); child 0 overwrites the
// parent only after the group has read it.
__global__ __launch_bounds__(256) void k_table_level(const uint4 *__restrict__ blocks, uint32_t format,
                                                     const uint4 *__restrict__ overflow, uint4 *__restrict__ table, uint32_t level) {
    const uint32_t sub = threadIdx.x & (kGroup - 1);
    const uint64_t parents = 1ull << (2u * (level - 1u));
    const uint64_t ngroups = (uint64_t(gridDim.x) * blockDim.x) / kGroup;
    for (uint64_t p = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) / kGroup; p < parents; p += ngroups) {
        const uint4 e = table[p];
        const uint64_t l = (uint64_t(e.y) << 32) | e.x, h = (uint64_t(e.w) << 32) | e.z;
        Range mine{0, 0};  // lane c of the group keeps child c
        for (uint32_t c = 0; c < 4u; ++c) {
            const uint32_t s = c == 3u ? 5u : c + 1u;  // A C G T
            const Range r = (l == h) ? Range{0, 0} : constrain_any(format, blocks, overflow, s, l, h, sub);
            if (sub == c) mine = r;
        }
        if (sub < 4u)
            table[p + uint64_t(sub) * parents] =
                make_uint4(uint32_t(mine.l), uint32_t(mine.l >> 32), uint32_t(mine.h), uint32_t(mine.h >> 32));
    }
}

// bit (j & mask) of the presence filter is set when table entry j is a non-empty range
__global__ __launch_bounds__(256) void k_table_filter(const uint4 *__restrict__ table, uint64_t entries, uint32_t mask,
                                                      uint32_t *__restrict__ filter) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t j = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < entries; j += stride) {
        const uint4 e = table[j];
        if (e.x != e.z || e.y != e.w) {
            const uint32_t i = uint32_t(j) & mask;
            atomicOr(&filter[i >> 5], 1u << (i & 31u));
        }
    }
}

// ---- packed table: two more levels in the HBM of one -------------------------------------------
// occ2(a, b, pos) + K[a][b] by ONE thread straight from global memory (pair block layout: rank_ops.hpp)
__device__ __forceinline__ uint64_t pair_bound_thread(const uint4 *__restrict__ pair_blocks, const uint64_t *__restrict__ pair_super,
                                                      bool stride96, uint32_t a2, uint32_t b2, uint64_t pos) {
    const uint64_t pb = pair_block_of(pos, stride96);
    const uint4 *blk = pair_blocks + pb * 8;
    const uint4 a0 = blk[0], a1 = blk[1], b0 = blk[2], b1 = blk[3], v = blk[kPairValidChunk];
    const uint32_t na0 = (a2 & 1u) - 1u, na1 = ((a2 >> 1) & 1u) - 1u, nb0 = (b2 & 1u) - 1u, nb1 = ((b2 >> 1) & 1u) - 1u;
    const uint32_t m0 = (a0.x ^ na0) & (a1.x ^ na1) & (b0.x ^ nb0) & (b1.x ^ nb1) & v.x;
    const uint32_t m1 = (a0.y ^ na0) & (a1.y ^ na1) & (b0.y ^ nb0) & (b1.y ^ nb1) & v.y;
    const uint32_t m2 = (a0.z ^ na0) & (a1.z ^ na1) & (b0.z ^ nb0) & (b1.z ^ nb1) & v.z;
    const uint32_t m3 = (a0.w ^ na0) & (a1.w ^ na1) & (b0.w ^ nb0) & (b1.w ^ nb1) & v.w;
    const uint32_t r = uint32_t(pos - pair_block_start(pb, stride96)), p = a2 * 4u + b2;
    const uint64_t t = (1ull << (r & 63u)) - 1ull;
    const bool upper = r >= 64u;
    const uint64_t lo = upper ? ~0ull : t, hi = upper ? t : 0ull;
    const uint32_t cnt = uint32_t(__popc(m0 & uint32_t(lo))) + uint32_t(__popc(m1 & uint32_t(lo >> 32))) +
                         uint32_t(__popc(m2 & uint32_t(hi))) + uint32_t(__popc(m3 & uint32_t(hi >> 32)));
    const uint32_t field = uint32_t(reinterpret_cast<const uint16_t *>(blk + kPairLoChunk)[p]) |
                           (uint32_t(reinterpret_cast<const uint8_t *>(blk + kPairHiChunk)[p]) << 16);
    return pair_super[(pb >> kPairSuperBlocks) * 16u + p] + field + cnt;
}

// One half-wave (32 lanes) writes one packed line: lane i < 30 extends flat entry
// (t mod 4^flat_depth) by the two symbols in t's top four bits, t = 30 line + i.
__global__ __launch_bounds__(256) void k_table_pack(const uint4 *__restrict__ flat, uint32_t flat_depth,
                                                    const uint4 *__restrict__ pair_blocks, const uint64_t *__restrict__ pair_super,
                                                    uint32_t stride96, uint32_t *__restrict__ packed, uint64_t nlines,
                                                    unsigned long long *__restrict__ escape_count, uint4 *__restrict__ side,
                                                    unsigned long long *__restrict__ side_cursor) {
    // Second pass (side != nullptr): only the lines the first pass marked ESCAPE are visited again; each takes the next
    // free group of the side array, stores its 30 ranges there as flat {l, h} entries and names the group in its base word.
    const uint32_t lane = threadIdx.x & 63u, i = lane & 31u, team_first = lane & 32u;
    const uint64_t entries = 1ull << (2u * (flat_depth + 2u)), parent_mask = (1ull << (2u * flat_depth)) - 1ull;
    const uint64_t nteams = (uint64_t(gridDim.x) * blockDim.x) / 32;
    for (uint64_t line = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) / 32; line < nlines; line += nteams) {
        const uint64_t t = line * kPackedPerLine + i;
        const bool valid = i < kPackedPerLine && t < entries;
        uint64_t l = 0, h = 0;
        if (side != nullptr && (packed[line * 32 + 1] & 0x80000000u) == 0u) continue;  // (team-uniform)
        if (valid) {
            const uint4 e = flat[t & parent_mask];
            l = (uint64_t(e.y) << 32) | e.x;
            h = (uint64_t(e.w) << 32) | e.z;
            if (l != h) {
                const uint32_t a2 = uint32_t(t >> (2u * flat_depth)) & 3u, b2 = uint32_t(t >> (2u * flat_depth + 2u)) & 3u;
                l = pair_bound_thread(pair_blocks, pair_super, stride96 != 0u, a2, b2, l);
                h = pair_bound_thread(pair_blocks, pair_super, stride96 != 0u, a2, b2, h);
            }
        }
        const bool nonempty = valid && l != h;
        // base = the first non-empty range's l (ranges of consecutive indices are consecutive, so l is monotone)
        const uint64_t ne = __ballot(nonempty);
        const uint32_t mine32 = uint32_t(ne >> team_first);
        const int first = mine32 ? __ffs(int(mine32)) - 1 : 0;
        const uint64_t base = (uint64_t(uint32_t(__shfl(int(uint32_t(l >> 32)), int(team_first) + first))) << 32) |
                              uint32_t(__shfl(int(uint32_t(l)), int(team_first) + first));
        const uint64_t dl = nonempty ? l - base : 0ull, w = nonempty ? h - l : 0ull;
        const bool wide = dl > 0xFFFFull || w > 0xFFFFull;
        const bool escape = (uint32_t(__ballot(wide) >> team_first)) != 0u;
        uint32_t *out = packed + line * 32;
        if (side != nullptr) {  // an escape line (see above): flat entries into its group of the side array
            unsigned long long g = 0;
            if (i == 30u) g = atomicAdd(side_cursor, 1ull);
            const uint64_t group = (uint64_t(uint32_t(__shfl(int(uint32_t(g >> 32)), int(team_first) + 30))) << 32) |
                                   uint32_t(__shfl(int(uint32_t(g)), int(team_first) + 30));
            if (i < kPackedPerLine) side[group * kSidePerLine + i] = make_uint4(uint32_t(l), uint32_t(l >> 32), uint32_t(h), uint32_t(h >> 32));
            if (i == 30u) {
                const uint64_t b = kPackedEscape | group;
                out[0] = uint32_t(b);
                out[1] = uint32_t(b >> 32);
            }
            continue;
        }
        if (i < kPackedPerLine) out[2u + i] = uint32_t(dl & 0xFFFFu) | (uint32_t(w & 0xFFFFu) << 16);
        if (i == 30u) {
            const uint64_t b = (mine32 ? base : 0ull) | (escape ? kPackedEscape : 0ull);
            out[0] = uint32_t(b);
            out[1] = uint32_t(b >> 32);
            if (escape && escape_count != nullptr) atomicAdd(escape_count, 1ull);
        }
    }
}

// ---- what the DATA look like: how wide is the range of a k-mer that is present? -------------------------------
// Sample g starts at a pseudo-random row r and walks backwards: the symbol stored at row r (the one that precedes
// suffix r in the text) is prepended to the pattern -- one step of the backward search [l, h) -> constrain(s, [l, h))
// -- and r moves on to LF(r), which lies inside the new range by construction.  After `steps` symbols h - l is the
// number of occurrences of a `steps`-mer that is known to occur: about the coverage on a real read set, 1 on a
// stream of independent symbols.  out[g] = 0 for walks that met '$' / 'N'.  Plane blocks only.
__global__ __launch_bounds__(256) void k_probe_widths(const uint4 *__restrict__ blocks, uint64_t total, uint32_t nsamples, uint32_t steps,
                                                      uint64_t seed, uint64_t *__restrict__ out) {
    const uint32_t sub = threadIdx.x & (kGroup - 1), lane = threadIdx.x & 63u;
    const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (g >= nsamples || total == 0) return;
    uint64_t z = seed + uint64_t(g + 1u) * 0x9E3779B97F4A7C15ull;  // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    uint64_t r = z % total, l = 0, h = total;
    bool ok = true;
    for (uint32_t t = 0; t < steps && ok; ++t) {
        const uint4 c = blocks[(r >> 8) * kGroup + sub];
        const uint32_t bit = uint32_t(r) & 31u;
        const uint32_t mine = ((c.x >> bit) & 1u) | (((c.y >> bit) & 1u) << 1) | (((c.z >> bit) & 1u) << 2);
        const uint32_t s = uint32_t(__shfl(int(mine), int((lane & ~7u) + ((uint32_t(r) & 255u) >> 5))));
        if (!is_acgt(s)) { ok = false; break; }  // group-uniform
        r = constrain(blocks, s, r, r + 1, sub).l;
        const Range q = constrain(blocks, s, l, h, sub);
        l = q.l;
        h = q.h;
    }
    if (sub == 0u) out[g] = ok ? h - l : 0ull;
}

// ---- how fast does the memory system serve random 128-byte lines of THIS allocation? ---------------------------------------------
// (msbwt_rle_probe_line_rate: a diagnostic.  Round 5 asked it whether the "two modes" of the C4-sized lines are visible to a plain
// gather over the arrays involved -- they are not, profiles/r05_lab/two_modes.log.)  8 lanes read one line, 8 independent lines in flight per lane, `iters`
// rounds; the lines follow a counter-based hash, the contents do not matter (nothing is written but a sink word that never is).
__global__ __launch_bounds__(256) void k_probe_lines(const uint4 *__restrict__ base, uint64_t nlines, uint32_t iters, uint32_t *__restrict__ sink) {
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, group = tid >> 3;
    const uint32_t piece = uint32_t(tid) & 7u;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            uint64_t z = (group * 0x9E3779B97F4A7C15ull) + uint64_t(it) * 8u + u;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            v[u] = base[__umul64hi(z, nlines) * 8u + piece];
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9E3779B9u && sink != nullptr) sink[0] = acc;  // (keeps the loads alive)
}

__global__ void k_table_root(uint4 *table, uint64_t total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) table[0] = make_uint4(0u, 0u, uint32_t(total), uint32_t(total >> 32));
}

__global__ __launch_bounds__(256) void k_constrain_ranges(const uint4 *__restrict__ blocks, uint32_t format,
                                                          const uint4 *__restrict__ overflow, uint64_t total,
                                                          const uint8_t *__restrict__ syms,
                                                          const uint64_t *__restrict__ l, const uint64_t *__restrict__ h,
                                                          uint64_t n, uint64_t *__restrict__ out_l,
                                                          uint64_t *__restrict__ out_h, uint32_t *__restrict__ flags,
                                                          uint64_t *__restrict__ done, uint64_t done_seq) {
    const uint32_t sub = threadIdx.x & (kGroup - 1);
    const uint64_t ngroups = (uint64_t(gridDim.x) * blockDim.x) / kGroup;
    for (uint64_t i = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) / kGroup; i < n; i += ngroups) {
        const uint32_t s = syms[i];
        const uint64_t li = l[i], hi = h[i];
        Range r{~0ull, ~0ull};
        uint32_t err = 0;
        if (s >= 6u) err = kFlagInvalidSymbol;
        else if (li > hi || hi > total) err = kFlagInvalidRange;
        if (err) {
            if (sub == 0) atomicOr(flags, err);
        } else {
            r = constrain_any(format, blocks, overflow, s, li, hi, sub);
        }
        if (sub == 0) {
            out_l[i] = r.l;
            out_h[i] = r.h;
        }
    }
    if (done != nullptr) {  // one-wave launches only (IndexView::done): the results are out before the word changes
        __threadfence_system();
        if (threadIdx.x == 0u) __hip_atomic_store(done, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Which search kernel (IndexView::search_kernel, msbwt_rle_set_search_kernel): the one-query-
// per-lane kernel (lanes.hip) for long searches -- many symbols left after the table, throughput
// set by random lines in flight -- and the 8-lane-group kernel above for short ones, where setup
// (staging, table lookups) dominates and its 32 waves per CU hide that latency better.
// Automatic choice: lanes.hip whenever the pair index exists and k >= 6.  Measured over C2, C4 and the
// human-scale index, k = 4..31, random and read-derived queries, batches of 10^5..3 x 10^8:
// the lanes kernel is 1.3-2x faster whenever queries survive the table (read-derived k-mers, any index in
// the HBM regime) and level with this file's kernel on random k-mers that end in the filter or the table;
// this kernel keeps very short k-mers (k < 6: wide ranges, two lines per step), batches without a pair
// index, and run blocks.
inline bool long_search(const IndexView &ix, uint32_t k) { return ix.pair_blocks != nullptr && k >= 6u; }

inline bool use_lanes_kernel(const IndexView &ix, uint32_t k) {
    // run blocks (round 4; k > 32 since round 5): the lanes kernel decodes them lane by lane for 6 <= k <= 64 (single-symbol steps);
    // shorter k-mers of that format stay with this file's kernel
    if (ix.block_format != kBlocksPlanes) return ix.search_kernel != kSearchGroups && k >= 6u && k <= uint32_t(kMaxTiledK);
    return ix.search_kernel == kSearchLanes || (ix.search_kernel == kSearchAuto && long_search(ix, k));
}

template <bool kReads>
void launch_tiled(bool longk, dim3 grid, hipStream_t stream, const IndexView &ix, const QuerySource &src, uint32_t *flags) {
    const uint4 *blocks = static_cast<const uint4 *>(ix.blocks);
    const uint4 *table = static_cast<const uint4 *>(ix.table.entries);
    const uint32_t depth = uint32_t(ix.table.depth);
    const uint32_t *filter = table ? ix.table.filter : nullptr;
    const uint32_t filter_mask = filter ? uint32_t((1ull << (2 * ix.table.filter_depth)) - 1ull) : 0u;
    const uint32_t packed = ix.table.packed ? 1u : 0u, format = uint32_t(ix.block_format);
    const uint4 *overflow = static_cast<const uint4 *>(ix.overflow);
    const uint4 *side = table ? static_cast<const uint4 *>(ix.table.side) : nullptr;
    if (longk)  // 33 <= k <= 64
        hipLaunchKernelGGL((k_count_kmers_tiled<kReads, 6>), grid, dim3(256), 0, stream, blocks, format, overflow, ix.total, table, depth, packed, filter, filter_mask, side, src, flags);
    else
        hipLaunchKernelGGL((k_count_kmers_tiled<kReads, 3>), grid, dim3(256), 0, stream, blocks, format, overflow, ix.total, table, depth, packed, filter, filter_mask, side, src, flags);
}

}  // namespace

int search_kernel_for(const IndexView &ix, uint32_t k) {
    if (k < 1 || k > uint32_t(kMaxTiledK)) return 0;
    return use_lanes_kernel(ix, k) ? kSearchLanes : kSearchGroups;
}

bool lanes_serves(const IndexView &ix, uint32_t k) { return k >= 1 && k <= uint32_t(kMaxTiledK) && use_lanes_kernel(ix, k); }

hipError_t launch_count_kmers(const IndexView &ix, const uint8_t *kmers, uint32_t k, uint64_t n,
                              uint64_t *counts, uint32_t *flags, hipStream_t stream, const uint8_t *inline_kmer) {
    if (n == 0) return hipSuccess;
    if (inline_kmer != nullptr && !(n == 1 && lanes_serves(ix, k))) return hipErrorInvalidValue;
    const uint4 *blocks = static_cast<const uint4 *>(ix.blocks);
    const bool aligned = (reinterpret_cast<uintptr_t>(kmers) & 15u) == 0;
    if (k >= 1 && k <= uint32_t(kMaxTiledK) && aligned) {
        const uint64_t tiles = (n + kTile - 1) / kTile;
        QuerySource src{};
        src.data = kmers;
        src.n = n;
        src.k = k;
        src.out_fwd = counts;
        if (inline_kmer != nullptr) {
            src.inline_n = 1;
            __builtin_memcpy(src.inline_kmer, inline_kmer, k);
        }
        if (use_lanes_kernel(ix, k)) return launch_lanes(ix, src, false, true, flags, stream);
        launch_tiled<false>(k > uint32_t(kMaxShortK), dim3(grid_for(tiles * 64, k > uint32_t(kMaxShortK) ? 4 : 6)), stream, ix, src, flags);
    } else {
        hipLaunchKernelGGL(k_count_kmers_generic, dim3(grid_for(n * kGroup)), dim3(256), 0, stream, blocks, uint32_t(ix.block_format),
                           static_cast<const uint4 *>(ix.overflow), ix.total, kmers, k, n, counts, flags);
    }
    return hipGetLastError();
}

hipError_t launch_count_packed(const IndexView &ix, const uint64_t *packed, uint32_t k, uint64_t n, uint64_t *counts,
                               const uint32_t *out_index, uint32_t *flags, hipStream_t stream, uint32_t stride_words, bool place_inline) {
    if (k < 1 || k > uint32_t(kMaxTiledK) || ix.block_format != kBlocksPlanes || ((out_index != nullptr || place_inline) && n > 0xFFFFFFFFull))
        return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    QuerySource src{};
    src.data = reinterpret_cast<const uint8_t *>(packed);
    src.n = n;
    src.k = k;
    src.out_fwd = counts;
    src.packed = 1;
    src.out_index = out_index;
    src.packed_stride = stride_words;
    src.place_inline = place_inline ? 1u : 0u;
    return launch_lanes(ix, src, false, true, flags, stream);
}

hipError_t launch_count_read_kmers(const IndexView &ix, const uint8_t *reads, uint32_t read_len, uint64_t n_reads,
                                   uint32_t k, bool ascii, uint64_t *out_fwd, uint64_t *out_rc, uint32_t *flags,
                                   hipStream_t stream) {
    if (k < 1 || k > uint32_t(kMaxTiledK) || k > read_len || (!out_fwd && !out_rc)) return hipErrorInvalidValue;
    if (n_reads == 0) return hipSuccess;
    QuerySource src{};
    src.data = reads;
    src.k = k;
    src.read_len = read_len;
    src.windows = read_len - k + 1;
    src.strands = (out_fwd ? 1u : 0u) | (out_rc ? 2u : 0u);
    src.ascii = ascii ? 1u : 0u;
    src.out_fwd = out_fwd;
    src.out_rc = out_rc;
    src.n_reads = n_reads;
    src.n = n_reads * src.windows * (src.strands == 3u ? 2u : 1u);
    if (use_lanes_kernel(ix, k)) return launch_lanes(ix, src, true, true, flags, stream);
    const uint64_t tiles = (src.n + kTile - 1) / kTile;
    launch_tiled<true>(k > uint32_t(kMaxShortK), dim3(grid_for(tiles * 64, k > uint32_t(kMaxShortK) ? 4 : 6)), stream, ix, src, flags);
    return hipGetLastError();
}

hipError_t launch_count_ragged_read_kmers(const IndexView &ix, const uint8_t *reads, const uint64_t *read_off,
                                          const uint64_t *win_off, uint64_t n_reads, uint64_t n_windows, uint32_t k,
                                          bool ascii, uint64_t *out_fwd, uint64_t *out_rc, uint32_t *flags,
                                          hipStream_t stream) {
    if (k < 1 || k > uint32_t(kMaxTiledK) || (!out_fwd && !out_rc)) return hipErrorInvalidValue;
    if (n_reads == 0 || n_windows == 0) return hipSuccess;
    QuerySource src{};
    src.data = reads;
    src.k = k;
    src.strands = (out_fwd ? 1u : 0u) | (out_rc ? 2u : 0u);
    src.ascii = ascii ? 1u : 0u;
    src.out_fwd = out_fwd;
    src.out_rc = out_rc;
    src.read_off = read_off;
    src.win_off = win_off;
    src.n_reads = n_reads;
    src.n = n_windows * (src.strands == 3u ? 2u : 1u);
    if (use_lanes_kernel(ix, k)) return launch_lanes(ix, src, true, true, flags, stream);
    const uint64_t tiles = (src.n + kTile - 1) / kTile;
    launch_tiled<true>(k > uint32_t(kMaxShortK), dim3(grid_for(tiles * 64, k > uint32_t(kMaxShortK) ? 4 : 6)), stream, ix, src, flags);
    return hipGetLastError();
}

hipError_t launch_constrain_ranges(const IndexView &ix, const uint8_t *syms, const uint64_t *l,
                                   const uint64_t *h, uint64_t n, uint64_t *out_l, uint64_t *out_h,
                                   uint32_t *flags, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const bool one_wave = ix.done != nullptr && n <= 64 / kGroup;  // a polled completion word needs a single wave
    hipLaunchKernelGGL(k_constrain_ranges, dim3(one_wave ? 1 : grid_for(n * kGroup)), dim3(one_wave ? 64 : 256), 0, stream,
                       static_cast<const uint4 *>(ix.blocks), uint32_t(ix.block_format), static_cast<const uint4 *>(ix.overflow), ix.total, syms,
                       l, h, n, out_l, out_h, flags, one_wave ? ix.done : nullptr, ix.done_seq);
    return hipGetLastError();
}

hipError_t launch_build_table(const IndexView &ix, int depth, void *entries, hipStream_t stream) {
    if (depth < 1 || depth > 16) return hipErrorInvalidValue;
    uint4 *table = static_cast<uint4 *>(entries);
    hipLaunchKernelGGL(k_table_root, dim3(1), dim3(64), 0, stream, table, ix.total);
    for (int level = 1; level <= depth; ++level) {
        const uint64_t parents = 1ull << (2 * (level - 1));
        hipLaunchKernelGGL(k_table_level, dim3(grid_for(parents * kGroup)), dim3(256), 0, stream,
                           static_cast<const uint4 *>(ix.blocks), uint32_t(ix.block_format), static_cast<const uint4 *>(ix.overflow), table,
                           uint32_t(level));
    }
    return hipGetLastError();
}

hipError_t launch_pack_table(const IndexView &ix, int flat_depth, const void *flat_entries, void *packed_entries,
                             unsigned long long *escape_count, void *side, unsigned long long *side_cursor, hipStream_t stream) {
    if (flat_depth < 1 || flat_depth > 16 || !ix.pair_blocks || !ix.pair_super || (side && !side_cursor)) return hipErrorInvalidValue;
    const uint64_t nlines = packed_table_bytes(flat_depth + 2) / 128;
    hipLaunchKernelGGL(k_table_pack, dim3(grid_for(nlines * 32)), dim3(256), 0, stream, static_cast<const uint4 *>(flat_entries),
                       uint32_t(flat_depth), static_cast<const uint4 *>(ix.pair_blocks), ix.pair_super, ix.pair_stride96 ? 1u : 0u,
                       static_cast<uint32_t *>(packed_entries), nlines, escape_count, static_cast<uint4 *>(side), side_cursor);
    return hipGetLastError();
}

hipError_t launch_probe_widths(const IndexView &ix, uint32_t nsamples, uint32_t steps, uint64_t seed, uint64_t *d_out, hipStream_t stream) {
    if (ix.block_format != kBlocksPlanes || nsamples == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_probe_widths, dim3((nsamples * kGroup + 255) / 256), dim3(256), 0, stream, static_cast<const uint4 *>(ix.blocks), ix.total,
                       nsamples, steps, seed, d_out);
    return hipGetLastError();
}

hipError_t launch_probe_lines(const void *base, uint64_t bytes, uint32_t iters, uint64_t *lines_touched, uint32_t *sink, hipStream_t stream) {
    const uint64_t nlines = bytes / 128;
    if (!base || nlines == 0) return hipErrorInvalidValue;
    const uint32_t blocks = 256u * 8u;  // 8 resident blocks of 256 threads per CU
    hipLaunchKernelGGL(k_probe_lines, dim3(blocks), dim3(256), 0, stream, static_cast<const uint4 *>(base), nlines, iters, sink);
    if (lines_touched) *lines_touched = uint64_t(blocks) * 256u / 8u * iters * 8u;
    return hipGetLastError();
}

hipError_t launch_build_filter(const void *entries, int depth, int filter_depth, uint32_t *filter, hipStream_t stream) {
    if (depth < 1 || filter_depth < 1 || filter_depth > depth || filter_depth > 16) return hipErrorInvalidValue;
    const uint64_t n = 1ull << (2 * depth);
    const uint32_t mask = uint32_t((1ull << (2 * filter_depth)) - 1ull);
    hipLaunchKernelGGL(k_table_filter, dim3(grid_for(n)), dim3(256), 0, stream, static_cast<const uint4 *>(entries), n, mask, filter);
    return hipGetLastError();
}

}  // namespace msbwt
