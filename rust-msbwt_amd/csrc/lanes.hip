// gfx950 (MI355X, CDNA4): count_kmers with ONE QUERY PER LANE and the index lines staged through
// LDS -- the search kernel for long searches (k-mers with many symbols left after the suffix
// table), where throughput is set by how many random 128-byte lines are in flight.
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
//      -- 64..128 lines in flight per wave instead of 8..16;
//   3. after s_waitcnt vmcnt(0) every lane reads its own line(s) back from LDS (bank-conflict
//      free, see line_base) and ranks both bounds itself: XOR / AND / popcount on the bit planes,
//      no cross-lane reduction at all.
//
// Lanes whose query is finished take the next undecided query from a ring in LDS that phase 1
// (stage the tile's bytes, validate, pack, suffix-table lookup -- the same code as the tiled
// kernel, search_common.hpp) keeps topped up, so all 64 lanes stay busy whatever the mix of
// early exits.  Counts go straight to the caller's buffer.  Block layouts: plane_index.hpp,
// rank_ops.hpp.  One wave per workgroup, 22 KiB of LDS each: 7 waves per CU.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.hpp"
#include "rank_ops.hpp"
#include "search_common.hpp"

namespace msbwt {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_void;

constexpr int kRing = 128;       // undecided queries waiting for a lane (>= 64 left over + 64 new)
constexpr int kRegions = 16;     // LDS-DMA regions of 8 lines: 64 first-bound lines + up to 64 second-bound lines
constexpr int kLineSlots = kRegions * 8;

template <int kWords>
struct LaneScratchT {
    static constexpr int kMaxK = kWords * 32 / 3;  // 32 or 64
    // Region i (one LDS-DMA instruction: lane j writes 16 bytes at 16 j) starts at uint4 index
    // region_base(i): every odd region is pushed 128 bytes further, so that the 64 lanes'
    // read-back of "chunk j of my line" touches every bank exactly once per 16 lanes (lanes 16 m ..
    // 16 m + 15 own the lines of regions 2 m and 2 m + 1, whose bank phases differ by 128 bytes).
    // During phase 1 the same memory stages the tile's query bytes (2 or 4 KiB).
    uint4 lines[(kRegions / 2) * 136];   // 17 KiB
    uint64_t list[kLineSlots];        // this step's line addresses, 0 = none (1 KiB)
    WorkItemT<kWords> ring[kRing];    // 4 or 6 KiB
};

// uint4 index of the first of region i's 64 sixteen-byte pieces: pairs of regions take 136 pieces,
// the odd one starting 72 in (64 + 8 of padding)
__host__ __device__ constexpr uint32_t region_base(uint32_t i) { return 136u * (i >> 1) + 72u * (i & 1u); }

// uint4 index of chunk 0 of the line in slot `s` (chunk j sits at line_base(s) + (j ^ (s & 7)))
__device__ __forceinline__ uint32_t line_base(uint32_t s) { return region_base(s >> 3) + 8u * (s & 7u); }

__device__ __forceinline__ uint32_t below(int r, int first) {  // mask of the bits [0, r - first) clamped to a word
    const int n = r - first;
    return n <= 0 ? 0u : (n >= 32 ? ~0u : ((1u << n) - 1u));
}

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
        L.m[j] = (c.x ^ x0) & (c.y ^ x1) & (c.z ^ x2);
        L.meta[j] = c.w;
    }
}

// start_index[s] + rank(s, pos) from the line of pos's block (plane_index.hpp layout)
__device__ __forceinline__ uint64_t plane_line_bound(const PlaneLine &L, uint32_t s, uint64_t pos) {
    const int r = int(uint32_t(pos) & 255u);
    uint32_t cnt = 0, lo = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        cnt += uint32_t(__popc(L.m[j] & below(r, 32 * j)));
        if (j < 6) lo = (s == uint32_t(j)) ? L.meta[j] : lo;
    }
    const uint32_t hi = (((s >> 2) ? L.meta[7] : L.meta[6]) >> ((s & 3u) * 8u)) & 0xFFu;
    return ((uint64_t(hi) << 32) | lo) + cnt;
}

// ---- rank of one bound from a staged pair-block line (two-symbol step, rank_ops.hpp) -------------
struct PairLine {
    uint32_t m[4];   // per 32 positions: 1 where (S, S2) == (a, b)
    uint32_t field;  // the block's 24-bit header count of the pair
};

__device__ __forceinline__ void read_pair_line(const uint4 *lines, uint32_t slot, uint32_t a2, uint32_t b2, PairLine &L) {
    const uint32_t base = line_base(slot), g = slot & 7u, p = a2 * 4u + b2;
    const uint32_t pa = ((a2 & 1u) ? 0x0000FFFFu : 0u) | ((a2 & 2u) ? 0xFFFF0000u : 0u);
    const uint32_t pb = ((b2 & 1u) ? 0x0000FFFFu : 0u) | ((b2 & 2u) ? 0xFFFF0000u : 0u);
    L.field = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const uint4 c = lines[base + (j ^ g)];
        uint32_t t = ~(c.x ^ pa) & ~(c.y ^ pb);  // 1 where a plane bit equals the wanted bit (both symbols)
        t = t & (t >> 16) & c.z & 0xFFFFu;       // both planes of both symbols, and the position is valid
        if (j & 1u) L.m[j >> 1] |= t << 16; else L.m[j >> 1] = t;
        L.field = ((p >> 1) == j) ? pair_chunk_field(c, p) : L.field;
    }
}

__device__ __forceinline__ uint64_t pair_line_bound(const PairLine &L, uint64_t super_base, uint64_t pos) {
    const int r = int(uint32_t(pos) & 127u);
    uint32_t cnt = L.field;
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt += uint32_t(__popc(L.m[w] & below(r, 32 * w)));
    return super_base + cnt;
}

template <bool kReads, bool kPair, int kWords>
__global__ __launch_bounds__(64) void k_count_kmers_lanes(const uint4 *__restrict__ blocks, uint64_t total,
                                                          const uint4 *__restrict__ table, uint32_t depth,
                                                          const uint32_t *__restrict__ filter, uint32_t filter_mask,
                                                          const uint4 *__restrict__ pair_blocks,
                                                          const uint64_t *__restrict__ pair_super, const QuerySource src,
                                                          uint32_t *__restrict__ flags, uint64_t *__restrict__ debug) {
    using Scratch = LaneScratchT<kWords>;
    using WorkItem = WorkItemT<kWords>;
    constexpr int kPieces = Scratch::kMaxK / 16;  // 16-byte pieces of a tile per lane: 2 or 4
    __shared__ Scratch ws;
    const uint8_t *__restrict__ kmers = src.data;
    const uint32_t k = src.k;
    const uint64_t n = src.n;
    const uint32_t lane = threadIdx.x;
    const uint8_t *stage_bytes = reinterpret_cast<const uint8_t *>(ws.lines);
    const uint64_t ntiles = (n + kTile - 1) / kTile;
    const uint64_t wave_id = blockIdx.x, nwaves = gridDim.x;
    const bool use_table = table != nullptr && depth > 0 && k >= depth;
    const TableEnv env{table, depth, use_table, filter, filter_mask, total};
    // this lane's part in the line fetches: 16 bytes (chunk dma_chunk) of the line in list slot 8 i + dma_group
    const uint32_t dma_group = lane >> 3, dma_chunk_bytes = ((lane & 7u) ^ dma_group) * 16u;

    uint64_t next_tile = wave_id;
    uint32_t seq = 0;                    // tiles this wave has set up so far (slot = seq * 64 + lane of the tile)
    uint32_t ring_head = 0, ring_count = 0;  // wave-uniform
    uint32_t filter_pause = 0;           // as in the tiled kernel: the filter rests while nearly everything passes

    bool have = false;
    uint64_t l = 0, h = 0;
    uint32_t w[kWords], rem = 0, slot = 0;
#pragma unroll
    for (int i = 0; i < kWords; ++i) w[i] = 0;

    for (;;) {
        // ---- phase 1: set up tiles until at least 64 undecided queries wait (or no tiles are left) ----
        while (ring_count < 64u && next_tile < ntiles) {
            const uint64_t tile = next_tile;
            next_tile += nwaves;
            const uint64_t q0 = tile * kTile;
            const uint32_t in_tile = uint32_t(min(uint64_t(kTile), n - q0));
            const bool filter_now = filter != nullptr && filter_pause == 0;
            bool looked_up = false, passed = false;
            if (!kReads) {  // the tile's bytes (contiguous, 16-byte aligned) go through LDS
                uint4 staged[kPieces];
#pragma unroll
                for (int i = 0; i < kPieces; ++i) staged[i] = load_piece(kmers + q0 * k, in_tile * k, lane + 64u * i);
#pragma unroll
                for (int i = 0; i < kPieces; ++i) ws.lines[lane + 64u * i] = staged[i];
            }
            wave_lds_sync();
            bool pending = false;
            uint64_t pl = 0, ph = 0, result = 0;
            uint32_t pw[kWords], prem = 0;
#pragma unroll
            for (int i = 0; i < kWords; ++i) pw[i] = 0;
            if (lane < in_tile) {
                pending = prepare_query<kReads, kWords>(src, env, stage_bytes + lane * k, q0 + lane, filter_now, flags, pl, ph,
                                                        pw, prem, result, looked_up, passed);
                if (!pending) store_count<kReads>(src, q0 + lane, result);
            }
            if (filter != nullptr) {
                if (filter_now) {
                    const uint32_t nlook = uint32_t(__popcll(__ballot(looked_up))), npass = uint32_t(__popcll(__ballot(passed)));
                    if (nlook > 0 && npass * 10u >= nlook * 9u) filter_pause = 7;
                } else {
                    --filter_pause;
                }
            }
            const uint64_t pend_mask = __ballot(pending);
            if (pending) {
                const uint32_t at = __builtin_amdgcn_mbcnt_hi(uint32_t(pend_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(pend_mask), 0u));
                WorkItem it;
                it.l_lo = uint32_t(pl); it.l_hi = uint32_t(pl >> 32);
                it.h_lo = uint32_t(ph); it.h_hi = uint32_t(ph >> 32);
#pragma unroll
                for (int i = 0; i < kWords; ++i) it.w[i] = pw[i];
                it.rem_slot = prem | ((seq * 64u + lane) << 8);
                ws.ring[(ring_head + ring_count + at) & (kRing - 1)] = it;
            }
            ring_count += uint32_t(__popcll(pend_mask));
            ++seq;
            wave_lds_sync();  // the ring entries are visible; the staged bytes may be overwritten
        }
        // ---- idle lanes take the waiting queries, in lane order ----
        uint64_t busy = __ballot(have);
        if (busy != ~0ull && ring_count > 0u) {
            const uint64_t idle = ~busy;
            const uint32_t my = __builtin_amdgcn_mbcnt_hi(uint32_t(idle >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle), 0u));
            if (!have && my < ring_count) {
                const uint4 *it = reinterpret_cast<const uint4 *>(&ws.ring[(ring_head + my) & (kRing - 1)]);
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
            const uint32_t taken = min(ring_count, uint32_t(__popcll(idle)));
            ring_head = (ring_head + taken) & (kRing - 1);
            ring_count -= taken;
            busy = __ballot(have);
        }
        // a range outside the index would turn into a wild line address: end such a query with
        // u64::MAX and a status flag instead (never seen on a well-formed index; cheap insurance)
        const bool broken = have && (h > total || l > h);
        if (broken) {
            atomicOr(flags, kFlagInternal);
            if (debug != nullptr && atomicCAS(reinterpret_cast<unsigned long long *>(debug), 0ull, 1ull) == 0ull) {
                debug[1] = l;
                debug[2] = h;
                debug[3] = (uint64_t(rem) << 32) | slot;
                debug[4] = (uint64_t(w[1]) << 32) | w[0];
                debug[5] = (uint64_t(blockIdx.x) << 32) | lane;
            }
            store_count<kReads>(src, (wave_id + uint64_t(slot >> 6) * nwaves) * kTile + (slot & 63u), ~0ull);
            have = false;
        }
        busy = __ballot(have);
        if (busy == 0ull) break;  // nothing in flight, nothing waiting, no tiles left

        // ---- one search step of every busy lane ----
        const uint32_t s1 = w[0] & 7u, s2 = (w[0] >> 3) & 7u;
        const bool pair = kPair && have && rem >= 2u && (acgt_bit(s1) & acgt_bit(s2)) != 0u;
        const uint32_t a2 = acgt_code(s1) & 3u, b2 = acgt_code(s2) & 3u;
        const uint32_t shift = pair ? uint32_t(kPairShift) : 8u;
        const uint64_t base = pair ? reinterpret_cast<uint64_t>(pair_blocks) : reinterpret_cast<uint64_t>(blocks);
        const uint64_t bl = l >> shift, bh = h >> shift;
        const bool second = have && bh != bl;
        const uint64_t second_mask = __ballot(second);
        const uint32_t nsecond = uint32_t(__popcll(second_mask));
        const uint32_t second_rank = __builtin_amdgcn_mbcnt_hi(uint32_t(second_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(second_mask), 0u));
        const uint32_t slot_l = lane, slot_h = second ? 64u + second_rank : lane;
        ws.list[lane] = have ? base + bl * 128u : 0ull;
        if (second) ws.list[slot_h] = base + bh * 128u;
        uint64_t super_l = 0, super_h = 0;  // pair steps: K[a][b] + occ2 at the superblock start (L2-resident table)
        if (pair) {
            const uint32_t p = a2 * 4u + b2;
            super_l = pair_super[(l >> kPairSuperShift) * 16u + p];
            super_h = pair_super[(h >> kPairSuperShift) * 16u + p];
        }
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < kRegions; ++i) {
            const bool wanted = i < 8 ? ((busy >> (8 * i)) & 0xFFull) != 0ull : nsecond > uint32_t(8 * (i - 8));  // wave-uniform
            if (wanted) {
                const uint32_t idx = 8u * i + dma_group;
                uint64_t a = ws.list[idx];
                if (i >= 8 && idx >= 64u + nsecond) a = 0;  // stale entry of an earlier step
                if (a != 0ull)
                    __builtin_amdgcn_global_load_lds((global_void *)(a + dma_chunk_bytes), (lds_void *)&ws.lines[region_base(i)], 16, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): every line has landed in LDS
        wave_lds_sync();
        if (have) {
            uint64_t nl, nh;
            if (pair) {
                PairLine L;
                read_pair_line(ws.lines, slot_l, a2, b2, L);
                nl = pair_line_bound(L, super_l, l);
                if (second) read_pair_line(ws.lines, slot_h, a2, b2, L);
                nh = pair_line_bound(L, super_h, h);
                consume_symbols<kWords>(w, 6);
                rem -= 2u;
            } else {
                PlaneLine L;
                read_plane_line(ws.lines, slot_l, s1, L);
                nl = plane_line_bound(L, s1, l);
                if (second) read_plane_line(ws.lines, slot_h, s1, L);
                nh = plane_line_bound(L, s1, h);
                consume_symbols<kWords>(w, 3);
                --rem;
            }
            l = nl;
            h = nh;
            if (rem == 0u || l == h) {
                const uint64_t q = (wave_id + uint64_t(slot >> 6) * nwaves) * kTile + (slot & 63u);
                store_count<kReads>(src, q, h - l);
                have = false;
            }
        }
        wave_lds_sync();  // the next step (or phase 1) overwrites the lines
    }
}

// The wave's 24-bit slot numbers allow 2^18 tiles per wave and launch
constexpr uint64_t kMaxTilesPerWave = 1ull << 18;

template <bool kReads>
void launch_shape(bool pair, bool longk, dim3 grid, hipStream_t stream, const IndexView &ix, const QuerySource &src,
                  uint32_t *flags) {
    const uint4 *blocks = static_cast<const uint4 *>(ix.blocks);
    const uint4 *table = static_cast<const uint4 *>(ix.table.entries);
    const uint32_t depth = uint32_t(ix.table.depth);
    const uint32_t *filter = table ? ix.table.filter : nullptr;
    const uint32_t filter_mask = filter ? uint32_t((1ull << (2 * ix.table.filter_depth)) - 1ull) : 0u;
    const uint4 *pair_blocks = static_cast<const uint4 *>(ix.pair_blocks);
#define MSBWT_LAUNCH(P, W) \
    hipLaunchKernelGGL((k_count_kmers_lanes<kReads, P, W>), grid, dim3(64), 0, stream, blocks, ix.total, table, depth, filter, filter_mask, pair_blocks, ix.pair_super, src, flags, ix.debug)
    if (longk) {
        if (pair) MSBWT_LAUNCH(true, 6); else MSBWT_LAUNCH(false, 6);
    } else {
        if (pair) MSBWT_LAUNCH(true, 3); else MSBWT_LAUNCH(false, 3);
    }
#undef MSBWT_LAUNCH
}

}  // namespace

hipError_t launch_lanes(const IndexView &ix, const QuerySource &src, bool reads, bool pair, uint32_t *flags,
                        hipStream_t stream) {
    if (src.k < 1 || src.k > uint32_t(kMaxTiledK)) return hipErrorInvalidValue;
    if (!reads && (reinterpret_cast<uintptr_t>(src.data) & 15u) != 0) return hipErrorInvalidValue;
    if (src.n == 0) return hipSuccess;
    pair = pair && ix.pair_blocks != nullptr;
    const uint64_t tiles = (src.n + kTile - 1) / kTile;
    // 7 one-wave workgroups per CU (LDS-bound) on 256 CUs; fewer for small batches
    const uint64_t waves = tiles < 7 * 256 ? tiles : 7 * 256;
    if ((tiles + waves - 1) / waves > kMaxTilesPerWave) return hipErrorInvalidValue;  // > 3e10 queries: split the batch
    if (reads) launch_shape<true>(pair, src.k > uint32_t(kMaxShortK), dim3(uint32_t(waves)), stream, ix, src, flags);
    else launch_shape<false>(pair, src.k > uint32_t(kMaxShortK), dim3(uint32_t(waves)), stream, ix, src, flags);
    return hipGetLastError();
}

}  // namespace msbwt
