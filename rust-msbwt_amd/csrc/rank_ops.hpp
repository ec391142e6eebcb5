// Device-side rank primitives shared by the query kernels (kernels.hip) and the pair-index
// builder (pair_index.hip).  Include from HIP translation units only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace msbwt {
namespace {

constexpr int kGroup = 8;       // lanes per query in the search phase

// x (op) value of x in the lane selected by the DPP control, all 64 lanes
template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_add(uint32_t x) {
    return x + uint32_t(__builtin_amdgcn_update_dpp(0, int(x), kCtrl, 0xF, 0xF, true));
}
// Sum over each aligned group of 8 lanes; every lane of the group gets the total.
__device__ __forceinline__ uint32_t group_sum(uint32_t x) {
    x = dpp_add<0xB1>(x);   // quad_perm [1,0,3,2]: lane ^ 1
    x = dpp_add<0x4E>(x);   // quad_perm [2,3,0,1]: lane ^ 2
    x = dpp_add<0x141>(x);  // row_half_mirror: lane -> 7 - lane (the other quad of the 8)
    return x;
}

struct Range {
    uint64_t l, h;
};

// New range for prepending symbol s (0..5) to [l, h): start_index[s] + rank(s, l / h).
// Called by all 8 lanes of a group with identical (s, l, h); `sub` = lane index in group.
__device__ __forceinline__ Range constrain(const uint4 *__restrict__ blocks, uint32_t s, uint64_t l,
                                           uint64_t h, uint32_t sub) {
    const uint4 cl = blocks[(l >> 8) * kGroup + sub];
    const uint4 ch = blocks[(h >> 8) * kGroup + sub];
    // a symbol matches s iff every plane bit equals the corresponding bit of s
    const uint32_t x0 = (s & 1u) ? 0u : ~0u, x1 = (s & 2u) ? 0u : ~0u, x2 = (s & 4u) ? 0u : ~0u;
    const int nl = min(max(int(uint32_t(l) & 255u) - int(sub * 32u), 0), 32);
    const int nh = min(max(int(uint32_t(h) & 255u) - int(sub * 32u), 0), 32);
    const uint32_t ml = nl >= 32 ? ~0u : ((1u << nl) - 1u);
    const uint32_t mh = nh >= 32 ? ~0u : ((1u << nh) - 1u);
    const uint32_t cnt_l = __popc((cl.x ^ x0) & (cl.y ^ x1) & (cl.z ^ x2) & ml);  // <= 32, sum <= 255
    const uint32_t cnt_h = __popc((ch.x ^ x0) & (ch.y ^ x1) & (ch.z ^ x2) & mh);
    // the block's 40-bit bound A[s]: low word in chunk s, high byte in chunk 6 (s<4) or 7
    const bool owns_lo = (sub == s);
    const bool owns_hi = (sub == 6u + (s >> 2));
    const uint32_t sh = (s & 3u) * 8u;
    const uint32_t lo_l = owns_lo ? cl.w : 0u, lo_h = owns_lo ? ch.w : 0u;
    const uint32_t hi_l = owns_hi ? ((cl.w >> sh) & 0xFFu) : 0u, hi_h = owns_hi ? ((ch.w >> sh) & 0xFFu) : 0u;
    // four byte-wide fields never carry into each other: counts sum to <= 255, one lane owns hi
    const uint32_t packed = group_sum(cnt_l | (hi_l << 8) | (cnt_h << 16) | (hi_h << 24));
    const uint32_t base_l = group_sum(lo_l), base_h = group_sum(lo_h);
    Range r;
    r.l = ((uint64_t((packed >> 8) & 0xFFu) << 32) | base_l) + (packed & 0xFFu);
    r.h = ((uint64_t(packed >> 24) << 32) | base_h) + ((packed >> 16) & 0xFFu);
    return r;
}


__device__ __forceinline__ uint32_t low_bits(int n) { return n >= 32 ? ~0u : ((1u << n) - 1u); }  // n in 0..32

// ---- run blocks (run_index.hpp): the memory-lean block format -----------------------------------
// start_index[s] + rank(s, pos) from pos's R512 block, by the 8 lanes of a group: lanes 0,1 hold the
// header, lanes 2..7 decode 16 one-byte runs each; a prefix sum over the lanes gives every lane the
// block offset of its first run, each lane clips its runs against the target offset and the group
// sums the matches.  An OVERFLOW block ranks from two plane-shaped lines of the side array instead.
constexpr uint32_t kRunOverflowFlag = 0x80000000u;

// Both positions p0 <= p1 must lie in the same block when `both` is set; then .h is p1's bound.
__device__ __forceinline__ Range rank_runs_group(const uint4 *__restrict__ blocks, const uint4 *__restrict__ overflow,
                                                 uint32_t s, uint64_t p0, uint64_t p1, bool both, uint32_t sub) {
    const uint32_t lane = threadIdx.x & 63u, group_base = lane & ~7u;
    const int r0 = int(uint32_t(p0) & 511u), r1 = int(uint32_t(p1) & 511u);
    const uint4 c = blocks[(p0 >> 9) * 8 + sub];
    // header: A[s] low word in word s (lane s>>2, component s&3), high byte in word 6/7 (lane 1)
    const uint32_t w0 = uint32_t(__shfl(int(s == 0u ? c.x : s == 1u ? c.y : s == 2u ? c.z : c.w), int(group_base)));
    const uint32_t w1 = uint32_t(__shfl(int(s == 4u ? c.x : c.y), int(group_base + 1)));
    const uint32_t lo = (s & 4u) ? w1 : w0;
    const uint32_t hi03 = uint32_t(__shfl(int(c.z), int(group_base + 1))), hi45 = uint32_t(__shfl(int(c.w), int(group_base + 1)));
    const uint32_t hi = (((s >> 2) ? hi45 : hi03) >> ((s & 3u) * 8u)) & 0xFFu;
    uint32_t cnt = 0;  // matches before r0 in the low half, before r1 in the high half (each <= 512)
    if (hi45 & kRunOverflowFlag) {  // group-uniform: the block's symbols live in two plane blocks of the side array, each
                                    // with its own header -- the plane rank of the line that holds each bound
        const uint4 *two = overflow + uint64_t(uint32_t(__shfl(int(c.x), int(group_base + 2)))) * 16;
        const uint64_t q0 = uint64_t(r0), q1 = both ? uint64_t(r1) : uint64_t(r0);  // positions inside the pair of plane blocks
        return constrain(two, s, q0, q1, sub);
    } else {
        const uint32_t word[4] = {c.x, c.y, c.z, c.w};
        uint32_t mine = 0;  // symbols my 16 runs cover (lanes 0,1: none)
        if (sub >= 2u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) mine = __builtin_amdgcn_sad_u8((word[j] >> 3) & 0x1F1F1F1Fu, 0u, mine);
        }
        uint32_t inc = mine;  // inclusive prefix over the 8 lanes
        for (int d = 1; d < 8; d <<= 1) {
            const uint32_t y = uint32_t(__shfl_up(int(inc), d, 8));
            if (int(sub) >= d) inc += y;
        }
        int cur = int(inc - mine);
        if (sub >= 2u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t run = (word[j] >> (8 * b)) & 0xFFu;
                    const int len = int(run >> 3);
                    if ((run & 7u) == s) {
                        cnt += uint32_t(min(max(r0 - cur, 0), len));
                        if (both) cnt += uint32_t(min(max(r1 - cur, 0), len)) << 16;
                    }
                    cur += len;
                }
            }
        }
    }
    cnt = group_sum(cnt);
    const uint64_t a = (uint64_t(hi) << 32) | lo;
    Range out;
    out.l = a + (cnt & 0xFFFFu);
    out.h = a + (cnt >> 16);
    return out;
}

// constrain() on whichever block format the index has (uniform per launch): 0 = plane blocks, 1 = run blocks
__device__ __forceinline__ Range constrain_any(uint32_t format, const uint4 *__restrict__ blocks, const uint4 *__restrict__ overflow,
                                               uint32_t s, uint64_t l, uint64_t h, uint32_t sub) {
    if (format == 0u) return constrain(blocks, s, l, h, sub);
    if ((l >> 9) == (h >> 9)) return rank_runs_group(blocks, overflow, s, l, h, true, sub);  // one line, one decode for both bounds
    Range r;
    r.l = rank_runs_group(blocks, overflow, s, l, l, false, sub).l;
    r.h = rank_runs_group(blocks, overflow, s, h, h, false, sub).l;
    return r;
}

// ---- pair blocks: two symbols per search step ------------------------------------------------
// With S the BWT and S2[i] = S[LF(i)], two consecutive steps (first a, then b) collapse into
//     p'' = K[a][b] + occ2(a, b, p),   K[a][b] = C[b] + occ(b, C[a]),
//     occ2(a, b, p) = #{ i < p : S[i] = a and S2[i] = b }
// (rows C[a] + j, j = 0.., are exactly the positions holding a, in order).  A pair block is
// 128 bytes for 128 positions, laid out as whole-block bit planes so that ONE lane can rank a
// position with a handful of 32-bit operations (the one-query-per-lane kernel, lanes.hip):
//     chunk 0 (words 0-3)    a0: bit i of word w = low bit of the 2-bit code of S[128 b + 32 w + i]
//     chunk 1 (words 4-7)    a1: its high bit             (A C G T -> 0 1 2 3)
//     chunk 2, 3             b0, b1: the same for S2
//     chunk 4 (words 16-19)  valid: S and S2 are both ACGT
//     chunk 5, 6             16 x u16: low halves of the 24-bit counts occ2(., ., block start),
//                            relative to the block's 2^24-position superblock, pair p = 4 a + b
//     chunk 7                16 x u8: their high bytes
// `super` holds K + occ2(superblock start) as u64 x 16 per superblock.
// Blocks come in two spacings.  stride 128: block b covers [128 b, 128 b + 128).  stride 96
// ("overlapping"): block b covers [96 b, 96 b + 128) -- every block also holds the first 32
// positions of the next one, so a range up to 32 wide that starts in a block's own 96 positions is
// ranked from ONE line (on real 30x data ranges stay ~25 wide to the last step: with stride 128 every
// fifth step needs a second line).  1.33 bytes per symbol instead of 1.  Header counts are those at
// the block's first position either way; a superblock is 2^17 blocks.
constexpr int kPairSuperBlocks = 17;  // log2(blocks per superblock)

__device__ __forceinline__ uint64_t pair_block_of(uint64_t pos, bool stride96) { return stride96 ? (pos >> 5) / 3u : pos >> 7; }
__device__ __forceinline__ uint64_t pair_block_start(uint64_t blk, bool stride96) { return stride96 ? blk * 96u : blk << 7; }
constexpr int kPairValidChunk = 4, kPairLoChunk = 5, kPairHiChunk = 7;

// branch-free: codes 1,2,3,5 are bits 1,2,3,5 of 0x2E; A C G T -> 0 1 2 3 is s - 1 - (s >> 2)
__device__ __forceinline__ uint32_t acgt_code(uint32_t s) { return s - 1u - (s >> 2); }  // s in {1,2,3,5}
__device__ __forceinline__ uint32_t acgt_bit(uint32_t s) { return (0x2Eu >> (s & 7u)) & 1u; }
__device__ __forceinline__ bool is_acgt(uint32_t s) { return acgt_bit(s) != 0u; }

// ---- 8-lane groups, one bound per quad (the search loop of kernels.hip) ---------------------------
// Lanes 0-3 of the group work on bound l, lanes 4-7 on bound h; lane q of a quad holds chunks q
// and q+4 of its bound's block (one load instruction fetches the first halves of both lines,
// the other the second halves).  Each lane computes masks, header field and base for ONE bound
// only; a two-step quad sum and one cross-quad exchange finish the step.  Fewer instructions
// per query step than handling both bounds in every lane.
__device__ __forceinline__ uint32_t quad_sum(uint32_t x) {
    x = dpp_add<0xB1>(x);  // lane ^ 1
    x = dpp_add<0x4E>(x);  // lane ^ 2
    return x;
}

__device__ __forceinline__ uint64_t other_quad(uint64_t x) {  // value held by the group's other quad
    const uint32_t lo = uint32_t(__builtin_amdgcn_update_dpp(0, int(uint32_t(x)), 0x141, 0xF, 0xF, true));
    const uint32_t hi = uint32_t(__builtin_amdgcn_update_dpp(0, int(uint32_t(x >> 32)), 0x141, 0xF, 0xF, true));
    return (uint64_t(hi) << 32) | lo;
}

__device__ __forceinline__ Range constrain_split(const uint4 *__restrict__ blocks, uint32_t s, uint64_t l, uint64_t h,
                                                 uint32_t sub) {
    const bool upper = (sub & 4u) != 0;
    const uint32_t q = sub & 3u;
    const uint64_t pos = upper ? h : l;
    const uint4 *b = blocks + (pos >> 8) * 8 + q;
    const uint4 c0 = b[0], c1 = b[4];
    const uint32_t x0 = (s & 1u) ? 0u : ~0u, x1 = (s & 2u) ? 0u : ~0u, x2 = (s & 4u) ? 0u : ~0u;
    const int r = int(uint32_t(pos) & 255u) - int(q * 32u);  // chunk q covers [32q, 32q+32), chunk q+4 is 128 further
    const uint32_t cnt = __popc((c0.x ^ x0) & (c0.y ^ x1) & (c0.z ^ x2) & low_bits(min(max(r, 0), 32))) +
                         __popc((c1.x ^ x0) & (c1.y ^ x1) & (c1.z ^ x2) & low_bits(min(max(r - 128, 0), 32)));
    // header: low word of A[s] in chunk s = lane s&3, its chunk s>>2; high byte in chunk 6/7 = lane 2/3, second chunk
    const uint32_t lo = ((s >> 2) ? c1.w : c0.w) & ((q == (s & 3u)) ? ~0u : 0u);
    const uint32_t hi = (c1.w >> ((s & 3u) * 8u)) & ((q == 2u + (s >> 2)) ? 0xFFu : 0u);
    const uint32_t packed = quad_sum(cnt | (hi << 8));  // count <= 255, one lane owns hi
    const uint32_t base = quad_sum(lo);
    const uint64_t mine = ((uint64_t(packed >> 8) << 32) | base) + (packed & 0xFFu);
    const uint64_t theirs = other_quad(mine);
    Range out;
    out.l = upper ? theirs : mine;
    out.h = upper ? mine : theirs;
    return out;
}

}  // namespace
}  // namespace msbwt
