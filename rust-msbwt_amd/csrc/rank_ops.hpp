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


// ---- pair blocks: two symbols per search step ------------------------------------------------
// With S the BWT and S2[i] = S[LF(i)], two consecutive steps (first a, then b) collapse into
//     p'' = K[a][b] + occ2(a, b, p),   K[a][b] = C[b] + occ(b, C[a]),
//     occ2(a, b, p) = #{ i < p : S[i] = a and S2[i] = b }
// (rows C[a] + j, j = 0.., are exactly the positions holding a, in order).  A pair block is
// 128 bytes for 128 positions: chunk j (16 B) = { a-planes, b-planes, valid | header bits } of
// positions 16j..16j+15, where a/b are 2-bit ACGT codes (A C G T -> 0..3) and `valid` marks
// positions whose S and S2 are both ACGT.  The 8 x 48 header bits hold the 16 pair counts
// occ2(.,., block start) relative to the block's 2^24-position superblock (24 bits each,
// pairs 2j and 2j+1 in chunk j); `super` holds K + occ2(superblock start) as u64 x 16.
constexpr int kPairShift = 7;         // 128 positions per pair block
constexpr int kPairSuperShift = 24;   // 2^24 positions per superblock

// branch-free: codes 1,2,3,5 are bits 1,2,3,5 of 0x2E; A C G T -> 0 1 2 3 is s - 1 - (s >> 2)
__device__ __forceinline__ uint32_t acgt_code(uint32_t s) { return s - 1u - (s >> 2); }  // s in {1,2,3,5}
__device__ __forceinline__ uint32_t acgt_bit(uint32_t s) { return (0x2Eu >> (s & 7u)) & 1u; }
__device__ __forceinline__ bool is_acgt(uint32_t s) { return acgt_bit(s) != 0u; }

// matches of pair (a2, b2) among the first n (0..16) positions of one pair chunk
__device__ __forceinline__ uint32_t pair_chunk_count(const uint4 c, uint32_t a2, uint32_t b2, int n) {
    const uint32_t pa = ((a2 & 1u) ? 0x0000FFFFu : 0u) | ((a2 & 2u) ? 0xFFFF0000u : 0u);
    const uint32_t pb = ((b2 & 1u) ? 0x0000FFFFu : 0u) | ((b2 & 2u) ? 0xFFFF0000u : 0u);
    const uint32_t ea = ~(c.x ^ pa), eb = ~(c.y ^ pb);  // 1 where the plane bit equals the wanted bit
    const uint32_t m = ea & (ea >> 16) & eb & (eb >> 16) & c.z & ((1u << n) - 1u) & 0xFFFFu;
    return uint32_t(__popc(m));
}

// 24-bit header field of pair p held by this chunk (p>>1 must be the chunk index); branch-free
__device__ __forceinline__ uint32_t pair_chunk_field(const uint4 c, uint32_t p) {
    const uint32_t even = (c.z >> 16) | ((c.w & 0xFFu) << 16), odd = c.w >> 8;
    return (p & 1u) ? odd : even;
}

// ---- the same two steps for 4-lane groups: a lane owns two adjacent chunks (32 bytes) of the
// block, a wave carries 16 queries, and the group sum is two quad_perm steps --------------------
__device__ __forceinline__ uint32_t quad_sum(uint32_t x) {
    x = dpp_add<0xB1>(x);  // lane ^ 1
    x = dpp_add<0x4E>(x);  // lane ^ 2
    return x;
}

__device__ __forceinline__ uint32_t low_bits(int n) { return n >= 32 ? ~0u : ((1u << n) - 1u); }  // n in 0..32

__device__ __forceinline__ Range constrain_quad(const uint4 *__restrict__ blocks, uint32_t s, uint64_t l, uint64_t h,
                                                uint32_t sub) {
    const uint4 *bl = blocks + (l >> 8) * 8 + 2u * sub;
    const uint4 *bh = blocks + (h >> 8) * 8 + 2u * sub;
    const uint4 l0 = bl[0], l1 = bl[1], h0 = bh[0], h1 = bh[1];
    const uint32_t x0 = (s & 1u) ? 0u : ~0u, x1 = (s & 2u) ? 0u : ~0u, x2 = (s & 4u) ? 0u : ~0u;
    const int rl = int(uint32_t(l) & 255u) - int(sub * 64u), rh = int(uint32_t(h) & 255u) - int(sub * 64u);
    const uint32_t cnt_l = __popc((l0.x ^ x0) & (l0.y ^ x1) & (l0.z ^ x2) & low_bits(min(max(rl, 0), 32))) +
                           __popc((l1.x ^ x0) & (l1.y ^ x1) & (l1.z ^ x2) & low_bits(min(max(rl - 32, 0), 32)));
    const uint32_t cnt_h = __popc((h0.x ^ x0) & (h0.y ^ x1) & (h0.z ^ x2) & low_bits(min(max(rh, 0), 32))) +
                           __popc((h1.x ^ x0) & (h1.y ^ x1) & (h1.z ^ x2) & low_bits(min(max(rh - 32, 0), 32)));
    // header: low word of A[s] in chunk s = lane s>>1, its chunk s&1; high byte in chunk 6/7 = lane 3
    const uint32_t owns_lo = (sub == (s >> 1)) ? ~0u : 0u, owns_hi = (sub == 3u) ? 0xFFu : 0u;
    const uint32_t sh = (s & 3u) * 8u;
    const uint32_t lo_l = ((s & 1u) ? l1.w : l0.w) & owns_lo, lo_h = ((s & 1u) ? h1.w : h0.w) & owns_lo;
    const uint32_t hi_l = (((s >> 2) ? l1.w : l0.w) >> sh) & owns_hi, hi_h = (((s >> 2) ? h1.w : h0.w) >> sh) & owns_hi;
    const uint32_t packed = quad_sum(cnt_l | (hi_l << 8) | (cnt_h << 16) | (hi_h << 24));  // counts sum to <= 255
    const uint32_t base_l = quad_sum(lo_l), base_h = quad_sum(lo_h);
    Range r;
    r.l = ((uint64_t((packed >> 8) & 0xFFu) << 32) | base_l) + (packed & 0xFFu);
    r.h = ((uint64_t(packed >> 24) << 32) | base_h) + ((packed >> 16) & 0xFFu);
    return r;
}

__device__ __forceinline__ Range constrain2_quad(const uint4 *__restrict__ pair_blocks, const uint64_t *__restrict__ super,
                                                 uint32_t a2, uint32_t b2, uint64_t l, uint64_t h, uint32_t sub) {
    const uint4 *bl = pair_blocks + (l >> kPairShift) * 8 + 2u * sub;
    const uint4 *bh = pair_blocks + (h >> kPairShift) * 8 + 2u * sub;
    const uint4 l0 = bl[0], l1 = bl[1], h0 = bh[0], h1 = bh[1];
    const uint32_t p = a2 * 4u + b2;
    const uint64_t kl = super[(l >> kPairSuperShift) * 16u + p];
    const uint64_t kh = super[(h >> kPairSuperShift) * 16u + p];
    const int rl = int(uint32_t(l) & 127u) - int(sub * 32u), rh = int(uint32_t(h) & 127u) - int(sub * 32u);
    const uint32_t cnt_l = pair_chunk_count(l0, a2, b2, min(max(rl, 0), 16)) + pair_chunk_count(l1, a2, b2, min(max(rl - 16, 0), 16));
    const uint32_t cnt_h = pair_chunk_count(h0, a2, b2, min(max(rh, 0), 16)) + pair_chunk_count(h1, a2, b2, min(max(rh - 16, 0), 16));
    // pair p lives in chunk p>>1 = lane p>>2, its chunk (p>>1)&1
    const uint32_t owner = (sub == (p >> 2)) ? ~0u : 0u;
    const bool second = ((p >> 1) & 1u) != 0;
    const uint32_t tl = quad_sum(cnt_l | ((pair_chunk_field(second ? l1 : l0, p) << 8) & owner));
    const uint32_t th = quad_sum(cnt_h | ((pair_chunk_field(second ? h1 : h0, p) << 8) & owner));
    Range r;
    r.l = kl + (tl >> 8) + (tl & 0xFFu);
    r.h = kh + (th >> 8) + (th & 0xFFu);
    return r;
}

// ---- 8-lane groups, one bound per quad ----------------------------------------------------------
// Lanes 0-3 of the group work on bound l, lanes 4-7 on bound h; lane q of a quad holds chunks q
// and q+4 of its bound's block (one load instruction fetches the first halves of both lines,
// the other the second halves).  Each lane computes masks, header field and base for ONE bound
// only; a two-step quad sum and one cross-quad exchange finish the step.  Fewer instructions
// per query step than handling both bounds in every lane (the kernels are issue-bound).
__device__ __forceinline__ uint64_t other_quad(uint64_t x) {  // value held by the group's other quad
    const uint32_t lo = uint32_t(__builtin_amdgcn_update_dpp(0, int(uint32_t(x)), 0x141, 0xF, 0xF, true));
    const uint32_t hi = uint32_t(__builtin_amdgcn_update_dpp(0, int(uint32_t(x >> 32)), 0x141, 0xF, 0xF, true));
    return (uint64_t(hi) << 32) | lo;
}

// A step is written as its loads (issue_*) and its arithmetic (finish_*); constrain_split /
// constrain2_split chain the two halves.
struct StepLoads {
    uint4 c0, c1;
    uint64_t k;  // pair steps: superblock base; unused for single steps
};

__device__ __forceinline__ StepLoads issue_single(const uint4 *__restrict__ blocks, uint64_t l, uint64_t h, uint32_t sub) {
    const uint64_t pos = (sub & 4u) ? h : l;
    const uint4 *b = blocks + (pos >> 8) * 8 + (sub & 3u);
    StepLoads L;
    L.c0 = b[0];
    L.c1 = b[4];
    L.k = 0;
    return L;
}

__device__ __forceinline__ Range finish_single(const StepLoads &L, uint32_t s, uint64_t l, uint64_t h, uint32_t sub) {
    const bool upper = (sub & 4u) != 0;
    const uint32_t q = sub & 3u;
    const uint64_t pos = upper ? h : l;
    const uint4 c0 = L.c0, c1 = L.c1;
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

__device__ __forceinline__ StepLoads issue_pair(const uint4 *__restrict__ pair_blocks, const uint64_t *__restrict__ super,
                                                uint32_t a2, uint32_t b2, uint64_t l, uint64_t h, uint32_t sub) {
    const uint64_t pos = (sub & 4u) ? h : l;
    const uint4 *b = pair_blocks + (pos >> kPairShift) * 8 + (sub & 3u);
    StepLoads L;
    L.c0 = b[0];
    L.c1 = b[4];
    L.k = super[__builtin_amdgcn_alignbit(uint32_t(pos >> 32), uint32_t(pos), kPairSuperShift) * 16u + (a2 * 4u + b2)];
    return L;
}

__device__ __forceinline__ Range finish_pair(const StepLoads &L, uint32_t a2, uint32_t b2, uint64_t l, uint64_t h, uint32_t sub) {
    const bool upper = (sub & 4u) != 0;
    const uint32_t q = sub & 3u;
    const uint64_t pos = upper ? h : l;
    const uint32_t p = a2 * 4u + b2;
    const int r = int(uint32_t(pos) & 127u) - int(q * 16u);  // chunk q covers [16q, 16q+16), chunk q+4 is 64 further
    const uint32_t cnt = pair_chunk_count(L.c0, a2, b2, min(max(r, 0), 16)) + pair_chunk_count(L.c1, a2, b2, min(max(r - 64, 0), 16));
    // pair p lives in chunk p>>1 = lane (p>>1)&3, its chunk p>>3
    const uint32_t owner = (q == ((p >> 1) & 3u)) ? ~0u : 0u;
    const uint32_t t = quad_sum(cnt | ((pair_chunk_field((p >> 3) ? L.c1 : L.c0, p) << 8) & owner));
    const uint64_t mine = L.k + ((t >> 8) + (t & 0xFFu));
    const uint64_t theirs = other_quad(mine);
    Range out;
    out.l = upper ? theirs : mine;
    out.h = upper ? mine : theirs;
    return out;
}

__device__ __forceinline__ Range constrain_split(const uint4 *__restrict__ blocks, uint32_t s, uint64_t l, uint64_t h,
                                                 uint32_t sub) {
    return finish_single(issue_single(blocks, l, h, sub), s, l, h, sub);
}

__device__ __forceinline__ Range constrain2_split(const uint4 *__restrict__ pair_blocks, const uint64_t *__restrict__ super,
                                                  uint32_t a2, uint32_t b2, uint64_t l, uint64_t h, uint32_t sub) {
    return finish_pair(issue_pair(pair_blocks, super, a2, b2, l, h, sub), a2, b2, l, h, sub);
}

// Uniform access to the two group shapes
template <int kLanes>
struct GroupOps;
template <>
struct GroupOps<8> {
    static __device__ __forceinline__ Range step(const uint4 *b, uint32_t s, uint64_t l, uint64_t h, uint32_t sub) {
        return constrain_split(b, s, l, h, sub);
    }
    static __device__ __forceinline__ Range step2(const uint4 *pb, const uint64_t *sup, uint32_t a2, uint32_t b2, uint64_t l,
                                                  uint64_t h, uint32_t sub) {
        return constrain2_split(pb, sup, a2, b2, l, h, sub);
    }
};
template <>
struct GroupOps<4> {
    static __device__ __forceinline__ Range step(const uint4 *b, uint32_t s, uint64_t l, uint64_t h, uint32_t sub) {
        return constrain_quad(b, s, l, h, sub);
    }
    static __device__ __forceinline__ Range step2(const uint4 *pb, const uint64_t *sup, uint32_t a2, uint32_t b2, uint64_t l,
                                                  uint64_t h, uint32_t sub) {
        return constrain2_quad(pb, sup, a2, b2, l, h, sub);
    }
};

}  // namespace
}  // namespace msbwt
