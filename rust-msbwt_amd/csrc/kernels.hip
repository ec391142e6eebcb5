// gfx950 (MI355X, CDNA4) kernels for batched FM-index backward search over plane blocks.
// Integer / bit work bound by random 128-byte fetches: no MFMA anywhere.
//
// Work decomposition: an 8-lane group owns one query (64-lane wave = 8 queries in flight).
// For a rank the group's 8 lanes load the 8 x 16-byte chunks of one 128-byte block with a
// single coalesced global_load_dwordx4, each lane popcounts its 32 symbols, and the group
// sums with three DPP steps (quad_perm xor 1, xor 2, row_half_mirror) -- no LDS, no
// barriers.  Layout of a block: plane_index.hpp.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace msbwt {
namespace {

constexpr int kGroup = 8;  // lanes per query

// x + (value of x in the lane selected by the DPP control), all 64 lanes
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

// ---- count_kmers, version 1: one query per group at a time, symbols read as needed ----
__global__ __launch_bounds__(256) void k_count_kmers(const uint4 *__restrict__ blocks, uint64_t total,
                                                     const uint8_t *__restrict__ kmers, uint32_t k, uint64_t n,
                                                     uint64_t *__restrict__ counts, uint32_t *__restrict__ flags) {
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
            for (uint32_t i = k; i-- > 0 && r.l != r.h;) r = constrain(blocks, kmer[i], r.l, r.h, sub);
            result = r.h - r.l;
        }
        if (sub == 0) counts[q] = result;
    }
}

__global__ __launch_bounds__(256) void k_constrain_ranges(const uint4 *__restrict__ blocks, uint64_t total,
                                                          const uint8_t *__restrict__ syms,
                                                          const uint64_t *__restrict__ l, const uint64_t *__restrict__ h,
                                                          uint64_t n, uint64_t *__restrict__ out_l,
                                                          uint64_t *__restrict__ out_h, uint32_t *__restrict__ flags) {
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
            r = constrain(blocks, s, li, hi, sub);
        }
        if (sub == 0) {
            out_l[i] = r.l;
            out_h[i] = r.h;
        }
    }
}

inline uint32_t grid_for(uint64_t n_groups_wanted) {
    // 256 CUs x 8 blocks of 256 threads fill the chip; smaller batches get just enough blocks
    const uint64_t blocks = (n_groups_wanted * kGroup + 255) / 256;
    return uint32_t(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));
}

}  // namespace

hipError_t launch_count_kmers(const IndexView &ix, const uint8_t *kmers, uint32_t k, uint64_t n,
                              uint64_t *counts, uint32_t *flags, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_count_kmers, dim3(grid_for(n)), dim3(256), 0, stream,
                       static_cast<const uint4 *>(ix.blocks), ix.total, kmers, k, n, counts, flags);
    return hipGetLastError();
}

hipError_t launch_constrain_ranges(const IndexView &ix, const uint8_t *syms, const uint64_t *l,
                                   const uint64_t *h, uint64_t n, uint64_t *out_l, uint64_t *out_h,
                                   uint32_t *flags, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_constrain_ranges, dim3(grid_for(n)), dim3(256), 0, stream,
                       static_cast<const uint4 *>(ix.blocks), ix.total, syms, l, h, n, out_l, out_h, flags);
    return hipGetLastError();
}

hipError_t launch_build_table(const IndexView &, int, void *, hipStream_t) { return hipErrorNotSupported; }

}  // namespace msbwt
