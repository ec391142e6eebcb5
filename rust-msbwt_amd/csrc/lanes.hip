// The one-query-per-lane search kernel's launch (kernel and helpers: lanes_kernel.hpp): this translation unit holds the direct-table
// instantiations (kSparse = 0) and those of the 24-bit-tag sparse table (1); the other layouts live in lanes_wide.hip, lanes_xwide.hip
// and lanes_tier.hip.
#include "lanes_kernel.hpp"

namespace msbwt {
namespace {

template <bool kReads, bool kPacked>
hipError_t launch_shape(bool pair, bool longk, hipStream_t stream, const IndexView &ix, const QuerySource &src, uint32_t *flags) {
    // the sparse table serves every query that is at least as long as its entries (shorter ones: the direct table)
    // -- one kernel per table LAYOUT, so that each carries a single scan (round 6: with the three complete layouts behind launch-uniform
    // branches in one kernel the lookup-heavy lines lost 8-13 % against round 5's single-layout kernel on the same box)
    if (const SparseView *sp = sparse_for(ix, src.k)) {
        if (sp->tier) return launch_lanes_sparse_tier(kReads, kPacked, pair, longk, stream, ix, src, flags);
        if (sparse_xwide(sp->depth)) return launch_lanes_sparse_xwide(kReads, kPacked, pair, longk, stream, ix, src, flags);
        if (sparse_wide(sp->depth)) return launch_lanes_sparse_wide(kReads, kPacked, pair, longk, stream, ix, src, flags);
        return launch_sparse_shape<kReads, kPacked, 1>(pair, longk, stream, ix, src, flags);
    }
    if (!pair) return longk ? launch_variant<kReads, false, 6, false, kPacked, 0>(stream, ix, src, flags) : launch_variant<kReads, false, 3, false, kPacked, 0>(stream, ix, src, flags);
    if (ix.pair_stride96) return longk ? launch_variant<kReads, true, 6, true, kPacked, 0>(stream, ix, src, flags) : launch_variant<kReads, true, 3, true, kPacked, 0>(stream, ix, src, flags);
    return longk ? launch_variant<kReads, true, 6, false, kPacked, 0>(stream, ix, src, flags) : launch_variant<kReads, true, 3, false, kPacked, 0>(stream, ix, src, flags);
}

}  // namespace

hipError_t launch_lanes(const IndexView &ix, const QuerySource &src, bool reads, bool pair, uint32_t *flags,
                        hipStream_t stream) {
    if (src.k < 1 || src.k > uint32_t(kMaxTiledK)) return hipErrorInvalidValue;
    if (!reads && (reinterpret_cast<uintptr_t>(src.data) & ((src.packed && !src.place_inline) ? 7u : 15u)) != 0) return hipErrorInvalidValue;
    if (src.n == 0) return hipSuccess;
    pair = pair && ix.pair_blocks != nullptr && ix.block_format == kBlocksPlanes;
    if (ix.block_format != kBlocksPlanes && src.packed) return hipErrorInvalidValue;  // (packed queries on run blocks are unpacked by the caller)
    const bool longk = src.k > uint32_t(kMaxShortK);
    if (reads) return src.packed ? hipErrorInvalidValue : launch_shape<true, false>(pair, longk, stream, ix, src, flags);
    if ((src.out_index != nullptr || src.place_inline != 0u) && !src.packed) return hipErrorInvalidValue;  // (counts are placed for packed queries only)
    return src.packed ? launch_shape<false, true>(pair, longk, stream, ix, src, flags) : launch_shape<false, false>(pair, longk, stream, ix, src, flags);
}

}  // namespace msbwt
