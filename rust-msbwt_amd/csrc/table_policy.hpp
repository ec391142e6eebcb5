// Automatic suffix-table depths (host only; no HIP here, so that the CPU suite can pin the policy through
// msbwt_auto_table_depths).  Every search step is at least one random 128-byte line and the memory
// system serves a fixed number of such lines per second whether they come from HBM or the Infinity Cache
// (tools/ubench_granule.hip: 4.86 x 10^10 lines/s), so each table level removes one or two line fetches
// per surviving query for the price of memory only -- which is what a 288 GB part has.
#pragma once
#include <algorithm>
#include <cstdint>

#include "kernels.hpp"

namespace msbwt {

struct TableChoice {
    int flat;    // levels of the flat table that is built first (16 bytes per entry); 0 = no table
    int packed;  // levels of the packed table it is turned into (flat + 2), 0 = the flat table stays
};

// Flat table: the deepest that the data warrant (4^depth <= T) within max(1 GiB, 2 x the block array),
// at most 15 levels (16 GiB).
// `allowance`: what a memory budget (msbwt_rle_set_memory_budget) leaves for the table; ~0 = no budget.
constexpr uint64_t kNoBudget = ~uint64_t(0);
inline int auto_flat_table_depth(uint64_t total, uint64_t block_bytes, uint64_t allowance = kNoBudget) {
    const uint64_t budget = std::min(allowance, std::max<uint64_t>(uint64_t(1) << 30, 2 * block_bytes));
    int d = 0;
    while (d < 15 && (uint64_t(4) << (2 * d)) <= total && (uint64_t(64) << (2 * d)) <= budget) ++d;
    return d;
}

// Beside a pair index the flat table is only the (temporary) parent of a packed one, two levels deeper
// (kernels.hpp, launch_pack_table): aim for the deepest packed table -- at most 17 levels, 73 GB --
// that the data warrant and HBM allows (its lines take at most half of what is free, and parent and packed table
// fit side by side while packing).  "Warrant": at most kTableEntriesPerSymbol entries per BWT symbol.  Most entries
// of such a table are empty, but a PRESENT k-mer's entry never is, and every two levels spare it a pair step: round 2
// stopped at 16 entries per symbol (C3: depth 15); with 256 a 2 x 10^8-symbol index gets the full depth 17 (73 GB that
// a 288 GB part has to spare): C3 fused 8.9 -> 9.8 x 10^9 windows/s.  Toy indexes stay small (T = 10: depth 5).
constexpr uint64_t kTableEntriesPerSymbol = 256;
inline TableChoice choose_table_depths(uint64_t total, uint64_t block_bytes, uint64_t free_bytes, bool pair_index, bool packing_allowed,
                                       uint64_t allowance = kNoBudget) {
    TableChoice c{auto_flat_table_depth(total, block_bytes, allowance), 0};
    if (!pair_index || !packing_allowed) return c;
    // Under a memory budget the packed lines must fit what the budget leaves -- all of it: the caller has said how much the
    // index may hold, so the "half of what is free" courtesy does not apply to the allowance (it still does to the HBM that is
    // actually free, and the flat parent, which is freed after packing, only has to fit there).
    const auto fits = [&](int p) {
        const uint64_t flat_b = uint64_t(16) << (2 * (p - 2));
        return (uint64_t(1) << (2 * p)) <= kTableEntriesPerSymbol * total && 2 * packed_table_bytes(p) <= free_bytes && flat_b + packed_table_bytes(p) <= free_bytes &&
               packed_table_bytes(p) <= allowance;
    };
    if (allowance != kNoBudget) {  // the deepest packed table the allowance holds, whatever the flat table's own depth would be
        c.flat = 0;
        for (int p = 17; p >= 3; --p)
            if (fits(p)) {
                c.flat = p - 2;
                c.packed = p;
                return c;
            }
        c.flat = auto_flat_table_depth(total, block_bytes, allowance);
        return c;
    }
    for (int p = 17; p - 2 > c.flat; --p)
        if (fits(p)) {
            c.flat = p - 2;
            break;
        }
    if (c.flat > 0 && fits(c.flat + 2)) c.packed = c.flat + 2;
    return c;
}

// ---- spacing of the pair blocks (rank_ops.hpp): 128 = disjoint, 96 = overlapping (1.33 bytes per symbol) --------
// Overlapping blocks rank a range up to 32 wide from ONE line.  Whether that pays is a property of the DATA: on
// a real 30x read set the range of a present k-mer stays about as wide as the coverage down to the last step
// (every fifth pair step would need a second line from disjoint blocks), on a stream of independent symbols
// it collapses to width 1 within a few steps and the overlap buys nothing.  The loader measures which it is:
// launch_probe_widths (kernels.hpp) walks a few thousand pseudo-random rows backwards and reports how often the
// kProbeSteps-mer read off each of them occurs; `typical_width` is the median.
constexpr uint32_t kProbeSamples = 4096, kProbeSteps = 24;
constexpr double kWideRangeThreshold = 8.0;  // from here on a second line per pair step is common enough to pay 0.33 bytes per symbol

// bytes96 / bytes128: pair blocks + superblock table (+ build scratch) at either spacing; expected_table_bytes: the
// suffix table the loader is going to build beside DISJOINT blocks (its policy budgets against those; the
// overlap is paid from the reserve); free_bytes: HBM free once the plane blocks are in place; an eighth of the
// device's HBM stays free for the caller's batches.  typical_width < 0 = unknown.
inline int choose_pair_stride(uint64_t bytes96, uint64_t expected_table_bytes, uint64_t free_bytes, uint64_t hbm_total_bytes, double typical_width) {
    if (bytes96 <= free_bytes / 4) return 96;  // cheap (C3 / C4-sized indexes): taken whatever the data look like
    if (typical_width < kWideRangeThreshold) return 128;
    return bytes96 + expected_table_bytes + hbm_total_bytes / 8 <= free_bytes ? 96 : 128;
}

// peak HBM the table takes while it is built with `free_bytes` free (flat parent + packed lines side by side)
inline uint64_t expected_table_bytes(uint64_t total, uint64_t block_bytes, uint64_t free_bytes, bool pair_index, bool packing_allowed,
                                     uint64_t allowance = kNoBudget) {
    const TableChoice c = choose_table_depths(total, block_bytes, free_bytes, pair_index, packing_allowed, allowance);
    const uint64_t flat_b = c.flat ? uint64_t(16) << (2 * c.flat) : 0;
    return c.packed ? flat_b + packed_table_bytes(c.packed) : flat_b;
}

// ---- run blocks: does the DEVICE builder fit?  (capi.cpp, build_run_index; pinned by a CPU test through msbwt_run_build_fits_device) ----
// Asked once the RLE bytes are in HBM and the totals are known: the plane blocks (128 bytes per 256 symbols) and the run blocks (128
// bytes per 512 symbols, + up to an eighth in overflow blocks) must fit the free HBM side by side, with a 32nd of it as slack.  About
// 0.8 byte per symbol at its peak against 0.3 for the finished index -- an index that loaded in this format only BECAUSE it is lean is
// built on the host, as until round 3.
inline bool run_build_fits_device(uint64_t total_symbols, uint64_t free_bytes) {
    const uint64_t planes = (total_symbols / 256 + 1) * 128, runs = (total_symbols / 512 + 1) * 128;
    const uint64_t peak = planes + runs + runs / 8;
    return peak <= free_bytes - free_bytes / 32;
}

// ---- a memory budget for the whole index (msbwt_rle_set_memory_budget): the analogue of the reference's only space / time
// knob, `bin_power` (rle_bwt.rs:309-322) -------------------------------------------------------------------------------------
// The plane blocks (0.5 byte per symbol) are always built.  What the budget leaves goes, in this order, to the pair blocks
// (two symbols per step: the largest single gain; disjoint blocks, 1 byte per symbol, whenever they fit with a little room
// for a table), then to the deepest packed suffix table that fits (each two
// levels spare a query a step), then -- if the probe says ranges stay wide -- to the overlapping pair blocks (+0.33 byte per
// symbol).  A pure function of sizes, pinned by a CPU test through msbwt_auto_index_plan.
struct IndexPlan {
    bool pair;       // build pair blocks
    int stride;      // 96 or 128 (meaningful when pair)
    int flat, packed;
    uint64_t bytes;  // HBM the finished index holds under this plan (blocks + pair blocks + table; filter and side array not counted)
};

// pair128_bytes / pair96_bytes: pair blocks + superblock table at either spacing (pair_index_sizes)
inline IndexPlan plan_index(uint64_t total, uint64_t free_bytes, uint64_t hbm_total_bytes, double typical_width, uint64_t budget,
                            uint64_t pair128_bytes, uint64_t pair96_bytes) {
    const uint64_t nblocks = total / 256 + 1, plane_b = nblocks * 128;
    IndexPlan p{false, 128, 0, 0, plane_b};
    if (total == 0) return p;
    const uint64_t left0 = budget == 0 ? kNoBudget : (budget > plane_b ? budget - plane_b : 0);
    // two symbols per step halve the lines of every query, which no table depth that the same bytes would buy does: the pair
    // blocks come first whenever they fit with a sixteenth of their size to spare for a table
    p.pair = pair128_bytes <= free_bytes / 2 && (left0 == kNoBudget || pair128_bytes + pair128_bytes / 16 <= left0);
    uint64_t left = left0, free_left = free_bytes;
    if (p.pair) {
        if (left != kNoBudget) left -= pair128_bytes;
        free_left -= pair128_bytes;
    }
    const TableChoice c = choose_table_depths(total, plane_b, free_left, p.pair, true, left);
    p.flat = c.flat;
    p.packed = c.packed;
    const uint64_t table_b = c.packed ? packed_table_bytes(c.packed) : (c.flat ? uint64_t(16) << (2 * c.flat) : 0);
    if (left != kNoBudget) left = left > table_b ? left - table_b : 0;
    if (p.pair) {
        const uint64_t extra = pair96_bytes > pair128_bytes ? pair96_bytes - pair128_bytes : 0;
        const bool affordable = left == kNoBudget || extra <= left;
        const int want = choose_pair_stride(pair96_bytes, table_b, free_bytes, hbm_total_bytes, typical_width);
        p.stride = (want == 96 && affordable) ? 96 : 128;
    }
    p.bytes = plane_b + (p.pair ? (p.stride == 96 ? pair96_bytes : pair128_bytes) : 0) + table_b;
    return p;
}

}  // namespace msbwt
