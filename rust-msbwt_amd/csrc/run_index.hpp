// Run blocks: the memory-lean alternative to plane blocks (msbwt_rle_set_block_format) -- the
// layout BASELINE.json's north_star sketches after src/run_block_av_flat.rs:43-56,97-125:
// fixed-width runs + per-block occurrence counts, one 128-byte line per rank.
//
// R512 block = 128 bytes = BWT positions [512 b, 512 b + 512):
//   words 0..7   the same header as a plane block: for s = 0..5 the 40-bit value
//                A[s] = start_index[s] + occ(s, 512 b) (low words in words 0..5, high bytes in words
//                6 and 7); bit 31 of word 7 = OVERFLOW
//   bytes 32..127  96 one-byte runs, byte = sym | len << 3 with len 1..31 (0 = unused slot); a BWT
//                run is cut at block borders and into pieces of at most 31
//   overflow     a block that needs more than 96 pieces keeps, instead of runs, the index (word 8) of
//                TWO PLANE BLOCKS in a side array (plane_index.hpp layout, each with its own header: the
//                counts at 512 b and at 512 b + 256), so that any rank inside it is ONE more line: a
//                second, dependent fetch for the rare low-run-length block.  (Until round 3 the two lines
//                shared the run block's header and a rank could need both.)
// About 0.30 bytes per symbol on 30x short-read BWTs (plane blocks: 0.50), 1.4x the work per rank
// (DESIGN.md section 2).  No pair index in this format; since round 4 the lane-per-query kernel
// (lanes.hip) reads it for 6 <= k <= 32.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "rle_codec.hpp"

namespace msbwt {

constexpr int kRunShift = 9;              // 512 positions per run block
constexpr int kRunsPerBlock = 96;
constexpr uint32_t kRunOverflowBit = 0x80000000u;  // in header word 7

inline uint64_t run_block_count(uint64_t total) { return (total >> kRunShift) + 1; }

struct RunIndex {
    std::vector<uint32_t> blocks;    // 32 words per block
    std::vector<uint32_t> overflow;  // 64 words per overflowing block
    uint64_t nblocks = 0, noverflow = 0;
};

// Builds the run blocks of an RLE stream on the host (threads <= 0 picks a default).
void build_run_blocks(const uint8_t *rle, size_t n, const Totals &totals, RunIndex *out, int threads);

}  // namespace msbwt
