// Device layout of the index ("plane blocks"), built on the host at load time.
//
// One block = 128 bytes = one L2 line = 256 consecutive BWT symbols, block b covering
// positions [256*b, 256*b + 256).  A block is 8 chunks of 16 bytes; chunk j (j = 0..7)
// covers the 32 symbols [256*b + 32*j, +32) and is the uint4
//     { plane0, plane1, plane2, meta_j }
// where bit i of plane p is bit p of the 3-bit symbol code at position 256*b + 32*j + i.
// The eight meta words of a block hold, for each symbol s in 0..5, the 40-bit value
//     A[s] = start_index[s] + (number of s in bwt[0 .. 256*b))
// i.e. the new range bound for position 256*b:  meta_s = low 32 bits of A[s] (s = 0..5),
// meta_6 = bits 32..39 of A[0..3] (one byte each), meta_7 = bits 32..39 of A[4], A[5].
//
// A rank is therefore ONE coalesced 128-byte fetch by 8 lanes (16 B each), three XORs, two
// ANDs and a popcount per lane, and an 8-lane reduction -- no run decoding on the query
// path.  The RLE stream is expanded once, at load time; 0.5 byte per symbol, which is what
// 288 GB of HBM is for (a 30x human BWT, 9e10 symbols, is 45 GB).
//
// Number of blocks = total/256 + 1, so that position == total always has a block.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "rle_codec.hpp"

namespace msbwt {

constexpr int kBlockShift = 8;                  // 256 symbols per block
constexpr uint64_t kBlockSymbols = 1ull << kBlockShift;
constexpr size_t kBlockBytes = 128;
constexpr uint64_t kMaxTotal = (1ull << 40) - 1;  // A[s] must fit 40 bits

inline uint64_t plane_block_count(uint64_t total) { return (total >> kBlockShift) + 1; }

// Expands the RLE stream into plane blocks.  `out` must hold plane_block_count(total)*32
// uint32 words.  threads <= 0 picks a default.
void build_plane_blocks(const uint8_t *rle, size_t n, const Totals &totals, uint32_t *out, int threads);

}  // namespace msbwt
