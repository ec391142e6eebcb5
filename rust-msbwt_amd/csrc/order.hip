// Batch order: in which order a batch of k-mers walks the index most cheaply -- the key (round 3), and since round 4 the
// in-library pass that puts a dense batch into that order on the device, inside the launch.
//
// A backward search consumes a k-mer from its LAST symbol.  After j symbols the ranges of different queries are laid out in
// the BWT in lexicographic order of their j-symbol suffixes, and the suffix table holds its entries in exactly that order
// for j = table depth.  A batch whose queries are ordered by  (the last 17 symbols as a string, then the symbols before them
// going leftwards)  therefore reads table lines in ascending order, starts from ascending ranges, and keeps that order
// within each symbol class through the following steps: neighbouring lanes and waves touch neighbouring lines, which the
// L2 / Infinity Cache and the DRAM pages reward -- measured 2.0x on dense batches (C4 read-derived, C3), +15 % on the
// human-scale default batch.
//
// key = sum over t < min(k, 17) of code(kmer[k-1-t]) << (28 + 2 t)  (A C G T -> 0..3: the table index, most significant)
//     | the next up to 14 symbols to the left, kmer[k-18] most significant, in the low 28 bits;
// a '$' / 'N' / invalid symbol among the symbols used gives UINT64_MAX (such queries sort last).
//
// The ordering pass (launch_order_batch / launch_order_finish) is a two-level bucket sort by the top bits of that key, not a
// radix sort of all of it (measured, C4, batch ordered outside the timed region by the top B bits only: 16.7 ms unordered,
// B = 12 14.4, 16 13.5, 20 9.6, 24 and more 8.4).  Every pass is shaped so that what it scatters stays inside a window the L2
// holds -- scattered 8-byte writes over the whole batch cost as much as the search saves (3.6 ms for 10^8 counts):
//   k_order_pack     rows of symbol codes -> 2-bit packed queries (8 bytes per 31-mer; the search kernel reads those directly,
//                    QuerySource::packed) + per-chunk histograms of the level-0 bucket; a query that cannot be packed ('$',
//                    'N', an invalid code) goes onto an exception list and is counted from its row afterwards (k_count_listed),
//                    in stream order, so its result is the one that stays;
//   k_order_offsets / k_order_starts   histogram -> where each chunk's share of each bucket starts;
//   k_order_scatter  level 0, the one global pass: a FEW persistent workgroups take the chunks in turn and send every query
//                    to its bucket (LDS cursors) -- few, so that the lines being filled (one per bucket and array) stay in L2
//                    until they are whole; with every workgroup of the chip at it they were written back in pieces;
//   k_order_level    level 1: one workgroup per level-0 bucket splits it by the next key bits inside the bucket's own window,
//                    and notes for every query WHERE IT CAME FROM in that window;
//   (search)         the lanes kernel counts the packed, ordered batch and writes every count to that place: neighbouring
//                    queries write into one window, not all over the batch;
//   k_order_unsort   level 0 backwards, chunk by chunk: the counts of a chunk's queries are read from the buckets it sent
//                    them to (runs of consecutive places) and written to the caller's buffer inside the chunk's own window.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "order.hpp"
#include "rank_ops.hpp"
#include "search_common.hpp"

namespace msbwt {

__host__ __device__ inline uint64_t order_key_of(const uint8_t *kmer, uint32_t k) {
    const uint32_t depth = k < 17u ? k : 17u, more = k - depth < 14u ? k - depth : 14u;
    uint64_t key = 0;
    for (uint32_t t = 0; t < depth; ++t) {
        const uint32_t s = kmer[k - 1u - t];
        if (s != 1u && s != 2u && s != 3u && s != 5u) return ~0ull;
        key |= uint64_t(s - 1u - (s >> 2)) << (28u + 2u * t);
    }
    for (uint32_t j = 0; j < more; ++j) {
        const uint32_t s = kmer[k - 1u - depth - j];
        if (s != 1u && s != 2u && s != 3u && s != 5u) return ~0ull;
        key |= uint64_t(s - 1u - (s >> 2)) << (26u - 2u * j);
    }
    return key;
}

namespace {

constexpr uint32_t kOrderThreads = 256;

__global__ __launch_bounds__(256) void k_order_keys(const uint8_t *__restrict__ kmers, uint32_t k, uint64_t n, uint64_t *__restrict__ keys) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t q = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < n; q += stride) keys[q] = order_key_of(kmers + q * k, k);
}

// the bucket of a packed query: the top `bits` bits of the low 2 x reach bits of its first word (reach = the symbols the
// suffix table stands for: the table index, in the table's own order), `drop` of them dropped from the top first
__device__ __forceinline__ uint32_t bucket_of(uint64_t word0, uint32_t reach, uint32_t drop, uint32_t bits) {
    const uint32_t have = 2u * reach - drop;  // key bits left
    const uint64_t key = reach >= 32u && drop == 0u ? word0 : (word0 & ((1ull << have) - 1ull));
    return have > bits ? uint32_t(key >> (have - bits)) : uint32_t(key << (bits - have)) & ((1u << bits) - 1u);
}

// Row q of an n x k matrix of symbol codes -> 2-bit words (the layout of QuerySource::packed: the last symbol in bits 0-1 of
// word 0).  Returns false when the row holds something that two bits cannot say ('$', 'N', a code >= 6).  The 32 or 64 bytes
// that END at the row's last symbol arrive as wide unaligned loads (`safe`: they lie inside the caller's buffer) and are
// converted four symbols per instruction, as pack_row_swar does for the search kernel (search_common.hpp).
template <int kWordsPerQuery>
__device__ __forceinline__ bool pack_row(const uint8_t *__restrict__ row, uint32_t k, bool safe, uint64_t (&words)[kWordsPerQuery]) {
    constexpr uint32_t kBlock = 32u * kWordsPerQuery, D = kBlock / 4u;
    uint32_t raw[D];
    if (safe) {
        __builtin_memcpy(raw, row + k - kBlock, kBlock);
    } else {  // the batch's first rows: bytewise, nothing in front of the buffer is touched
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) raw[d] = 0;
        for (uint32_t i = 0; i < k; ++i) {
            const uint32_t at = kBlock - k + i, byte = row[i];
#pragma unroll
            for (uint32_t d = 0; d < D; ++d)
                if ((at >> 2) == d) raw[d] |= byte << ((at & 3u) * 8u);
        }
    }
    const uint32_t first = kBlock - k;  // the row's first byte inside the block
    uint32_t bad = 0;
#pragma unroll
    for (int w = 0; w < kWordsPerQuery; ++w) words[w] = 0;
#pragma unroll
    for (uint32_t u = 0; u < D; ++u) {  // steps 4u .. 4u+3 = bytes 3..0 of dword D-1-u
        const uint32_t d = D - 1u - u, dm = tail_byte_mask(d, first);
        const uint32_t x = raw[d] & dm;
        bad |= ((x + 0x7A7A7A7Au) | x) & 0x80808080u;                       // a byte >= 6
        const uint32_t ys = (x & 0x07070707u & dm) | (0x01010101u & ~dm);   // bytes outside the row read as 'A'
        const uint32_t low2 = ys & 0x03030303u;
        bad |= (low2 - 0x01010101u) & ~low2 & 0x80808080u;                  // a byte that is 0 or 4: '$' / 'N'
        const uint32_t c = ys - 0x01010101u - ((ys >> 2) & 0x01010101u);    // A C G T -> 0..3
        const uint32_t z2 = ((c >> 8) | (c << 2)) & 0x000F000Fu;
        const uint64_t g = ((z2 >> 16) | (z2 << 4)) & 0xFFu;                // the four 2-bit codes, step 4u lowest
        words[u >> 3] |= g << (8u * (u & 7u));
    }
    // (bytes outside the row became 'A' = 0: the words hold zeros beyond 2k bits)
    return bad == 0u;
}

constexpr int kUnroll = 4;       // queries a thread of k_order_pack has in flight per loop iteration
constexpr int kPassUnroll = 8;   // ... of the scatter passes (they are latency-bound otherwise)
constexpr uint32_t kPassThreads = 1024;

// Chunk c = the queries [c * chunk, min(n, (c + 1) * chunk)), one workgroup each; hist[c * nbuckets + b] = how many of them
// fall into level-0 bucket b.  With rows != nullptr the queries are packed here first (packed_out receives them); a row that
// cannot be packed is appended to the exception list and travels on as an all-'A' query.
template <int kWordsPerQuery>
__global__ __launch_bounds__(256) void k_order_pack(const uint8_t *__restrict__ rows, const uint64_t *__restrict__ packed_in,
                                                    uint64_t *__restrict__ packed_out, uint32_t k, uint64_t n, uint32_t chunk, uint32_t reach,
                                                    uint32_t bits0, uint32_t *__restrict__ hist, uint32_t *__restrict__ exceptions,
                                                    unsigned long long *__restrict__ nexceptions) {
    extern __shared__ uint32_t lds_hist[];
    const uint32_t nbuckets = 1u << bits0;
    for (uint32_t b = threadIdx.x; b < nbuckets; b += kOrderThreads) lds_hist[b] = 0u;
    __syncthreads();
    const uint64_t lo = uint64_t(blockIdx.x) * chunk, hi = min(n, lo + chunk);
    for (uint64_t base = lo + threadIdx.x; base < hi; base += kOrderThreads * kUnroll) {
        uint64_t words[kUnroll][kWordsPerQuery];
        bool ok[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {  // all loads first
            const uint64_t q = base + uint64_t(u) * kOrderThreads;
            ok[u] = true;
            if (q < hi) {
                if (rows != nullptr) ok[u] = pack_row<kWordsPerQuery>(rows + q * k, k, q * k + k >= 32u * kWordsPerQuery, words[u]);
                else words[u][0] = packed_in[q * kWordsPerQuery];
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint64_t q = base + uint64_t(u) * kOrderThreads;
            if (q >= hi) continue;
            if (rows != nullptr) {
                if (!ok[u]) {
#pragma unroll
                    for (int w = 0; w < kWordsPerQuery; ++w) words[u][w] = 0;
                    exceptions[atomicAdd(nexceptions, 1ull)] = uint32_t(q);
                }
#pragma unroll
                for (int w = 0; w < kWordsPerQuery; ++w) packed_out[q * kWordsPerQuery + w] = words[u][w];
            }
            atomicAdd(&lds_hist[bucket_of(words[u][0], reach, 0u, bits0)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbuckets; b += kOrderThreads) hist[uint64_t(blockIdx.x) * nbuckets + b] = lds_hist[b];
}

// hist[c][b] -> the number of bucket-b queries in chunks before c (in place); totals[b] = the bucket's size.  One workgroup
// per bucket: its threads take the chunks' counts in turn (a column of the histogram), scanned through LDS.
__global__ __launch_bounds__(256) void k_order_offsets(uint32_t *__restrict__ hist, uint32_t nchunks, uint32_t nbuckets, uint32_t *__restrict__ totals) {
    __shared__ uint32_t part[kOrderThreads];
    const uint32_t b = blockIdx.x;
    const uint32_t per = (nchunks + kOrderThreads - 1) / kOrderThreads, first = threadIdx.x * per, last = min(nchunks, first + per);
    uint32_t sum = 0;
    for (uint32_t w = first; w < last; ++w) sum += hist[uint64_t(w) * nbuckets + b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t t = 0; t < kOrderThreads; ++t) {
            const uint32_t c = part[t];
            part[t] = acc;
            acc += c;
        }
        totals[b] = acc;
    }
    __syncthreads();
    uint32_t acc = part[threadIdx.x];
    for (uint32_t w = first; w < last; ++w) {
        const uint32_t c = hist[uint64_t(w) * nbuckets + b];
        hist[uint64_t(w) * nbuckets + b] = acc;
        acc += c;
    }
}

// starts[b] = sum of totals[< b], starts[nbuckets] = n: one workgroup, nbuckets <= 8192
__global__ __launch_bounds__(256) void k_order_starts(const uint32_t *__restrict__ totals, uint32_t nbuckets, uint32_t *__restrict__ starts) {
    __shared__ uint32_t part[kOrderThreads];
    const uint32_t per = (nbuckets + kOrderThreads - 1) / kOrderThreads, first = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t i = first; i < min(nbuckets, first + per); ++i) sum += totals[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t t = 0; t < kOrderThreads; ++t) {
            const uint32_t c = part[t];
            part[t] = acc;
            acc += c;
        }
        starts[nbuckets] = acc;
    }
    __syncthreads();
    uint32_t acc = part[threadIdx.x];
    for (uint32_t i = first; i < min(nbuckets, first + per); ++i) {
        starts[i] = acc;
        acc += totals[i];
    }
}

// An ELEMENT of the ordered batch: the query's kWordsPerQuery words + one more word -- after level 0 its index in the caller's
// batch, after level 1 its place after level 0 (where its count goes).  One array, one store per element: the passes are
// bound by how many separate stores they issue (about 10^8 per millisecond), not by bytes.
template <int kWordsPerQuery>
__device__ __forceinline__ void store_element(uint64_t *__restrict__ out, uint64_t at, const uint64_t (&words)[kWordsPerQuery], uint32_t tag) {
    if constexpr (kWordsPerQuery == 1) {
        *reinterpret_cast<uint4 *>(out + at * 2) = make_uint4(uint32_t(words[0]), uint32_t(words[0] >> 32), tag, 0u);
    } else {
        out[at * 3] = words[0];
        out[at * 3 + 1] = words[1];
        out[at * 3 + 2] = tag;
    }
}

// part[0 .. 1024) -> exclusive prefix sums + `base`, by the workgroup's first wave (16 values per lane, then across the lanes);
// the caller synchronises before and after
__device__ __forceinline__ void scan_1024(uint32_t *part, uint32_t base) {
    if (threadIdx.x < 64u) {
        uint32_t mine[16], total = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            mine[i] = total;
            total += part[threadIdx.x * 16u + i];
        }
        uint32_t incl = total;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (int(threadIdx.x) >= d) incl += up;
        }
        const uint32_t before = base + incl - total;
#pragma unroll
        for (int i = 0; i < 16; ++i) part[threadIdx.x * 16u + i] = before + mine[i];
    }
}

// counters[0 .. n) (LDS) -> their exclusive prefix sums + `base`, in place, by the whole workgroup (n <= 4096)
__device__ __forceinline__ void scan_in_place(uint32_t *counters, uint32_t n, uint32_t *part, uint32_t base) {
    const uint32_t per = (n + kPassThreads - 1) / kPassThreads, first = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t i = first; i < min(n, first + per); ++i) sum += counters[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    scan_1024(part, base);
    __syncthreads();
    uint32_t acc = part[threadIdx.x];
    for (uint32_t i = first; i < min(n, first + per); ++i) {
        const uint32_t c = counters[i];
        counters[i] = acc;
        acc += c;
    }
    __syncthreads();
}

// A TILE of elements goes out in RUNS (round 4, last shape of the passes).  The passes cost about a millisecond per 10^8
// separate stores whatever their size, so a workgroup first sorts the tile it holds in registers by bucket INSIDE LDS -- a
// counting sort: LDS atomics give every element its rank in its bucket, a scan the bucket's place in the tile -- and then
// stores the tile in sorted order: neighbouring lanes hold neighbouring elements of one bucket and their stores go to
// neighbouring addresses (one request per run instead of one per element).
//   kPer elements per thread (words, tag, bucket, live); nb buckets; cursor[b] (LDS) = where bucket b's next element goes
//   in `out`, advanced by what the tile adds; tile_start (LDS, nb) and stage (LDS, kPer x 1024 elements) are scratch.
template <int kWordsPerQuery>
struct OrderTile {
    static constexpr int kPer = kWordsPerQuery == 1 ? 8 : 4;            // 8192 16-byte or 4096 24-byte elements: 128 / 96 KiB of LDS
    static constexpr uint32_t kElems = uint32_t(kPer) * kPassThreads;
    static constexpr uint32_t kStride = kWordsPerQuery + 1;
};

template <int kWordsPerQuery>
__device__ __forceinline__ void scatter_tile_in_runs(const uint64_t (&words)[OrderTile<kWordsPerQuery>::kPer][kWordsPerQuery],
                                                     const uint32_t (&tag)[OrderTile<kWordsPerQuery>::kPer],
                                                     const uint32_t (&bucket)[OrderTile<kWordsPerQuery>::kPer], uint32_t live_mask, uint32_t ntile,
                                                     uint32_t nb, uint32_t reach, uint32_t drop, uint32_t bits, uint32_t *tile_start, uint32_t *cursor,
                                                     uint32_t *part, uint64_t *stage, uint64_t *__restrict__ out) {
    constexpr int kPer = OrderTile<kWordsPerQuery>::kPer;
    constexpr uint32_t kStride = OrderTile<kWordsPerQuery>::kStride;
    for (uint32_t b = threadIdx.x; b < nb; b += kPassThreads) tile_start[b] = 0u;
    __syncthreads();
    uint32_t rank[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) rank[u] = (live_mask >> u) & 1u ? atomicAdd(&tile_start[bucket[u]], 1u) : 0u;
    __syncthreads();
    // what each bucket gets from this tile (kept by the thread that owns the bucket), then the buckets' places in the tile
    const uint32_t per = (nb + kPassThreads - 1) / kPassThreads, first = threadIdx.x * per;
    uint32_t added[4];  // nb <= 4096
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) added[i] = (i < per && first + i < nb) ? tile_start[first + i] : 0u;
    scan_in_place(tile_start, nb, part, 0u);
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        if (((live_mask >> u) & 1u) == 0u) continue;
        const uint32_t at = tile_start[bucket[u]] + rank[u];
#pragma unroll
        for (int w = 0; w < kWordsPerQuery; ++w) stage[at * kStride + w] = words[u][w];
        stage[at * kStride + kWordsPerQuery] = tag[u];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const uint32_t at = threadIdx.x + uint32_t(u) * kPassThreads;  // neighbouring lanes, neighbouring places of the sorted tile
        if (at >= ntile) continue;
        uint64_t e[kWordsPerQuery];
#pragma unroll
        for (int w = 0; w < kWordsPerQuery; ++w) e[w] = stage[at * kStride + w];
        const uint32_t b = bucket_of(e[0], reach, drop, bits);
        store_element<kWordsPerQuery>(out, cursor[b] + (at - tile_start[b]), e, uint32_t(stage[at * kStride + kWordsPerQuery]));
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i)
        if (i < per && first + i < nb) cursor[first + i] += added[i];
    __syncthreads();
}

// Level 0, the one global pass: the workgroups take the chunks of k_order_pack in turn; every query of a chunk goes to its
// place in its bucket, tile by tile (scatter_tile_in_runs), as an element that carries its index in the caller's batch.
template <int kWordsPerQuery>
__global__ __launch_bounds__(1024) void k_order_scatter(const uint64_t *__restrict__ packed_in, uint64_t n, uint32_t chunk, uint32_t nchunks,
                                                        uint32_t reach, uint32_t bits0, const uint32_t *__restrict__ offsets,
                                                        const uint32_t *__restrict__ starts, uint64_t *__restrict__ elems_out) {
    using Tile = OrderTile<kWordsPerQuery>;
    extern __shared__ uint64_t order_lds[];  // stage, then tile_start[nb], cursor[nb]
    __shared__ uint32_t part[kPassThreads];
    const uint32_t nb = 1u << bits0;
    uint64_t *stage = order_lds;
    uint32_t *tile_start = reinterpret_cast<uint32_t *>(order_lds + size_t(Tile::kElems) * Tile::kStride), *cursor = tile_start + nb;
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < nb; b += kPassThreads) cursor[b] = starts[b] + offsets[uint64_t(c) * nb + b];
        __syncthreads();
        const uint64_t lo = uint64_t(c) * chunk, hi = min(n, lo + chunk);
        for (uint64_t tile = lo; tile < hi; tile += Tile::kElems) {
            uint64_t words[Tile::kPer][kWordsPerQuery];
            uint32_t tag[Tile::kPer], bucket[Tile::kPer], live = 0;
#pragma unroll
            for (int u = 0; u < Tile::kPer; ++u) {
                const uint64_t q = tile + uint64_t(u) * kPassThreads + threadIdx.x;
#pragma unroll
                for (int w = 0; w < kWordsPerQuery; ++w) words[u][w] = q < hi ? packed_in[q * kWordsPerQuery + w] : 0ull;
                tag[u] = uint32_t(q);
                bucket[u] = bucket_of(words[u][0], reach, 0u, bits0);
                live |= q < hi ? 1u << u : 0u;
            }
            scatter_tile_in_runs<kWordsPerQuery>(words, tag, bucket, live, uint32_t(min(uint64_t(Tile::kElems), hi - tile)), nb, reach, 0u, bits0, tile_start,
                                                 cursor, part, stage, elems_out);
        }
    }
}

// Level 1: every level-0 bucket (`nparents` of them, parent_starts) is split once more by the NEXT `bits` key bits by ONE
// workgroup, inside the bucket's own window, however large the bucket is: a counting pass over the bucket gives every
// sub-bucket its place, then the bucket goes out tile by tile in runs.  The element written carries the level-0 place the
// query came from: the search writes its count THERE (inside the same window), k_order_unsort takes it from there.
template <int kWordsPerQuery>
__global__ __launch_bounds__(1024) void k_order_level(const uint64_t *__restrict__ elems_in, const uint32_t *__restrict__ parent_starts,
                                                      uint32_t nparents, uint32_t reach, uint32_t drop, uint32_t bits,
                                                      uint64_t *__restrict__ elems_out) {
    using Tile = OrderTile<kWordsPerQuery>;
    constexpr uint32_t kStride = Tile::kStride;
    extern __shared__ uint64_t order_lds[];  // stage, then tile_start[nb], cursor[nb]
    __shared__ uint32_t part[kPassThreads];
    const uint32_t nb = 1u << bits;
    uint64_t *stage = order_lds;
    uint32_t *tile_start = reinterpret_cast<uint32_t *>(order_lds + size_t(Tile::kElems) * kStride), *cursor = tile_start + nb;
    for (uint32_t parent = blockIdx.x; parent < nparents; parent += gridDim.x) {
        const uint32_t lo = parent_starts[parent], hi = parent_starts[parent + 1];
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < nb; b += kPassThreads) cursor[b] = 0u;
        __syncthreads();
        for (uint32_t base = lo + threadIdx.x; base < hi; base += kPassThreads * kPassUnroll) {
            uint64_t w0[kPassUnroll];
#pragma unroll
            for (int u = 0; u < kPassUnroll; ++u) {
                const uint32_t q = base + uint32_t(u) * kPassThreads;
                w0[u] = q < hi ? elems_in[uint64_t(q) * kStride] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < kPassUnroll; ++u)
                if (base + uint32_t(u) * kPassThreads < hi) atomicAdd(&cursor[bucket_of(w0[u], reach, drop, bits)], 1u);
        }
        __syncthreads();
        scan_in_place(cursor, nb, part, lo);
        for (uint32_t tile = lo; tile < hi; tile += Tile::kElems) {
            uint64_t words[Tile::kPer][kWordsPerQuery];
            uint32_t tag[Tile::kPer], bucket[Tile::kPer], live = 0;
#pragma unroll
            for (int u = 0; u < Tile::kPer; ++u) {
                const uint32_t q = tile + uint32_t(u) * kPassThreads + threadIdx.x;
#pragma unroll
                for (int w = 0; w < kWordsPerQuery; ++w) words[u][w] = q < hi ? elems_in[uint64_t(q) * kStride + w] : 0ull;
                tag[u] = q;
                bucket[u] = bucket_of(words[u][0], reach, drop, bits);
                live |= q < hi ? 1u << u : 0u;
            }
            scatter_tile_in_runs<kWordsPerQuery>(words, tag, bucket, live, min(Tile::kElems, hi - tile), nb, reach, drop, bits, tile_start, cursor, part, stage,
                                                 elems_out);
        }
    }
}

// Level 0 backwards, one workgroup per chunk: chunk c's queries sit in the buckets as runs of consecutive places -- bucket b:
// from offsets[c][b] into the bucket, as many as the next chunk's offset says -- where the level-0 element names each one's
// index in the caller's batch (inside the chunk's own window) and counts[] holds its count.  The runs are laid end to end
// (their lengths scanned in LDS) and the threads walk that sequence, a wave reading neighbouring places; the counts are
// gathered into the chunk's window IN LDS and leave for the caller's buffer as whole lines.
__global__ __launch_bounds__(1024) void k_order_unsort(const uint64_t *__restrict__ counts, const uint64_t *__restrict__ elems0, uint32_t stride,
                                                       const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ totals,
                                                       const uint32_t *__restrict__ starts, uint64_t n, uint32_t chunk, uint32_t nchunks,
                                                       uint32_t nbuckets, uint64_t *__restrict__ out) {
    extern __shared__ uint64_t window[];  // chunk counts, then run_start[nbuckets], prefix[nbuckets + 1]
    __shared__ uint32_t part[kPassThreads];
    uint32_t *run_start = reinterpret_cast<uint32_t *>(window + chunk), *prefix = run_start + nbuckets;
    const uint32_t c = blockIdx.x;
    const uint64_t first_query = uint64_t(c) * chunk;
    const uint32_t per = (nbuckets + kPassThreads - 1) / kPassThreads, first = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t b = first; b < min(nbuckets, first + per); ++b) {
        const uint32_t off = offsets[uint64_t(c) * nbuckets + b];
        const uint32_t next = c + 1u < nchunks ? offsets[uint64_t(c + 1u) * nbuckets + b] : totals[b];
        run_start[b] = starts[b] + off;
        prefix[b] = next - off;  // the run's length, for now
        sum += next - off;
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    scan_1024(part, 0u);
    __syncthreads();
    uint32_t acc = part[threadIdx.x];
    for (uint32_t b = first; b < min(nbuckets, first + per); ++b) {
        const uint32_t len = prefix[b];
        prefix[b] = acc;
        acc += len;
    }
    if (threadIdx.x == kPassThreads - 1u) prefix[nbuckets] = acc;
    __syncthreads();
    const uint32_t total = prefix[nbuckets];  // = the chunk's queries
    for (uint32_t base = threadIdx.x; base < total; base += kPassThreads * kPassUnroll) {
        uint32_t j[kPassUnroll], to[kPassUnroll];
        uint64_t value[kPassUnroll];
#pragma unroll
        for (int u = 0; u < kPassUnroll; ++u) {
            const uint32_t e = base + uint32_t(u) * kPassThreads;
            j[u] = 0u;
            if (e < total) {  // the run that holds element e: the last one that starts at or before it
                uint32_t lo = 0, hi = nbuckets;
                while (hi - lo > 1u) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (prefix[mid] <= e) lo = mid; else hi = mid;
                }
                j[u] = run_start[lo] + (e - prefix[lo]);
            }
        }
#pragma unroll
        for (int u = 0; u < kPassUnroll; ++u) {
            const bool live = base + uint32_t(u) * kPassThreads < total;
            to[u] = live ? uint32_t(elems0[uint64_t(j[u]) * stride + (stride - 1u)]) : 0u;
            value[u] = live ? counts[j[u]] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kPassUnroll; ++u)
            if (base + uint32_t(u) * kPassThreads < total) window[to[u] - uint32_t(first_query)] = value[u];
    }
    __syncthreads();
    const uint32_t mine = uint32_t(min(uint64_t(chunk), n - first_query));
    for (uint32_t i = threadIdx.x; i < mine; i += kPassThreads) out[first_query + i] = window[i];
}

// counts[list[i]] = count_kmer(row list[i]) for the *nlist queries of the exception list (rows that two bits cannot say):
// the any-k kernel's loop, one 8-lane group per query, read through the list.  Launched with a fixed grid: the list's
// length is only known on the device.
__global__ __launch_bounds__(256) void k_count_listed(const uint4 *__restrict__ blocks, uint64_t total, const uint8_t *__restrict__ kmers, uint32_t k,
                                                      const uint32_t *__restrict__ list, const unsigned long long *__restrict__ nlist,
                                                      uint64_t *__restrict__ counts, uint32_t *__restrict__ flags) {
    const uint32_t sub = threadIdx.x & (kGroup - 1);
    const uint64_t ngroups = (uint64_t(gridDim.x) * blockDim.x) / kGroup, n = *nlist;
    for (uint64_t i = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) / kGroup; i < n; i += ngroups) {
        const uint64_t q = list[i];
        const uint8_t *kmer = kmers + q * k;
        uint32_t bad = 0;
        for (uint32_t j = sub; j < k; j += kGroup) bad |= (kmer[j] >= 6u) ? 1u : 0u;  // the reference asserts (msbwt_core.rs:127)
        bad = group_sum(bad);
        uint64_t result;
        if (bad) {
            result = ~0ull;
            if (sub == 0) atomicOr(flags, 1u);  // kFlagInvalidSymbol
        } else {
            Range r{0, total};
            bool broken = false;
            for (uint32_t j = k; j-- > 0 && r.l != r.h && !(broken = r.h > total || r.l > r.h);) r = constrain(blocks, kmer[j], r.l, r.h, sub);
            result = broken ? ~0ull : r.h - r.l;
            if (broken && sub == 0) atomicOr(flags, 4u);  // kFlagInternal
        }
        if (sub == 0) counts[q] = result;
    }
}

// 2-bit words -> rows of symbol codes (indexes the lanes kernel does not serve: run blocks)
__global__ __launch_bounds__(256) void k_unpack_rows(const uint64_t *__restrict__ packed, uint32_t k, uint64_t n, uint8_t *__restrict__ rows) {
    const uint32_t words = k > 32u ? 2u : 1u;
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t q = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < n; q += stride)
        for (uint32_t t = 0; t < k; ++t) {
            const uint32_t c = uint32_t(packed[q * words + (t >> 5)] >> (2u * (t & 31u))) & 3u;
            rows[q * k + k - 1u - t] = uint8_t(c + 1u + (c == 3u ? 1u : 0u));
        }
}

inline uint64_t round_up(uint64_t v, uint64_t to) { return (v + to - 1) / to * to; }

}  // namespace

void order_keys_host(const uint8_t *kmers, uint32_t k, uint64_t n, uint64_t *keys) {
    for (uint64_t q = 0; q < n; ++q) keys[q] = order_key_of(kmers + q * k, k);
}

hipError_t launch_order_keys(const uint8_t *d_kmers, uint32_t k, uint64_t n, uint64_t *d_keys, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_order_keys, dim3(uint32_t(blocks > 256 * 32 ? 256 * 32 : blocks)), dim3(256), 0, stream, d_kmers, k, n, d_keys);
    return hipGetLastError();
}

namespace {
uint32_t env_or(const char *name, uint32_t fallback) {
    const char *e = std::getenv(name);
    return e && std::atoi(e) > 0 ? uint32_t(std::atoi(e)) : fallback;
}
}  // namespace

OrderPlan plan_order(uint64_t n, uint32_t k, uint32_t reach, uint32_t bits, bool from_rows) {
    OrderPlan p{};
    p.n = n;
    p.k = k;
    p.words = k > 32u ? 2u : 1u;
    p.reach = std::min(std::max(reach, 1u), std::min(k, 32u));
    bits = std::max(1u, std::min(std::min(bits, 2u * p.reach), 23u));
    // chunks of 16 Ki queries: a chunk's counts (128 KiB) fit the LDS of the workgroup that returns them to the caller's order
    p.chunk = 16384;
    p.nchunks = uint32_t((n + p.chunk - 1) / p.chunk);
    // Level 0 is the one GLOBAL pass, at most 2^11 buckets: what a chunk adds to a bucket is then a run of about 8 elements,
    // a whole line; the remaining bits (at most 12: LDS counters) follow in level 1, inside each bucket's window.
    uint32_t b0 = 3;
    while (b0 < 11u && (uint64_t(8) << (b0 + 1)) <= std::min<uint64_t>(p.chunk, n)) ++b0;
    p.bits0 = std::min(b0, bits);
    p.bits1 = std::min(bits - p.bits0, 11u);
    p.wg0 = std::min(p.nchunks, env_or("MSBWT_ORDER_WG0", 1024));
    p.wg1 = std::min(1u << p.bits0, env_or("MSBWT_ORDER_WG1", 1024));
    const uint64_t nb0 = 1ull << p.bits0;
    uint64_t at = 0;
    auto take = [&](uint64_t bytes) { const uint64_t here = at; at += round_up(bytes, 256); return here; };
    p.off_packed_rows = from_rows ? take(n * p.words * 8) : 0;   // the packed batch in the caller's order, when it arrives as rows
    p.off_elems0 = take(n * (p.words + 1) * 8);                  // after level 0: {query, index in the caller's batch}
    p.off_elems1 = p.bits1 ? take(n * (p.words + 1) * 8) : p.off_elems0;   // after level 1: {query, place after level 0}
    p.off_counts = take(n * 8);                                  // the counts, by place after level 0
    p.off_hist = take(uint64_t(p.nchunks) * nb0 * 4);
    p.off_totals = take(nb0 * 4);
    p.off_starts = take((nb0 + 1) * 4);
    p.off_exceptions = take(from_rows ? n * 4 : 4);
    p.off_nexceptions = take(8);
    p.scratch_bytes = at;
    p.from_rows = from_rows;
    return p;
}

hipError_t launch_order_batch(const OrderPlan &p, const uint8_t *d_rows, const uint64_t *d_packed, void *d_scratch, hipStream_t stream,
                              const uint64_t **ordered, bool *place_inline, uint64_t **counts) {
    if (p.n == 0 || p.n > 0xFFFFFFFFull || (p.from_rows ? d_rows == nullptr : d_packed == nullptr)) return hipErrorInvalidValue;
    char *s = static_cast<char *>(d_scratch);
    uint64_t *rows_packed = reinterpret_cast<uint64_t *>(s + p.off_packed_rows), *e0 = reinterpret_cast<uint64_t *>(s + p.off_elems0),
             *e1 = reinterpret_cast<uint64_t *>(s + p.off_elems1);
    uint32_t *hist = reinterpret_cast<uint32_t *>(s + p.off_hist), *totals = reinterpret_cast<uint32_t *>(s + p.off_totals),
             *starts = reinterpret_cast<uint32_t *>(s + p.off_starts), *exceptions = reinterpret_cast<uint32_t *>(s + p.off_exceptions);
    unsigned long long *nexc = reinterpret_cast<unsigned long long *>(s + p.off_nexceptions);
    const uint32_t b0 = p.bits0, nb0 = 1u << b0;
    hipError_t e = hipMemsetAsync(nexc, 0, 8, stream);
    if (e != hipSuccess) return e;
    const uint64_t *src_packed = p.from_rows ? rows_packed : d_packed;
    if (p.words == 1)
        hipLaunchKernelGGL((k_order_pack<1>), dim3(p.nchunks), dim3(kOrderThreads), nb0 * 4, stream, p.from_rows ? d_rows : nullptr, d_packed, rows_packed, p.k,
                           p.n, p.chunk, p.reach, b0, hist, exceptions, nexc);
    else
        hipLaunchKernelGGL((k_order_pack<2>), dim3(p.nchunks), dim3(kOrderThreads), nb0 * 4, stream, p.from_rows ? d_rows : nullptr, d_packed, rows_packed, p.k,
                           p.n, p.chunk, p.reach, b0, hist, exceptions, nexc);
    hipLaunchKernelGGL(k_order_offsets, dim3(nb0), dim3(kOrderThreads), 0, stream, hist, p.nchunks, nb0, totals);
    hipLaunchKernelGGL(k_order_starts, dim3(1), dim3(kOrderThreads), 0, stream, totals, nb0, starts);
    // the tiles are staged in LDS: 128 KiB (96 KiB for two-word queries) + two arrays of one u32 per bucket -- one workgroup per CU
    const auto lds_for = [&](uint32_t nbuckets) { return size_t(p.words == 1 ? OrderTile<1>::kElems * 16u : OrderTile<2>::kElems * 24u) + size_t(nbuckets) * 8; };
    const int lds_cap = 160 * 1024 - 4096 - 64;  // (per call: the attribute belongs to the current device)
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_order_scatter<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_cap)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_order_scatter<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_cap)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_order_level<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_cap)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_order_level<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_cap)) != hipSuccess) return e;
    if (p.words == 1)
        hipLaunchKernelGGL((k_order_scatter<1>), dim3(p.wg0), dim3(kPassThreads), lds_for(nb0), stream, src_packed, p.n, p.chunk, p.nchunks, p.reach, b0, hist, starts, e0);
    else
        hipLaunchKernelGGL((k_order_scatter<2>), dim3(p.wg0), dim3(kPassThreads), lds_for(nb0), stream, src_packed, p.n, p.chunk, p.nchunks, p.reach, b0, hist, starts, e0);
    *ordered = e0;
    *place_inline = false;  // one level only: the search counts in level-0 order and its counts are in place already
    if (p.bits1) {
        if (p.words == 1)
            hipLaunchKernelGGL((k_order_level<1>), dim3(p.wg1), dim3(kPassThreads), lds_for(1u << p.bits1), stream, e0, starts, nb0, p.reach, b0, p.bits1, e1);
        else
            hipLaunchKernelGGL((k_order_level<2>), dim3(p.wg1), dim3(kPassThreads), lds_for(1u << p.bits1), stream, e0, starts, nb0, p.reach, b0, p.bits1, e1);
        *ordered = e1;
        *place_inline = true;
    }
    *counts = reinterpret_cast<uint64_t *>(s + p.off_counts);
    return hipGetLastError();
}

hipError_t launch_order_finish(const OrderPlan &p, void *d_scratch, uint64_t *d_out, hipStream_t stream) {
    char *s = static_cast<char *>(d_scratch);
    const uint32_t nb0 = 1u << p.bits0;
    const size_t lds = size_t(p.chunk) * 8 + (size_t(2) * nb0 + 1) * 4;
    // (per call, not once per process: the attribute belongs to the current device, and replicas live on several)
    const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void *>(k_order_unsort), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096 - 64);
    if (raised != hipSuccess) return raised;
    hipLaunchKernelGGL(k_order_unsort, dim3(p.nchunks), dim3(kPassThreads), lds, stream, reinterpret_cast<const uint64_t *>(s + p.off_counts),
                       reinterpret_cast<const uint64_t *>(s + p.off_elems0), p.words + 1u, reinterpret_cast<const uint32_t *>(s + p.off_hist),
                       reinterpret_cast<const uint32_t *>(s + p.off_totals), reinterpret_cast<const uint32_t *>(s + p.off_starts), p.n, p.chunk, p.nchunks, nb0, d_out);
    return hipGetLastError();
}

hipError_t launch_unpack_rows(const uint64_t *d_packed, uint32_t k, uint64_t n, uint8_t *d_rows, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_unpack_rows, dim3(uint32_t(std::min<uint64_t>(256 * 16, (n + 255) / 256))), dim3(256), 0, stream, d_packed, k, n, d_rows);
    return hipGetLastError();
}

hipError_t launch_count_exceptions(const OrderPlan &p, const void *d_blocks, uint64_t total, const uint8_t *d_rows, void *d_scratch, uint64_t *d_counts,
                                   uint32_t *d_flags, hipStream_t stream) {
    if (!p.from_rows) return hipSuccess;
    char *s = static_cast<char *>(d_scratch);
    hipLaunchKernelGGL(k_count_listed, dim3(256), dim3(256), 0, stream, static_cast<const uint4 *>(d_blocks), total, d_rows, p.k,
                       reinterpret_cast<const uint32_t *>(s + p.off_exceptions), reinterpret_cast<const unsigned long long *>(s + p.off_nexceptions), d_counts,
                       d_flags);
    return hipGetLastError();
}

}  // namespace msbwt
