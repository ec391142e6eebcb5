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
// The ordering pass (launch_order_batch) is a bucket sort by the top bits of that key, not a radix sort of all of it
// (measured, C4, batch ordered outside the timed region by the top B bits only: 16.7 ms unordered, B = 12 14.4, 16 13.5,
// 20 9.6, 24 and more 8.4):
//   k_order_pack   rows of symbol codes -> 2-bit packed queries (8 bytes per 31-mer; the search kernel reads those directly,
//                  QuerySource::packed) + per-workgroup histograms of the coarse bucket in LDS; a query that cannot be
//                  packed ('$', 'N', an invalid code) goes onto an exception list and is counted from its row afterwards
//                  (k_count_listed), in stream order, so its result is the one that stays;
//   k_order_offsets / k_order_starts   histogram -> where each workgroup's share of each bucket starts;
//   k_order_scatter   every query to its coarse bucket (LDS cursors), with its index in the caller's batch;
//   k_order_level  one workgroup per bucket of the level before: a further bucket pass on the next bits, inside a window that
//                  stays in L2 / the Infinity Cache.
// The search kernel then counts the packed, ordered batch and writes every count to its query's own place in the
// caller's buffer (QuerySource::out_index): nothing is moved back.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

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

constexpr int kUnroll = 4;  // queries a thread has in flight per loop iteration (the passes are latency-bound otherwise)

// Workgroup w owns the queries [w * chunk, min(n, (w + 1) * chunk)); hist[w * nbuckets + b] = how many of them fall into
// level-0 bucket b.  With rows != nullptr the queries are packed here first (packed_out receives them); a row that cannot
// be packed is appended to the exception list and travels on as an all-'A' query.
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

// hist[w][b] -> the number of bucket-b queries in workgroups before w (in place); totals[b] = the bucket's size.  One workgroup
// per bucket: its threads take the workgroups' counts in turn (a column of the histogram), scanned through LDS.
__global__ __launch_bounds__(256) void k_order_offsets(uint32_t *__restrict__ hist, uint32_t nwg, uint32_t nbuckets, uint32_t *__restrict__ totals) {
    __shared__ uint32_t part[kOrderThreads];
    const uint32_t b = blockIdx.x;
    const uint32_t per = (nwg + kOrderThreads - 1) / kOrderThreads, first = threadIdx.x * per, last = min(nwg, first + per);
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

// Level 0, the one global pass: every query of workgroup w's chunk (the chunks of k_order_pack) to its place in its bucket.
// Few buckets (2^10 at most), so that a workgroup's share of a bucket is a run of whole lines.
template <int kWordsPerQuery>
__global__ __launch_bounds__(256) void k_order_scatter(const uint64_t *__restrict__ packed_in, uint64_t n, uint32_t chunk, uint32_t reach,
                                                       uint32_t bits0, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ starts,
                                                       uint64_t *__restrict__ packed_out, uint32_t *__restrict__ index_out) {
    extern __shared__ uint32_t cursor[];
    const uint32_t nbuckets = 1u << bits0;
    for (uint32_t b = threadIdx.x; b < nbuckets; b += kOrderThreads) cursor[b] = starts[b] + offsets[uint64_t(blockIdx.x) * nbuckets + b];
    __syncthreads();
    const uint64_t lo = uint64_t(blockIdx.x) * chunk, hi = min(n, lo + chunk);
    for (uint64_t base = lo + threadIdx.x; base < hi; base += kOrderThreads * kUnroll) {
        uint64_t words[kUnroll][kWordsPerQuery];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint64_t q = base + uint64_t(u) * kOrderThreads;
#pragma unroll
            for (int w = 0; w < kWordsPerQuery; ++w) words[u][w] = q < hi ? packed_in[q * kWordsPerQuery + w] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint64_t q = base + uint64_t(u) * kOrderThreads;
            if (q >= hi) continue;
            const uint32_t at = atomicAdd(&cursor[bucket_of(words[u][0], reach, 0u, bits0)], 1u);
#pragma unroll
            for (int w = 0; w < kWordsPerQuery; ++w) packed_out[uint64_t(at) * kWordsPerQuery + w] = words[u][w];
            index_out[at] = uint32_t(q);
        }
    }
}

// A further level: every bucket of the level before (`nparents` of them, parent_starts) is split once more by the NEXT `bits`
// key bits -- count, scan, scatter, by ONE workgroup inside the bucket's own window (L2- or Infinity-Cache-resident), however
// large the bucket is.  child_starts (optional): the starts of the nparents x 2^bits buckets of this level, + the end.
template <int kWordsPerQuery>
__global__ __launch_bounds__(256) void k_order_level(const uint64_t *__restrict__ packed_in, const uint32_t *__restrict__ index_in,
                                                     const uint32_t *__restrict__ parent_starts, uint32_t nparents, uint32_t reach, uint32_t drop,
                                                     uint32_t bits, uint64_t *__restrict__ packed_out, uint32_t *__restrict__ index_out,
                                                     uint32_t *__restrict__ child_starts) {
    extern __shared__ uint32_t fine[];  // 2^bits counters, then cursors
    __shared__ uint32_t part[kOrderThreads];
    const uint32_t nfine = 1u << bits;
    for (uint32_t parent = blockIdx.x; parent < nparents; parent += gridDim.x) {
        const uint32_t lo = parent_starts[parent], hi = parent_starts[parent + 1];
        __syncthreads();  // (the previous parent's cursors are no longer read)
        for (uint32_t b = threadIdx.x; b < nfine; b += kOrderThreads) fine[b] = 0u;
        __syncthreads();
        for (uint32_t base = lo + threadIdx.x; base < hi; base += kOrderThreads * kUnroll) {
            uint64_t w0[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const uint32_t q = base + uint32_t(u) * kOrderThreads;
                w0[u] = q < hi ? packed_in[uint64_t(q) * kWordsPerQuery] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)
                if (base + uint32_t(u) * kOrderThreads < hi) atomicAdd(&fine[bucket_of(w0[u], reach, drop, bits)], 1u);
        }
        __syncthreads();
        const uint32_t per = nfine / kOrderThreads > 0 ? nfine / kOrderThreads : 1u, first = threadIdx.x * per;
        uint32_t sum = 0;
        for (uint32_t i = first; i < min(nfine, first + per); ++i) sum += fine[i];
        part[threadIdx.x] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t acc = lo;
            for (uint32_t t = 0; t < kOrderThreads; ++t) {
                const uint32_t c = part[t];
                part[t] = acc;
                acc += c;
            }
        }
        __syncthreads();
        uint32_t acc = part[threadIdx.x];
        for (uint32_t i = first; i < min(nfine, first + per); ++i) {
            const uint32_t c = fine[i];
            fine[i] = acc;
            if (child_starts != nullptr) child_starts[uint64_t(parent) * nfine + i] = acc;
            acc += c;
        }
        if (child_starts != nullptr && parent + 1u == nparents && threadIdx.x == 0u) child_starts[uint64_t(nparents) * nfine] = hi;
        __syncthreads();
        for (uint32_t base = lo + threadIdx.x; base < hi; base += kOrderThreads * kUnroll) {
            uint64_t words[kUnroll][kWordsPerQuery];
            uint32_t idx[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const uint32_t q = base + uint32_t(u) * kOrderThreads;
#pragma unroll
                for (int w = 0; w < kWordsPerQuery; ++w) words[u][w] = q < hi ? packed_in[uint64_t(q) * kWordsPerQuery + w] : 0ull;
                idx[u] = q < hi ? index_in[q] : 0u;
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (base + uint32_t(u) * kOrderThreads >= hi) continue;
                const uint32_t at = atomicAdd(&fine[bucket_of(words[u][0], reach, drop, bits)], 1u);
#pragma unroll
                for (int w = 0; w < kWordsPerQuery; ++w) packed_out[uint64_t(at) * kWordsPerQuery + w] = words[u][w];
                index_out[at] = idx[u];
            }
        }
    }
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

OrderPlan plan_order(uint64_t n, uint32_t k, uint32_t reach, uint32_t bits, bool from_rows) {
    OrderPlan p{};
    p.n = n;
    p.k = k;
    p.words = k > 32u ? 2u : 1u;
    p.reach = std::min(std::max(reach, 1u), std::min(k, 32u));
    bits = std::max(1u, std::min(std::min(bits, 2u * p.reach), 34u));
    // Level 0 is the one GLOBAL pass: at most 10 bits, so that what a workgroup adds to a bucket is a run of whole lines
    // (4096 buckets made 64-byte runs: 4.5 ms for 10^8 queries against 1 ms); the remaining bits follow in passes of at
    // most 12 (LDS counters), each inside its parent bucket's window.
    p.nlevels = 0;
    for (uint32_t left = bits; left > 0 && p.nlevels < 3;) {
        const uint32_t cap = p.nlevels == 0 ? 10u : 12u, take = std::min(left, cap);
        p.level_bits[p.nlevels++] = take;
        left -= take;
    }
    // chunks of about 64 Ki queries, at most 2048 workgroups, a multiple of the workgroup size
    const uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>(2048, (n + 65535) / 65536));
    p.chunk = uint32_t(round_up((n + want - 1) / want, kOrderThreads));
    p.nwg = uint32_t((n + p.chunk - 1) / p.chunk);
    const uint64_t nb0 = 1ull << p.level_bits[0];
    uint64_t at = 0;
    auto take = [&](uint64_t bytes) { const uint64_t here = at; at += round_up(bytes, 256); return here; };
    // two (packed, index) buffers to ping-pong between the levels, + the packed batch in the caller's order when it arrives as rows
    p.off_packed_rows = from_rows ? take(n * p.words * 8) : 0;
    p.off_packed[0] = take(n * p.words * 8);
    p.off_index[0] = take(n * 4);
    p.off_packed[1] = p.nlevels > 1 ? take(n * p.words * 8) : p.off_packed[0];
    p.off_index[1] = p.nlevels > 1 ? take(n * 4) : p.off_index[0];
    p.off_hist = take(uint64_t(p.nwg) * nb0 * 4);
    p.off_totals = take(nb0 * 4);
    p.off_starts[0] = take((nb0 + 1) * 4);
    p.off_starts[1] = p.nlevels > 2 ? take(((nb0 << p.level_bits[1]) + 1) * 4) : 0;
    p.off_exceptions = take(from_rows ? n * 4 : 4);
    p.off_nexceptions = take(8);
    p.scratch_bytes = at;
    p.from_rows = from_rows;
    return p;
}

hipError_t launch_order_batch(const OrderPlan &p, const uint8_t *d_rows, const uint64_t *d_packed, void *d_scratch, hipStream_t stream,
                              const uint64_t **ordered, const uint32_t **out_index) {
    if (p.n == 0 || p.n > 0xFFFFFFFFull || p.nlevels < 1 || (p.from_rows ? d_rows == nullptr : d_packed == nullptr)) return hipErrorInvalidValue;
    char *s = static_cast<char *>(d_scratch);
    uint64_t *rows_packed = reinterpret_cast<uint64_t *>(s + p.off_packed_rows);
    uint64_t *pk[2] = {reinterpret_cast<uint64_t *>(s + p.off_packed[0]), reinterpret_cast<uint64_t *>(s + p.off_packed[1])};
    uint32_t *ix[2] = {reinterpret_cast<uint32_t *>(s + p.off_index[0]), reinterpret_cast<uint32_t *>(s + p.off_index[1])};
    uint32_t *hist = reinterpret_cast<uint32_t *>(s + p.off_hist), *totals = reinterpret_cast<uint32_t *>(s + p.off_totals);
    uint32_t *starts[2] = {reinterpret_cast<uint32_t *>(s + p.off_starts[0]), reinterpret_cast<uint32_t *>(s + p.off_starts[1])};
    uint32_t *exceptions = reinterpret_cast<uint32_t *>(s + p.off_exceptions);
    unsigned long long *nexc = reinterpret_cast<unsigned long long *>(s + p.off_nexceptions);
    const uint32_t b0 = p.level_bits[0], nb0 = 1u << b0;
    hipError_t e = hipMemsetAsync(nexc, 0, 8, stream);
    if (e != hipSuccess) return e;
    const uint64_t *src_packed = p.from_rows ? rows_packed : d_packed;
    if (p.words == 1)
        hipLaunchKernelGGL((k_order_pack<1>), dim3(p.nwg), dim3(kOrderThreads), nb0 * 4, stream, p.from_rows ? d_rows : nullptr, d_packed, rows_packed, p.k, p.n,
                           p.chunk, p.reach, b0, hist, exceptions, nexc);
    else
        hipLaunchKernelGGL((k_order_pack<2>), dim3(p.nwg), dim3(kOrderThreads), nb0 * 4, stream, p.from_rows ? d_rows : nullptr, d_packed, rows_packed, p.k, p.n,
                           p.chunk, p.reach, b0, hist, exceptions, nexc);
    hipLaunchKernelGGL(k_order_offsets, dim3(nb0), dim3(kOrderThreads), 0, stream, hist, p.nwg, nb0, totals);
    hipLaunchKernelGGL(k_order_starts, dim3(1), dim3(kOrderThreads), 0, stream, totals, nb0, starts[0]);
    if (p.words == 1)
        hipLaunchKernelGGL((k_order_scatter<1>), dim3(p.nwg), dim3(kOrderThreads), nb0 * 4, stream, src_packed, p.n, p.chunk, p.reach, b0, hist, starts[0], pk[0], ix[0]);
    else
        hipLaunchKernelGGL((k_order_scatter<2>), dim3(p.nwg), dim3(kOrderThreads), nb0 * 4, stream, src_packed, p.n, p.chunk, p.reach, b0, hist, starts[0], pk[0], ix[0]);
    int cur = 0;
    uint32_t drop = b0, nparents = nb0;
    for (uint32_t level = 1; level < p.nlevels; ++level) {
        const uint32_t bits = p.level_bits[level];
        const bool more = level + 1 < p.nlevels;
        const uint32_t *parents = starts[(level - 1) & 1];
        uint32_t *children = more ? starts[level & 1] : nullptr;
        const uint32_t grid = std::min<uint32_t>(nparents, 256u * 64u);
        if (p.words == 1)
            hipLaunchKernelGGL((k_order_level<1>), dim3(grid), dim3(kOrderThreads), (1u << bits) * 4, stream, pk[cur], ix[cur], parents, nparents, p.reach, drop, bits,
                               pk[cur ^ 1], ix[cur ^ 1], children);
        else
            hipLaunchKernelGGL((k_order_level<2>), dim3(grid), dim3(kOrderThreads), (1u << bits) * 4, stream, pk[cur], ix[cur], parents, nparents, p.reach, drop, bits,
                               pk[cur ^ 1], ix[cur ^ 1], children);
        cur ^= 1;
        drop += bits;
        nparents <<= bits;
    }
    *ordered = pk[cur];
    *out_index = ix[cur];
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
