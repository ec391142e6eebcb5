// Device-side index build (load path): RLE bytes in HBM -> plane blocks, without ever holding
// the expanded index on the host.  Three kernels and a small scan:
//
//   k_tile_sums   every 4 KiB tile of RLE bytes -> {symbols, per-symbol counts} it encodes
//   k_scan_tiles  exclusive prefix over the tiles (7 x u64 per tile)
//   k_paint       every byte is one "sub-run" (digit << 5*index-in-its-run symbols of one
//                 code): a block-level scan gives its BWT position and the symbol counts
//                 before it; the thread ORs its bits into the planes (atomicOr: neighbouring
//                 sub-runs share words) and writes the header of every block whose first
//                 position it covers.  Sub-runs of >= 2048 symbols go to a list ...
//   k_paint_long  ... and are filled by one workgroup each with plain stores.
//
// This is what breaks the carry chain of the on-disk format (the weight of a byte depends on
// how many bytes of the same symbol precede it, reference src/rle_bwt.rs:360-371): the
// exponent is found by looking back at most 12 bytes, everything else is a prefix sum.
#include <hip/hip_runtime.h>

#include "device_build.hpp"
#include "plane_index.hpp"

namespace msbwt {
namespace {

constexpr int kThreads = 256;
constexpr int kBytesPerThread = 16;
constexpr int kTileBytes = kThreads * kBytesPerThread;  // 4096
constexpr uint64_t kLongRun = 2048;                      // symbols; longer sub-runs are deferred
constexpr int kMaxDigits = 8;                            // 32^8 = 2^40: more digits cannot fit T < 2^40

struct Seven {
    uint64_t v[7];  // [0..5] per-symbol counts, [6] all symbols
};

struct LongRun {
    uint64_t pos, len;
    uint64_t occ[6];  // symbol counts before pos
    uint32_t sym, pad;
};

__device__ __forceinline__ uint64_t wave_inclusive_scan(uint64_t x) {
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    return x;
}

// Exclusive scan of one u64 per thread across the 256-thread block; *block_total gets the sum.
__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t x, uint64_t *lds4, uint64_t *block_total) {
    const uint64_t inc = wave_inclusive_scan(x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // lds4 may still be read from the previous call
    if (lane == 63) lds4[wave] = inc;
    __syncthreads();
    uint64_t base = 0, total = 0;
    for (int w = 0; w < kThreads / 64; ++w) {
        const uint64_t t = lds4[w];
        if (w < wave) base += t;
        total += t;
    }
    *block_total = total;
    return base + inc - x;
}

// The 16 bytes a thread owns (as four dwords) and how their first byte continues a run that
// started earlier: `carry` = number of bytes right before them with the symbol of byte 0.
struct ThreadBytes {
    uint32_t w[4];
    int valid;  // how many of the 16 bytes exist
    int carry;
};

__device__ __forceinline__ uint32_t byte_at(const ThreadBytes &tb, int i) {
    const uint32_t lo = (i & 4) ? tb.w[1] : tb.w[0], hi = (i & 4) ? tb.w[3] : tb.w[2];
    return (((i & 8) ? hi : lo) >> ((i & 3) * 8)) & 0xFFu;
}

__device__ __forceinline__ void load_thread_bytes(const uint8_t *__restrict__ rle, uint64_t n, uint64_t first,
                                                  ThreadBytes *tb) {
    // rle is 16-byte aligned and first is a multiple of 16; never read past n
    tb->valid = first >= n ? 0 : int(min(uint64_t(16), n - first));
    tb->w[0] = tb->w[1] = tb->w[2] = tb->w[3] = 0;
    tb->carry = 0;
    if (tb->valid == 16) {
        const uint4 c = *reinterpret_cast<const uint4 *>(rle + first);
        tb->w[0] = c.x; tb->w[1] = c.y; tb->w[2] = c.z; tb->w[3] = c.w;
    } else {
        for (int i = 0; i < tb->valid; ++i) {
            const uint32_t b = uint32_t(rle[first + i]) << ((i & 3) * 8);
            if (i < 4) tb->w[0] |= b; else if (i < 8) tb->w[1] |= b; else if (i < 12) tb->w[2] |= b; else tb->w[3] |= b;
        }
    }
    if (first >= 16 && tb->valid > 0) {
        const uint4 p = *reinterpret_cast<const uint4 *>(rle + first - 16);
        const uint32_t pw[4] = {p.x, p.y, p.z, p.w};
        const uint32_t sym0 = tb->w[0] & 7u;
        bool run = true;
#pragma unroll
        for (int j = 15; j >= 0; --j) {
            run = run && (((pw[j >> 2] >> ((j & 3) * 8)) & 7u) == sym0);
            tb->carry += run ? 1 : 0;
        }
    }
}

// Calls fn(sym, value) for each of the thread's sub-runs in order: value = digit << 5 * (index
// of the byte inside its run), the index being a recurrence over the bytes.  Returns error bits.
template <class Fn>
__device__ __forceinline__ uint32_t for_each_subrun(const ThreadBytes &tb, Fn &&fn) {
    uint32_t bad = 0, prev_sym = 8;
    int e = tb.carry;
    for (int i = 0; i < tb.valid; ++i) {
        const uint32_t byte = byte_at(tb, i), sym = byte & 7u, digit = byte >> 3;
        if (i > 0) e = (sym == prev_sym) ? e + 1 : 0;
        prev_sym = sym;
        if (sym >= 6u) bad |= kBuildBadSymbol;
        if (e >= kMaxDigits && digit) bad |= kBuildTooLarge;
        fn(sym, e < kMaxDigits ? (uint64_t(digit) << (5 * e)) : uint64_t(0));
    }
    return bad;
}

// acc[sym] += v without a runtime-indexed register array
__device__ __forceinline__ void add_to_symbol(uint64_t acc[7], uint32_t sym, uint64_t v) {
#pragma unroll
    for (int s = 0; s < 6; ++s) acc[s] += (uint32_t(s) == sym) ? v : 0;
}

__global__ __launch_bounds__(kThreads) void k_tile_sums(const uint8_t *__restrict__ rle, uint64_t n,
                                                        Seven *__restrict__ tile_sums, uint32_t *__restrict__ flags,
                                                        unsigned long long *__restrict__ long_count) {
    __shared__ uint64_t red[7][kThreads / 64];
    for (uint64_t tile = blockIdx.x; tile * kTileBytes < n; tile += gridDim.x) {
        const uint64_t first = tile * kTileBytes + uint64_t(threadIdx.x) * kBytesPerThread;
        ThreadBytes tb;
        load_thread_bytes(rle, n, first, &tb);
        uint64_t acc[7] = {0, 0, 0, 0, 0, 0, 0};
        uint32_t longs = 0;
        const uint32_t bad = for_each_subrun(tb, [&](uint32_t sym, uint64_t v) {
            add_to_symbol(acc, sym, v);
            acc[6] += v;
            longs += (v >= kLongRun) ? 1u : 0u;
        });
        if (bad) atomicOr(flags, bad);
        if (longs) atomicAdd(long_count, (unsigned long long)longs);
        // block reduction of the 7 sums
        __syncthreads();
        for (int k = 0; k < 7; ++k) {
            uint64_t x = acc[k];
            for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d);
            if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = x;
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            uint64_t s = 0;
            for (int w = 0; w < kThreads / 64; ++w) s += red[threadIdx.x][w];
            tile_sums[tile].v[threadIdx.x] = s;
        }
    }
}

// In-place exclusive scan over the tiles; totals[0..6] gets the grand totals.  One workgroup.
__global__ __launch_bounds__(1024) void k_scan_tiles(Seven *__restrict__ tiles, uint64_t ntiles, uint64_t *__restrict__ totals) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry[7];
    if (threadIdx.x < 7) carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t base = 0; base < ntiles; base += 1024) {
        const uint64_t t = base + threadIdx.x;
        for (int k = 0; k < 7; ++k) {
            const uint64_t x = t < ntiles ? tiles[t].v[k] : 0;
            const uint64_t inc = wave_inclusive_scan(x);
            if (lane == 63) wave_sum[wave] = inc;
            __syncthreads();
            uint64_t before = carry[k], all = 0;
            for (int w = 0; w < 16; ++w) {
                if (w < wave) before += wave_sum[w];
                all += wave_sum[w];
            }
            if (t < ntiles) tiles[t].v[k] = before + inc - x;
            __syncthreads();
            if (threadIdx.x == 0) carry[k] += all;
            __syncthreads();
        }
    }
    if (threadIdx.x < 7) totals[threadIdx.x] = carry[threadIdx.x];
}

// Header words of block `b` from the bounds A[s] (layout: plane_index.hpp)
__device__ __forceinline__ void store_meta(uint32_t *__restrict__ blocks, uint64_t b, const uint64_t A[6]) {
    uint32_t *blk = blocks + b * 32;
    uint32_t hi_a = 0, hi_b = 0;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        blk[4 * s + 3] = uint32_t(A[s]);
        const uint32_t hi = uint32_t(A[s] >> 32) & 0xFFu;
        if (s < 4) hi_a |= hi << (8 * s);
        else hi_b |= hi << (8 * (s - 4));
    }
    blk[4 * 6 + 3] = hi_a;
    blk[4 * 7 + 3] = hi_b;
}

// OR symbol `sym` into positions [pos, pos+len) of the planes, word by word.
__device__ __forceinline__ void paint_atomic(uint32_t *__restrict__ blocks, uint32_t sym, uint64_t pos, uint64_t len) {
    const uint64_t end = pos + len;
    for (uint64_t w = pos >> 5; w <= (end - 1) >> 5; ++w) {
        const uint32_t lo = uint32_t(max(pos, w << 5) - (w << 5));
        const uint32_t hi = uint32_t(min(end, (w << 5) + 32) - (w << 5));  // 1..32
        const uint32_t mask = (hi == 32u ? ~0u : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        uint32_t *word = blocks + (w >> 3) * 32 + (w & 7u) * 4;
        if (sym & 1u) atomicOr(word + 0, mask);
        if (sym & 2u) atomicOr(word + 1, mask);
        if (sym & 4u) atomicOr(word + 2, mask);
    }
}

// Headers of every block whose first position lies in [pos, pos+len): occ[] = counts before pos.
__device__ __forceinline__ void write_headers(uint32_t *__restrict__ blocks, const uint64_t *__restrict__ start_index,
                                              uint32_t sym, uint64_t pos, uint64_t len, const uint64_t occ[6]) {
    for (uint64_t B = (pos + 255) & ~uint64_t(255); B < pos + len; B += 256) {
        uint64_t A[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) A[s] = start_index[s] + occ[s] + (uint32_t(s) == sym ? B - pos : 0);
        store_meta(blocks, B >> 8, A);
    }
}

__global__ __launch_bounds__(kThreads) void k_paint(const uint8_t *__restrict__ rle, uint64_t n,
                                                    const Seven *__restrict__ tile_base,
                                                    const uint64_t *__restrict__ start_index,
                                                    uint32_t *__restrict__ blocks, LongRun *__restrict__ long_runs,
                                                    unsigned long long *__restrict__ long_cursor) {
    __shared__ uint64_t lds4[kThreads / 64];
    for (uint64_t tile = blockIdx.x; tile * kTileBytes < n; tile += gridDim.x) {
        const uint64_t first = tile * kTileBytes + uint64_t(threadIdx.x) * kBytesPerThread;
        ThreadBytes tb;
        load_thread_bytes(rle, n, first, &tb);
        uint64_t mine[7] = {0, 0, 0, 0, 0, 0, 0};
        (void)for_each_subrun(tb, [&](uint32_t sym, uint64_t v) {
            add_to_symbol(mine, sym, v);
            mine[6] += v;
        });
        // position and per-symbol counts at this thread's first byte
        uint64_t at[7], dummy;
        for (int k = 0; k < 7; ++k) at[k] = tile_base[tile].v[k] + block_exclusive_scan(mine[k], lds4, &dummy);
        (void)for_each_subrun(tb, [&](uint32_t sym, uint64_t v) {
            if (v == 0) return;
            const uint64_t pos = at[6];
            if (v >= kLongRun) {
                const unsigned long long slot = atomicAdd(long_cursor, 1ull);
                LongRun lr;
                lr.pos = pos;
                lr.len = v;
                for (int s = 0; s < 6; ++s) lr.occ[s] = at[s];
                lr.sym = sym;
                lr.pad = 0;
                long_runs[slot] = lr;
            } else {
                if (sym) paint_atomic(blocks, sym, pos, v);
                write_headers(blocks, start_index, sym, pos, v, at);
            }
            add_to_symbol(at, sym, v);
            at[6] += v;
        });
    }
}

// One workgroup per long sub-run: partial words at both ends by atomicOr, whole words in
// between by plain stores (nothing else writes them), headers for every covered block.
__global__ __launch_bounds__(kThreads) void k_paint_long(const LongRun *__restrict__ long_runs, uint64_t nlong,
                                                         const uint64_t *__restrict__ start_index,
                                                         uint32_t *__restrict__ blocks) {
    for (uint64_t item = blockIdx.x; item < nlong; item += gridDim.x) {
        const LongRun lr = long_runs[item];
        const uint64_t pos = lr.pos, end = lr.pos + lr.len;
        const uint64_t first_full = (pos + 31) >> 5, last_full = end >> 5;  // words [first_full, last_full) are whole
        if (lr.sym) {
            if (threadIdx.x == 0 && (pos & 31u)) paint_atomic(blocks, lr.sym, pos, min(end, first_full << 5) - pos);
            if (threadIdx.x == 1 && (end & 31u) && (last_full >= first_full)) paint_atomic(blocks, lr.sym, last_full << 5, end - (last_full << 5));
            for (uint64_t w = first_full + threadIdx.x; w < last_full; w += kThreads) {
                uint32_t *word = blocks + (w >> 3) * 32 + (w & 7u) * 4;
                if (lr.sym & 1u) word[0] = ~0u;
                if (lr.sym & 2u) word[1] = ~0u;
                if (lr.sym & 4u) word[2] = ~0u;
            }
        }
        const uint64_t B0 = (pos + 255) & ~uint64_t(255);
        for (uint64_t B = B0 + uint64_t(threadIdx.x) * 256; B < end; B += uint64_t(kThreads) * 256) {
            uint64_t A[6];
#pragma unroll
            for (int s = 0; s < 6; ++s) A[s] = start_index[s] + lr.occ[s] + (uint32_t(s) == lr.sym ? B - pos : 0);
            store_meta(blocks, B >> 8, A);
        }
    }
}

// Header of the block that starts exactly at T (no sub-run covers that position).
__global__ void k_final_header(const uint64_t *__restrict__ start_index, const uint64_t *__restrict__ totals,
                               uint32_t *__restrict__ blocks) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && (totals[6] & 255u) == 0) {
        uint64_t A[6];
        for (int s = 0; s < 6; ++s) A[s] = start_index[s] + totals[s];
        store_meta(blocks, totals[6] >> 8, A);
    }
}

inline uint32_t grid_for_tiles(uint64_t ntiles) { return uint32_t(ntiles < 1 ? 1 : (ntiles > 4096 ? 4096 : ntiles)); }

}  // namespace

size_t device_build_scratch_bytes(size_t n) {
    const uint64_t ntiles = (n + kTileBytes - 1) / kTileBytes;
    return size_t(ntiles + 1) * sizeof(Seven) + 256;
}

hipError_t device_build_pass1(const uint8_t *d_rle, size_t n, void *d_scratch, DeviceBuildState *st, hipStream_t stream) {
    // scratch: [totals 7 x u64 | start_index 6 x u64 | flags u32 | long_count u64 | long_cursor u64] + tile sums
    uint8_t *base = static_cast<uint8_t *>(d_scratch);
    st->d_totals = reinterpret_cast<uint64_t *>(base);
    st->d_start_index = st->d_totals + 7;
    st->d_flags = reinterpret_cast<uint32_t *>(st->d_start_index + 6);
    st->d_long_count = reinterpret_cast<unsigned long long *>(st->d_start_index + 8);
    st->d_long_cursor = st->d_long_count + 1;
    st->d_tiles = base + 256;
    st->ntiles = (n + kTileBytes - 1) / kTileBytes;
    hipError_t e = hipMemsetAsync(base, 0, 256, stream);
    if (e != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tile_sums, dim3(grid_for_tiles(st->ntiles)), dim3(kThreads), 0, stream, d_rle, uint64_t(n),
                       static_cast<Seven *>(st->d_tiles), st->d_flags, st->d_long_count);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, stream, static_cast<Seven *>(st->d_tiles), st->ntiles, st->d_totals);
    return hipGetLastError();
}

hipError_t device_build_pass2(const uint8_t *d_rle, size_t n, const DeviceBuildState &st, void *d_long_runs,
                              uint64_t nlong, void *d_blocks, hipStream_t stream) {
    uint32_t *blocks = static_cast<uint32_t *>(d_blocks);
    if (n) {
        hipLaunchKernelGGL(k_paint, dim3(grid_for_tiles(st.ntiles)), dim3(kThreads), 0, stream, d_rle, uint64_t(n),
                           static_cast<const Seven *>(st.d_tiles), st.d_start_index, blocks,
                           static_cast<LongRun *>(d_long_runs), st.d_long_cursor);
        if (nlong)
            hipLaunchKernelGGL(k_paint_long, dim3(uint32_t(nlong > 2048 ? 2048 : nlong)), dim3(kThreads), 0, stream,
                               static_cast<const LongRun *>(d_long_runs), nlong, st.d_start_index, blocks);
    }
    hipLaunchKernelGGL(k_final_header, dim3(1), dim3(64), 0, stream, st.d_start_index, st.d_totals, blocks);
    return hipGetLastError();
}

size_t device_build_long_run_bytes(uint64_t nlong) { return size_t(nlong ? nlong : 1) * sizeof(LongRun); }

}  // namespace msbwt
