// Builds the pair blocks (two symbols per search step; layout and maths: rank_ops.hpp) from the
// finished plane blocks, entirely on the device:
//
//   k_pair_paint      per plane block: for each ACGT symbol a the positions holding a map, in
//                     order, to the consecutive rows A[a], A[a]+1, ... (A[a] = the block's
//                     header value), so S2[i] = S[LF(i)] is read from a contiguous row range;
//                     every lane owns one word of each of the five planes of its pair block
//                     outright (plain stores, no atomics)
//   k_pair_tile_sums  16 pair counts per tile of 1024 pair blocks
//   k_pair_scan       exclusive prefix over tiles
//   k_pair_headers    24-bit per-block header fields, relative to the 2^24-position superblock
//   k_pair_super      superblock table K[a][b] + occ2(a, b, superblock start)
#include <hip/hip_runtime.h>

#include "pair_index.hpp"
#include "rank_ops.hpp"

namespace msbwt {
namespace {

constexpr int kTilePairBlocks = 1024;  // pair blocks per tile; 128 tiles = one superblock (2^17 blocks)
constexpr int kThreads = 256;
constexpr int kBlocksPerThread = kTilePairBlocks / kThreads;  // 4

struct Sixteen {
    uint64_t v[16];
};

struct StartIndex {
    uint64_t c[6];
};

// ---- K[a][b] = C[b] + occ(b, C[a]) via the single-step rank ---------------------------------
__global__ __launch_bounds__(128) void k_pair_consts(const uint4 *__restrict__ blocks, StartIndex start, uint64_t *__restrict__ K) {
    const uint32_t sub = threadIdx.x & 7u, p = threadIdx.x >> 3;  // 16 groups, one per pair
    const uint32_t a2 = p >> 2, b2 = p & 3u;
    const uint32_t a = a2 == 3u ? 5u : a2 + 1u, b = b2 == 3u ? 5u : b2 + 1u;
    const Range r = constrain(blocks, b, start.c[a], start.c[a], sub);
    if (sub == 0) K[p] = r.l;
}

// exclusive prefix of x over the lanes of an aligned 8-lane group
__device__ __forceinline__ uint32_t group_exclusive_scan(uint32_t x, uint32_t sub) {
    uint32_t inc = x;
    for (int d = 1; d < 8; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d, 8);
        if (int(sub) >= d) inc += y;
    }
    return inc - x;
}

__global__ __launch_bounds__(256) void k_pair_paint(const uint4 *__restrict__ blocks, uint64_t nblocks, uint4 *__restrict__ pair_blocks,
                                                    uint32_t stride96) {
    const uint32_t sub = threadIdx.x & 7u;
    const uint32_t lane = threadIdx.x & 63u, group_base = lane & ~7u;
    const uint64_t ngroups = (uint64_t(gridDim.x) * blockDim.x) / 8;
    for (uint64_t b = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) / 8; b < nblocks; b += ngroups) {
        const uint4 c = blocks[b * 8 + sub];
        // header words of the whole block: low words from chunks 1,2,3,5, high bytes from chunks 6,7
        const uint32_t hi03 = __shfl(c.w, int(group_base + 6)), hi45 = __shfl(c.w, int(group_base + 7));
        uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0, valid = 0;  // my 32 positions' bits of the five planes
        uint64_t cached_chunk = ~0ull;
        uint4 tc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t a2 = 0; a2 < 4; ++a2) {
            const uint32_t a = a2 == 3u ? 5u : a2 + 1u;
            const uint32_t lo = __shfl(c.w, int(group_base + a));
            const uint32_t hi = a < 4u ? (hi03 >> (8u * a)) & 0xFFu : (hi45 >> (8u * (a - 4u))) & 0xFFu;
            const uint64_t A = (uint64_t(hi) << 32) | lo;  // row of the block's first a
            const uint32_t x0 = (a & 1u) ? 0u : ~0u, x1 = (a & 2u) ? 0u : ~0u, x2 = (a & 4u) ? 0u : ~0u;
            uint32_t mask = (c.x ^ x0) & (c.y ^ x1) & (c.z ^ x2);
            uint64_t target = A + group_exclusive_scan(uint32_t(__popc(mask)), sub);
            while (mask) {
                const uint32_t q = uint32_t(__ffs(int(mask))) - 1u;  // position in my 32
                mask &= mask - 1u;
                const uint64_t chunk = target >> 5;                 // plane chunk that holds row `target`
                if (chunk != cached_chunk) {
                    tc = blocks[chunk];
                    cached_chunk = chunk;
                }
                const uint32_t bit = uint32_t(target) & 31u;
                const uint32_t s2 = ((tc.x >> bit) & 1u) | (((tc.y >> bit) & 1u) << 1) | (((tc.z >> bit) & 1u) << 2);
                if (is_acgt(s2)) {
                    const uint32_t b2 = acgt_code(s2);
                    a0 |= (a2 & 1u) << q;
                    a1 |= (a2 >> 1) << q;
                    b0 |= (b2 & 1u) << q;
                    b1 |= (b2 >> 1) << q;
                    valid |= 1u << q;
                }
                ++target;
            }
        }
        // my 32 positions are 32-position word W = 8 b + sub of the BWT.  stride 128: word W & 3 of pair
        // block W >> 2.  stride 96: word W % 3 of block W / 3 -- and, when that is word 0, also word 3 of
        // the block before (its look-ahead).
        const uint64_t W = b * 8 + sub;
        const uint64_t pb = stride96 ? W / 3u : W >> 2;
        const uint32_t slot = stride96 ? uint32_t(W - pb * 3u) : uint32_t(W & 3u);
        uint32_t *out = reinterpret_cast<uint32_t *>(pair_blocks + pb * 8) + slot;
        out[0] = a0;
        out[4] = a1;
        out[8] = b0;
        out[12] = b1;
        out[16] = valid;
        if (stride96 && slot == 0u && pb > 0) {
            uint32_t *prev = reinterpret_cast<uint32_t *>(pair_blocks + (pb - 1) * 8) + 3;
            prev[0] = a0;
            prev[4] = a1;
            prev[8] = b0;
            prev[12] = b1;
            prev[16] = valid;
        }
    }
}

// the 16 pair counts of one pair block's OWN positions (its first `words` 32-position words: 4, or 3
// when blocks overlap), added into acc[]
__device__ __forceinline__ void add_block_pair_counts(const uint4 *__restrict__ blk, uint32_t acc[16], int words) {
    const uint4 a0 = blk[0], a1 = blk[1], b0 = blk[2], b1 = blk[3], v = blk[kPairValidChunk];
    const uint32_t A0[4] = {a0.x, a0.y, a0.z, a0.w}, A1[4] = {a1.x, a1.y, a1.z, a1.w};
    const uint32_t B0[4] = {b0.x, b0.y, b0.z, b0.w}, B1[4] = {b1.x, b1.y, b1.z, b1.w};
    const uint32_t V[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w >= words) break;
        uint32_t ia[4], ib[4];  // positions whose first / second symbol is A C G T
        ia[0] = ~A0[w] & ~A1[w] & V[w]; ia[1] = A0[w] & ~A1[w] & V[w]; ia[2] = ~A0[w] & A1[w] & V[w]; ia[3] = A0[w] & A1[w] & V[w];
        ib[0] = ~B0[w] & ~B1[w]; ib[1] = B0[w] & ~B1[w]; ib[2] = ~B0[w] & B1[w]; ib[3] = B0[w] & B1[w];
#pragma unroll
        for (uint32_t p = 0; p < 16; ++p) acc[p] += uint32_t(__popc(ia[p >> 2] & ib[p & 3u]));
    }
}

__global__ __launch_bounds__(kThreads) void k_pair_tile_sums(const uint4 *__restrict__ pair_blocks, uint64_t npair,
                                                             Sixteen *__restrict__ tiles, int words) {
    __shared__ uint32_t red[16][kThreads / 64];
    for (uint64_t tile = blockIdx.x; tile * kTilePairBlocks < npair; tile += gridDim.x) {
        uint32_t acc[16] = {0};
        for (int i = 0; i < kBlocksPerThread; ++i) {
            const uint64_t pb = tile * kTilePairBlocks + uint64_t(threadIdx.x) * kBlocksPerThread + i;
            if (pb < npair) add_block_pair_counts(pair_blocks + pb * 8, acc, words);
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            uint32_t x = acc[p];
            for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d);
            if ((threadIdx.x & 63) == 0) red[p][threadIdx.x >> 6] = x;
        }
        __syncthreads();
        if (threadIdx.x < 16) {
            uint64_t s = 0;
            for (int w = 0; w < kThreads / 64; ++w) s += red[threadIdx.x][w];
            tiles[tile].v[threadIdx.x] = s;
        }
    }
}

// in-place exclusive scan of the 16-vectors over tiles; one workgroup of 1024
__global__ __launch_bounds__(1024) void k_pair_scan(Sixteen *__restrict__ tiles, uint64_t ntiles) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry[16];
    if (threadIdx.x < 16) carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t base = 0; base < ntiles; base += 1024) {
        const uint64_t t = base + threadIdx.x;
        for (int p = 0; p < 16; ++p) {
            const uint64_t x = t < ntiles ? tiles[t].v[p] : 0;
            uint64_t inc = x;
            for (int d = 1; d < 64; d <<= 1) {
                const uint64_t y = __shfl_up(inc, d);
                if (lane >= d) inc += y;
            }
            if (lane == 63) wave_sum[wave] = inc;
            __syncthreads();
            uint64_t before = carry[p], all = 0;
            for (int w = 0; w < 16; ++w) {
                if (w < wave) before += wave_sum[w];
                all += wave_sum[w];
            }
            if (t < ntiles) tiles[t].v[p] = before + inc - x;
            __syncthreads();
            if (threadIdx.x == 0) carry[p] += all;
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_pair_headers(uint4 *__restrict__ pair_blocks, uint64_t npair,
                                                           const Sixteen *__restrict__ tiles, int words) {
    __shared__ uint32_t wave_tot[kThreads / 64];
    for (uint64_t tile = blockIdx.x; tile * kTilePairBlocks < npair; tile += gridDim.x) {
        const uint64_t first = tile * kTilePairBlocks + uint64_t(threadIdx.x) * kBlocksPerThread;
        // pass A: my blocks' totals -> where my first block starts inside the tile
        uint32_t mine[16] = {0};
        for (int i = 0; i < kBlocksPerThread; ++i)
            if (first + i < npair) add_block_pair_counts(pair_blocks + (first + i) * 8, mine, words);
        uint32_t run[16];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            uint32_t inc = mine[p];
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(inc, d);
                if (lane >= d) inc += y;
            }
            __syncthreads();
            if (lane == 63) wave_tot[wave] = inc;
            __syncthreads();
            uint32_t before = 0;
            for (int w = 0; w < wave; ++w) before += wave_tot[w];
            run[p] = before + inc - mine[p];
        }
        // pass B: header fields = counts before the block, relative to its superblock start
        const uint64_t sb_tile = tile & ~uint64_t(127);  // 128 tiles of 1024 blocks = one superblock
        uint32_t rel[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) rel[p] = uint32_t(tiles[tile].v[p] - tiles[sb_tile].v[p]) + run[p];
        for (int i = 0; i < kBlocksPerThread; ++i) {
            const uint64_t pb = first + i;
            if (pb >= npair) break;
            uint4 *blk = pair_blocks + pb * 8;
            uint32_t add[16] = {0};
            add_block_pair_counts(blk, add, words);
            uint32_t lo[8], hi[4] = {0, 0, 0, 0};  // 16 x u16 low halves, 16 x u8 high bytes
#pragma unroll
            for (int j = 0; j < 8; ++j) lo[j] = (rel[2 * j] & 0xFFFFu) | ((rel[2 * j + 1] & 0xFFFFu) << 16);
#pragma unroll
            for (int p = 0; p < 16; ++p) hi[p >> 2] |= ((rel[p] >> 16) & 0xFFu) << (8 * (p & 3));
            blk[kPairLoChunk] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            blk[kPairLoChunk + 1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
            blk[kPairHiChunk] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
#pragma unroll
            for (int p = 0; p < 16; ++p) rel[p] += add[p];
        }
    }
}

__global__ void k_pair_super(const Sixteen *__restrict__ tiles, uint64_t ntiles, const uint64_t *__restrict__ K,
                             uint64_t nsuper, uint64_t *__restrict__ super) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nsuper * 16) return;
    const uint64_t sb = i >> 4, p = i & 15u, tile = sb * 128;
    super[i] = K[p] + (tile < ntiles ? tiles[tile].v[p] : 0);
}

}  // namespace

PairIndexSizes pair_index_sizes(uint64_t nblocks, int stride) {
    PairIndexSizes s;
    s.pair_blocks = stride == 96 ? (nblocks * 256) / 96 + 2 : 2 * nblocks;
    s.tiles = (s.pair_blocks + kTilePairBlocks - 1) / kTilePairBlocks;
    s.supers = (s.tiles + 127) / 128;
    s.pair_block_bytes = size_t(s.pair_blocks) * 128;
    s.super_bytes = size_t(s.supers) * 16 * sizeof(uint64_t);
    s.scratch_bytes = size_t(s.tiles) * sizeof(Sixteen) + 16 * sizeof(uint64_t);
    return s;
}

hipError_t build_pair_index(const void *d_blocks, uint64_t nblocks, const uint64_t start_index[6], void *d_pair_blocks,
                            void *d_super, void *d_scratch, hipStream_t stream, int stride) {
    const PairIndexSizes sz = pair_index_sizes(nblocks, stride);
    const uint32_t stride96 = stride == 96 ? 1u : 0u;
    const int words = stride96 ? 3 : 4;
    // with overlapping blocks not every (block, word) is painted (the tail): start from zeros
    hipError_t zeroed = hipMemsetAsync(d_pair_blocks, 0, sz.pair_block_bytes, stream);
    if (zeroed != hipSuccess) return zeroed;
    const uint4 *blocks = static_cast<const uint4 *>(d_blocks);
    uint4 *pair = static_cast<uint4 *>(d_pair_blocks);
    uint64_t *K = static_cast<uint64_t *>(d_scratch);
    Sixteen *tiles = reinterpret_cast<Sixteen *>(K + 16);
    StartIndex st;
    for (int s = 0; s < 6; ++s) st.c[s] = start_index[s];
    hipLaunchKernelGGL(k_pair_consts, dim3(1), dim3(128), 0, stream, blocks, st, K);
    const uint64_t groups_blocks = (nblocks * 8 + 255) / 256;
    hipLaunchKernelGGL(k_pair_paint, dim3(uint32_t(groups_blocks > 8192 ? 8192 : (groups_blocks ? groups_blocks : 1))), dim3(256), 0,
                       stream, blocks, nblocks, pair, stride96);
    const uint32_t tgrid = uint32_t(sz.tiles > 4096 ? 4096 : (sz.tiles ? sz.tiles : 1));
    hipLaunchKernelGGL(k_pair_tile_sums, dim3(tgrid), dim3(kThreads), 0, stream, pair, sz.pair_blocks, tiles, words);
    hipLaunchKernelGGL(k_pair_scan, dim3(1), dim3(1024), 0, stream, tiles, sz.tiles);
    hipLaunchKernelGGL(k_pair_headers, dim3(tgrid), dim3(kThreads), 0, stream, pair, sz.pair_blocks, tiles, words);
    hipLaunchKernelGGL(k_pair_super, dim3(uint32_t((sz.supers * 16 + 255) / 256)), dim3(256), 0, stream, tiles, sz.tiles, K,
                       sz.supers, static_cast<uint64_t *>(d_super));
    return hipGetLastError();
}

}  // namespace msbwt
