// gfx950 builder of the sparse suffix table (sparse_table.hpp): frontier expansion with the index's own rank code.
//
// A node is a suffix that occurs: {key, l, h}.  The seeds are the non-empty entries of the flat direct table (depth p); one
// expansion takes every node two symbols further by ONE pair step -- all 16 children of a node come from the same one or
// two pair-block lines (a line holds the counts of all 16 pairs), so a level costs about one random line per node, not 32
// ranks -- or, for the odd last level, one symbol further from its plane-block lines.  Children are appended to the next
// frontier in whatever order the blocks finish (the table is hashed: order means nothing), one atomic per workgroup.  The
// last level does not materialise its children: the sizing pass counts them, the fill pass puts each into its bucket
// (an atomic slot counter per bucket; a full bucket sends the entry on to the next one).  The parents are worked through
// in chunks that keep the frontiers inside a fixed scratch allocation; the sizing pass finds the chunking, the fill pass
// replays it.  Integer work bound by random lines: no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rank_ops.hpp"
#include "sparse_build.hpp"

namespace msbwt {
namespace {

struct Node {
    uint64_t key, l, h;
};

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
// cursor block (u64 words) at the start of the scratch
constexpr int kCurLevel = 0;    // [0 .. 32): nodes appended to the frontier of depth d (reset per chunk)
constexpr int kAccLevel = 32;   // [32 .. 64): the same summed over all chunks that completed
constexpr int kAccEscape = 64;  // [64 .. 96): nodes 255 or more wide, per depth
constexpr int kCurEscape = 96;  // [96 .. 128): the same for the chunk in hand
constexpr int kOverflow = 128, kFailed = 129, kSideCursor = 130, kDisplaced = 131, kFiltered = 132;
constexpr int kCurSingle = 160;  // [160 .. 192): nodes exactly 1 wide (suffixes that occur once), per depth, of the chunk in hand
constexpr int kAccSingle = 192;  // [192 .. 224): the same summed over all chunks that completed
constexpr int kCursorWords = 224;
static_assert(kSparseMaxDepth < 32, "a cursor per depth the expansion can reach");

// this thread's first of `mine` consecutive slots behind *cursor: one atomic per workgroup.  Every thread of the block calls it.
__device__ __forceinline__ uint64_t reserve(uint32_t mine, unsigned long long *cursor) {
    __shared__ uint32_t wave_total[kWaves];
    __shared__ unsigned long long block_base;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = uint32_t(__shfl_up(int(inc), d));
        if (int(lane) >= d) inc += y;
    }
    if (lane == 63u) wave_total[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t x = wave_total[w];
            wave_total[w] = t;
            t += x;
        }
        block_base = t ? atomicAdd(cursor, (unsigned long long)t) : 0ull;
    }
    __syncthreads();
    const uint64_t first = block_base + wave_total[wave] + (inc - mine);
    __syncthreads();  // (the shared words are rewritten by the next call)
    return first;
}

// adds the block's sum of `mine` to *acc
__device__ __forceinline__ void block_add(uint32_t mine, unsigned long long *acc) {
    __shared__ uint32_t part[kWaves];
    uint32_t s = mine;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += uint32_t(__shfl_xor(int(s), d));
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kWaves; ++w) t += part[w];
        if (t) atomicAdd(acc, (unsigned long long)t);
    }
    __syncthreads();
}

// ---- seeds: the non-empty entries [i0, i0 + np) of the flat table (or the root) ------------------------------------------
__global__ __launch_bounds__(kThreads) void k_sparse_seed(const uint4 *__restrict__ flat, uint64_t i0, uint64_t np, uint64_t total, Node *__restrict__ out,
                                                          uint64_t cap, unsigned long long *__restrict__ cur, uint32_t depth) {
    for (uint64_t base = uint64_t(blockIdx.x) * kThreads; base < np; base += uint64_t(gridDim.x) * kThreads) {
        const uint64_t i = base + threadIdx.x;
        Node nd{0, 0, 0};
        bool keep = false;
        if (i < np) {
            if (flat == nullptr) {
                nd = Node{0, 0, total};
            } else {
                const uint4 e = flat[i0 + i];
                nd = Node{i0 + i, (uint64_t(e.y) << 32) | e.x, (uint64_t(e.w) << 32) | e.z};
            }
            keep = nd.l != nd.h;
        }
        const uint64_t at = reserve(keep ? 1u : 0u, cur + kCurLevel + depth);
        if (keep) {
            if (at < cap) out[at] = nd;
            else atomicOr(cur + kOverflow, 1ull);
        }
    }
}

// ---- what one pair-block line says about a position: the 16 pair counts before it, relative to the superblock ---------------
struct PairSixteen {
    uint32_t rel[16];  // header field + matches among the block's first r positions
};

__device__ __forceinline__ void pair_line_sixteen(const uint4 *__restrict__ blk, uint32_t r, PairSixteen &out) {
    const uint4 a0 = blk[0], a1 = blk[1], b0 = blk[2], b1 = blk[3], v = blk[kPairValidChunk];
    const uint4 h5 = blk[kPairLoChunk], h6 = blk[kPairLoChunk + 1], h7 = blk[kPairHiChunk];
    const uint32_t A0[4] = {a0.x, a0.y, a0.z, a0.w}, A1[4] = {a1.x, a1.y, a1.z, a1.w}, B0[4] = {b0.x, b0.y, b0.z, b0.w},
                   B1[4] = {b1.x, b1.y, b1.z, b1.w}, V[4] = {v.x, v.y, v.z, v.w};
    const uint32_t lo16[8] = {h5.x, h5.y, h5.z, h5.w, h6.x, h6.y, h6.z, h6.w}, hi8[4] = {h7.x, h7.y, h7.z, h7.w};
    uint32_t am[4][4], bm[4][4];  // [code][word]: positions (among the first r) whose S / S2 is that code
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t low = low_bits(min(max(int(r) - 32 * w, 0), 32));
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) {
            const uint32_t n0 = (c & 1u) - 1u, n1 = ((c >> 1) & 1u) - 1u;
            am[c][w] = (A0[w] ^ n0) & (A1[w] ^ n1) & V[w] & low;
            bm[c][w] = (B0[w] ^ n0) & (B1[w] ^ n1);
        }
    }
#pragma unroll
    for (uint32_t p = 0; p < 16; ++p) {
        const uint32_t a = p >> 2, b = p & 3u;
        uint32_t cnt = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) cnt += uint32_t(__popc(am[a][w] & bm[b][w]));
        const uint32_t field = ((lo16[p >> 1] >> ((p & 1u) * 16u)) & 0xFFFFu) | (((hi8[p >> 2] >> ((p & 3u) * 8u)) & 0xFFu) << 16);
        out.rel[p] = field + cnt;
    }
}

struct FillEnv {
    uint4 *lines;        // the table (zeroed); nullptr = sizing pass
    uint32_t *counts;    // slot counter per bucket
    uint4 *side;
    uint32_t nbuckets, probe, n;  // n = 2 depth
    uint64_t nside;      // entries the side array holds
    uint32_t tier;       // two-tier form: ranges 1 wide set filter bits in their own bucket instead of taking an entry
};

// puts the entry of (key, [nl, nh)) into its bucket; `cur`: the cursor block (side cursor, displaced, failed)
__device__ __forceinline__ void sparse_insert(const FillEnv &env, uint64_t key, uint64_t nl, uint64_t nh, unsigned long long *cur) {
    const uint64_t x = sparse_mix(key, env.n);
    uint32_t b = sparse_bucket(x, env.n, env.nbuckets);
    const uint32_t depth = env.n >> 1;
    const bool wide = sparse_wide(depth), xwide = sparse_xwide(depth);  // launch-uniform: the layout follows the depth (sparse_table.hpp)
    const bool tier = env.tier != 0u;
    const uint32_t tag = sparse_tag(x, depth), tag_hi = sparse_tag_hi(x, depth), slots = sparse_slots(depth, tier);
    if (tier && nh - nl == 1u) {  // occurs once: four bits of one filter word of its OWN bucket (never displaced), no entry
        const uint32_t f = sparse_filter_hash(tag);
        atomicOr(reinterpret_cast<uint32_t *>(env.lines + uint64_t(b) * 8u) + kTierFilterWord + sparse_filter_word(f), sparse_filter_mask(f));
        atomicAdd(cur + kFiltered, 1ull);
        return;
    }
    uint64_t lval = nl;
    uint32_t wf = uint32_t(nh - nl);
    if (nh - nl >= kSparseEscapeWidth) {
        const uint64_t idx = atomicAdd(cur + kSideCursor, 1ull);
        if (idx >= env.nside) {  // (more wide ranges than the sizing pass counted: never on a consistent build -- refuse rather than write outside)
            atomicOr(cur + kFailed, 2ull);
            return;
        }
        env.side[idx] = make_uint4(uint32_t(nl), uint32_t(nl >> 32), uint32_t(nh), uint32_t(nh >> 32));
        lval = idx;
        wf = kSparseEscapeWidth;
    }
    for (uint32_t dist = 0; dist <= env.probe; ++dist, ++b) {
        const uint32_t slot = atomicAdd(env.counts + b, 1u);
        if (slot < slots) {
            uint32_t *line = reinterpret_cast<uint32_t *>(env.lines + uint64_t(b) * 8u);
            if (tier && wide) {
                line[slot] = tag;
                line[kTierWideL0Word + slot] = uint32_t(lval);
                reinterpret_cast<uint8_t *>(line)[kTierWideHiByte + slot] = uint8_t(lval >> 32);
                reinterpret_cast<uint8_t *>(line)[kTierWideWidthByte + slot] = uint8_t(wf);
            } else if (tier) {
                line[slot] = tag | (wf << kSparseTagBits);
                line[kTierL0Word + slot] = uint32_t(lval);
                reinterpret_cast<uint8_t *>(line)[kTierHiByte + slot] = uint8_t(lval >> 32);
            } else if (xwide) {
                line[slot] = tag;
                line[kSparseXL0Word + slot] = uint32_t(lval);
                reinterpret_cast<uint8_t *>(line)[kSparseXTagHiByte + slot] = uint8_t(tag_hi);
                reinterpret_cast<uint8_t *>(line)[kSparseXHiByte + slot] = uint8_t(lval >> 32);
                reinterpret_cast<uint8_t *>(line)[kSparseXWidthByte + slot] = uint8_t(wf);
            } else if (wide) {
                line[slot] = tag;
                line[kSparseWideL0Word + slot] = uint32_t(lval);
                reinterpret_cast<uint8_t *>(line)[kSparseWideHiByte + slot] = uint8_t(lval >> 32);
                reinterpret_cast<uint8_t *>(line)[kSparseWideWidthByte + slot] = uint8_t(wf);
            } else {
                line[slot] = tag | (wf << kSparseTagBits);
                line[kSparseL0Word + slot] = uint32_t(lval);
                reinterpret_cast<uint8_t *>(line)[kSparseHiByte + slot] = uint8_t(lval >> 32);
            }
            if (dist != 0u) atomicAdd(cur + kDisplaced, 1ull);
            return;
        }
    }
    atomicOr(cur + kFailed, 1ull);
}

// ---- one pair step: every node of `in` -> its non-empty children two symbols deeper --------------------------------------------
// kFinal = false: children appended to `out`;  true: counted (env.lines == nullptr) or inserted into the table.
template <bool kFinal>
__global__ __launch_bounds__(kThreads) void k_sparse_expand_pair(const Node *__restrict__ in, uint64_t in_cap, uint32_t depth,
                                                                 const uint4 *__restrict__ pair_blocks, const uint64_t *__restrict__ pair_super,
                                                                 uint32_t stride96, Node *__restrict__ out, uint64_t cap,
                                                                 unsigned long long *__restrict__ cur, FillEnv env) {
    // An earlier level of this chunk overflowed its frontier: the slots of the thread whose reservation straddled the end were never
    // written, so this level must not read them (raw allocation contents would become wild line addresses).  The host takes the whole
    // chunk again at half the size.
    if (cur[kOverflow] != 0ull) return;
    const uint64_t n = min(uint64_t(cur[kCurLevel + depth]), in_cap);
    const bool s96 = stride96 != 0u;
    for (uint64_t base = uint64_t(blockIdx.x) * kThreads; base < n; base += uint64_t(gridDim.x) * kThreads) {
        const uint64_t i = base + threadIdx.x;
        uint32_t nonempty = 0, wide = 0, single = 0;
        PairSixteen L, H;
        Node nd{0, 0, 0};
        uint64_t sb_l = 0, sb_h = 0;
        if (i < n) {
            nd = in[i];
            const uint64_t bl = pair_block_of(nd.l, s96), start_l = pair_block_start(bl, s96);
            const bool same = (nd.h - start_l) < 128u;
            const uint64_t bh = same ? bl : pair_block_of(nd.h, s96);
            pair_line_sixteen(pair_blocks + bl * 8u, uint32_t(nd.l - start_l), L);
            pair_line_sixteen(pair_blocks + bh * 8u, uint32_t(nd.h - pair_block_start(bh, s96)), H);
            sb_l = bl >> kPairSuperBlocks;
            sb_h = bh >> kPairSuperBlocks;
            if (sb_l == sb_h) {
#pragma unroll
                for (uint32_t p = 0; p < 16; ++p) {
                    nonempty |= (H.rel[p] != L.rel[p] ? 1u : 0u) << p;
                    wide |= (H.rel[p] - L.rel[p] >= kSparseEscapeWidth ? 1u : 0u) << p;
                    single |= (H.rel[p] - L.rel[p] == 1u ? 1u : 0u) << p;
                }
            } else {  // rare: the bounds lie in different superblocks
#pragma unroll
                for (uint32_t p = 0; p < 16; ++p) {
                    const uint64_t nl = pair_super[sb_l * 16u + p] + L.rel[p], nh = pair_super[sb_h * 16u + p] + H.rel[p];
                    nonempty |= (nh != nl ? 1u : 0u) << p;
                    wide |= (nh - nl >= kSparseEscapeWidth ? 1u : 0u) << p;
                    single |= (nh - nl == 1u ? 1u : 0u) << p;
                }
            }
        }
        const uint32_t mine = uint32_t(__popc(nonempty));
        if (kFinal) {
            if (env.lines == nullptr) {
                block_add(mine, cur + kCurLevel + depth + 2u);
                block_add(uint32_t(__popc(wide)), cur + kCurEscape + depth + 2u);
                block_add(uint32_t(__popc(single)), cur + kCurSingle + depth + 2u);
                continue;
            }
        } else {
            block_add(uint32_t(__popc(wide)), cur + kCurEscape + depth + 2u);
            block_add(uint32_t(__popc(single)), cur + kCurSingle + depth + 2u);
        }
        uint64_t at = kFinal ? 0ull : reserve(mine, cur + kCurLevel + depth + 2u);
        if (!kFinal && mine != 0u && at + mine > cap) {
            atomicOr(cur + kOverflow, 1ull);
            nonempty = 0;
        }
#pragma unroll
        for (uint32_t p = 0; p < 16; ++p) {
            if (((nonempty >> p) & 1u) == 0u) continue;
            const uint64_t nl = pair_super[sb_l * 16u + p] + L.rel[p], nh = pair_super[sb_h * 16u + p] + H.rel[p];
            const uint64_t key = nd.key | (uint64_t(p >> 2) << (2u * depth)) | (uint64_t(p & 3u) << (2u * depth + 2u));
            if (kFinal) sparse_insert(env, key, nl, nh, cur);
            else out[at++] = Node{key, nl, nh};
        }
    }
}

// ---- the odd last level: one symbol further from the plane blocks ------------------------------------------------------------
// start_index[s] + rank(s, pos) for the four ACGT symbols by ONE thread straight from the plane block (plane_index.hpp)
__device__ __forceinline__ void plane_line_four(const uint4 *__restrict__ blk, uint32_t r, uint64_t (&out)[4]) {
    uint32_t cnt[4] = {0, 0, 0, 0}, meta[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint4 c = blk[j];
        meta[j] = c.w;
        const uint32_t low = low_bits(min(max(int(r) - 32 * j, 0), 32));
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t s = q == 3u ? 5u : q + 1u;
            const uint32_t x0 = (s & 1u) ? 0u : ~0u, x1 = (s & 2u) ? 0u : ~0u, x2 = (s & 4u) ? 0u : ~0u;
            cnt[q] += uint32_t(__popc((c.x ^ x0) & (c.y ^ x1) & (c.z ^ x2) & low));
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t s = q == 3u ? 5u : q + 1u;
        const uint32_t hi = (((s >> 2) ? meta[7] : meta[6]) >> ((s & 3u) * 8u)) & 0xFFu;
        out[q] = ((uint64_t(hi) << 32) | meta[s]) + cnt[q];
    }
}

__global__ __launch_bounds__(kThreads) void k_sparse_expand_plane(const Node *__restrict__ in, uint64_t in_cap, uint32_t depth,
                                                                  const uint4 *__restrict__ blocks, unsigned long long *__restrict__ cur, FillEnv env) {
    if (cur[kOverflow] != 0ull) return;  // (as in k_sparse_expand_pair)
    const uint64_t n = min(uint64_t(cur[kCurLevel + depth]), in_cap);
    for (uint64_t base = uint64_t(blockIdx.x) * kThreads; base < n; base += uint64_t(gridDim.x) * kThreads) {
        const uint64_t i = base + threadIdx.x;
        uint32_t mine = 0, wide = 0, single = 0;
        uint64_t nl[4], nh[4];
        Node nd{0, 0, 0};
        if (i < n) {
            nd = in[i];
            plane_line_four(blocks + (nd.l >> 8) * 8u, uint32_t(nd.l) & 255u, nl);
            plane_line_four(blocks + (nd.h >> 8) * 8u, uint32_t(nd.h) & 255u, nh);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mine += nh[q] != nl[q] ? 1u : 0u;
                wide += nh[q] - nl[q] >= kSparseEscapeWidth ? 1u : 0u;
                single += nh[q] - nl[q] == 1u ? 1u : 0u;
            }
        }
        if (env.lines == nullptr) {
            block_add(mine, cur + kCurLevel + depth + 1u);
            block_add(wide, cur + kCurEscape + depth + 1u);
            block_add(single, cur + kCurSingle + depth + 1u);
            continue;
        }
        if (i < n) {
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q)
                if (nh[q] != nl[q]) sparse_insert(env, nd.key | (uint64_t(q) << (2u * depth)), nl[q], nh[q], cur);
        }
    }
}

// a chunk has completed: its level counts join the totals, the chunk cursors start again from 0
__global__ void k_sparse_tally(unsigned long long *cur) {
    const uint32_t i = threadIdx.x;
    if (i < 32u) {
        cur[kAccLevel + i] += cur[kCurLevel + i];
        cur[kAccEscape + i] += cur[kCurEscape + i];
        cur[kAccSingle + i] += cur[kCurSingle + i];
        cur[kCurLevel + i] = 0ull;
        cur[kCurEscape + i] = 0ull;
        cur[kCurSingle + i] = 0ull;
    }
}

// bytes 126..127 of every bucket: how many entries wanted it
__global__ __launch_bounds__(256) void k_sparse_headers(uint4 *__restrict__ lines, const uint32_t *__restrict__ counts, uint64_t nlines, uint32_t header_byte) {
    for (uint64_t b = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; b < nlines; b += uint64_t(gridDim.x) * blockDim.x)
        reinterpret_cast<uint16_t *>(lines + b * 8u)[header_byte / 2] = uint16_t(min(counts[b], 0xFFFFu));
}

uint32_t grid_for_nodes(uint64_t n) {
    const uint64_t blocks = (n + kThreads - 1) / kThreads;
    return uint32_t(std::min<uint64_t>(std::max<uint64_t>(blocks, 1), 256ull * 8));
}

struct Work {
    unsigned long long *cur;
    Node *a, *b;
    uint64_t cap;
};

Work carve(void *d_work, size_t work_bytes) {
    Work w;
    w.cur = static_cast<unsigned long long *>(d_work);
    const size_t head = 4096;
    w.cap = work_bytes > head ? (work_bytes - head) / (2 * sizeof(Node)) : 0;
    w.a = reinterpret_cast<Node *>(static_cast<char *>(d_work) + head);
    w.b = w.a + w.cap;
    return w;
}

struct Chunk {
    uint64_t i0, np;
};

// Enqueues one chunk of parents through every level up to `depth` (the last level counts or fills).
hipError_t run_chunk(const IndexView &ix, const void *flat, int flat_depth, int depth, const Chunk &c, const Work &w, const FillEnv &env, hipStream_t stream) {
    const uint4 *pair_blocks = static_cast<const uint4 *>(ix.pair_blocks);
    Node *in = w.a, *out = w.b;
    hipLaunchKernelGGL(k_sparse_seed, dim3(grid_for_nodes(c.np)), dim3(kThreads), 0, stream, static_cast<const uint4 *>(flat), c.i0, c.np, ix.total, in, w.cap,
                       w.cur, uint32_t(flat_depth));
    // every level is launched for the most nodes a frontier can hold (the kernels read the true count from the cursor)
    const uint32_t grid = grid_for_nodes(w.cap);
    for (int cur_depth = flat_depth; cur_depth < depth;) {
        const int left = depth - cur_depth;
        if (left >= 2) {
            if (left == 2)
                hipLaunchKernelGGL((k_sparse_expand_pair<true>), dim3(grid), dim3(kThreads), 0, stream, in, w.cap, uint32_t(cur_depth), pair_blocks, ix.pair_super,
                                   ix.pair_stride96 ? 1u : 0u, out, w.cap, w.cur, env);
            else
                hipLaunchKernelGGL((k_sparse_expand_pair<false>), dim3(grid), dim3(kThreads), 0, stream, in, w.cap, uint32_t(cur_depth), pair_blocks, ix.pair_super,
                                   ix.pair_stride96 ? 1u : 0u, out, w.cap, w.cur, env);
            cur_depth += 2;
        } else {
            hipLaunchKernelGGL(k_sparse_expand_plane, dim3(grid), dim3(kThreads), 0, stream, in, w.cap, uint32_t(cur_depth), static_cast<const uint4 *>(ix.blocks),
                               w.cur, env);
            cur_depth += 1;
        }
        std::swap(in, out);
    }
    return hipGetLastError();
}

// the chunking the sizing pass found, replayed by the fill pass (one builder at a time per process: the loader holds the handle's lock,
// and the list is keyed by what it was made for)
struct ChunkPlan {
    const void *flat = nullptr;
    int flat_depth = -1, depth = -1;
    uint64_t cap = 0;
    std::vector<Chunk> chunks;
};
thread_local ChunkPlan g_plan;

}  // namespace

size_t sparse_work_bytes(uint64_t free_bytes) {
    // two frontiers of up to 2^27 nodes (6.4 GB) when HBM is plentiful, an eighth of what is free otherwise, 2^16 nodes at least
    const uint64_t want = uint64_t(2) * sizeof(Node) << 27, least = uint64_t(2) * sizeof(Node) << 16;
    return size_t(4096 + std::max(least, std::min(want, free_bytes / 8)));
}

hipError_t sparse_count_levels(const IndexView &ix, const void *flat_entries, int flat_depth, int max_depth, void *d_work, size_t work_bytes,
                               SparseBuildReport *report, hipStream_t stream) {
    if (!ix.pair_blocks || !ix.pair_super || ix.block_format != kBlocksPlanes || max_depth > kSparseMaxDepth || flat_depth < 0 || flat_depth >= max_depth)
        return hipErrorInvalidValue;
    if (flat_entries == nullptr) flat_depth = 0;
    const Work w = carve(d_work, work_bytes);
    if (w.cap < 1024) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(w.cur, 0, kCursorWords * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const FillEnv none{nullptr, nullptr, nullptr, 0, 0, uint32_t(2 * max_depth), 0, 0};
    const uint64_t parents = flat_entries ? (uint64_t(1) << (2 * flat_depth)) : 1;
    g_plan = ChunkPlan{flat_entries, flat_depth, max_depth, w.cap, {}};
    // Chunks of parents: the first is small, the following ones are sized by what the last one's largest frontier was, aiming at a
    // quarter of the buffer; a chunk that overflows it is taken again at half the size (nothing of it has been tallied).
    uint64_t np = std::min<uint64_t>(parents, uint64_t(1) << 16);
    std::vector<unsigned long long> cur(kCursorWords);
    for (uint64_t i0 = 0; i0 < parents;) {
        const Chunk c{i0, std::min(np, parents - i0)};
        e = run_chunk(ix, flat_entries, flat_depth, max_depth, c, w, none, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(cur.data(), w.cur, cur.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (cur[kOverflow] != 0) {
            if (c.np == 1) return hipErrorOutOfMemory;  // one parent's descendants do not fit the scratch
            np = std::max<uint64_t>(1, c.np / 2);
            e = hipMemsetAsync(w.cur + kCurLevel, 0, 32 * sizeof(unsigned long long), stream);
            if (e == hipSuccess) e = hipMemsetAsync(w.cur + kCurEscape, 0, 32 * sizeof(unsigned long long), stream);
            if (e == hipSuccess) e = hipMemsetAsync(w.cur + kCurSingle, 0, 32 * sizeof(unsigned long long), stream);
            if (e == hipSuccess) e = hipMemsetAsync(w.cur + kOverflow, 0, sizeof(unsigned long long), stream);
            if (e != hipSuccess) return e;
            continue;
        }
        hipLaunchKernelGGL(k_sparse_tally, dim3(1), dim3(64), 0, stream, w.cur);
        g_plan.chunks.push_back(c);
        i0 += c.np;
        unsigned long long widest = 1;
        for (int d = flat_depth; d < max_depth; ++d) widest = std::max(widest, cur[kCurLevel + d]);  // (the last level is not materialised)
        const double per_parent = double(widest) / double(c.np);
        np = uint64_t(std::max(1.0, std::min(double(uint64_t(1) << 26), double(w.cap) / 4.0 / std::max(per_parent, 1e-6))));
    }
    e = hipMemcpyAsync(cur.data(), w.cur, cur.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    *report = SparseBuildReport{};
    report->parent_depth = flat_depth;
    for (int d = 0; d <= kSparseMaxDepth; ++d) {
        report->distinct[d] = cur[kAccLevel + d];
        report->escapes[d] = cur[kAccEscape + d];
        report->singles[d] = cur[kAccSingle + d];
    }
    return hipSuccess;
}

hipError_t sparse_fill(const IndexView &ix, const void *flat_entries, int flat_depth, int depth, bool tier, void *lines, uint64_t nbuckets, uint32_t probe, void *side,
                       uint64_t nside, void *d_counts, void *d_work, size_t work_bytes, SparseBuildReport *report, hipStream_t stream) {
    if (flat_entries == nullptr) flat_depth = 0;
    if (!ix.pair_blocks || !lines || !d_counts || depth < kSparseMinDepth || depth > kSparseMaxDepth || flat_depth >= depth || nbuckets == 0 ||
        nbuckets + probe > 0xFFFFFFFFull || int(probe) > sparse_probe_limit(depth, nbuckets) || (tier && depth > kTierMaxDepth))
        return hipErrorInvalidValue;
    const Work w = carve(d_work, work_bytes);
    const uint64_t nlines = nbuckets + probe;
    hipError_t e = hipMemsetAsync(w.cur, 0, kCursorWords * sizeof(unsigned long long), stream);
    if (e == hipSuccess) e = hipMemsetAsync(lines, 0, nlines * 128, stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_counts, 0, nlines * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const FillEnv env{static_cast<uint4 *>(lines), static_cast<uint32_t *>(d_counts), static_cast<uint4 *>(side), uint32_t(nbuckets), probe, uint32_t(2 * depth),
                      side ? nside : 0, tier ? 1u : 0u};
    // The chunking of the sizing pass holds for every depth up to the one it was made for, of either parity: the pair levels are the
    // same ones (a frontier at depth d does not depend on where the expansion ends), and the last level is never materialised.  Without
    // one (a fill that was not preceded by its sizing pass): small chunks.
    std::vector<Chunk> chunks;
    if (g_plan.flat == flat_entries && g_plan.flat_depth == flat_depth && g_plan.depth >= depth && g_plan.cap == w.cap) {
        chunks = g_plan.chunks;
    } else {
        const uint64_t parents = flat_entries ? (uint64_t(1) << (2 * flat_depth)) : 1;
        for (uint64_t i0 = 0; i0 < parents; i0 += 4096) chunks.push_back(Chunk{i0, std::min<uint64_t>(4096, parents - i0)});
    }
    for (const Chunk &c : chunks) {
        e = run_chunk(ix, flat_entries, flat_depth, depth, c, w, env, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_sparse_tally, dim3(1), dim3(64), 0, stream, w.cur);
    }
    hipLaunchKernelGGL(k_sparse_headers, dim3(uint32_t(std::min<uint64_t>((nlines + 255) / 256, 2048))), dim3(256), 0, stream, static_cast<uint4 *>(lines),
                       static_cast<const uint32_t *>(d_counts), nlines, tier ? kTierHeaderByte : kSparseHeaderByte);
    std::vector<unsigned long long> cur(kCursorWords);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(cur.data(), w.cur, cur.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    if (cur[kOverflow] != 0) return hipErrorOutOfMemory;
    if (cur[kFailed] != 0) return hipErrorInvalidValue;  // some entry found no slot: more buckets, please
    report->depth = depth;
    report->nbuckets = nbuckets;
    report->nescapes = cur[kSideCursor];
    report->displaced = cur[kDisplaced];
    report->tier = tier;
    report->filtered = cur[kFiltered];
    report->entries = report->distinct[depth] - (tier ? cur[kFiltered] : 0ull);
    return hipSuccess;
}

}  // namespace msbwt
