// Run-block index prototype (NOT part of the product): the layout BASELINE.json's north_star
// sketches -- fixed-width runs + per-block occurrence counts (src/run_block_av_flat.rs:43-56,
// 97-125), LDS-free cooperative run decoding with a DPP prefix sum -- built next to the product's
// plane blocks so that the two can be measured on the same stream with the same launch shape
// (tools/runblock_bench.py).  DESIGN.md section 2 quotes the result.
//
// R512 block = 128 bytes = one line = BWT positions [512 b, 512 b + 512):
//   chunks 0,1 (32 B)  the same eight header words as a plane block: for s = 0..5 the 40-bit value
//                      A[s] = start_index[s] + occ(s, 512 b) (low words in words 0..5, high bytes in
//                      words 6, 7); bit 31 of word 7 = OVERFLOW
//   chunks 2..7 (96 B) 96 one-byte runs, byte = sym | len << 3 with len 1..31 (0 = unused slot);
//                      a BWT run is cut at block borders and into pieces of at most 31
//   overflow           a block that would need more than 96 pieces stores, instead of runs, the index
//                      of two plane blocks (256 B, 512 positions) in a side array: a second,
//                      dependent fetch, for the rare low-run-length block only
// bytes/symbol: 0.25 + overflow (plane blocks: 0.5; the reference's RleBWT: ~0.17 of RLE bytes +
// 0.22 of sampled counts at bin_power 8).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kRunsPerBlock = 96;
constexpr uint32_t kOverflowBit = 0x80000000u;

struct Lab {
    uint4 *d_blocks = nullptr;    // R512 blocks
    uint4 *d_overflow = nullptr;  // pairs of plane blocks
    uint64_t nblocks = 0, noverflow = 0, total = 0, pieces = 0;
};

template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_add(uint32_t x) {
    return x + uint32_t(__builtin_amdgcn_update_dpp(0, int(x), kCtrl, 0xF, 0xF, true));
}
__device__ __forceinline__ uint32_t group_sum(uint32_t x) {  // over an aligned group of 8 lanes
    x = dpp_add<0xB1>(x);
    x = dpp_add<0x4E>(x);
    x = dpp_add<0x141>(x);
    return x;
}

// rank over 512 positions held as two plane blocks (the overflow representation), 8 lanes
__device__ __forceinline__ uint32_t plane_pair_count(const uint4 *two_blocks, uint32_t s, uint32_t r, uint32_t sub) {
    const uint32_t x0 = (s & 1u) ? 0u : ~0u, x1 = (s & 2u) ? 0u : ~0u, x2 = (s & 4u) ? 0u : ~0u;
    uint32_t cnt = 0;
    for (int half = 0; half < 2; ++half) {
        const uint4 c = two_blocks[half * 8 + sub];
        const int n = min(max(int(r) - int(half * 256 + sub * 32), 0), 32);
        const uint32_t m = n >= 32 ? ~0u : ((1u << n) - 1u);
        cnt += uint32_t(__popc((c.x ^ x0) & (c.y ^ x1) & (c.z ^ x2) & m));
    }
    return group_sum(cnt);
}

// out[i] = start_index[s] + rank(s, pos[i]); 8 lanes per query, one 128-byte line per rank:
// lanes 0,1 hold the header, lanes 2..7 decode 16 runs each; a prefix sum over the lanes gives
// every lane the block offset of its first run, each lane clips its runs against the target offset
// and the group sums the matches (the north_star's "prefix-sum to find the straddling run").
__global__ __launch_bounds__(256) void k_rank_runs(const uint4 *__restrict__ blocks, const uint4 *__restrict__ overflow,
                                                   const uint8_t *__restrict__ syms, const uint64_t *__restrict__ pos, uint64_t n,
                                                   uint64_t *__restrict__ out) {
    const uint32_t sub = threadIdx.x & 7u;
    const uint32_t lane = threadIdx.x & 63u, group_base = lane & ~7u;
    const uint64_t ngroups = (uint64_t(gridDim.x) * blockDim.x) / 8;
    for (uint64_t i = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) / 8; i < n; i += ngroups) {
        const uint32_t s = syms[i];
        const uint64_t p = pos[i];
        const uint32_t r = uint32_t(p) & 511u;
        const uint4 c = blocks[(p >> 9) * 8 + sub];
        // header: A[s] low word in word s (lane s>>2, component s&3), high byte in word 6/7 (lane 1)
        const uint32_t w0 = uint32_t(__shfl(int(s & 4u ? 0 : (s == 0 ? c.x : s == 1 ? c.y : s == 2 ? c.z : c.w)), int(group_base)));
        const uint32_t w1 = uint32_t(__shfl(int(s == 4 ? c.x : c.y), int(group_base + 1)));
        const uint32_t lo = (s & 4u) ? w1 : w0;
        const uint32_t hi03 = uint32_t(__shfl(int(c.z), int(group_base + 1))), hi45 = uint32_t(__shfl(int(c.w), int(group_base + 1)));
        const uint32_t hi = (((s >> 2) ? hi45 : hi03) >> ((s & 3u) * 8u)) & 0xFFu;
        uint32_t cnt;
        if (hi45 & kOverflowBit) {  // group-uniform
            const uint32_t idx = uint32_t(__shfl(int(c.x), int(group_base + 2)));
            cnt = plane_pair_count(overflow + uint64_t(idx) * 16, s, r, sub);
        } else {
            const uint32_t word[4] = {c.x, c.y, c.z, c.w};
            uint32_t mine = 0;  // symbols my 16 runs cover (lanes 0,1: none)
            if (sub >= 2u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) mine = __builtin_amdgcn_sad_u8((word[j] >> 3) & 0x1F1F1F1Fu, 0u, mine);
            }
            // exclusive prefix over the 8 lanes (3 DPP steps within the row of 16 lanes)
            uint32_t inc = mine;
            for (int d = 1; d < 8; d <<= 1) {
                const uint32_t y = uint32_t(__shfl_up(int(inc), d, 8));
                if (int(sub) >= d) inc += y;
            }
            int cur = int(inc - mine);
            uint32_t c_local = 0;
            if (sub >= 2u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t run = (word[j] >> (8 * b)) & 0xFFu;
                        const int len = int(run >> 3);
                        const int take = min(max(int(r) - cur, 0), len);
                        c_local += ((run & 7u) == s) ? uint32_t(take) : 0u;
                        cur += len;
                    }
                }
            }
            cnt = group_sum(c_local);
        }
        if (sub == 0) out[i] = ((uint64_t(hi) << 32) | lo) + cnt;
    }
}

}  // namespace

extern "C" {

// Builds the R512 index of an RLE byte stream (msbwt_core.rs:3-14 format) on the host and uploads
// it.  start_index = exclusive prefix of the symbol totals ($ A C G N T).  Returns an opaque handle.
void *rb_build(const uint8_t *rle, size_t n) {
    // pass 1: totals
    uint64_t counts[6] = {0};
    {
        uint8_t prev = 255;
        uint64_t mult = 1;
        for (size_t i = 0; i < n; ++i) {
            const uint8_t sym = rle[i] & 7u;
            if (sym >= 6) return nullptr;
            mult = (sym == prev) ? mult * 32 : 1;
            counts[sym] += uint64_t(rle[i] >> 3) * mult;
            prev = sym;
        }
    }
    uint64_t start[6], total = 0;
    for (int s = 0; s < 6; ++s) { start[s] = total; total += counts[s]; }
    Lab *lab = new Lab();
    lab->total = total;
    lab->nblocks = (total >> 9) + 1;
    std::vector<uint32_t> blocks(size_t(lab->nblocks) * 32, 0);
    std::vector<uint32_t> over;
    // pass 2: walk the runs, cut at block borders and at 31
    uint64_t occ[6] = {0};
    uint64_t pos = 0, blk = 0;
    std::vector<uint8_t> pieces;            // of the current block
    std::vector<uint8_t> symbols_of_block;  // only materialised for an overflowing block
    uint64_t block_occ[6] = {0};
    auto header = [&](uint32_t *w, const uint64_t at[6], bool overflow) {
        uint32_t hi03 = 0, hi45 = 0;
        for (int s = 0; s < 6; ++s) {
            const uint64_t a = start[s] + at[s];
            w[s] = uint32_t(a);
            if (s < 4) hi03 |= uint32_t((a >> 32) & 0xFF) << (8 * s);
            else hi45 |= uint32_t((a >> 32) & 0xFF) << (8 * (s - 4));
        }
        w[6] = hi03;
        w[7] = hi45 | (overflow ? kOverflowBit : 0u);
    };
    auto flush = [&]() {
        uint32_t *w = &blocks[size_t(blk) * 32];
        const bool overflow = pieces.size() > size_t(kRunsPerBlock);
        header(w, block_occ, overflow);
        if (!overflow) {
            std::memcpy(reinterpret_cast<uint8_t *>(w) + 32, pieces.data(), pieces.size());
        } else {  // expand the block's symbols into two plane blocks (chunk layout of plane_index.hpp, headers unused)
            std::vector<uint8_t> sym;
            for (uint8_t pc : pieces) sym.insert(sym.end(), size_t(pc >> 3), uint8_t(pc & 7u));
            sym.resize(512, 0);
            const uint32_t idx = uint32_t(over.size() / 64);
            over.resize(over.size() + 64, 0);
            uint32_t *o = &over[size_t(idx) * 64];
            for (int i = 0; i < 512; ++i)
                for (int p = 0; p < 3; ++p)
                    if ((sym[i] >> p) & 1u) o[(i >> 8) * 32 + ((i & 255) >> 5) * 4 + p] |= 1u << (i & 31);
            w[8] = idx;
            ++lab->noverflow;
        }
        lab->pieces += pieces.size();
        pieces.clear();
        ++blk;
        for (int s = 0; s < 6; ++s) block_occ[s] = occ[s];
    };
    auto emit = [&](uint8_t sym, uint64_t len) {
        while (len) {
            const uint64_t room = 512 - (pos & 511);
            uint64_t take = len < room ? len : room;
            uint64_t left = take;
            while (left) {
                const uint64_t piece = left < 31 ? left : 31;
                // a long run inside one block: 17 pieces cover 512, well under 96
                pieces.push_back(uint8_t(sym | (piece << 3)));
                left -= piece;
            }
            occ[sym] += take;
            pos += take;
            len -= take;
            if ((pos & 511) == 0) flush();
        }
    };
    {
        uint8_t prev = 255;
        uint64_t mult = 1, run = 0;
        for (size_t i = 0; i < n; ++i) {
            const uint8_t sym = rle[i] & 7u;
            if (sym != prev) {
                if (prev != 255 && run) emit(prev, run);
                prev = sym;
                mult = 1;
                run = 0;
            } else {
                mult *= 32;
            }
            run += uint64_t(rle[i] >> 3) * mult;
        }
        if (prev != 255 && run) emit(prev, run);
    }
    while (blk < lab->nblocks) flush();  // the last, partial block (and the block of position == total)
    if (hipMalloc(reinterpret_cast<void **>(&lab->d_blocks), blocks.size() * 4) != hipSuccess ||
        hipMemcpy(lab->d_blocks, blocks.data(), blocks.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    if (!over.empty() && (hipMalloc(reinterpret_cast<void **>(&lab->d_overflow), over.size() * 4) != hipSuccess ||
                          hipMemcpy(lab->d_overflow, over.data(), over.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) return nullptr;
    return lab;
}

void rb_info(const void *h, uint64_t *total, uint64_t *nblocks, uint64_t *noverflow, uint64_t *pieces) {
    const Lab *lab = static_cast<const Lab *>(h);
    *total = lab->total;
    *nblocks = lab->nblocks;
    *noverflow = lab->noverflow;
    *pieces = lab->pieces;
}

// out[i] = start_index[sym] + rank(sym, pos[i]) for device arrays; returns the average kernel ms over `iters` launches
double rb_rank(const void *h, const void *d_syms, const void *d_pos, uint64_t n, void *d_out, int iters) {
    const Lab *lab = static_cast<const Lab *>(h);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const uint64_t want = (n * 8 + 255) / 256;
    const dim3 grid(uint32_t(want > 2048 ? 2048 : (want ? want : 1)));
    hipLaunchKernelGGL(k_rank_runs, grid, dim3(256), 0, 0, lab->d_blocks, lab->d_overflow, static_cast<const uint8_t *>(d_syms),
                       static_cast<const uint64_t *>(d_pos), n, static_cast<uint64_t *>(d_out));
    (void)hipEventRecord(a);
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(k_rank_runs, grid, dim3(256), 0, 0, lab->d_blocks, lab->d_overflow, static_cast<const uint8_t *>(d_syms),
                           static_cast<const uint64_t *>(d_pos), n, static_cast<uint64_t *>(d_out));
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return hipGetLastError() == hipSuccess ? double(ms) / iters : -1.0;
}

void rb_free(void *h) {
    Lab *lab = static_cast<Lab *>(h);
    if (!lab) return;
    if (lab->d_blocks) (void)hipFree(lab->d_blocks);
    if (lab->d_overflow) (void)hipFree(lab->d_overflow);
    delete lab;
}

}  // extern "C"
