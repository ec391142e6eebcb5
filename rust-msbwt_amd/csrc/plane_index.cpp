#include "plane_index.hpp"

#include <algorithm>
#include <cstring>
#include <thread>

namespace msbwt {
namespace {

// Writes the header (meta words) of block `b` from the running bounds A[].
inline void write_meta(uint32_t *block, const uint64_t A[kAlphabet]) {
    uint32_t hi_a = 0, hi_b = 0;
    for (int s = 0; s < kAlphabet; ++s) {
        block[4 * s + 3] = uint32_t(A[s]);
        const uint32_t hi = uint32_t(A[s] >> 32) & 0xFFu;
        if (s < 4) hi_a |= hi << (8 * s);
        else hi_b |= hi << (8 * (s - 4));
    }
    block[4 * 6 + 3] = hi_a;
    block[4 * 7 + 3] = hi_b;
}

// Sets symbol `sym` at in-block positions [from, to) (the planes start out zero).
inline void paint(uint32_t *block, uint8_t sym, unsigned from, unsigned to) {
    if (sym == 0 || from >= to) return;
    for (unsigned j = from >> 5; j <= (to - 1) >> 5; ++j) {
        const unsigned lo = std::max(from, j * 32) - j * 32;
        const unsigned hi = std::min(to, j * 32 + 32) - j * 32;  // 1..32
        const uint32_t mask = (hi == 32 ? ~0u : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        if (sym & 1u) block[4 * j + 0] |= mask;
        if (sym & 2u) block[4 * j + 1] |= mask;
        if (sym & 4u) block[4 * j + 2] |= mask;
    }
}

struct Segment {
    size_t byte_begin, byte_end;  // run-aligned byte range of the RLE stream
    uint64_t pos;                 // BWT position of its first symbol
    uint64_t occ[kAlphabet];      // symbol counts before it
};

// Expands the runs of one segment.  Blocks that a segment only partly covers are shared
// with a neighbour: plane bits are OR-ed in (disjoint bit ranges); a block's header is written
// whenever the running position reaches the block's first symbol.
void expand_segment(const uint8_t *rle, const Segment &seg, const Totals &t, uint32_t *out) {
    uint64_t A[kAlphabet];
    for (int s = 0; s < kAlphabet; ++s) A[s] = t.start_index[s] + seg.occ[s];
    uint64_t pos = seg.pos;
    if ((pos & (kBlockSymbols - 1)) == 0) write_meta(out + (pos >> kBlockShift) * 32, A);
    for_each_run(rle + seg.byte_begin, seg.byte_end - seg.byte_begin, [&](uint8_t sym, uint64_t len) {
        while (len > 0) {
            const unsigned off = unsigned(pos & (kBlockSymbols - 1));
            const uint64_t take = std::min<uint64_t>(len, kBlockSymbols - off);
            paint(out + (pos >> kBlockShift) * 32, sym, off, off + unsigned(take));
            pos += take;
            len -= take;
            A[sym] += take;
            if ((pos & (kBlockSymbols - 1)) == 0) write_meta(out + (pos >> kBlockShift) * 32, A);
        }
    });
}

}  // namespace

void build_plane_blocks(const uint8_t *rle, size_t n, const Totals &totals, uint32_t *out, int threads) {
    const uint64_t nblocks = plane_block_count(totals.total);
    if (threads <= 0) threads = int(std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency())));
    if (n < (1u << 20)) threads = 1;

    // cut the byte stream into run-aligned segments and prefix their symbol counts
    std::vector<Segment> segs;
    size_t begin = 0;
    for (int t = 0; t < threads && begin < n; ++t) {
        size_t end = (t == threads - 1) ? n : std::max(begin + 1, n * size_t(t + 1) / size_t(threads));
        while (end < n && (rle[end] & 7u) == (rle[end - 1] & 7u)) ++end;  // do not split a run
        Segment s{};
        s.byte_begin = begin;
        s.byte_end = end;
        segs.push_back(s);
        begin = end;
    }
    {
        std::vector<Totals> part(segs.size());
        std::vector<std::thread> pool;
        for (size_t i = 0; i < segs.size(); ++i)
            pool.emplace_back([&, i] { compute_totals(rle + segs[i].byte_begin, segs[i].byte_end - segs[i].byte_begin, &part[i]); });
        for (auto &th : pool) th.join();
        uint64_t pos = 0, occ[kAlphabet] = {0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < segs.size(); ++i) {
            segs[i].pos = pos;
            std::memcpy(segs[i].occ, occ, sizeof occ);
            pos += part[i].total;
            for (int s = 0; s < kAlphabet; ++s) occ[s] += part[i].symbol_counts[s];
        }
    }

    // zero the blocks in parallel slices, then paint.  Two passes so that a block shared by
    // two segments is never zeroed after a neighbour painted it.
    {
        std::vector<std::thread> pool;
        const int zt = std::max(1, threads);
        for (int t = 0; t < zt; ++t)
            pool.emplace_back([&, t] {
                const uint64_t lo = nblocks * uint64_t(t) / uint64_t(zt), hi = nblocks * uint64_t(t + 1) / uint64_t(zt);
                std::memset(out + lo * 32, 0, size_t(hi - lo) * kBlockBytes);
            });
        for (auto &th : pool) th.join();
    }
    if (segs.empty()) {
        Segment s{};
        segs.push_back(s);  // empty stream: just the header of block 0
    }
    // Neighbouring segments paint disjoint bit ranges, but possibly of the same 32-bit word
    // (a segment boundary inside a block), and both may write the header of a block that
    // starts exactly on their border (same values).  Running the even and the odd segments
    // in two phases keeps neighbours apart without any atomics.
    for (int phase = 0; phase < 2; ++phase) {
        std::vector<std::thread> pool;
        for (size_t i = size_t(phase); i < segs.size(); i += 2)
            pool.emplace_back([&, i] { expand_segment(rle, segs[i], totals, out); });
        for (auto &th : pool) th.join();
    }
}

}  // namespace msbwt
