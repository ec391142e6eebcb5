#include "run_index.hpp"

#include <algorithm>
#include <cstring>
#include <thread>

namespace msbwt {
namespace {

struct Segment {
    size_t byte_begin, byte_end;  // run-aligned byte range of the RLE stream
    uint64_t pos;                 // BWT position of its first symbol
    uint64_t occ[kAlphabet];      // symbol counts before it
};

// Walks the runs of the stream from a run-aligned byte offset.
struct RunCursor {
    const uint8_t *rle;
    size_t i, n;
    bool next(uint8_t *sym, uint64_t *len) {
        if (i >= n) return false;
        const uint8_t s = rle[i] & 7u;
        uint64_t l = 0, weight = 1;
        do {
            l += uint64_t(rle[i] >> 3) * weight;
            weight <<= 5;
            ++i;
        } while (i < n && (rle[i] & 7u) == s);
        *sym = s;
        *len = l;
        return true;
    }
};

void write_header(uint32_t *w, const Totals &t, const uint64_t occ[kAlphabet], bool overflow) {
    uint32_t hi03 = 0, hi45 = 0;
    for (int s = 0; s < kAlphabet; ++s) {
        const uint64_t a = t.start_index[s] + occ[s];
        w[s] = uint32_t(a);
        if (s < 4) hi03 |= uint32_t((a >> 32) & 0xFFu) << (8 * s);
        else hi45 |= uint32_t((a >> 32) & 0xFFu) << (8 * (s - 4));
    }
    w[6] = hi03;
    w[7] = hi45 | (overflow ? kRunOverflowBit : 0u);
}

// One worker builds the blocks [first, last): it finds its first position inside `seg` (the byte
// segment that contains it) and walks the runs from there.
struct Worker {
    std::vector<uint32_t> overflow;        // 64 words per overflowing block, in block order
    std::vector<uint64_t> overflow_block;  // which blocks they belong to
};

void build_range(const uint8_t *rle, size_t n, const Totals &t, const Segment &seg, uint64_t first, uint64_t last,
                 uint32_t *blocks, Worker *out) {
    RunCursor cur{rle, seg.byte_begin, n};
    uint64_t pos = seg.pos, occ[kAlphabet];
    std::memcpy(occ, seg.occ, sizeof occ);
    const uint64_t start = first << kRunShift, stop = std::min(last << kRunShift, t.total);
    uint8_t sym = 0;
    uint64_t len = 0;
    // skip to `start` (it may fall inside a run)
    while (pos < start) {
        if (!len && !cur.next(&sym, &len)) break;
        const uint64_t take = std::min(len, start - pos);
        pos += take;
        occ[sym] += take;
        len -= take;
    }
    uint8_t pieces[2048];
    for (uint64_t b = first; b < last; ++b) {
        uint32_t *w = blocks + b * 32;
        const uint64_t block_end = std::min((b + 1) << kRunShift, stop);
        uint64_t at_start[kAlphabet];
        std::memcpy(at_start, occ, sizeof occ);
        size_t np = 0;
        while (pos < block_end) {
            if (!len && !cur.next(&sym, &len)) break;
            uint64_t take = std::min(len, block_end - pos);
            pos += take;
            occ[sym] += take;
            len -= take;
            while (take) {  // at most 512 symbols per block: 17 pieces of 31 at most per run part, 512 pieces in all
                const uint64_t piece = std::min<uint64_t>(take, 31);
                pieces[np++] = uint8_t(sym | (piece << 3));
                take -= piece;
            }
        }
        const bool overflow = np > size_t(kRunsPerBlock);
        write_header(w, t, at_start, overflow);
        if (!overflow) {
            std::memcpy(reinterpret_cast<uint8_t *>(w) + 32, pieces, np);
        } else {  // the block's symbols as TWO plane blocks (plane_index.hpp): 8 chunks of {plane0, plane1, plane2, meta} each,
                  // the meta words holding A[s] at the block's own first position -- 512 b and 512 b + 256
            const size_t base = out->overflow.size();
            out->overflow.resize(base + 64, 0);
            uint32_t *o = out->overflow.data() + base;
            uint64_t at_half[kAlphabet];
            std::memcpy(at_half, at_start, sizeof at_half);
            unsigned i = 0;
            for (size_t p = 0; p < np; ++p)
                for (unsigned c = 0; c < unsigned(pieces[p] >> 3); ++c, ++i) {
                    if (i < 256u) ++at_half[pieces[p] & 7u];
                    for (int pl = 0; pl < 3; ++pl)
                        if ((pieces[p] >> pl) & 1u) o[(i >> 8) * 32 + ((i & 255u) >> 5) * 4 + pl] |= 1u << (i & 31u);
                }
            for (int half = 0; half < 2; ++half) {
                uint32_t meta[8];
                write_header(meta, t, half ? at_half : at_start, false);
                for (int j = 0; j < 8; ++j) o[half * 32 + j * 4 + 3] = meta[j];
            }
            out->overflow_block.push_back(b);
        }
    }
}

}  // namespace

void build_run_blocks(const uint8_t *rle, size_t n, const Totals &totals, RunIndex *out, int threads) {
    const uint64_t nblocks = run_block_count(totals.total);
    if (threads <= 0) threads = int(std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency())));
    if (n < (1u << 20)) threads = 1;
    out->nblocks = nblocks;
    out->blocks.assign(size_t(nblocks) * 32, 0);
    // run-aligned byte segments with their start positions and symbol counts
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
    if (segs.empty()) segs.push_back(Segment{});
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
    // every worker owns a contiguous range of blocks and starts from the byte segment that holds its first position
    const int workers = int(std::min<uint64_t>(uint64_t(threads), nblocks));
    std::vector<Worker> parts(static_cast<size_t>(workers));
    {
        std::vector<std::thread> pool;
        for (int w = 0; w < workers; ++w) {
            const uint64_t first = nblocks * uint64_t(w) / uint64_t(workers), last = nblocks * uint64_t(w + 1) / uint64_t(workers);
            size_t k = segs.size() - 1;
            while (k > 0 && segs[k].pos > (first << kRunShift)) --k;
            pool.emplace_back([&, w, first, last, k] { build_range(rle, n, totals, segs[k], first, last, out->blocks.data(), &parts[size_t(w)]); });
        }
        for (auto &th : pool) th.join();
    }
    // overflow lines of all workers, in block order; each overflowing block learns its index (word 8)
    out->overflow.clear();
    out->noverflow = 0;
    for (const Worker &p : parts) {
        for (size_t j = 0; j < p.overflow_block.size(); ++j) out->blocks[size_t(p.overflow_block[j]) * 32 + 8] = uint32_t(out->noverflow + j);
        out->overflow.insert(out->overflow.end(), p.overflow.begin(), p.overflow.end());
        out->noverflow += p.overflow_block.size();
    }
}

}  // namespace msbwt
