// Device code shared by the two count_kmers kernels (kernels.hip: 8-lane groups, lanes.hip: one
// query per lane): where a tile's queries come from, how one lane validates / packs its query
// and looks it up in the suffix table, and where counts go.  HIP translation units only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"
#include "rank_ops.hpp"

namespace msbwt {

// Where a tile's queries come from and where their counts go.
//   matrix mode: `data` = n x k symbol codes (the batch API); counts to out_fwd[q].
//   reads mode:  `data` = n_reads x read_len bytes; query = one k-mer window of one read, on
//                the forward strand and/or reverse-complemented (string_util.rs:12,45-50);
//                bytes are symbol codes or ASCII (string_util.rs:15-32 mapping).  Query
//                preparation -- convert_stoi, windowing, reverse_complement_i -- happens here,
//                in registers, instead of on the host.
struct QuerySource {
    const uint8_t *data;
    uint64_t n;          // queries (reads mode: windows x strands)
    uint32_t k;
    uint32_t read_len;   // reads mode
    uint32_t windows;    // read_len - k + 1
    uint32_t strands;    // bit 0: forward wanted, bit 1: reverse complement wanted
    uint32_t ascii;
    uint64_t *out_fwd, *out_rc;
    // ragged reads (different lengths): read r occupies data[read_off[r] .. read_off[r+1]) and
    // owns the global windows [win_off[r], win_off[r+1]); nullptr = fixed read_len
    const uint64_t *read_off, *win_off;
    uint64_t n_reads;
    // single-query calls (the trait's count_kmer, msbwt_core.rs:124): the k <= 64 symbols travel INSIDE the kernel
    // arguments, which the wave reads at start-up anyway -- one PCIe round trip less than fetching them from the
    // host's buffer.  inline_n = 1: the batch is this one query (matrix mode, lanes kernel only).
    uint32_t inline_n;
    uint4 inline_kmer[4];
    // matrix mode, PACKED queries (lanes kernel, k <= 64): `data` = n x ceil(k / 32) u64 words, two bits per symbol (A C G T ->
    // 0..3), the k-mer read as a base-4 number with its FIRST symbol most significant -- the last symbol sits in bits 0-1 of
    // word 0, symbol k - 33 in bits 0-1 of word 1.  8 bytes per 31-mer instead of 31; '$' / 'N' cannot be said.
    uint32_t packed;
    // where query v's count goes: out_fwd[out_index[v]] (an ordered batch is counted in index order and its counts return to
    // the caller's order, order.hip); nullptr = out_fwd[v].  Matrix mode, lanes kernel; n <= 2^32.
    const uint32_t *out_index;
    // packed queries as ELEMENTS of packed_stride u64 words each (0 = just the ceil(k / 32) words); with place_inline the word
    // after the query's own holds the place its count goes to (low 32 bits) -- one 16-byte load per 31-mer brings both
    // (what the ordering passes of order.hip hand over: one store per element instead of two)
    uint32_t packed_stride, place_inline;
};

namespace {

constexpr int kTile = 64;       // queries per wave tile
constexpr int kMaxShortK = 32;  // 3 dwords of packed symbols
constexpr int kMaxTiledK = 64;  // 6 dwords

// compiler-level ordering of one wave's LDS writes before its later LDS reads (the LDS
// executes a wave's operations in issue order; no s_barrier is needed inside a wave)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// kWords = dwords of packed symbols a query carries: 3 (k <= 32) or 6 (k <= 64).
template <int kWords>
struct alignas(16) WorkItemT {  // one undecided query: 32 bytes (kWords 3) or 48 (kWords 6)
    uint32_t l_lo, l_hi, h_lo, h_hi;
    uint32_t w[kWords];   // remaining symbols, 3 bits each, next step in the low bits
    uint32_t rem_slot;    // remaining steps | slot << 8
};

// the suffix table and its presence filter as the setup code sees them
struct TableEnv {
    const uint4 *table;
    uint32_t depth;        // symbols a table entry stands for
    bool use_table;        // table present and k >= depth
    bool packed;           // entry format: false = flat {l, h} (16 B), true = packed lines (kernels.hpp)
    const uint32_t *filter;
    uint32_t filter_mask;
    uint64_t total;
    const uint4 *side;     // flat entries of the packed table's escape lines (nullptr: such queries search from scratch)
};

// Packed suffix table (TableView::packed): a 128-byte line = u64 base | 30 x u32 { l - base : 16, h - l : 16 }
// for 30 consecutive table indices (consecutive indices are consecutive ranges, so the deltas are
// small); bit 63 of base = ESCAPE: some entry of the line does not fit 16 bits (the suffixes of a high-copy repeat).
// The other bits of an escape line's base then name its group in the SIDE array: 32 flat 16-byte {l, h} entries (30
// used), so that a query of such a line costs ONE more line fetch -- like any search step -- instead of a search from
// scratch (round 4; the reference's constrain_range costs the same for any range width, rle_bwt.rs:202-287).
// 4.27 bytes per entry instead of 16: two more table levels in the same HBM.
constexpr uint32_t kPackedPerLine = 30;
constexpr uint32_t kSidePerLine = 32;  // 16-byte slots of a side group (512 bytes: four lines)
constexpr uint64_t kPackedEscape = 1ull << 63;

// The raw bits of one table entry, as loaded (the loads may stay in flight):
//   flat:   {l_lo, l_hi, h_lo, h_hi};   packed: {base_lo, base_hi, entry, slot in the line}
__device__ __forceinline__ uint4 table_fetch(const TableEnv &env, uint64_t tidx) {
    if (!env.packed) return env.table[tidx];
    const uint64_t line = tidx / kPackedPerLine;
    const uint32_t slot = uint32_t(tidx - line * kPackedPerLine);
    const uint2 base = *reinterpret_cast<const uint2 *>(env.table + line * 8);
    const uint32_t e = reinterpret_cast<const uint32_t *>(env.table + line * 8)[2u + slot];
    return make_uint4(base.x, base.y, e, slot);
}

// -> range; returns false for an entry of an escape line: l = index of its 16-byte entry in the side array then
// (meaningful when env.side != nullptr; without a side array the query searches from [0, total))
__device__ __forceinline__ bool table_decode(const TableEnv &env, const uint4 raw, uint64_t &l, uint64_t &h) {
    if (!env.packed) {
        l = (uint64_t(raw.y) << 32) | raw.x;
        h = (uint64_t(raw.w) << 32) | raw.z;
        return true;
    }
    const uint64_t base = (uint64_t(raw.y) << 32) | raw.x;
    if ((base & kPackedEscape) != 0ull) {
        l = (base & ~kPackedEscape) * kSidePerLine + raw.w;
        h = l;
        return false;
    }
    l = base + (raw.z & 0xFFFFu);
    h = l + (raw.z >> 16);
    return true;
}

__device__ __forceinline__ uint32_t ascii_to_code(uint32_t c) {
    if (c == 0x24u) return 0u;  // '$'
    c &= 0xDFu;                 // fold lower case
    return c == 0x41u ? 1u : c == 0x43u ? 2u : c == 0x47u ? 3u : c == 0x54u ? 5u : 4u;
}
__device__ __forceinline__ uint32_t complement_code(uint32_t s) {  // $ACGNT -> $TGCNA; 6,7 stay invalid
    return s == 1u ? 5u : s == 5u ? 1u : s == 2u ? 3u : s == 3u ? 2u : s;
}

// drops the `bits` lowest bits of the packed symbols (3 or 6: one or two consumed symbols)
template <int kWords>
__device__ __forceinline__ void consume_symbols(uint32_t (&w)[kWords], int bits) {
#pragma unroll
    for (int i = 0; i + 1 < kWords; ++i) w[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], bits);
    w[kWords - 1] >>= bits;
}

// Piece `piece` (16 bytes) of a tile's `nbytes` contiguous query bytes starting at src_bytes
// (16-byte aligned); the batch's ragged end is read bytewise so that nothing past the caller's
// buffer is touched.
__device__ __forceinline__ uint4 load_piece(const uint8_t *__restrict__ src_bytes, uint32_t nbytes, uint32_t piece) {
    if (piece * 16u >= nbytes) return make_uint4(0, 0, 0, 0);
    if (piece * 16u + 16u <= nbytes) return *reinterpret_cast<const uint4 *>(src_bytes + piece * 16u);
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    for (uint32_t b = piece * 16u; b < nbytes; ++b) {
        const uint32_t v = uint32_t(src_bytes[b]) << ((b & 3u) * 8u), word = (b & 15u) >> 2;
        if (word == 0u) w0 |= v; else if (word == 1u) w1 |= v; else if (word == 2u) w2 |= v; else w3 |= v;
    }
    return make_uint4(w0, w1, w2, w3);
}

// count of query v (global index) goes to its place in the caller's output
template <bool kReads>
__device__ __forceinline__ void store_count(const QuerySource &src, uint64_t v, uint64_t value) {
    if (!kReads) {
        // (the counts of a placed batch -- order.hip -- are scattered 8-byte stores; tried for them in round 4: non-temporal
        // stores 18.1 ms against 18.5 ms plain on C4, write-through past the L2 (sc1) 18.7: not what they cost)
        src.out_fwd[v] = value;
    } else if (src.strands == 3u) {
        ((v & 1u) ? src.out_rc : src.out_fwd)[v >> 1] = value;
    } else {
        (src.strands == 2u ? src.out_rc : src.out_fwd)[v] = value;
    }
}

// ---- one lane = one query: setup in three pieces -----------------------------------------------
// pack_query:   read the k symbols (matrix mode: this lane's staged bytes in LDS; reads mode: the
//               window of global query v), note symbols >= 6 (the reference asserts,
//               msbwt_core.rs:127), pack 3 bits/symbol in search order (last symbol first) and form
//               the suffix-table index of the first `depth` steps.
// table lookup: env.table[tidx] behind the presence filter -- issued by the caller, so that the
//               lanes kernel can leave the load in flight across a search step.
// unpack_words: drop the table's symbols, hand the rest over as kWords dwords.
// bits |= value << pos over an array of u64 words (pos is a compile-time constant after unrolling; value < 8)
template <int kBits>
__device__ __forceinline__ void constexpr_shift_or(uint64_t (&bits)[kBits], uint32_t value, uint32_t pos) {
    const uint32_t word = pos >> 6, off = pos & 63u;
#pragma unroll
    for (int j = 0; j < kBits; ++j) {
        if (word == uint32_t(j)) bits[j] |= uint64_t(value) << off;
        if (j > 0 && word == uint32_t(j - 1) && off > 61u) bits[j] |= uint64_t(value) >> (64u - off);
    }
}

// bits |= value << pos for a 12-bit value (pos is a compile-time constant after unrolling)
template <int kBits>
__device__ __forceinline__ void constexpr_shift_or12(uint64_t (&bits)[kBits], uint32_t value, uint32_t pos) {
    const uint32_t word = pos >> 6, off = pos & 63u;
#pragma unroll
    for (int j = 0; j < kBits; ++j) {
        if (word == uint32_t(j)) bits[j] |= uint64_t(value) << off;
        if (j > 0 && word == uint32_t(j - 1) && off > 52u) bits[j] |= uint64_t(value) >> (64u - off);
    }
}

template <int kWords>
struct PackedQuery {
    static constexpr int kBits = (kWords + 1) / 2;
    uint64_t bits[kBits];  // symbol of step t (t = 0 first) at bits [3t, 3t+3) of the little-endian words
    uint64_t tidx;         // table index of steps 0..depth-1 (A C G T -> 0..3, step t at bits [2t, 2t+2))
    bool bad;              // a symbol code >= 6
    bool acgt;             // steps 0..depth-1 are all ACGT: the table applies
};

// Bytes kept free in front of a tile's staged query bytes (matrix mode): pack_query reads the 32 or 64
// bytes that END at a query's last symbol, which for the tile's first queries start before the tile.
constexpr uint32_t kStageLead = 64;

// bytes `first`.. of a block are wanted: the mask of dword d's wanted bytes (wave-uniform arguments: scalar work)
__device__ __forceinline__ uint32_t tail_byte_mask(uint32_t d, uint32_t first) {
    const int lo = int(first) - int(4u * d);  // first wanted byte inside this dword
    return lo <= 0 ? ~0u : (lo >= 4 ? 0u : (~0u << (8 * lo)));
}

// pack_query for a k-byte ROW of symbol codes in LDS (matrix queries, LDS-staged read windows), four symbols per
// instruction (round 3; the byte-at-a-time version below cost ~5 of a random query's 6.5 VALU wave-instructions).
// The block of kBlock bytes that ENDS at the row's last byte is read as dwords; step t (t = 0 first) is byte
// kBlock-1-t, i.e. byte 3 - (t & 3) of dword D-1-(t >> 2):
//   symbols >= 6   per byte, bit 7 of ((x + 0x7A) | x);
//   3-bit packing  four bytes b0..b3 of a dword become the 12-bit group b3 | b2 << 3 | b1 << 6 | b0 << 9 (search
//                  order: the highest address first) in two shift-or-mask rounds, group u lands at bit 12 u;
//   table index    bytes outside the table's reach read as 'A' so that nothing borrows from them; A C G T -> 0..3 is
//                  y - 1 - (y >> 2) per byte, the four 2-bit codes of a dword are gathered the same way; a '$' / 'N'
//                  inside the reach (low two bits zero) turns the table off for this query.
// kReach: how many steps deep a table index may reach in this kernel (24: direct tables and the 24-bit-tag sparse table, as until round 5;
// 32: the deeper sparse layouts) -- the index costs a dozen instructions per four symbols, and the lookup-heavy lines of small indexes
// notice them (round 6, same box: C2 read-derived 21-mers 0.273 -> 0.297 ms with every kernel paying for 32).
template <int kWords, int kReach = 32>
__device__ __forceinline__ void pack_row_swar(uint32_t k, uint32_t depth, const uint8_t *row, PackedQuery<kWords> &pq) {
    constexpr int kBits = PackedQuery<kWords>::kBits;
    constexpr uint32_t kBlock = kWords == 3 ? 32u : 64u, D = kBlock / 4u;
    uint32_t raw[D];
    __builtin_memcpy(raw, row + k - kBlock, kBlock);  // LDS; the caller keeps kStageLead bytes in front of the tile
    const uint32_t first_k = kBlock - k, first_d = kBlock - min(depth, k);
#pragma unroll
    for (int j = 0; j < kBits; ++j) pq.bits[j] = 0;
    uint32_t bad = 0, nonacgt = 0;
    uint64_t tidx = 0;
#pragma unroll
    for (uint32_t u = 0; u < D; ++u) {  // steps 4u .. 4u+3
        const uint32_t d = D - 1u - u;
        const uint32_t x = raw[d] & tail_byte_mask(d, first_k);
        bad |= ((x + 0x7A7A7A7Au) | x) & 0x80808080u;
        const uint32_t y = x & 0x07070707u;
        const uint32_t z = ((y >> 8) | (y << 3)) & 0x003F003Fu;
        const uint32_t g = ((z >> 16) | (z << 6)) & 0xFFFu;
        constexpr_shift_or12<kBits>(pq.bits, g, 12u * u);
        if (u < uint32_t(kReach) / 4u) {  // the tables of this kernel reach at most kReach steps deep (sparse_table.hpp)
            const uint32_t dm = tail_byte_mask(d, first_d);
            const uint32_t ys = (y & dm) | (0x01010101u & ~dm);
            const uint32_t low2 = ys & 0x03030303u;
            nonacgt |= (low2 - 0x01010101u) & ~low2 & 0x80808080u;  // some byte is 0 or 4 ('$' / 'N'; >= 6 is `bad`)
            const uint32_t c = ys - 0x01010101u - ((ys >> 2) & 0x01010101u);
            const uint32_t z2 = ((c >> 8) | (c << 2)) & 0x000F000Fu;
            tidx |= uint64_t(((z2 >> 16) | (z2 << 4)) & 0xFFu) << (8u * u);
        }
    }
    pq.tidx = tidx & ((1ull << (2u * min(depth, uint32_t(kReach >= 32 ? 31 : kReach)))) - 1ull);
    pq.bad = bad != 0u;
    pq.acgt = nonacgt == 0u;
}

// pack_query for a query handed over as 2-bit codes (QuerySource::packed): `lo` = symbols k-1 .. k-32 from bit 0 up (search
// order), `hi` = symbols k-33 .. (kWords == 6 only).  Nothing to validate; the table index is the low 2 x depth bits as they are.
template <int kWords>
__device__ __forceinline__ void pack_two_bit(uint32_t k, uint32_t depth, uint64_t lo, uint64_t hi, PackedQuery<kWords> &pq) {
    constexpr int kBits = PackedQuery<kWords>::kBits;
    constexpr uint32_t kMaxK = kWords == 3 ? 32u : 64u;
#pragma unroll
    for (int j = 0; j < kBits; ++j) pq.bits[j] = 0;
    // symbols beyond k read as 'A' (code 1): never searched (rem counts the steps), but the words stay well-formed
    if (k < 32u) lo &= (1ull << (2u * k)) - 1ull;
    if (k < 64u && k > 32u) hi &= (1ull << (2u * (k - 32u))) - 1ull;
    if (k <= 32u) hi = 0ull;
#pragma unroll
    for (uint32_t u = 0; u < kMaxK / 4u; ++u) {  // four symbols per round: 8 bits -> the 12-bit group of steps 4u .. 4u+3
        const uint32_t x = uint32_t((u < 8u ? lo : hi) >> (8u * (u & 7u))) & 0xFFu;
        const uint32_t y = (x & 0x03u) | ((x & 0x0Cu) << 1) | ((x & 0x30u) << 2) | ((x & 0xC0u) << 3);  // 2-bit fields on a 3-bit pitch
        const uint32_t g = y + 0x249u + ((y & (y >> 1)) & 0x249u);  // A C G T -> 1 2 3 5: + 1, and + 1 more where the code is 3
        constexpr_shift_or12<kBits>(pq.bits, g, 12u * u);
    }
    const uint32_t reach = min(depth, k);
    pq.tidx = reach >= 32u ? lo : (lo & ((1ull << (2u * reach)) - 1ull));
    pq.bad = false;
    pq.acgt = true;
}

template <bool kReads, int kWords, int kReach = 32>
__device__ __forceinline__ void pack_query(const QuerySource &src, uint32_t depth, const uint8_t *staged, uint64_t v,
                                           PackedQuery<kWords> &pq) {
    if constexpr (!kReads) {
        (void)v;
        pack_row_swar<kWords, kReach>(src.k, depth, staged, pq);
        return;
    }
    constexpr int kBits = PackedQuery<kWords>::kBits;
    constexpr uint32_t kBlock = kWords == 3 ? 32u : 64u;  // >= k: the bytes of a query arrive as ONE block of wide loads
    const uint32_t k = src.k;
    // The search consumes a k-mer from its last symbol: step t reads byte k-1-t of a forward query --
    // byte kBlock-1-t of the block that ENDS at the query's last byte -- and comp(byte t) of a window
    // taken as its reverse complement (q'[j] = comp(window[k-1-j])): byte t of the block that STARTS at
    // the window.  Either way the byte of step t sits at a compile-time place in registers: no
    // byte-sized load (and its wait) per symbol.
    uint32_t raw[kBlock / 4];
    bool rc = false;
    if (!kReads) {
        __builtin_memcpy(raw, staged + k - kBlock, kBlock);  // LDS; the caller keeps kStageLead bytes in front of the tile
    } else {  // window g of read r, forward or reverse-complemented
        const uint64_t g = src.strands == 3u ? (v >> 1) : v;
        rc = src.strands == 3u ? (v & 1u) != 0 : src.strands == 2u;
        uint64_t first, data_len;  // the window's first byte; bytes in the caller's buffer
        if (src.win_off == nullptr) {
            first = (g / src.windows) * src.read_len + (g % src.windows);
            data_len = src.n_reads * src.read_len;
        } else {  // last read whose first window is <= g (reads shorter than k own none)
            uint64_t lo = 0, hi = src.n_reads;
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (src.win_off[mid] <= g) lo = mid; else hi = mid;
            }
            first = src.read_off[lo] + (g - src.win_off[lo]);
            data_len = src.read_off[src.n_reads];
        }
        const uint64_t block = rc ? first : first + k - kBlock;  // may lie partly outside the buffer (wraps below 0)
        if (block + kBlock <= data_len && block <= first) {
            __builtin_memcpy(raw, src.data + block, kBlock);
        } else {  // the batch's first / last windows: bytewise, nothing outside the caller's buffer is touched
#pragma unroll
            for (uint32_t i = 0; i < kBlock / 4; ++i) raw[i] = 0;
            for (uint32_t i = 0; i < kBlock; ++i) {
                const uint64_t at = block + i;  // wraps for bytes in front of the buffer
                const uint32_t byte = at < data_len ? uint32_t(src.data[at]) : 0u;
#pragma unroll
                for (uint32_t j = 0; j < kBlock / 4; ++j)
                    if ((i >> 2) == j) raw[j] |= byte << ((i & 3u) * 8u);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kBits; ++j) pq.bits[j] = 0;
    uint32_t bad = 0, acgt = 1;
    uint64_t tidx = 0;
#pragma unroll
    for (uint32_t t = 0; t < kBlock; ++t) {
        if (t < k) {  // wave-uniform
            const uint32_t from_end = (raw[(kBlock - 1u - t) >> 2] >> (((kBlock - 1u - t) & 3u) * 8u)) & 0xFFu;
            uint32_t s = from_end;
            if (kReads) {
                const uint32_t from_start = (raw[t >> 2] >> ((t & 3u) * 8u)) & 0xFFu;
                s = rc ? from_start : from_end;
                if (src.ascii) s = ascii_to_code(s);
                if (rc) s = complement_code(s);
            }
            bad |= (s >= 6u) ? 1u : 0u;
            constexpr_shift_or<kBits>(pq.bits, s & 7u, 3u * t);
            if (t < depth) {
                acgt &= acgt_bit(s);
                tidx |= uint64_t(acgt_code(s) & 3u) << (2u * t);
            }
        }
    }
    pq.tidx = tidx;
    pq.bad = bad != 0u;
    pq.acgt = acgt != 0u;
}

// the remaining symbols after `skip` steps (0 or the table depth), as kWords dwords
template <int kWords>
__device__ __forceinline__ void unpack_words(const PackedQuery<kWords> &pq, uint32_t skip, uint32_t (&w)[kWords]) {
    constexpr int kBits = PackedQuery<kWords>::kBits;
    uint64_t bits[kBits];
    // 0, or 3..87 (the sparse table reaches 29 symbols deep): whole words first, then the bits (wave-uniform)
    const bool whole = 3u * skip >= 64u;
    const uint32_t sh = 3u * skip - (whole ? 64u : 0u);
#pragma unroll
    for (int j = 0; j < kBits; ++j) {
        const uint64_t lo = whole ? (j + 1 < kBits ? pq.bits[j + 1] : 0ull) : pq.bits[j];
        const uint64_t hi = whole ? (j + 2 < kBits ? pq.bits[j + 2] : 0ull) : (j + 1 < kBits ? pq.bits[j + 1] : 0ull);
        bits[j] = sh == 0u ? lo : ((lo >> sh) | (hi << (64u - sh)));
    }
#pragma unroll
    for (int i = 0; i < kWords; ++i) w[i] = uint32_t(bits[i >> 1] >> ((i & 1) * 32));
}

// The whole setup of one query, table lookup included (the 8-lane-group kernel).  Returns true
// when a search is still needed (l, h, w, rem filled), false when `result` already is the count.
template <bool kReads, int kWords>
__device__ __forceinline__ bool prepare_query(const QuerySource &src, const TableEnv &env, const uint8_t *staged, uint64_t v,
                                              bool filter_now, uint32_t *__restrict__ flags, uint64_t &l, uint64_t &h,
                                              uint32_t (&w)[kWords], uint32_t &rem, uint64_t &result, bool &looked_up,
                                              bool &passed) {
    PackedQuery<kWords> pq;
    pack_query<kReads, kWords>(src, env.depth, staged, v, pq);
    l = 0;
    h = env.total;
    rem = src.k;
    if (pq.bad) {
        result = ~0ull;
        atomicOr(flags, kFlagInvalidSymbol);
        return false;
    }
    uint32_t skip = 0;
    if (env.use_table && pq.acgt) {
        // L2-resident presence bit first: an absent suffix never touches the table line
        bool maybe = true;
        if (filter_now) {
            const uint32_t fi = uint32_t(pq.tidx) & env.filter_mask;
            maybe = ((env.filter[fi >> 5] >> (fi & 31u)) & 1u) != 0u;
            looked_up = true;
            passed = maybe;
        }
        if (!maybe) {
            l = h = 0;  // empty range: count 0
            skip = env.depth;
            rem = src.k - env.depth;
        } else if (table_decode(env, table_fetch(env, pq.tidx), l, h)) {
            skip = env.depth;
            rem = src.k - env.depth;
        } else if (env.side != nullptr) {  // escape line of a packed table: its flat entry, one dependent load further
            const uint4 e = env.side[l];
            l = (uint64_t(e.y) << 32) | e.x;
            h = (uint64_t(e.w) << 32) | e.z;
            skip = env.depth;
            rem = src.k - env.depth;
        } else {  // no side array: this query searches from scratch
            l = 0;
            h = env.total;
        }
    }
    if (rem == 0u || l == h) {
        result = h - l;
        return false;
    }
    unpack_words<kWords>(pq, skip, w);
    return true;
}

// 256 CUs x blocks_per_cu resident blocks of 256 threads fill the chip; smaller batches get just enough blocks
inline uint32_t grid_for(uint64_t threads_wanted, uint32_t blocks_per_cu = 8) {
    const uint64_t blocks = (threads_wanted + 255) / 256, cap = 256ull * blocks_per_cu;
    return uint32_t(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

}  // namespace

// lanes.hip: the one-query-per-lane search kernel (LDS-staged lines), 1 <= src.k <= 64; matrix
// mode needs a 16-byte-aligned batch.
hipError_t launch_lanes(const IndexView &ix, const QuerySource &src, bool reads, bool pair, uint32_t *flags,
                        hipStream_t stream);

}  // namespace msbwt
