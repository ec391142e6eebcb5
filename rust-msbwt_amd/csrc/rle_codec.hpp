// Host-side codecs for the run-length-encoded BWT byte stream (load path, not the hot path).
// Format: src/msbwt_core.rs:3-14 of the reference -- byte = symbol | digit << 3, consecutive
// bytes of one symbol are base-32 digits of one run, least significant first.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace msbwt {

constexpr int kAlphabet = 6;  // $ A C G N T

struct Totals {
    uint64_t symbol_counts[kAlphabet];
    uint64_t start_index[kAlphabet];  // exclusive prefix sum in alphabet order
    uint64_t end_index[kAlphabet];
    uint64_t total;
    uint64_t runs;  // maximal same-symbol byte groups
};

// Calls fn(symbol, length) for every run of the stream, in order.  A run is a maximal group
// of consecutive bytes with the same symbol; its length is sum(digit_i * 32^i) (may be 0).
template <class Fn>
inline void for_each_run(const uint8_t *bytes, size_t n, Fn &&fn) {
    size_t i = 0;
    while (i < n) {
        const uint8_t sym = bytes[i] & 7u;
        uint64_t len = 0, weight = 1;
        do {
            len += uint64_t(bytes[i] >> 3) * weight;
            weight <<= 5;
            ++i;
        } while (i < n && (bytes[i] & 7u) == sym);
        fn(sym, len);
    }
}

// Symbol totals of the stream (what rle_bwt.rs:352-384 computes at load time).
// Returns false if a byte carries symbol code 6 or 7.
bool compute_totals(const uint8_t *bytes, size_t n, Totals *out);

// ASCII "$ACGNT" (+ ignored '\n') -> RLE bytes.  Returns false on any other byte.
bool encode_text(const uint8_t *ascii, size_t n, std::vector<uint8_t> *out);
// (symbol, count) runs -> RLE bytes (digits of each run; a zero-length run writes nothing).
void encode_runs(const uint8_t *syms, const uint64_t *counts, size_t nruns, std::vector<uint8_t> *out);

void ascii_to_codes(const uint8_t *ascii, size_t n, uint8_t *out);
void codes_to_ascii(const uint8_t *codes, size_t n, uint8_t *out);
void reverse_complement_codes(const uint8_t *codes, size_t n, uint8_t *out);

}  // namespace msbwt
