#include "rle_codec.hpp"

#include <cstring>

namespace msbwt {

bool compute_totals(const uint8_t *bytes, size_t n, Totals *out) {
    std::memset(out, 0, sizeof(*out));
    bool ok = true;
    for_each_run(bytes, n, [&](uint8_t sym, uint64_t len) {
        if (sym >= kAlphabet) { ok = false; return; }
        out->symbol_counts[sym] += len;
        out->runs += 1;
    });
    uint64_t acc = 0;
    for (int s = 0; s < kAlphabet; ++s) {
        out->start_index[s] = acc;
        acc += out->symbol_counts[s];
        out->end_index[s] = acc;
    }
    out->total = acc;
    return ok;
}

namespace {

struct AsciiTables {
    uint8_t strict[256];   // $ACGNT -> 0..5, everything else 255   (bwt_converter.rs:27-31)
    uint8_t lenient[256];  // $ACGTacgt -> code, everything else 4  (string_util.rs:15-32)
    AsciiTables() {
        std::memset(strict, 255, sizeof strict);
        std::memset(lenient, 4, sizeof lenient);
        const char *alpha = "$ACGNT";
        for (int i = 0; i < kAlphabet; ++i) {
            strict[uint8_t(alpha[i])] = uint8_t(i);
            lenient[uint8_t(alpha[i])] = uint8_t(i);
            if (alpha[i] != '$') lenient[uint8_t(alpha[i] + 32)] = uint8_t(i);
        }
    }
};
const AsciiTables kTables;

inline void push_digits(uint8_t code, uint64_t count, std::vector<uint8_t> *out) {
    for (; count > 0; count >>= 5) out->push_back(uint8_t(code | ((count & 31u) << 3)));
}

}  // namespace

bool encode_text(const uint8_t *ascii, size_t n, std::vector<uint8_t> *out) {
    out->clear();
    uint8_t open_code = 0;  // an empty '$' run is open at the start (writes nothing if unused)
    uint64_t open_len = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t code = kTables.strict[ascii[i]];
        if (code == 255) {
            if (ascii[i] == '\n') continue;  // newlines never break a run
            return false;
        }
        if (code == open_code) {
            ++open_len;
        } else {
            push_digits(open_code, open_len, out);
            open_code = code;
            open_len = 1;
        }
    }
    push_digits(open_code, open_len, out);
    return true;
}

void encode_runs(const uint8_t *syms, const uint64_t *counts, size_t nruns, std::vector<uint8_t> *out) {
    out->clear();
    for (size_t i = 0; i < nruns; ++i) push_digits(syms[i], counts[i], out);
}

void ascii_to_codes(const uint8_t *ascii, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) out[i] = kTables.lenient[ascii[i]];
}

void codes_to_ascii(const uint8_t *codes, size_t n, uint8_t *out) {
    static const char alpha[kAlphabet + 1] = "$ACGNT";
    for (size_t i = 0; i < n; ++i) out[i] = uint8_t(codes[i] < kAlphabet ? alpha[codes[i]] : '?');
}

void reverse_complement_codes(const uint8_t *codes, size_t n, uint8_t *out) {
    static const uint8_t comp[8] = {0, 5, 3, 2, 4, 1, 6, 7};  // $<->$ A<->T C<->G N<->N
    for (size_t i = 0; i < n; ++i) out[i] = comp[codes[n - 1 - i] & 7u];
}

}  // namespace msbwt
