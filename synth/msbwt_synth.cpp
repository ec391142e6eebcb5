// Synthetic workload generators for tests and bench.py (TEST / BENCH INFRASTRUCTURE, not
// part of the product): a random genome, Illumina-like reads, the multi-string BWT of a read
// set in the reference's ordering, and random k-mer queries.  Everything is integer and
// seeded with splitmix64, so the CPU box and the GPU box produce identical inputs.
//
// MSBWT ordering (what the reference's msbwt2-build produces and bwt_util::naive_bwt
// defines, src/bwt_util.rs:154-171): suffixes of read+'$' sorted with $ < A < C < G < N < T;
// equal suffixes are ordered by their owning read's lexicographic rank.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <thread>
#include <vector>

namespace {

inline uint64_t splitmix64(uint64_t &state) {
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// stateless form: the i-th output of the stream seeded with `seed`
inline uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t s = seed + i * 0x9E3779B97F4A7C15ull;
    return splitmix64(s);
}

const uint8_t kBase[4] = {1, 2, 3, 5};  // A C G T

int g_default_threads = 0;  // synth_set_threads: bench.py bounds the host threads of a rank (nproc / world)

int pick_threads(int threads) {
    if (threads > 0) return threads;
    if (g_default_threads > 0) return g_default_threads;
    unsigned hw = std::thread::hardware_concurrency();
    return int(std::max(1u, std::min(hw, 16u)));
}

template <class Fn>
void parallel_for(int threads, size_t n, Fn fn) {  // fn(thread, begin, end)
    threads = int(std::min<size_t>(size_t(threads), std::max<size_t>(n, 1)));
    if (threads <= 1) { fn(0, size_t(0), n); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([=] { fn(t, n * size_t(t) / size_t(threads), n * size_t(t + 1) / size_t(threads)); });
    for (auto &th : pool) th.join();
}

struct Suffix {
    uint64_t key;  // first 21 symbols, 3 bits each, first symbol most significant
    uint64_t pos;  // position in the concatenated text
};

}  // namespace

extern "C" {

// genome[i] in {A,C,G,T} codes, uniform i.i.d.
void synth_genome(uint64_t length, uint64_t seed, uint8_t *out) {
    parallel_for(pick_threads(0), size_t(length), [=](int, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) out[i] = kBase[splitmix64_at(seed, i) & 3u];
    });
}

// A genome WITH REPEATS (round 4; the human genome is about half repeats and the search cost of a k-mer follows its copy
// number): a uniform i.i.d. background, onto which are pasted, in this order (later pastes overwrite earlier ones),
//   * interspersed families: family f has a random consensus of fam_len[f] bases and fam_copies[f] copies at uniform
//     positions; a copy is the whole consensus, or (fam_trunc[f] != 0) a 5'-truncated piece of uniform length in
//     [fam_len/20, fam_len] (L1-like); half of the copies are reverse-complemented; every base of a copy is substituted
//     with the copy's own divergence, drawn uniformly from [div_lo, div_hi] (parts per million) -- young and old copies;
//   * satellite arrays: tandem repeats of a sat_monomer-base monomer (one monomer per array family of ~100 kb arrays),
//     neighbouring monomers diverged by sat_div_ppm, sat_bases bases in all;
//   * microsatellites / homopolymer tracts: units of 1..6 bases repeated over 20..80 bases, micro_bases bases in all.
// out_class[i] (optional): 0 = background, 1 + f = family f, 250 = satellite, 251 = microsatellite.
// Everything is integer arithmetic on one splitmix64 stream: the CPU box and the GPU box build the same genome.
void synth_repeat_genome(uint64_t length, uint64_t seed, uint32_t nfam, const uint32_t *fam_len, const uint32_t *fam_copies,
                         const uint32_t *fam_div_lo_ppm, const uint32_t *fam_div_hi_ppm, const uint8_t *fam_trunc,
                         uint64_t sat_bases, uint32_t sat_monomer, uint32_t sat_div_ppm, uint64_t micro_bases, uint8_t *out,
                         uint8_t *out_class) {
    synth_genome(length, seed, out);
    if (out_class) std::memset(out_class, 0, size_t(length));
    uint64_t st = seed ^ 0x5EED5EED5EED5EEDull;
    auto mutate = [&](uint8_t c, uint32_t ppm) -> uint8_t {  // substitution with probability ppm / 1e6
        const uint64_t x = splitmix64(st);
        if (uint32_t(x % 1000000u) >= ppm) return c;
        uint32_t idx = 0;
        while (kBase[idx] != c) ++idx;
        return kBase[(idx + 1 + uint32_t(x >> 32) % 3u) & 3u];
    };
    auto comp = [](uint8_t c) -> uint8_t { return c == 1 ? 5 : c == 5 ? 1 : c == 2 ? 3 : 2; };
    std::vector<uint8_t> cons;
    for (uint32_t f = 0; f < nfam; ++f) {
        const uint32_t L = fam_len[f];
        if (L == 0 || L >= length) continue;
        cons.resize(L);
        for (uint32_t i = 0; i < L; ++i) cons[i] = kBase[splitmix64(st) & 3u];
        for (uint32_t c = 0; c < fam_copies[f]; ++c) {
            uint32_t piece = L;
            if (fam_trunc[f]) piece = std::max<uint32_t>(L / 20u, 1u) + uint32_t(splitmix64(st) % (L - std::max<uint32_t>(L / 20u, 1u) + 1u));
            const uint32_t from = L - piece;  // 5'-truncated: the 3' end is kept
            const uint64_t at = splitmix64(st) % (length - piece + 1);
            const bool rc = (splitmix64(st) & 1u) != 0;
            const uint32_t span = fam_div_hi_ppm[f] - fam_div_lo_ppm[f];
            const uint32_t div = fam_div_lo_ppm[f] + (span ? uint32_t(splitmix64(st) % (span + 1u)) : 0u);
            for (uint32_t i = 0; i < piece; ++i) {
                const uint8_t b = rc ? comp(cons[from + piece - 1 - i]) : cons[from + i];
                out[at + i] = mutate(b, div);
                if (out_class) out_class[at + i] = uint8_t(1 + f);
            }
        }
    }
    if (sat_monomer > 0 && sat_monomer < length) {
        std::vector<uint8_t> mono(sat_monomer), cur(sat_monomer);
        uint64_t placed = 0;
        while (placed < sat_bases) {
            for (uint32_t i = 0; i < sat_monomer; ++i) mono[i] = kBase[splitmix64(st) & 3u];
            const uint64_t arr = std::min<uint64_t>(std::min<uint64_t>(sat_bases - placed, 50000 + splitmix64(st) % 100000), length / 2);
            const uint64_t at = splitmix64(st) % (length - arr + 1);
            cur = mono;
            for (uint64_t i = 0; i < arr; ++i) {
                const uint32_t j = uint32_t(i % sat_monomer);
                if (j == 0 && i) for (uint32_t t = 0; t < sat_monomer; ++t) cur[t] = mutate(mono[t], sat_div_ppm);
                out[at + i] = cur[j];
                if (out_class) out_class[at + i] = 250;
            }
            placed += std::max<uint64_t>(arr, 1);
        }
    }
    for (uint64_t placed = 0; placed < micro_bases;) {
        const uint32_t unit = 1u + uint32_t(splitmix64(st) % 6u), tract = 20u + uint32_t(splitmix64(st) % 61u);
        if (tract >= length) break;
        uint8_t u[6];
        for (uint32_t i = 0; i < unit; ++i) u[i] = kBase[splitmix64(st) & 3u];
        const uint64_t at = splitmix64(st) % (length - tract + 1);
        for (uint32_t i = 0; i < tract; ++i) {
            out[at + i] = u[i % unit];
            if (out_class) out_class[at + i] = 251;
        }
        placed += tract;
    }
}

// n reads of `len` bases: uniform start, forward strand, each base substituted by one of
// the other three with probability err_per_million / 1e6.  out: n x len codes.
void synth_reads(const uint8_t *genome, uint64_t glen, uint64_t n, uint32_t len, uint64_t seed,
                 uint32_t err_per_million, uint8_t *out) {
    parallel_for(pick_threads(0), size_t(n), [=](int, size_t b, size_t e) {
        for (size_t r = b; r < e; ++r) {
            uint64_t st = seed ^ (0xD1B54A32D192ED03ull * (r + 1));
            const uint64_t start = splitmix64(st) % (glen - len + 1);
            uint8_t *dst = out + r * len;
            for (uint32_t i = 0; i < len; ++i) {
                uint8_t c = genome[start + i];
                const uint64_t x = splitmix64(st);
                if (uint32_t(x % 1000000u) < err_per_million) {
                    const uint32_t which = uint32_t(x >> 32) % 3u;  // one of the other three
                    uint32_t idx = 0;
                    while (kBase[idx] != c) ++idx;
                    c = kBase[(idx + 1 + which) & 3u];
                }
                dst[i] = c;
            }
        }
    });
}

// n x k uniform k-mers over {A,C,G,T}
void synth_random_kmers(uint64_t n, uint32_t k, uint64_t seed, uint8_t *out) {
    parallel_for(pick_threads(0), size_t(n), [=](int, size_t b, size_t e) {
        for (size_t q = b; q < e; ++q) {
            uint64_t st = seed ^ (0xA0761D6478BD642Full * (q + 1));
            uint64_t bits = 0;
            int left = 0;
            for (uint32_t i = 0; i < k; ++i) {
                if (left == 0) { bits = splitmix64(st); left = 32; }
                out[q * k + i] = kBase[bits & 3u];
                bits >>= 2;
                --left;
            }
        }
    });
}

// Multi-string BWT of n reads (codes 1..5, no 0) given as a flat buffer with offsets[n+1].
// Writes one symbol code per BWT position to out_symbols (sum(len)+n of them).
// Returns 0, or -1 on bad input / out of memory.
int synth_build_msbwt(const uint8_t *reads, const uint64_t *offsets, uint64_t n, uint8_t *out_symbols,
                      int threads) {
    threads = pick_threads(threads);
    const bool verbose = std::getenv("SYNTH_VERBOSE") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[synth] %-10s %.2fs\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    };
    for (uint64_t i = 0; i < offsets[n]; ++i)
        if (reads[i] == 0 || reads[i] > 5) return -1;
    // 1. lexicographic order of the reads ('$' terminator smallest => shorter prefix first)
    std::vector<uint64_t> order(n);
    std::iota(order.begin(), order.end(), uint64_t(0));
    auto read_less = [&](uint64_t a, uint64_t b) {
        const uint64_t la = offsets[a + 1] - offsets[a], lb = offsets[b + 1] - offsets[b];
        const int c = std::memcmp(reads + offsets[a], reads + offsets[b], size_t(std::min(la, lb)));
        if (c) return c < 0;
        if (la != lb) return la < lb;
        return a < b;
    };
    {   // parallel: sort slices, then merge pairwise
        std::vector<size_t> cuts;
        for (int t = 0; t <= threads; ++t) cuts.push_back(size_t(n) * size_t(t) / size_t(threads));
        parallel_for(threads, size_t(threads), [&](int, size_t b, size_t e) {
            for (size_t s = b; s < e; ++s) std::sort(order.begin() + cuts[s], order.begin() + cuts[s + 1], read_less);
        });
        for (size_t width = 1; width < size_t(threads); width *= 2) {
            std::vector<std::thread> pool;
            for (size_t s = 0; s + width < size_t(threads); s += 2 * width) {
                const size_t lo = cuts[s], mid = cuts[s + width], hi = cuts[std::min(s + 2 * width, size_t(threads))];
                pool.emplace_back([&, lo, mid, hi] {
                    std::inplace_merge(order.begin() + lo, order.begin() + mid, order.begin() + hi, read_less);
                });
            }
            for (auto &th : pool) th.join();
        }
    }
    lap("sort reads");
    // 2. text = sorted reads, each followed by a 0 terminator
    const uint64_t N = offsets[n] + n;
    std::vector<uint64_t> tstart(n + 1);
    tstart[0] = 0;
    for (uint64_t r = 0; r < n; ++r) tstart[r + 1] = tstart[r] + (offsets[order[r] + 1] - offsets[order[r]]) + 1;
    // plain malloc: the pages are first touched by the parallel loops below (a value-
    // initialising container would fault all of them in on one thread)
    struct Free { void operator()(void *p) const { std::free(p); } };
    std::unique_ptr<uint8_t, Free> text_mem(static_cast<uint8_t *>(std::malloc(N + 32)));
    std::unique_ptr<Suffix, Free> a_mem(static_cast<Suffix *>(std::malloc((N + 1) * sizeof(Suffix))));
    std::unique_ptr<Suffix, Free> b_mem(static_cast<Suffix *>(std::malloc((N + 1) * sizeof(Suffix))));
    if (!text_mem || !a_mem || !b_mem) return -1;
    uint8_t *text = text_mem.get();
    Suffix *a = a_mem.get(), *b = b_mem.get();
    std::memset(text + N, 0, 32);
    parallel_for(threads, size_t(n), [&](int, size_t rb, size_t re) {
        for (size_t r = rb; r < re; ++r) {
            const uint64_t src = offsets[order[r]], len = tstart[r + 1] - tstart[r] - 1;
            std::memcpy(text + tstart[r], reads + src, size_t(len));
            text[tstart[r] + len] = 0;
            // keys, from the terminator backwards: key(p) = text[p] << 60 | key(p+1) >> 3
            uint64_t key = 0;
            for (uint64_t i = len + 1; i-- > 0;) {
                const uint64_t p = tstart[r] + i;
                key = (uint64_t(text[p]) << 60) | (key >> 3);
                a[p].key = key;
                a[p].pos = p;
            }
        }
    });
    lap("keys");
    // 3. bucket by the first 4 symbols (top 12 key bits), then sort buckets in parallel
    constexpr int kBucketBits = 12;
    constexpr size_t kBuckets = size_t(1) << kBucketBits;
    std::vector<std::vector<uint64_t>> hist(size_t(threads), std::vector<uint64_t>(kBuckets, 0));
    parallel_for(threads, size_t(N), [&](int t, size_t lo, size_t hi) {
        auto &h = hist[size_t(t)];
        for (size_t i = lo; i < hi; ++i) ++h[a[i].key >> (63 - kBucketBits)];
    });
    std::vector<uint64_t> bucket_begin(kBuckets + 1, 0);
    {
        uint64_t acc = 0;
        for (size_t k = 0; k < kBuckets; ++k) {
            bucket_begin[k] = acc;
            for (int t = 0; t < threads; ++t) {
                const uint64_t c = hist[size_t(t)][k];
                hist[size_t(t)][k] = acc;  // becomes this thread's write cursor
                acc += c;
            }
        }
        bucket_begin[kBuckets] = acc;
    }
    parallel_for(threads, size_t(N), [&](int t, size_t lo, size_t hi) {
        auto &cur = hist[size_t(t)];
        for (size_t i = lo; i < hi; ++i) b[cur[a[i].key >> (63 - kBucketBits)]++] = a[i];
    });
    a_mem.reset();
    lap("scatter");
    const uint8_t *tx = text;
    auto deep_less = [tx](const Suffix &x, const Suffix &y) {
        // equal 21-symbol keys without a terminator: keep comparing from symbol 21
        for (uint64_t i = 21;; ++i) {
            const uint8_t cx = tx[x.pos + i], cy = tx[y.pos + i];
            if (cx != cy) return cx < cy;
            if (cx == 0) return x.pos < y.pos;  // same suffix: owner's rank decides
        }
    };
    std::atomic<size_t> next_bucket{0};
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    const size_t k = next_bucket.fetch_add(1);
                    if (k >= kBuckets) break;
                    Suffix *lo = b + bucket_begin[k], *hi = b + bucket_begin[k + 1];
                    if (hi - lo < 2) continue;
                    std::sort(lo, hi, [](const Suffix &x, const Suffix &y) {
                        return x.key != y.key ? x.key < y.key : x.pos < y.pos;
                    });
                    // refine runs of equal keys whose 21 symbols hold no terminator
                    for (Suffix *g = lo; g < hi;) {
                        Suffix *e = g + 1;
                        while (e < hi && e->key == g->key) ++e;
                        if (e - g > 1 && (g->key & 7u) != 0) std::sort(g, e, deep_less);
                        g = e;
                    }
                }
            });
        for (auto &th : pool) th.join();
    }
    lap("sort");
    // 4. the BWT symbol of a suffix is the text symbol before it ('$' before a read start)
    parallel_for(threads, size_t(N), [&](int, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint64_t p = b[i].pos;
            out_symbols[i] = p == 0 ? 0 : tx[p - 1];
        }
    });
    lap("emit");
    return 0;
}

// symbols (one code per position) -> RLE bytes; returns the byte count (out may be NULL)
uint64_t synth_rle_encode(const uint8_t *symbols, uint64_t n, uint8_t *out, uint64_t cap) {
    uint64_t w = 0, i = 0;
    while (i < n) {
        const uint8_t s = symbols[i];
        uint64_t j = i + 1;
        while (j < n && symbols[j] == s) ++j;
        for (uint64_t len = j - i; len > 0; len >>= 5) {
            if (out && w < cap) out[w] = uint8_t(s | ((len & 31u) << 3));
            ++w;
        }
        i = j;
    }
    return w;
}

// Structure-equivalent synthetic RLE stream (NOT a real BWT): runs over {A,C,G,T} with a few
// '$'/'N', lengths geometric with the given mean, no two neighbouring runs of one symbol.
// Generated in independent seeded chunks (so it parallelises and is reproducible for a
// given chunk count), stitched so that chunk borders never merge two runs.  The returned
// buffer is malloc'ed; free it with synth_free.  Used for index sizes that cannot be
// suffix-sorted here.
// Run lengths: geometric with the given mean, sampled through a 64 Ki-entry inverse-CDF
// table (one lookup per run instead of a log()); lengths are capped at 255.
struct GeomTable {
    std::vector<uint8_t> len;
    explicit GeomTable(double mean_run) : len(65536) {
        const double p = 1.0 / (mean_run < 1.0 ? 1.0 : mean_run);
        double cdf = 0.0, pk = p;  // P(len = k) = p (1-p)^(k-1)
        uint32_t k = 1, filled = 0;
        while (filled < 65536) {
            cdf += pk;
            const uint32_t upto = (k >= 255) ? 65536u : uint32_t(std::min(65536.0, cdf * 65536.0 + 0.5));
            for (; filled < upto; ++filled) len[filled] = uint8_t(k);
            pk *= (1.0 - p);
            ++k;
        }
    }
};

static void gen_chunk(uint64_t target, const GeomTable &geom, double mean_run, uint64_t seed, std::vector<uint8_t> *enc) {
    uint64_t st = seed, total = 0;
    uint8_t prev = 255;
    enc->resize(size_t(double(target) / (mean_run < 1.0 ? 1.0 : mean_run) * 1.3) + 4096);
    uint8_t *out = enc->data();
    size_t w = 0, cap = enc->size();
    bool first = true;
    while (total < target) {
        if (w + 4 > cap) {  // rare: the estimate was short
            enc->resize(cap + cap / 4 + 4096);
            out = enc->data();
            cap = enc->size();
        }
        const uint64_t x = splitmix64(st);
        uint64_t len = geom.len[(x >> 32) & 0xFFFFu];
        if (len > target - total) len = target - total;
        const bool last = (total + len == target);
        uint8_t sym;
        const uint32_t pick = uint32_t(x & 1023u);
        if (pick < 8) sym = 0; else if (pick < 12) sym = 4; else sym = kBase[(x >> 10) & 3u];
        // chunk borders can never merge two runs: a chunk starts with G or T and ends with A or C
        if (first) sym = (x >> 20) & 1u ? 3 : 5;
        if (last && !first) sym = (x >> 21) & 1u ? 1 : 2;
        if (sym == prev) {
            if (last) sym = uint8_t(prev == 1 ? 2 : 1);
            else {
                sym = kBase[(x >> 12) & 3u];
                if (sym == prev) sym = uint8_t(prev == 1 ? 2 : 1);
            }
        }
        for (uint64_t l = len; l > 0; l >>= 5) out[w++] = uint8_t(sym | ((l & 31u) << 3));
        total += len;
        prev = sym;
        first = false;
    }
    enc->resize(w);
}

void synth_set_threads(int threads) { g_default_threads = threads; }

uint8_t *synth_rle_stream(uint64_t target_symbols, double mean_run, uint64_t seed, int chunks,
                          uint64_t *out_bytes, uint64_t *out_total) {
    if (chunks < 1) chunks = 1;
    if (uint64_t(chunks) * 4 > target_symbols) chunks = 1;
    std::vector<std::vector<uint8_t>> enc(static_cast<size_t>(chunks));
    const GeomTable geom(mean_run);
    {   // dynamic schedule: chunk sizes are equal but first-touch page faults are not
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < pick_threads(0); ++t)
            pool.emplace_back([&] {
                for (;;) {
                    const size_t c = next.fetch_add(1);
                    if (c >= size_t(chunks)) break;
                    // 128-bit product: target_symbols * c overflows 64 bits at human scale
                    const uint64_t lo = uint64_t((unsigned __int128)target_symbols * c / uint64_t(chunks));
                    const uint64_t hi = uint64_t((unsigned __int128)target_symbols * (c + 1) / uint64_t(chunks));
                    gen_chunk(hi - lo, geom, mean_run, seed + 0x632BE59BD9B4E019ull * (c + 1), &enc[c]);
                }
            });
        for (auto &th : pool) th.join();
    }
    uint64_t bytes = 0;
    std::vector<uint64_t> off(static_cast<size_t>(chunks) + 1, 0);
    for (size_t c = 0; c < size_t(chunks); ++c) { off[c] = bytes; bytes += enc[c].size(); }
    uint8_t *out = static_cast<uint8_t *>(std::malloc(bytes ? bytes : 1));
    if (!out) return nullptr;
    parallel_for(pick_threads(0), size_t(chunks), [&](int, size_t b, size_t e) {
        for (size_t c = b; c < e; ++c)
            if (!enc[c].empty()) std::memcpy(out + off[c], enc[c].data(), enc[c].size());
    });
    if (out_bytes) *out_bytes = bytes;
    if (out_total) *out_total = target_symbols;
    return out;
}

// Run-length histogram of an RLE stream, per symbol: hist[s * cap + min(len, cap - 1)] += 1 for every run
// (consecutive same-symbol bytes are the base-32 digits of ONE run, least significant first).
void synth_run_histogram(const uint8_t *rle, uint64_t n, uint64_t *hist, uint64_t cap) {
    uint64_t i = 0;
    while (i < n) {
        const uint8_t s = rle[i] & 7u;
        uint64_t len = 0, weight = 1;
        while (i < n && (rle[i] & 7u) == s) {
            len += uint64_t(rle[i] >> 3) * weight;
            weight <<= 5;
            ++i;
        }
        if (s < 6) ++hist[uint64_t(s) * cap + std::min(len, cap - 1)];
    }
}

// The same kind of stream as synth_rle_stream, with (symbol, run length) drawn from a MEASURED histogram
// ("run-length histogram taken from C4's BWT"): len_table[s * 65536 + u] is the inverse CDF
// of symbol s's run lengths at u / 65536, sym_cdf[s] the cumulative share of runs with symbol <= s scaled to
// 2^32.  Symbols stay independent of their neighbours (except that two neighbouring runs differ), so this is
// still not a BWT: ranges of present k-mers collapse to width 1 as before.
static void gen_chunk_hist(uint64_t target, const uint16_t *len_table, const uint32_t *sym_cdf, double mean_run, uint64_t seed,
                           std::vector<uint8_t> *enc) {
    uint64_t st = seed, total = 0;
    uint8_t prev = 255;
    enc->resize(size_t(double(target) / mean_run * 1.3) + 4096);
    size_t w = 0;
    bool first = true;
    auto pick = [&](uint32_t u) -> uint8_t {
        for (uint8_t s = 0; s < 5; ++s)
            if (u < sym_cdf[s]) return s;
        return 5;
    };
    while (total < target) {
        if (w + 8 > enc->size()) enc->resize(enc->size() + enc->size() / 4 + 4096);
        uint8_t *out = enc->data();
        uint64_t x = splitmix64(st);
        uint8_t sym = pick(uint32_t(x));
        while (sym == prev) sym = pick(uint32_t(splitmix64(st)));
        // chunk borders can never merge two runs: a chunk starts with G or T and ends with A or C
        if (first) sym = (x >> 40) & 1u ? 3 : 5;
        uint64_t len = len_table[size_t(sym) * 65536 + ((x >> 44) & 0xFFFFu)];
        if (len == 0) len = 1;
        if (len >= target - total) {  // the chunk's last run
            len = target - total;
            if (!first) {
                sym = (x >> 41) & 1u ? 1 : 2;
                if (sym == prev) sym = uint8_t(prev == 1 ? 2 : 1);
            }
        }
        for (uint64_t l = len; l > 0; l >>= 5) out[w++] = uint8_t(sym | ((l & 31u) << 3));
        total += len;
        prev = sym;
        first = false;
    }
    enc->resize(w);
}

uint8_t *synth_rle_stream_hist(uint64_t target_symbols, const uint16_t *len_table, const uint32_t *sym_cdf, double mean_run, uint64_t seed,
                               int chunks, int threads, uint64_t *out_bytes, uint64_t *out_total) {
    if (chunks < 1) chunks = 1;
    if (uint64_t(chunks) * 4 > target_symbols) chunks = 1;
    std::vector<std::vector<uint8_t>> enc(static_cast<size_t>(chunks));
    {
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < pick_threads(threads); ++t)
            pool.emplace_back([&] {
                for (;;) {
                    const size_t c = next.fetch_add(1);
                    if (c >= size_t(chunks)) break;
                    const uint64_t lo = uint64_t((unsigned __int128)target_symbols * c / uint64_t(chunks));
                    const uint64_t hi = uint64_t((unsigned __int128)target_symbols * (c + 1) / uint64_t(chunks));
                    gen_chunk_hist(hi - lo, len_table, sym_cdf, mean_run < 1.0 ? 1.0 : mean_run, seed + 0x632BE59BD9B4E019ull * (c + 1), &enc[c]);
                }
            });
        for (auto &th : pool) th.join();
    }
    uint64_t bytes = 0;
    std::vector<uint64_t> off(static_cast<size_t>(chunks) + 1, 0);
    for (size_t c = 0; c < size_t(chunks); ++c) { off[c] = bytes; bytes += enc[c].size(); }
    uint8_t *out = static_cast<uint8_t *>(std::malloc(bytes ? bytes : 1));
    if (!out) return nullptr;
    parallel_for(pick_threads(threads), size_t(chunks), [&](int, size_t b, size_t e) {
        for (size_t c = b; c < e; ++c)
            if (!enc[c].empty()) std::memcpy(out + off[c], enc[c].data(), enc[c].size());
    });
    if (out_bytes) *out_bytes = bytes;
    if (out_total) *out_total = target_symbols;
    return out;
}

void synth_free(void *p) { std::free(p); }

}  // extern "C"
