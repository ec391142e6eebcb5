// msbwt_hip.hpp -- header-only C++17 mirror of the reference's `msbwt_core::BWT` trait and of
// `rle_bwt::RleBWT`, over the C ABI of msbwt_hip.h.
//
// The reference is Rust (src/msbwt_core.rs:28-162, src/rle_bwt.rs); where no Rust toolchain is
// available this is the compiled-language host side: same names, same argument meaning, same
// error behaviour (io errors -> std::system_error / msbwt::UnexpectedEof, the reference's
// panics -> msbwt::Panic).  The Rust `impl BWT for GpuRleBWT` over the same symbols is in
// INTEGRATION.md.  Every query runs on the GPU; there is no CPU fallback.
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <system_error>
#include <vector>

#include "msbwt_hip.h"

namespace msbwt {

constexpr std::size_t VC_LEN = MSBWT_VC_LEN;            // src/msbwt_core.rs:4
constexpr std::size_t LETTER_BITS = MSBWT_LETTER_BITS;  // :6
constexpr std::size_t NUMBER_BITS = MSBWT_NUMBER_BITS;  // :8
constexpr std::size_t NUM_POWER = MSBWT_NUM_POWER;      // :10
constexpr std::uint8_t MASK = MSBWT_MASK;               // :12
constexpr std::uint8_t COUNT_MASK = MSBWT_COUNT_MASK;   // :14

/// Half-open range [l, h) of the BWT (src/msbwt_core.rs:18-24).
struct BWTRange {
    std::uint64_t l = 0;
    std::uint64_t h = 0;
    bool operator==(const BWTRange &o) const { return l == o.l && h == o.h; }
    bool operator!=(const BWTRange &o) const { return !(*this == o); }
};

/// What the reference does with `panic!` / `assert!` / an out-of-bounds index.
struct Panic : std::runtime_error {
    int code;
    Panic(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
/// io::ErrorKind::UnexpectedEof (src/rle_bwt.rs:129-148).
struct UnexpectedEof : std::runtime_error {
    explicit UnexpectedEof(const std::string &m) : std::runtime_error(m) {}
};

/// The trait surface (src/msbwt_core.rs:28-162).
class BWT {
  public:
    virtual ~BWT() = default;
    virtual void load_vector(const std::vector<std::uint8_t> &bwt) = 0;
    virtual void load_numpy_file(const std::string &filename) = 0;
    virtual std::uint64_t get_symbol_count(std::uint8_t symbol) const = 0;
    virtual std::uint64_t get_total_size() const = 0;
    virtual BWTRange constrain_range(std::uint8_t sym, const BWTRange &input_range) const = 0;
    virtual std::uint64_t count_kmer(const std::vector<std::uint8_t> &kmer) const = 0;
};

/// Drop-in for `RleBWT` (src/rle_bwt.rs:14-24) whose index lives in the HBM of an MI355X.
class RleBWT final : public BWT {
  public:
    RleBWT() : RleBWT(8) {}                                    // RleBWT::new(), :297
    static RleBWT with_bin_power(std::uint8_t bin_power, int device = -1) { return RleBWT(bin_power, device); }  // :309
    explicit RleBWT(std::uint8_t bin_power, int device = -1) : raw_(msbwt_rle_new_on_device(bin_power, device)) {
        if (!raw_) throw std::bad_alloc();
    }
    ~RleBWT() override { msbwt_rle_free(raw_); }
    RleBWT(RleBWT &&o) noexcept : raw_(o.raw_) { o.raw_ = nullptr; }
    RleBWT &operator=(RleBWT &&o) noexcept {
        if (this != &o) {
            msbwt_rle_free(raw_);
            raw_ = o.raw_;
            o.raw_ = nullptr;
        }
        return *this;
    }
    RleBWT(const RleBWT &) = delete;
    RleBWT &operator=(const RleBWT &) = delete;

    void load_vector(const std::vector<std::uint8_t> &bwt) override {
        check(msbwt_rle_load_vector(raw_, bwt.data(), bwt.size()));
    }
    void load_numpy_file(const std::string &filename) override { check(msbwt_rle_load_numpy_file(raw_, filename.c_str())); }
    std::uint64_t get_symbol_count(std::uint8_t symbol) const override {
        if (symbol >= VC_LEN) throw Panic(MSBWT_ERR_INVALID_SYMBOL, "index out of bounds: symbol >= 6");  // :174
        return msbwt_rle_get_symbol_count(raw_, symbol);
    }
    std::uint64_t get_total_size() const override { return msbwt_rle_get_total_size(raw_); }
    BWTRange constrain_range(std::uint8_t sym, const BWTRange &r) const override {
        BWTRange out;
        check(msbwt_rle_constrain_range(raw_, sym, r.l, r.h, &out.l, &out.h));
        return out;
    }
    std::uint64_t count_kmer(const std::vector<std::uint8_t> &kmer) const override {
        std::uint64_t out = 0;
        check(msbwt_rle_count_kmer(raw_, kmer.data(), kmer.size(), &out));
        return out;
    }

    // ---- batch forms (the GPU entry points proper) ----
    /// kmers: n x k symbol codes, row-major.
    std::vector<std::uint64_t> count_kmers(const std::vector<std::uint8_t> &kmers, std::size_t k) const {
        const std::size_t n = k ? kmers.size() / k : 0;
        if (k && kmers.size() % k) throw std::invalid_argument("kmers.size() is not a multiple of k");
        std::vector<std::uint64_t> out(n);
        check(msbwt_rle_count_kmers(raw_, kmers.data(), k, n, out.data()));
        return out;
    }
    std::vector<BWTRange> constrain_ranges(const std::vector<std::uint8_t> &syms, const std::vector<BWTRange> &ranges) const {
        if (syms.size() != ranges.size()) throw std::invalid_argument("syms and ranges differ in length");
        std::vector<std::uint64_t> l(ranges.size()), h(ranges.size()), ol(ranges.size()), oh(ranges.size());
        for (std::size_t i = 0; i < ranges.size(); ++i) {
            l[i] = ranges[i].l;
            h[i] = ranges[i].h;
        }
        check(msbwt_rle_constrain_ranges(raw_, syms.data(), l.data(), h.data(), ranges.size(), ol.data(), oh.data()));
        std::vector<BWTRange> out(ranges.size());
        for (std::size_t i = 0; i < ranges.size(); ++i) out[i] = BWTRange{ol[i], oh[i]};
        return out;
    }
    /// Every k-mer window of n equal-length ASCII reads; second = counts of the reverse complements.
    std::pair<std::vector<std::uint64_t>, std::vector<std::uint64_t>> count_read_kmers(const std::string &reads,
                                                                                     std::size_t read_len, std::size_t k) const {
        if (!read_len || reads.size() % read_len || k < 1 || k > read_len) throw std::invalid_argument("bad read_len / k");
        const std::size_t n = reads.size() / read_len, w = read_len - k + 1;
        std::vector<std::uint64_t> fwd(n * w), rc(n * w);
        check(msbwt_rle_count_read_kmers(raw_, reinterpret_cast<const std::uint8_t *>(reads.data()), read_len, n, k, 1,
                                         fwd.data(), rc.data()));
        return {std::move(fwd), std::move(rc)};
    }

    /// k-mers over ACGT as 2-bit words (pack_2bit): ceil(k / 32) words each, 8 bytes per 31-mer across PCIe instead of 31.
    std::vector<std::uint64_t> count_kmers_packed(const std::vector<std::uint64_t> &words, std::size_t k) const {
        const std::size_t per = k > 32 ? 2 : 1;
        if (k < 1 || k > 64 || words.size() % per) throw std::invalid_argument("bad k / words");
        std::vector<std::uint64_t> out(words.size() / per);
        check(msbwt_rle_count_kmers_packed(raw_, words.data(), k, out.size(), out.data(), 64));
        return out;
    }
    /// ... with 32-bit counts (half the bytes on the way back; Panic(MSBWT_ERR_OVERFLOW) if a count does not fit)
    std::vector<std::uint32_t> count_kmers_packed_u32(const std::vector<std::uint64_t> &words, std::size_t k) const {
        const std::size_t per = k > 32 ? 2 : 1;
        if (k < 1 || k > 64 || words.size() % per) throw std::invalid_argument("bad k / words");
        std::vector<std::uint32_t> out(words.size() / per);
        check(msbwt_rle_count_kmers_packed(raw_, words.data(), k, out.size(), out.data(), 32));
        return out;
    }
    /// n x k symbol codes (A C G T = 1 2 3 5) -> 2-bit words: the k-mer as a base-4 number, first symbol most significant
    static std::vector<std::uint64_t> pack_2bit(const std::vector<std::uint8_t> &kmers, std::size_t k) {
        if (k < 1 || k > 64 || kmers.size() % k) throw std::invalid_argument("bad k / kmers");
        std::vector<std::uint64_t> words(kmers.size() / k * (k > 32 ? 2 : 1));
        if (msbwt_kmers_pack_2bit(kmers.data(), k, kmers.size() / k, words.data()) != MSBWT_OK)
            throw std::invalid_argument("a symbol outside A C G T cannot be packed into two bits");
        return words;
    }

    /// Device batch (pointers on this handle's GPU), asynchronous on `hip_stream`; device_status() reports bad input.
    void count_kmers_device(const void *d_kmers, std::size_t k, std::size_t n, void *d_out, void *hip_stream = nullptr) const {
        check(msbwt_rle_count_kmers_device(raw_, d_kmers, k, n, d_out, hip_stream));
    }
    void device_status(void *hip_stream = nullptr) const { check(msbwt_rle_device_status(raw_, hip_stream)); }
    /// One process per GPU: every rank ends up with all ranks' counts (d_all[r * n_mine + i] = rank r's d_mine[i]).
    /// wire_bits 16 / 32 narrows the counts on the wire; device_status() throws Panic(MSBWT_ERR_OVERFLOW) if one did not fit.
    void allgather_counts(void *comm, const void *d_mine, std::size_t n_mine, void *d_all, int wire_bits = 64,
                          void *hip_stream = nullptr) const {
        check(msbwt_rle_allgather_counts(raw_, comm, d_mine, n_mine, d_all, wire_bits, hip_stream));
    }

    void set_table_depth(int depth) { check(msbwt_rle_set_table_depth(raw_, depth)); }
    void set_table_packed(int mode) { check(msbwt_rle_set_table_packed(raw_, mode)); }
    void set_pair_index(int mode) { check(msbwt_rle_set_pair_index(raw_, mode)); }
    void set_search_kernel(int mode) { check(msbwt_rle_set_search_kernel(raw_, mode)); }
    /// HBM the loaded index may hold (0 = no budget): the space / time knob, as bin_power is the reference's (rle_bwt.rs:309-322)
    void set_memory_budget(std::uint64_t bytes) { check(msbwt_rle_set_memory_budget(raw_, bytes)); }
    /// 1 = the library orders every batch it can before counting it; 0 and -1 (the default) = never: a switch, there is no automatic mode
    void set_batch_order(int mode) { check(msbwt_rle_set_batch_order(raw_, mode)); }
    /// sparse suffix table (the suffixes that occur, one hashed 128-byte bucket per lookup): -1 = automatic (default), 0 = off, 16..31 = that depth
    void set_sparse_table(int depth) { check(msbwt_rle_set_sparse_table(raw_, depth)); }
    /// the k this index will mostly be asked about (0 = unknown): the automatic sparse table goes as deep as min(k, 31) where its table fits, instead of 23
    void set_query_length(int k) { check(msbwt_rle_set_query_length(raw_, k)); }
    int get_query_length() const { return msbwt_rle_get_query_length(raw_); }
    /// two-tier form of the sparse table (entries for the suffixes that occur at least twice, filter bits for the rest: read sets with errors):
    /// -1 = where the complete table of a depth does not fit (default), 0 = never, 1 = always
    void set_sparse_tiers(int mode) { check(msbwt_rle_set_sparse_tiers(raw_, mode)); }
    bool get_sparse_tiers() const { return msbwt_rle_get_sparse_tiers(raw_) != 0; }
    /// the second, shallower sparse level (17-symbol suffixes, k undeclared): -1 = automatic (default), 0 = never
    void set_sparse_second(int mode) { check(msbwt_rle_set_sparse_second(raw_, mode)); }
    int get_sparse_table() const { return msbwt_rle_get_sparse_table(raw_); }
    void set_table_side(int mode) { check(msbwt_rle_set_table_side(raw_, mode)); }
    std::uint64_t device_bytes() const { return msbwt_rle_device_bytes(raw_); }
    int search_kernel_for(size_t k) const { return msbwt_rle_search_kernel_for(raw_, k); }
    msbwt_rle *raw() const { return raw_; }

    /// A copy of the loaded index on another GPU of the node (GPU -> GPU, no rebuild).
    RleBWT replicate(int device) const {
        msbwt_rle *other = msbwt_rle_replicate(raw_, device);
        if (!other) throw Panic(MSBWT_ERR_HIP, msbwt_rle_last_error(raw_));
        return RleBWT(other);
    }
    /// One batch sharded over several replicas (one per GPU); counts in query order.
    static std::vector<std::uint64_t> count_kmers_multi(const std::vector<const RleBWT *> &replicas,
                                                        const std::vector<std::uint8_t> &kmers, std::size_t k) {
        if (replicas.empty()) throw std::invalid_argument("no replicas");
        const std::size_t n = k ? kmers.size() / k : 0;
        std::vector<const msbwt_rle *> raws;
        for (const RleBWT *r : replicas) raws.push_back(r->raw_);
        std::vector<std::uint64_t> out(n);
        const int code = msbwt_rle_count_kmers_multi(raws.data(), raws.size(), kmers.data(), k, n, out.data());
        if (code != MSBWT_OK) throw Panic(code, "count_kmers_multi failed");
        return out;
    }

  private:
    explicit RleBWT(msbwt_rle *adopted) : raw_(adopted) {}
    void check(int code) const {
        if (code == MSBWT_OK) return;
        const std::string msg = msbwt_rle_last_error(raw_);
        switch (code) {
            case MSBWT_ERR_IO: throw std::system_error(std::make_error_code(std::errc::io_error), msg);
            case MSBWT_ERR_UNEXPECTED_EOF: throw UnexpectedEof(msg);
            default: throw Panic(code, msg);
        }
    }
    msbwt_rle *raw_;
};

/// An RCCL communicator over the GPUs of a one-process-per-GPU job, made through the library (librccl.so is bound at run
/// time).  Rank 0 calls RankComm::unique_id(), ships the bytes to the other ranks, every rank constructs a RankComm with its
/// GPU current; RleBWT::allgather_counts takes raw().
class RankComm {
  public:
    static std::array<unsigned char, MSBWT_COMM_ID_BYTES> unique_id() {
        std::array<unsigned char, MSBWT_COMM_ID_BYTES> id{};
        if (msbwt_comm_get_unique_id(id.data()) != MSBWT_OK) throw Panic(MSBWT_ERR_RCCL, "msbwt_comm_get_unique_id (is librccl.so there?)");
        return id;
    }
    RankComm(int nranks, const std::array<unsigned char, MSBWT_COMM_ID_BYTES> &id, int rank) {
        if (msbwt_comm_init_rank(&raw_, nranks, id.data(), rank) != MSBWT_OK) throw Panic(MSBWT_ERR_RCCL, "msbwt_comm_init_rank");
    }
    ~RankComm() {
        if (raw_) msbwt_comm_destroy(raw_);
    }
    RankComm(const RankComm &) = delete;
    RankComm &operator=(const RankComm &) = delete;
    void *raw() const { return raw_; }

  private:
    void *raw_ = nullptr;
};

// ---- string_util (src/string_util.rs) and bwt_converter (src/bwt_converter.rs) ----
inline std::vector<std::uint8_t> convert_stoi(const std::string &seq) {
    std::vector<std::uint8_t> out(seq.size());
    msbwt_convert_stoi(reinterpret_cast<const std::uint8_t *>(seq.data()), seq.size(), out.data());
    return out;
}
inline std::string convert_itos(const std::vector<std::uint8_t> &iseq) {
    std::string out(iseq.size(), '\0');
    msbwt_convert_itos(iseq.data(), iseq.size(), reinterpret_cast<std::uint8_t *>(&out[0]));
    return out;
}
inline std::vector<std::uint8_t> reverse_complement_i(const std::vector<std::uint8_t> &seq) {
    std::vector<std::uint8_t> out(seq.size());
    msbwt_reverse_complement_i(seq.data(), seq.size(), out.data());
    return out;
}
inline std::vector<std::uint8_t> convert_to_vec(const std::string &bwt) {
    const auto *p = reinterpret_cast<const std::uint8_t *>(bwt.data());
    const std::size_t need = msbwt_convert_to_vec(p, bwt.size(), nullptr, 0);
    if (need == SIZE_MAX) throw Panic(MSBWT_ERR_INVALID_SYMBOL, "Unexpected symbol in input");
    std::vector<std::uint8_t> out(need);
    msbwt_convert_to_vec(p, bwt.size(), out.data(), need);
    return out;
}
inline void save_bwt_numpy(const std::vector<std::uint8_t> &bwt, const std::string &filename) {
    if (msbwt_save_bwt_numpy(bwt.data(), bwt.size(), filename.c_str()) != MSBWT_OK)
        throw std::system_error(std::make_error_code(std::errc::io_error), "cannot write " + filename);
}

}  // namespace msbwt
