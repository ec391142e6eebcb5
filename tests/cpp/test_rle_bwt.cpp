// The reference's own RleBWT tests (src/rle_bwt.rs:478-710) and doc-tests
// (src/msbwt_core.rs:37-122, src/lib.rs:21-29), written against the C++ mirror of the trait
// (include/msbwt_hip.hpp) -- they read like the originals with the type substituted.
// Needs an MI355X.  argv[1] = path of tests/golden/two_string.npy, argv[2] = a scratch dir.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "msbwt_hip.hpp"

using namespace msbwt;

static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
            ++failures;                                                      \
        }                                                                    \
    } while (0)

// naive_bwt(["CCGT","N","ACG"]) and naive_bwt(["CCGTACGTA","GGTACAGTA","ACGACGACG"])
// (src/bwt_util.rs:154-171; the literals are pinned in tests/golden/reference_vectors.json)
static const char *kBwtSmall = "GTN$$ACCC$G";
static const char *kBwtThree = "GAATTTGG$TCA$AAAACCCC$CAGCGGGG";

static void test_load_rlebwt_from_npy(const std::string &dir) {  // rle_bwt.rs:478-503
    const std::vector<uint8_t> compressed = convert_to_vec(kBwtSmall);
    const std::string filename = dir + "/temp_data_cpp.npy";
    save_bwt_numpy(compressed, filename);
    RleBWT bwt;
    bwt.load_numpy_file(filename);
    const uint64_t expected_totals[6] = {3, 1, 3, 2, 1, 1};
    for (int i = 0; i < 6; ++i) CHECK(bwt.get_symbol_count(uint8_t(i)) == expected_totals[i]);
}

static void test_constrain_range() {  // rle_bwt.rs:603-675
    const std::string bwt_stream = kBwtSmall;
    const std::vector<uint8_t> bwt_int_form = convert_stoi(bwt_stream);
    const std::vector<uint8_t> compressed = convert_to_vec(bwt_stream);
    CHECK(compressed.size() == 8);
    for (uint8_t bin_power = 1; bin_power < 5; ++bin_power) {
        RleBWT bwt = RleBWT::with_bin_power(bin_power);
        bwt.load_vector(compressed);
        uint64_t start_index[6], end_index[6], acc = 0;
        for (int s = 0; s < 6; ++s) {
            start_index[s] = acc;
            acc += bwt.get_symbol_count(uint8_t(s));
            end_index[s] = acc;
        }
        const BWTRange initial_range{0, bwt_stream.size()};
        for (uint8_t sym = 0; sym < VC_LEN; ++sym)
            CHECK(bwt.constrain_range(sym, initial_range) == (BWTRange{start_index[sym], end_index[sym]}));
        for (uint8_t sym = 0; sym < VC_LEN; ++sym) {
            uint64_t sym_count = 0;
            for (size_t ind = 0; ind < bwt_stream.size() + 1; ++ind) {
                CHECK(bwt.constrain_range(sym, BWTRange{0, ind}) == (BWTRange{start_index[sym], start_index[sym] + sym_count}));
                CHECK(bwt.constrain_range(sym, BWTRange{ind, bwt_stream.size()}) ==
                      (BWTRange{start_index[sym] + sym_count, end_index[sym]}));
                if (ind < bwt_stream.size() && bwt_int_form[ind] == sym) ++sym_count;
            }
        }
    }
}

static void test_count_kmer() {  // rle_bwt.rs:677-710
    const std::vector<std::string> data = {"CCGTACGTA", "GGTACAGTA", "ACGACGACG"};
    const std::vector<uint8_t> compressed = convert_to_vec(kBwtThree);
    for (uint8_t bin_power = 1; bin_power < 5; ++bin_power) {
        RleBWT bwt = RleBWT::with_bin_power(bin_power);
        bwt.load_vector(compressed);
        for (uint8_t c = 0; c < VC_LEN; ++c) CHECK(bwt.get_symbol_count(c) == bwt.count_kmer({c}));
        for (const std::string &seq : data) CHECK(bwt.count_kmer(convert_stoi(seq)) == 1);
        CHECK(bwt.count_kmer(convert_stoi("ACG")) == 4);
        CHECK(bwt.count_kmer(convert_stoi("CC")) == 1);
        CHECK(bwt.count_kmer(convert_stoi("TAC")) == 2);
    }
}

static void doc_tests(const std::string &two_string_npy) {
    {   // msbwt_core.rs:110-122, :68-73, :83-88 -- strings "ACGT" and "CCGG"
        RleBWT bwt;
        bwt.load_vector(convert_to_vec("TG$$CAGCCG"));
        CHECK(bwt.count_kmer({1, 2, 3, 5}) == 1);
        CHECK(bwt.count_kmer({2, 3}) == 2);
        CHECK(bwt.count_kmer(convert_stoi("CG")) == 2);
        CHECK(bwt.get_symbol_count(0) == 2);
        CHECK(bwt.get_total_size() == 10);
    }
    {   // lib.rs:21-29, README.md:62-70
        RleBWT bwt;
        bwt.load_numpy_file(two_string_npy);
        CHECK(bwt.count_kmer(convert_stoi("ACGT")) == 1);
        CHECK(bwt.count_kmer(convert_stoi("TGCA")) == 1);
        CHECK(bwt.count_kmer(convert_stoi("$")) == 2);
    }
    // string_util.rs:36-88
    CHECK(convert_stoi("ACGTN$") == (std::vector<uint8_t>{1, 2, 3, 5, 4, 0}));
    CHECK(convert_itos({0, 1, 2, 3, 4, 5}) == "$ACGNT");
    CHECK(reverse_complement_i({0, 1, 2, 3, 4, 5}) == (std::vector<uint8_t>{1, 4, 2, 3, 5, 0}));
}

static void error_behaviour(const std::string &dir) {
    RleBWT bwt;
    bwt.load_vector(convert_to_vec(kBwtSmall));
    bool panicked = false;
    try { bwt.count_kmer({1, 6}); } catch (const Panic &p) { panicked = p.code == MSBWT_ERR_INVALID_SYMBOL; }  // msbwt_core.rs:127
    CHECK(panicked);
    bool io_failed = false;
    try { bwt.load_numpy_file(dir + "/does_not_exist.npy"); } catch (const std::system_error &) { io_failed = true; }
    CHECK(io_failed);
    // batch forms agree with the single calls
    RleBWT three;
    three.load_vector(convert_to_vec(kBwtThree));
    const std::vector<uint8_t> flat = {1, 2, 3, 2, 2, 0, 5, 1, 2};  // ACG CC$ TAC
    const std::vector<uint64_t> counts = three.count_kmers(flat, 3);
    CHECK(counts.size() == 3 && counts[0] == 4 && counts[2] == 2);
    // replicas (here on the same device) and a batch sharded over them
    const RleBWT copy = three.replicate(-1);
    CHECK(copy.get_total_size() == three.get_total_size() && copy.count_kmer(convert_stoi("ACG")) == 4);
    const std::vector<uint64_t> sharded = RleBWT::count_kmers_multi({&three, &copy}, flat, 3);
    CHECK(sharded == counts);
    // round 4: the same k-mers as 2-bit words (no '$' there: the middle one is replaced), 64- and 32-bit counts; a memory
    // budget and the library's own batch order change nothing
    const std::vector<uint8_t> acgt = {1, 2, 3, 2, 2, 3, 5, 1, 2};  // ACG CCG TAC
    const std::vector<uint64_t> words = RleBWT::pack_2bit(acgt, 3);
    CHECK(words.size() == 3 && words[0] == ((0ull << 4) | (1ull << 2) | 2ull));  // ACG = 0 1 2 base 4, first symbol most significant
    const std::vector<uint64_t> packed = three.count_kmers_packed(words, 3);
    const std::vector<uint32_t> packed32 = three.count_kmers_packed_u32(words, 3);
    CHECK(packed == three.count_kmers(acgt, 3) && packed[0] == 4 && packed[2] == 2);
    CHECK(packed32.size() == 3 && packed32[0] == 4 && packed32[1] == packed[1] && packed32[2] == 2);
    three.set_memory_budget(1 << 20);
    three.set_batch_order(1);
    CHECK(three.count_kmers(acgt, 3) == packed && three.device_bytes() > 0);
    three.set_memory_budget(0);
    three.set_batch_order(-1);
    // round 5: a sparse suffix table of the smallest depth (every string here is shorter: nothing occurs) changes nothing either
    three.set_sparse_table(16);
    CHECK(three.get_sparse_table() == 16 && three.count_kmers(acgt, 3) == packed && three.count_kmer(convert_stoi("ACGACGACG")) == 1);
    three.set_sparse_table(-1);
    const auto both = three.count_read_kmers("CCGTACGTAGGTACAGTA", 9, 3);
    CHECK(both.first.size() == 14 && both.first[2] == three.count_kmer(convert_stoi("GTA")));
    CHECK(both.second[0] == three.count_kmer(reverse_complement_i(convert_stoi("CCG"))));
}

int main(int argc, char **argv) {
    if (argc < 3) {
        std::printf("usage: %s two_string.npy scratch_dir\n", argv[0]);
        return 2;
    }
    test_load_rlebwt_from_npy(argv[2]);
    test_constrain_range();
    test_count_kmer();
    doc_tests(argv[1]);
    error_behaviour(argv[2]);
    std::printf(failures ? "%d check(s) failed\n" : "all C++ trait-mirror tests passed\n", failures);
    return failures ? 1 : 0;
}
