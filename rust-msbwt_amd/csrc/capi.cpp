// extern "C" boundary (include/msbwt_hip.h): handle management, load path, batch plumbing.
// The query work itself is in kernels.hip; nothing here computes a rank on the CPU.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/msbwt_hip.h"
#include "device_build.hpp"
#include "gather.hpp"
#include "host_pipeline.hpp"
#include "kernels.hpp"
#include "table_policy.hpp"
#include "npy_io.hpp"
#include "order.hpp"
#include "pair_index.hpp"
#include "plane_index.hpp"
#include "rle_codec.hpp"
#include "run_build.hpp"
#include "run_index.hpp"
#include "sparse_build.hpp"
#include "sparse_policy.hpp"

using namespace msbwt;

struct msbwt_rle {
    int device = 0;
    uint8_t bin_power = 8;
    bool loaded = false;
    Totals totals{};
    void *d_blocks = nullptr;
    uint64_t nblocks = 0;
    int block_format = kBlocksPlanes;         // format of d_blocks
    int wanted_block_format = kBlocksPlanes;  // takes effect at the next load
    void *d_overflow = nullptr;               // run blocks: plane-shaped lines of the overflowing blocks
    uint64_t overflow_bytes = 0;
    void *d_pair_blocks = nullptr;  // optional pair index (two symbols per step)
    void *d_pair_super = nullptr;
    uint64_t pair_bytes = 0;
    uint64_t pair_overlap_bytes = 0;  // what overlapping pair blocks take beyond disjoint ones (0 unless the data-driven policy chose them)
    int wanted_pair = -1;           // -1 = on when it fits comfortably, 0 = off, 1 = on
    int pair_stride = 128;          // spacing of the pair blocks in HBM: 128, or 96 (overlapping)
    int wanted_pair_stride = 0;     // 0 = automatic (table_policy.hpp: cheap -> 96; else 96 when the data keep ranges wide and it fits)
    double typical_width = -1.0;    // median occurrence count of a present 24-mer, probed at load time (-1: not probed)
    void *d_table = nullptr;
    int table_depth = 0;         // symbols a table entry stands for (of the table currently in HBM)
    bool table_packed = false;   // packed lines (two levels deeper than the flat table it was made from)
    size_t table_bytes = 0;
    void *d_table_side = nullptr;    // packed table: flat entries of its escape lines (512 bytes per line), or nullptr
    uint64_t table_side_bytes = 0;
    uint64_t table_lines = 0, table_escape_lines = 0;  // of the packed table in HBM
    int wanted_table_side = 1;       // 0 = no side array (queries of escape lines search from scratch, as until round 3)
    // sparse suffix table (sparse_table.hpp): ranges of the suffixes that occur, deeper than the direct table reaches
    void *d_sparse = nullptr;        // (nbuckets + probe) lines of 128 bytes
    void *d_sparse_side = nullptr;   // 16-byte {l, h} entries of its ESCAPE entries
    uint64_t sparse_bytes = 0, sparse_side_bytes = 0;
    uint32_t sparse_nbuckets = 0, sparse_probe = 0;
    int sparse_depth = 0;
    bool sparse_tier = false;        // the table in HBM is of the two-tier form (entries for the suffixes at least 2 wide, filter bits for the rest)
    // second, shallower sparse table (round 6; k undeclared): serves the queries shorter than the first one's entries (17 <= k < 23), which would
    // otherwise fall to the direct table -- shallow beside a sparse table -- and lose 1.5-2.5 x against the index without one
    void *d_sparse2 = nullptr, *d_sparse2_side = nullptr;
    uint64_t sparse2_bytes = 0, sparse2_side_bytes = 0, sparse2_entries = 0;
    uint32_t sparse2_nbuckets = 0, sparse2_probe = 0;
    int sparse2_depth = 0;
    bool sparse2_tier = false;
    int wanted_second = -1;          // -1 = automatic (k undeclared, the deep direct table does not fit, this one does), 0 = never
    int wanted_tiers = -1;           // -1 = two-tier where the complete table of a depth does not fit, 0 = complete tables only, 1 = two-tier only
    int wanted_streaming = -1;       // index lines fetched non-temporally: -1 = when the random-access arrays dwarf the caches, 0 = never, 1 = always
    int wanted_sparse = -1;          // -1 = automatic (beside a pair index, as deep as the data and HBM allow, at most 23 -- or what query_length says), 0 = off, 16..28 = that depth
    int query_length = 0;            // the k the index will mostly be asked about (msbwt_rle_set_query_length), 0 = unknown
    SparseBuildReport sparse_report{};
    bool counting = false;           // search counters wanted (msbwt_rle_set_search_counters)
    int wanted_table_packed = -1; // -1 = pack when the data warrants it and it fits, 0 = never, 1 = whenever a pair index exists
    uint32_t *d_filter = nullptr;   // presence bits over the low 2*filter_depth index bits of the table
    int filter_depth = 0;
    int wanted_filter = -1;         // -1 = keep it when it can reject something, 0 = off
    int wanted_table_depth = -1; // -1 = pick from the index size
    int search_kernel = kSearchAuto;
    // Tile-ticket counter blocks of the lanes kernel (kernels.hpp, kTicketBytes each): a launch takes a block whose
    // previous launch has COMPLETED (its event says so) or a new one, so two launches in flight on different
    // streams never share counters however many there are.
    struct TicketSlot {
        void *counters = nullptr;
        hipEvent_t done = nullptr;
        bool used = false;  // `done` has been recorded at least once
        hipStream_t last_stream = nullptr;  // the stream of the launch that used it last
        void *order_scratch = nullptr;      // scratch of the batch-ordering pass (order.hip) of the launch that holds the slot
        size_t order_bytes = 0;
    };
    uint64_t memory_budget = 0;  // bytes of HBM the index may hold (0 = no budget): msbwt_rle_set_memory_budget
    bool planned = false;        // a budget is in force: `plan` (table_policy.hpp, plan_index) decides the optional structures
    IndexPlan plan{};
    int wanted_order = -1;   // batch order: 1 = whenever the passes apply; 0 and -1 (automatic: see order_pays) = never
    int order_bits = 22;     // key bits the bucket passes order by (11 in the global pass + 11 inside each bucket)
    std::vector<TicketSlot> tickets;
    // device status block (128 bytes): word 0 = flags of the host-pointer entry points (handle
    // stream), word 1 = flags of the *_device entry points (caller streams; read and cleared only by
    // msbwt_rle_device_status), bytes 64.. = 8 x u64 record of a failed device consistency check
    uint32_t *d_flags = nullptr;
    hipStream_t stream = nullptr;  // used by the host-pointer entry points
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    HostPipeline pipe;             // pinned, triple-buffered path of the host-pointer batch entry points
    // Small host batches (the trait's single-query calls above all): queries and results travel through ONE
    // mapped, coherent host buffer that the kernel reads and writes directly -- no copies, no memset, no flag
    // read-back; one launch and one stream synchronisation per call.
    void *d_gather = nullptr;      // scratch of msbwt_rle_allgather_counts (narrow wire widths)
    size_t gather_bytes = 0;
    hipStream_t gather_stream = nullptr;  // msbwt_rle_count_kmers_allgather_device: the all-gathers of a batch's pieces run here, beside the search
    std::vector<hipEvent_t> piece_events;
    uint8_t *mail = nullptr;       // host address
    uint8_t *d_mail = nullptr;     // the same buffer as the device sees it
    uint64_t mail_seq = 0;         // completion word of the mailbox: the kernel of call i writes i
    bool timing = false;
    std::vector<hipEvent_t> events;  // start/stop pairs not yet read back
    double timed_ms = 0.0;
    uint64_t timed_launches = 0;
    std::mutex mu;
    std::string err;
};

namespace {

constexpr int kMaxTableDepth = 16;  // 4^16 x 16 B = 64 GiB
constexpr uint64_t kStreamLinesFrom = uint64_t(4) << 30;  // random-access arrays from here on are read with the non-temporal hint (view_of)

constexpr size_t kStatusBytes = 1024;  // flag words, debug record (bytes 64..128), search counters (bytes 128..256)
constexpr size_t kCountersOffset = 128;
static_assert(MSBWT_SEARCH_COUNTERS == kSearchCounters, "the header's counter block is the kernels'");
static_assert(10 + kSparseMaxDepth + 1 <= 42 && 45 + kSparseMaxDepth + 1 <= 80 && 80 + kSparseMaxDepth + 1 <= MSBWT_SPARSE_INFO_WORDS,
              "msbwt_rle_sparse_table_info: [10 + d] distinct, [42] filtered, [45 + d] wide, [80 + d] once");
constexpr size_t kPackScratchOffset = 256;  // two u64 of the table packer (escape-line count, side-array cursor)
constexpr size_t kMaxTimedEvents = 256;  // start/stop pairs kept before timed_launch folds them into the running sum
constexpr int kHostFlags = 0, kDeviceFlags = 1;  // words of the status block

const char *kVersion = "rust-msbwt_amd 0.1.0 (gfx950 plane-block index)";

// Makes the handle's device current for the scope, restoring the caller's afterwards (the
// caller may be a torch process with its own current device).
class DeviceScope {
  public:
    explicit DeviceScope(int device) {
        err_ = hipGetDevice(&prev_);
        if (err_ == hipSuccess && prev_ != device) {
            err_ = hipSetDevice(device);
            switched_ = err_ == hipSuccess;
        }
        ok_ = err_ == hipSuccess;
    }
    ~DeviceScope() {
        if (switched_) (void)hipSetDevice(prev_);
    }
    bool ok() const { return ok_; }
    std::string why() const { return std::string("no usable HIP device: ") + hipGetErrorString(err_); }

  private:
    hipError_t err_ = hipSuccess;
    int prev_ = 0;
    bool ok_ = false, switched_ = false;
};

int fail(msbwt_rle *h, int code, const std::string &msg) {
    if (h) h->err = msg;
    return code;
}

int hip_fail(msbwt_rle *h, hipError_t e, const char *what) {
    return fail(h, MSBWT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(h, expr)                                      \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return hip_fail(h, e_, #expr);  \
    } while (0)

// random 128-byte lines per second the memory system serves from this allocation right now (0: could not be measured)
double line_rate_of(const void *p, size_t bytes, hipStream_t stream) {
    hipEvent_t a = nullptr, b = nullptr;
    uint64_t lines = 0;
    float ms = 0.f;
    hipError_t e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    if (e == hipSuccess) e = launch_probe_lines(p, bytes, 4, nullptr, nullptr, stream);  // warm-up: page tables, clocks
    if (e == hipSuccess) e = hipEventRecord(a, stream);
    if (e == hipSuccess) e = launch_probe_lines(p, bytes, 48, &lines, nullptr, stream);  // 2.5 x 10^7 lines: about 0.6 ms
    if (e == hipSuccess) e = hipEventRecord(b, stream);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (e != hipSuccess || ms <= 0.f) {
        (void)hipGetLastError();
        return 0.0;
    }
    return double(lines) / (double(ms) * 1e-3);
}

void release_sparse(msbwt_rle *h) {
    if (h->d_sparse) (void)hipFree(h->d_sparse);
    if (h->d_sparse_side) (void)hipFree(h->d_sparse_side);
    h->d_sparse = h->d_sparse_side = nullptr;
    h->sparse_bytes = h->sparse_side_bytes = 0;
    h->sparse_nbuckets = h->sparse_probe = 0;
    h->sparse_depth = 0;
    h->sparse_tier = false;
    h->sparse_report = SparseBuildReport{};
    if (h->d_sparse2) (void)hipFree(h->d_sparse2);
    if (h->d_sparse2_side) (void)hipFree(h->d_sparse2_side);
    h->d_sparse2 = h->d_sparse2_side = nullptr;
    h->sparse2_bytes = h->sparse2_side_bytes = h->sparse2_entries = 0;
    h->sparse2_nbuckets = h->sparse2_probe = 0;
    h->sparse2_depth = 0;
    h->sparse2_tier = false;
}

void release_index(msbwt_rle *h) {
    if (h->d_blocks) (void)hipFree(h->d_blocks);
    if (h->d_overflow) (void)hipFree(h->d_overflow);
    h->d_overflow = nullptr;
    h->overflow_bytes = 0;
    if (h->d_table) (void)hipFree(h->d_table);
    if (h->d_table_side) (void)hipFree(h->d_table_side);
    h->d_table_side = nullptr;
    h->table_side_bytes = h->table_lines = h->table_escape_lines = 0;
    if (h->d_filter) (void)hipFree(h->d_filter);
    h->d_filter = nullptr;
    h->filter_depth = 0;
    release_sparse(h);
    if (h->d_pair_blocks) (void)hipFree(h->d_pair_blocks);
    if (h->d_pair_super) (void)hipFree(h->d_pair_super);
    h->d_blocks = h->d_table = h->d_pair_blocks = h->d_pair_super = nullptr;
    h->pair_bytes = 0;
    h->nblocks = 0;
    h->table_depth = 0;
    h->table_packed = false;
    h->table_bytes = 0;
    h->typical_width = -1.0;
    h->totals = Totals{};  // an unloaded handle reports 0 symbols, not the previous BWT's
    h->loaded = false;
}

IndexView view_of(msbwt_rle *h) {
    IndexView v;
    v.blocks = h->d_blocks;
    v.block_format = h->block_format;
    v.overflow = h->d_overflow;
    v.nblocks = h->nblocks;
    v.total = h->totals.total;
    v.table.entries = h->d_table;
    v.table.depth = h->d_table ? h->table_depth : 0;
    v.table.packed = h->d_table && h->table_packed;
    v.table.filter = h->d_table ? h->d_filter : nullptr;
    v.table.filter_depth = h->filter_depth;
    v.table.side = (h->d_table && h->table_packed) ? h->d_table_side : nullptr;
    v.counters = (h->counting && h->d_flags) ? reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(h->d_flags) + kCountersOffset) : nullptr;
    v.pair_blocks = h->d_pair_blocks;
    v.pair_super = static_cast<const uint64_t *>(h->d_pair_super);
    v.pair_stride96 = h->d_pair_blocks && h->pair_stride == 96;
    v.search_kernel = h->search_kernel;
    {   // lines used once should not evict what is reused -- once the arrays the search reads at random (pair blocks, else the blocks
        // themselves) are far beyond what L2 (8 x 4 MB) and the Infinity Cache (256 MB) hold: 4 GiB and up
        const uint64_t hot = h->d_pair_blocks ? h->pair_bytes : h->nblocks * kBlockBytes;
        v.stream_lines = h->wanted_streaming > 0 || (h->wanted_streaming < 0 && hot >= kStreamLinesFrom);
    }
    if (h->d_sparse && (h->d_pair_blocks || h->block_format == kBlocksRuns)) {  // (run blocks: built from pair blocks that are gone again)
        v.sparse.lines = h->d_sparse;
        v.sparse.nbuckets = h->sparse_nbuckets;
        v.sparse.depth = uint32_t(h->sparse_depth);
        v.sparse.probe = h->sparse_probe;
        v.sparse.side = h->d_sparse_side;
        v.sparse.tier = h->sparse_tier ? 1u : 0u;
        if (h->d_sparse2) {
            v.sparse2.lines = h->d_sparse2;
            v.sparse2.nbuckets = h->sparse2_nbuckets;
            v.sparse2.depth = uint32_t(h->sparse2_depth);
            v.sparse2.probe = h->sparse2_probe;
            v.sparse2.side = h->d_sparse2_side;
            v.sparse2.tier = h->sparse2_tier ? 1u : 0u;
        }
    }
    v.debug = h->d_flags ? reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(h->d_flags) + 64) : nullptr;
    return v;  // tile_counter: with_tickets()
}

// Runs `launch(view)` with a ticket-counter block that no launch still in flight uses, and marks the block busy
// until everything enqueued on `stream` so far -- the launch included -- has completed.  The caller holds h->mu.
template <class Launch>
hipError_t with_slot(msbwt_rle *h, hipStream_t stream, Launch &&launch) {
    // Launches queued back to back on ONE stream are ordered by the stream itself (the memset of the counters waits for the
    // previous kernel), so they share a block without asking its event: a caller that enqueues N asynchronous launches
    // gets one block, not N allocations inside its launch path.
    // (NOT for hipStreamPerThread: that one handle value stands for a different queue in every host thread, so two threads' launches
    // "on the same stream" may run side by side -- they go by the completion event like launches on different streams)
    msbwt_rle::TicketSlot *slot = nullptr;
    for (auto &s : h->tickets)
        if (stream != hipStreamPerThread && s.used && s.last_stream == stream) {
            slot = &s;
            break;
        }
    for (auto &s : h->tickets)
        if (!slot && (!s.used || hipEventQuery(s.done) == hipSuccess)) slot = &s;
    (void)hipGetLastError();  // hipErrorNotReady from a busy slot is not an error
    if (!slot) {
        msbwt_rle::TicketSlot fresh;
        hipError_t e = hipMalloc(&fresh.counters, kTicketBytes);
        if (std::getenv("MSBWT_VERBOSE")) std::fprintf(stderr, "[msbwt] launch slot: ticket counters %p\n", fresh.counters);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&fresh.done, hipEventDisableTiming);
        if (e != hipSuccess) {
            if (fresh.counters) (void)hipFree(fresh.counters);
            return e;
        }
        h->tickets.push_back(fresh);
        slot = &h->tickets.back();
    }
    IndexView v = view_of(h);
    v.tile_counter = slot->counters;
    hipError_t e = launch(v, *slot);
    // recorded even after a failed launch: the memset of the counters may already be queued
    const hipError_t r = hipEventRecord(slot->done, stream);
    slot->used = true;
    slot->last_stream = stream;
    return e != hipSuccess ? e : r;
}

template <class Launch>
hipError_t with_tickets(msbwt_rle *h, hipStream_t stream, Launch &&launch) {
    return with_slot(h, stream, [&](const IndexView &v, msbwt_rle::TicketSlot &) { return launch(v); });
}

int ensure_runtime(msbwt_rle *h) {
    if (!h->stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    if (!h->d_flags) {
        HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&h->d_flags), kStatusBytes));
        if (std::getenv("MSBWT_VERBOSE")) std::fprintf(stderr, "[msbwt] status block %p\n", static_cast<void *>(h->d_flags));
        HIP_TRY(h, hipMemset(h->d_flags, 0, kStatusBytes));
    }
    return MSBWT_OK;
}

// mailbox layout (bytes); kMailQueries queries of at most kMailKmerBytes in all
constexpr size_t kMailQueries = 64, kMailKmerBytes = 4096;
constexpr size_t kMailDone = 0;  // u64 completion word (lanes kernel, one wave)
constexpr size_t kMailKmers = 64, kMailCounts = kMailKmers + kMailKmerBytes, kMailSyms = kMailCounts + 8 * kMailQueries,
                 kMailL = kMailSyms + 64, kMailH = kMailL + 8 * kMailQueries, kMailOutL = kMailH + 8 * kMailQueries,
                 kMailOutH = kMailOutL + 8 * kMailQueries, kMailBytes = kMailOutH + 8 * kMailQueries;

int ensure_mail(msbwt_rle *h) {
    if (h->mail) return MSBWT_OK;
    void *host = nullptr, *dev = nullptr;
    // coherent explicitly: the host polls a word the kernel writes (HIP_HOST_COHERENT=0 in the environment must not turn every
    // single-query call into a 2 ms spin)
    HIP_TRY(h, hipHostMalloc(&host, kMailBytes, hipHostMallocMapped | hipHostMallocCoherent));
    const hipError_t e = hipHostGetDevicePointer(&dev, host, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(host);
        return hip_fail(h, e, "hipHostGetDevicePointer");
    }
    std::memset(host, 0, kMailBytes);
    h->mail = static_cast<uint8_t *>(host);
    h->d_mail = static_cast<uint8_t *>(dev);
    return MSBWT_OK;
}

int ensure_stage(msbwt_rle *h, size_t bytes) {
    if (bytes <= h->stage_bytes) return MSBWT_OK;
    if (h->d_stage) (void)hipFree(h->d_stage);
    h->d_stage = nullptr;
    h->stage_bytes = 0;
    HIP_TRY(h, hipMalloc(&h->d_stage, bytes));
    h->stage_bytes = bytes;
    return MSBWT_OK;
}

// Presence filter over the finished table: 4^min(12, depth) bits (<= 2 MiB, L2-sized).  Kept
// only if it can reject something (less than 90 % of its bits set) -- on a large genome every
// 12-mer occurs and the filter would be a wasted lookup.
int rebuild_filter(msbwt_rle *h) {
    if (h->d_filter) (void)hipFree(h->d_filter);
    h->d_filter = nullptr;
    h->filter_depth = 0;
    if (!h->d_table || h->wanted_filter == 0 || h->table_depth < 6) return MSBWT_OK;
    const int fd = std::min(12, h->table_depth);
    const size_t words = (size_t(1) << (2 * fd)) / 32;
    uint32_t *filter = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&filter), words * sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(filter, 0, words * sizeof(uint32_t), h->stream);
    if (e == hipSuccess) e = launch_build_filter(h->d_table, h->table_depth, fd, filter, h->stream);
    std::vector<uint32_t> host(words);
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), filter, words * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        (void)hipFree(filter);
        return hip_fail(h, e, "build presence filter");
    }
    uint64_t set = 0;
    for (uint32_t w : host) set += uint64_t(__builtin_popcount(w));
    if (double(set) > 0.9 * double(words * 32)) {
        (void)hipFree(filter);
        return MSBWT_OK;
    }
    h->d_filter = filter;
    h->filter_depth = fd;
    return MSBWT_OK;
}

// Sparse suffix table (sparse_table.hpp) from the flat direct table that is in HBM right now (its parent; none: from the root).
// Optional structure: when nothing fits (or a step fails for want of memory) the handle simply has none -- unless a depth was
// asked for explicitly, which is then an error.  keep_free: bytes that what is built afterwards (the packed direct table) still
// needs; allowance: what a memory budget leaves for this table (kNoBudget: none in force).
// deep_direct_depth: the flat depth of the DEEP direct table that is kept beside the sparse table when HBM is plentiful (rebuild_table; 0 =
// not in question) -- where that one fits no second sparse level is built.
constexpr int kSparseSecondDepth = 17;  // entries of the second, shallower level (what the packed direct table of round 4 reached)

bool deep_direct_fits(const msbwt_rle *h, int flat_depth_wanted) {
    if (flat_depth_wanted <= 0 || flat_depth_wanted + 2 > 18 || h->planned) return false;
    size_t free_b = 0, total_b = 0;
    const uint64_t flat_deep = (uint64_t(1) << (2 * flat_depth_wanted)) * 16, packed = packed_table_bytes(flat_depth_wanted + 2);
    const uint64_t need = flat_deep + packed + packed / 8;  // (the packer's side array of escape lines: an eighth at most in practice)
    return hipMemGetInfo(&free_b, &total_b) == hipSuccess && uint64_t(free_b) + h->table_bytes >= need + uint64_t(total_b) / 8;
}

int build_sparse(msbwt_rle *h, uint64_t keep_free, uint64_t allowance, int deep_direct_depth = 0) {
    release_sparse(h);
    const bool verbose = std::getenv("MSBWT_VERBOSE") != nullptr;
    const bool explicit_depth = h->wanted_sparse > 0;
    const void *flat = (h->d_table && !h->table_packed) ? h->d_table : nullptr;
    const int flat_depth = flat ? h->table_depth : 0;
    const int max_depth = explicit_depth ? h->wanted_sparse : sparse_auto_max_depth(h->query_length);
    if (max_depth <= flat_depth || max_depth < kSparseMinDepth) return explicit_depth ? fail(h, MSBWT_ERR_INVALID_ARG, "sparse table depth must exceed the direct table's") : MSBWT_OK;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return MSBWT_OK;
    struct Temps {
        void *work = nullptr, *counts = nullptr, *lines = nullptr, *side = nullptr;
        ~Temps() {
            for (void *p : {work, counts, lines, side})
                if (p) (void)hipFree(p);
        }
    } tmp;
    // (never more scratch than the index can fill: the nodes of a level are disjoint non-empty ranges, at most `total` of them -- a toy
    // index must not pay for a 6 GB allocation per build)
    const size_t work_bytes = std::min<size_t>(sparse_work_bytes(free_b), 4096 + 2 * 24 * size_t(std::max<uint64_t>(h->totals.total + 1024, 4096)));
    auto optional = [&](hipError_t e, const char *what) -> int {  // an optional structure gives way; an explicit wish does not
        (void)hipGetLastError();
        if (explicit_depth) return hip_fail(h, e, what);
        if (verbose) std::fprintf(stderr, "[msbwt] sparse table: %s: %s -- none built\n", what, hipGetErrorString(e));
        return MSBWT_OK;
    };
    hipError_t e = hipMalloc(&tmp.work, work_bytes);
    if (e != hipSuccess) return optional(e, "scratch");
    // (the frontiers start out as zeros, not as whatever the allocation held: a node {0, 0, 0} is harmless wherever it is read)
    e = hipMemsetAsync(tmp.work, 0, work_bytes, h->stream);
    if (e != hipSuccess) return optional(e, "scratch");
    SparseBuildReport rep;
    e = sparse_count_levels(view_of(h), flat, flat_depth, max_depth, tmp.work, work_bytes, &rep, h->stream);
    if (e != hipSuccess) return optional(e, "sizing pass");
    if (verbose) {
        std::fprintf(stderr, "[msbwt] sparse table: distinct suffixes by length:");
        for (int d = flat_depth; d <= max_depth; ++d)
            if (rep.distinct[d]) std::fprintf(stderr, " %d: %llu (%llu wide, %llu once)", d, (unsigned long long)rep.distinct[d], (unsigned long long)rep.escapes[d], (unsigned long long)rep.singles[d]);
        std::fprintf(stderr, "\n");
    }
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return MSBWT_OK;
    const uint64_t spare = keep_free + total_b / 8;  // an eighth of the HBM stays free for the caller's batches
    // (an explicit depth wins over a memory budget, like the other explicit settings: only the HBM itself limits it)
    const uint64_t avail = std::min<uint64_t>(explicit_depth ? kNoBudget : allowance, uint64_t(free_b) > spare ? uint64_t(free_b) - spare : 0);
    // the depth: a pure function of the counts and the bytes (sparse_policy.hpp, pinned by a CPU test through msbwt_auto_sparse_depth)
    // (the two-tier form sends the suffixes that occur once down the direct table's path: it needs that table's side array for escape lines)
    const int tiers = h->wanted_table_side == 0 ? 0 : h->wanted_tiers;
    const SparseChoice choice = choose_sparse_depth(rep.distinct, rep.escapes, flat_depth, max_depth, avail, explicit_depth ? max_depth : 0, rep.singles, tiers);
    const int chosen = choice.depth;
    uint64_t nbuckets = choice.nbuckets;
    if (!chosen && explicit_depth) return fail(h, MSBWT_ERR_HIP, "the sparse table of the requested depth does not fit in HBM");
    if (!chosen) {
        h->sparse_report = rep;  // (the distinct counts are worth keeping: msbwt_rle_sparse_table_info)
        if (verbose) std::fprintf(stderr, "[msbwt] sparse table: no depth fits %.2f GB -- none built\n", double(avail) / 1e9);
        return MSBWT_OK;
    }
    const uint64_t nside = rep.escapes[chosen];
    if (nside) {
        e = hipMalloc(&tmp.side, nside * 16);
        if (e != hipSuccess) return optional(e, "side array");
    }
    for (int attempt = 0;; ++attempt) {
        const int probe = sparse_probe_limit(chosen, nbuckets);
        if (probe < 1) return optional(hipErrorInvalidValue, "bucket count");
        const uint64_t lines = nbuckets + uint64_t(probe);
        e = hipMalloc(&tmp.lines, lines * 128);
        if (e == hipSuccess) e = hipMalloc(&tmp.counts, lines * sizeof(uint32_t));
        if (e != hipSuccess) return optional(e, "bucket lines");
        e = sparse_fill(view_of(h), flat, flat_depth, chosen, choice.tier, tmp.lines, nbuckets, uint32_t(probe), tmp.side, nside, tmp.counts, tmp.work, work_bytes, &rep, h->stream);
        if (e == hipSuccess) {
            h->sparse_probe = uint32_t(probe);
            h->sparse_bytes = lines * 128;
            break;
        }
        if (e != hipErrorInvalidValue || attempt == 3) return optional(e, "fill pass");
        (void)hipFree(tmp.lines);  // some entry found no slot within the probe limit: a quarter more buckets
        (void)hipFree(tmp.counts);
        tmp.lines = tmp.counts = nullptr;
        nbuckets += nbuckets / 4;
        // the larger table must still fit what the first one was chosen within (the budget, the eighth of HBM left to the caller) and the format
        const uint64_t again = nbuckets + kSparseMaxProbe;
        if (again > 0xFFFFFFFFull || again * 128 + again * sizeof(uint32_t) + nside * 16 > avail) return optional(hipErrorOutOfMemory, "fill pass (no room for more buckets)");
    }
    h->d_sparse = tmp.lines;
    h->d_sparse_side = tmp.side;
    tmp.lines = tmp.side = nullptr;
    h->sparse_side_bytes = nside * 16;
    h->sparse_nbuckets = uint32_t(nbuckets);
    h->sparse_depth = chosen;
    h->sparse_tier = choice.tier;
    h->sparse_report = rep;
    if (verbose)
        std::fprintf(stderr, "[msbwt] sparse table: depth %d%s, %llu entries in %u buckets (%.2f per bucket, %llu displaced, %llu in the side array, %llu in the filters), %.2f GB\n", chosen,
                     choice.tier ? " two-tier" : "", (unsigned long long)rep.entries, h->sparse_nbuckets, double(rep.entries) / double(nbuckets), (unsigned long long)rep.displaced,
                     (unsigned long long)rep.nescapes, (unsigned long long)rep.filtered, double(h->sparse_bytes + h->sparse_side_bytes) / 1e9);
    // ---- a second, shallower level for the queries this table is too deep for (sparse_for, kernels.hpp) ----------------------------------
    // With k undeclared the table above is 23 deep and k = 17..22 fall to the direct table, which stays at packed depth 15 beside a sparse
    // table: measured at human scale (round 6, present k-mers), k = 17 / 19 / 21 run 2.5 / 1.8 / 1.5 x slower than on the index WITHOUT a
    // sparse table (packed depth 17).  Where the deep direct table itself fits (rebuild_table keeps it then) nothing is needed; otherwise the
    // same sizing counts and chunk plan fill a table of the suffixes of 17 symbols -- when it fits what is left, an eighth of the device still
    // free.  A declared k gets none (the caller has said what it will ask), an explicit depth neither.
    // (run blocks are the memory-lean format: no second level there)
    if (!explicit_depth && h->wanted_second != 0 && h->wanted_block_format == kBlocksPlanes && h->query_length == 0 && chosen > kSparseSecondDepth &&
        !deep_direct_fits(h, deep_direct_depth) &&
        hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const uint64_t used = h->sparse_bytes + h->sparse_side_bytes;
        const uint64_t avail2 = std::min<uint64_t>(allowance > used ? allowance - used : 0, uint64_t(free_b) > spare ? uint64_t(free_b) - spare : 0);
        const SparseChoice second = choose_sparse_depth(rep.distinct, rep.escapes, flat_depth, std::min(kSparseSecondDepth, chosen - 1), avail2, 0, rep.singles, tiers);
        if (second.depth) {
            Temps two;  // (its own buffers: the first table's are the handle's by now)
            SparseBuildReport rep2 = rep;
            const int probe2 = sparse_probe_limit(second.depth, second.nbuckets);
            const uint64_t lines2 = second.nbuckets + uint64_t(std::max(probe2, 0)), nside2 = rep.escapes[second.depth];
            e = probe2 >= 1 ? hipSuccess : hipErrorInvalidValue;
            if (e == hipSuccess && nside2) e = hipMalloc(&two.side, nside2 * 16);
            if (e == hipSuccess) e = hipMalloc(&two.lines, lines2 * 128);
            if (e == hipSuccess) e = hipMalloc(&two.counts, lines2 * sizeof(uint32_t));
            if (e == hipSuccess)
                e = sparse_fill(view_of(h), flat, flat_depth, second.depth, second.tier, two.lines, second.nbuckets, uint32_t(probe2), two.side, nside2, two.counts, tmp.work, work_bytes, &rep2,
                                h->stream);
            if (e == hipSuccess) {
                h->d_sparse2 = two.lines;
                h->d_sparse2_side = two.side;
                two.lines = two.side = nullptr;
                h->sparse2_bytes = lines2 * 128;
                h->sparse2_side_bytes = nside2 * 16;
                h->sparse2_entries = rep2.entries;
                h->sparse2_nbuckets = uint32_t(second.nbuckets);
                h->sparse2_probe = uint32_t(probe2);
                h->sparse2_depth = second.depth;
                h->sparse2_tier = second.tier;
                if (verbose)
                    std::fprintf(stderr, "[msbwt] sparse table, second level: depth %d%s, %llu entries in %u buckets, %.2f GB (serves %d <= k < %d)\n", second.depth,
                                 second.tier ? " two-tier" : "", (unsigned long long)rep2.entries, h->sparse2_nbuckets, double(h->sparse2_bytes + h->sparse2_side_bytes) / 1e9,
                                 second.depth, chosen);
            } else {  // optional: an entry without a slot, no memory -- the index simply has no second level
                (void)hipGetLastError();
                if (verbose) std::fprintf(stderr, "[msbwt] sparse table, second level: %s -- none built\n", hipGetErrorString(e));
            }
        }
    }
    return MSBWT_OK;
}

// Direct table beside a sparse one: only queries shorter than the sparse table's entries (and those with '$' / 'N' among their last
// symbols) still read it, so it stays small -- packed depth 15 (4.6 GB) at most.
constexpr int kDirectDepthBesideSparse = 13;  // levels of the flat table (the packed one: + 2)

int rebuild_table(msbwt_rle *h, bool allow_sparse = true) {
    // (run blocks: their sparse table was built at load time from temporary plane and pair blocks -- build_sparse_for_runs -- and does
    // not depend on the direct table rebuilt here; it goes with the index, or by msbwt_rle_set_sparse_table(0))
    if (h->block_format == kBlocksPlanes) release_sparse(h);
    if (h->d_filter) (void)hipFree(h->d_filter);
    h->d_filter = nullptr;
    h->filter_depth = 0;
    if (h->d_table) (void)hipFree(h->d_table);
    h->d_table = nullptr;
    h->table_depth = 0;
    h->table_packed = false;
    h->table_bytes = 0;
    if (h->d_table_side) (void)hipFree(h->d_table_side);
    h->d_table_side = nullptr;
    h->table_side_bytes = h->table_lines = h->table_escape_lines = 0;
    // Automatic depths come from ONE decision (table_policy.hpp, pinned by a CPU test through
    // msbwt_auto_table_depths): beside a pair index the flat table is built as deep as the packed one needs.
    const bool automatic = h->wanted_table_depth < 0;
    int depth = h->wanted_table_depth;
    bool pack = h->d_pair_blocks != nullptr && h->wanted_table_packed > 0;  // an explicit depth is packed only on request
    if (automatic && h->planned) {  // a memory budget is in force: the plan has sized the table (table_policy.hpp, plan_index)
        depth = h->plan.flat;
        pack = h->plan.packed != 0 && h->d_pair_blocks != nullptr && h->wanted_table_packed != 0;
    } else if (automatic) {
        size_t free_b = 0, total_b = 0;
        const bool know_free = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
        // the table budgets against DISJOINT pair blocks: what overlapping ones take on top was checked against the
        // reserve when they were chosen (choose_pair_stride)
        const TableChoice c = choose_table_depths(h->totals.total, h->nblocks * kBlockBytes, know_free ? uint64_t(free_b) + h->pair_overlap_bytes : 0,
                                                  h->d_pair_blocks != nullptr, h->wanted_table_packed != 0);
        depth = c.flat;
        pack = c.packed != 0 || (h->d_pair_blocks != nullptr && h->wanted_table_packed > 0);  // mode 1: whenever a pair index exists
    }
    // The sparse table (sparse_table.hpp) is tried whenever a pair index exists; the automatic direct table then stays small.
    // Should no sparse depth fit (a read set whose error k-mers outnumber the genome's many times over), the direct table is built
    // again as if there were no such thing.
    const bool try_sparse = allow_sparse && h->wanted_sparse != 0 && h->d_pair_blocks != nullptr && h->block_format == kBlocksPlanes && h->totals.total > 0;
    bool capped = false;
    const int uncapped_depth = depth;
    if (try_sparse && automatic && depth > kDirectDepthBesideSparse) {
        depth = kDirectDepthBesideSparse;
        capped = true;
    }
    // (run blocks behind a sparse table -- built at load time, build_sparse_for_runs: the lean format keeps its flat direct table at depth 13,
    // 1 GB instead of 17, for the queries the sparse table does not serve)
    if (h->block_format == kBlocksRuns && h->d_sparse && automatic && depth > kDirectDepthBesideSparse) depth = kDirectDepthBesideSparse;
    if (depth <= 0 && !try_sparse) return MSBWT_OK;
    if (depth + 2 > 18) pack = false;
    auto build_flat = [&](int d) -> int {
        const size_t bytes = (size_t(1) << (2 * d)) * 16;
        void *tab = nullptr;
        HIP_TRY(h, hipMalloc(&tab, bytes));
        hipError_t e = launch_build_table(view_of(h), d, tab, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            (void)hipFree(tab);
            if (e == hipErrorNotSupported) return MSBWT_OK;  // kernel set without a table
            return hip_fail(h, e, "build suffix table");
        }
        h->d_table = tab;
        h->table_depth = d;
        h->table_bytes = bytes;
        return rebuild_filter(h);  // from the flat table, before it may be packed away
    };
    int rc = depth > 0 ? build_flat(depth) : MSBWT_OK;
    if (rc) return rc;
    if (try_sparse) {
        uint64_t allowance = kNoBudget;
        if (h->planned) {  // what the budget leaves once blocks, pair blocks and the direct table are paid for
            const uint64_t direct = (pack && h->d_table) ? packed_table_bytes(depth + 2) : uint64_t(h->table_bytes);
            const uint64_t held = h->nblocks * kBlockBytes + h->pair_bytes + direct;
            allowance = h->memory_budget > held ? h->memory_budget - held : 0;
        }
        rc = build_sparse(h, (pack && h->d_table) ? packed_table_bytes(depth + 2) : 0, allowance, (capped && pack) ? uncapped_depth : 0);
        if (rc) return rc;
        if (!h->d_sparse && capped) {  // no depth fit: the direct table as if there were no sparse one (the distinct counts stay on record)
            const SparseBuildReport counted = h->sparse_report;
            const int again = rebuild_table(h, false);
            h->sparse_report = counted;
            return again;
        }
    }
    // Beside a sparse table the direct table serves the queries SHORTER than that table's entries (and those with '$' / 'N' among their
    // last symbols).  Capped at packed depth 15 those lose against the index without a sparse table (round 6, human scale, present
    // k-mers: k = 17 2.5 x, k = 19 1.8 x, k = 21 1.5 x slower than behind the packed depth-17 table) -- so where HBM is plentiful (a
    // chr20-sized index: 15 GB of 288) the deep direct table is kept AS WELL: nothing is lost for any k.  Not under a memory budget
    // (the plan has sized the table), and not where it would take the eighth of the device left to the caller's batches (deep_direct_fits);
    // there build_sparse has tried a second, shallower sparse level instead.
    if (capped && h->d_sparse && !h->d_sparse2 && h->d_table && pack && deep_direct_fits(h, uncapped_depth)) {
        if (h->d_filter) (void)hipFree(h->d_filter);
        h->d_filter = nullptr;
        h->filter_depth = 0;
        (void)hipFree(h->d_table);
        h->d_table = nullptr;
        h->table_depth = 0;
        h->table_bytes = 0;
        depth = uncapped_depth;
        rc = build_flat(depth);
        if (rc) return rc;
    }
    if (!h->d_table || !pack) return rc;
    // Packed form, two levels deeper (kernels.hpp, launch_pack_table): every level removes a line fetch
    // per query, and the first step after a shallow table is the expensive one (wide ranges straddle
    // blocks).  Needs the pair index.
    const uint64_t pbytes = packed_table_bytes(depth + 2);
    void *packed = nullptr;
    unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(h->d_flags) + kPackScratchOffset);  // [0] escape lines, [1] side cursor
    unsigned long long escapes = 0;
    hipError_t e = hipMalloc(&packed, pbytes);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, 16, h->stream);
    if (e == hipSuccess) e = launch_pack_table(view_of(h), depth, h->d_table, packed, d_cnt, nullptr, nullptr, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&escapes, d_cnt, sizeof escapes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    // Escape lines (some delta beyond 16 bits: the suffixes of high-copy repeats) get their ranges as flat entries in a side
    // array, 512 bytes per line, filled by a second pass over those lines only.  Optional: without it (no memory, or
    // MSBWT_TABLE_SIDE=0) their queries search from scratch.
    void *side = nullptr;
    if (e == hipSuccess && escapes > 0 && h->wanted_table_side != 0) {
        if (hipMalloc(&side, size_t(escapes) * 512) != hipSuccess) {
            (void)hipGetLastError();
            side = nullptr;
        } else {
            e = launch_pack_table(view_of(h), depth, h->d_table, packed, nullptr, side, d_cnt + 1, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        }
    }
    if (e != hipSuccess) {
        if (side) (void)hipFree(side);
        if (packed) (void)hipFree(packed);
        (void)hipGetLastError();
        if (!automatic || h->wanted_table_packed > 0) return hip_fail(h, e, "pack suffix table");
        // optional structure: the handle keeps a flat table -- within the flat table's OWN budget, not the
        // deeper parent that was only meant to be packed away
        const int own = auto_flat_table_depth(h->totals.total, h->nblocks * kBlockBytes);
        if (own < depth) {
            if (h->d_filter) (void)hipFree(h->d_filter);
            h->d_filter = nullptr;
            h->filter_depth = 0;
            (void)hipFree(h->d_table);
            h->d_table = nullptr;
            h->table_depth = 0;
            h->table_bytes = 0;
            return own > 0 ? build_flat(own) : MSBWT_OK;
        }
        return MSBWT_OK;
    }
    (void)hipFree(h->d_table);
    h->d_table = packed;
    h->table_depth = depth + 2;
    h->table_packed = true;
    h->table_bytes = pbytes;
    h->d_table_side = side;
    h->table_side_bytes = side ? uint64_t(escapes) * 512 : 0;
    h->table_lines = pbytes / 128;
    h->table_escape_lines = escapes;
    if (h->d_sparse && h->sparse_tier && escapes > 0 && !side) {
        // the two-tier table sends queries down this table's path, and an escape line without its side entry cannot be followed from
        // there (the query's first symbols are gone): no room for the side array -> the index as if there were no sparse table
        const SparseBuildReport counted = h->sparse_report;
        const int again = rebuild_table(h, false);
        h->sparse_report = counted;
        return again;
    }
    return MSBWT_OK;
}

int rebuild_pair_index(msbwt_rle *h);

// Run blocks with a sparse table (round 6): the table is built while the PLANE blocks of the load are still in HBM -- temporary pair blocks
// (stride 128) and a flat parent table beside them, then the usual sizing and fill passes -- and only the table stays: pair blocks and
// parent are freed again before the planes become run blocks.  Optional: whatever does not fit leaves the index without a sparse table.
// The caller has made the handle look like a plane-block index (d_blocks = planes, totals, nblocks).
int build_sparse_for_runs(msbwt_rle *h) {
    const bool verbose = std::getenv("MSBWT_VERBOSE") != nullptr;
    if (h->wanted_sparse == 0 || h->totals.total == 0) return MSBWT_OK;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return MSBWT_OK;
    const PairIndexSizes sz = pair_index_sizes(h->nblocks, 128);
    // what the conversion will need beside the planes: the run blocks and their overflow blocks -- counted from the planes, as the conversion does
    const uint64_t run_bytes = run_block_count(h->totals.total) * kBlockBytes;
    uint64_t run_peak = run_bytes + run_bytes / 8;
    {
        unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(h->d_flags) + kPackScratchOffset), nover = 0;
        hipError_t e = launch_run_block_count(h->d_blocks, h->nblocks, h->totals.total, d_cnt, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&nover, d_cnt, sizeof nover, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) run_peak = run_bytes + uint64_t(nover) * 256;
        else (void)hipGetLastError();
    }
    const int parent = std::min(kDirectDepthBesideSparse, std::max(0, auto_flat_table_depth(h->totals.total, h->nblocks * kBlockBytes)));
    const uint64_t parent_bytes = parent > 0 ? (uint64_t(1) << (2 * parent)) * 16 : 0;
    if (sz.pair_block_bytes + sz.super_bytes + sz.scratch_bytes + parent_bytes + run_peak > uint64_t(free_b) - uint64_t(free_b) / 32) {
        if (verbose) std::fprintf(stderr, "[msbwt] run blocks: no room for the temporary pair blocks of a sparse-table build -- none built\n");
        return h->wanted_sparse > 0 ? fail(h, MSBWT_ERR_HIP, "the sparse table of the requested depth cannot be built: no room for its temporary pair blocks") : MSBWT_OK;
    }
    const int saved_pair = h->wanted_pair, saved_stride = h->wanted_pair_stride;
    const bool saved_planned = h->planned;
    h->wanted_pair = 1;
    h->wanted_pair_stride = 128;
    h->planned = false;
    int rc = rebuild_pair_index(h);
    h->wanted_pair = saved_pair;
    h->wanted_pair_stride = saved_stride;
    h->planned = saved_planned;
    auto drop_temps = [&]() {
        if (h->d_table) (void)hipFree(h->d_table);
        h->d_table = nullptr;
        h->table_depth = 0;
        h->table_bytes = 0;
        h->table_packed = false;
        if (h->d_filter) (void)hipFree(h->d_filter);
        h->d_filter = nullptr;
        h->filter_depth = 0;
        if (h->d_pair_blocks) (void)hipFree(h->d_pair_blocks);
        if (h->d_pair_super) (void)hipFree(h->d_pair_super);
        h->d_pair_blocks = h->d_pair_super = nullptr;
        h->pair_bytes = 0;
        h->pair_overlap_bytes = 0;
        h->pair_stride = 128;
    };
    if (rc || !h->d_pair_blocks) {
        drop_temps();
        (void)hipGetLastError();
        if (h->wanted_sparse > 0) return rc ? rc : fail(h, MSBWT_ERR_HIP, "the sparse table of the requested depth cannot be built: no pair blocks");
        h->err.clear();
        return MSBWT_OK;
    }
    if (parent > 0) {
        void *tab = nullptr;
        hipError_t e = hipMalloc(&tab, parent_bytes);
        if (e == hipSuccess) e = launch_build_table(view_of(h), parent, tab, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            if (tab) (void)hipFree(tab);
            (void)hipGetLastError();
            tab = nullptr;
        }
        h->d_table = tab;
        h->table_depth = tab ? parent : 0;
        h->table_bytes = tab ? parent_bytes : 0;
        h->table_packed = false;
    }
    // what the budget leaves once the run blocks and their (flat) direct table are paid for; the conversion's peak stays free.  Run blocks
    // are the memory-LEAN format: left to itself the table (with its build scratch) may take twice what the finished blocks take and no more
    // (human scale: 26.8 GB of run blocks -> 53.6 GB: the depth-23 table, 42 GB, or for a declared k = 31 the depth-27 one, 49 GB; a 3e7-symbol
    // stream, whose depth-23 table the tags would force to 4.3 GB: none) -- an explicit depth or a memory budget says otherwise.
    uint64_t allowance = h->wanted_sparse < 0 ? 2 * run_peak : kNoBudget;
    if (h->memory_budget) {
        const uint64_t held = run_peak + parent_bytes;
        allowance = h->memory_budget > held ? h->memory_budget - held : 0;
    }
    rc = build_sparse(h, run_peak, allowance);
    drop_temps();
    if (rc && h->wanted_sparse <= 0) {
        release_sparse(h);
        h->err.clear();
        rc = MSBWT_OK;
    }
    return rc;
}

// A memory budget (msbwt_rle_set_memory_budget) turns the automatic choices into ONE plan, made once the plane blocks are in
// HBM and the data have been probed (table_policy.hpp, plan_index: pair blocks, then the deepest packed table, then
// overlapping pair blocks).  Run blocks have no optional structures but the flat table.
void make_plan(msbwt_rle *h) {
    h->planned = false;
    if (h->memory_budget == 0 || h->block_format != kBlocksPlanes) return;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    const PairIndexSizes wide = pair_index_sizes(h->nblocks, 96), narrow = pair_index_sizes(h->nblocks, 128);
    h->plan = plan_index(h->totals.total, free_b, total_b, h->typical_width, h->memory_budget, narrow.pair_block_bytes + narrow.super_bytes,
                         wide.pair_block_bytes + wide.super_bytes);
    h->planned = true;
}

// How wide is the range of a k-mer that occurs?  (kernels.hpp, launch_probe_widths: the median over a few thousand
// sampled 24-mers; -1 when it cannot be told.)  Cheap: microseconds of kernel time, one 32 KiB read-back.
double probe_typical_width(msbwt_rle *h) {
    if (h->block_format != kBlocksPlanes || h->totals.total == 0) return -1.0;
    uint64_t *d_out = nullptr;
    std::vector<uint64_t> widths(kProbeSamples);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_out), widths.size() * sizeof(uint64_t));
    if (e == hipSuccess) e = launch_probe_widths(view_of(h), kProbeSamples, kProbeSteps, 0x6D73627774ull, d_out, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(widths.data(), d_out, widths.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return -1.0;
    }
    widths.erase(std::remove(widths.begin(), widths.end(), uint64_t(0)), widths.end());  // walks that met '$' / 'N'
    if (widths.size() < 64) return -1.0;
    std::nth_element(widths.begin(), widths.begin() + widths.size() / 2, widths.end());
    return double(widths[widths.size() / 2]);
}

// Pair index (two symbols per step, rank_ops.hpp): 1 byte/symbol on top of the plane blocks,
// built on the device from them.  Default policy: build it when it fits in half of what is
// still free in HBM after the blocks (it is a pure speed-for-memory trade).
int rebuild_pair_index(msbwt_rle *h) {
    if (h->d_pair_blocks) (void)hipFree(h->d_pair_blocks);
    if (h->d_pair_super) (void)hipFree(h->d_pair_super);
    h->d_pair_blocks = h->d_pair_super = nullptr;
    h->pair_bytes = 0;
    h->pair_overlap_bytes = 0;
    if (h->wanted_pair == 0 || h->totals.total == 0 || h->block_format != kBlocksPlanes) return MSBWT_OK;  // built from plane blocks
    // Spacing (table_policy.hpp, choose_pair_stride): an explicit wish is taken literally; otherwise overlapping
    // blocks (stride 96, 1.33 bytes per symbol: ranges up to 32 wide from one line) when they are cheap in HBM, and
    // when they are not, when the DATA keep the ranges of present k-mers wide (probed above) and the bigger blocks
    // fit beside the table that is about to be built.
    size_t free_b = 0, total_b = 0;
    const bool know_free = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    int stride = h->wanted_pair_stride;
    bool by_data = false;
    if (h->planned && h->wanted_pair < 0 && !h->plan.pair) return MSBWT_OK;  // the memory budget has no room for pair blocks
    if (h->planned && stride != 96 && stride != 128) {
        stride = h->plan.stride;
        by_data = true;  // (the plan has checked the fit)
    } else if (stride != 96 && stride != 128) {
        const PairIndexSizes wide = pair_index_sizes(h->nblocks, 96), narrow = pair_index_sizes(h->nblocks, 128);
        const uint64_t bytes96 = wide.pair_block_bytes + wide.super_bytes + wide.scratch_bytes, bytes128 = narrow.pair_block_bytes + narrow.super_bytes;
        const uint64_t after128 = uint64_t(free_b) > bytes128 ? uint64_t(free_b) - bytes128 : 0;
        const uint64_t table_b = expected_table_bytes(h->totals.total, h->nblocks * kBlockBytes, after128, true, h->wanted_table_packed != 0);
        stride = know_free ? choose_pair_stride(bytes96, table_b, free_b, total_b, h->typical_width) : 128;
        by_data = stride == 96 && bytes96 > uint64_t(free_b) / 4;
        if (by_data) h->pair_overlap_bytes = wide.pair_block_bytes + wide.super_bytes - bytes128;
    }
    const PairIndexSizes sz = pair_index_sizes(h->nblocks, stride);
    if (h->wanted_pair < 0 && !by_data) {  // (the data-driven choice has checked its own fit)
        if (!know_free || sz.pair_block_bytes + sz.scratch_bytes > free_b / 2) return MSBWT_OK;
    }
    void *scratch = nullptr;
    hipError_t e = hipMalloc(&h->d_pair_blocks, sz.pair_block_bytes);
    if (e == hipSuccess) e = hipMalloc(&h->d_pair_super, sz.super_bytes);
    if (e == hipSuccess) e = hipMalloc(&scratch, sz.scratch_bytes);
    if (e == hipSuccess) e = build_pair_index(h->d_blocks, h->nblocks, h->totals.start_index, h->d_pair_blocks, h->d_pair_super, scratch, h->stream, stride);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (scratch) (void)hipFree(scratch);
    if (e != hipSuccess) {
        if (h->d_pair_blocks) (void)hipFree(h->d_pair_blocks);
        if (h->d_pair_super) (void)hipFree(h->d_pair_super);
        h->d_pair_blocks = h->d_pair_super = nullptr;
        h->pair_overlap_bytes = 0;
        if (h->wanted_pair < 0 && e == hipErrorOutOfMemory) return MSBWT_OK;  // optional structure
        return hip_fail(h, e, "build pair index");
    }
    h->pair_stride = stride;
    h->pair_bytes = sz.pair_block_bytes + sz.super_bytes;
    return MSBWT_OK;
}

// Index build on the host (kept for MSBWT_BUILD=host and for cross-checking the device
// builder): expand into pinned memory, upload.
int build_on_host(msbwt_rle *h, const uint8_t *rle, size_t n, Totals *t_out) {
    Totals t;
    if (!compute_totals(rle, n, &t)) return fail(h, MSBWT_ERR_INVALID_SYMBOL, "RLE stream holds a symbol code >= 6");
    if (t.total > kMaxTotal) return fail(h, MSBWT_ERR_TOO_LARGE, "BWT has 2^40 symbols or more");
    const uint64_t nblocks = plane_block_count(t.total);
    const size_t bytes = size_t(nblocks) * kBlockBytes;
    uint32_t *host = nullptr;
    // pinned staging so the upload runs at PCIe rate; fall back to pageable memory
    const bool pinned = hipHostMalloc(reinterpret_cast<void **>(&host), bytes, hipHostMallocDefault) == hipSuccess;
    if (!pinned) {
        host = static_cast<uint32_t *>(std::malloc(bytes));
        if (!host) return fail(h, MSBWT_ERR_IO, "out of host memory while building the index");
    }
    build_plane_blocks(rle, n, t, host, 0);
    hipError_t e = hipMalloc(&h->d_blocks, bytes);
    if (e == hipSuccess) e = hipMemcpy(h->d_blocks, host, bytes, hipMemcpyHostToDevice);
    if (pinned) (void)hipHostFree(host);
    else std::free(host);
    if (e != hipSuccess) return hip_fail(h, e, "upload index");
    *t_out = t;
    return MSBWT_OK;
}

// Index build on the device (default): upload the RLE bytes, expand them in HBM
// (device_build.hip).  The expanded index (0.5 B/symbol) never exists on the host.
constexpr int kBuildOnHostInstead = 1000;  // (internal) the run-block path's planes do not fit beside its runs: nothing is left allocated

int build_on_device(msbwt_rle *h, const uint8_t *rle, size_t n, Totals *t_out, bool for_run_blocks = false) {
    struct Temps {
        void *rle = nullptr, *scratch = nullptr, *longs = nullptr;
        ~Temps() {
            if (rle) (void)hipFree(rle);
            if (scratch) (void)hipFree(scratch);
            if (longs) (void)hipFree(longs);
        }
    } tmp;
    HIP_TRY(h, hipMalloc(&tmp.rle, n + 32));
    if (n) HIP_TRY(h, hipMemcpyAsync(tmp.rle, rle, n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMalloc(&tmp.scratch, device_build_scratch_bytes(n)));
    DeviceBuildState st;
    HIP_TRY(h, device_build_pass1(static_cast<const uint8_t *>(tmp.rle), n, tmp.scratch, &st, h->stream));
    uint64_t head[32];  // totals[7], start_index[6], flags, long_count, ...
    HIP_TRY(h, hipMemcpyAsync(head, tmp.scratch, sizeof head, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const uint32_t flags = *reinterpret_cast<const uint32_t *>(&head[13]);
    const uint64_t nlong = head[15];
    if (flags & kBuildBadSymbol) return fail(h, MSBWT_ERR_INVALID_SYMBOL, "RLE stream holds a symbol code >= 6");
    Totals t{};
    uint64_t acc = 0;
    for (int s = 0; s < kAlphabet; ++s) {
        t.symbol_counts[s] = head[s];
        t.start_index[s] = acc;
        acc += head[s];
        t.end_index[s] = acc;
    }
    t.total = acc;
    if ((flags & kBuildTooLarge) || t.total > kMaxTotal || acc != head[6])
        return fail(h, MSBWT_ERR_TOO_LARGE, "BWT has 2^40 symbols or more");
    const uint64_t nblocks = plane_block_count(t.total);
    const size_t bytes = size_t(nblocks) * kBlockBytes;
    if (for_run_blocks) {  // planes AND runs must fit (table_policy.hpp); MSBWT_RUN_BUILD_FREE=<bytes>: tests pretend that much is free
        size_t free_b = 0, total_b = 0;
        uint64_t free_now = (hipMemGetInfo(&free_b, &total_b) == hipSuccess) ? uint64_t(free_b) : ~uint64_t(0);
        if (const char *env = std::getenv("MSBWT_RUN_BUILD_FREE")) free_now = std::strtoull(env, nullptr, 10);
        if (!run_build_fits_device(t.total, free_now)) return kBuildOnHostInstead;
    }
    {
        const hipError_t e = hipMalloc(&h->d_blocks, bytes);
        if (e == hipErrorOutOfMemory && for_run_blocks) {
            (void)hipGetLastError();
            h->d_blocks = nullptr;
            return kBuildOnHostInstead;
        }
        if (e != hipSuccess) return hip_fail(h, e, "hipMalloc(plane blocks)");
    }
    HIP_TRY(h, hipMemsetAsync(h->d_blocks, 0, bytes, h->stream));
    HIP_TRY(h, hipMemcpyAsync(st.d_start_index, t.start_index, sizeof t.start_index, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMalloc(&tmp.longs, device_build_long_run_bytes(nlong)));
    HIP_TRY(h, device_build_pass2(static_cast<const uint8_t *>(tmp.rle), n, st, tmp.longs, nlong, h->d_blocks, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *t_out = t;
    return MSBWT_OK;
}

// Run blocks (the memory-lean format, run_index.hpp).  Default (round 4): on the device -- the RLE bytes are expanded into
// plane blocks as for the default format, every run block is made from its two plane blocks (run_build.hip), and the plane
// blocks are freed: 73 GB for a moment instead of 28 GB at human scale, seconds instead of half a minute.  MSBWT_BUILD=host:
// built on the host and uploaded.
int build_run_index(msbwt_rle *h, const uint8_t *rle, size_t n, Totals *t_out) {
    const char *mode = std::getenv("MSBWT_BUILD");
    auto sparse_while_planes = [&]() -> int {  // (the planes are in h->d_blocks: the handle looks like a plane-block index for a moment)
        if (h->wanted_sparse == 0) return MSBWT_OK;
        h->block_format = kBlocksPlanes;
        h->totals = *t_out;
        h->nblocks = plane_block_count(t_out->total);
        const int rc = build_sparse_for_runs(h);
        h->block_format = kBlocksRuns;
        return rc;
    };
    // The device path holds the plane blocks (0.5 byte per symbol), the RLE bytes and its scratch for a moment, and then the run
    // blocks beside the planes: about 0.8 byte per symbol at its peak against 0.3 for the finished index.  An index whose planes do
    // not fit beside its runs is built on the host instead (as until round 3) -- decided beforehand from the free HBM where the totals
    // can be told (run_build_fits_device), and again on the way should an allocation fail after all.
    bool on_device = !(mode && std::strcmp(mode, "host") == 0);
    if (on_device) {
        int rc = build_on_device(h, rle, n, t_out, true);  // h->d_blocks = plane blocks
        if (rc == kBuildOnHostInstead) {
            if (std::getenv("MSBWT_VERBOSE")) std::fprintf(stderr, "[msbwt] run blocks: the device builder's peak does not fit the free HBM -- built on the host\n");
            on_device = false;
        } else if (rc) {
            return rc;
        }
    }
    if (on_device) {
        const int rcs = sparse_while_planes();
        if (rcs) return rcs;
        void *planes = h->d_blocks;
        h->d_blocks = nullptr;
        const uint64_t nplanes = plane_block_count(t_out->total), nruns = run_block_count(t_out->total);
        unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(h->d_flags) + kPackScratchOffset);
        unsigned long long nover = 0;
        hipError_t e = launch_run_block_count(planes, nplanes, t_out->total, d_cnt, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&nover, d_cnt, sizeof nover, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipMalloc(&h->d_blocks, size_t(nruns) * kBlockBytes);
        if (e == hipSuccess && nover) {
            h->overflow_bytes = uint64_t(nover) * 256;
            e = hipMalloc(&h->d_overflow, h->overflow_bytes);
        }
        if (e == hipSuccess) e = launch_run_block_write(planes, nplanes, t_out->total, d_cnt, h->d_blocks, h->d_overflow, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(planes);
        if (e == hipSuccess) return MSBWT_OK;
        if (h->d_blocks) (void)hipFree(h->d_blocks);
        if (h->d_overflow) (void)hipFree(h->d_overflow);
        h->d_blocks = h->d_overflow = nullptr;
        h->overflow_bytes = 0;
        if (e != hipErrorOutOfMemory) return hip_fail(h, e, "build run blocks on the device");
        (void)hipGetLastError();  // no room for the run blocks beside the planes: the planes are gone now, the host builder takes over
        if (std::getenv("MSBWT_VERBOSE")) std::fprintf(stderr, "[msbwt] run blocks: out of memory on the device path -- built on the host\n");
    }
    Totals t;
    if (!compute_totals(rle, n, &t)) return fail(h, MSBWT_ERR_INVALID_SYMBOL, "RLE stream holds a symbol code >= 6");
    if (t.total > kMaxTotal) return fail(h, MSBWT_ERR_TOO_LARGE, "BWT has 2^40 symbols or more");
    RunIndex ri;
    build_run_blocks(rle, n, t, &ri, 0);
    HIP_TRY(h, hipMalloc(&h->d_blocks, ri.blocks.size() * sizeof(uint32_t)));
    HIP_TRY(h, hipMemcpy(h->d_blocks, ri.blocks.data(), ri.blocks.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (!ri.overflow.empty()) {
        h->overflow_bytes = ri.overflow.size() * sizeof(uint32_t);
        HIP_TRY(h, hipMalloc(&h->d_overflow, h->overflow_bytes));
        HIP_TRY(h, hipMemcpy(h->d_overflow, ri.overflow.data(), h->overflow_bytes, hipMemcpyHostToDevice));
    }
    *t_out = t;
    return MSBWT_OK;
}

// Common tail of both load entry points: build the blocks in HBM, then the pair index and the table.
// The previous index is released FIRST (two human-scale indexes do not fit one GPU): a failed load
// leaves the handle unloaded -- total size and symbol counts 0, queries MSBWT_ERR_NOT_LOADED.
int install(msbwt_rle *h, const uint8_t *rle, size_t n) {
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    release_index(h);
    int rc = ensure_runtime(h);
    if (rc) return rc;
    // MSBWT_VERBOSE=1: one stderr line per load stage (the reference logs its load milestones with
    // log::info!, rle_bwt.rs:62,149,347,383)
    const bool verbose = std::getenv("MSBWT_VERBOSE") != nullptr;
    auto clock = std::chrono::steady_clock::now();
    auto stage = [&](const char *what, uint64_t bytes) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[msbwt] load: %-22s %7.2f s  %8.2f GB in HBM\n", what, std::chrono::duration<double>(now - clock).count(), double(bytes) / 1e9);
        clock = now;
    };
    Totals t{};
    const char *mode = std::getenv("MSBWT_BUILD");
    h->block_format = h->wanted_block_format;
    if (h->block_format == kBlocksRuns) rc = build_run_index(h, rle, n, &t);
    else rc = (mode && std::strcmp(mode, "host") == 0) ? build_on_host(h, rle, n, &t) : build_on_device(h, rle, n, &t);
    if (rc) {
        release_index(h);
        return rc;
    }
    h->totals = t;
    h->nblocks = h->block_format == kBlocksRuns ? run_block_count(t.total) : plane_block_count(t.total);
    h->loaded = true;
    stage(h->block_format == kBlocksRuns ? "run blocks" : "plane blocks", h->nblocks * kBlockBytes + h->overflow_bytes);
    h->typical_width = probe_typical_width(h);
    make_plan(h);
    rc = rebuild_pair_index(h);  // first: the table may be packed with its help
    if (!rc) stage(h->pair_stride == 96 ? "pair blocks, stride 96" : "pair blocks, stride 128", h->pair_bytes);
    if (!rc) rc = rebuild_table(h);
    if (!rc) stage(h->table_packed ? "suffix table, packed" : "suffix table, flat", h->table_bytes);
    if (!rc && h->d_sparse) stage("sparse suffix table", h->sparse_bytes + h->sparse_side_bytes);
    if (rc) {
        release_index(h);
        return rc;
    }
    if (verbose)
        std::fprintf(stderr, "[msbwt] load: %llu symbols, table depth %d, a present %u-mer occurs %.0f times (median), %.2f GB of HBM in all\n",
                     (unsigned long long)t.total, h->table_depth, kProbeSteps, h->typical_width,
                     double(h->nblocks * kBlockBytes + h->overflow_bytes + h->pair_bytes + h->table_bytes + h->sparse_bytes + h->sparse_side_bytes) / 1e9);
    if (verbose)  // where the arrays landed (run-to-run differences of up to 15 % on one box follow the process, not the clocks: profiles/r04_lab)
        std::fprintf(stderr, "[msbwt] load: blocks %p pair blocks %p pair super %p table %p side %p filter %p\n", h->d_blocks, h->d_pair_blocks,
                     static_cast<void *>(h->d_pair_super), h->d_table, h->d_table_side, static_cast<void *>(h->d_filter));
    h->err.clear();
    return MSBWT_OK;
}

// Reads and clears one flag word of the status block on `stream` (synchronises it).
int read_flags(msbwt_rle *h, hipStream_t stream, int which, uint32_t *flags) {
    HIP_TRY(h, hipMemcpyAsync(flags, h->d_flags + which, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(h, hipMemsetAsync(h->d_flags + which, 0, sizeof(uint32_t), stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
    return MSBWT_OK;
}

int flags_to_code(msbwt_rle *h, uint32_t flags) {
    if (flags & kFlagInternal) {
        uint64_t rec[8] = {0};
        (void)hipMemcpy(rec, reinterpret_cast<char *>(h->d_flags) + 64, sizeof rec, hipMemcpyDeviceToHost);
        (void)hipMemset(reinterpret_cast<char *>(h->d_flags) + 64, 0, sizeof rec);
        char buf[256];
        std::snprintf(buf, sizeof buf, "device consistency check failed: range [%llu, %llu) outside the index (rem|slot %#llx, symbols %#llx, wave|lane %#llx)",
                      (unsigned long long)rec[1], (unsigned long long)rec[2], (unsigned long long)rec[3], (unsigned long long)rec[4],
                      (unsigned long long)rec[5]);
        return fail(h, MSBWT_ERR_INTERNAL, buf);
    }
    if (flags & kFlagInvalidSymbol) return fail(h, MSBWT_ERR_INVALID_SYMBOL, "a query holds a symbol code >= 6");
    if (flags & kFlagInvalidRange) return fail(h, MSBWT_ERR_INVALID_RANGE, "a range has l > h or h > total size");
    if (flags & kFlagNarrowOverflow) return fail(h, MSBWT_ERR_OVERFLOW, "a count does not fit the wire width of the all-gather: repeat it with 64 bits");
    return MSBWT_OK;
}

// Folds the recorded start/stop pairs into the running sum.  wait = true (msbwt_rle_kernel_time_ms): waits for the kernels
// they bracket; wait = false (inside an asynchronous launch): only the pairs whose kernel has completed -- in stream order, so
// stopping at the first pending one loses nothing -- and never blocks the caller.
int drain_timing_events(msbwt_rle *h, bool wait = true) {
    int rc = MSBWT_OK;
    size_t kept = 0;
    for (size_t i = 0; i + 1 < h->events.size(); i += 2) {
        float ms = 0.f;
        if (!wait && hipEventQuery(h->events[i + 1]) != hipSuccess) {
            (void)hipGetLastError();
            h->events[kept++] = h->events[i];
            h->events[kept++] = h->events[i + 1];
            continue;
        }
        hipError_t e = hipEventSynchronize(h->events[i + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, h->events[i], h->events[i + 1]);
        if (e == hipSuccess) {
            h->timed_ms += ms;
            h->timed_launches += 1;
        } else if (!rc) {
            rc = hip_fail(h, e, "kernel timing events");
        }
        (void)hipEventDestroy(h->events[i]);
        (void)hipEventDestroy(h->events[i + 1]);
    }
    h->events.resize(kept);
    return rc;
}

// Runs `launch` (which enqueues the count kernel on `stream`); when kernel timing is on, brackets
// it with HIP events on that same stream (read back by msbwt_rle_kernel_time_ms).
template <class Launch>
int timed_launch(msbwt_rle *h, hipStream_t stream, Launch &&launch) {
    if (!h->timing) {
        HIP_TRY(h, launch());
        return MSBWT_OK;
    }
    hipEvent_t start = nullptr, stop = nullptr;
    hipError_t e = hipEventCreate(&start);
    if (e == hipSuccess) e = hipEventCreate(&stop);
    if (e == hipSuccess) e = hipEventRecord(start, stream);
    if (e == hipSuccess) e = launch();
    if (e == hipSuccess) e = hipEventRecord(stop, stream);
    if (e != hipSuccess) {  // nothing is left behind on the error path
        if (start) (void)hipEventDestroy(start);
        if (stop) (void)hipEventDestroy(stop);
        return hip_fail(h, e, "count kernel launch");
    }
    h->events.push_back(start);
    h->events.push_back(stop);
    // a caller that never reads the timer must not grow this forever: completed pairs are folded away without blocking; only a
    // caller with more than 16 x kMaxTimedEvents launches IN FLIGHT is made to wait
    if (h->events.size() >= 2 * kMaxTimedEvents) return drain_timing_events(h, h->events.size() >= 32 * kMaxTimedEvents);
    return MSBWT_OK;
}

// Batch order (order.hip): is this launch to be put through the ordering passes?  Mode 1: whenever they apply (lanes kernel
// on a pair index, 12 <= k <= 64, 4096 <= n < 2^32).  Automatic (-1, the default) is NEVER, on the measurements of round 4
// (profiles/r04_lab/library_batch_order.log, one box, pass off / on): 10^8 read-derived 31-mers over the C4 index 16.8 ->
// 16.1 ms, with repeats 18.9 -> 18.7, C3 3.13 -> 2.97 -- a few per cent where the batch is dense -- against 3.4 -> 9.3 ms on
// random 31-mers (which end in the table: nothing to order for), 5.5 -> 6.6 on 3 x 10^7 queries and 55.7 -> 79.9 at human
// scale.  The ordered search itself is worth 2x (8.4 ms when the caller hands the batch over sorted, msbwt_kmer_order_keys),
// but packing, two bucket passes and returning the counts to the caller's order cost 6 ms of it for 10^8 queries, and the
// search pays 1.7 ms more for a 22-bit order and placed count stores.  Nothing a launch knows beforehand tells the first
// case from the others by a margin that would justify the risk, so the pass stays a switch.
bool order_pays(msbwt_rle *h, const IndexView &v, size_t k, size_t n) {
    if (h->wanted_order <= 0 || v.block_format != kBlocksPlanes || v.pair_blocks == nullptr) return false;
    return k >= 12 && k <= 64 && n >= 4096 && n <= 0xFFFFFFFFull && lanes_serves(v, uint32_t(k));
}

// grows the slot's ordering scratch; false = no memory for it (the launch then runs unordered)
bool ensure_order_scratch(msbwt_rle::TicketSlot &slot, size_t bytes) {
    if (bytes <= slot.order_bytes) return true;
    if (slot.order_scratch) (void)hipFree(slot.order_scratch);  // (waits for the device: nothing still reads it)
    slot.order_scratch = nullptr;
    slot.order_bytes = 0;
    if (hipMalloc(&slot.order_scratch, bytes) != hipSuccess) {
        (void)hipGetLastError();
        slot.order_scratch = nullptr;
        return false;
    }
    slot.order_bytes = bytes;
    return true;
}

uint32_t order_reach(const IndexView &v, size_t k) {  // the symbols the bucket key reads: the table's own index, else 17
    const uint32_t depth = (v.table.entries && v.table.depth > 0 && size_t(v.table.depth) <= k) ? uint32_t(v.table.depth) : 17u;
    return uint32_t(std::min<size_t>(depth, k));
}

int launch_count(msbwt_rle *h, const uint8_t *d_kmers, size_t k, size_t n, uint64_t *d_out, hipStream_t stream, int which) {
    if (k > 0xFFFFFFFFull) return fail(h, MSBWT_ERR_INVALID_ARG, "k does not fit 32 bits");
    return timed_launch(h, stream, [&] {
        return with_slot(h, stream, [&](const IndexView &v, msbwt_rle::TicketSlot &slot) {
            if (order_pays(h, v, k, n)) {
                const OrderPlan plan = plan_order(n, uint32_t(k), order_reach(v, k), uint32_t(h->order_bits), true);
                if (ensure_order_scratch(slot, plan.scratch_bytes)) {
                    const uint64_t *ordered = nullptr;
                    bool inline_place = false;
                    uint64_t *counts = nullptr;
                    hipError_t e = launch_order_batch(plan, d_kmers, nullptr, slot.order_scratch, stream, &ordered, &inline_place, &counts);
                    if (e == hipSuccess) e = launch_count_packed(v, ordered, uint32_t(k), n, counts, nullptr, h->d_flags + which, stream, plan.words + 1, inline_place);
                    if (e == hipSuccess) e = launch_order_finish(plan, slot.order_scratch, d_out, stream);
                    if (e == hipSuccess) e = launch_count_exceptions(plan, v.blocks, v.total, d_kmers, slot.order_scratch, d_out, h->d_flags + which, stream);
                    return e;
                }
            }
            return launch_count_kmers(v, d_kmers, uint32_t(k), n, d_out, h->d_flags + which, stream);
        });
    });
}

// the same for queries handed over as 2-bit words (include/msbwt_hip.h, msbwt_rle_count_kmers_packed_device)
int launch_count_2bit(msbwt_rle *h, const uint64_t *d_packed, size_t k, size_t n, uint64_t *d_out, hipStream_t stream, int which) {
    if (k < 1 || k > 64) return fail(h, MSBWT_ERR_INVALID_ARG, "packed queries need 1 <= k <= 64");
    if (reinterpret_cast<uintptr_t>(d_packed) & 7u) return fail(h, MSBWT_ERR_INVALID_ARG, "packed queries must be 8-byte aligned");
    return timed_launch(h, stream, [&] {
        return with_slot(h, stream, [&](const IndexView &v, msbwt_rle::TicketSlot &slot) {
            if (v.block_format != kBlocksPlanes) {  // run blocks: no kernel reads 2-bit words; unpack into rows first
                const size_t row_bytes = (n * k + 255) / 256 * 256;
                if (!ensure_order_scratch(slot, row_bytes)) return hipErrorOutOfMemory;
                uint8_t *rows = static_cast<uint8_t *>(slot.order_scratch);
                hipError_t e = launch_unpack_rows(d_packed, uint32_t(k), n, rows, stream);
                if (e == hipSuccess) e = launch_count_kmers(v, rows, uint32_t(k), n, d_out, h->d_flags + which, stream);
                return e;
            }
            if (order_pays(h, v, k, n)) {
                const OrderPlan plan = plan_order(n, uint32_t(k), order_reach(v, k), uint32_t(h->order_bits), false);
                if (ensure_order_scratch(slot, plan.scratch_bytes)) {
                    const uint64_t *ordered = nullptr;
                    bool inline_place = false;
                    uint64_t *counts = nullptr;
                    hipError_t e = launch_order_batch(plan, nullptr, d_packed, slot.order_scratch, stream, &ordered, &inline_place, &counts);
                    if (e == hipSuccess) e = launch_count_packed(v, ordered, uint32_t(k), n, counts, nullptr, h->d_flags + which, stream, plan.words + 1, inline_place);
                    if (e == hipSuccess) e = launch_order_finish(plan, slot.order_scratch, d_out, stream);
                    return e;
                }
            }
            return launch_count_packed(v, d_packed, uint32_t(k), n, d_out, nullptr, h->d_flags + which, stream);
        });
    });
}

}  // namespace

extern "C" {

const char *msbwt_version(void) { return kVersion; }

msbwt_rle *msbwt_rle_new_on_device(uint8_t bin_power, int device) {
    msbwt_rle *h = new (std::nothrow) msbwt_rle();
    if (!h) return nullptr;
    h->bin_power = bin_power;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    h->device = device;
    if (const char *env = std::getenv("MSBWT_TABLE_DEPTH")) h->wanted_table_depth = std::max(-1, std::min(std::atoi(env), kMaxTableDepth));
    if (const char *env = std::getenv("MSBWT_TABLE_PACKED")) h->wanted_table_packed = std::atoi(env) ? 1 : 0;
    if (const char *env = std::getenv("MSBWT_TABLE_SIDE")) h->wanted_table_side = std::atoi(env) ? 1 : 0;
    if (const char *env = std::getenv("MSBWT_STREAM_LINES")) h->wanted_streaming = std::strcmp(env, "auto") == 0 ? -1 : (std::atoi(env) ? 1 : 0);
    if (const char *env = std::getenv("MSBWT_QUERY_K")) h->query_length = std::max(0, std::atoi(env));
    if (const char *env = std::getenv("MSBWT_SPARSE_TABLE")) {
        const int d = std::strcmp(env, "auto") == 0 ? -1 : std::atoi(env);
        h->wanted_sparse = (d == 0 || d == -1 || (d >= kSparseMinDepth && d <= kSparseMaxDepth)) ? d : -1;
    }
    if (const char *env = std::getenv("MSBWT_SPARSE_SECOND")) h->wanted_second = std::strcmp(env, "auto") == 0 ? -1 : (std::atoi(env) ? -1 : 0);
    if (const char *env = std::getenv("MSBWT_SPARSE_TIERS")) h->wanted_tiers = std::strcmp(env, "auto") == 0 ? -1 : (std::atoi(env) ? 1 : 0);
    if (const char *env = std::getenv("MSBWT_PAIR_INDEX")) h->wanted_pair = std::atoi(env) ? 1 : 0;
    if (const char *env = std::getenv("MSBWT_PAIR_STRIDE")) h->wanted_pair_stride = std::atoi(env);
    if (const char *env = std::getenv("MSBWT_FILTER")) h->wanted_filter = std::atoi(env) ? -1 : 0;
    if (const char *env = std::getenv("MSBWT_BLOCKS")) h->wanted_block_format = std::strcmp(env, "runs") == 0 ? kBlocksRuns : kBlocksPlanes;
    if (const char *env = std::getenv("MSBWT_MEMORY_BUDGET")) h->memory_budget = std::strtoull(env, nullptr, 10);
    if (const char *env = std::getenv("MSBWT_ORDER")) h->wanted_order = std::strcmp(env, "auto") == 0 ? -1 : (std::atoi(env) ? 1 : 0);
    if (const char *env = std::getenv("MSBWT_ORDER_BITS")) h->order_bits = std::max(1, std::min(std::atoi(env), 36));
    if (const char *env = std::getenv("MSBWT_SEARCH"))
        h->search_kernel = std::strcmp(env, "groups") == 0 ? kSearchGroups : std::strcmp(env, "lanes") == 0 ? kSearchLanes : kSearchAuto;
    return h;
}

msbwt_rle *msbwt_rle_new(uint8_t bin_power) { return msbwt_rle_new_on_device(bin_power, -1); }

void msbwt_rle_free(msbwt_rle *h) {
    if (!h) return;
    {
        DeviceScope scope(h->device);
        release_index(h);
        for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
        for (auto &t : h->tickets) {
            if (t.done) (void)hipEventDestroy(t.done);
            if (t.counters) (void)hipFree(t.counters);
            if (t.order_scratch) (void)hipFree(t.order_scratch);
        }
        h->pipe.release();
        if (h->mail) (void)hipHostFree(h->mail);
        if (h->d_gather) (void)hipFree(h->d_gather);
        for (hipEvent_t e : h->piece_events) (void)hipEventDestroy(e);
        if (h->gather_stream) (void)hipStreamDestroy(h->gather_stream);
        if (h->d_stage) (void)hipFree(h->d_stage);
        if (h->d_flags) (void)hipFree(h->d_flags);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
}

int msbwt_rle_load_vector(msbwt_rle *h, const uint8_t *rle_bytes, size_t len) {
    if (!h || (!rle_bytes && len)) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    return install(h, rle_bytes, len);
}

int msbwt_rle_load_numpy_file(msbwt_rle *h, const char *utf8_path) {
    if (!h || !utf8_path) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    MappedPayload payload;  // mapped, not read: the upload reads it straight from the page cache
    std::string msg;
    switch (map_npy_payload(utf8_path, &payload, &msg)) {
        case NpyStatus::kOk: break;
        case NpyStatus::kIo: return fail(h, MSBWT_ERR_IO, msg);
        case NpyStatus::kUnexpectedEof: return fail(h, MSBWT_ERR_UNEXPECTED_EOF, msg);
        case NpyStatus::kBadHeader: return fail(h, MSBWT_ERR_BAD_HEADER, msg);
    }
    return install(h, payload.data(), payload.size());
}

uint64_t msbwt_rle_get_symbol_count(const msbwt_rle *h, uint8_t symbol) {
    return (h && symbol < kAlphabet) ? h->totals.symbol_counts[symbol] : 0;
}

uint64_t msbwt_rle_get_total_size(const msbwt_rle *h) { return h ? h->totals.total : 0; }

int msbwt_rle_count_kmers_device(const msbwt_rle *ch, const void *d_kmers, size_t k, size_t n,
                                 void *d_out_counts, void *hip_stream) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    if (n && (!d_out_counts || (!d_kmers && k))) return fail(h, MSBWT_ERR_INVALID_ARG, "null device pointer");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    return launch_count(h, static_cast<const uint8_t *>(d_kmers), k, n, static_cast<uint64_t *>(d_out_counts),
                        static_cast<hipStream_t>(hip_stream), kDeviceFlags);
}

int msbwt_rle_count_kmers_packed_device(const msbwt_rle *ch, const void *d_kmers2bit, size_t k, size_t n, void *d_out_counts, void *hip_stream) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    if (n && (!d_out_counts || !d_kmers2bit)) return fail(h, MSBWT_ERR_INVALID_ARG, "null device pointer");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    if (n == 0) return MSBWT_OK;
    return launch_count_2bit(h, static_cast<const uint64_t *>(d_kmers2bit), k, n, static_cast<uint64_t *>(d_out_counts), static_cast<hipStream_t>(hip_stream),
                             kDeviceFlags);
}

int msbwt_rle_count_kmers_packed(const msbwt_rle *ch, const uint64_t *kmers2bit, size_t k, size_t n, void *out_counts, int count_bits) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    if (k < 1 || k > 64 || (count_bits != 64 && count_bits != 32)) return fail(h, MSBWT_ERR_INVALID_ARG, "packed queries need 1 <= k <= 64 and 64- or 32-bit counts");
    if (n && (!out_counts || !kmers2bit)) return fail(h, MSBWT_ERR_INVALID_ARG, "null pointer");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    if (n == 0) return MSBWT_OK;
    int rc = ensure_runtime(h);
    if (rc) return rc;
    // pipelined like msbwt_rle_count_kmers: 8 (16) bytes per query in, 8 or 4 out; the 32-bit form counts into a device
    // buffer and narrows on the kernels' stream (a count beyond 32 bits is reported, not truncated silently)
    const size_t words = k > 32 ? 2 : 1, chunk = size_t(1) << 22;
    if (count_bits == 32 && (rc = ensure_stage(h, std::min(n, chunk) * sizeof(uint64_t))) != MSBWT_OK) return rc;
    std::vector<HostArray> ins(1), outs(1);
    ins[0].in = kmers2bit;
    ins[0].item_bytes = words * sizeof(uint64_t);
    outs[0].out = out_counts;
    outs[0].item_bytes = size_t(count_bits / 8);
    int launch_rc = MSBWT_OK;
    const hipError_t e = h->pipe.run(n, chunk, ins, outs, h->stream,
                                     [&](size_t, size_t m, void *const *d_in, void *const *d_out, hipStream_t stream) -> hipError_t {
                                         uint64_t *d_counts = count_bits == 64 ? static_cast<uint64_t *>(d_out[0]) : static_cast<uint64_t *>(h->d_stage);
                                         launch_rc = launch_count_2bit(h, static_cast<const uint64_t *>(d_in[0]), k, m, d_counts, stream, kHostFlags);
                                         if (launch_rc) return hipErrorUnknown;
                                         return count_bits == 64 ? hipSuccess : launch_narrow_counts32(d_counts, static_cast<uint32_t *>(d_out[0]), m, h->d_flags + kHostFlags, stream);
                                     });
    if (launch_rc) return launch_rc;
    if (e != hipSuccess) return hip_fail(h, e, "count_kmers_packed pipeline");
    uint32_t flags = 0;
    rc = read_flags(h, h->stream, kHostFlags, &flags);
    return rc ? rc : flags_to_code(h, flags);
}

int msbwt_kmers_pack_2bit(const uint8_t *kmers, size_t k, size_t n, uint64_t *out_words) {
    if (k < 1 || k > 64 || (n && (!kmers || !out_words))) return MSBWT_ERR_INVALID_ARG;
    const size_t words = k > 32 ? 2 : 1;
    for (size_t q = 0; q < n; ++q) {
        uint64_t w[2] = {0, 0};
        for (size_t t = 0; t < k; ++t) {  // step t = symbol k-1-t, two bits each from bit 0 of word 0 up
            const uint8_t s = kmers[q * k + k - 1 - t];
            if (s != 1 && s != 2 && s != 3 && s != 5) return MSBWT_ERR_INVALID_SYMBOL;
            w[t >> 5] |= uint64_t(s - 1 - (s >> 2)) << (2 * (t & 31));
        }
        for (size_t i = 0; i < words; ++i) out_words[q * words + i] = w[i];
    }
    return MSBWT_OK;
}

int msbwt_rle_set_batch_order(msbwt_rle *h, int mode) {
    if (!h || mode < -1 || mode > 1) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->wanted_order = mode;
    return MSBWT_OK;
}

int msbwt_rle_get_batch_order(const msbwt_rle *h) { return h ? h->wanted_order : 0; }

int msbwt_rle_batch_order_for(const msbwt_rle *ch, size_t k, size_t n) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h || !h->loaded) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    return order_pays(h, view_of(h), k, n) ? 1 : 0;
}

int msbwt_rle_constrain_ranges_device(const msbwt_rle *ch, const void *d_syms, const void *d_l, const void *d_h,
                                      size_t n, void *d_out_l, void *d_out_h, void *hip_stream) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    if (n && (!d_syms || !d_l || !d_h || !d_out_l || !d_out_h)) return fail(h, MSBWT_ERR_INVALID_ARG, "null device pointer");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    HIP_TRY(h, launch_constrain_ranges(view_of(h), static_cast<const uint8_t *>(d_syms),
                                       static_cast<const uint64_t *>(d_l), static_cast<const uint64_t *>(d_h), n,
                                       static_cast<uint64_t *>(d_out_l), static_cast<uint64_t *>(d_out_h),
                                       h->d_flags + kDeviceFlags, static_cast<hipStream_t>(hip_stream)));
    return MSBWT_OK;
}

// Enqueues the fused read -> k-mer count kernel; the caller holds h->mu and has made the
// handle's device current.
static int launch_read_kmers_locked(msbwt_rle *h, const void *d_reads, size_t read_len, size_t n_reads, size_t k,
                                    int ascii, void *d_out_fwd, void *d_out_rc, hipStream_t stream, int which) {
    return timed_launch(h, stream, [&] {
        return with_tickets(h, stream, [&](const IndexView &v) {
            return launch_count_read_kmers(v, static_cast<const uint8_t *>(d_reads), uint32_t(read_len), n_reads, uint32_t(k), ascii != 0,
                                           static_cast<uint64_t *>(d_out_fwd), static_cast<uint64_t *>(d_out_rc), h->d_flags + which, stream);
        });
    });
}

int msbwt_rle_count_read_kmers_device(const msbwt_rle *ch, const void *d_reads, size_t read_len, size_t n_reads,
                                      size_t k, int ascii, void *d_out_fwd, void *d_out_rc, void *hip_stream) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    if (k < 1 || k > 64 || k > read_len || read_len > 0xFFFFFFFFull || (!d_out_fwd && !d_out_rc) || (n_reads && !d_reads))
        return fail(h, MSBWT_ERR_INVALID_ARG, "count_read_kmers needs 1 <= k <= min(64, read_len) and an output");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    return launch_read_kmers_locked(h, d_reads, read_len, n_reads, k, ascii, d_out_fwd, d_out_rc,
                                    static_cast<hipStream_t>(hip_stream), kDeviceFlags);
}

int msbwt_rle_count_read_kmers(const msbwt_rle *ch, const uint8_t *reads, size_t read_len, size_t n_reads, size_t k,
                               int ascii, uint64_t *out_fwd, uint64_t *out_rc) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);  // held throughout: the staging buffer is per handle
    if (k < 1 || k > 64 || k > read_len || read_len > 0xFFFFFFFFull || (!out_fwd && !out_rc) || (n_reads && !reads))
        return fail(h, MSBWT_ERR_INVALID_ARG, "count_read_kmers needs 1 <= k <= min(64, read_len) and an output");
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    const size_t windows = read_len - k + 1;
    // pipelined: chunks of reads holding ~2 Mi windows travel host -> pinned -> HBM -> pinned -> host
    const size_t chunk = std::max<size_t>(1, (size_t(1) << 21) / windows);
    std::vector<HostArray> ins(1), outs;
    ins[0].in = reads;
    ins[0].item_bytes = read_len;
    if (out_fwd) { HostArray a; a.out = out_fwd; a.item_bytes = windows * sizeof(uint64_t); outs.push_back(a); }
    if (out_rc) { HostArray a; a.out = out_rc; a.item_bytes = windows * sizeof(uint64_t); outs.push_back(a); }
    int launch_rc = MSBWT_OK;
    const hipError_t e = h->pipe.run(n_reads, chunk, ins, outs, h->stream,
                                     [&](size_t, size_t m, void *const *d_in, void *const *d_out, hipStream_t stream) -> hipError_t {
                                         void *d_f = out_fwd ? d_out[0] : nullptr, *d_c = out_rc ? d_out[out_fwd ? 1 : 0] : nullptr;
                                         launch_rc = launch_read_kmers_locked(h, d_in[0], read_len, m, k, ascii, d_f, d_c, stream, kHostFlags);
                                         return launch_rc ? hipErrorUnknown : hipSuccess;
                                     });
    if (launch_rc) return launch_rc;
    if (e != hipSuccess) return hip_fail(h, e, "count_read_kmers pipeline");
    uint32_t flags = 0;
    const int rc = read_flags(h, h->stream, kHostFlags, &flags);
    return rc ? rc : flags_to_code(h, flags);
}

int msbwt_rle_count_ragged_read_kmers(const msbwt_rle *ch, const uint8_t *reads, const uint64_t *read_offsets,
                                      size_t n_reads, size_t k, int ascii, uint64_t *out_fwd, uint64_t *out_rc,
                                      uint64_t *out_windows) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (k < 1 || k > 64 || (n_reads && (!read_offsets || !reads)))
        return fail(h, MSBWT_ERR_INVALID_ARG, "count_ragged_read_kmers needs 1 <= k <= 64 and offsets");
    // window prefix: read r owns [win[r], win[r+1])
    std::vector<uint64_t> win(n_reads + 1, 0);
    for (size_t r = 0; r < n_reads; ++r) {
        if (read_offsets[r + 1] < read_offsets[r]) return fail(h, MSBWT_ERR_INVALID_ARG, "read offsets must not decrease");
        const uint64_t len = read_offsets[r + 1] - read_offsets[r];
        win[r + 1] = win[r] + (len >= k ? len - k + 1 : 0);
    }
    const uint64_t total_windows = win[n_reads];
    if (out_windows) *out_windows = total_windows;
    if (!out_fwd && !out_rc) return MSBWT_OK;
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    if (total_windows == 0) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    // batches of whole reads holding at most ~4 Mi windows (at least one read)
    uint32_t all_flags = 0;
    for (size_t r0 = 0; r0 < n_reads;) {
        size_t r1 = r0 + 1;
        while (r1 < n_reads && win[r1 + 1] - win[r0] <= (uint64_t(1) << 22)) ++r1;
        const uint64_t nwin = win[r1] - win[r0], nbytes = read_offsets[r1] - read_offsets[r0];
        const size_t m = r1 - r0;
        if (nwin) {
            const size_t off_bytes = (m + 1) * sizeof(uint64_t);
            const size_t read_bytes = (size_t(nbytes) + 15) / 16 * 16;
            int rc = ensure_stage(h, read_bytes + 2 * off_bytes + 2 * nwin * sizeof(uint64_t) + 64);
            if (rc) return rc;
            uint8_t *d_r = static_cast<uint8_t *>(h->d_stage);
            uint64_t *d_roff = reinterpret_cast<uint64_t *>(d_r + read_bytes);
            uint64_t *d_woff = d_roff + (m + 1), *d_f = d_woff + (m + 1), *d_c = d_f + nwin;
            std::vector<uint64_t> roff(m + 1), woff(m + 1);  // rebased to the batch
            for (size_t i = 0; i <= m; ++i) {
                roff[i] = read_offsets[r0 + i] - read_offsets[r0];
                woff[i] = win[r0 + i] - win[r0];
            }
            HIP_TRY(h, hipMemcpyAsync(d_r, reads + read_offsets[r0], nbytes, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(d_roff, roff.data(), off_bytes, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(d_woff, woff.data(), off_bytes, hipMemcpyHostToDevice, h->stream));
            rc = timed_launch(h, h->stream, [&] {
                return with_tickets(h, h->stream, [&](const IndexView &v) {
                    return launch_count_ragged_read_kmers(v, d_r, d_roff, d_woff, m, nwin, uint32_t(k), ascii != 0, out_fwd ? d_f : nullptr,
                                                          out_rc ? d_c : nullptr, h->d_flags, h->stream);
                });
            });
            if (rc) return rc;
            if (out_fwd) HIP_TRY(h, hipMemcpyAsync(out_fwd + win[r0], d_f, nwin * 8, hipMemcpyDeviceToHost, h->stream));
            if (out_rc) HIP_TRY(h, hipMemcpyAsync(out_rc + win[r0], d_c, nwin * 8, hipMemcpyDeviceToHost, h->stream));
            uint32_t flags = 0;
            rc = read_flags(h, h->stream, kHostFlags, &flags);  // synchronises: roff/woff may go out of scope
            if (rc) return rc;
            all_flags |= flags;
        }
        r0 = r1;
    }
    return flags_to_code(h, all_flags);
}

int msbwt_rle_device_status(const msbwt_rle *ch, void *hip_stream) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->d_flags) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    uint32_t flags = 0;
    int rc = read_flags(h, static_cast<hipStream_t>(hip_stream), kDeviceFlags, &flags);
    return rc ? rc : flags_to_code(h, flags);
}

int msbwt_rle_count_kmers(const msbwt_rle *ch, const uint8_t *kmers, size_t k, size_t n, uint64_t *out_counts) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    if (n && (!out_counts || (!kmers && k))) return fail(h, MSBWT_ERR_INVALID_ARG, "null pointer");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    if (n == 0) return MSBWT_OK;
    if (n <= kMailQueries && n * k <= kMailKmerBytes) {
        // the trait's single-query shape (msbwt_core.rs:124: one k-mer per call) and other tiny batches: through the
        // mailbox.  Every error ends its query with u64::MAX -- no real count is that large -- so the status word is
        // only read back when one shows up.
        int rc = ensure_runtime(h);
        if (!rc) rc = ensure_mail(h);
        if (rc) return rc;
        uint64_t *counts = reinterpret_cast<uint64_t *>(h->mail + kMailCounts);
        IndexView v = view_of(h);  // no ticket counters: at most one tile
        // The lanes kernel announces completion in the mailbox itself: poll that word (about 5 us cheaper than a stream
        // synchronisation on this runtime); a single query even travels inside the kernel arguments.
        const bool poll = k <= 0xFFFFFFFFull && lanes_serves(v, uint32_t(k));
        const bool inlined = poll && n == 1;
        if (k && !inlined) std::memcpy(h->mail + kMailKmers, kmers, n * k);
        volatile uint64_t *done = reinterpret_cast<volatile uint64_t *>(h->mail + kMailDone);
        const uint64_t seq = ++h->mail_seq;
        if (poll) {
            v.done = reinterpret_cast<uint64_t *>(h->d_mail + kMailDone);
            v.done_seq = seq;
        }
        rc = timed_launch(h, h->stream, [&] {
            return launch_count_kmers(v, h->d_mail + kMailKmers, uint32_t(k), n, reinterpret_cast<uint64_t *>(h->d_mail + kMailCounts),
                                      h->d_flags + kHostFlags, h->stream, inlined ? kmers : nullptr);
        });
        if (rc) return rc;
        bool seen = false;
        if (poll) {  // bounded: a kernel that never reports (a fault) is left to the synchronisation below, which says why
            const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
            for (unsigned spins = 0; !(seen = *done == seq); ++spins)
                if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() > give_up) break;
        }
        if (!seen) HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::atomic_thread_fence(std::memory_order_acquire);  // the counts are read after the completion word
        bool flagged = false;
        for (size_t i = 0; i < n; ++i) {
            out_counts[i] = counts[i];
            flagged |= counts[i] == ~0ull;
        }
        if (!flagged) return MSBWT_OK;
        uint32_t flags = 0;
        rc = read_flags(h, h->stream, kHostFlags, &flags);
        return rc ? rc : flags_to_code(h, flags);
    }
    // pipelined: chunks of 2 Mi queries travel host -> pinned -> HBM -> pinned -> host, copies and
    // kernels overlapping on three streams (host_pipeline.hpp)
    std::vector<HostArray> ins(1), outs(1);
    ins[0].in = kmers;
    ins[0].item_bytes = k;
    outs[0].out = out_counts;
    outs[0].item_bytes = sizeof(uint64_t);
    int launch_rc = MSBWT_OK;
    const hipError_t e = h->pipe.run(n, size_t(1) << 21, ins, outs, h->stream,
                                     [&](size_t, size_t m, void *const *d_in, void *const *d_out, hipStream_t stream) -> hipError_t {
                                         launch_rc = launch_count(h, static_cast<const uint8_t *>(d_in[0]), k, m,
                                                                  static_cast<uint64_t *>(d_out[0]), stream, kHostFlags);
                                         return launch_rc ? hipErrorUnknown : hipSuccess;
                                     });
    if (launch_rc) return launch_rc;
    if (e != hipSuccess) return hip_fail(h, e, "count_kmers pipeline");
    uint32_t flags = 0;
    const int rc = read_flags(h, h->stream, kHostFlags, &flags);
    return rc ? rc : flags_to_code(h, flags);
}

int msbwt_rle_constrain_ranges(const msbwt_rle *ch, const uint8_t *syms, const uint64_t *l, const uint64_t *hh,
                               size_t n, uint64_t *out_l, uint64_t *out_h) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    if (n && (!syms || !l || !hh || !out_l || !out_h)) return fail(h, MSBWT_ERR_INVALID_ARG, "null pointer");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    if (n == 0) return MSBWT_OK;
    if (n <= kMailQueries) {  // BWT::constrain_range, one range per call (msbwt_core.rs:99): through the mailbox, as above
        int rc = ensure_runtime(h);
        if (!rc) rc = ensure_mail(h);
        if (rc) return rc;
        std::memcpy(h->mail + kMailSyms, syms, n);
        std::memcpy(h->mail + kMailL, l, n * sizeof(uint64_t));
        std::memcpy(h->mail + kMailH, hh, n * sizeof(uint64_t));
        IndexView v = view_of(h);
        volatile uint64_t *done = reinterpret_cast<volatile uint64_t *>(h->mail + kMailDone);
        const uint64_t seq = ++h->mail_seq;
        const bool poll = n <= 8;  // one wave of 8-lane groups: the kernel announces completion in the mailbox
        if (poll) {
            v.done = reinterpret_cast<uint64_t *>(h->d_mail + kMailDone);
            v.done_seq = seq;
        }
        HIP_TRY(h, launch_constrain_ranges(v, h->d_mail + kMailSyms, reinterpret_cast<const uint64_t *>(h->d_mail + kMailL),
                                           reinterpret_cast<const uint64_t *>(h->d_mail + kMailH), n, reinterpret_cast<uint64_t *>(h->d_mail + kMailOutL),
                                           reinterpret_cast<uint64_t *>(h->d_mail + kMailOutH), h->d_flags + kHostFlags, h->stream));
        bool seen = false;
        if (poll) {
            const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
            for (unsigned spins = 0; !(seen = *done == seq); ++spins)
                if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() > give_up) break;
        }
        if (!seen) HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::atomic_thread_fence(std::memory_order_acquire);
        const uint64_t *ol = reinterpret_cast<const uint64_t *>(h->mail + kMailOutL), *oh = reinterpret_cast<const uint64_t *>(h->mail + kMailOutH);
        bool flagged = false;
        for (size_t i = 0; i < n; ++i) {
            out_l[i] = ol[i];
            out_h[i] = oh[i];
            flagged |= ol[i] == ~0ull;  // an invalid symbol or range ends as {u64::MAX, u64::MAX}
        }
        if (!flagged) return MSBWT_OK;
        uint32_t flags = 0;
        rc = read_flags(h, h->stream, kHostFlags, &flags);
        return rc ? rc : flags_to_code(h, flags);
    }
    std::vector<HostArray> ins(3), outs(2);
    ins[0].in = syms; ins[0].item_bytes = 1;
    ins[1].in = l; ins[1].item_bytes = sizeof(uint64_t);
    ins[2].in = hh; ins[2].item_bytes = sizeof(uint64_t);
    outs[0].out = out_l; outs[0].item_bytes = sizeof(uint64_t);
    outs[1].out = out_h; outs[1].item_bytes = sizeof(uint64_t);
    const hipError_t e = h->pipe.run(n, size_t(1) << 21, ins, outs, h->stream,
                                     [&](size_t, size_t m, void *const *d_in, void *const *d_out, hipStream_t stream) -> hipError_t {
                                         return launch_constrain_ranges(view_of(h), static_cast<const uint8_t *>(d_in[0]),
                                                                        static_cast<const uint64_t *>(d_in[1]), static_cast<const uint64_t *>(d_in[2]), m,
                                                                        static_cast<uint64_t *>(d_out[0]), static_cast<uint64_t *>(d_out[1]),
                                                                        h->d_flags + kHostFlags, stream);
                                     });
    if (e != hipSuccess) return hip_fail(h, e, "constrain_ranges pipeline");
    uint32_t flags = 0;
    const int rc = read_flags(h, h->stream, kHostFlags, &flags);
    return rc ? rc : flags_to_code(h, flags);
}

int msbwt_rle_constrain_range(const msbwt_rle *h, uint8_t sym, uint64_t l, uint64_t hh, uint64_t *out_l,
                              uint64_t *out_h) {
    if (!out_l || !out_h) return MSBWT_ERR_INVALID_ARG;
    return msbwt_rle_constrain_ranges(h, &sym, &l, &hh, 1, out_l, out_h);
}

int msbwt_rle_count_kmer(const msbwt_rle *h, const uint8_t *kmer, size_t k, uint64_t *out_count) {
    if (!out_count) return MSBWT_ERR_INVALID_ARG;
    return msbwt_rle_count_kmers(h, kmer, k, 1, out_count);
}

// ---- several devices of one node: replicas of one index, batches sharded over them --------------
msbwt_rle *msbwt_rle_replicate(const msbwt_rle *csrc, int device) {
    msbwt_rle *src = const_cast<msbwt_rle *>(csrc);
    if (!src) return nullptr;
    std::lock_guard<std::mutex> lock(src->mu);
    if (!src->loaded) {
        fail(src, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
        return nullptr;
    }
    msbwt_rle *h = msbwt_rle_new_on_device(src->bin_power, device);
    if (!h) return nullptr;
    h->wanted_table_depth = src->wanted_table_depth;
    h->wanted_table_packed = src->wanted_table_packed;
    h->wanted_table_side = src->wanted_table_side;
    h->wanted_sparse = src->wanted_sparse;
    h->wanted_tiers = src->wanted_tiers;
    h->wanted_second = src->wanted_second;
    h->query_length = src->query_length;
    h->wanted_streaming = src->wanted_streaming;
    h->wanted_block_format = src->wanted_block_format;
    h->block_format = src->block_format;
    h->wanted_pair = src->wanted_pair;
    h->wanted_pair_stride = src->wanted_pair_stride;
    h->pair_stride = src->pair_stride;
    h->wanted_filter = src->wanted_filter;
    h->search_kernel = src->search_kernel;
    h->wanted_order = src->wanted_order;
    h->order_bits = src->order_bits;
    h->memory_budget = src->memory_budget;
    h->planned = src->planned;
    h->plan = src->plan;
    auto give_up = [&](hipError_t e, const char *what) -> msbwt_rle * {
        hip_fail(src, e, what);
        msbwt_rle_free(h);
        return nullptr;
    };
    DeviceScope scope(h->device);
    if (!scope.ok()) {
        fail(src, MSBWT_ERR_HIP, scope.why());
        msbwt_rle_free(h);
        return nullptr;
    }
    if (ensure_runtime(h) != MSBWT_OK) return give_up(hipErrorUnknown, "replicate: runtime setup");
    if (h->device != src->device) {  // direct GPU -> GPU copies (xGMI) when the pair allows it; staged by the runtime otherwise
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, h->device, src->device) == hipSuccess && can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(src->device, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
    }
    const PairIndexSizes psz = pair_index_sizes(src->nblocks, src->pair_stride);
    struct Piece { void *const *from; void **to; size_t bytes; };
    const Piece pieces[] = {
        {&src->d_blocks, &h->d_blocks, size_t(src->nblocks) * kBlockBytes},
        {&src->d_overflow, &h->d_overflow, src->d_overflow ? size_t(src->overflow_bytes) : 0},
        {&src->d_table, &h->d_table, src->d_table ? src->table_bytes : 0},
        {&src->d_table_side, &h->d_table_side, src->d_table_side ? size_t(src->table_side_bytes) : 0},
        {reinterpret_cast<void *const *>(&src->d_filter), reinterpret_cast<void **>(&h->d_filter), src->d_filter ? (size_t(1) << (2 * src->filter_depth)) / 8 : 0},
        {&src->d_pair_blocks, &h->d_pair_blocks, src->d_pair_blocks ? psz.pair_block_bytes : 0},
        {&src->d_pair_super, &h->d_pair_super, src->d_pair_super ? psz.super_bytes : 0},
        {&src->d_sparse, &h->d_sparse, src->d_sparse ? size_t(src->sparse_bytes) : 0},
        {&src->d_sparse_side, &h->d_sparse_side, src->d_sparse_side ? size_t(src->sparse_side_bytes) : 0},
        {&src->d_sparse2, &h->d_sparse2, src->d_sparse2 ? size_t(src->sparse2_bytes) : 0},
        {&src->d_sparse2_side, &h->d_sparse2_side, src->d_sparse2_side ? size_t(src->sparse2_side_bytes) : 0},
    };
    for (const Piece &p : pieces) {
        if (!p.bytes || !*p.from) continue;
        hipError_t e = hipMalloc(p.to, p.bytes);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(*p.to, h->device, *p.from, src->device, p.bytes, h->stream);
        if (e != hipSuccess) return give_up(e, "replicate: copy index to the other device");
    }
    const hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return give_up(e, "replicate: copy index to the other device");
    h->totals = src->totals;
    h->nblocks = src->nblocks;
    h->overflow_bytes = src->overflow_bytes;
    h->table_depth = src->table_depth;
    h->table_packed = src->table_packed;
    h->table_bytes = src->table_bytes;
    h->table_side_bytes = src->table_side_bytes;
    h->table_lines = src->table_lines;
    h->table_escape_lines = src->table_escape_lines;
    h->sparse_bytes = src->sparse_bytes;
    h->sparse_side_bytes = src->sparse_side_bytes;
    h->sparse_nbuckets = src->sparse_nbuckets;
    h->sparse_probe = src->sparse_probe;
    h->sparse_depth = src->sparse_depth;
    h->sparse_tier = src->sparse_tier;
    h->sparse_report = src->sparse_report;
    h->sparse2_bytes = src->sparse2_bytes;
    h->sparse2_side_bytes = src->sparse2_side_bytes;
    h->sparse2_entries = src->sparse2_entries;
    h->sparse2_nbuckets = src->sparse2_nbuckets;
    h->sparse2_probe = src->sparse2_probe;
    h->sparse2_depth = src->sparse2_depth;
    h->sparse2_tier = src->sparse2_tier;
    h->typical_width = src->typical_width;
    h->pair_overlap_bytes = src->pair_overlap_bytes;
    h->filter_depth = src->filter_depth;
    h->pair_bytes = src->pair_bytes;
    h->loaded = true;
    return h;
}

}  // extern "C"

namespace {

// contiguous shards starting at multiples of 16 items (16-byte aligned rows for any k; sharded.py has the same rule)
void shard_of(size_t n, size_t world, size_t rank, size_t *lo, size_t *hi) {
    const size_t units = (n + 15) / 16, base = units / world, extra = units % world;
    const size_t lo_u = rank * base + std::min(rank, extra), hi_u = lo_u + base + (rank < extra ? 1 : 0);
    *lo = std::min(n, lo_u * 16);
    *hi = std::min(n, hi_u * 16);
}

// runs work(r) for every replica on its own host thread; returns the first non-zero code
template <class Work>
int on_every_replica(size_t n_replicas, Work &&work) {
    std::vector<int> rc(n_replicas, MSBWT_OK);
    std::vector<std::thread> threads;
    for (size_t r = 1; r < n_replicas; ++r) threads.emplace_back([&, r] { rc[r] = work(r); });
    rc[0] = work(0);
    for (auto &t : threads) t.join();
    for (int c : rc)
        if (c) return c;
    return MSBWT_OK;
}

}  // namespace

extern "C" {

int msbwt_rle_count_kmers_multi(const msbwt_rle *const *replicas, size_t n_replicas, const uint8_t *kmers, size_t k, size_t n,
                                uint64_t *out_counts) {
    if (!replicas || n_replicas == 0) return MSBWT_ERR_INVALID_ARG;
    for (size_t r = 0; r < n_replicas; ++r)
        if (!replicas[r]) return MSBWT_ERR_INVALID_ARG;
    if (n && (!out_counts || (!kmers && k))) return MSBWT_ERR_INVALID_ARG;
    // one host thread and one pinned pipeline per replica; every shard's counts land directly in the
    // caller's buffer -- the "gather" is the D2H copies themselves
    return on_every_replica(n_replicas, [&](size_t r) {
        size_t lo, hi;
        shard_of(n, n_replicas, r, &lo, &hi);
        return hi > lo ? msbwt_rle_count_kmers(replicas[r], kmers + lo * k, k, hi - lo, out_counts + lo) : MSBWT_OK;
    });
}

int msbwt_rle_count_read_kmers_multi(const msbwt_rle *const *replicas, size_t n_replicas, const uint8_t *reads, size_t read_len,
                                     size_t n_reads, size_t k, int ascii, uint64_t *out_fwd, uint64_t *out_rc) {
    if (!replicas || n_replicas == 0 || k < 1 || k > read_len) return MSBWT_ERR_INVALID_ARG;
    for (size_t r = 0; r < n_replicas; ++r)
        if (!replicas[r]) return MSBWT_ERR_INVALID_ARG;
    const size_t windows = read_len - k + 1;
    return on_every_replica(n_replicas, [&](size_t r) {
        size_t lo, hi;
        shard_of(n_reads, n_replicas, r, &lo, &hi);
        if (hi <= lo) return int(MSBWT_OK);
        return msbwt_rle_count_read_kmers(replicas[r], reads + lo * read_len, read_len, hi - lo, k, ascii,
                                          out_fwd ? out_fwd + lo * windows : nullptr, out_rc ? out_rc + lo * windows : nullptr);
    });
}

int msbwt_rle_count_kmers_multi_device(const msbwt_rle *const *replicas, size_t n_replicas, const void *d_kmers, size_t k, size_t n,
                                       void *d_out_counts) {
    if (!replicas || n_replicas == 0) return MSBWT_ERR_INVALID_ARG;
    for (size_t r = 0; r < n_replicas; ++r)
        if (!replicas[r]) return MSBWT_ERR_INVALID_ARG;
    if (n && (!d_out_counts || (!d_kmers && k))) return MSBWT_ERR_INVALID_ARG;
    const int home = replicas[0]->device;
    const uint8_t *src = static_cast<const uint8_t *>(d_kmers);
    uint64_t *dst = static_cast<uint64_t *>(d_out_counts);
    // enqueue every shard on its replica's stream: shard in by peer copy, kernel, counts back by peer copy.  An
    // error ends the enqueueing but NOT the call: the replicas already at work are drained below before the
    // first error is returned, so that nothing still writes into d_out_counts when the caller gets it back.
    auto enqueue = [&](size_t r) -> int {
        msbwt_rle *h = const_cast<msbwt_rle *>(replicas[r]);
        size_t lo, hi;
        shard_of(n, n_replicas, r, &lo, &hi);
        if (hi <= lo) return MSBWT_OK;
        std::lock_guard<std::mutex> lock(h->mu);
        if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
        DeviceScope scope(h->device);
        if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
        const size_t m = hi - lo;
        // MSBWT_FORCE_PEER_COPIES=1: take the staging + peer-copy path even on the home device (tests on one GPU)
        static const bool force_peer = [] { const char *e = std::getenv("MSBWT_FORCE_PEER_COPIES"); return e && std::atoi(e) != 0; }();
        if (h->device == home && !(force_peer && r > 0)) return launch_count(h, src + lo * k, k, m, dst + lo, h->stream, kHostFlags);
        const size_t kmer_bytes = (m * k + 255) / 256 * 256;
        int rc = ensure_stage(h, kmer_bytes + m * sizeof(uint64_t));
        if (rc) return rc;
        uint8_t *d_k = static_cast<uint8_t *>(h->d_stage);
        uint64_t *d_c = reinterpret_cast<uint64_t *>(d_k + kmer_bytes);
        if (k) HIP_TRY(h, hipMemcpyPeerAsync(d_k, h->device, src + lo * k, home, m * k, h->stream));
        rc = launch_count(h, d_k, k, m, d_c, h->stream, kHostFlags);
        if (rc) return rc;
        HIP_TRY(h, hipMemcpyPeerAsync(dst + lo, home, d_c, h->device, m * sizeof(uint64_t), h->stream));
        return MSBWT_OK;
    };
    int first = MSBWT_OK;
    for (size_t r = 0; r < n_replicas && !first; ++r) first = enqueue(r);
    // the counts are complete when every replica's stream has drained
    for (size_t r = 0; r < n_replicas; ++r) {
        msbwt_rle *h = const_cast<msbwt_rle *>(replicas[r]);
        std::lock_guard<std::mutex> lock(h->mu);
        if (!h->stream) continue;
        DeviceScope scope(h->device);
        uint32_t flags = 0;
        int rc = scope.ok() ? read_flags(h, h->stream, kHostFlags, &flags) : fail(h, MSBWT_ERR_HIP, scope.why());
        if (!rc) rc = flags_to_code(h, flags);
        if (rc && !first) first = rc;
    }
    return first;
}

// ---- one process per GPU: the final count gather over RCCL ------------------------------------------------
int msbwt_comm_get_unique_id(void *out_id) {
    std::string why;
    if (!out_id) return MSBWT_ERR_INVALID_ARG;
    return comm_unique_id(out_id, &why) ? MSBWT_OK : MSBWT_ERR_RCCL;
}

int msbwt_comm_init_rank(void **out_comm, int nranks, const void *id, int rank) {
    std::string why;
    if (!out_comm || !id || nranks < 1 || rank < 0 || rank >= nranks) return MSBWT_ERR_INVALID_ARG;
    if (!comm_init_rank(out_comm, nranks, id, rank, &why)) {
        std::fprintf(stderr, "[msbwt] msbwt_comm_init_rank: %s\n", why.c_str());
        return MSBWT_ERR_RCCL;
    }
    return MSBWT_OK;
}

int msbwt_comm_destroy(void *comm) {
    std::string why;
    if (!comm) return MSBWT_ERR_INVALID_ARG;
    return comm_destroy(comm, &why) ? MSBWT_OK : MSBWT_ERR_RCCL;
}

int msbwt_rle_allgather_counts(const msbwt_rle *ch, void *comm, const void *d_mine, size_t n_mine, void *d_all, int wire_bits,
                               void *hip_stream) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!comm || (wire_bits != 64 && wire_bits != 32 && wire_bits != 16) || (n_mine && (!d_mine || !d_all)))
        return fail(h, MSBWT_ERR_INVALID_ARG, "allgather_counts needs a communicator, buffers and a wire width of 64, 32 or 16 bits");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    int rc = ensure_runtime(h);
    if (rc) return rc;
    std::string why;
    const int nranks = comm_ranks(comm, &why);
    if (nranks < 1) return fail(h, MSBWT_ERR_RCCL, why);
    const size_t need = allgather_scratch_bytes(n_mine, nranks, wire_bits);
    if (need > h->gather_bytes) {  // (hipFree waits for the device: no gather still reads the old buffer)
        if (h->d_gather) (void)hipFree(h->d_gather);
        h->d_gather = nullptr;
        h->gather_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->d_gather, need));
        h->gather_bytes = need;
    }
    const hipError_t e = allgather_counts(comm, nranks, static_cast<const uint64_t *>(d_mine), n_mine, static_cast<uint64_t *>(d_all), wire_bits,
                                          h->d_gather, h->d_flags + kDeviceFlags, static_cast<hipStream_t>(hip_stream), &why);
    if (e == hipSuccess) return MSBWT_OK;
    return why.empty() ? hip_fail(h, e, "all-gather of the counts") : fail(h, MSBWT_ERR_RCCL, why);
}

// One batch counted and gathered as a PIPELINE (a caller with a single batch otherwise sees kernel + gather + widening one after the other):
// the rank's shard is cut into pieces; piece i is searched on the caller's stream while the counts of piece i - 1 travel -- narrowed,
// ncclAllGather, placed -- on a second stream of the handle.
int msbwt_rle_count_kmers_allgather_device(const msbwt_rle *ch, void *comm, const void *d_kmers, size_t k, size_t n_mine, void *d_mine_counts, void *d_all,
                                           int wire_bits, int out_bits, int pieces, void *hip_stream) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    if (!comm || (wire_bits != 64 && wire_bits != 32 && wire_bits != 16) || (out_bits != 64 && out_bits != wire_bits) || pieces < 1 || pieces > 64 || k < 1 ||
        (n_mine && (!d_kmers || !d_mine_counts || !d_all)))
        return fail(h, MSBWT_ERR_INVALID_ARG, "count_kmers_allgather needs a communicator, buffers, a wire width of 64 / 32 / 16 bits, counts left at that width or widened to 64, 1..64 pieces");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    int rc = ensure_runtime(h);
    if (rc) return rc;
    if (n_mine == 0) return MSBWT_OK;
    std::string why;
    const int nranks = comm_ranks(comm, &why);
    if (nranks < 1) return fail(h, MSBWT_ERR_RCCL, why);
    if (!h->gather_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->gather_stream, hipStreamNonBlocking));
    while (h->piece_events.size() < size_t(pieces) + 1) {
        hipEvent_t e = nullptr;
        HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->piece_events.push_back(e);
    }
    const size_t need = allgather_pieces_scratch_bytes(n_mine, nranks, wire_bits);
    if (need > h->gather_bytes) {  // (hipFree waits for the device: no gather still reads the old buffer)
        if (h->d_gather) (void)hipFree(h->d_gather);
        h->d_gather = nullptr;
        h->gather_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->d_gather, need));
        h->gather_bytes = need;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    // pieces of whole 16-query units (rows of any k then start 16-byte aligned: the fast kernels), the last one takes what is left
    const size_t per = allgather_piece_queries(n_mine, pieces);  // (gather.hpp: at most `pieces` pieces, whatever n_mine)
    // the gather stream starts behind everything the caller has queued so far (its buffers may still be in use there)
    HIP_TRY(h, hipEventRecord(h->piece_events[size_t(pieces)], stream));
    HIP_TRY(h, hipStreamWaitEvent(h->gather_stream, h->piece_events[size_t(pieces)], 0));
    size_t piece = 0;
    for (size_t off = 0; off < n_mine; off += per, ++piece) {
        const size_t len = std::min(per, n_mine - off);
        if (piece >= size_t(pieces)) {  // (cannot happen: allgather_piece_queries cuts at most `pieces` pieces)
            rc = fail(h, MSBWT_ERR_INTERNAL, "count_kmers_allgather: more pieces than events");
            break;
        }
        rc = launch_count(h, static_cast<const uint8_t *>(d_kmers) + off * k, k, len, static_cast<uint64_t *>(d_mine_counts) + off, stream, kDeviceFlags);
        if (rc) break;
        hipError_t e = hipEventRecord(h->piece_events[piece], stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(h->gather_stream, h->piece_events[piece], 0);
        if (e != hipSuccess) {
            rc = hip_fail(h, e, "piece event");
            break;
        }
        e = allgather_piece(comm, nranks, static_cast<const uint64_t *>(d_mine_counts), n_mine, off, len, d_all, wire_bits, out_bits, h->d_gather,
                            h->d_flags + kDeviceFlags, h->gather_stream, &why);
        if (e != hipSuccess) {
            rc = why.empty() ? hip_fail(h, e, "all-gather of a piece of the counts") : fail(h, MSBWT_ERR_RCCL, why);
            break;
        }
    }
    // The caller's stream continues once the last piece has arrived -- also after an error in the middle: pieces already queued on the
    // gather stream still write d_gather and d_all, so the caller's stream must not run ahead of them (after an RCCL error the
    // communicator is unusable and other ranks may be left inside ncclAllGather: the caller tears the job down).
    const hipError_t j1 = hipEventRecord(h->piece_events[size_t(pieces)], h->gather_stream);
    const hipError_t j2 = j1 == hipSuccess ? hipStreamWaitEvent(stream, h->piece_events[size_t(pieces)], 0) : j1;
    if (j2 != hipSuccess) {
        (void)hipStreamSynchronize(h->gather_stream);
        if (!rc) rc = hip_fail(h, j2, "join of the gather stream");
    }
    return rc;
}

size_t msbwt_allgather_piece_queries(size_t n_mine, int pieces) { return allgather_piece_queries(n_mine, pieces); }

// ---- batch order keys (order.hip): sort a batch by them and it walks the index in ascending order -----------------------
int msbwt_kmer_order_keys(const uint8_t *kmers, size_t k, size_t n, uint64_t *out_keys) {
    if (k < 1 || k > 0xFFFFFFFFull || (n && (!kmers || !out_keys))) return MSBWT_ERR_INVALID_ARG;
    order_keys_host(kmers, uint32_t(k), n, out_keys);
    return MSBWT_OK;
}

int msbwt_rle_kmer_order_keys_device(const msbwt_rle *ch, const void *d_kmers, size_t k, size_t n, void *d_out_keys, void *hip_stream) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (k < 1 || k > 0xFFFFFFFFull || (n && (!d_kmers || !d_out_keys))) return fail(h, MSBWT_ERR_INVALID_ARG, "order keys need 1 <= k and buffers");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    HIP_TRY(h, launch_order_keys(static_cast<const uint8_t *>(d_kmers), uint32_t(k), n, static_cast<uint64_t *>(d_out_keys), static_cast<hipStream_t>(hip_stream)));
    return MSBWT_OK;
}

int msbwt_rle_set_table_depth(msbwt_rle *h, int depth) {
    if (!h || depth > kMaxTableDepth) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->wanted_table_depth = depth;
    if (!h->loaded) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    return rebuild_table(h);
}

int msbwt_rle_get_table_depth(const msbwt_rle *h) { return h ? h->table_depth : 0; }

int msbwt_rle_set_pair_index(msbwt_rle *h, int mode) {
    if (!h || mode < -1 || mode > 1) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->wanted_pair = mode;
    if (!h->loaded) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    const int rc = rebuild_pair_index(h);
    return rc ? rc : rebuild_table(h);  // the table's packed form exists only beside a pair index
}

int msbwt_rle_get_pair_index(const msbwt_rle *h) { return (h && h->d_pair_blocks) ? 1 : 0; }

int msbwt_rle_set_pair_stride(msbwt_rle *h, int stride) {
    if (!h || (stride != 0 && stride != 96 && stride != 128)) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->wanted_pair_stride = stride;
    if (!h->loaded) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    const int rc = rebuild_pair_index(h);
    return rc ? rc : rebuild_table(h);
}

int msbwt_rle_get_pair_stride(const msbwt_rle *h) { return (h && h->d_pair_blocks) ? h->pair_stride : 0; }

int msbwt_rle_set_presence_filter(msbwt_rle *h, int mode) {
    if (!h || mode < -1 || mode > 1) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->wanted_filter = mode == 0 ? 0 : -1;
    if (!h->loaded) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    return rebuild_table(h);  // the filter is made from the flat table, which a packed table no longer holds
}

int msbwt_rle_set_table_packed(msbwt_rle *h, int mode) {
    if (!h || mode < -1 || mode > 1) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->wanted_table_packed = mode;
    if (!h->loaded) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    return rebuild_table(h);
}

int msbwt_rle_get_table_packed(const msbwt_rle *h) { return (h && h->d_table && h->table_packed) ? 1 : 0; }

int msbwt_rle_set_memory_budget(msbwt_rle *h, uint64_t bytes) {
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->memory_budget = bytes;
    if (!h->loaded) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    // the optional structures are rebuilt under the new budget (the plan counts the memory they hold now as free).  Run blocks: their sparse
    // table was built at load time and cannot be rebuilt (the plane blocks it came from are gone) -- it stays while the index fits the budget
    if (h->block_format == kBlocksPlanes || (bytes != 0 && msbwt_rle_device_bytes(h) > bytes)) release_sparse(h);
    if (h->d_table) (void)hipFree(h->d_table);
    if (h->d_table_side) (void)hipFree(h->d_table_side);
    if (h->d_filter) (void)hipFree(h->d_filter);
    if (h->d_pair_blocks) (void)hipFree(h->d_pair_blocks);
    if (h->d_pair_super) (void)hipFree(h->d_pair_super);
    h->d_table = h->d_table_side = h->d_pair_blocks = h->d_pair_super = nullptr;
    h->d_filter = nullptr;
    h->filter_depth = 0;
    h->table_depth = 0;
    h->table_packed = false;
    h->table_bytes = 0;
    h->table_side_bytes = h->table_lines = h->table_escape_lines = 0;
    h->pair_bytes = h->pair_overlap_bytes = 0;
    make_plan(h);
    int rc = rebuild_pair_index(h);
    if (!rc) rc = rebuild_table(h);
    if (rc) return rc;
    // a budget that cannot be met is said, not silently exceeded (the call still succeeds: the index works)
    h->err.clear();
    if (bytes != 0 && h->block_format != kBlocksPlanes) h->err = "memory budget: the run-block format has no optional structures to plan; the budget is not applied";
    else if (bytes != 0 && bytes < h->nblocks * kBlockBytes) h->err = "memory budget: below the plane blocks themselves, which are built all the same";
    return MSBWT_OK;
}

uint64_t msbwt_rle_get_memory_budget(const msbwt_rle *h) { return h ? h->memory_budget : 0; }

int msbwt_auto_index_plan(uint64_t total_symbols, uint64_t free_hbm_bytes, uint64_t hbm_total_bytes, double typical_width, uint64_t budget_bytes,
                          int *pair_index, int *pair_stride, int *flat_depth, int *packed_depth, uint64_t *index_bytes) {
    if (!pair_index || !pair_stride || !flat_depth || !packed_depth) return MSBWT_ERR_INVALID_ARG;
    const uint64_t nblocks = plane_block_count(total_symbols);
    const PairIndexSizes wide = pair_index_sizes(nblocks, 96), narrow = pair_index_sizes(nblocks, 128);
    const IndexPlan p = plan_index(total_symbols, free_hbm_bytes, hbm_total_bytes, typical_width, budget_bytes, narrow.pair_block_bytes + narrow.super_bytes,
                                   wide.pair_block_bytes + wide.super_bytes);
    *pair_index = p.pair ? 1 : 0;
    *pair_stride = p.pair ? p.stride : 0;
    *flat_depth = p.flat;
    *packed_depth = p.packed;
    if (index_bytes) *index_bytes = p.bytes;
    return MSBWT_OK;
}

int msbwt_rle_set_table_side(msbwt_rle *h, int mode) {
    if (!h || mode < 0 || mode > 1) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->wanted_table_side = mode;
    if (!h->loaded) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    return rebuild_table(h);
}

int msbwt_rle_table_info(const msbwt_rle *h, uint64_t *lines, uint64_t *escape_lines, uint64_t *side_bytes) {
    if (!h) return MSBWT_ERR_INVALID_ARG;
    const bool packed = h->d_table && h->table_packed;
    if (lines) *lines = packed ? h->table_lines : 0;
    if (escape_lines) *escape_lines = packed ? h->table_escape_lines : 0;
    if (side_bytes) *side_bytes = packed ? h->table_side_bytes : 0;
    return MSBWT_OK;
}

// ---- sparse suffix table (sparse_table.hpp) ----------------------------------------------------------------------------------
int msbwt_rle_set_sparse_table(msbwt_rle *h, int depth) {
    if (!h || !(depth == -1 || depth == 0 || (depth >= kSparseMinDepth && depth <= kSparseMaxDepth))) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->wanted_sparse = depth;
    if (!h->loaded) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    if (h->block_format != kBlocksPlanes) {  // run blocks: the table is built at load time only (from plane blocks that are gone); 0 drops it now
        if (depth == 0) release_sparse(h);
        return MSBWT_OK;
    }
    return rebuild_table(h);
}

int msbwt_rle_get_sparse_table(const msbwt_rle *h) { return (h && h->d_sparse) ? h->sparse_depth : 0; }

int msbwt_rle_set_sparse_tiers(msbwt_rle *h, int mode) {
    if (!h || mode < -1 || mode > 1) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    const bool changes = mode != h->wanted_tiers;
    h->wanted_tiers = mode;
    if (!h->loaded || !changes || h->wanted_sparse == 0 || h->block_format != kBlocksPlanes) return MSBWT_OK;  // (run blocks: at the next load)
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    return rebuild_table(h);
}

int msbwt_rle_get_sparse_tiers(const msbwt_rle *h) { return (h && h->d_sparse && h->sparse_tier) ? 1 : 0; }

int msbwt_rle_set_sparse_second(msbwt_rle *h, int mode) {
    if (!h || mode < -1 || mode > 0) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    const bool changes = mode != h->wanted_second;
    h->wanted_second = mode;
    if (!h->loaded || !changes || h->wanted_sparse == 0 || h->block_format != kBlocksPlanes) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    return rebuild_table(h);
}

int msbwt_rle_set_query_length(msbwt_rle *h, int k) {
    if (!h || k < 0) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    const bool changes = sparse_auto_max_depth(k) != sparse_auto_max_depth(h->query_length);
    h->query_length = k;
    if (!h->loaded || !changes || h->wanted_sparse >= 0 || h->block_format != kBlocksPlanes) return MSBWT_OK;  // (an explicit depth, or none at all, does not follow the hint; run blocks: at the next load)
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    return rebuild_table(h);
}

int msbwt_rle_get_query_length(const msbwt_rle *h) { return h ? h->query_length : 0; }

int msbwt_auto_sparse_max_depth(int query_length) { return sparse_auto_max_depth(query_length); }

int msbwt_rle_sparse_table_info(const msbwt_rle *ch, uint64_t *out) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h || !out) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    std::memset(out, 0, MSBWT_SPARSE_INFO_WORDS * sizeof(uint64_t));
    const SparseBuildReport &r = h->sparse_report;
    if (h->d_sparse) {
        out[0] = uint64_t(h->sparse_depth);
        out[1] = r.entries;
        out[2] = h->sparse_nbuckets;
        out[3] = h->sparse_bytes;
        out[4] = r.nescapes;
        out[5] = h->sparse_side_bytes;
        out[6] = r.displaced;
        out[8] = h->sparse_tier ? 1 : 0;
        out[9] = h->sparse_probe;
        out[42] = r.filtered;
        out[43] = uint64_t(h->sparse2_depth);
        out[44] = h->sparse2_bytes + h->sparse2_side_bytes;
    }
    out[7] = uint64_t(r.parent_depth);
    for (int d = 0; d <= kSparseMaxDepth; ++d) {
        out[10 + d] = r.distinct[d];
        out[45 + d] = r.escapes[d];
        out[80 + d] = r.singles[d];
    }
    return MSBWT_OK;
}

int msbwt_sparse_hash(uint64_t key, int depth, uint64_t nbuckets, uint32_t *bucket, uint32_t *tag) {
    if (depth < kSparseMinDepth || depth > kSparseMaxDepth || nbuckets == 0 || nbuckets > 0xFFFFFFFFull || !bucket || !tag) return MSBWT_ERR_INVALID_ARG;
    const uint64_t x = sparse_mix(key, uint32_t(2 * depth));
    *bucket = sparse_bucket(x, uint32_t(2 * depth), uint32_t(nbuckets));
    *tag = sparse_tag(x, uint32_t(depth));
    return MSBWT_OK;
}

int msbwt_sparse_hash64(uint64_t key, int depth, uint64_t nbuckets, uint32_t *bucket, uint64_t *tag) {
    if (depth < kSparseMinDepth || depth > kSparseMaxDepth || nbuckets == 0 || nbuckets > 0xFFFFFFFFull || !bucket || !tag) return MSBWT_ERR_INVALID_ARG;
    const uint64_t x = sparse_mix(key, uint32_t(2 * depth));
    *bucket = sparse_bucket(x, uint32_t(2 * depth), uint32_t(nbuckets));
    *tag = (uint64_t(sparse_tag_hi(x, uint32_t(depth))) << 32) | sparse_tag(x, uint32_t(depth));
    return MSBWT_OK;
}

int msbwt_auto_sparse_depth(const uint64_t *distinct, const uint64_t *wide, int parent_depth, uint64_t avail_bytes, int query_length, int *depth, uint64_t *table_bytes) {
    if (!distinct || !wide || !depth || parent_depth < 0 || parent_depth > 16 || query_length < 0) return MSBWT_ERR_INVALID_ARG;
    const SparseChoice c = choose_sparse_depth(distinct, wide, parent_depth, sparse_auto_max_depth(query_length), avail_bytes, 0);
    *depth = c.depth;
    if (table_bytes) *table_bytes = c.bytes;
    return MSBWT_OK;
}

int msbwt_auto_sparse_choice(const uint64_t *distinct, const uint64_t *wide, const uint64_t *singles, int parent_depth, uint64_t avail_bytes, int query_length,
                             int tiers, int *depth, int *two_tier, uint64_t *table_bytes) {
    if (!distinct || !wide || !singles || !depth || !two_tier || parent_depth < 0 || parent_depth > 16 || query_length < 0 || tiers < -1 || tiers > 1)
        return MSBWT_ERR_INVALID_ARG;
    const SparseChoice c = choose_sparse_depth(distinct, wide, parent_depth, sparse_auto_max_depth(query_length), avail_bytes, 0, singles, tiers);
    *depth = c.depth;
    *two_tier = c.tier ? 1 : 0;
    if (table_bytes) *table_bytes = c.bytes;
    return MSBWT_OK;
}

int msbwt_sparse_filter_bits(uint64_t tag, uint32_t *word, uint32_t *mask) {
    if (!word || !mask) return MSBWT_ERR_INVALID_ARG;
    const uint32_t f = sparse_filter_hash(uint32_t(tag));
    *word = sparse_filter_word(f);
    *mask = sparse_filter_mask(f);
    return MSBWT_OK;
}

int msbwt_sparse_table_shape(int depth, uint64_t entries, uint64_t *nbuckets, int *probe) {
    if (depth < kSparseMinDepth || depth > kSparseMaxDepth || !nbuckets || !probe) return MSBWT_ERR_INVALID_ARG;
    *nbuckets = sparse_buckets_for(depth, entries);
    *probe = sparse_probe_limit(depth, *nbuckets);
    return MSBWT_OK;
}

size_t msbwt_rle_download_sparse_table(const msbwt_rle *ch, void *out_lines, size_t cap_bytes, void *out_side, size_t cap_side_bytes) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return SIZE_MAX;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->loaded || !h->d_sparse) return SIZE_MAX;
    DeviceScope scope(h->device);
    if (!scope.ok()) return SIZE_MAX;
    if (out_lines && cap_bytes >= h->sparse_bytes && hipMemcpy(out_lines, h->d_sparse, h->sparse_bytes, hipMemcpyDeviceToHost) != hipSuccess) return SIZE_MAX;
    if (out_side && h->d_sparse_side && cap_side_bytes >= h->sparse_side_bytes &&
        hipMemcpy(out_side, h->d_sparse_side, h->sparse_side_bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return SIZE_MAX;
    return size_t(h->sparse_bytes);
}

int msbwt_rle_set_line_streaming(msbwt_rle *h, int mode) {
    if (!h || mode < -1 || mode > 1) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->wanted_streaming = mode;
    return MSBWT_OK;
}

int msbwt_rle_get_line_streaming(const msbwt_rle *ch) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h || !h->loaded) return 0;
    std::lock_guard<std::mutex> lock(h->mu);
    return view_of(h).stream_lines ? 1 : 0;
}

int msbwt_rle_probe_line_rate(const msbwt_rle *ch, int which, double *lines_per_second) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h || !lines_per_second || which < 0 || which > 3) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->loaded) return fail(h, MSBWT_ERR_NOT_LOADED, "no BWT loaded");
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    const void *p = which == 0 ? h->d_blocks : which == 1 ? h->d_pair_blocks : which == 2 ? h->d_sparse : h->d_table;
    const uint64_t bytes = which == 0 ? h->nblocks * kBlockBytes : which == 1 ? pair_index_sizes(h->nblocks, h->pair_stride).pair_block_bytes
                           : which == 2 ? h->sparse_bytes : uint64_t(h->table_bytes);
    *lines_per_second = 0.0;
    if (!p || bytes < 128) return MSBWT_OK;
    *lines_per_second = line_rate_of(p, bytes, h->stream);
    return MSBWT_OK;
}

int msbwt_rle_set_search_counters(msbwt_rle *h, int enabled) {
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->counting = enabled != 0;
    return MSBWT_OK;
}

int msbwt_rle_search_counters(const msbwt_rle *ch, uint64_t *out, void *hip_stream) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h || !out) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    std::memset(out, 0, MSBWT_SEARCH_COUNTERS * sizeof(uint64_t));
    if (!h->d_flags) return MSBWT_OK;
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    char *d_counters = reinterpret_cast<char *>(h->d_flags) + kCountersOffset;
    HIP_TRY(h, hipMemcpyAsync(out, d_counters, MSBWT_SEARCH_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(h, hipMemsetAsync(d_counters, 0, MSBWT_SEARCH_COUNTERS * sizeof(uint64_t), stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
    return MSBWT_OK;
}

int msbwt_rle_get_presence_filter(const msbwt_rle *h) { return (h && h->d_filter) ? h->filter_depth : 0; }

int msbwt_rle_set_block_format(msbwt_rle *h, int format) {
    if (!h || (format != kBlocksPlanes && format != kBlocksRuns)) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->wanted_block_format = format;  // the next load builds it
    return MSBWT_OK;
}

int msbwt_rle_get_block_format(const msbwt_rle *h) { return h ? (h->loaded ? h->block_format : h->wanted_block_format) : 0; }

int msbwt_auto_table_depths(uint64_t total_symbols, uint64_t free_hbm_bytes, int pair_index, int *flat_depth, int *packed_depth) {
    if (!flat_depth || !packed_depth) return MSBWT_ERR_INVALID_ARG;
    const TableChoice c = choose_table_depths(total_symbols, plane_block_count(total_symbols) * kBlockBytes, free_hbm_bytes, pair_index != 0, true);
    *flat_depth = c.flat;
    *packed_depth = c.packed;
    return MSBWT_OK;
}

int msbwt_run_build_fits_device(uint64_t total_symbols, uint64_t free_hbm_bytes) { return run_build_fits_device(total_symbols, free_hbm_bytes) ? 1 : 0; }

int msbwt_auto_pair_stride(uint64_t total_symbols, uint64_t free_hbm_bytes, uint64_t hbm_total_bytes, double typical_width, int *stride) {
    if (!stride) return MSBWT_ERR_INVALID_ARG;
    const uint64_t nblocks = plane_block_count(total_symbols);
    const PairIndexSizes wide = pair_index_sizes(nblocks, 96), narrow = pair_index_sizes(nblocks, 128);
    const uint64_t bytes96 = wide.pair_block_bytes + wide.super_bytes + wide.scratch_bytes, bytes128 = narrow.pair_block_bytes + narrow.super_bytes;
    const uint64_t after128 = free_hbm_bytes > bytes128 ? free_hbm_bytes - bytes128 : 0;
    *stride = choose_pair_stride(bytes96, expected_table_bytes(total_symbols, nblocks * kBlockBytes, after128, true, true), free_hbm_bytes,
                                 hbm_total_bytes, typical_width);
    return MSBWT_OK;
}

double msbwt_rle_get_typical_range_width(const msbwt_rle *h) { return h ? h->typical_width : -1.0; }

int msbwt_rle_set_search_kernel(msbwt_rle *h, int mode) {
    if (!h || mode < kSearchAuto || mode > kSearchLanes) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->search_kernel = mode;
    return MSBWT_OK;
}

int msbwt_rle_get_search_kernel(const msbwt_rle *h) { return h ? h->search_kernel : 0; }

int msbwt_rle_search_kernel_for(const msbwt_rle *h, size_t k) {
    if (!h || !h->loaded) return MSBWT_ERR_INVALID_ARG;
    if (k > 0xFFFFFFFFull) return 0;
    msbwt_rle *m = const_cast<msbwt_rle *>(h);
    std::lock_guard<std::mutex> lock(m->mu);
    return search_kernel_for(view_of(m), uint32_t(k));
}

uint64_t msbwt_rle_device_bytes(const msbwt_rle *h) {
    if (!h || !h->loaded) return 0;
    return h->nblocks * kBlockBytes + h->overflow_bytes + (h->d_table ? uint64_t(h->table_bytes) + h->table_side_bytes : 0) + h->pair_bytes +
           (h->d_filter ? (uint64_t(1) << (2 * h->filter_depth)) / 8 : 0) + h->sparse_bytes + h->sparse_side_bytes + h->sparse2_bytes + h->sparse2_side_bytes;
}

int msbwt_rle_set_kernel_timing(msbwt_rle *h, int enabled) {
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->timing = enabled != 0;
    return MSBWT_OK;
}

int msbwt_rle_kernel_time_ms(const msbwt_rle *ch, double *avg_ms, uint64_t *launches) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return MSBWT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceScope scope(h->device);
    if (!scope.ok()) return fail(h, MSBWT_ERR_HIP, scope.why());
    const int rc = drain_timing_events(h);
    if (rc) return rc;
    if (avg_ms) *avg_ms = h->timed_launches ? h->timed_ms / double(h->timed_launches) : 0.0;
    if (launches) *launches = h->timed_launches;
    h->timed_ms = 0.0;
    h->timed_launches = 0;
    return MSBWT_OK;
}

int msbwt_rle_device_ordinal(const msbwt_rle *h) { return h ? h->device : -1; }

const char *msbwt_rle_last_error(const msbwt_rle *ch) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h) return "null handle";
    // query threads may be failing into h->err right now: copy it under the handle lock into a
    // buffer owned by the calling thread (valid until this thread's next call)
    thread_local std::string mine;
    std::lock_guard<std::mutex> lock(h->mu);
    mine = h->err;
    return mine.c_str();
}

size_t msbwt_rle_download_blocks(const msbwt_rle *ch, void *out_blocks, size_t cap_blocks) {
    msbwt_rle *h = const_cast<msbwt_rle *>(ch);
    if (!h || !h->loaded || h->block_format != kBlocksPlanes) return SIZE_MAX;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceScope scope(h->device);
    if (!scope.ok()) return SIZE_MAX;
    if (out_blocks && cap_blocks >= h->nblocks &&
        hipMemcpy(out_blocks, h->d_blocks, size_t(h->nblocks) * kBlockBytes, hipMemcpyDeviceToHost) != hipSuccess)
        return SIZE_MAX;
    return size_t(h->nblocks);
}

size_t msbwt_build_plane_blocks(const uint8_t *rle_bytes, size_t len, void *out_blocks, size_t cap_blocks,
                                uint64_t *out_total) {
    Totals t;
    if ((!rle_bytes && len) || !compute_totals(rle_bytes, len, &t) || t.total > kMaxTotal) return SIZE_MAX;
    if (out_total) *out_total = t.total;
    const uint64_t nblocks = plane_block_count(t.total);
    if (out_blocks && cap_blocks >= nblocks) build_plane_blocks(rle_bytes, len, t, static_cast<uint32_t *>(out_blocks), 0);
    return size_t(nblocks);
}

size_t msbwt_build_run_blocks(const uint8_t *rle_bytes, size_t len, void *out_blocks, size_t cap_blocks, void *out_overflow,
                              size_t cap_overflow, uint64_t *out_total, uint64_t *out_noverflow) {
    Totals t;
    if ((!rle_bytes && len) || !compute_totals(rle_bytes, len, &t) || t.total > kMaxTotal) return SIZE_MAX;
    RunIndex ri;
    build_run_blocks(rle_bytes, len, t, &ri, 0);
    if (out_total) *out_total = t.total;
    if (out_noverflow) *out_noverflow = ri.noverflow;
    if (out_blocks && cap_blocks >= ri.nblocks) std::memcpy(out_blocks, ri.blocks.data(), ri.blocks.size() * sizeof(uint32_t));
    if (out_overflow && cap_overflow >= ri.noverflow && ri.noverflow) std::memcpy(out_overflow, ri.overflow.data(), ri.overflow.size() * sizeof(uint32_t));
    return size_t(ri.nblocks);
}

size_t msbwt_convert_to_vec(const uint8_t *ascii, size_t n, uint8_t *out, size_t cap) {
    std::vector<uint8_t> enc;
    if (!encode_text(ascii, n, &enc)) return SIZE_MAX;
    if (out) std::memcpy(out, enc.data(), std::min(cap, enc.size()));
    return enc.size();
}

static int npy_code(NpyStatus s) {
    switch (s) {
        case NpyStatus::kOk: return MSBWT_OK;
        case NpyStatus::kIo: return MSBWT_ERR_IO;
        case NpyStatus::kUnexpectedEof: return MSBWT_ERR_UNEXPECTED_EOF;
        default: return MSBWT_ERR_BAD_HEADER;
    }
}

int msbwt_save_bwt_numpy(const uint8_t *rle_bytes, size_t n, const char *utf8_path) {
    if (!utf8_path || (!rle_bytes && n)) return MSBWT_ERR_INVALID_ARG;
    std::string msg;
    return npy_code(write_npy_payload(utf8_path, rle_bytes, n, &msg));
}

int msbwt_save_bwt_runs_numpy(const uint8_t *syms, const uint64_t *counts, size_t nruns, const char *utf8_path) {
    if (!utf8_path || (nruns && (!syms || !counts))) return MSBWT_ERR_INVALID_ARG;
    std::vector<uint8_t> enc;
    encode_runs(syms, counts, nruns, &enc);
    std::string msg;
    return npy_code(write_npy_payload(utf8_path, enc.data(), enc.size(), &msg));
}

void msbwt_convert_stoi(const uint8_t *ascii, size_t n, uint8_t *out_codes) { ascii_to_codes(ascii, n, out_codes); }
void msbwt_convert_itos(const uint8_t *codes, size_t n, uint8_t *out_ascii) { codes_to_ascii(codes, n, out_ascii); }
void msbwt_reverse_complement_i(const uint8_t *codes, size_t n, uint8_t *out_codes) {
    reverse_complement_codes(codes, n, out_codes);
}

}  // extern "C"
