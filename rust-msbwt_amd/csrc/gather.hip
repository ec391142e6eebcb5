// The exchange step of the one-process-per-GPU form: count_kmer calls are independent and
// read-only (`&self`, src/msbwt_core.rs:125), the batch is sharded over the ranks, and ONE collective -- an
// all-gather of the counts over RCCL / xGMI -- leaves every rank with all of them.  8 bytes per count on the
// wire can cost more than computing it, so the counts may travel as 16- or 32-bit integers: narrowed here,
// widened on arrival, an overflow raising a status flag instead of truncating silently.
//
// RCCL is bound at run time (dlopen): a single-GPU host never needs it, and a host that already has an RCCL
// in its process (PyTorch bundles one) shares that instance instead of mapping a second.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: nothing here links against librccl

#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <string>

#include "gather.hpp"
#include "kernels.hpp"

namespace msbwt {
namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string why;
};

const Rccl &rccl() {
    static const Rccl bound = [] {
        Rccl r;
        // an instance that is already mapped wins (one RCCL per process), then the system one
        // MSBWT_RCCL_LIB names THE library to use: an explicit wish is taken literally (no search beside it)
        const char *env = std::getenv("MSBWT_RCCL_LIB");
        const bool named = env && *env;
        if (named) r.lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        for (const char *name : {"librccl.so", "librccl.so.1"})
            if (!r.lib && !named) r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if (!r.lib && !named) r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) {
            const char *msg = dlerror();  // once: the call clears the error it reports
            r.why = std::string("librccl.so not found: ") + (msg ? msg : "");
            return r;
        }
        r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.comm_count = reinterpret_cast<decltype(r.comm_count)>(dlsym(r.lib, "ncclCommCount"));
        r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.lib, "ncclAllGather"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.comm_count || !r.all_gather || !r.error_string) {
            r.why = "librccl.so lacks an expected symbol";
            r.lib = nullptr;
        }
        return r;
    }();
    return bound;
}

std::string describe(ncclResult_t e) {
    const Rccl &r = rccl();
    return std::string("RCCL: ") + (r.error_string ? r.error_string(e) : "error");
}

// out[i] = in[i] as a narrower unsigned integer; a value that does not fit sets kFlagNarrowOverflow
template <class Narrow>
__global__ __launch_bounds__(256) void k_narrow_counts(const uint64_t *__restrict__ in, Narrow *__restrict__ out, uint64_t n,
                                                       uint32_t *__restrict__ flags) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    bool lost = false;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t v = in[i];
        out[i] = Narrow(v);
        lost |= uint64_t(Narrow(v)) != v;
    }
    if (__ballot(lost) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(flags, kFlagNarrowOverflow);
}

template <class Narrow>
__global__ __launch_bounds__(256) void k_widen_counts(const Narrow *__restrict__ in, uint64_t *__restrict__ out, uint64_t n) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = uint64_t(in[i]);
}

// A gathered PIECE goes to its place: in[r * len + j] (rank r's counts of the piece) -> out[r * n_mine + off + j], widened to Out
template <class In, class Out>
__global__ __launch_bounds__(256) void k_place_piece(const In *__restrict__ in, Out *__restrict__ out, uint64_t len, uint64_t n_mine, uint64_t off, uint32_t nranks) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x, n = len * nranks;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t r = i / len, j = i - r * len;
        out[r * n_mine + off + j] = Out(in[i]);
    }
}

uint32_t copy_grid(uint64_t n) { return uint32_t(std::min<uint64_t>(256 * 8, std::max<uint64_t>(1, (n + 255) / 256))); }

}  // namespace

hipError_t launch_narrow_counts32(const uint64_t *d_in, uint32_t *d_out, uint64_t n, uint32_t *flags, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((k_narrow_counts<uint32_t>), dim3(copy_grid(n)), dim3(256), 0, stream, d_in, d_out, n, flags);
    return hipGetLastError();
}

bool rccl_available(std::string *why) {
    const Rccl &r = rccl();
    if (why) *why = r.why;
    return r.lib != nullptr;
}

bool comm_unique_id(void *out_id, std::string *why) {
    const Rccl &r = rccl();
    if (!r.lib) { *why = r.why; return false; }
    static_assert(sizeof(ncclUniqueId) == kCommIdBytes, "MSBWT_COMM_ID_BYTES");
    const ncclResult_t e = r.get_unique_id(static_cast<ncclUniqueId *>(out_id));
    if (e != ncclSuccess) { *why = describe(e); return false; }
    return true;
}

bool comm_init_rank(void **out_comm, int nranks, const void *id, int rank, std::string *why) {
    const Rccl &r = rccl();
    if (!r.lib) { *why = r.why; return false; }
    ncclUniqueId uid;
    __builtin_memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    const ncclResult_t e = r.comm_init_rank(&comm, nranks, uid, rank);
    if (e != ncclSuccess) { *why = describe(e); return false; }
    *out_comm = comm;
    return true;
}

bool comm_destroy(void *comm, std::string *why) {
    const Rccl &r = rccl();
    if (!r.lib) { *why = r.why; return false; }
    const ncclResult_t e = r.comm_destroy(static_cast<ncclComm_t>(comm));
    if (e != ncclSuccess) { *why = describe(e); return false; }
    return true;
}

int comm_ranks(void *comm, std::string *why) {
    const Rccl &r = rccl();
    if (!r.lib) { *why = r.why; return -1; }
    int n = 0;
    const ncclResult_t e = r.comm_count(static_cast<ncclComm_t>(comm), &n);
    if (e != ncclSuccess) { *why = describe(e); return -1; }
    return n;
}

hipError_t allgather_counts(void *comm, int nranks, const uint64_t *d_mine, size_t n_mine, uint64_t *d_all, int wire_bits, void *d_scratch,
                            uint32_t *flags, hipStream_t stream, std::string *why) {
    const Rccl &r = rccl();
    if (!r.lib) { *why = r.why; return hipErrorNotSupported; }
    if (n_mine == 0) return hipSuccess;
    const size_t wire_bytes = size_t(wire_bits) / 8;
    const void *send = d_mine;
    void *recv = d_all;
    if (wire_bits != 64) {  // scratch: [n_mine narrow | n_mine x nranks narrow]
        uint8_t *s = static_cast<uint8_t *>(d_scratch);
        send = s;
        recv = s + (n_mine * wire_bytes + 255) / 256 * 256;
        if (wire_bits == 16) hipLaunchKernelGGL(k_narrow_counts<uint16_t>, dim3(copy_grid(n_mine)), dim3(256), 0, stream, d_mine, reinterpret_cast<uint16_t *>(s), n_mine, flags);
        else hipLaunchKernelGGL(k_narrow_counts<uint32_t>, dim3(copy_grid(n_mine)), dim3(256), 0, stream, d_mine, reinterpret_cast<uint32_t *>(s), n_mine, flags);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // raw bytes on the wire: RCCL has no 16-bit integer type, and the reduction-free all-gather does not care
    const ncclResult_t e = r.all_gather(send, recv, n_mine * wire_bytes, ncclUint8, static_cast<ncclComm_t>(comm), stream);
    if (e != ncclSuccess) { *why = describe(e); return hipErrorUnknown; }
    if (wire_bits != 64) {
        const uint64_t n_all = uint64_t(n_mine) * uint64_t(nranks);
        if (wire_bits == 16) hipLaunchKernelGGL(k_widen_counts<uint16_t>, dim3(copy_grid(n_all)), dim3(256), 0, stream, static_cast<const uint16_t *>(recv), d_all, n_all);
        else hipLaunchKernelGGL(k_widen_counts<uint32_t>, dim3(copy_grid(n_all)), dim3(256), 0, stream, static_cast<const uint32_t *>(recv), d_all, n_all);
        return hipGetLastError();
    }
    return hipSuccess;
}

// One piece [off, off + len) of this rank's n_mine counts: narrowed to the wire width, all-gathered, and every rank's part of the piece
// put at its place in d_all (out_bits wide: 64, or the wire width -- "narrow at destination", no widening pass).  d_scratch as sized by
// allgather_pieces_scratch_bytes: [n_mine narrow | n_mine x nranks narrow]; pieces use disjoint parts of it, so several may be in flight.
hipError_t allgather_piece(void *comm, int nranks, const uint64_t *d_mine, size_t n_mine, size_t off, size_t len, void *d_all, int wire_bits, int out_bits,
                           void *d_scratch, uint32_t *flags, hipStream_t stream, std::string *why) {
    const Rccl &r = rccl();
    if (!r.lib) { *why = r.why; return hipErrorNotSupported; }
    if (len == 0) return hipSuccess;
    const size_t wb = size_t(wire_bits) / 8;
    uint8_t *s = static_cast<uint8_t *>(d_scratch);
    uint8_t *send = s + off * wb, *recv = s + (n_mine * wb + 255) / 256 * 256 + off * wb * size_t(nranks);
    const dim3 g(copy_grid(len)), b(256);
    if (wire_bits == 16) hipLaunchKernelGGL(k_narrow_counts<uint16_t>, g, b, 0, stream, d_mine + off, reinterpret_cast<uint16_t *>(send), len, flags);
    else if (wire_bits == 32) hipLaunchKernelGGL(k_narrow_counts<uint32_t>, g, b, 0, stream, d_mine + off, reinterpret_cast<uint32_t *>(send), len, flags);
    else hipLaunchKernelGGL(k_narrow_counts<uint64_t>, g, b, 0, stream, d_mine + off, reinterpret_cast<uint64_t *>(send), len, flags);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return he;
    const ncclResult_t e = r.all_gather(send, recv, len * wb, ncclUint8, static_cast<ncclComm_t>(comm), stream);
    if (e != ncclSuccess) { *why = describe(e); return hipErrorUnknown; }
    const dim3 ga(copy_grid(len * size_t(nranks)));
    const uint32_t nr = uint32_t(nranks);
    if (wire_bits == 16 && out_bits == 64) hipLaunchKernelGGL((k_place_piece<uint16_t, uint64_t>), ga, b, 0, stream, reinterpret_cast<const uint16_t *>(recv), static_cast<uint64_t *>(d_all), len, n_mine, off, nr);
    else if (wire_bits == 16) hipLaunchKernelGGL((k_place_piece<uint16_t, uint16_t>), ga, b, 0, stream, reinterpret_cast<const uint16_t *>(recv), static_cast<uint16_t *>(d_all), len, n_mine, off, nr);
    else if (wire_bits == 32 && out_bits == 64) hipLaunchKernelGGL((k_place_piece<uint32_t, uint64_t>), ga, b, 0, stream, reinterpret_cast<const uint32_t *>(recv), static_cast<uint64_t *>(d_all), len, n_mine, off, nr);
    else if (wire_bits == 32) hipLaunchKernelGGL((k_place_piece<uint32_t, uint32_t>), ga, b, 0, stream, reinterpret_cast<const uint32_t *>(recv), static_cast<uint32_t *>(d_all), len, n_mine, off, nr);
    else hipLaunchKernelGGL((k_place_piece<uint64_t, uint64_t>), ga, b, 0, stream, reinterpret_cast<const uint64_t *>(recv), static_cast<uint64_t *>(d_all), len, n_mine, off, nr);
    return hipGetLastError();
}

size_t allgather_pieces_scratch_bytes(size_t n_mine, int nranks, int wire_bits) {
    const size_t wb = size_t(wire_bits) / 8;
    return (n_mine * wb + 255) / 256 * 256 + n_mine * wb * size_t(nranks);
}

size_t allgather_scratch_bytes(size_t n_mine, int nranks, int wire_bits) {
    if (wire_bits == 64) return 0;
    const size_t wire_bytes = size_t(wire_bits) / 8;
    return (n_mine * wire_bytes + 255) / 256 * 256 + n_mine * wire_bytes * size_t(nranks);
}

}  // namespace msbwt
