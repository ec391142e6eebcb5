// The RCCL all-gather of the counts (gather.hip): the one exchange step of the one-process-per-GPU form.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace msbwt {

constexpr size_t kCommIdBytes = 128;  // sizeof(ncclUniqueId)

bool rccl_available(std::string *why);
bool comm_unique_id(void *out_id, std::string *why);
bool comm_init_rank(void **out_comm, int nranks, const void *id, int rank, std::string *why);
bool comm_destroy(void *comm, std::string *why);
int comm_ranks(void *comm, std::string *why);  // -1 on error

// d_all[r * n_mine + i] = rank r's d_mine[i], on every rank; wire_bits 64, 32 or 16 (narrower widths need
// allgather_scratch_bytes of device scratch; a count that does not fit sets kFlagNarrowOverflow in *flags).
// Asynchronous on `stream`.
hipError_t allgather_counts(void *comm, int nranks, const uint64_t *d_mine, size_t n_mine, uint64_t *d_all, int wire_bits, void *d_scratch,
                            uint32_t *flags, hipStream_t stream, std::string *why);
size_t allgather_scratch_bytes(size_t n_mine, int nranks, int wire_bits);
// The same for ONE piece [off, off + len) of the rank's counts (msbwt_rle_count_kmers_allgather_device: a batch counted and gathered as a
// pipeline): narrowed, all-gathered, every rank's part put at d_all[r * n_mine + off + j] as out_bits-wide integers (64, or the wire width).
hipError_t allgather_piece(void *comm, int nranks, const uint64_t *d_mine, size_t n_mine, size_t off, size_t len, void *d_all, int wire_bits, int out_bits,
                           void *d_scratch, uint32_t *flags, hipStream_t stream, std::string *why);
size_t allgather_pieces_scratch_bytes(size_t n_mine, int nranks, int wire_bits);
// Queries per piece when a shard of n_mine queries is cut into at most `pieces` pieces of whole 16-query units (rows of any k then start
// 16-byte aligned: the fast kernels); the last piece takes what is left.  ceil, not floor: n_mine = 1087, pieces = 64 -> 32 per piece,
// 34 pieces (floor gave 16 per piece and 68 pieces for 65 events).  Pinned by a CPU test through msbwt_allgather_piece_queries.
inline size_t allgather_piece_queries(size_t n_mine, int pieces) {
    const size_t unit = 16, p = pieces < 1 ? 1 : size_t(pieces);
    const size_t per = ((n_mine + p - 1) / p + unit - 1) / unit * unit;
    return per < unit ? unit : per;
}
// d_out[i] = d_in[i] as u32; a count that does not fit sets kFlagNarrowOverflow in *flags (the 32-bit count outputs of the
// packed host entry point)
hipError_t launch_narrow_counts32(const uint64_t *d_in, uint32_t *d_out, uint64_t n, uint32_t *flags, hipStream_t stream);

}  // namespace msbwt
