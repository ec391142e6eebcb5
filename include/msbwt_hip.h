/*
 * msbwt_hip.h -- C ABI of the MI355X (gfx950) implementation of rust-msbwt's RleBWT
 * count_kmer / constrain_range path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Each
 * entry point names the reference interface it replaces (file:line relative to the
 * reference repo root).  The Rust side a maintainer would add -- `impl BWT for GpuRleBWT`
 * over these symbols -- is shown in INTEGRATION.md.
 *
 * Every query entry point runs on the GPU.  There is no CPU fallback: when no HIP device
 * is usable the calls return MSBWT_ERR_HIP.
 *
 * Threading: query calls on a loaded handle may be made from several host threads (they
 * serialise on the handle's stream); load calls need exclusive access, like `&mut self`
 * in the reference.
 */
#ifndef MSBWT_HIP_H
#define MSBWT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/msbwt_core.rs:3-14 */
#define MSBWT_VC_LEN 6      /* $ A C G N T */
#define MSBWT_LETTER_BITS 3
#define MSBWT_NUMBER_BITS 5
#define MSBWT_NUM_POWER 32
#define MSBWT_MASK 0x07
#define MSBWT_COUNT_MASK 0x1F

/* Return codes.  The reference reports file problems as io::Error and panics on malformed
 * headers / symbols >= 6; nothing unwinds across this ABI, so both become codes that the
 * Rust shim maps back. */
#define MSBWT_OK 0
#define MSBWT_ERR_IO (-1)              /* io::Error from open/metadata/read            */
#define MSBWT_ERR_UNEXPECTED_EOF (-2)  /* io::ErrorKind::UnexpectedEof (size mismatch) */
#define MSBWT_ERR_BAD_HEADER (-3)      /* reference panics (rle_bwt.rs:91-93,115,123-125) */
#define MSBWT_ERR_INVALID_SYMBOL (-4)  /* reference panics (msbwt_core.rs:127, rle_bwt.rs:212) */
#define MSBWT_ERR_INVALID_RANGE (-5)   /* l > h or h > total (reference: OOB panic / underflow) */
#define MSBWT_ERR_HIP (-6)             /* HIP runtime error or no usable device         */
#define MSBWT_ERR_NOT_LOADED (-7)
#define MSBWT_ERR_TOO_LARGE (-8)       /* total symbols >= 2^40 (device block counts are 40-bit) */
#define MSBWT_ERR_INVALID_ARG (-9)
#define MSBWT_ERR_INTERNAL (-10)       /* a device-side consistency check failed (a bug, not bad input) */
#define MSBWT_ERR_OVERFLOW (-11)       /* a count did not fit the wire width of msbwt_rle_allgather_counts */
#define MSBWT_ERR_RCCL (-12)           /* RCCL missing (librccl.so is bound at run time) or an RCCL call failed */

typedef struct msbwt_rle msbwt_rle; /* opaque; replaces `struct RleBWT` (src/rle_bwt.rs:14-24) */

/* RleBWT::with_bin_power (src/rle_bwt.rs:309-322); RleBWT::new() == bin_power 8 (:297).
 * bin_power is accepted for interface compatibility; results never depend on it (the
 * device index always uses its own fixed block size).  `device` is the HIP ordinal, -1 =
 * the current device.  Returns NULL when out of memory. */
msbwt_rle *msbwt_rle_new(uint8_t bin_power);
msbwt_rle *msbwt_rle_new_on_device(uint8_t bin_power, int device);
void msbwt_rle_free(msbwt_rle *bwt);

/* BWT::load_vector (src/msbwt_core.rs:43, src/rle_bwt.rs:59-66).  Copies: the caller keeps
 * (or drops) its Vec<u8>.  The bytes are uploaded and expanded into the query layout on the
 * device (what calculate_totals + construct_fmindex, src/rle_bwt.rs:352-467, do on the CPU);
 * MSBWT_BUILD=host in the environment selects the host-side builder instead. */
int msbwt_rle_load_vector(msbwt_rle *bwt, const uint8_t *rle_bytes, size_t len);
/* BWT::load_numpy_file (src/msbwt_core.rs:58, src/rle_bwt.rs:81-155). */
int msbwt_rle_load_numpy_file(msbwt_rle *bwt, const char *utf8_path);
/* BWT::get_symbol_count (src/msbwt_core.rs:75, src/rle_bwt.rs:172-175); 0 if symbol >= 6 */
uint64_t msbwt_rle_get_symbol_count(const msbwt_rle *bwt, uint8_t symbol);
/* BWT::get_total_size (src/msbwt_core.rs:90, src/rle_bwt.rs:190-193) */
uint64_t msbwt_rle_get_total_size(const msbwt_rle *bwt);
/* BWT::constrain_range (src/msbwt_core.rs:99, src/rle_bwt.rs:202-287).  BWTRange is not
 * repr(C), so l/h travel as scalars. */
int msbwt_rle_constrain_range(const msbwt_rle *bwt, uint8_t sym, uint64_t l, uint64_t h,
                              uint64_t *out_l, uint64_t *out_h);
/* BWT::count_kmer (src/msbwt_core.rs:124-161).  kmer: k symbol codes 0..5. */
int msbwt_rle_count_kmer(const msbwt_rle *bwt, const uint8_t *kmer, size_t k, uint64_t *out_count);

/* ---- batch extensions: the GPU entry points proper (the reference has no batch API; each
 * element is exactly one count_kmer / constrain_range call) ---- */
/* kmers: n x k row-major, one byte per symbol (codes 0..5); out_counts: n.  Host pointers. */
int msbwt_rle_count_kmers(const msbwt_rle *bwt, const uint8_t *kmers, size_t k, size_t n,
                          uint64_t *out_counts);
int msbwt_rle_constrain_ranges(const msbwt_rle *bwt, const uint8_t *syms, const uint64_t *l,
                               const uint64_t *h, size_t n, uint64_t *out_l, uint64_t *out_h);
/* Same, with DEVICE pointers on the handle's device; asynchronous on `hip_stream` (a
 * hipStream_t, NULL = default stream).  Invalid input is reported by
 * msbwt_rle_device_status(), which synchronises the stream (the device entry points keep their
 * own status word: a host-pointer call on another thread never sees or clears it).
 * Alignment: d_kmers should be 16-byte aligned -- then 1 <= k <= 64 runs the fast kernels; an
 * unaligned batch is still counted correctly, by the slower generic kernel.
 * Streams: launches queued on ONE stream share a block of tile-ticket counters (the stream orders them); launches on different
 * streams -- and on hipStreamPerThread, whose one handle value stands for a different queue in every host thread -- get blocks of
 * their own, handed back when the launch that used them has completed. */
int msbwt_rle_count_kmers_device(const msbwt_rle *bwt, const void *d_kmers, size_t k, size_t n,
                                 void *d_out_counts, void *hip_stream);
int msbwt_rle_constrain_ranges_device(const msbwt_rle *bwt, const void *d_syms, const void *d_l,
                                      const void *d_h, size_t n, void *d_out_l, void *d_out_h,
                                      void *hip_stream);
int msbwt_rle_device_status(const msbwt_rle *bwt, void *hip_stream);

/* ---- compact queries: two bits per symbol (no reference counterpart; the trait's count_kmer takes one byte per symbol,
 * src/msbwt_core.rs:125, codes as string_util.rs:15-67 makes them) ----
 * A k-mer over ACGT, 1 <= k <= 64, as ceil(k / 32) u64 words: the k-mer read as a base-4 number with A C G T -> 0 1 2 3 and
 * its FIRST symbol most significant -- the usual 2-bit k-mer of k-mer counters: the last symbol sits in bits 0-1 of word 0,
 * symbol k-33 (k > 32) in bits 0-1 of word 1; bits beyond 2k are ignored.  8 bytes per 31-mer instead of 31: the host entry
 * point moves 8 + 8 (or 8 + 4) bytes per query over PCIe instead of 31 + 8, and the search kernel reads the words as they are.
 * '$' and 'N' cannot be said: such queries go through msbwt_rle_count_kmers.  Counts are those of the byte form, always.
 * msbwt_kmers_pack_2bit: n x k symbol codes -> words (MSBWT_ERR_INVALID_SYMBOL for a code outside A C G T).
 * count_bits: 64 (uint64_t out_counts[n]) or 32 (uint32_t out_counts[n]; a count that does not fit makes the call return
 * MSBWT_ERR_OVERFLOW).  Device form: d_kmers2bit 8-byte aligned, u64 counts, asynchronous like msbwt_rle_count_kmers_device. */
int msbwt_kmers_pack_2bit(const uint8_t *kmers, size_t k, size_t n, uint64_t *out_words);
int msbwt_rle_count_kmers_packed(const msbwt_rle *bwt, const uint64_t *kmers2bit, size_t k, size_t n, void *out_counts, int count_bits);
int msbwt_rle_count_kmers_packed_device(const msbwt_rle *bwt, const void *d_kmers2bit, size_t k, size_t n, void *d_out_counts, void *hip_stream);

/* Fused query preparation + counting: every k-mer window of every read, on the forward strand
 * and/or reverse-complemented, without materialising the n x k query matrix.  Replaces the
 * host-side loop  convert_stoi (src/string_util.rs:63-67) -> windows -> reverse_complement_i
 * (:45-50) -> count_kmer (src/msbwt_core.rs:124-161)  that consumers of the crate write.
 * reads: n_reads x read_len bytes, ASCII (ascii != 0: A/a C/c G/g T/t $ as in
 * string_util.rs:15-32, everything else N) or symbol codes (ascii == 0).  1 <= k <= 64,
 * k <= read_len.  out_fwd / out_rc: n_reads x (read_len - k + 1) counts; either may be NULL.
 * out_rc[r][w] is the count of reverse_complement_i(window w of read r). */
int msbwt_rle_count_read_kmers(const msbwt_rle *bwt, const uint8_t *reads, size_t read_len, size_t n_reads,
                               size_t k, int ascii, uint64_t *out_fwd, uint64_t *out_rc);
int msbwt_rle_count_read_kmers_device(const msbwt_rle *bwt, const void *d_reads, size_t read_len, size_t n_reads,
                                      size_t k, int ascii, void *d_out_fwd, void *d_out_rc, void *hip_stream);
/* The same for reads of different lengths (host pointers): read r is
 * reads[read_offsets[r] .. read_offsets[r+1]).  Counts are written window-major in read order:
 * read r owns out[W_r .. W_r + max(0, len_r - k + 1)) with W_r the running sum over earlier
 * reads; *out_windows (optional) receives the total.  out arrays must hold that many u64 --
 * call with out_fwd == out_rc == NULL to obtain the total only. */
int msbwt_rle_count_ragged_read_kmers(const msbwt_rle *bwt, const uint8_t *reads, const uint64_t *read_offsets,
                                      size_t n_reads, size_t k, int ascii, uint64_t *out_fwd, uint64_t *out_rc,
                                      uint64_t *out_windows);

/* ---- several GPUs of one node (no reference counterpart: the crate is single-threaded) ----
 * count_kmer calls are independent and read-only (`&self`, src/msbwt_core.rs:125), so the path
 * shards over queries: every device holds a replica of the index, a batch is cut into contiguous
 * shards that start at multiples of 16 queries, and the counts of all shards end up in ONE buffer.
 *
 * msbwt_rle_replicate: a new, independent handle on `device` holding a copy of src's loaded index --
 * blocks, suffix table, presence filter and pair index travel GPU -> GPU (hipMemcpyPeerAsync: over
 * xGMI when peer access is available) instead of being uploaded and rebuilt per device.  NULL on
 * error (msbwt_rle_last_error(src) says why).  `device` may equal src's own (tests). */
msbwt_rle *msbwt_rle_replicate(const msbwt_rle *src, int device);
/* Host batch over `n_replicas` handles (normally one per GPU; replicas[0] may be the original):
 * one host thread and one pinned pipeline per replica, each shard's counts are copied straight
 * into out_counts -- no collective is needed.  Same arguments and results as
 * msbwt_rle_count_kmers / msbwt_rle_count_read_kmers on a single handle. */
int msbwt_rle_count_kmers_multi(const msbwt_rle *const *replicas, size_t n_replicas, const uint8_t *kmers,
                                size_t k, size_t n, uint64_t *out_counts);
int msbwt_rle_count_read_kmers_multi(const msbwt_rle *const *replicas, size_t n_replicas, const uint8_t *reads,
                                     size_t read_len, size_t n_reads, size_t k, int ascii, uint64_t *out_fwd,
                                     uint64_t *out_rc);
/* Device batch: d_kmers and d_out_counts live on replicas[0]'s device.  Shard r is copied to
 * replica r's device, counted there and its counts are copied back into d_out_counts, all by peer
 * copies on that replica's stream (the gather over xGMI).  Synchronous: d_kmers must be complete
 * before the call, d_out_counts is complete when it returns. */
int msbwt_rle_count_kmers_multi_device(const msbwt_rle *const *replicas, size_t n_replicas, const void *d_kmers,
                                       size_t k, size_t n, void *d_out_counts);

/* ---- one process per GPU (the shape `bench.py --gpus N` and an MPI-style Rust host use): every rank holds its
 * own handle with a full index, counts its shard of the batch with msbwt_rle_count_kmers_device, and ONE
 * collective leaves every rank with all counts -- ncclAllGather over RCCL / xGMI, the only exchange step of this
 * path (queries are `&self`, src/msbwt_core.rs:125: no rank needs anything from another one before that).
 * librccl.so is bound at run time (an RCCL already mapped into the process, e.g. PyTorch's, is shared);
 * without it these calls return MSBWT_ERR_RCCL.
 *
 * Communicator bootstrap, for hosts that do not link RCCL themselves: rank 0 obtains an id
 * (MSBWT_COMM_ID_BYTES opaque bytes), hands it to the other ranks by whatever channel the host has (a file, MPI,
 * an environment variable), and every rank calls msbwt_comm_init_rank with its GPU current.  A communicator the
 * host created itself (ncclComm_t) is accepted just as well. */
#define MSBWT_COMM_ID_BYTES 128
int msbwt_comm_get_unique_id(void *out_id);
int msbwt_comm_init_rank(void **out_comm, int nranks, const void *id, int rank);
int msbwt_comm_destroy(void *comm);
/* d_all[r * n_mine + i] = rank r's d_mine[i] (u64 counts, device pointers on the handle's device; every rank
 * passes the same n_mine -- pad the last shard), asynchronous on `hip_stream`.  wire_bits = 64, or 32 / 16: the
 * counts travel narrowed and are widened on arrival (8 bytes per count over xGMI can cost more than the search
 * itself); a count that does not fit makes msbwt_rle_device_status return MSBWT_ERR_OVERFLOW -- repeat the
 * gather with 64.  Calls on one handle must be ordered by the caller (one stream): they share scratch memory. */
int msbwt_rle_allgather_counts(const msbwt_rle *bwt, void *comm, const void *d_mine, size_t n_mine, void *d_all, int wire_bits,
                               void *hip_stream);

/* ONE batch counted and gathered as a pipeline.  A caller that counts its shard with msbwt_rle_count_kmers_device and then calls
 * msbwt_rle_allgather_counts sees kernel + gather + widening one after the other (a steady stream of batches hides the gather behind the
 * next batch's kernel; a single batch cannot).  This call cuts the rank's shard of n_mine queries into `pieces` (1..64; pieces of whole
 * 16-query units) and searches piece i on hip_stream while the counts of piece i - 1 travel -- narrowed to wire_bits, ncclAllGather,
 * put at their place -- on a second stream of the handle; hip_stream continues when the last piece has arrived.
 * d_kmers: n_mine x k symbol codes (this rank's shard); d_mine_counts: n_mine u64 (this rank's counts, also an output);
 * d_all[r * n_mine + i] = rank r's count i as out_bits-wide unsigned integers: 64, or the wire width ("narrow at destination": no
 * widening pass and an array a quarter or half the size -- 8 GB less to write for the 10^9-query line of BASELINE configs[4]).
 * Every rank passes the same n_mine, wire_bits, out_bits and pieces.  A count that does not fit the wire width makes
 * msbwt_rle_device_status return MSBWT_ERR_OVERFLOW.  Calls on one handle must be ordered by the caller (they share scratch). */
int msbwt_rle_count_kmers_allgather_device(const msbwt_rle *bwt, void *comm, const void *d_kmers, size_t k, size_t n_mine, void *d_mine_counts,
                                           void *d_all, int wire_bits, int out_bits, int pieces, void *hip_stream);
/* How that call cuts a shard (a pure function, no device needed): queries per piece -- whole 16-query units, at most `pieces` pieces
 * for any n_mine, the last one takes what is left. */
size_t msbwt_allgather_piece_queries(size_t n_mine, int pieces);

/* ---- batch order (no reference counterpart) ----
 * The order in which a batch is handed over does not change a single count, but it changes how fast they come: a batch whose
 * k-mers are ordered by (their last 17 symbols as a string, then the symbols before those going leftwards) keeps neighbouring
 * queries on neighbouring index lines through the search -- measured up to 2.0x on dense batches over the DIRECT suffix table
 * (10^8 read-derived 31-mers over a 2 x 10^9-symbol BWT, rounds 3-4).  msbwt_kmer_order_keys hands out the 64-bit key to sort by
 * (ascending), for callers that hold a sorted k-mer list anyway or count a batch more than once.
 * kmers: n x k symbol codes; a '$' / 'N' / invalid symbol among the (at most 31) symbols the key reads gives UINT64_MAX. */
int msbwt_kmer_order_keys(const uint8_t *kmers, size_t k, size_t n, uint64_t *out_keys);
/* The library can also order a batch itself, inside the launch (round 4): the queries are packed to two bits per symbol,
 * bucket-ordered on the device by the top 22 key bits (MSBWT_ORDER_BITS) in two passes, counted in index order, and every count
 * is returned to its query's own place in the caller's buffer -- the caller sees its order.  It is a SWITCH, off unless asked for:
 * mode 1 = whenever the passes apply (pair index, 12 <= k <= 64, 4096 <= n < 2^32); 0 = never; -1 = the default, which is never
 * too (kept as a value so that callers written against round 4 keep working).  There is no automatic mode: ordering 10^8 queries
 * and un-ordering their counts costs about 6 ms of the 8 ms the ordered search saved on the densest batches of round 4, sparse or
 * random batches lost 30 % to 2.7x -- and since round 5 the default index looks its queries up in a HASHED sparse suffix table,
 * whose lookups an order cannot help: the same dense batches now LOSE 6-14 % with the pass forced on (round 5, `library_ordered` beside
 * the c4_* lines of bench.py: 7.5 -> 6.8 and 7.9 -> 7.4 x 10^9 q/s).  MSBWT_ORDER=0|1 in the environment sets the initial mode.
 * Applies to msbwt_rle_count_kmers[_device] and the packed forms.  msbwt_rle_batch_order_for: 1 if a batch of n k-symbol queries
 * would be ordered now.  Results never change. */
int msbwt_rle_set_batch_order(msbwt_rle *bwt, int mode);
int msbwt_rle_get_batch_order(const msbwt_rle *bwt);
int msbwt_rle_batch_order_for(const msbwt_rle *bwt, size_t k, size_t n);
int msbwt_rle_kmer_order_keys_device(const msbwt_rle *bwt, const void *d_kmers, size_t k, size_t n, void *d_out_keys, void *hip_stream);

/* ---- tuning / introspection (no reference counterpart) ---- */
/* Depth of the precomputed suffix table (the reference's stubbed kmer_cache,
 * src/msbwt_core.rs:133-146): ranges for every ACGT suffix of length `depth` are computed
 * on the device at load time and replace the first `depth` steps of each query.  0 turns it
 * off, a negative value restores the automatic choice (the deepest table the data warrants
 * within max(1 GiB, 2 x block bytes), at most 15), the maximum is 16 (64 GiB).  Takes effect
 * immediately if an index is loaded.  Results never change. */
int msbwt_rle_set_table_depth(msbwt_rle *bwt, int depth);
int msbwt_rle_get_table_depth(const msbwt_rle *bwt);
/* Packed suffix table: beside a pair index the finished table can be extended by TWO more levels
 * and stored as 128-byte lines of 30 entries (16-bit deltas; 4.27 bytes per entry instead of 16),
 * e.g. depth 17 in 73 GB for a 30x human BWT.  One line fetch per query as before, one search step
 * fewer -- and it is the widest-range step, the one that usually needs two lines.  mode 1 = on
 * (whenever a pair index exists), 0 = off, -1 = automatic (default: when the table depth itself is
 * automatic, 4^(depth+2) <= 256 x total symbols and the lines fit in half of the free HBM -- the flat
 * parent table is then built as deep as that packed table needs, up to 15 levels, and freed;
 * MSBWT_TABLE_PACKED=0/1 overrides).
 * msbwt_rle_get_table_depth reports the effective depth (flat depth + 2).  Lines whose deltas do not
 * fit 16 bits are marked ESCAPE and keep their ranges in a side array (msbwt_rle_set_table_side, below).  Results never change.
 * Beside a sparse suffix table (msbwt_rle_set_sparse_table, the default since round 5) the automatic direct table stays at packed
 * depth 15 at most: only queries shorter than the sparse table's depth still read it. */
int msbwt_rle_set_table_packed(msbwt_rle *bwt, int mode);
/* Escape lines of the packed table -- lines holding a range 2^16 or more wide, or further than that from the line's
 * first range: the suffixes of high-copy repeats (Alu-, L1-like families, satellites) -- keep their 30 ranges as flat
 * 16-byte entries in a SIDE array (512 bytes per such line): a query that lands on one pays one more line fetch, like any
 * search step, instead of searching from scratch (the reference's constrain_range costs the same for any range width,
 * src/rle_bwt.rs:202-287).  mode 1 = on (default), 0 = off (MSBWT_TABLE_SIDE=0 in the environment).  Results never change.
 * msbwt_rle_table_info: lines of the packed table in HBM, how many are escape lines, bytes of the side array (all 0
 * without a packed table). */
int msbwt_rle_set_table_side(msbwt_rle *bwt, int mode);
int msbwt_rle_table_info(const msbwt_rle *bwt, uint64_t *lines, uint64_t *escape_lines, uint64_t *side_bytes);
/* The automatic choice, as a pure function (no device needed): levels of the flat table built first and of
 * the packed table it becomes (0 = stays flat), for an index of `total_symbols` symbols with `free_hbm_bytes`
 * of HBM free once plane and pair blocks are in place. */
int msbwt_auto_table_depths(uint64_t total_symbols, uint64_t free_hbm_bytes, int pair_index, int *flat_depth, int *packed_depth);
int msbwt_rle_get_table_packed(const msbwt_rle *bwt);
/* Sparse suffix table (the reference's stubbed kmer_cache, src/msbwt_core.rs:133-146 / src/rle_bwt.rs:332-346, taken past what a
 * direct-address table can hold): the ranges of the `depth`-symbol suffixes that OCCUR in the BWT, 16 <= depth <= 31, as a hashed
 * table of 128-byte buckets of 14 entries (12 with 32-bit tags from depth 25 on, 11 with 40-bit tags from depth 30 on; layout and hash: rust-msbwt_amd/csrc/sparse_table.hpp).  The direct table above has 4^depth
 * entries whatever the data -- 73 GB at depth 17, of which a 30x human read set can fill 17 % and a chr20-sized one 0.4 % -- while a
 * table of the present suffixes reaches depth 23 in about 14 bytes per distinct 23-mer: every present 31-mer is three pair steps
 * (of seven) shorter.  One lookup = one random 128-byte line, fetched like any search step's; the table is complete, so a miss is
 * count 0 (the early exit of src/msbwt_core.rs:151-153); entries 255 or more wide (suffixes of high-copy repeats) keep their range
 * in a side array, one more line.  Built on the device by frontier expansion with the index's own rank code (each present d-mer ->
 * its <= 16 present (d+2)-mers by one pair step), which also yields the DISTINCT counts per depth that size it.  Needs the pair
 * index; the one-query-per-lane kernel uses it for every batch with k >= depth, shorter k-mers (and k-mers with '$' / 'N' among
 * their last `depth` symbols) use the direct table, which then stays small (packed depth 15 at most when automatic).
 * depth: -1 = automatic (default: the deepest depth <= 23 whose table fits HBM -- and a memory budget, if one is set -- with an
 * eighth of the device left free; none if no depth fits, e.g. a read set whose error k-mers outnumber everything), 0 = off, 16..31 =
 * exactly that depth (an error if it cannot be built; a table serves k >= its depth: a caller that counts 31-mers gains two pair
 * steps per query from depth 27 -- 3 index lines instead of 5 -- and loses the table for k = 23..26).  MSBWT_SPARSE_TABLE=<depth>|auto|0 in the environment sets the initial mode.
 * Takes effect immediately if an index is loaded.  Results never change.
 * msbwt_rle_get_sparse_table: depth of the table in HBM, 0 = none.
 * msbwt_rle_sparse_table_info: out[MSBWT_SPARSE_INFO_WORDS] = [0] depth, [1] entries, [2] buckets, [3] bytes of the bucket lines,
 * [4] entries in the side array, [5] its bytes, [6] entries displaced to a later bucket, [7] depth of the direct table the build
 * started from, [9] buckets a lookup may go beyond its own, [10 + d] DISTINCT d-symbol suffixes that occur (d = 0..31; 0 where the
 * build did not pass: it advances two symbols at a time), [45 + d] of which 255 or more wide, [80 + d] of which exactly 1 wide (suffixes
 * that occur once: on reads with errors, mostly error k-mers), [8] 1 = the table is of the two-tier form, [42] suffixes in its filters,
 * [43] depth of the SECOND, shallower level (0 = none) and [44] its bytes: with k undeclared the table is 23 deep and serves k >= 23; where
 * the deep direct table of an index without a sparse table (packed depth 17) does not fit beside it but a table of the 17-symbol suffixes
 * does, that one is built as well and serves 17 <= k < 23 (MSBWT_SPARSE_SECOND=0: never), so that no k loses to the index without a
 * sparse table.
 * The counts are kept even when no table was built.
 * TWO-TIER form (msbwt_rle_set_sparse_tiers; rust-msbwt_amd/csrc/sparse_table.hpp): the complete table's size follows the distinct
 * suffixes, and on reads WITH errors most of those occur once (a 30x human read set with 0.5 % substitutions: 1.3e10 distinct 23-mers,
 * 185 GB, of which about 3e9 -- the genome's -- occur more than once).  The two-tier table keeps entries only for the suffixes whose
 * range is at least 2 wide and sets, for each of the others, four bits of a filter inside its own bucket line (no false negatives):
 * a lookup that finds its tag is served as before, one that finds neither tag nor filter bits is count 0, and one whose filter bits are
 * set continues through the direct table and the search -- the reference's own path (src/rle_bwt.rs:202-287) -- so every count stays
 * exact whatever the filter says.  mode: -1 = automatic (default: the two-tier form of a depth where its complete table does not fit
 * HBM or the memory budget -- tried before the next shallower depth), 0 = complete tables only, 1 = two-tier only (tests, measurements).
 * MSBWT_SPARSE_TIERS=auto|0|1 sets the initial mode.  Takes effect immediately if an index is loaded.  Results never change.
 * msbwt_rle_get_sparse_tiers: 1 when the table in HBM is of the two-tier form, else 0.
 * msbwt_sparse_hash / msbwt_sparse_table_shape: the table's hash and sizing as pure functions (no device needed).
 * msbwt_rle_download_sparse_table: copies the bucket lines (and the side array) to the host; returns the bytes of the lines,
 * SIZE_MAX without a table or on error (tests check every entry against the oracle). */
#define MSBWT_SPARSE_INFO_WORDS 120
int msbwt_rle_set_sparse_table(msbwt_rle *bwt, int depth);
int msbwt_rle_get_sparse_table(const msbwt_rle *bwt);
int msbwt_rle_set_sparse_tiers(msbwt_rle *bwt, int mode);
int msbwt_rle_get_sparse_tiers(const msbwt_rle *bwt);
/* The second, shallower sparse level ([43] / [44] of msbwt_rle_sparse_table_info): -1 = automatic (default: k undeclared, plane blocks, the
 * deep direct table does not fit beside the sparse table but a table of the 17-symbol suffixes does), 0 = never.  MSBWT_SPARSE_SECOND=0|auto
 * sets the initial mode.  Takes effect immediately if an index is loaded.  Results never change. */
int msbwt_rle_set_sparse_second(msbwt_rle *bwt, int mode);
int msbwt_rle_sparse_table_info(const msbwt_rle *bwt, uint64_t *out);
int msbwt_sparse_hash(uint64_t key, int depth, uint64_t nbuckets, uint32_t *bucket, uint32_t *tag);
int msbwt_sparse_hash64(uint64_t key, int depth, uint64_t nbuckets, uint32_t *bucket, uint64_t *tag); /* the whole tag: 24, 32 or 40 bits by depth */
int msbwt_sparse_table_shape(int depth, uint64_t entries, uint64_t *nbuckets, int *probe);
/* The automatic depth as a pure function (no device needed; rust-msbwt_amd/csrc/sparse_policy.hpp): distinct[d] / wide[d] for d = 0..31
 * as msbwt_rle_sparse_table_info reports them ([10 + d], [45 + d]), the depth of the direct table the count started from, and the bytes
 * the table and its build scratch may take -> the depth the loader would build (0 = none fits) and the bytes of that table. */
int msbwt_auto_sparse_depth(const uint64_t *distinct, const uint64_t *wide, int parent_depth, uint64_t avail_bytes, int query_length, int *depth, uint64_t *table_bytes);
/* ... with the two-tier form in the choice: singles[d] as [80 + d] above, tiers as msbwt_rle_set_sparse_tiers -> also *two_tier (0 / 1). */
int msbwt_auto_sparse_choice(const uint64_t *distinct, const uint64_t *wide, const uint64_t *singles, int parent_depth, uint64_t avail_bytes, int query_length,
                             int tiers, int *depth, int *two_tier, uint64_t *table_bytes);
/* The two-tier filter as a pure function: the filter word (0..7, i.e. word 23 + that of the bucket line) and the four bits of a key with this tag. */
int msbwt_sparse_filter_bits(uint64_t tag, uint32_t *word, uint32_t *mask);
/* The k the index will mostly be asked about (0 = unknown, the default; MSBWT_QUERY_K in the environment sets the initial value).  The
 * reference's count_kmer takes any k per call and so does this library -- results never depend on the hint -- but a hashed table of
 * d-mers serves k >= d only, and every two symbols of depth save a present k-mer one index line: with k unknown the automatic sparse
 * table stops at depth 23 (every k >= 23 is served: 5 lines for a present 31-mer); a caller that declares k = 31 gets depth 31 -- the table's range IS the count, one line per query
 * (depth 29: 2 lines, 1.92e10 present 31-mers/s at 30x-human scale; depth 27: 3 lines, 1.34e10; k unknown: 5 lines, 8.4e9) -- and k = 21 gets depth 21.  Shorter k-mers than the table's depth use the
 * direct table as before.  Only the AUTOMATIC depth follows the hint (msbwt_rle_set_sparse_table(-1)); it takes effect immediately if an
 * index is loaded (the tables are rebuilt).  msbwt_auto_sparse_max_depth: the rule as a pure function. */
int msbwt_rle_set_query_length(msbwt_rle *bwt, int k);
int msbwt_rle_get_query_length(const msbwt_rle *bwt);
int msbwt_auto_sparse_max_depth(int query_length);
size_t msbwt_rle_download_sparse_table(const msbwt_rle *bwt, void *out_lines, size_t cap_bytes, void *out_side, size_t cap_side_bytes);
/* Presence filter: one bit per ACGT suffix of length min(12, table depth), set when some table
 * entry with that suffix is a non-empty range; at most 2 MiB, so it lives in L2 and decides
 * absent k-mers (random queries, small genomes) without fetching the table line.  Built on the
 * device from the table; dropped automatically when more than 90 % of its bits are set (it
 * could reject almost nothing).  mode 0 = off, otherwise automatic (MSBWT_FILTER=0 in the
 * environment turns it off).  get returns the filter depth, 0 if none.  Results never change. */
int msbwt_rle_set_presence_filter(msbwt_rle *bwt, int mode);
int msbwt_rle_get_presence_filter(const msbwt_rle *bwt);
/* Pair index: a second block array (1 byte per symbol) that stores, next to each BWT symbol,
 * the symbol one LF step further, so that one search step consumes TWO k-mer symbols for one
 * line fetch per bound (maths: rust-msbwt_amd/csrc/rank_ops.hpp).  Built on the device from
 * the loaded index.  mode 1 = on, 0 = off, -1 = on when it fits in half of the free HBM
 * (default; MSBWT_PAIR_INDEX=0/1 in the environment overrides).  Results never change. */
int msbwt_rle_set_pair_index(msbwt_rle *bwt, int mode);
int msbwt_rle_get_pair_index(const msbwt_rle *bwt);
/* Spacing of the pair blocks: 128 = disjoint blocks (1 byte per symbol); 96 = overlapping blocks that
 * also hold the first 32 positions of their successor (1.33 bytes per symbol), so that a range up to
 * 32 wide is ranked from ONE line -- on real 30x data ranges stay ~25 wide to the last step and every
 * fifth step would otherwise fetch a second line.  0 = automatic (default; MSBWT_PAIR_STRIDE=96|128
 * overrides): 96 when that takes at most a quarter of the free HBM; otherwise the DATA decide -- at load time a
 * few thousand present 24-mers are read off the index by LF walks and counted, and overlapping blocks are built
 * when their median count (msbwt_rle_get_typical_range_width) is >= 8 and the bigger blocks fit beside the
 * suffix table with an eighth of the HBM to spare.  get returns 0 without a pair index.  Results never change. */
int msbwt_rle_set_pair_stride(msbwt_rle *bwt, int stride);
int msbwt_rle_get_pair_stride(const msbwt_rle *bwt);
/* Median number of occurrences of a 24-mer that is present in the index (4096 LF walks from pseudo-random rows,
 * probed once per load): about the coverage on a real read set, 1 on a stream of independent symbols; the width
 * the ranges of surviving queries keep through the search.  -1 = not probed (run blocks, empty index). */
double msbwt_rle_get_typical_range_width(const msbwt_rle *bwt);
/* The automatic spacing as a pure function (no device needed): for an index of `total_symbols` symbols with
 * `free_hbm_bytes` free once the plane blocks are in place on a device of `hbm_total_bytes`, given the typical
 * range width the probe reports (negative = unknown). */
int msbwt_auto_pair_stride(uint64_t total_symbols, uint64_t free_hbm_bytes, uint64_t hbm_total_bytes, double typical_width, int *stride);
/* A memory budget for the whole index -- the analogue of the reference's one space / time knob, `bin_power`
 * (src/rle_bwt.rs:309-322; here results never depend on it, only speed).  bytes = HBM the loaded index may hold, 0 = no budget
 * (default; MSBWT_MEMORY_BUDGET=<bytes> in the environment sets the initial value).  The plane blocks (0.5 byte per symbol) are
 * always built; what the budget leaves goes, in this order, to the pair blocks (whenever they fit), to the
 * deepest packed suffix table that fits, and to overlapping pair blocks when the data keep ranges wide -- one plan, made
 * at load time (or at once, if an index is loaded: the optional structures are rebuilt).  Explicit settings
 * (msbwt_rle_set_table_depth, _set_pair_index, _set_pair_stride, an explicit _set_sparse_table depth) still win over the plan.
 * The sparse suffix table takes what the budget leaves once blocks, pair blocks and a direct table of packed depth 15 are paid for
 * (the deepest depth that fits; none if nothing does -- the direct table is then as deep as the plan says).  Not counted against
 * the budget: the presence filter (2 MiB at most) and the side arrays of the tables (bytes per high-copy suffix).  A budget below
 * the plane blocks cannot be met -- they are built all the same -- and run blocks have no optional structure to plan: both are
 * said by msbwt_rle_last_error after the call, which still returns MSBWT_OK.
 * msbwt_auto_index_plan: the plan as a pure function (no device needed) for an index of `total_symbols` symbols with
 * `free_hbm_bytes` free once its plane blocks are in place; *index_bytes (optional) = what the planned index holds
 * (presence filter, at most 2 MiB, and the table's side array not counted). */
int msbwt_rle_set_memory_budget(msbwt_rle *bwt, uint64_t bytes);
uint64_t msbwt_rle_get_memory_budget(const msbwt_rle *bwt);
int msbwt_auto_index_plan(uint64_t total_symbols, uint64_t free_hbm_bytes, uint64_t hbm_total_bytes, double typical_width, uint64_t budget_bytes,
                          int *pair_index, int *pair_stride, int *flat_depth, int *packed_depth, uint64_t *index_bytes);
/* Block format of the index, chosen BEFORE a load (MSBWT_BLOCKS=runs in the environment sets the
 * initial choice): 0 = bit-plane blocks (default: 0.5 byte per symbol, fastest, the only format the
 * pair index, the packed table and the batch-ordering pass work on), 1 = run blocks -- the layout of the
 * reference's run_block_av_flat (src/run_block_av_flat.rs:43-56,97-125): 128-byte blocks of 512 positions
 * with per-block counts and 96 one-byte runs, ~0.3 byte per symbol on 30x short-read BWTs, single-symbol
 * steps only (no pair index).  Built on the device from the RLE bytes like the default format; for
 * 6 <= k <= 64 the lane-per-query kernel serves it too (each lane decodes its own block's runs from LDS):
 * a 30x human-scale BWT in 44 GB at 1.8 x 10^9 present 31-mers/s (bit planes + pair blocks + table:
 * 238 GB, 5.4 x 10^9).  For replicas that must leave HBM to others.  Results never change. */
/* msbwt_run_build_fits_device (pure, no device needed): 1 when the DEVICE builder of the run-block format fits `free_hbm_bytes` of free
 * HBM for an index of that many symbols -- it holds the plane blocks (0.5 byte per symbol) beside the run blocks (~0.3) for a moment;
 * when it does not (or an allocation fails on the way) the load builds the run blocks on the host and uploads them, as MSBWT_BUILD=host
 * always does: an index that fits only BECAUSE the format is lean still loads. */
int msbwt_run_build_fits_device(uint64_t total_symbols, uint64_t free_hbm_bytes);
int msbwt_rle_set_block_format(msbwt_rle *bwt, int format);
int msbwt_rle_get_block_format(const msbwt_rle *bwt);
/* Search kernel for 1 <= k <= 64: 0 = automatic (default: 2 whenever a pair index exists and k >= 6, and on run
 * blocks for k >= 6), 1 = 8 lanes per query, lines in registers (kernels.hip; one symbol per step; needs no pair index),
 * 2 = one query per lane, lines staged through LDS by LDS-DMA (lanes.hip; two symbols per step,
 * 8x more random lines in flight per wave).  MSBWT_SEARCH=groups|lanes in the environment
 * sets the initial mode.  Results never change. */
int msbwt_rle_set_search_kernel(msbwt_rle *bwt, int mode);
int msbwt_rle_get_search_kernel(const msbwt_rle *bwt);
/* The kernel a batch of k-symbol queries runs on with the index and mode as they are now: 1 or 2 as
 * above, 0 for k > 64 (generic 8-lane kernel, no suffix table); negative = error. */
int msbwt_rle_search_kernel_for(const msbwt_rle *bwt, size_t k);
/* Search counters (measurement aid, off by default): with them on, every wave of the one-query-per-lane kernel adds what its
 * queries did to a block of MSBWT_SEARCH_COUNTERS u64 -- [0] search steps of a wave, [1] steps taken by a query (one line
 * fetch each), [2] of which two-symbol steps, [3] steps that needed a second line (range over two blocks), [4] steps a query
 * waited because the second-line slots of its wave were taken, [5] queries whose packed-table line is an escape line,
 * [6] of which searched from scratch (no side array), [7] queries decided by the table / presence filter alone, [8] queries
 * that entered the search, [9] first lines fetched, [10] sparse-table bucket lines fetched (included in [1]), [11] of which did not
 * hold the key although the bucket had displaced entries (the lookup went on to the next bucket), [12] of [10], lookups that rode
 * along with the search of the tile before theirs (in second-line slots that step left free) instead of taking a step of their own, [13] waves of the
 * persistent kernel that were dealt at least one tile; the rest 0.  msbwt_rle_search_counters copies the block out and
 * zeroes it (synchronises `hip_stream`, on which the counted launches ran).  Results never change. */
#define MSBWT_SEARCH_COUNTERS 16
int msbwt_rle_set_search_counters(msbwt_rle *bwt, int enabled);
int msbwt_rle_search_counters(const msbwt_rle *bwt, uint64_t *out, void *hip_stream);
/* Cache policy of the index lines.  The one-query-per-lane kernel uses every line it fetches once; on an index whose random-access
 * arrays (the pair blocks, else the blocks themselves) are far larger than L2 and the 256 MB Infinity Cache those lines only evict what
 * IS reused (the superblock table, the query stream), so they are fetched with the non-temporal hint: at 30x-human scale 36.1 ms per
 * 3 x 10^8 present 31-mers (35.6-36.6 over four alternating runs) against 38.4 ms (36.7-39.8) with the default policy; a small index,
 * whose pair blocks half live in the Infinity Cache, LOSES (C3 fused 14.9 -> 19.8 ms).  mode -1 = automatic (default: from 4 GiB of such
 * arrays on), 0 = never, 1 = always; MSBWT_STREAM_LINES=0|1|auto in the environment sets the initial mode.  get: what launches on the
 * loaded index do now.  Results never change. */
int msbwt_rle_set_line_streaming(msbwt_rle *bwt, int mode);
int msbwt_rle_get_line_streaming(const msbwt_rle *bwt);
/* Placement diagnostic: random 128-byte lines per second the memory system serves right now from one of the index's arrays
 * (which: 0 = plane / run blocks, 1 = pair blocks, 2 = the sparse table's bucket lines, 3 = the direct suffix table; 0.0 when the
 * index has no such array) -- about a millisecond of gathering, nothing is written.  (Round 5 used it on the "two modes" of C4-sized
 * indexes -- the same launch 12 or 14 ms per instance of the index: a plain gather does not see them, profiles/r05_lab/two_modes.log.) */
int msbwt_rle_probe_line_rate(const msbwt_rle *bwt, int which, double *lines_per_second);
/* Bytes of HBM held by the index (blocks + table + filter + pair index). */
uint64_t msbwt_rle_device_bytes(const msbwt_rle *bwt);
/* Average duration in ms of the count kernel launches since the last reset, measured with
 * HIP events on the launch stream (bench.py's roofline uses it); resets the accumulator. */
int msbwt_rle_kernel_time_ms(const msbwt_rle *bwt, double *avg_ms, uint64_t *launches);
int msbwt_rle_set_kernel_timing(msbwt_rle *bwt, int enabled);
int msbwt_rle_device_ordinal(const msbwt_rle *bwt);
const char *msbwt_rle_last_error(const msbwt_rle *bwt);
/* Host half of the load path, exposed for inspection (no device needed): expands an RLE
 * stream into the plane blocks the kernels read (layout: rust-msbwt_amd/csrc/plane_index.hpp).
 * Returns the number of 128-byte blocks (= total/256 + 1); writes at most `cap_blocks` of
 * them to out_blocks (may be NULL), the symbol total to *out_total. SIZE_MAX on bad input. */
size_t msbwt_build_plane_blocks(const uint8_t *rle_bytes, size_t len, void *out_blocks, size_t cap_blocks,
                                uint64_t *out_total);
/* The same for the run-block format (layout: rust-msbwt_amd/csrc/run_index.hpp): returns the number
 * of 128-byte blocks (= total/512 + 1); out_overflow receives 256 bytes per overflowing block
 * (*out_noverflow of them).  Call with NULL outputs to size.  SIZE_MAX on bad input. */
size_t msbwt_build_run_blocks(const uint8_t *rle_bytes, size_t len, void *out_blocks, size_t cap_blocks,
                              void *out_overflow, size_t cap_overflow, uint64_t *out_total, uint64_t *out_noverflow);
/* Copies the index as it sits in HBM back to the host (same layout as
 * msbwt_build_plane_blocks); returns the block count, SIZE_MAX on error.  Lets tests compare
 * the device-side builder with the host one. */
size_t msbwt_rle_download_blocks(const msbwt_rle *bwt, void *out_blocks, size_t cap_blocks);
const char *msbwt_version(void);

/* ---- codecs either side of the path ---- */
/* convert_to_vec (src/bwt_converter.rs:26-80): ASCII "$ACGNT" (+ '\n', ignored) -> RLE
 * bytes.  Returns the number of bytes needed/written; out may be NULL to size.  Returns
 * SIZE_MAX for a byte outside the alphabet (the reference panics). */
size_t msbwt_convert_to_vec(const uint8_t *ascii, size_t n, uint8_t *out, size_t cap);
/* save_bwt_numpy (src/bwt_converter.rs:102-130): 96-byte NumPy v1.0 header + payload */
int msbwt_save_bwt_numpy(const uint8_t *rle_bytes, size_t n, const char *utf8_path);
/* save_bwt_runs_numpy (src/bwt_converter.rs:151-184) */
int msbwt_save_bwt_runs_numpy(const uint8_t *syms, const uint64_t *counts, size_t nruns,
                              const char *utf8_path);
/* string_util (src/string_util.rs:3-88) */
void msbwt_convert_stoi(const uint8_t *ascii, size_t n, uint8_t *out_codes);
void msbwt_convert_itos(const uint8_t *codes, size_t n, uint8_t *out_ascii);
void msbwt_reverse_complement_i(const uint8_t *codes, size_t n, uint8_t *out_codes);

#ifdef __cplusplus
}
#endif
#endif
