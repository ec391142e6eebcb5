/*
 * msbwt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's RleBWT count_kmer / constrain_range
 * hot path and of the small codecs either side of it.  It exists so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg have something to check the
 * HIP path against.  Nothing in the product (rust-msbwt_amd/, include/) may include,
 * link or call it.
 *
 * Parity pinning: the reference is Rust and cannot be built in this image (no
 * cargo/rustc, no network), so there is no oracle/_ref.  The restatement is pinned by
 * every golden vector the reference's own tests hold for this path (SURVEY.md 8c,
 * G1..G10) -- see tests/test_oracle_golden.py.
 *
 * Each function cites the reference file:line (relative to the reference repo root)
 * whose behaviour it restates.
 */
#ifndef MSBWT_ORACLE_H
#define MSBWT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/msbwt_core.rs:3-14 */
#define ORC_VC_LEN 6
#define ORC_LETTER_BITS 3
#define ORC_NUMBER_BITS 5
#define ORC_NUM_POWER 32
#define ORC_MASK 0x07
#define ORC_COUNT_MASK 0x1F

/* return codes */
#define ORC_OK 0
#define ORC_ERR_IO (-1)          /* open/metadata/short read: io::Error            */
#define ORC_ERR_EOF (-2)         /* size mismatch: io::ErrorKind::UnexpectedEof    */
#define ORC_ERR_HEADER (-3)      /* the reference panics on these headers          */
#define ORC_ERR_SYMBOL (-4)      /* the reference panics (assert / OOB index)      */
#define ORC_ERR_RANGE (-5)       /* position outside [0,total] or l > h            */

/* src/rle_bwt.rs:14-24 -- same fields, same (SoA) layout */
typedef struct orc_rle_bwt {
    uint8_t *bwt;
    size_t bwt_len;
    uint64_t symbol_counts[ORC_VC_LEN];
    uint64_t start_index[ORC_VC_LEN];
    uint64_t end_index[ORC_VC_LEN];
    uint64_t *fm_index[ORC_VC_LEN];
    uint64_t *ref_index;
    size_t index_length;
    uint64_t total_size;
    uint8_t bin_power;
    uint64_t bin_size;
} orc_rle_bwt;

/* Counters that define the roofline denominator: one "bin visit" =
 * 56 B of sample + the RLE bytes the scan loop consumed. */
typedef struct orc_stats {
    uint64_t queries;
    uint64_t steps;       /* constrain_range calls executed             */
    uint64_t visits;      /* bin visits (1 or 2 per step)               */
    uint64_t scan_bytes;  /* RLE bytes consumed by the two scan loops   */
} orc_stats;

/* ---- src/rle_bwt.rs ---- */
orc_rle_bwt *orc_rle_new(uint8_t bin_power);                       /* :297-322 */
void orc_rle_free(orc_rle_bwt *b);
int orc_rle_load_vector(orc_rle_bwt *b, const uint8_t *bytes, size_t n); /* :59-66 */
int orc_rle_load_numpy_file(orc_rle_bwt *b, const char *path);     /* :81-155 */
uint64_t orc_rle_get_symbol_count(const orc_rle_bwt *b, uint8_t s); /* :172-175 */
uint64_t orc_rle_get_total_size(const orc_rle_bwt *b);             /* :190-193 */
int orc_rle_constrain_range(const orc_rle_bwt *b, uint8_t sym, uint64_t l, uint64_t h,
                            uint64_t *out_l, uint64_t *out_h, orc_stats *st); /* :202-287 */
/* src/msbwt_core.rs:124-161 */
int orc_rle_count_kmer(const orc_rle_bwt *b, const uint8_t *kmer, size_t k,
                       uint64_t *out, orc_stats *st);
/* batch helpers (no reference counterpart: a loop over count_kmer, optionally over
 * nthreads pthreads with a static partition -- legal because queries take &self) */
int orc_rle_count_kmers(const orc_rle_bwt *b, const uint8_t *kmers, size_t k, size_t n,
                        uint64_t *out, int nthreads, orc_stats *st);
int orc_rle_constrain_ranges(const orc_rle_bwt *b, const uint8_t *syms, const uint64_t *l,
                             const uint64_t *h, size_t n, uint64_t *out_l, uint64_t *out_h);
/* accessors for the sampled index (tests compare them with rle_bwt.rs:520-600) */
size_t orc_rle_index_length(const orc_rle_bwt *b);
const uint64_t *orc_rle_ref_index(const orc_rle_bwt *b);
const uint64_t *orc_rle_fm_index(const orc_rle_bwt *b, int sym);
const uint64_t *orc_rle_start_index(const orc_rle_bwt *b);
const uint64_t *orc_rle_end_index(const orc_rle_bwt *b);

/* ---- src/bwt_converter.rs ---- */
/* :26-80; returns bytes written (or needed when out==NULL), (size_t)-1 on bad symbol */
size_t orc_convert_to_vec(const uint8_t *ascii, size_t n, uint8_t *out, size_t cap);
int orc_save_bwt_numpy(const uint8_t *bytes, size_t n, const char *path);    /* :102-130 */
int orc_save_bwt_runs_numpy(const uint8_t *syms, const uint64_t *counts, size_t nruns,
                            const char *path);                               /* :151-184 */

/* ---- src/string_util.rs ---- */
void orc_convert_stoi(const uint8_t *ascii, size_t n, uint8_t *out);         /* :15-32,63-67 */
void orc_convert_itos(const uint8_t *codes, size_t n, uint8_t *out);         /* :6-9,80-88 */
void orc_reverse_complement_i(const uint8_t *codes, size_t n, uint8_t *out); /* :12,45-50 */

/* ---- src/bwt_util.rs:154-171 ---- */
/* strings: nstr NUL-terminated ASCII strings; out must hold sum(len+1) bytes; returns
 * the BWT length */
size_t orc_naive_bwt(const char *const *strings, size_t nstr, uint8_t *out);

/* ---- independent cross-checks (no shared code with the scan above) ---- */
/* expand RLE bytes to one symbol code per position; returns symbols written (or needed
 * when out==NULL) */
uint64_t orc_decompress(const uint8_t *bytes, size_t n, uint8_t *out, uint64_t cap);
/* rank by brute force over the plain symbol vector */
uint64_t orc_rank_bruteforce(const uint8_t *symbols, uint64_t n, uint8_t sym, uint64_t pos);
/* src/run_block_av_flat.rs:97-125 restated over a flat u16 run array */
uint64_t orc_runblock_count(const uint16_t *runs, size_t nruns, uint64_t position, uint8_t symbol);

#ifdef __cplusplus
}
#endif
#endif
