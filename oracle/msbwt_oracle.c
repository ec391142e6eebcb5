/*
 * msbwt_oracle.c -- TEST INFRASTRUCTURE ONLY (see msbwt_oracle.h).
 *
 * Plain-C CPU restatement of the reference's RleBWT query path.  Parity is pinned by the
 * reference's own golden vectors (tests/test_oracle_golden.py); there is no oracle/_ref
 * because the reference is Rust and this image has no Rust toolchain.
 */
#define _GNU_SOURCE
#include "msbwt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

/* ------------------------------------------------------------------------------------
 * RleBWT construction
 * ---------------------------------------------------------------------------------- */

/* src/rle_bwt.rs:297-322 (new() == with_bin_power(8)) */
orc_rle_bwt *orc_rle_new(uint8_t bin_power) {
    orc_rle_bwt *b = (orc_rle_bwt *)calloc(1, sizeof(*b));
    if (!b) return NULL;
    b->bin_power = bin_power;
    b->bin_size = (uint64_t)1 << bin_power;
    return b;
}

static void drop_index(orc_rle_bwt *b) {
    for (int s = 0; s < ORC_VC_LEN; ++s) {
        free(b->fm_index[s]);
        b->fm_index[s] = NULL;
    }
    free(b->ref_index);
    b->ref_index = NULL;
    b->index_length = 0;
}

void orc_rle_free(orc_rle_bwt *b) {
    if (!b) return;
    drop_index(b);
    free(b->bwt);
    free(b);
}

/* src/rle_bwt.rs:352-384 -- one pass; a byte whose symbol equals the previous byte's is
 * the next base-32 digit of the same run. */
static void calculate_totals(orc_rle_bwt *b) {
    uint8_t prev = 255;
    uint64_t weight = 1;
    memset(b->symbol_counts, 0, sizeof(b->symbol_counts));
    for (size_t i = 0; i < b->bwt_len; ++i) {
        uint8_t v = b->bwt[i];
        uint8_t sym = v & ORC_MASK;
        weight = (sym == prev) ? weight * ORC_NUM_POWER : 1;
        prev = sym;
        b->symbol_counts[sym] += (uint64_t)(v >> ORC_LETTER_BITS) * weight;
    }
    uint64_t acc = 0;
    for (int s = 0; s < ORC_VC_LEN; ++s) {
        b->start_index[s] = acc;
        acc += b->symbol_counts[s];
        b->end_index[s] = acc;
    }
    b->total_size = b->end_index[ORC_VC_LEN - 1];
}

/* src/rle_bwt.rs:387-467.  A sample for bin boundary B is taken when the run that covers
 * position B is *finished*: it records the byte offset of that run's first byte and the
 * symbol counts before that run (run-aligned, not boundary-aligned).  The initial state
 * (prev symbol 0, weight 1, empty run) makes a leading '$' byte a continuation digit. */
static int construct_fmindex(orc_rle_bwt *b) {
    drop_index(b);
    size_t n = (size_t)ceil((double)b->total_size / (double)b->bin_size) + 1; /* :390 */
    for (int s = 0; s < ORC_VC_LEN; ++s) {
        b->fm_index[s] = (uint64_t *)calloc(n, sizeof(uint64_t));
        if (!b->fm_index[s]) return ORC_ERR_IO;
    }
    b->ref_index = (uint64_t *)calloc(n, sizeof(uint64_t));
    if (!b->ref_index) return ORC_ERR_IO;
    b->index_length = n;

    uint64_t before[ORC_VC_LEN] = {0, 0, 0, 0, 0, 0}; /* counts before the open run */
    uint64_t run_len = 0, weight = 1, boundary = 0, pos = 0;
    size_t bin = 0, run_first_byte = 0;
    uint8_t run_sym = 0;

    for (size_t x = 0; x <= b->bwt_len; ++x) {
        int at_end = (x == b->bwt_len);
        uint8_t v = at_end ? 0 : b->bwt[x];
        uint8_t sym = v & ORC_MASK;
        if (!at_end && sym == run_sym) { /* :414-417 */
            run_len += (uint64_t)(v >> ORC_LETTER_BITS) * weight;
            weight *= ORC_NUM_POWER;
            continue;
        }
        /* run [pos, pos+run_len) is complete: sample every boundary it covers (:421-428,
         * and :443-450 for the final run) */
        while (pos + run_len > boundary) {
            b->ref_index[bin] = (uint64_t)run_first_byte;
            for (int s = 0; s < ORC_VC_LEN; ++s) b->fm_index[s][bin] = before[s];
            boundary += b->bin_size;
            ++bin;
        }
        if (at_end) break;
        before[run_sym] += run_len; /* :431-437 */
        pos += run_len;
        run_sym = sym;
        run_first_byte = x;
        run_len = (uint64_t)(v >> ORC_LETTER_BITS);
        weight = ORC_NUM_POWER;
    }
    /* :453-457 the last entry is the end of the stream */
    before[run_sym] += run_len;
    b->ref_index[n - 1] = (uint64_t)b->bwt_len;
    for (int s = 0; s < ORC_VC_LEN; ++s) b->fm_index[s][n - 1] = before[s];
    return ORC_OK;
}

/* src/rle_bwt.rs:324-348 */
static int standard_init(orc_rle_bwt *b) {
    calculate_totals(b);
    return construct_fmindex(b);
}

/* src/rle_bwt.rs:59-66 (the reference takes ownership; this copies) */
int orc_rle_load_vector(orc_rle_bwt *b, const uint8_t *bytes, size_t n) {
    free(b->bwt);
    b->bwt = (uint8_t *)malloc(n ? n : 1);
    if (!b->bwt) return ORC_ERR_IO;
    if (n) memcpy(b->bwt, bytes, n);
    b->bwt_len = n;
    return standard_init(b);
}

/* ---- the NumPy v1.0 header.  The reference (src/rle_bwt.rs:115-125) rewrites the
 * python-dict text into JSON with seven textual replacements, parses it with serde_json
 * and takes ["shape"][0].as_u64(); anything serde_json rejects is a panic.  Restated as:
 * the same replacements, then a small strict JSON reader. ---- */

typedef struct {
    const char *p;
    const char *end;
    int depth_shape; /* 1 while inside the top-level "shape" value */
    int have_shape0;
    uint64_t shape0;
} jparse;

static void jskip(jparse *j) {
    while (j->p < j->end && (*j->p == ' ' || *j->p == '\n' || *j->p == '\t' || *j->p == '\r')) ++j->p;
}

static int jvalue(jparse *j, int top_key_is_shape, int array_pos, int depth);

static int jstring(jparse *j, char *out, size_t cap) {
    if (j->p >= j->end || *j->p != '"') return -1;
    ++j->p;
    size_t n = 0;
    while (j->p < j->end && *j->p != '"') {
        char c = *j->p++;
        if (c == '\\') {
            if (j->p >= j->end) return -1;
            c = *j->p++;
        }
        if (out && n + 1 < cap) out[n++] = c;
    }
    if (j->p >= j->end) return -1;
    ++j->p;
    if (out) out[n] = 0;
    return 0;
}

static int jnumber(jparse *j, int *is_u64, uint64_t *val) {
    const char *s = j->p;
    int neg = 0, integral = 1;
    if (j->p < j->end && *j->p == '-') { neg = 1; ++j->p; }
    if (j->p >= j->end || *j->p < '0' || *j->p > '9') return -1;
    uint64_t v = 0;
    while (j->p < j->end && *j->p >= '0' && *j->p <= '9') v = v * 10 + (uint64_t)(*j->p++ - '0');
    if (j->p < j->end && (*j->p == '.' || *j->p == 'e' || *j->p == 'E')) {
        integral = 0;
        while (j->p < j->end && (*j->p == '.' || *j->p == 'e' || *j->p == 'E' || *j->p == '+' ||
                                 *j->p == '-' || (*j->p >= '0' && *j->p <= '9')))
            ++j->p;
    }
    (void)s;
    *is_u64 = integral && !neg;
    *val = v;
    return 0;
}

static int jvalue(jparse *j, int in_shape, int array_pos, int depth) {
    jskip(j);
    if (j->p >= j->end) return -1;
    char c = *j->p;
    if (c == '{') {
        ++j->p;
        jskip(j);
        if (j->p < j->end && *j->p == '}') { ++j->p; return 0; }
        for (;;) {
            char key[64];
            jskip(j);
            if (jstring(j, key, sizeof key)) return -1;
            jskip(j);
            if (j->p >= j->end || *j->p != ':') return -1;
            ++j->p;
            int is_shape = (depth == 0 && strcmp(key, "shape") == 0);
            if (jvalue(j, is_shape, -1, depth + 1)) return -1;
            jskip(j);
            if (j->p < j->end && *j->p == ',') { ++j->p; continue; }
            if (j->p < j->end && *j->p == '}') { ++j->p; return 0; }
            return -1;
        }
    }
    if (c == '[') {
        ++j->p;
        jskip(j);
        if (j->p < j->end && *j->p == ']') { ++j->p; return 0; }
        for (int idx = 0;; ++idx) {
            if (jvalue(j, 0, (in_shape && depth == 1) ? idx : -1, depth + 1)) return -1;
            jskip(j);
            if (j->p < j->end && *j->p == ',') { ++j->p; continue; }
            if (j->p < j->end && *j->p == ']') { ++j->p; return 0; }
            return -1;
        }
    }
    if (c == '"') return jstring(j, NULL, 0);
    if (c == '-' || (c >= '0' && c <= '9')) {
        int is_u64;
        uint64_t v;
        if (jnumber(j, &is_u64, &v)) return -1;
        if (array_pos == 0 && is_u64) { j->have_shape0 = 1; j->shape0 = v; }
        return 0;
    }
    if (j->end - j->p >= 4 && !memcmp(j->p, "true", 4)) { j->p += 4; return 0; }
    if (j->end - j->p >= 5 && !memcmp(j->p, "false", 5)) { j->p += 5; return 0; }
    if (j->end - j->p >= 4 && !memcmp(j->p, "null", 4)) { j->p += 4; return 0; }
    return -1;
}

/* replace every occurrence of `from` by `to` (|to| <= |from|), in place */
static size_t replace_all(char *s, size_t n, const char *from, const char *to) {
    size_t lf = strlen(from), lt = strlen(to), w = 0, r = 0;
    while (r < n) {
        if (r + lf <= n && !memcmp(s + r, from, lf)) {
            memcpy(s + w, to, lt);
            w += lt;
            r += lf;
        } else {
            s[w++] = s[r++];
        }
    }
    return w;
}

static int is_utf8(const uint8_t *s, size_t n) {
    size_t i = 0;
    while (i < n) {
        uint8_t c = s[i];
        size_t need = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : 99;
        if (need == 99 || i + need >= n + (need == 0)) return 0;
        for (size_t k = 1; k <= need; ++k)
            if ((s[i + k] >> 6) != 2) return 0;
        i += need + 1;
    }
    return 1;
}

static int parse_npy_shape0(const uint8_t *hdr, size_t n, uint64_t *shape0) {
    if (!is_utf8(hdr, n)) return ORC_ERR_HEADER; /* String::from_utf8(..).unwrap() */
    char *s = (char *)malloc(n + 1);
    if (!s) return ORC_ERR_IO;
    memcpy(s, hdr, n);
    /* the seven replacements of src/rle_bwt.rs:116-122, in order */
    n = replace_all(s, n, "'", "\"");
    n = replace_all(s, n, "False", "false");
    n = replace_all(s, n, "(", "[");
    n = replace_all(s, n, ")", "]");
    n = replace_all(s, n, ", }", "}");
    n = replace_all(s, n, ", ]", "]");
    n = replace_all(s, n, ",]", "]");
    jparse j = {s, s + n, 0, 0, 0};
    int rc = jvalue(&j, 0, -1, 0);
    if (!rc) {
        jskip(&j);
        if (j.p != j.end) rc = -1; /* trailing garbage: serde_json rejects it */
    }
    free(s);
    if (rc || !j.have_shape0) return ORC_ERR_HEADER;
    *shape0 = j.shape0;
    return ORC_OK;
}

/* src/rle_bwt.rs:81-155.  Magic, version and dtype are NOT checked (as in the reference). */
int orc_rle_load_numpy_file(orc_rle_bwt *b, const char *path) {
    struct stat st;
    if (stat(path, &st)) return ORC_ERR_IO; /* :84 */
    uint64_t file_size = (uint64_t)st.st_size;
    FILE *f = fopen(path, "rb"); /* :88 */
    if (!f) return ORC_ERR_IO;
    uint8_t fixed[10];
    if (fread(fixed, 1, 10, f) != 10) { fclose(f); return ORC_ERR_HEADER; } /* :91-93 panic */
    size_t header_len = (size_t)fixed[8] + 256 * (size_t)fixed[9]; /* :96 */
    size_t skip = 10 + header_len;
    if (skip % 16) skip = (skip / 16 + 1) * 16; /* :98-100 */
    uint8_t *hdr = (uint8_t *)malloc(skip - 10 + 1);
    if (!hdr) { fclose(f); return ORC_ERR_IO; }
    if (fread(hdr, 1, skip - 10, f) != skip - 10) { /* :102-112 read_exact -> UnexpectedEof */
        free(hdr); fclose(f);
        return ORC_ERR_EOF;
    }
    uint64_t expected = 0;
    int rc = parse_npy_shape0(hdr, skip - 10, &expected);
    free(hdr);
    if (rc) { fclose(f); return rc; }
    uint64_t disk = file_size - skip; /* :128 */
    if (expected != disk) { fclose(f); return ORC_ERR_EOF; } /* :129-136 */
    uint8_t *buf = (uint8_t *)malloc(disk ? disk : 1);
    if (!buf) { fclose(f); return ORC_ERR_IO; }
    size_t got = fread(buf, 1, disk, f);
    fclose(f);
    if (got != disk) { free(buf); return ORC_ERR_EOF; } /* :141-148 */
    free(b->bwt);
    b->bwt = buf;
    b->bwt_len = disk;
    return standard_init(b); /* :152 */
}

uint64_t orc_rle_get_symbol_count(const orc_rle_bwt *b, uint8_t s) { return b->symbol_counts[s]; }
uint64_t orc_rle_get_total_size(const orc_rle_bwt *b) { return b->total_size; }
size_t orc_rle_index_length(const orc_rle_bwt *b) { return b->index_length; }
const uint64_t *orc_rle_ref_index(const orc_rle_bwt *b) { return b->ref_index; }
const uint64_t *orc_rle_fm_index(const orc_rle_bwt *b, int sym) { return b->fm_index[sym]; }
const uint64_t *orc_rle_start_index(const orc_rle_bwt *b) { return b->start_index; }
const uint64_t *orc_rle_end_index(const orc_rle_bwt *b) { return b->end_index; }

/* ------------------------------------------------------------------------------------
 * The hot path
 * ---------------------------------------------------------------------------------- */

/* State of the byte scan of src/rle_bwt.rs:216-238.  `pos` symbols are committed, the open
 * run has symbol `run_sym` and `run_len` symbols decoded so far (possibly not all of its
 * digits yet), `acc` is start_index[sym] + occurrences of sym among the committed ones. */
typedef struct {
    size_t byte;
    uint64_t pos, run_len, weight, acc;
    uint8_t run_sym;
} scan_state;

static inline void scan_open(const orc_rle_bwt *b, uint8_t sym, size_t bin, scan_state *s) {
    s->byte = (size_t)b->ref_index[bin]; /* :205 / :251 */
    s->pos = 0;
    for (int x = 0; x < ORC_VC_LEN; ++x) s->pos += b->fm_index[x][bin]; /* :206-209 */
    s->acc = b->start_index[sym] + b->fm_index[sym][bin];                /* :211-214 */
    s->run_sym = 255;
    s->run_len = 0;
    s->weight = 1;
}

/* :221-238 / :264-281: consume bytes while the decoded prefix ends before `target` */
static inline uint64_t scan_to(const orc_rle_bwt *b, uint8_t sym, uint64_t target, scan_state *s) {
    uint64_t consumed = 0;
    while (s->pos + s->run_len < target) {
        uint8_t v = b->bwt[s->byte];
        uint8_t c = v & ORC_MASK;
        if (c == s->run_sym) {
            s->run_len += (uint64_t)(v >> ORC_LETTER_BITS) * s->weight;
            s->weight *= ORC_NUM_POWER;
        } else {
            if (s->run_sym == sym) s->acc += s->run_len;
            s->pos += s->run_len;
            s->run_len = (uint64_t)(v >> ORC_LETTER_BITS);
            s->run_sym = c;
            s->weight = ORC_NUM_POWER;
        }
        ++s->byte;
        ++consumed;
    }
    return consumed;
}

/* src/rle_bwt.rs:202-287.  The reference panics on sym >= 6 (index out of bounds, :212) and
 * on a bin beyond the index; it has no defined result for l > h in the same bin
 * (u64 underflow at :284).  Those become error codes here. */
int orc_rle_constrain_range(const orc_rle_bwt *b, uint8_t sym, uint64_t l, uint64_t h,
                            uint64_t *out_l, uint64_t *out_h, orc_stats *st) {
    if (sym >= ORC_VC_LEN) return ORC_ERR_SYMBOL;
    if (l > h || h > b->total_size) return ORC_ERR_RANGE;
    size_t bin_l = (size_t)(l >> b->bin_power); /* :204 */
    scan_state s;
    scan_open(b, sym, bin_l, &s);
    uint64_t consumed = scan_to(b, sym, l, &s);
    uint64_t visits = 1;
    /* :240-243 the partial run */
    *out_l = s.acc + ((s.run_sym == sym) ? l - s.pos : 0);

    size_t bin_h = (size_t)(h >> b->bin_power); /* :246 */
    if (bin_h != bin_l) {                       /* :250-262 restart from bin_h's sample */
        scan_open(b, sym, bin_h, &s);
        ++visits;
    }                                           /* else :247-249 keep scanning */
    consumed += scan_to(b, sym, h, &s);
    *out_h = s.acc + ((s.run_sym == sym) ? h - s.pos : 0); /* :283-285 */
    if (st) {
        st->steps += 1;
        st->visits += visits;
        st->scan_bytes += consumed;
    }
    return ORC_OK;
}

/* src/msbwt_core.rs:124-161 */
int orc_rle_count_kmer(const orc_rle_bwt *b, const uint8_t *kmer, size_t k, uint64_t *out,
                       orc_stats *st) {
    for (size_t i = 0; i < k; ++i)
        if (kmer[i] >= ORC_VC_LEN) return ORC_ERR_SYMBOL; /* :127 assert -> panic */
    uint64_t l = 0, h = b->total_size;                    /* :128-131 */
    if (st) st->queries += 1;
    for (size_t i = k; i-- > 0;) {                        /* :150 last symbol first */
        if (l == h) { *out = 0; return ORC_OK; }          /* :151-153 */
        uint64_t nl, nh;
        int rc = orc_rle_constrain_range(b, kmer[i], l, h, &nl, &nh, st);
        if (rc) return rc;
        l = nl;
        h = nh;
    }
    *out = h - l;                                         /* :160 */
    return ORC_OK;
}

typedef struct {
    const orc_rle_bwt *b;
    const uint8_t *kmers;
    size_t k, lo, hi;
    uint64_t *out;
    orc_stats st;
    int want_stats; /* 0: the loop runs without the byte counters (the timed CPU baseline) */
    int rc;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *j = (batch_job *)arg;
    for (size_t q = j->lo; q < j->hi; ++q) {
        int rc = orc_rle_count_kmer(j->b, j->kmers + q * j->k, j->k, &j->out[q], j->want_stats ? &j->st : NULL);
        if (rc) { j->rc = rc; j->out[q] = UINT64_MAX; }
    }
    return NULL;
}

int orc_rle_count_kmers(const orc_rle_bwt *b, const uint8_t *kmers, size_t k, size_t n,
                        uint64_t *out, int nthreads, orc_stats *st) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n && n > 0) nthreads = (int)n;
    batch_job *jobs = (batch_job *)calloc((size_t)nthreads, sizeof(batch_job));
    pthread_t *tid = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    if (!jobs || !tid) { free(jobs); free(tid); return ORC_ERR_IO; }
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].b = b; jobs[t].kmers = kmers; jobs[t].k = k; jobs[t].out = out; jobs[t].want_stats = st != NULL;
        jobs[t].lo = n * (size_t)t / (size_t)nthreads;
        jobs[t].hi = n * (size_t)(t + 1) / (size_t)nthreads;
    }
    if (nthreads == 1) {
        batch_worker(&jobs[0]);
    } else {
        for (int t = 0; t < nthreads; ++t) pthread_create(&tid[t], NULL, batch_worker, &jobs[t]);
        for (int t = 0; t < nthreads; ++t) pthread_join(tid[t], NULL);
    }
    int rc = ORC_OK;
    for (int t = 0; t < nthreads; ++t) {
        if (jobs[t].rc) rc = jobs[t].rc;
        if (st) {
            st->queries += jobs[t].st.queries;
            st->steps += jobs[t].st.steps;
            st->visits += jobs[t].st.visits;
            st->scan_bytes += jobs[t].st.scan_bytes;
        }
    }
    free(jobs);
    free(tid);
    return rc;
}

int orc_rle_constrain_ranges(const orc_rle_bwt *b, const uint8_t *syms, const uint64_t *l,
                             const uint64_t *h, size_t n, uint64_t *out_l, uint64_t *out_h) {
    int rc = ORC_OK;
    for (size_t i = 0; i < n; ++i) {
        int r = orc_rle_constrain_range(b, syms[i], l[i], h[i], &out_l[i], &out_h[i], NULL);
        if (r) { rc = r; out_l[i] = out_h[i] = UINT64_MAX; }
    }
    return rc;
}

/* ------------------------------------------------------------------------------------
 * src/bwt_converter.rs
 * ---------------------------------------------------------------------------------- */

static int sym_code(uint8_t ch) {
    switch (ch) {
        case '$': return 0;
        case 'A': return 1;
        case 'C': return 2;
        case 'G': return 3;
        case 'N': return 4;
        case 'T': return 5;
        default: return 255;
    }
}

static size_t emit_run(uint8_t code, uint64_t count, uint8_t *out, size_t cap, size_t w) {
    /* :51-55 base-32 digits, least significant first; zero digits are written */
    while (count > 0) {
        if (out && w < cap) out[w] = (uint8_t)(code | (((uint8_t)count & ORC_COUNT_MASK) << ORC_LETTER_BITS));
        ++w;
        count >>= ORC_NUMBER_BITS;
    }
    return w;
}

/* src/bwt_converter.rs:26-80: '\n' is skipped (also inside a run), any other byte outside
 * "$ACGNT" is a panic -> (size_t)-1 */
size_t orc_convert_to_vec(const uint8_t *ascii, size_t n, uint8_t *out, size_t cap) {
    uint8_t curr = '$';
    uint64_t count = 0;
    size_t w = 0;
    for (size_t i = 0; i < n; ++i) {
        uint8_t ch = ascii[i];
        if (ch == curr) {
            ++count;
        } else if (sym_code(ch) == 255) {
            if (ch != '\n') return (size_t)-1;
        } else {
            w = emit_run((uint8_t)sym_code(curr), count, out, cap, w);
            curr = ch;
            count = 1;
        }
    }
    return emit_run((uint8_t)sym_code(curr), count, out, cap, w);
}

/* the 96-byte header of src/bwt_converter.rs:107-127: placeholder of 95 spaces + '\n',
 * then the dict text is written over its start */
static int write_npy(const char *path, const uint8_t *payload, size_t n) {
    FILE *f = fopen(path, "wb");
    if (!f) return ORC_ERR_IO;
    char head[96];
    memset(head, ' ', 95);
    head[95] = '\n';
    static const char prefix[] = "\x93NUMPY\x01\x00\x56\x00{'descr': '|u1', 'fortran_order': False, 'shape': (";
    char text[96];
    size_t plen = sizeof(prefix) - 1;
    memcpy(text, prefix, plen);
    int m = snprintf(text + plen, sizeof(text) - plen, "%llu, ), }", (unsigned long long)n);
    if (m < 0 || plen + (size_t)m > 95) { fclose(f); return ORC_ERR_HEADER; }
    memcpy(head, text, plen + (size_t)m);
    int ok = fwrite(head, 1, 96, f) == 96 && (n == 0 || fwrite(payload, 1, n, f) == n);
    if (fclose(f)) ok = 0;
    return ok ? ORC_OK : ORC_ERR_IO;
}

int orc_save_bwt_numpy(const uint8_t *bytes, size_t n, const char *path) { /* :102-130 */
    return write_npy(path, bytes, n);
}

int orc_save_bwt_runs_numpy(const uint8_t *syms, const uint64_t *counts, size_t nruns,
                            const char *path) { /* :151-184 */
    size_t need = 0;
    for (size_t i = 0; i < nruns; ++i) need = emit_run(syms[i], counts[i], NULL, 0, need);
    uint8_t *buf = (uint8_t *)malloc(need ? need : 1);
    if (!buf) return ORC_ERR_IO;
    size_t w = 0;
    for (size_t i = 0; i < nruns; ++i) w = emit_run(syms[i], counts[i], buf, need, w);
    int rc = write_npy(path, buf, w);
    free(buf);
    return rc;
}

/* ------------------------------------------------------------------------------------
 * src/string_util.rs
 * ---------------------------------------------------------------------------------- */

void orc_convert_stoi(const uint8_t *ascii, size_t n, uint8_t *out) { /* :15-32 */
    for (size_t i = 0; i < n; ++i) {
        switch (ascii[i]) {
            case '$': out[i] = 0; break;
            case 'A': case 'a': out[i] = 1; break;
            case 'C': case 'c': out[i] = 2; break;
            case 'G': case 'g': out[i] = 3; break;
            case 'T': case 't': out[i] = 5; break;
            default: out[i] = 4; break; /* N, n and every other byte */
        }
    }
}

void orc_convert_itos(const uint8_t *codes, size_t n, uint8_t *out) { /* :6-9 */
    static const char tab[6] = {'$', 'A', 'C', 'G', 'N', 'T'};
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)tab[codes[i]];
}

void orc_reverse_complement_i(const uint8_t *codes, size_t n, uint8_t *out) { /* :12,45-50 */
    static const uint8_t comp[6] = {0, 5, 3, 2, 4, 1};
    for (size_t i = 0; i < n; ++i) out[i] = comp[codes[n - 1 - i]];
}

/* ------------------------------------------------------------------------------------
 * src/bwt_util.rs:154-171 -- sort every rotation of s+"$", doubled so that strings of
 * different length break ties by their owner; byte order ('$' < 'A' < ... < 'T').
 * ---------------------------------------------------------------------------------- */

typedef struct { char *rot; size_t len; } rotation;

static int rot_cmp(const void *a, const void *b) {
    const rotation *x = (const rotation *)a, *y = (const rotation *)b;
    size_t m = x->len < y->len ? x->len : y->len;
    int c = memcmp(x->rot, y->rot, m);
    if (c) return c;
    return (x->len > y->len) - (x->len < y->len);
}

size_t orc_naive_bwt(const char *const *strings, size_t nstr, uint8_t *out) {
    size_t total = 0;
    for (size_t i = 0; i < nstr; ++i) total += strlen(strings[i]) + 1;
    rotation *rots = (rotation *)calloc(total ? total : 1, sizeof(rotation));
    size_t r = 0;
    for (size_t i = 0; i < nstr; ++i) {
        size_t m = strlen(strings[i]) + 1; /* with '$' */
        char *ds = (char *)malloc(m);
        memcpy(ds, strings[i], m - 1);
        ds[m - 1] = '$';
        for (size_t l = 0; l < m; ++l) {
            /* ds[l..] + ds + ds[..l] */
            char *rot = (char *)malloc(2 * m);
            memcpy(rot, ds + l, m - l);
            memcpy(rot + (m - l), ds, m);
            memcpy(rot + (m - l) + m, ds, l);
            rots[r].rot = rot;
            rots[r].len = 2 * m;
            ++r;
        }
        free(ds);
    }
    qsort(rots, total, sizeof(rotation), rot_cmp);
    for (size_t i = 0; i < total; ++i) {
        out[i] = (uint8_t)rots[i].rot[rots[i].len - 1];
        free(rots[i].rot);
    }
    free(rots);
    return total;
}

/* ------------------------------------------------------------------------------------
 * Independent cross-checks
 * ---------------------------------------------------------------------------------- */

uint64_t orc_decompress(const uint8_t *bytes, size_t n, uint8_t *out, uint64_t cap) {
    uint64_t w = 0;
    size_t i = 0;
    while (i < n) {
        uint8_t sym = bytes[i] & ORC_MASK;
        uint64_t len = 0, weight = 1;
        while (i < n && (bytes[i] & ORC_MASK) == sym) {
            len += (uint64_t)(bytes[i] >> ORC_LETTER_BITS) * weight;
            weight *= ORC_NUM_POWER;
            ++i;
        }
        if (out) {
            uint64_t m = (w + len <= cap) ? len : (cap > w ? cap - w : 0);
            memset(out + w, sym, (size_t)m);
        }
        w += len;
    }
    return w;
}

uint64_t orc_rank_bruteforce(const uint8_t *symbols, uint64_t n, uint8_t sym, uint64_t pos) {
    uint64_t c = 0;
    if (pos > n) pos = n;
    for (uint64_t i = 0; i < pos; ++i) c += (symbols[i] == sym);
    return c;
}

/* src/run_block_av_flat.rs:97-125: add every run to its symbol's total until the prefix
 * reaches `position`, then take the overshoot back from the last run's symbol */
uint64_t orc_runblock_count(const uint16_t *runs, size_t nruns, uint64_t position, uint8_t symbol) {
    uint64_t totals[ORC_VC_LEN] = {0, 0, 0, 0, 0, 0};
    uint64_t end = 0;
    unsigned last_sym = 0;
    size_t i = 0;
    while (end < position && i < nruns) {
        last_sym = runs[i] & 0x7u;          /* :50-56 decode_run */
        uint64_t len = runs[i] >> 3;
        totals[last_sym] += len;
        end += len;
        ++i;
    }
    if (end > position) totals[last_sym] -= end - position;
    return totals[symbol];
}
