/* Latency of the trait-shaped calls through the C ABI: the shape of the reference's own benchmark loop
 * (benches/ds_benchmarks.rs:96-99: one count_kmer per iteration).  Loads a comp_msbwt.npy, draws random
 * ACGT k-mers and times msbwt_rle_count_kmer / msbwt_rle_constrain_range one call at a time, then
 * msbwt_rle_count_kmers for growing batch sizes (the break-even against a CPU loop is where microseconds
 * per query drop below the CPU's).
 *
 *   gcc -O2 -Iinclude examples/call_latency.c -Lrust-msbwt_amd -lmsbwt_hip -Wl,-rpath,$PWD/rust-msbwt_amd -o call_latency
 *   ./call_latency synth/cache/c2_..._comp_msbwt.npy 21
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "msbwt_hip.h"

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e6 + (double)t.tv_nsec * 1e-3;
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s comp_msbwt.npy K [calls]\n", argv[0]);
        return 2;
    }
    const size_t k = (size_t)atoi(argv[2]);
    const size_t calls = argc > 3 ? (size_t)atol(argv[3]) : 20000;
    msbwt_rle *bwt = msbwt_rle_new(8);
    if (!bwt || msbwt_rle_load_numpy_file(bwt, argv[1]) != MSBWT_OK) {
        fprintf(stderr, "load failed: %s\n", bwt ? msbwt_rle_last_error(bwt) : "out of memory");
        return 1;
    }
    const size_t maxn = 1u << 16;
    uint8_t *q = (uint8_t *)malloc(maxn * k);
    uint64_t *out = (uint64_t *)calloc(maxn, sizeof(uint64_t));
    static const uint8_t base[4] = {1, 2, 3, 5};
    uint64_t st = 88172645463325252ull;
    for (size_t i = 0; i < maxn * k; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        q[i] = base[st & 3u];
    }
    uint64_t sink = 0, c = 0, ol = 0, oh = 0;
    for (size_t i = 0; i < 200; ++i) msbwt_rle_count_kmer(bwt, q + (i % maxn) * k, k, &c); /* warm-up */
    double t0 = now_us();
    for (size_t i = 0; i < calls; ++i) {
        if (msbwt_rle_count_kmer(bwt, q + (i % maxn) * k, k, &c) != MSBWT_OK) return 1;
        sink += c;
    }
    double t1 = now_us();
    printf("msbwt_rle_count_kmer      k=%zu: %7.2f us per call (%zu calls)\n", k, (t1 - t0) / (double)calls, calls);
    const uint64_t total = msbwt_rle_get_total_size(bwt);
    t0 = now_us();
    for (size_t i = 0; i < calls; ++i) {
        if (msbwt_rle_constrain_range(bwt, base[i & 3u], (i * 7919u) % (total / 2 + 1), total / 2 + (i * 104729u) % (total / 2 + 1), &ol, &oh) != MSBWT_OK) return 1;
        sink += oh - ol;
    }
    t1 = now_us();
    printf("msbwt_rle_constrain_range      : %7.2f us per call\n", (t1 - t0) / (double)calls);
    for (size_t n = 1; n <= maxn; n *= 4) {
        const size_t reps = n <= 64 ? 4000 : (n <= 4096 ? 400 : 40);
        msbwt_rle_count_kmers(bwt, q, k, n, out);
        t0 = now_us();
        for (size_t r = 0; r < reps; ++r)
            if (msbwt_rle_count_kmers(bwt, q, k, n, out) != MSBWT_OK) return 1;
        t1 = now_us();
        printf("msbwt_rle_count_kmers n=%6zu: %8.2f us per call, %8.4f us per query\n", n, (t1 - t0) / (double)reps, (t1 - t0) / (double)reps / (double)n);
    }
    printf("(checksum %llu)\n", (unsigned long long)(sink + out[0]));
    msbwt_rle_free(bwt);
    free(q);
    free(out);
    return 0;
}
