/* Minimal C host for the C ABI: load a comp_msbwt.npy and count the k-mers given on the
 * command line (what README.md:62-70 of the reference does in Rust).
 *
 *   gcc -Iinclude examples/count_kmers.c -Lrust-msbwt_amd -lmsbwt_hip -Wl,-rpath,$PWD/rust-msbwt_amd -o count_kmers
 *   ./count_kmers tests/golden/two_string.npy ACGT TGCA CCCC
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "msbwt_hip.h"

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s comp_msbwt.npy KMER [KMER...]   (all k-mers of one length)\n", argv[0]);
        return 2;
    }
    msbwt_rle *bwt = msbwt_rle_new(8); /* RleBWT::new() */
    if (!bwt) return 1;
    int rc = msbwt_rle_load_numpy_file(bwt, argv[1]);
    if (rc != MSBWT_OK) {
        fprintf(stderr, "load failed (%d): %s\n", rc, msbwt_rle_last_error(bwt));
        msbwt_rle_free(bwt);
        return 1;
    }
    const size_t n = (size_t)(argc - 2), k = strlen(argv[2]);
    uint8_t *codes = (uint8_t *)malloc(n * k ? n * k : 1);
    uint64_t *counts = (uint64_t *)calloc(n, sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) {
        if (strlen(argv[i + 2]) != k) {
            fprintf(stderr, "all k-mers must have the same length\n");
            return 2;
        }
        msbwt_convert_stoi((const uint8_t *)argv[i + 2], k, codes + i * k); /* string_util::convert_stoi */
    }
    rc = msbwt_rle_count_kmers(bwt, codes, k, n, counts); /* one launch for the whole batch */
    if (rc != MSBWT_OK) {
        fprintf(stderr, "count failed (%d): %s\n", rc, msbwt_rle_last_error(bwt));
        return 1;
    }
    printf("total symbols: %llu\n", (unsigned long long)msbwt_rle_get_total_size(bwt));
    for (size_t i = 0; i < n; ++i) printf("%s\t%llu\n", argv[i + 2], (unsigned long long)counts[i]);
    free(codes);
    free(counts);
    msbwt_rle_free(bwt);
    return 0;
}
