/* One rank of a one-process-per-GPU job, as a plain C host (the shape an MPI-style Rust host would have): load the
 * index on this rank's GPU, count this rank's shard of a k-mer batch with the device entry point, and leave every rank
 * with ALL counts through the library's RCCL all-gather (msbwt_rle_allgather_counts) -- the path's one exchange step
 * (queries are `&self`, src/msbwt_core.rs:125).  The RCCL id travels through a file here; MPI_Bcast or an
 * environment variable would do as well.
 *
 *   gcc -std=gnu11 -Iinclude -I/opt/rocm/include examples/rank_gather.c -Lrust-msbwt_amd -lmsbwt_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/rust-msbwt_amd -Wl,-rpath,/opt/rocm/lib -o rank_gather
 *   ./rank_gather comp_msbwt.npy RANK NRANKS /tmp/msbwt.id ACGT TGCA CCCC AAAA      (start one per GPU)
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "msbwt_hip.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        if ((call) != hipSuccess) { fprintf(stderr, "HIP error at %s\n", #call); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s comp_msbwt.npy RANK NRANKS ID_FILE KMER [KMER...]   (all k-mers of one length)\n", argv[0]);
        return 2;
    }
    const int rank = atoi(argv[2]), nranks = atoi(argv[3]);
    const size_t n = (size_t)(argc - 5), k = strlen(argv[5]);
    int ndev = 0;
    CHECK(hipGetDeviceCount(&ndev));
    CHECK(hipSetDevice(rank % (ndev > 0 ? ndev : 1)));
    /* the communicator: rank 0 makes the id, everyone reads it */
    unsigned char id[MSBWT_COMM_ID_BYTES];
    if (rank == 0) {
        if (msbwt_comm_get_unique_id(id) != MSBWT_OK) { fprintf(stderr, "no RCCL\n"); return 1; }
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", argv[4]);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 1;
        fclose(f);
        rename(tmp, argv[4]);
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 600 && !(f = fopen(argv[4], "rb")); ++tries) usleep(100000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) return 1;
        fclose(f);
    }
    void *comm = NULL;
    if (msbwt_comm_init_rank(&comm, nranks, id, rank) != MSBWT_OK) return 1;

    msbwt_rle *bwt = msbwt_rle_new_on_device(8, -1);
    if (!bwt || msbwt_rle_load_numpy_file(bwt, argv[1]) != MSBWT_OK) {
        fprintf(stderr, "load failed: %s\n", bwt ? msbwt_rle_last_error(bwt) : "out of memory");
        return 1;
    }
    /* the whole batch is known to every rank; this rank counts rows [lo, hi) -- shards start at multiples of 16 queries */
    uint8_t *codes = (uint8_t *)malloc(n * k);
    for (size_t i = 0; i < n; ++i) msbwt_convert_stoi((const uint8_t *)argv[i + 5], k, codes + i * k);
    const size_t units = (n + 15) / 16, per = (units + (size_t)nranks - 1) / (size_t)nranks * 16; /* = n_mine of every rank (padded) */
    size_t lo = (size_t)rank * per, hi = lo + per;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
    void *d_kmers = NULL, *d_mine = NULL, *d_all = NULL;
    CHECK(hipMalloc(&d_kmers, n * k + 16));
    CHECK(hipMalloc(&d_mine, per * sizeof(uint64_t)));
    CHECK(hipMalloc(&d_all, per * (size_t)nranks * sizeof(uint64_t)));
    CHECK(hipMemset(d_mine, 0, per * sizeof(uint64_t)));
    CHECK(hipMemcpy(d_kmers, codes, n * k, hipMemcpyHostToDevice));
    if (hi > lo && msbwt_rle_count_kmers_device(bwt, (const uint8_t *)d_kmers + lo * k, k, hi - lo, d_mine, NULL) != MSBWT_OK) return 1;
    if (msbwt_rle_allgather_counts(bwt, comm, d_mine, per, d_all, 16, NULL) != MSBWT_OK) {
        fprintf(stderr, "gather failed: %s\n", msbwt_rle_last_error(bwt));
        return 1;
    }
    int rc = msbwt_rle_device_status(bwt, NULL); /* synchronises; MSBWT_ERR_OVERFLOW = a count needs more than 16 bits */
    if (rc == MSBWT_ERR_OVERFLOW) {
        if (msbwt_rle_allgather_counts(bwt, comm, d_mine, per, d_all, 64, NULL) != MSBWT_OK) return 1;
        rc = msbwt_rle_device_status(bwt, NULL);
    }
    if (rc != MSBWT_OK) { fprintf(stderr, "count failed (%d): %s\n", rc, msbwt_rle_last_error(bwt)); return 1; }
    uint64_t *all = (uint64_t *)malloc(per * (size_t)nranks * sizeof(uint64_t));
    CHECK(hipMemcpy(all, d_all, per * (size_t)nranks * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) printf("rank %d: %s\t%llu\n", rank, argv[i + 5], (unsigned long long)all[i]); /* shard r sits at r * per */
    msbwt_comm_destroy(comm);
    msbwt_rle_free(bwt);
    return 0;
}
