"""GPU parity at the sizes of the BASELINE.json configs (not only on toy genomes): the HIP path
through the C ABI against the CPU oracle on large samples, plus whole-batch equality of the two
search kernels.  Needs an MI355X and a few minutes (the C4 index alone takes about a minute of
host time to build)."""
import os
import sys

import numpy as np
import pytest

import rust_msbwt_amd as msbwt
from rust_msbwt_amd import RleBWT
from oracle import oracle as orc
from conftest import ROOT

pytestmark = pytest.mark.gpu

NCPU = min(os.cpu_count() or 1, 16)


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch, torch.device("cuda", 0)


def _count_matrix(torch, dev, bwt, d_q):
    n, k = d_q.shape
    out = torch.empty(n, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    bwt.count_kmers_device(d_q.data_ptr(), k, n, out.data_ptr(), stream)
    bwt.device_status(stream)
    return out


def _revcomp(codes):
    comp = np.array([0, 5, 3, 2, 4, 1], dtype=np.uint8)  # string_util.rs:12
    return np.ascontiguousarray(comp[codes[:, ::-1]])


def test_c3_full_all_read_kmers_fused_both_strands():
    """BASELINE configs[2] at full size: ALL 31-mers of ALL 1 547 217 reads (1.857e8 windows), forward
    and reverse-complemented, prepared in-kernel; 1e6 sampled windows per strand against the oracle."""
    import synth
    torch, dev = _torch()
    npy, reads = synth.workload_index("c3")
    k = 31
    nread, rlen = reads.shape
    wins = rlen - k + 1
    bwt = RleBWT(device=0)
    bwt.load_numpy_file(npy)
    ref = orc.OracleRleBWT()
    ref.load_numpy_file(npy)
    assert bwt.get_total_size() == ref.get_total_size() == nread * (rlen + 1)
    d_reads = torch.from_numpy(reads).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    outs = {}
    for mode in ("lanes", "groups"):
        bwt.set_search_kernel(mode)
        fwd = torch.empty(nread * wins, dtype=torch.int64, device=dev)
        rc = torch.empty(nread * wins, dtype=torch.int64, device=dev)
        bwt.count_read_kmers_device(d_reads.data_ptr(), rlen, nread, k, False, fwd.data_ptr(), rc.data_ptr(), stream)
        bwt.device_status(stream)
        outs[mode] = (fwd, rc)
    # the two search kernels agree on every one of the 3.7e8 counts
    assert torch.equal(outs["lanes"][0], outs["groups"][0]) and torch.equal(outs["lanes"][1], outs["groups"][1])
    fwd, rc = outs["lanes"]
    assert int(fwd.min()) >= 1  # every window occurs at least in its own read
    rng = np.random.default_rng(31)
    ids = np.sort(rng.choice(nread * wins, size=1_000_000, replace=False))
    windows = np.ascontiguousarray(reads[(ids // wins)[:, None], (ids % wins)[:, None] + np.arange(k)[None, :]])
    d_ids = torch.from_numpy(ids).to(dev)
    exp_f = ref.count_kmers(windows, nthreads=NCPU)
    exp_r = ref.count_kmers(_revcomp(windows), nthreads=NCPU)
    assert np.array_equal(fwd[d_ids].cpu().numpy().astype(np.uint64), exp_f)
    assert np.array_equal(rc[d_ids].cpu().numpy().astype(np.uint64), exp_r)
    # the explicit n x k matrix path gives the same counts for the sampled windows
    got_m = _count_matrix(torch, dev, bwt, torch.from_numpy(windows).to(dev))
    assert np.array_equal(got_m.cpu().numpy().astype(np.uint64), exp_f)


def test_c4_real_bwt_random_and_read_derived_31mers():
    """BASELINE configs[3]'s index and batch at full size on one GPU: the MSBWT of 12 888 833 reads
    (1.95e9 symbols, built here in the reference's ordering), 1e8 random 31-mers (1e6 sampled against
    the oracle, whole batch lanes == groups) and 2e7 read-derived 31-mers."""
    import synth
    torch, dev = _torch()
    npy, reads = synth.workload_index("c4")
    k = 31
    bwt = RleBWT(device=0)
    bwt.load_numpy_file(npy)
    ref = orc.OracleRleBWT()
    ref.load_numpy_file(npy)
    total = ref.get_total_size()
    assert bwt.get_total_size() == total == reads.shape[0] * (reads.shape[1] + 1)
    assert [bwt.get_symbol_count(s) for s in range(6)] == [ref.get_symbol_count(s) for s in range(6)]
    rng = np.random.default_rng(41)
    for name, q in (("random", synth.random_kmers(100_000_000, k, synth.CONFIGS["c4"]["qseed"])),
                    ("reads", synth.read_kmers(reads, k, limit=20_000_000, seed=7))):
        d_q = torch.from_numpy(q).to(dev)
        bwt.set_search_kernel("lanes")
        a = _count_matrix(torch, dev, bwt, d_q)
        bwt.set_search_kernel("groups")
        b = _count_matrix(torch, dev, bwt, d_q)
        assert torch.equal(a, b), name
        ids = np.sort(rng.choice(len(q), size=1_000_000, replace=False))
        exp = ref.count_kmers(q[ids], nthreads=NCPU)
        assert np.array_equal(a[torch.from_numpy(ids).to(dev)].cpu().numpy().astype(np.uint64), exp), name
        if name == "reads":
            assert int(a.min()) >= 1
        del d_q, a, b
    # Round 6: the TWO-TIER form of the sparse table on this read set (0.5 % substitutions: 2.4e8 distinct 23-mers, 3.7 x its genome's, most of
    # them error k-mers that occur once).  Solid, once-only and absent 31-mers -- read windows, and read windows with one symbol changed --
    # count n / 1 / 0 exactly as with the complete table and as the oracle says; the table holds entries only for the suffixes that occur
    # at least twice.
    bwt.set_search_kernel("lanes")
    q = synth.read_kmers(reads, k, limit=12_000_000, seed=9)
    mut = q[:4_000_000].copy()
    mut[np.arange(len(mut)), rng.integers(0, k, size=len(mut))] = np.array([1, 2, 3, 5], dtype=np.uint8)[rng.integers(0, 4, size=len(mut))]
    q = np.ascontiguousarray(np.concatenate([q, mut]))
    d_q = torch.from_numpy(q).to(dev)
    complete = _count_matrix(torch, dev, bwt, d_q)
    info0 = bwt.sparse_table_info()
    assert info0["depth"] == 23 and not info0["two_tier"]
    bwt.set_sparse_tiers(1)
    info1 = bwt.sparse_table_info()
    assert bwt.get_sparse_tiers() and info1["depth"] == 23 and info1["two_tier"]
    assert info1["entries"] + info1["filtered"] == info0["entries"] == info1["distinct"][23] and info1["filtered"] == info1["once"][23] > info1["entries"]
    bwt.set_search_counters(True)
    tiered = _count_matrix(torch, dev, bwt, d_q)
    cnt = bwt.search_counters(torch.cuda.current_stream(dev).cuda_stream)
    bwt.set_search_counters(False)
    assert torch.equal(tiered, complete)
    assert cnt["tier_fallbacks"] > 0.05 * len(q)      # the once-only suffixes (and a few false positives) took the direct table's path
    ids = np.sort(rng.choice(len(q), size=1_000_000, replace=False))
    exp = ref.count_kmers(q[ids], nthreads=NCPU)
    assert np.array_equal(tiered[torch.from_numpy(ids).to(dev)].cpu().numpy().astype(np.uint64), exp)
    assert (exp == 0).sum() > 50_000 and (exp == 1).sum() > 50_000 and (exp > 1).sum() > 500_000
    # k between the direct table's depth and the sparse table's, and beyond 32
    for kk in (17, 21, 22, 23, 24, 40):
        qs = np.ascontiguousarray(synth.read_kmers(reads, kk, limit=1_000_000, seed=kk))
        assert np.array_equal(_count_matrix(torch, dev, bwt, torch.from_numpy(qs).to(dev)).cpu().numpy().astype(np.uint64), ref.count_kmers(qs, nthreads=NCPU)), kk
    bwt.set_sparse_tiers(-1)
    assert not bwt.get_sparse_tiers() and torch.equal(_count_matrix(torch, dev, bwt, d_q), complete)


def test_stream_beyond_2_pow_33_symbols_hbm_regime():
    """A 9e9-symbol synthetic RLE stream (total > 2^33: positions and counts need more than 32 bits, the
    index is far larger than the Infinity Cache): present (LF-walk) and absent 31-mers and batched
    constrain_range at l, h > 2^32, all against the oracle on the same stream."""
    import synth
    sys.path.insert(0, ROOT)
    import bench
    torch, dev = _torch()
    rle, total = synth.rle_stream(9_000_000_000, 6.0, 123)
    assert total > 2**33
    bwt = RleBWT(device=0)
    bwt.load_vector(rle)
    ref = orc.OracleRleBWT()
    ref.load_vector(rle)
    assert bwt.get_total_size() == total == ref.get_total_size()
    k = 31
    present = bench.walk_kmers(torch, np, bwt, dev, total, 2_000_000, k, 99)
    absent = bench.device_random_kmers(torch, dev, 0, 2_000_000, k, 98)
    d_q = torch.cat([present, absent])
    q = d_q.cpu().numpy()
    exp = ref.count_kmers(q, nthreads=NCPU)
    for mode in ("lanes", "groups"):
        bwt.set_search_kernel(mode)
        got = _count_matrix(torch, dev, bwt, d_q).cpu().numpy().astype(np.uint64)
        assert np.array_equal(got, exp), mode
    assert exp[:2_000_000].min() >= 1 and (exp[2_000_000:] == 0).mean() > 0.9
    # batched constrain_range with both bounds beyond 2^32
    rng = np.random.default_rng(17)
    n = 1_000_000
    a = rng.integers(2**32, total + 1, size=n, dtype=np.uint64)
    b = a + rng.integers(0, 5000, size=n).astype(np.uint64) * rng.integers(0, 2, size=n).astype(np.uint64)
    b = np.minimum(b, np.uint64(total))
    syms = rng.integers(0, 6, size=n).astype(np.uint8)
    gl, gh = bwt.constrain_ranges(syms, a, b)
    ol, oh = ref.constrain_ranges(syms, a, b)
    assert np.array_equal(gl, ol) and np.array_equal(gh, oh)
    assert int(gl.max()) > 2**32


def test_repeat_genome_read_set_built_on_the_gpu_and_counted():
    """bench.py --genome repeats in small: a 4e6-bp genome with the human repeat classes (synth.repeat_genome), 30x error-free reads,
    its exact MSBWT built by synth/bwt_reads.py msbwt_rle_repeats ON THE GPU -- the same bytes as the CPU build of the same tensors --
    loaded, and read-derived / walked / random k-mers counted against the oracle: high-copy k-mers (counts in the thousands), wide
    sparse-table entries in the side array, second lines at every step."""
    from synth import bwt_reads
    torch, dev = _torch()
    L = 150
    genome, cnt = bwt_reads.repeat_read_set(4_000_000, L, 30.0, 5)
    rle_cpu, totals, n_reads = bwt_reads.msbwt_rle_repeats(genome, cnt, L)
    rle, totals_gpu, n_gpu = bwt_reads.msbwt_rle_repeats(genome.to(dev), cnt.to(dev), L)
    assert n_gpu == n_reads and np.array_equal(totals, totals_gpu) and np.array_equal(rle, rle_cpu)
    total = int(totals.sum())
    bwt = RleBWT(device=0)
    bwt.load_vector(rle)
    ref = orc.OracleRleBWT()
    ref.load_vector(rle)
    assert bwt.get_total_size() == ref.get_total_size() == total == n_reads * (L + 1)
    assert bwt.get_sparse_table() >= 19
    info = bwt.sparse_table_info()
    assert info["side_entries"] > 0          # suffixes of high-copy repeats: 255 or more wide
    sys.path.insert(0, ROOT)
    import bench
    for k in (23, 31, 59):
        walked = bench.walk_kmers(torch, np, bwt, dev, total, 300_000, k, 11 + k)      # drawn like read windows: repeats over-represented
        d_q = torch.cat([walked, bench.device_random_kmers(torch, dev, 0, 50_000, k, 3 + k)])
        exp = ref.count_kmers(d_q.cpu().numpy(), nthreads=NCPU)
        assert exp[:300_000].min() >= 1 and exp.max() > 1000
        for mode in ("lanes", "groups"):
            bwt.set_search_kernel(mode)
            got = _count_matrix(torch, dev, bwt, d_q).cpu().numpy().astype(np.uint64)
            assert np.array_equal(got, exp), (k, mode)
    bwt.set_search_kernel("auto")
    # a whole read as the query: as many occurrences as reads spell it
    reads = bwt_reads.reads_of(genome[:200_000 + L], cnt[:200_000], L)[:20_000]
    assert np.array_equal(bwt.count_kmers(reads), ref.count_kmers(reads, nthreads=NCPU))


def test_human_scale_9e10_symbols_against_the_oracle():
    """The size and the index the metric is quoted on: the exact MSBWT of 5.96e8 error-free reads, 9e10 symbols (positions
    beyond 2^36, 40-bit header counts with high bytes up to 20, the full index: sparse suffix table of depth 23 -- and, with it switched off, the 238 GB index of
    rounds 2-4 with its depth-17 packed table --, overlapping pair blocks addressed at human scale): 2e6 present (LF-walk) +
    2e6 random 31-mers under both search kernels, and 1e6 batched constrain_range calls with l, h > 2^36, all
    against the oracle on the same stream.  Needs ~250 GB of HBM and ~20 GB of host memory; about two minutes."""
    import synth
    sys.path.insert(0, ROOT)
    import bench
    torch, dev = _torch()
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 250 * 10**9:
        pytest.skip("needs 250 GB of free HBM")
    # bench.py's default index: the exact MSBWT of a 30x error-free read set, built on the GPU (synth/bwt_reads.py)
    from synth import bwt_reads
    genome, cnt = bwt_reads.read_set(int(bench.HUMAN_SYMBOLS * 150 / (30 * 151)), 150, 30.0, 77, device=dev)
    rle, totals, n_reads = bwt_reads.msbwt_rle(genome, cnt, 150)
    del genome, cnt
    torch.cuda.empty_cache()
    total = int(totals.sum())
    assert total == n_reads * 151 and abs(total - bench.HUMAN_SYMBOLS) < 1e-3 * bench.HUMAN_SYMBOLS and total > 2**36
    bwt = RleBWT(device=0)
    bwt.load_vector(rle)
    # the default index since round 5: sparse suffix table of depth 23 (3e9 distinct 23-mers, ~42 GB), a small direct table beside it
    assert bwt.get_total_size() == total and bwt.get_pair_index() and bwt.get_sparse_table() == 23 and bwt.get_table_depth() == 15
    assert bwt.get_line_streaming()   # 120 GB of pair blocks: their lines are fetched with the non-temporal hint
    info = bwt.sparse_table_info()
    assert 2.9e9 < info["entries"] < 3.1e9 and info["entries"] == info["distinct"][23] and info["bytes"] < 50e9, info
    # a real 30x BWT: present k-mers keep ranges ~ coverage wide, and the loader answers with overlapping pair blocks
    assert bwt.get_typical_range_width() >= 20 and bwt.get_pair_stride() == 96
    ref = orc.OracleRleBWT()
    ref.load_vector(rle)
    assert ref.get_total_size() == total
    assert [bwt.get_symbol_count(s) for s in range(6)] == [ref.get_symbol_count(s) for s in range(6)]
    k = 31
    present = bench.walk_kmers(torch, np, bwt, dev, total, 2_000_000, k, 4242)
    absent = bench.device_random_kmers(torch, dev, 0, 2_000_000, k, 98)
    d_q = torch.cat([present, absent])
    q = d_q.cpu().numpy()
    exp = ref.count_kmers(q, nthreads=NCPU)
    assert exp[:2_000_000].min() >= 1
    for mode in ("lanes", "groups"):
        bwt.set_search_kernel(mode)
        got = _count_matrix(torch, dev, bwt, d_q).cpu().numpy().astype(np.uint64)
        assert np.array_equal(got, exp), mode
    # round 6: k between the direct table's depth and the sparse table's.  With k undeclared a SECOND sparse level (the 17-symbol suffixes,
    # 39 GB) serves 17 <= k < 23 -- the packed depth-17 direct table does not fit beside the first -- and k = 16 takes the direct table
    assert info["second_depth"] == 17 and 35e9 < info["second_bytes"] < 45e9 and bwt.device_bytes() < 260e9, info
    bwt.set_search_kernel("lanes")
    for kk in (16, 17, 19, 21, 22, 23, 24):
        short = torch.cat([present[:1_000_000, k - kk:], absent[:200_000, k - kk:]]).contiguous()
        got = _count_matrix(torch, dev, bwt, short).cpu().numpy().astype(np.uint64)
        assert np.array_equal(got, ref.count_kmers(short.cpu().numpy(), nthreads=NCPU)), kk
        assert got[:1_000_000].min() >= 1
    # without the sparse table the loader builds what rounds 2-4 measured: the depth-17 packed direct table (73 GB, 238 GB in all)
    bwt.set_sparse_table(0)
    assert bwt.get_sparse_table() == 0 and bwt.get_table_depth() == 17 and bwt.device_bytes() > 230e9
    bwt.set_search_kernel("lanes")
    got = _count_matrix(torch, dev, bwt, d_q).cpu().numpy().astype(np.uint64)
    assert np.array_equal(got, exp)
    bwt.set_sparse_table(-1)
    assert bwt.get_sparse_table() == 23 and bwt.get_table_depth() == 15
    # k > 32 takes the other instantiation of the lanes kernel (6 words of symbols): 59-mers, fmlrc's long k
    bwt.set_search_kernel("auto")
    long_q = bench.walk_kmers(torch, np, bwt, dev, total, 500_000, 59, 7)
    got = _count_matrix(torch, dev, bwt, long_q).cpu().numpy().astype(np.uint64)
    assert np.array_equal(got, ref.count_kmers(long_q.cpu().numpy(), nthreads=NCPU)) and got.min() >= 1
    # batched constrain_range with both bounds beyond 2^36
    rng = np.random.default_rng(23)
    n = 1_000_000
    a = rng.integers(2**36, total + 1, size=n, dtype=np.uint64)
    b = a + rng.integers(0, 5000, size=n).astype(np.uint64) * rng.integers(0, 2, size=n).astype(np.uint64)
    b = np.minimum(b, np.uint64(total))
    syms = rng.integers(0, 6, size=n).astype(np.uint8)
    gl, gh = bwt.constrain_ranges(syms, a, b)
    ol, oh = ref.constrain_ranges(syms, a, b)
    assert np.array_equal(gl, ol) and np.array_equal(gh, oh)
    assert int(gl.max()) > 2**36
