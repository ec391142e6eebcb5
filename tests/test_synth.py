"""The synthetic-workload generators: the fast MSBWT builder must reproduce the reference's
ordering (checked against the oracle's naive_bwt restatement), generators are deterministic."""
import numpy as np
import pytest

import synth
from oracle import oracle as orc


def _random_reads(rng, n, lo, hi, alphabet="ACGT", genome_len=200):
    genome = "".join(rng.choice(list(alphabet), size=genome_len))
    out = []
    for _ in range(n):
        length = int(rng.integers(lo, hi + 1))
        p = int(rng.integers(0, genome_len - length + 1))
        out.append(genome[p:p + length])
    return out


@pytest.mark.parametrize("seed", range(6))
def test_builder_matches_naive_bwt(seed):
    rng = np.random.default_rng(seed)
    alphabet = "ACGT" if seed % 2 == 0 else "ACGNT"
    reads = _random_reads(rng, 60, 1, 40, alphabet)
    reads += reads[:5]                                  # exact duplicates
    reads += ["A", "AA", "AAA", "ACA", "CA"]            # bwt_util.rs:201-236 tie-break cases
    sym = synth.build_msbwt_symbols([orc.convert_stoi(r) for r in reads], threads=3)
    assert orc.convert_itos(sym) == orc.naive_bwt(reads)


def test_builder_long_shared_prefixes():
    # suffixes equal for far more than the 21 packed symbols
    base = "ACGTTGCA" * 12
    reads = [base[i:i + 60] for i in range(0, 30, 3)] + [base[:70], base[:70], "T" * 50, "T" * 49]
    sym = synth.build_msbwt_symbols([orc.convert_stoi(r) for r in reads], threads=2)
    assert orc.convert_itos(sym) == orc.naive_bwt(reads)


def test_builder_reference_vectors(golden):
    for case in golden["G3_naive_bwt"]["cases"]:
        sym = synth.build_msbwt_symbols([orc.convert_stoi(r) for r in case["strings"]])
        assert orc.convert_itos(sym) == case["bwt"]


def test_fixed_length_reads_array_and_rle():
    g = synth.genome(3000, 5)
    assert set(np.unique(g)) <= {1, 2, 3, 5}
    rd = synth.reads(g, 300, 50, 6, 0.01)
    assert rd.shape == (300, 50)
    sym = synth.build_msbwt_symbols(rd)
    assert orc.convert_itos(sym) == orc.naive_bwt([orc.convert_itos(r) for r in rd])
    rle = synth.rle_encode(sym)
    assert np.array_equal(rle, orc.convert_to_vec(orc.convert_itos(sym)))
    assert np.array_equal(orc.decompress(rle), sym)


def test_generators_are_deterministic():
    assert np.array_equal(synth.genome(1000, 1), synth.genome(1000, 1))
    assert not np.array_equal(synth.genome(1000, 1), synth.genome(1000, 2))
    g = synth.genome(5000, 1)
    assert np.array_equal(synth.reads(g, 50, 100, 2), synth.reads(g, 50, 100, 2))
    q = synth.random_kmers(1000, 31, 3)
    assert np.array_equal(q, synth.random_kmers(1000, 31, 3))
    assert abs((q == 1).mean() - 0.25) < 0.02
    # error rate is about what was asked for
    clean = synth.reads(g, 2000, 100, 9, 0.0)
    noisy = synth.reads(g, 2000, 100, 9, 0.05)
    assert 0.03 < (clean != noisy).mean() < 0.07


def test_read_kmers_and_npy(tmp_path):
    g = synth.genome(2000, 3)
    rd = synth.reads(g, 20, 40, 4, 0.0)
    allk = synth.read_kmers(rd, 31)
    assert allk.shape == (20 * 10, 31)
    assert np.array_equal(allk[10], rd[1, 0:31])
    some = synth.read_kmers(rd, 31, limit=50, seed=1)
    assert some.shape == (50, 31)
    rle = synth.rle_encode(synth.build_msbwt_symbols(rd))
    a, b = str(tmp_path / "a.npy"), str(tmp_path / "b.npy")
    synth.write_npy(a, rle)
    orc.save_bwt_numpy(rle, b)
    assert open(a, "rb").read() == open(b, "rb").read()
    o = orc.OracleRleBWT()
    o.load_numpy_file(a)
    assert o.count_kmers(allk).min() >= 1           # every read k-mer is present


def test_rle_stream_is_well_formed():
    rle, total = synth.rle_stream(200000, 6.0, 7)
    plain = orc.decompress(rle)
    assert len(plain) == total == 200000
    runs = 1 + int((plain[1:] != plain[:-1]).sum())
    assert 4.0 < total / runs < 8.0


def test_workload_index_small_scale(tmp_path, monkeypatch):
    """The bench's config builder at 0.1 % scale: the cached .npy loads in the oracle and every
    read k-mer is present in it."""
    monkeypatch.setattr(synth, "CACHE", str(tmp_path))
    npy, rd = synth.workload_index("c2", scale=0.001)
    assert rd.shape == (1000, 100)
    o = orc.OracleRleBWT()
    o.load_numpy_file(npy)
    assert o.get_total_size() == 1000 * 101
    assert o.get_symbol_count(0) == 1000
    q = synth.read_kmers(rd, 21, limit=500, seed=1)
    assert o.count_kmers(q).min() >= 1
    npy2, rd2 = synth.workload_index("c2", scale=0.001)      # second call: served from the cache
    assert npy2 == npy and np.array_equal(rd, rd2)


@pytest.mark.parametrize("g,L,cov,seed", [(2000, 40, 8, 1), (20000, 60, 20, 2), (120000, 100, 30, 3), (1000, 150, 30, 4)])
def test_exact_bwt_of_a_read_set_without_suffix_sorting_the_reads(g, L, cov, seed):
    """synth/bwt_reads.py (the human-scale index of bench.py) against the suffix-sorting builder on the same reads:
    same symbol totals and the same count for every k-mer tried, k = 1 .. read length (the two streams differ only in
    the order of identical suffixes, which no '$'-free count depends on)."""
    from synth import bwt_reads
    from oracle import oracle as orc
    genome, cnt = bwt_reads.read_set(g, L, cov, seed)
    rle, totals, nreads = bwt_reads.msbwt_rle(genome, cnt, L)
    reads = bwt_reads.reads_of(genome, cnt, L)
    assert reads.shape == (nreads, L)
    true = synth.rle_encode(synth.build_msbwt_symbols(reads))
    a, b = orc.OracleRleBWT(), orc.OracleRleBWT()
    a.load_vector(rle)
    b.load_vector(true)
    assert a.get_total_size() == b.get_total_size() == nreads * (L + 1)
    assert [a.get_symbol_count(s) for s in range(6)] == [b.get_symbol_count(s) for s in range(6)] == [int(t) for t in totals]
    for k in (1, 2, 3, 4, 5, 8, 12, 17, 21, 29, 31, 33, L):
        q = np.concatenate([synth.read_kmers(reads, k, limit=2000, seed=k), synth.random_kmers(1000, k, seed + k)])
        assert np.array_equal(a.count_kmers(q), b.count_kmers(q)), k
    # a read-derived k-mer occurs at least as often as reads cover its window (exactly that, unless the genome repeats it)
    k = 25
    starts = np.repeat(np.arange(g), cnt.numpy().astype(np.int64))
    pos = starts[:200] + 3
    q = np.array([1, 2, 3, 5], dtype=np.uint8)[genome.numpy()[pos[:, None] + np.arange(k)[None, :]]]
    covering = np.array([int(cnt.numpy()[max(0, p + k - L):p + 1].sum()) for p in pos])
    got = a.count_kmers(q)
    assert (got >= covering).all() and (got == covering).mean() > 0.9


def _crafted_repeat_genome(g, L, seed):
    """random bases with exact copies far longer than a read, (AC)n, a homopolymer, an exact tandem array, a copy near the end"""
    rng = np.random.default_rng(seed)
    G = rng.integers(0, 4, size=g + L + 256).astype(np.uint8)
    seg = rng.integers(0, 4, size=700).astype(np.uint8)
    for at in (300, 1500, 2900, 3700):
        G[at:at + 700] = seg
    G[4500:4800] = np.tile(np.array([0, 1], dtype=np.uint8), 150)
    G[4900:4990] = 2
    G[5100:5100 + 37 * 12] = np.tile(rng.integers(0, 4, size=37).astype(np.uint8), 12)
    G[g - 200:g - 20] = seg[:180]
    return G


@pytest.mark.parametrize("kind,g,L,cov,seed", [("families", 40000, 60, 15, 2), ("families", 60000, 150, 20, 4), ("crafted", 6000, 40, 12, 5),
                                               ("crafted", 7000, 150, 30, 6)])
def test_exact_bwt_of_a_read_set_from_a_genome_with_repeats(kind, g, L, cov, seed):
    """synth/bwt_reads.py msbwt_rle_repeats (bench.py --genome repeats: the human-scale index over synth.repeat_genome) against the
    suffix-sorting builder on the same reads: stretches shared beyond 28 bases, beyond the read length, tandem arrays, homopolymers --
    same symbol totals, the same count for every k-mer tried around every depth the builder treats differently."""
    import torch
    from synth import bwt_reads
    from oracle import oracle as orc
    if kind == "families":
        genome, cnt = bwt_reads.repeat_read_set(g, L, cov, seed)
    else:
        genome = torch.from_numpy(_crafted_repeat_genome(g, L, seed))
        cnt = torch.from_numpy(np.random.default_rng(seed).poisson(cov / L, size=g).clip(max=255).astype(np.uint8))
    notes = []
    rle, totals, nreads = bwt_reads.msbwt_rle_repeats(genome, cnt, L, log=notes.append)
    assert any("placed one by one" in m and not m.startswith("0 ") for m in notes)      # the genome did exercise the explicit levels
    reads = bwt_reads.reads_of(genome, cnt, L)
    true = synth.rle_encode(synth.build_msbwt_symbols(reads))
    a, b = orc.OracleRleBWT(), orc.OracleRleBWT()
    a.load_vector(rle)
    b.load_vector(true)
    assert a.get_total_size() == b.get_total_size() == nreads * (L + 1)
    assert [a.get_symbol_count(s) for s in range(6)] == [b.get_symbol_count(s) for s in range(6)] == [int(t) for t in totals]
    for k in (1, 2, 3, 4, 8, 17, 27, 28, 29, 30, 31, 32, 33, 40, 59, 60, 61, 62, 63, 93, 124, 125, L - 1, L):
        if k > L:
            continue
        q = np.concatenate([synth.read_kmers(reads, k, limit=4000, seed=k), synth.random_kmers(300, k, seed + k)])
        assert np.array_equal(a.count_kmers(q), b.count_kmers(q)), k
    # the random-genome builder and this one agree where the former applies
    g2, c2 = bwt_reads.read_set(3000, L, cov, seed)
    g2 = torch.cat([g2, torch.randint(0, 4, (256,), dtype=torch.uint8, generator=torch.Generator().manual_seed(seed + 1000))])
    r1, t1, n1 = bwt_reads.msbwt_rle(g2, c2, L)
    r2, t2, n2 = bwt_reads.msbwt_rle_repeats(g2, c2, L)
    assert n1 == n2 and np.array_equal(t1, t2) and np.array_equal(r1, r2)


def test_repeat_genome_builder_on_random_structures():
    """msbwt_rle_repeats over 30 seeded genomes with random structures -- exact copies, tandem arrays, diverged copies, a low-complexity
    tail -- read lengths 30..155, coverage 3..60: totals and counts (k around every depth the builder treats differently) equal the
    suffix-sorting builder's.  (150 seeds of the same generator ran clean when the builder was written.)"""
    import torch
    from synth import bwt_reads
    from oracle import oracle as orc
    for seed in range(2000, 2030):
        rng = np.random.default_rng(seed)
        L = int(rng.integers(30, 156))
        g = int(rng.integers(L + 50, 6000))
        cov = float(rng.choice([3, 10, 30, 60]))
        G = rng.integers(0, 4, size=g + L + 256).astype(np.uint8)
        for _ in range(int(rng.integers(0, 8))):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                ln = int(rng.integers(20, min(1500, g // 2)))
                a, b = rng.integers(0, g - ln, size=2)
                G[b:b + ln] = G[a:a + ln].copy()
            elif kind == 1:
                unit, n = int(rng.integers(1, 60)), int(rng.integers(2, 40))
                ln = min(unit * n, g // 2)
                a = int(rng.integers(0, g - ln))
                G[a:a + ln] = np.tile(rng.integers(0, 4, size=unit).astype(np.uint8), n)[:ln]
            elif kind == 2:
                ln = int(rng.integers(50, min(800, g // 2)))
                a, b = rng.integers(0, g - ln, size=2)
                seg = G[a:a + ln].copy()
                mut = rng.random(ln) < 0.03
                seg[mut] = rng.integers(0, 4, size=int(mut.sum()))
                G[b:b + ln] = seg
            else:
                ln = int(rng.integers(10, 200))
                G[g - ln:g] = rng.integers(0, 2, size=ln)
        cnt = torch.from_numpy(rng.poisson(cov / L, size=g).clip(max=255).astype(np.uint8))
        if int(cnt.sum()) == 0:
            cnt[0] = 1
        genome = torch.from_numpy(G)
        rle, totals, nreads = bwt_reads.msbwt_rle_repeats(genome, cnt, L)
        reads = bwt_reads.reads_of(genome, cnt, L)
        a, b = orc.OracleRleBWT(), orc.OracleRleBWT()
        a.load_vector(rle)
        b.load_vector(synth.rle_encode(synth.build_msbwt_symbols(reads)))
        assert a.get_total_size() == b.get_total_size() == nreads * (L + 1), seed
        assert [a.get_symbol_count(c) for c in range(6)] == [b.get_symbol_count(c) for c in range(6)], seed
        for k in sorted({1, 3, 15, 27, 28, 29, 30, 31, 32, 61, 62, 63, 93, 94, 124, 125, L - 1, L}):
            if 1 <= k <= L:
                q = np.concatenate([synth.read_kmers(reads, k, limit=2000, seed=k), synth.random_kmers(100, k, seed + k)])
                assert np.array_equal(a.count_kmers(q), b.count_kmers(q)), (seed, L, g, k)


def test_histogram_stream_follows_the_measured_run_lengths():
    """synth.rle_stream(histogram=...): the independent-symbol stand-in draws (symbol, run length) from the committed histogram
    of config C4's real MSBWT (synth/c4_run_histogram.json): exact symbol total, no two neighbouring runs of
    one symbol, run-length distribution and mean as measured."""
    import json
    data = json.load(open(synth.HISTOGRAM_FILE))
    assert data["symbols"] == 1_946_213_783 and abs(data["mean_run"] - data["symbols"] / data["runs"]) < 1e-9
    rle, total = synth.rle_stream(30_000_000, 6.0, 5, histogram=synth.HISTOGRAM_FILE)
    assert total == 30_000_000
    sym = orc.decompress(rle)
    assert len(sym) == total
    hist = synth.run_histogram(rle)
    lengths = np.arange(hist.shape[1], dtype=np.uint64)
    assert int((hist * lengths[None, :]).sum()) == total          # runs decode to the symbol total: neighbours never merged
    mean = total / hist.sum()
    assert abs(mean - data["mean_run"]) < 0.15, mean
    real_a = np.array([data["lengths"]["1"].get(str(l), 0) for l in range(1, 9)], dtype=np.float64) / data["runs_per_symbol"]["1"]
    got_a = hist[1, 1:9] / hist[1].sum()
    assert np.abs(real_a - got_a).max() < 0.01, (real_a, got_a)  # P(run length = 1..8 | symbol A)
    assert hist[4].sum() == 0                                     # C4's reads hold no N
    # deterministic for a given (target, seed)
    again, _ = synth.rle_stream(30_000_000, 6.0, 5, histogram=synth.HISTOGRAM_FILE)
    assert np.array_equal(rle, again)
