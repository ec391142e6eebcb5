"""The oracle's byte scan against an independent decompress-and-count, on random RLE streams
that force multi-byte runs, zero digits, runs spanning many bins and every bin_power."""
import numpy as np
import pytest

from oracle import oracle as orc
from rle_random import random_stream, raw_byte_stream, runs_to_bytes


def _check(bytes_, bin_powers, seed, npos=300):
    plain = orc.decompress(bytes_)
    total = len(plain)
    rng = np.random.default_rng(seed)
    pref = np.zeros((6, total + 1), dtype=np.int64)
    for s in range(6):
        pref[s, 1:] = np.cumsum(plain == s)
    for bp in bin_powers:
        b = orc.OracleRleBWT(bp)
        b.load_vector(bytes_)
        assert b.get_total_size() == total
        start = b.start_index()
        assert [b.get_symbol_count(s) for s in range(6)] == [int(pref[s, total]) for s in range(6)]
        pos = np.unique(np.concatenate([rng.integers(0, total + 1, size=npos), [0, total],
                                        np.arange(0, total + 1, max(1, 1 << bp))[:50]]))
        l = rng.choice(pos, size=npos)
        h = rng.choice(pos, size=npos)
        l, h = np.minimum(l, h), np.maximum(l, h)
        syms = rng.integers(0, 6, size=npos).astype(np.uint8)
        ol, oh = b.constrain_ranges(syms, l.astype(np.uint64), h.astype(np.uint64))
        exp_l = np.array([start[s] + pref[s, p] for s, p in zip(syms, l)], dtype=np.uint64)
        exp_h = np.array([start[s] + pref[s, p] for s, p in zip(syms, h)], dtype=np.uint64)
        assert np.array_equal(ol, exp_l)
        assert np.array_equal(oh, exp_h)


@pytest.mark.parametrize("kind", ["ones", "short", "long", "mixed"])
def test_random_runs(kind):
    _check(random_stream(7, 400, kind), [1, 3, 8, 12], seed=11)


def test_raw_byte_streams():
    for seed in range(5):
        _check(raw_byte_stream(seed, 300), [2, 8], seed=seed)


def test_exact_multiples_and_edges():
    # total an exact multiple of the bin size; single huge run; leading '$' / non-'$'
    for syms, lens in [([1, 2], [256, 256]), ([0, 1], [1, 255]), ([3], [100000]), ([0], [512]),
                       ([5, 0, 5], [32, 1024, 32768])]:
        _check(runs_to_bytes(syms, lens), [4, 8], seed=3, npos=200)


def test_brute_force_rank_helper():
    plain = np.array([0, 1, 1, 2, 1], dtype=np.uint8)
    assert orc.rank_bruteforce(plain, 1, 0) == 0
    assert orc.rank_bruteforce(plain, 1, 3) == 2
    assert orc.rank_bruteforce(plain, 1, 5) == 3


def test_bin_power_8_multi_byte_runs_one_million_ranges():
    """The regime every large parity test leans on, pinned on the oracle itself: bin_power 8 (the reference's default,
    rle_bwt.rs:297), runs of two and three RLE bytes (lengths 32..1023 and 1024..32767, zero digits included), runs that
    span many 256-symbol bins -- the G6 property (rle_bwt.rs:603-675: constrain(sym, [l, h)) = start + rank at both ends)
    for 10^6 random (sym, l, h) against decompress-and-count, a sample of them against the structurally different
    RLEBlock::count restatement (run_block_av_flat.rs:97-125), and the G7 property (rle_bwt.rs:677-710) -- count_kmer --
    against a backward search written with nothing but prefix sums."""
    rng = np.random.default_rng(2024)
    nruns = 1500
    first = rng.integers(0, 6)
    syms = ((first + np.concatenate([[0], np.cumsum(rng.integers(1, 6, size=nruns - 1))])) % 6).astype(np.uint8)
    pick = rng.integers(0, 10, size=nruns)
    lens = np.where(pick < 4, rng.integers(32, 1024, size=nruns),                    # two bytes
                    np.where(pick < 7, rng.integers(1024, 32768, size=nruns),        # three bytes
                             np.where(pick < 8, rng.choice([32, 64, 1024, 2048, 32 * 33, 1024 * 31], size=nruns),  # zero digits
                                      rng.integers(1, 32, size=nruns)))).astype(np.uint64)   # one byte
    bytes_ = runs_to_bytes(syms, lens)
    same = (bytes_[1:] & 7) == (bytes_[:-1] & 7)
    assert same.sum() > nruns // 2                      # multi-byte runs dominate
    plain = orc.decompress(bytes_)
    total = len(plain)
    assert total == int(lens.sum()) and total > 2 ** 22
    pref = np.zeros((6, total + 1), dtype=np.int32)
    for s in range(6):
        np.cumsum(plain == s, out=pref[s, 1:])
    b = orc.OracleRleBWT(8)
    b.load_vector(bytes_)
    start = np.array(b.start_index(), dtype=np.int64)
    assert b.get_total_size() == total and [b.get_symbol_count(s) for s in range(6)] == [int(pref[s, total]) for s in range(6)]
    n = 1_000_000
    l = rng.integers(0, total + 1, size=n)
    h = rng.integers(0, total + 1, size=n)
    edge = rng.integers(0, 20, size=n)
    l = np.where(edge == 0, (l >> 8) << 8, l)           # bin boundaries
    h = np.where(edge == 1, np.minimum(((h >> 8) << 8) + 255, total), h)
    l, h = np.minimum(l, h), np.maximum(l, h)
    l[:3], h[:3] = [0, 0, total], [0, total, total]
    sy = rng.integers(0, 6, size=n).astype(np.uint8)
    ol, oh = b.constrain_ranges(sy, l.astype(np.uint64), h.astype(np.uint64))
    assert np.array_equal(ol.astype(np.int64), start[sy] + pref[sy, l])
    assert np.array_equal(oh.astype(np.int64), start[sy] + pref[sy, h])
    # the same answers from RLEBlock::count over the stream cut into 13-bit runs (a second algorithm)
    pieces_s, pieces_l = [], []
    for s, ln in zip(syms.tolist(), lens.tolist()):
        while ln > 0:
            take = min(ln, 8191)
            pieces_s.append(s)
            pieces_l.append(take)
            ln -= take
    runs16 = (np.array(pieces_s, dtype=np.uint16) | (np.array(pieces_l, dtype=np.uint16) << 3)).astype(np.uint16)
    for i in rng.integers(0, n, size=4000):
        assert int(ol[i]) == int(start[sy[i]]) + orc.runblock_count(runs16, int(l[i]), int(sy[i]))
        assert int(oh[i]) == int(start[sy[i]]) + orc.runblock_count(runs16, int(h[i]), int(sy[i]))
    # G7: count_kmer against a prefix-sum backward search; every 1-mer count is its symbol total
    for s in range(6):
        assert b.count_kmer([s]) == int(pref[s, total])
    for k, sticky in ((3, 0.0), (9, 0.0), (12, 0.85)):   # sticky: mostly homopolymer stretches, which long runs make frequent
        qs = rng.integers(0, 6, size=(100_000, k)).astype(np.uint8)
        for j in range(1, k):
            qs[:, j] = np.where(rng.random(len(qs)) < sticky, qs[:, j - 1], qs[:, j])
        lo = np.zeros(len(qs), dtype=np.int64)
        hi = np.full(len(qs), total, dtype=np.int64)
        for j in range(k - 1, -1, -1):
            c = qs[:, j]
            lo, hi = start[c] + pref[c, lo], start[c] + pref[c, hi]
        assert np.array_equal(b.count_kmers(qs).astype(np.int64), hi - lo), k
        assert k == 9 or (hi - lo > 0).sum() > 1000, k
