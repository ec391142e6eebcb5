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
