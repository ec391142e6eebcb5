"""Size-independent properties at BASELINE.json's full C2 size (1.01e8 symbols, 10 M queries),
checked on the GPU alone -- the oracle is too slow to sweep these volumes:

* every position of the (cyclic) text starts exactly one j-mer  =>  sum over ALL 6^j j-mers == T;
* every occurrence of w is preceded, and followed, by exactly one symbol
      =>  count(w) == sum_c count(c w) == sum_c count(w c)   over c in $ACGNT;
* determinism: two passes over the full 10 M-query batch are identical, and the batch's checksum
  agrees between the matrix entry point and the fused read-window entry point.
"""
import itertools

import numpy as np
import pytest

import rust_msbwt_amd as msbwt
import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    npy, reads = synth.workload_index("c2")
    b = msbwt.RleBWT()
    b.load_numpy_file(npy)
    b.npy = npy
    return b, reads


def test_c2_batch_against_the_oracle(c2):
    """configs[1] on its own inputs: the full 10 M random 21-mer batch is counted on the GPU and 10^5 of its queries, sampled
    uniformly, are checked against the oracle on the same comp_msbwt.npy -- plus 10^5 read-derived 21-mers (present ones:
    random 21-mers on a 3.3 Mbp genome almost all count 0)."""
    b, reads = c2
    ref = orc.OracleRleBWT(8)
    ref.load_numpy_file(b.npy)
    q = synth.random_kmers(10_000_000, 21, 3)          # the C2 batch itself (synth.CONFIGS["c2"]: qseed 3)
    got = b.count_kmers(q)
    ids = np.sort(np.random.default_rng(21).choice(len(q), size=100_000, replace=False))
    assert np.array_equal(got[ids], ref.count_kmers(np.ascontiguousarray(q[ids]), nthreads=8))
    rd = synth.read_kmers(reads, 21, limit=100_000, seed=22)
    got_rd = b.count_kmers(rd)
    assert np.array_equal(got_rd, ref.count_kmers(rd, nthreads=8))
    assert got_rd.min() >= 1 and int((got[ids] > 0).sum()) < 1000


def test_all_jmers_partition_the_text(c2):
    b, _ = c2
    total = b.get_total_size()
    assert total == 1_000_000 * 101
    for j in (1, 2, 3, 4):
        kmers = np.array(list(itertools.product(range(6), repeat=j)), dtype=np.uint8)
        assert int(b.count_kmers(kmers).sum()) == total, j


def test_extension_identities(c2):
    b, reads = c2
    rng = np.random.default_rng(0)
    for k in (5, 12, 20, 31, 40):
        base = np.concatenate([synth.read_kmers(reads, k, limit=150_000, seed=k), synth.random_kmers(50_000, k, k)])
        base_counts = b.count_kmers(base).astype(np.int64)
        left = np.zeros_like(base_counts)
        right = np.zeros_like(base_counts)
        for c in range(6):
            col = np.full((len(base), 1), c, dtype=np.uint8)
            left += b.count_kmers(np.hstack([col, base])).astype(np.int64)
            right += b.count_kmers(np.hstack([base, col])).astype(np.int64)
        assert np.array_equal(left, base_counts), k
        assert np.array_equal(right, base_counts), k
        assert base_counts[:150_000].min() >= 1


def test_full_batch_is_deterministic_and_matches_fused_path(c2):
    b, reads = c2
    q = synth.random_kmers(10_000_000, 21, 3)          # the C2 batch itself
    first = b.count_kmers(q)
    assert np.array_equal(first, b.count_kmers(q))
    b.set_pair_index(0)
    b.set_table_depth(9)
    assert np.array_equal(first, b.count_kmers(q))     # settings never change results
    b.set_table_depth(-1 if False else 13)
    b.set_pair_index(1)
    # fused read windows vs the same windows as an explicit matrix (200 k reads x 80 windows)
    sub = reads[:200_000]
    fwd, _ = b.count_read_kmers(sub, 21, ascii=False)
    wins = np.lib.stride_tricks.sliding_window_view(sub, 21, axis=1).reshape(-1, 21)
    assert np.array_equal(fwd.reshape(-1), b.count_kmers(np.ascontiguousarray(wins)))
