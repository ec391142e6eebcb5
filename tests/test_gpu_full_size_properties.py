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

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    npy, reads = synth.workload_index("c2")
    b = msbwt.RleBWT()
    b.load_numpy_file(npy)
    return b, reads


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
