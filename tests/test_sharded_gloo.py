"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the shard bounds
and the single all_gather; the per-rank worker is the CPU oracle here (the GPU worker is
covered by the -m gpu tests and bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rust_msbwt_amd as msbwt
from rust_msbwt_amd.sharded import ShardedCounter, shard_bounds, shard_capacity


def test_shard_bounds_cover_everything():
    for n in (0, 1, 2, 7, 64, 100, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) < 32 and max(sizes) <= shard_capacity(n, world)
            # every shard starts on a 16-query boundary: 16-byte aligned rows for any k (tiled kernel)
            assert all(a % 16 == 0 or a == n for a, _ in spans)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from rle_random import random_kmers, random_stream
        ref = orc.OracleRleBWT()
        ref.load_vector(random_stream(3, 5000, "short"))

        def cpu_worker(kmers):
            return torch.from_numpy(ref.count_kmers(kmers.numpy()).view(np.int64).copy())

        counter = ShardedCounter(count_local=cpu_worker)
        assert counter.world == world and counter.rank == rank
        q = random_kmers(9, n, k, alphabet=(0, 1, 2, 3, 4, 5))
        got = counter.count_kmers(torch.from_numpy(q))
        exp = ref.count_kmers(q)
        ok = np.array_equal(msbwt.sharded.as_u64(got), exp)
        # the pipelined form (pieces counted while earlier pieces travel): same counts for every cut and wire type
        for pieces in (1, 3, 4, 7):
            for wire in ("int16", "int32", "int64"):
                ok = ok and np.array_equal(msbwt.sharded.as_u64(counter.count_kmers_pipelined(torch.from_numpy(q), pieces=pieces, wire=wire)), exp)
        # a strided input (column slice of a wider matrix, a transposed matrix): the worker must see a dense copy
        wide = torch.from_numpy(np.concatenate([q, q[:, ::-1]], axis=1))

        def dense_worker(kmers):
            assert kmers.is_contiguous()
            return cpu_worker(kmers)

        strided = ShardedCounter(count_local=dense_worker)
        ok = ok and np.array_equal(msbwt.sharded.as_u64(strided.count_kmers(wide[:, :k])), exp)
        ok = ok and np.array_equal(msbwt.sharded.as_u64(strided.count_kmers(torch.from_numpy(np.ascontiguousarray(q.T)).t())), exp)
        # counts that need the int32 and the int64 wire formats (1-mers of a 5000-run stream are large)
        big = np.array([[1], [2], [3], [5], [0]], dtype=np.uint8)
        wide_worker_calls = []

        def fake_worker(kmers):  # exact but artificial magnitudes: 3, 70000, 2^40
            vals = {1: 3, 2: 70000, 3: 1 << 40, 5: 5, 0: 0}
            wide_worker_calls.append(len(kmers))
            return torch.tensor([vals[int(x[0])] for x in kmers.numpy()], dtype=torch.int64)

        for sel, want in (([0, 3], [3, 5]), ([0, 1, 3], [3, 70000, 5]), ([0, 1, 2, 3, 4], [3, 70000, 1 << 40, 5, 0])):
            c2 = ShardedCounter(count_local=fake_worker)
            out = c2.count_kmers(torch.from_numpy(big[sel]))
            ok = ok and out.tolist() == want
            # (a count beyond the pipelined form's fixed wire type: counted again the plain way, still exact)
            ok = ok and c2.count_kmers_pipelined(torch.from_numpy(big[sel]), pieces=2, wire="int16").tolist() == want
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1001), (2, 64), (3, 10), (2, 1)])
def test_sharded_count_matches_single_process(world, n):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, 6, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_device_generated_batch_is_the_same_whichever_rank_generates_which_rows():
    """bench.py's 1e9-query line at N > 1: the batch of random k-mers is defined chunk by chunk, a rank generates only the
    rows of its shard -- every split must reproduce the rows of the whole (CPU generator here, same code path)."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cpu")
    n, k, chunk = 10_000, 21, 1_024
    whole = bench.device_random_kmers(torch, dev, 0, n, k, 99, chunk=chunk)
    assert whole.shape == (n, k) and set(np.unique(whole.numpy())) <= {1, 2, 3, 5}
    for world in (2, 3, 8):
        parts = []
        for rank in range(world):
            lo, hi = shard_bounds(n, world, rank)
            parts.append(bench.device_random_kmers(torch, dev, lo, hi, k, 99, chunk=chunk))
        assert torch.equal(torch.cat(parts), whole), world
    assert not torch.equal(bench.device_random_kmers(torch, dev, 0, n, k, 100, chunk=chunk), whole)   # the seed matters
