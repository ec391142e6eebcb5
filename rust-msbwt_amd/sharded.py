"""Query sharding across the GPUs of one node.

count_kmer calls are independent and read-only (`&self`, src/msbwt_core.rs:125), so the path
shards over queries: every rank holds a replica of the index, takes a contiguous slice of the
batch, and the per-rank counts are exchanged with ONE collective -- an all_gather over RCCL
(xGMI) of `ceil(n/world)` u64 per rank.  No other communication exists on this path.

torch / torch.distributed are plumbing here (device buffers, streams, the process group).
"""
import numpy as np


SHARD_ALIGN = 16  # queries


def shard_bounds(n, world, rank):
    """Contiguous slices, balanced to within two 16-query units, every slice STARTING at a multiple of 16
    queries: row `lo` of an (n, k) byte matrix then sits at a 16-byte-aligned address whatever k
    is, which is what the tiled kernel's staged loads need (an unaligned batch silently takes the
    slow generic kernel -- include/msbwt_hip.h, "alignment")."""
    units = -(-n // SHARD_ALIGN)
    base, extra = divmod(units, world)
    lo_u = rank * base + min(rank, extra)
    hi_u = lo_u + base + (1 if rank < extra else 0)
    return min(n, lo_u * SHARD_ALIGN), min(n, hi_u * SHARD_ALIGN)


def shard_capacity(n, world):
    """Largest shard (the per-rank length of the all_gather buffers)."""
    if world <= 0:
        return n
    units = -(-n // SHARD_ALIGN)
    return min(n, -(-units // world) * SHARD_ALIGN)


class ShardedCounter:
    """Counts a batch that every rank can see, each rank doing its slice, all ranks ending
    up with the whole count vector.

    `count_local(kmers_slice) -> counts` is the per-rank worker; by default it runs the
    rank's `RleBWT` on its GPU (device-pointer entry point).  Tests inject a CPU worker to
    exercise the sharding and the collective under gloo.
    """

    def __init__(self, bwt=None, group=None, count_local=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bwt = bwt
        self.device = device
        self._count_local = count_local or self._count_on_gpu
        if count_local is None and bwt is None:
            raise ValueError("need an RleBWT (GPU worker) or an explicit count_local")

    def _count_on_gpu(self, d_kmers):
        torch = self.torch
        n, k = d_kmers.shape
        out = torch.empty(n, dtype=torch.int64, device=d_kmers.device)
        stream = torch.cuda.current_stream(d_kmers.device).cuda_stream
        self.bwt.count_kmers_device(d_kmers.data_ptr(), k, n, out.data_ptr(), stream)
        return out

    def count_kmers(self, kmers):
        """kmers: (n, k) uint8 tensor on this rank's device (same content on every rank).
        Returns int64[n] on the same device (bit pattern of the u64 counts)."""
        torch, dist = self.torch, self.dist
        n = kmers.shape[0]
        lo, hi = shard_bounds(n, self.world, self.rank)
        # the C ABI takes a dense row-major n x k matrix: a strided view (a column slice, an unfold() window view, a
        # transposed tensor) is copied; a row slice of a contiguous batch is contiguous already -- same storage, and
        # 16-byte aligned when `kmers` is
        mine = self._count_local(kmers[lo:hi].contiguous())
        if self.world == 1:
            return mine
        cap = shard_capacity(n, self.world)
        # 8 bytes per count over xGMI can cost more than computing it: agree on the narrowest
        # integer type that holds every rank's largest count exactly, ship raw bytes (neither
        # RCCL nor gloo has a 16-bit integer type), widen on arrival.
        top = mine.max().reshape(1).to(torch.int64) if hi > lo else torch.zeros(1, dtype=torch.int64, device=kmers.device)
        low = mine.min().reshape(1).to(torch.int64) if hi > lo else torch.zeros(1, dtype=torch.int64, device=kmers.device)
        span = torch.cat([top, -low])
        dist.all_reduce(span, op=dist.ReduceOp.MAX, group=self.group)
        biggest, smallest = int(span[0]), -int(span[1])
        wire = torch.int64
        if smallest >= 0:  # u64 counts >= 2^63 look negative as int64: keep them wide
            wire = torch.int16 if biggest < (1 << 15) else torch.int32 if biggest < (1 << 31) else torch.int64
        padded = torch.zeros(cap, dtype=wire, device=kmers.device)
        padded[:hi - lo] = mine.to(wire)
        narrow = torch.empty(cap * self.world, dtype=wire, device=kmers.device)
        dist.all_gather_into_tensor(narrow.view(torch.uint8), padded.view(torch.uint8), group=self.group)
        gathered = narrow.to(torch.int64)
        # drop the padding of the short shards
        pieces = []
        for r in range(self.world):
            a, b = shard_bounds(n, self.world, r)
            pieces.append(gathered[r * cap:r * cap + (b - a)])
        return torch.cat(pieces)

    def count_kmers_pipelined(self, kmers, pieces=4, wire="int32"):
        """The same result as count_kmers, as a PIPELINE for a caller with one batch: this rank's shard is counted in `pieces`
        pieces, and the all_gather of piece i (asynchronous) runs while piece i + 1 is counted.  The wire type is fixed beforehand
        (it cannot wait for the largest count): "int16" / "int32" / "int64"; should a count not fit it, the batch is counted again
        the plain way -- the result is always exact.  (The library's own form over RCCL: msbwt_rle_count_kmers_allgather_device.)"""
        torch, dist = self.torch, self.dist
        n = kmers.shape[0]
        lo, hi = shard_bounds(n, self.world, self.rank)
        if self.world == 1:
            return self._count_local(kmers[lo:hi].contiguous())
        cap = shard_capacity(n, self.world)
        wire_t = {"int16": torch.int16, "int32": torch.int32, "int64": torch.int64}[wire]
        limit = torch.iinfo(wire_t).max
        per = max(SHARD_ALIGN, -(-cap // pieces // SHARD_ALIGN) * SHARD_ALIGN)   # the same cut on every rank: whole 16-query units
        works, parts, fits = [], [], True
        for off in range(0, cap, per):
            length = min(per, cap - off)
            a, b = min(hi, lo + off), min(hi, lo + off + length)
            send = torch.zeros(length, dtype=wire_t, device=kmers.device)
            if b > a:
                mine = self._count_local(kmers[a:b].contiguous())
                fits = fits and bool(((mine >= 0) & (mine <= limit)).all())
                send[:b - a] = mine.to(wire_t)
            recv = torch.empty(length * self.world, dtype=wire_t, device=kmers.device)
            works.append(dist.all_gather_into_tensor(recv.view(torch.uint8), send.view(torch.uint8), group=self.group, async_op=True))
            parts.append((off, length, recv))
        for w in works:
            w.wait()
        ok = torch.tensor([1 if fits else 0], dtype=torch.int32, device=kmers.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok[0]) == 0:
            return self.count_kmers(kmers)
        out = torch.empty(n, dtype=torch.int64, device=kmers.device)
        for r in range(self.world):
            a, b = shard_bounds(n, self.world, r)
            for off, length, recv in parts:
                take = max(0, min(length, (b - a) - off))
                if take:
                    out[a + off:a + off + take] = recv[r * length:r * length + take].to(torch.int64)
        return out


def as_u64(t):
    """int64 tensor holding u64 bit patterns -> numpy uint64."""
    return t.detach().cpu().numpy().view(np.uint64)
