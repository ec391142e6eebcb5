#!/usr/bin/env python3
"""Randomised parity soak (GPU): many small random indexes x settings x query shapes against the
CPU oracle.  Not part of the default test suite; run as  python tools/stress_parity.py [seconds]."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rust_msbwt_amd as msbwt  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from rle_random import random_kmers, random_stream, raw_byte_stream  # noqa: E402
import synth  # noqa: E402


def real_bwt(rng):
    n, length = int(rng.integers(5, 400)), int(rng.integers(20, 120))
    g = synth.genome(int(rng.integers(length + 5, 3000)), int(rng.integers(1, 1 << 30)))
    rd = synth.reads(g, n, length, int(rng.integers(1, 1 << 30)), float(rng.choice([0.0, 0.01, 0.1])))
    if rng.random() < 0.3:                      # sprinkle N
        rd = rd.copy()
        rd[rng.random(rd.shape) < 0.01] = 4
    return synth.rle_encode(synth.build_msbwt_symbols(rd)), rd


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
    t0 = time.time()
    rounds = checks = 0
    last_report = t0
    while time.time() - t0 < budget:
        if time.time() - last_report > 60:  # a silent GPU job is taken for a hung one
            print("... %d indexes, %d query checks after %.0fs" % (rounds, checks, time.time() - t0), flush=True)
            last_report = time.time()
        kind = rng.choice(["bwt", "ones", "short", "long", "mixed", "raw"])
        reads = None
        if kind == "bwt":
            rle, reads = real_bwt(rng)
        elif kind == "raw":
            rle = raw_byte_stream(int(rng.integers(0, 1 << 30)), int(rng.integers(1, 3000)))
        else:
            rle = random_stream(int(rng.integers(0, 1 << 30)), int(rng.integers(1, 30000)), kind)
        o = orc.OracleRleBWT(int(rng.integers(1, 10)))
        o.load_vector(rle)
        total = o.get_total_size()
        b = msbwt.RleBWT()
        b.set_block_format("runs" if rng.random() < 0.25 else "planes")   # the memory-lean format now and then
        # (run blocks build their sparse table at load time only -- round 6 -- so its depth and form are drawn before the load)
        b.set_sparse_table(int(rng.choice([-1, -1, 0, 16, 17, 19, 20, 23, 25, 28])))
        b.set_sparse_tiers(int(rng.choice([-1, 0, 1, 1])))
        b.load_vector(rle)
        assert b.get_total_size() == total
        for _ in range(3):
            depth = int(rng.integers(0, 9))
            b.set_table_depth(depth)
            b.set_pair_stride(int(rng.choice([0, 96, 128])))
            b.set_pair_index(int(rng.integers(0, 2)))
            b.set_presence_filter(int(rng.integers(0, 2)))
            b.set_table_packed(int(rng.integers(-1, 2)))           # packed table when a pair index exists
            b.set_search_kernel(str(rng.choice(["auto", "groups", "lanes", "lanes"])))
            b.set_table_side(int(rng.integers(0, 2)))              # round 4: escape lines from the side array, or restarted
            b.set_batch_order(int(rng.choice([-1, 0, 1, 1])))      # ... the library's own ordering pass forced on half the time
            b.set_sparse_table(int(rng.choice([-1, -1, 0, 16, 17, 18, 19, 20, 23, 25, 27, 28, 30, 31])))   # round 5: sparse suffix table: automatic, off, or a depth
            b.set_sparse_tiers(int(rng.choice([-1, 0, 1, 1])))     # round 6: the two-tier form of the sparse table (filter -> direct table) forced on half the time
            b.set_line_streaming(int(rng.choice([-1, 0, 1, 1])))   # ... kernel revision 3: the non-temporal line loads forced on half the time
            if rng.random() < 0.15:                                # ... and a memory budget now and then (rebuilds the optional structures)
                b.set_memory_budget(int(b.device_bytes() * float(rng.choice([0.2, 0.5, 0.9]))) + 1)
            elif b.get_memory_budget():
                b.set_memory_budget(0)
            k = int(rng.integers(1, 72)) if rng.random() < 0.6 else int(rng.integers(16, 40))   # (often at and above the sparse table's depths)
            # mostly small batches, sometimes many tiles per wave (ring refill, setup running ahead)
            n = int(rng.integers(1, 700)) if rng.random() < 0.8 else int(rng.integers(5000, 60000))
            qs = [random_kmers(int(rng.integers(0, 1 << 30)), n, k),
                  random_kmers(int(rng.integers(0, 1 << 30)), n // 3 + 1, k, alphabet=(0, 1, 2, 3, 4, 5))]
            if reads is not None and reads.shape[1] >= k:
                qs.append(synth.read_kmers(reads, k, limit=n, seed=int(rng.integers(0, 1 << 30))))
            q = np.concatenate(qs)
            got, exp = b.count_kmers(q), o.count_kmers(q)
            assert np.array_equal(got, exp), (kind, depth, k, len(rle))
            checks += len(q)
            if k <= 64:                                            # round 4: the same k-mers as 2-bit words (ACGT rows only)
                acgt = ~np.isin(q, (0, 4)).any(axis=1)
                if acgt.any():
                    words = msbwt.rle_bwt.pack_2bit(q[acgt])
                    bits = int(rng.choice([64, 32]))
                    if bits == 64 or int(exp[acgt].max()) < 2 ** 32:
                        assert np.array_equal(b.count_kmers_packed(words, k, count_bits=bits).astype(np.uint64), exp[acgt]), (kind, depth, k, "packed")
                        checks += int(acgt.sum())
            # the trait's own shape: single calls and tiny batches (kernel-argument / mailbox path, polled completion)
            for i in rng.integers(0, len(q), size=4):
                assert b.count_kmer(q[int(i)]) == int(exp[int(i)]), (kind, depth, k)
            small = int(rng.integers(2, 65))
            assert np.array_equal(b.count_kmers(q[:small]), exp[:small]), (kind, depth, k, small)
            r1 = b.constrain_range(int(rng.integers(0, 6)), msbwt.BWTRange(0, total))
            # constrain_ranges around block / pair-block / superblock borders
            m = 300
            base = rng.integers(0, total + 1, size=m)
            snap = (base // 128) * 128 + rng.integers(-2, 3, size=m)
            pos = np.clip(np.where(rng.random(m) < 0.5, base, snap), 0, total)
            l, h = np.minimum(pos, pos[::-1]).astype(np.uint64), np.maximum(pos, pos[::-1]).astype(np.uint64)
            sy = rng.integers(0, 6, size=m).astype(np.uint8)
            gl, gh = b.constrain_ranges(sy, l, h)
            ol, oh = o.constrain_ranges(sy, l, h)
            assert np.array_equal(gl, ol) and np.array_equal(gh, oh), (kind, len(rle))
            if reads is not None and 1 <= k <= 64 and reads.shape[1] >= k:
                sub = reads[: min(len(reads), 40)]
                f, r = b.count_read_kmers(sub, k, ascii=False, revcomp=True)
                w = sub.shape[1] - k + 1
                wins = np.lib.stride_tricks.sliding_window_view(sub, k, axis=1).reshape(-1, k)
                assert np.array_equal(f.reshape(-1), o.count_kmers(wins))
                rcw = np.array([orc.reverse_complement_i(x) for x in wins], dtype=np.uint8)
                assert np.array_equal(r.reshape(-1), o.count_kmers(rcw))
                checks += 2 * len(wins)
                # one strand only (a tile is 64 windows then), ASCII input
                ascii_reads = np.frombuffer(b"$ACGNT", dtype=np.uint8)[sub]
                f1, _ = b.count_read_kmers(ascii_reads, k, ascii=True, revcomp=False)
                assert np.array_equal(f1, f)
                _, r1 = b.count_read_kmers(sub, k, ascii=False, forward=False, revcomp=True)
                assert np.array_equal(r1, r)
        rounds += 1
    print("stress ok: %d indexes, %d query checks in %.0fs" % (rounds, checks, time.time() - t0))


if __name__ == "__main__":
    main()
