"""CPU-only checks of the product's host side: the C-ABI library loads and exports every
symbol include/msbwt_hip.h declares, the codecs match the reference's vectors, and the
load-path layout builder produces blocks that decode back to the BWT.  No GPU compute."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import rust_msbwt_amd as msbwt
from conftest import ROOT, expand_case
from oracle import oracle as orc
from rle_random import random_stream, raw_byte_stream, runs_to_bytes

_lib = msbwt._lib


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "msbwt_hip.h")).read()
    declared = set(re.findall(r"\b(msbwt_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    L = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), "libmsbwt_hip.so lacks %s" % name
    # and the Python binding covers exactly the declared set
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert "gfx950" in msbwt.version()


def test_library_embeds_gfx950_code_object():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_codecs_match_reference_vectors(golden, tmp_path):
    for case in golden["G1_convert_to_vec"]["cases"]:
        got = msbwt.bwt_converter.convert_to_vec(expand_case(case))
        if "bytes" in case:
            assert got.tolist() == case["bytes"]
        else:
            assert len(got) == case["len"]
    with pytest.raises(ValueError):
        msbwt.bwt_converter.convert_to_vec("ACGX")
    g = golden["string_util"]
    for text, codes in g["stoi"]:
        assert msbwt.string_util.convert_stoi(text).tolist() == codes
    for codes, text in g["itos"]:
        assert msbwt.string_util.convert_itos(codes) == text
    for codes, rc in g["revcomp"]:
        assert msbwt.string_util.reverse_complement_i(codes).tolist() == rc
    # .npy writer: byte-identical with the oracle's (which is pinned by G2)
    g2 = golden["G2_npy"]
    for i, case in enumerate(g2["cases"]):
        a, b = str(tmp_path / ("a%d.npy" % i)), str(tmp_path / ("b%d.npy" % i))
        if case["kind"] == "bytes":
            rle = msbwt.bwt_converter.convert_to_vec(expand_case(case))
            msbwt.bwt_converter.save_bwt_numpy(rle, a)
            orc.save_bwt_numpy(rle, b)
        else:
            msbwt.bwt_converter.save_bwt_runs_numpy(case["runs"], a)
            orc.save_bwt_runs_numpy(case["runs"], b)
        assert open(a, "rb").read() == open(b, "rb").read()
        assert list(open(a, "rb").read()[96:]) == case["payload"]


def test_codec_agrees_with_oracle_on_random_text():
    rng = np.random.default_rng(5)
    text = "".join(rng.choice(list("$ACGNT\n"), p=[.05, .3, .2, .2, .05, .15, .05], size=5000))
    assert np.array_equal(msbwt.bwt_converter.convert_to_vec(text), orc.convert_to_vec(text))
    raw = bytes(rng.integers(0, 256, size=2000, dtype=np.uint8))
    assert np.array_equal(msbwt.string_util.convert_stoi(raw), orc.convert_stoi(raw))


def build_blocks(rle):
    rle = np.ascontiguousarray(rle, dtype=np.uint8)
    total = C.c_uint64()
    L = _lib.lib()
    n = L.msbwt_build_plane_blocks(rle.ctypes.data_as(C.c_void_p), rle.size, None, 0, C.byref(total))
    assert n != _lib.SIZE_MAX
    out = np.zeros((n, 8, 4), dtype=np.uint32)
    L.msbwt_build_plane_blocks(rle.ctypes.data_as(C.c_void_p), rle.size, out.ctypes.data_as(C.c_void_p), n, C.byref(total))
    return out, int(total.value)


def decode_blocks(blocks, total):
    """plane blocks -> (symbols, A) with A[b, s] = the 40-bit bound of block b."""
    bits = np.arange(32, dtype=np.uint32)
    planes = [((blocks[:, :, p][:, :, None] >> bits) & 1).reshape(len(blocks), 256) for p in range(3)]
    sym = (planes[0] | (planes[1] << 1) | (planes[2] << 2)).astype(np.uint8).reshape(-1)
    meta = blocks[:, :, 3].astype(np.uint64)
    A = np.zeros((len(blocks), 6), dtype=np.uint64)
    for s in range(6):
        hi = (meta[:, 6 + s // 4] >> np.uint64(8 * (s % 4))) & np.uint64(0xFF)
        A[:, s] = meta[:, s] | (hi << np.uint64(32))
    return sym, A


def check_layout(rle):
    blocks, total = build_blocks(rle)
    plain = orc.decompress(rle)
    assert total == len(plain)
    assert len(blocks) == total // 256 + 1
    sym, A = decode_blocks(blocks, total)
    assert np.array_equal(sym[:total], plain)
    assert not sym[total:].any(), "padding past the end must be zero"
    counts = np.array([(plain == s).sum() for s in range(6)], dtype=np.uint64)
    start = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
    for s in range(6):
        occ_at_block = np.concatenate([[0], np.cumsum(plain == s)])[np.arange(len(blocks)) * 256 if total else [0]]
        assert np.array_equal(A[:, s], start[s] + occ_at_block.astype(np.uint64)), s


@pytest.mark.parametrize("kind", ["ones", "short", "long", "mixed"])
def test_plane_blocks_random(kind):
    check_layout(random_stream(3, 500, kind))


def test_plane_blocks_edges():
    check_layout(np.zeros(0, dtype=np.uint8))                      # empty BWT: one header-only block
    check_layout(runs_to_bytes([1, 2], [256, 256]))                # exact multiple of the block size
    check_layout(runs_to_bytes([5], [100000]))                     # one run over many blocks
    check_layout(runs_to_bytes([0, 5, 0], [1, 254, 1]))
    for seed in range(3):
        check_layout(raw_byte_stream(seed, 200))                   # zero digits / zero-length runs
    check_layout(orc.convert_to_vec("GTN$$ACCC$G"))


def test_plane_blocks_threaded_build_matches_serial():
    # > 1 MiB of RLE bytes switches the builder to its multi-threaded path
    rle = random_stream(9, 1_200_000, "short")
    assert rle.size > (1 << 20)
    check_layout(rle)


def test_bad_input_is_rejected():
    bad = np.array([9, 14], dtype=np.uint8)  # symbol code 6
    total = C.c_uint64()
    assert _lib.lib().msbwt_build_plane_blocks(bad.ctypes.data_as(C.c_void_p), 2, None, 0, C.byref(total)) == _lib.SIZE_MAX


def test_default_bench_has_a_fresh_pmc_traffic_entry():
    """bench.py's roofline.frac is HBM-counter traffic / kernel time; the traffic comes from a committed
    rocprofv3 --pmc summary that is only valid for the kernel sources it was taken with.  The entry for
    the DEFAULT bench configuration must carry the stamp of the sources in the tree -- re-run
    tools/profile_bench.sh + tools/profile_collect.sh after touching a query kernel."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    stamp = bench.kernel_stamp()
    entries = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["entries"]
    fresh = [e for e in entries if e["workload"] == "human" and e["k"] == 31 and e.get("kernel_stamp") == stamp]
    assert fresh, "no profiles/traffic.json entry for the default bench matches kernel stamp %s" % stamp
    e = fresh[-1]
    # the default index is the exact MSBWT of a Poisson read set: ~9e10 symbols, the same number on every run (seeded)
    assert abs(e["bwt_symbols"] - bench.HUMAN_SYMBOLS) < 1e-3 * bench.HUMAN_SYMBOLS and e["query_kind"] == "walk"
    # also the extra line of the default run on the real C4 MSBWT
    c4 = [x for x in entries if x["workload"] == "c4" and x.get("query_kind") == "reads" and x.get("kernel_stamp") == stamp]
    assert c4 and os.path.exists(os.path.join(ROOT, c4[-1]["source"]))
    assert os.path.exists(os.path.join(ROOT, e["source"]))
    assert 0 < e["traffic_bytes_per_query"] < 31 * 2 * 184  # far below the reference algorithm's worst case


@pytest.mark.parametrize("kind", ["ones", "short", "long", "mixed", "raw"])
def test_run_blocks_decode_back_to_the_bwt(kind):
    """The host builder of the run-block format (csrc/run_index.cpp; no GPU needed): decoding the 96
    one-byte runs (or the plane-shaped overflow lines) of every block gives the BWT back, and every
    header equals start_index + symbol counts before the block."""
    nruns = {"ones": 3000, "short": 3000, "long": 40, "mixed": 150, "raw": 2500}[kind]
    rle = np.ascontiguousarray(raw_byte_stream(5, nruns) if kind == "raw" else random_stream(7, nruns, kind))
    sym = orc.decompress(rle)
    total = len(sym)
    assert total < 20_000_000
    L = msbwt._lib.lib()
    tot, nover = C.c_uint64(), C.c_uint64()
    nblocks = L.msbwt_build_run_blocks(rle.ctypes.data_as(C.c_void_p), rle.size, None, 0, None, 0, C.byref(tot), C.byref(nover))
    assert nblocks == total // 512 + 1 and tot.value == total
    blocks = np.zeros((nblocks, 32), dtype=np.uint32)
    over = np.zeros((max(nover.value, 1), 2, 8, 4), dtype=np.uint32)
    assert L.msbwt_build_run_blocks(rle.ctypes.data_as(C.c_void_p), rle.size, blocks.ctypes.data_as(C.c_void_p), nblocks,
                                    over.ctypes.data_as(C.c_void_p), nover.value, C.byref(tot), C.byref(nover)) == nblocks
    counts = np.bincount(sym, minlength=6)
    start = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
    # headers, vectorised: A[s] of block b = start[s] + #s before 512 b
    onehot = (sym[:, None] == np.arange(6, dtype=np.uint8)[None, :])
    before = np.concatenate([np.zeros((1, 6), dtype=np.uint64), np.cumsum(onehot, axis=0, dtype=np.uint64)])[np.minimum(np.arange(nblocks) * 512, total)]
    want_a = before + start[None, :]
    lo = blocks[:, :6].astype(np.uint64)
    hi = np.stack([(blocks[:, 6] >> (8 * s)) & 0xFF for s in range(4)] + [(blocks[:, 7] >> (8 * s)) & 0xFF for s in range(2)], axis=1).astype(np.uint64)
    assert np.array_equal((hi << np.uint64(32)) | lo, want_a)
    is_over = (blocks[:, 7] & np.uint32(0x80000000)) != 0
    assert int(is_over.sum()) == nover.value
    # payloads: every overflow block, and a sample of the ordinary ones (all of them when there are few)
    check = np.arange(nblocks) if nblocks <= 3000 else np.unique(np.concatenate([np.flatnonzero(is_over)[:300], np.random.default_rng(1).integers(0, nblocks, 2000), [0, nblocks - 1]]))
    for b in check:
        w = blocks[b]
        want = sym[512 * b:512 * b + 512]
        if is_over[b]:
            o = over[int(w[8])]
            for half in range(2):   # each of the two lines is a plane block of its own: A[s] at 512 b + 256 half in its meta words
                at = min(512 * b + 256 * half, total)
                a = start + np.bincount(sym[:at], minlength=6).astype(np.uint64)
                meta = o[half, :, 3].astype(np.uint64)
                got_a = [(((meta[6] >> np.uint64(8 * s)) & np.uint64(0xFF)) << np.uint64(32)) | meta[s] for s in range(4)] + \
                        [(((meta[7] >> np.uint64(8 * (s - 4))) & np.uint64(0xFF)) << np.uint64(32)) | meta[s] for s in (4, 5)]
                assert [int(x) for x in got_a] == [int(x) for x in a], (b, half)
            i = np.arange(len(want))
            words = o[i >> 8, (i & 255) >> 5]                      # (len, 4)
            got = (((words[:, 0] >> (i & 31)) & 1) | (((words[:, 1] >> (i & 31)) & 1) << 1) | (((words[:, 2] >> (i & 31)) & 1) << 2)).astype(np.uint8)
        else:
            runs = w[8:].view(np.uint8)
            got = np.repeat(runs & 7, runs >> 3).astype(np.uint8)
            assert (runs >> 3).max(initial=0) <= 31
        assert np.array_equal(got, want), b
    if kind == "ones":
        assert nover.value > 0      # run length 1: 512 pieces per block, every full block overflows


def test_sharded_ticket_scheme_covers_every_tile_once():
    """The lanes kernel's tile dealing (csrc/lanes.hip: a wave's first ticket is its own index, further
    tickets come from counter c of n as W + c + n * drawn, a wave stops at the first ticket beyond the
    batch) restated and run with waves advancing in random order: every tile is taken exactly once."""
    import random
    rng = random.Random(5)
    for _ in range(300):
        waves = rng.choice([1, 5, 8, 9, 64, 127, 128, 200, 3072])
        tiles = rng.randint(waves, waves * rng.choice([1, 2, 9, 40, 300]))  # the launcher never starts more waves than tiles
        grain = max(1, min(16, tiles // (waves * 8)))
        ncounters = min(16, max(1, waves >> 3))
        counters = [0] * ncounters
        taken = [0] * tiles

        def draw(w):
            c = (w >> 3) % ncounters
            t = counters[c]
            counters[c] += 1
            return (waves + c + t * ncounters) * grain

        state = {}  # wave -> [next_tile, seg_left, seg_after]
        for w in rng.sample(range(waves), waves):  # start-up in any order
            state[w] = [w * grain, grain, draw(w)]
        live = list(state)
        while live:
            w = rng.choice(live)
            nxt, left, after = state[w]
            if nxt >= tiles:
                live.remove(w)
                continue
            taken[nxt] += 1
            nxt, left = nxt + 1, left - 1
            if left == 0:
                nxt, left, after = after, grain, draw(w)
            state[w] = [nxt, left, after]
        assert all(t == 1 for t in taken), (waves, tiles, grain)


def test_automatic_table_depths_are_what_design_md_says():
    """csrc/table_policy.hpp through msbwt_auto_table_depths (no device): the depths DESIGN.md section 5 quotes
    for the configs, the no-pair-index and low-memory fall-backs, and the degenerate sizes."""
    GB = 10 ** 9
    auto = msbwt.auto_table_depths
    assert auto(90_000_000_000, 170 * GB) == (15, 17)        # human scale: 73 GB of packed lines beside 135 GB of blocks
    assert auto(1_946_213_783, 300 * GB) == (15, 17)         # C4: the flat parent is built deeper than its own budget allows
    assert auto(1_946_213_783, 100 * GB) == (14, 16)         # ... unless HBM is short: 2 x 73 GB do not fit
    assert auto(1_946_213_783, 20 * GB) == (13, 15)
    assert auto(233_629_767, 300 * GB) == (15, 17)           # C3: 73 entries per symbol -- most empty, a present k-mer's never
    assert auto(101_000_000, 300 * GB) == (15, 17)           # C2: 170 per symbol (the limit is 256)
    assert auto(60_000_000, 300 * GB) == (14, 16)            # below 6.7e7 symbols depth 17 would pass 256 entries per symbol
    assert auto(1_946_213_783, 300 * GB, pair_index=False) == (13, 0)   # no pair index: flat, within max(1 GiB, 2 x blocks)
    assert auto(90_000_000_000, 170 * GB, pair_index=False) == (15, 0)
    assert auto(10, 300 * GB) == (3, 5)                      # TG$$CAGCCG
    assert auto(0, 300 * GB) == (0, 0)
    assert auto(90_000_000_000, 0) == (15, 0)                # nothing known to be free: no packing
    for total in (5, 1000, 10 ** 6, 10 ** 9, 2 ** 39):
        flat, packed = auto(total, 300 * GB)
        assert 0 <= flat <= 15 and packed in (0, flat + 2) and (flat == 0 or 4 ** flat <= max(total, 4 ** (packed - 2) if packed else 0))
        assert packed == 0 or 4 ** packed <= 256 * total


def test_automatic_pair_stride_follows_memory_first_and_then_the_data():
    """csrc/table_policy.hpp through msbwt_auto_pair_stride (no device).  Small indexes take the overlapping
    (stride-96) pair blocks because they are cheap; a human-scale index takes them only when the load-time probe
    reports that present k-mers keep wide ranges (a real 30x read set: ~ the coverage; a stream of independent
    symbols: 1) and the bigger blocks fit beside the suffix table with an eighth of the HBM to spare."""
    GB = 10 ** 9
    hbm = 288 * 2 ** 30
    stride = msbwt.auto_pair_stride
    human, free_after_planes = 90_000_000_000, hbm - 45 * GB - 2 * GB
    assert stride(233_629_767, 300 * GB, hbm) == 96                         # C3: 0.3 GB of pair blocks
    assert stride(1_946_213_783, 300 * GB, hbm) == 96                       # C4: 2.6 GB
    assert stride(human, free_after_planes, hbm, typical_width=-1.0) == 128  # nothing known about the data
    assert stride(human, free_after_planes, hbm, typical_width=1.0) == 128   # the stand-in stream: independent symbols
    assert stride(human, free_after_planes, hbm, typical_width=7.0) == 128
    assert stride(human, free_after_planes, hbm, typical_width=8.0) == 96    # real data from ~10x coverage on
    assert stride(human, free_after_planes, hbm, typical_width=27.0) == 96   # 30x
    # ... but only when 120 GB of overlapping blocks fit beside the table (90 GB while it is packed) and the reserve
    assert stride(human, 240 * GB, hbm, typical_width=27.0) == 128
    assert stride(human, 100 * GB, hbm, typical_width=27.0) == 128
    # a 9e9-symbol index: 12 GB of overlapping blocks are cheap from the start
    assert stride(9_000_000_000, 280 * GB, hbm, typical_width=1.0) == 96
    for total in (10, 10 ** 6, 10 ** 9, 2 ** 39):
        for width in (-1.0, 2.0, 30.0):
            assert stride(total, 250 * GB, hbm, width) in (96, 128)


def test_gather_entry_points_reject_bad_arguments_without_touching_a_device():
    """The one-process-per-GPU entry points validate before they bind RCCL or touch HIP."""
    import ctypes as C
    L = _lib.lib()
    comm = C.c_void_p()
    ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    assert L.msbwt_comm_get_unique_id(None) == _lib.ERR_INVALID_ARG
    assert L.msbwt_comm_init_rank(None, 1, ident, 0) == _lib.ERR_INVALID_ARG
    assert L.msbwt_comm_init_rank(C.byref(comm), 0, ident, 0) == _lib.ERR_INVALID_ARG      # no ranks
    assert L.msbwt_comm_init_rank(C.byref(comm), 2, ident, 2) == _lib.ERR_INVALID_ARG      # rank outside the job
    assert L.msbwt_comm_init_rank(C.byref(comm), 1, None, 0) == _lib.ERR_INVALID_ARG
    assert L.msbwt_comm_destroy(None) == _lib.ERR_INVALID_ARG
    assert L.msbwt_rle_allgather_counts(None, None, None, 0, None, 64, None) == _lib.ERR_INVALID_ARG
    h = L.msbwt_rle_new(8)
    try:
        fake = C.c_void_p(1)
        assert L.msbwt_rle_allgather_counts(h, None, None, 0, None, 64, None) == _lib.ERR_INVALID_ARG      # no communicator
        assert L.msbwt_rle_allgather_counts(h, fake, None, 0, None, 48, None) == _lib.ERR_INVALID_ARG      # wire width
        assert L.msbwt_rle_allgather_counts(h, fake, None, 5, None, 64, None) == _lib.ERR_INVALID_ARG      # counts without buffers
        assert b"wire width" in L.msbwt_rle_last_error(h)
        assert L.msbwt_rle_get_typical_range_width(h) == -1.0 and L.msbwt_rle_get_typical_range_width(None) == -1.0
    finally:
        L.msbwt_rle_free(h)
    stride = C.c_int()
    assert L.msbwt_auto_pair_stride(10**9, 10**11, 3 * 10**11, 30.0, None) == _lib.ERR_INVALID_ARG
    assert L.msbwt_auto_pair_stride(10**9, 10**11, 3 * 10**11, 30.0, C.byref(stride)) == 0 and stride.value in (96, 128)


def test_bench_stdout_line_is_compact():
    """bench.py prints ONE line the driver must parse: round 5's had grown to 25 KB and was lost.  compact_record cuts a full record
    (here round 5's own, as committed) to the contract's keys + roofline + cpu_baseline + parity + scalar extras, well under 8 KB."""
    import json
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r05_lab", "bench_default_full_2.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000
    c = bench.compact_record(full, os.path.join(ROOT, "bench_extras.json"))
    line = json.dumps(c)
    assert len(line) < 4096 < bench.COMPACT_LIMIT
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "parity", "extras", "extras_file"):
        assert key in c, key
    assert c["value"] == full["value"] and c["roofline"]["frac"] == full["roofline"]["frac"] and c["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert c["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and c["cpu_baseline"]["cores"] == 1 and c["parity"] == {"checked": 2000000, "mismatches": 0}
    assert c["extras"]["c4_real_reads_qps"] == full["c4_real_reads"]["value"] and c["extras"]["c5_random_1e9_mismatches"] == 0
    assert "note" not in line and "telemetry" not in line and c["extras_file"] == "bench_extras.json"
    # the N > 1 shape (sharded batch, exchange step): per-rank and gather figures travel as scalars
    multi = dict(full, n_gpus=8, scaling="strong", weak_scaling={"value": 1.0e11, "note": "x" * 500},
                 ranks={"kernel_ms": [1.2] * 8, "kernel_ms_min": 1.1, "kernel_ms_max": 1.3, "exchange_ms_alone": 4.2, "note": "y" * 900},
                 native_gather={"value": 9.0e10, "equals_torch_path": True, "single_batch_latency_ms": 6.0, "single_batch_pipelined_ms": 4.5,
                                "single_batch_pipelined_narrow_destination_ms": 4.1, "note": "z" * 900})
    multi.pop("cpu_baseline", None)
    cm = bench.compact_record(multi, "bench_extras.json")
    assert len(json.dumps(cm)) < 4096 and cm["n_gpus"] == 8 and "cpu_baseline" not in cm
    assert cm["extras"]["weak_scaling_qps"] == 1.0e11 and cm["extras"]["native_gather_qps"] == 9.0e10 and cm["extras"]["ranks_kernel_ms_max"] == 1.3
    assert cm["extras"]["native_gather_single_batch_pipelined_ms"] == 4.5 and cm["extras"]["native_gather_equals_torch_path"] is True
    # the round-6 record (variant lines, budgeted C4 modes, short k, host path) as committed
    final = json.load(open(os.path.join(ROOT, "profiles", "r06_lab", "bench_extras_final.json")))
    cf = bench.compact_record(final, "bench_extras.json")
    assert len(json.dumps(cf)) < 4096
    for key in ("undeclared_k_qps", "headline_sparse_off_qps", "c4_budgeted_two_tier_qps", "c4_budgeted_two_tier_budgeted_qps", "c4_budgeted_fallback_qps",
                "short_k_worst_ratio_to_sparse_off", "host_api_bytes_qps", "host_api_packed_u32_qps", "c5_random_1e9_qps", "c4_real_reads_frac"):
        assert cf["extras"][key] > 0, key
    assert cf["value"] > 3e10 and cf["roofline"]["frac"] > 0.6 and cf["parity"]["mismatches"] == 0 and cf["extras"]["undeclared_k_counts_equal_headline"] is True
    # a record with absurdly long strings still fits
    full["config"]["workload"] = "x" * 100000
    full["roofline"]["traffic_source"] = "y" * 100000
    assert len(json.dumps(bench.compact_record(full, "e.json"))) < 4096


def test_pipelined_gather_cuts_a_shard_into_at_most_the_pieces_asked_for():
    """msbwt_rle_count_kmers_allgather_device holds pieces + 1 events: its cut (csrc/gather.hpp, allgather_piece_queries) never makes more
    pieces than that for any shard size -- round 5's floor-based cut made 68 of n_mine = 1087, pieces = 64 -- and every piece but the
    last is a whole number of 16-query units, like the Python twin's (sharded.count_kmers_pipelined)."""
    L = _lib.lib()
    for pieces in range(1, 65):
        for n in list(range(1, 2200)) + [1087, 543, 2111, 10**6 + 7, 3 * 10**8, 2**40 + 5]:
            per = L.msbwt_allgather_piece_queries(n, pieces)
            assert per >= 16 and per % 16 == 0
            made = -(-n // per)
            assert 1 <= made <= pieces, (n, pieces, per, made)
            assert per == max(16, -(-(-(-n // pieces)) // 16) * 16)
    assert L.msbwt_allgather_piece_queries(1087, 64) == 32 and L.msbwt_allgather_piece_queries(543, 32) == 32
    assert L.msbwt_allgather_piece_queries(0, 4) == 16 and L.msbwt_allgather_piece_queries(5, 0) == 16


def test_missing_rccl_is_an_error_code_not_a_crash():
    """A host without RCCL: the communicator entry points return MSBWT_ERR_RCCL (include/msbwt_hip.h) -- in a fresh process,
    with MSBWT_RCCL_LIB naming a library that does not exist (an explicit name is taken literally: no search beside it)."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, importlib, sys\n"
        "sys.path.insert(0, %r)\n"
        "L = importlib.import_module('rust-msbwt_amd')._lib\n"
        "ident = (C.c_uint8 * L.COMM_ID_BYTES)()\n"
        "comm = C.c_void_p()\n"
        "a = L.lib().msbwt_comm_get_unique_id(ident)\n"
        "b = L.lib().msbwt_comm_init_rank(C.byref(comm), 1, ident, 0)\n"
        "c = L.lib().msbwt_comm_destroy(C.c_void_p(1))\n"
        "print(a, b, c)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MSBWT_RCCL_LIB="/nonexistent/librccl_not_here.so")
    done = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert done.returncode == 0, done.stderr[-400:]
    assert done.stdout.split() == [str(_lib.ERR_RCCL)] * 3, done.stdout + done.stderr[-400:]


def test_memory_budget_plans():
    """csrc/table_policy.hpp, plan_index, through msbwt_auto_index_plan (no device): what a memory budget
    (msbwt_rle_set_memory_budget) makes of the configs -- plane blocks always, then pair blocks, then the deepest packed table
    that fits, then overlapping pair blocks; no budget = the automatic choices of the two tests above."""
    GB = 10 ** 9
    hbm = 288 * 2 ** 30
    plan = msbwt.auto_index_plan

    def free(total):
        return hbm - 2 * GB - (total // 256 + 1) * 128

    c2, c4, human = 101_000_000, 1_946_213_783, 90_000_000_000
    p = plan(c2, free(c2), hbm, 25.0, 0)
    assert (p["pair_index"], p["pair_stride"], p["flat_depth"], p["packed_depth"]) == (True, 96, 15, 17) and 73 * GB < p["index_bytes"] < 74 * GB
    p = plan(c2, free(c2), hbm, 25.0, 2 * GB)                # "C2 in 2 GB": everything but a shallower table
    assert (p["pair_index"], p["pair_stride"], p["flat_depth"], p["packed_depth"]) == (True, 96, 12, 14) and p["index_bytes"] <= 2 * GB
    p = plan(c4, free(c4), hbm, 23.0, 8 * GB)
    assert (p["pair_index"], p["flat_depth"], p["packed_depth"]) == (True, 13, 15) and p["index_bytes"] <= 8 * GB
    p = plan(c4, free(c4), hbm, 23.0, 2 * GB)                # 1 GB of plane blocks: no room for 2 GB of pair blocks
    assert (p["pair_index"], p["packed_depth"]) == (False, 0) and 0 < p["flat_depth"] <= 12 and p["index_bytes"] <= 2 * GB
    p = plan(human, free(human), hbm, 26.0, 0)
    assert (p["pair_index"], p["pair_stride"], p["flat_depth"], p["packed_depth"]) == (True, 96, 15, 17)
    p = plan(human, free(human), hbm, 26.0, 150 * GB)        # 45 + 90 GB of blocks, a depth-15 packed table in what is left
    assert (p["pair_index"], p["pair_stride"], p["packed_depth"]) == (True, 128, 15) and p["index_bytes"] <= 150 * GB
    p = plan(human, free(human), hbm, 26.0, 100 * GB)        # no room for pair blocks: plane blocks + a flat table
    assert (p["pair_index"], p["packed_depth"]) == (False, 0) and p["flat_depth"] == 15 and p["index_bytes"] <= 100 * GB
    p = plan(human, free(human), hbm, 26.0, 10 * GB)         # below the plane blocks themselves: they are built all the same
    assert (p["pair_index"], p["flat_depth"], p["packed_depth"]) == (False, 0, 0) and p["index_bytes"] > 10 * GB
    for total in (0, 10, 10 ** 6, 10 ** 9, 2 ** 39):
        for budget in (0, 10 ** 6, GB, 50 * GB, 300 * GB):
            p = plan(total, free(total) if total < 2 ** 38 else 10 * GB, hbm, 5.0, budget)
            planes = (total // 256 + 1) * 128 if total else 128
            assert p["packed_depth"] in (0, p["flat_depth"] + 2) and (budget == 0 or p["index_bytes"] <= max(budget, planes))


def test_two_bit_packing_layout():
    """msbwt_kmers_pack_2bit (host, no device): the k-mer as a base-4 number, A C G T -> 0 1 2 3, first symbol most significant,
    32 symbols per u64 -- the last symbol in bits 0-1 of word 0, symbol k-33 in bits 0-1 of word 1 -- checked against a
    restatement with Python integers; '$', 'N' and invalid codes are refused."""
    rng = np.random.default_rng(3)
    for k in (1, 2, 17, 31, 32, 33, 47, 64):
        qs = np.array([1, 2, 3, 5], dtype=np.uint8)[rng.integers(0, 4, size=(300, k))]
        words = msbwt.rle_bwt.pack_2bit(qs)
        assert words.shape == (300, 2 if k > 32 else 1)
        for row, w in zip(qs, words):
            value = 0
            for s in row:                      # first symbol most significant
                value = value * 4 + {1: 0, 2: 1, 3: 2, 5: 3}[int(s)]
            assert int(w[0]) == value & (2 ** 64 - 1) and (k <= 32 or int(w[1]) == value >> 64)
    for bad in (0, 4, 6, 255):
        q = np.array([[1, 2, bad, 3]], dtype=np.uint8)
        with pytest.raises(msbwt.MsbwtError):
            msbwt.rle_bwt.pack_2bit(q)


def test_tools_and_run_scripts_are_well_formed():
    """The measurement tools only ever run on the GPU box: a syntax error there costs a box.  Every tools/*.py compiles and
    every tools/**/*.sh passes `bash -n`."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = sorted(glob.glob(os.path.join(root, "tools", "*.py")))
    assert scripts
    for path in scripts:
        with open(path) as f:
            compile(f.read(), path, "exec")  # (syntax only: nothing is imported, nothing is written)
    shells = sorted(glob.glob(os.path.join(root, "tools", "*.sh")) + glob.glob(os.path.join(root, "tools", "runs_r*", "*.sh")))
    assert shells
    for path in shells:
        assert subprocess.run(["bash", "-n", path], capture_output=True).returncode == 0, path


def test_sparse_table_hash_and_shape_are_pure_functions():
    """csrc/sparse_table.hpp through msbwt_sparse_hash / msbwt_sparse_table_shape (no device): the mix is a bijection of 2 d-bit words --
    so bucket + tag identify a key among the keys one lookup can meet --, the buckets are contiguous windows of the mixed key, and the
    shape the builder picks leaves every lookup a probe limit of 7 buckets or more with tags unambiguous."""
    L = _lib.lib()
    rng = np.random.default_rng(1)
    for depth in (16, 19, 23, 24, 25, 27, 28, 29, 30, 31):
        n = 2 * depth
        tag_bits = 24 if depth <= 24 else 32 if depth <= 29 else 40   # the wide layout (12 entries, 32-bit tags) and the xwide one (11 entries, 40-bit tags)
        slots = 14 if depth <= 24 else 12 if depth <= 29 else 11
        for entries in (0, 1000, 3 * 10 ** 7, 3 * 10 ** 9):
            nb, probe = C.c_uint64(), C.c_int()
            assert L.msbwt_sparse_table_shape(depth, entries, C.byref(nb), C.byref(probe)) == 0
            assert nb.value >= entries / (9.0 * slots / 14) - 1 and 7 <= probe.value <= 15 and nb.value + probe.value < 2 ** 32
            window = (-(-2 ** 32 // nb.value)) << (n - 32)        # widest range of mixed keys one bucket takes
            assert (probe.value + 1) * window <= 2 ** tag_bits      # ... so that the tags of one probe sequence never collide
        # the mix loses nothing: with (nearly) 2^32 buckets the bucket is the mixed key's top 32 bits and the tag its low 24 -- together
        # every bit of a key of at most 48 bits -- so distinct keys must give distinct pairs
        keys = np.unique(rng.integers(0, 2 ** n, size=5000, dtype=np.uint64))
        b, t, t64 = C.c_uint32(), C.c_uint32(), C.c_uint64()
        mixes = set()
        for key in keys.tolist():
            assert L.msbwt_sparse_hash(key, depth, 2 ** 32 - 1, C.byref(b), C.byref(t)) == 0
            assert L.msbwt_sparse_hash64(key, depth, 2 ** 32 - 1, C.byref(b), C.byref(t64)) == 0
            assert t64.value < 2 ** tag_bits and (t64.value & 0xFFFFFFFF) == t.value
            mixes.add((b.value, t64.value))
        assert len(mixes) >= len(keys) - 1   # (the scaling by 2^32 - 1 folds the two lowest top values together)
        # and it spreads structured keys: consecutive keys (one symbol apart in the first position searched) over 1000 buckets
        hits = np.zeros(1000, dtype=np.int64)
        for key in range(20000):
            L.msbwt_sparse_hash(key, depth, 1000, C.byref(b), C.byref(t))
            hits[b.value] += 1
        assert hits.max() <= 60 and hits.min() >= 2, (hits.min(), hits.max())
    assert L.msbwt_sparse_hash(0, 15, 10, C.byref(b), C.byref(t)) == _lib.ERR_INVALID_ARG
    nb, probe = C.c_uint64(), C.c_int()
    assert L.msbwt_sparse_table_shape(32, 10, C.byref(nb), C.byref(probe)) == _lib.ERR_INVALID_ARG


def test_run_block_device_build_decision():
    """csrc/table_policy.hpp, run_build_fits_device, through msbwt_run_build_fits_device (no device): the device builder of the lean
    format needs the plane blocks beside the run blocks for a moment -- 0.5 + 0.25 (+ an eighth of that in overflow blocks) bytes per
    symbol, a 32nd of the free HBM as slack; an index that does not fit that is built on the host (and still loads)."""
    fits = _lib.lib().msbwt_run_build_fits_device
    human = 90_000_000_000
    assert fits(human, 288 * 2 ** 30) == 1           # 45 + 25.3 GB
    assert fits(human, 80 * 10 ** 9) == 1
    assert fits(human, 60 * 10 ** 9) == 0            # the finished 28 GB index fits, its builder's peak does not: host build
    assert fits(4 * human, 288 * 2 ** 30) == 1 and fits(5 * human, 288 * 2 ** 30) == 0
    assert fits(0, 1 << 20) == 1 and fits(1000, 100) == 0


def test_automatic_sparse_depth_follows_the_distinct_counts():
    """csrc/sparse_policy.hpp through msbwt_auto_sparse_depth (no device): the depth of the sparse suffix table is the deepest one the
    sizing pass reached whose table fits -- with the distinct counts the device builder measured on this repo's indexes (DESIGN.md 2,
    profiles/r05_lab/sparse_table.log), and with what a 30x human read set WITH errors would count (about 1.3e10 distinct 23-mers)."""
    def choose(distinct, avail, parent=13, wide=None, query_length=0):
        d, w = (C.c_uint64 * 32)(), (C.c_uint64 * 32)()
        for k, v in distinct.items():
            d[k] = v
        for k, v in (wide or {}).items():
            w[k] = v
        depth, nbytes = C.c_int(), C.c_uint64()
        assert _lib.lib().msbwt_auto_sparse_depth(d, w, parent, avail, query_length, C.byref(depth), C.byref(nbytes)) == 0
        return depth.value, nbytes.value

    GB = 10 ** 9
    human = {13: 67108864, 15: 1006827312, 17: 2735962851, 19: 2964035516, 21: 2979123385, 23: 2980069347}
    depth, nbytes = choose(human, 80 * GB)
    assert depth == 23 and 42 * GB < nbytes < 43 * GB            # 9 entries per 128-byte bucket: 14.2 bytes per distinct 23-mer
    assert choose(human, 41 * GB)[0] == 17                       # 19 and 21 are as large as 23 (the counts have saturated); 17 takes 38.9 + 1.2 GB
    assert choose(human, 30 * GB)[0] == 0                        # nothing fits: the loader builds the deep direct table instead
    c4 = {13: 62114725, 15: 173178299, 17: 201574823, 19: 215988570, 21: 228772228, 23: 240890440}
    depth, nbytes = choose(c4, 200 * GB)
    assert depth == 23 and 4.2 * GB < nbytes < 4.4 * GB           # 2^25 buckets: the least the 24-bit tags allow at depth 23
    assert choose(c4, 4 * GB)[0] == 21
    c4r_wide = {23: 121783}
    assert choose({13: 57811435, 15: 139915525, 17: 165712263, 19: 182311770, 21: 197456191, 23: 211785309}, 200 * GB, wide=c4r_wide) == (23, (2 ** 25 + 15) * 128 + 121783 * 16)
    c2 = {13: 8236020, 15: 9416785, 17: 10083004, 19: 10669535, 21: 11211118, 23: 11710447}
    assert choose(c2, 200 * GB)[0] == 21                         # depth 23 would need 16 x more buckets than its 1.2e7 entries fill
    c3 = {13: 15517768, 15: 19198740, 17: 21012907, 19: 22607694, 21: 24122220, 23: 25566761}
    assert choose(c3, 200 * GB)[0] == 23
    toy = {6: 2000, 8: 2900, 10: 2950, 12: 2960, 14: 2970, 16: 2975, 18: 2980, 20: 2985, 22: 2990, 23: 2992}
    depth, nbytes = choose(toy, 200 * GB, parent=6)
    assert depth == 18 and nbytes < 5 << 20                      # a toy index gets a toy table
    # reads with 0.5 % substitutions at human scale: errors, not the genome, set the counts
    noisy = {13: 67108864, 15: 1.07e9, 17: 9.0e9, 19: 1.05e10, 21: 1.17e10, 23: 1.29e10}
    noisy = {k: int(v) for k, v in noisy.items()}
    assert choose(noisy, 100 * GB)[0] == 0 and choose(noisy, 200 * GB)[0] == 23
    assert choose({}, 200 * GB)[0] == 0
    # a declared k moves the limit of the automatic depth (msbwt_rle_set_query_length): 31-mers get depth 27 -- 12 entries of 10 bytes per
    # bucket at the same 64 % load: 16.6 bytes per distinct 27-mer --, 21-mers depth 21, and k unknown stays at 23
    rule = _lib.lib().msbwt_auto_sparse_max_depth
    assert [rule(k) for k in (0, 12, 16, 21, 23, 25, 27, 28, 29, 31, 59)] == [23, 16, 16, 21, 23, 25, 27, 28, 29, 31, 31]
    human27 = dict(human)
    human27.update({25: 2980128505, 27: 2980132285, 29: 2980132543, 31: 2980132600})
    depth, nbytes = choose(human27, 80 * GB, query_length=31)
    assert depth == 31 and 53.5 * GB < nbytes < 54.5 * GB          # 40-bit tags: 11 entries of 11 bytes, 7 per bucket on average (18.1 bytes per distinct 31-mer)
    depth, nbytes = choose(human27, 80 * GB, query_length=29)
    assert depth == 29 and 68 * GB < nbytes < 69 * GB              # 32-bit tags at their limit: 2^29 buckets at least, 5.5 entries in each
    depth, nbytes = choose(human27, 52 * GB, query_length=31)
    assert depth == 27 and 49 * GB < nbytes < 50 * GB
    assert choose(human27, 80 * GB)[0] == 23 and choose(human27, 80 * GB, query_length=25)[0] == 25 and choose(human27, 80 * GB, query_length=27)[0] == 27
    assert choose(human27, 45 * GB, query_length=31)[0] == 23      # neither 29 nor 27 fits: the deepest that does
    c4_deep = dict(c4)
    c4_deep.update({25: 251000000, 27: 262000000, 29: 272000000})  # (reads with errors: every two symbols add singletons)
    depth, nbytes = choose(c4_deep, 200 * GB, query_length=29)
    assert depth == 27 and nbytes < 4.4 * GB                       # 69 GB for 2.7e8 entries is not worth two symbols
    assert choose(c2, 200 * GB, query_length=21)[0] == 21


def test_two_tier_sparse_table_is_chosen_where_the_complete_one_does_not_fit():
    """csrc/sparse_policy.hpp through msbwt_auto_sparse_choice (no device).  A 30x human read set WITH 0.5 % substitutions counts about 1.3e10
    distinct 23-mers of which the genome's 3e9 occur more than once: the complete depth-23 table needs 185 GB, the two-tier one
    -- entries for the suffixes that occur at least twice, filter bits for the rest -- about 66 GB, so with 120 GB free the two-tier form of
    depth 23 is what the loader builds; where the complete table fits it is preferred; tiers = 0 / 1 force either form."""
    def choose(distinct, once, avail, parent=13, tiers=-1, query_length=0, wide=None):
        d, w, o = (C.c_uint64 * 32)(), (C.c_uint64 * 32)(), (C.c_uint64 * 32)()
        for k, v in distinct.items():
            d[k] = int(v)
        for k, v in once.items():
            o[k] = int(v)
        for k, v in (wide or {}).items():
            w[k] = int(v)
        depth, tier, nbytes = C.c_int(), C.c_int(), C.c_uint64()
        assert _lib.lib().msbwt_auto_sparse_choice(d, w, o, parent, int(avail), query_length, tiers, C.byref(depth), C.byref(tier), C.byref(nbytes)) == 0
        return depth.value, tier.value, nbytes.value

    GB = 10 ** 9
    noisy = {13: 67108864, 15: 1.07e9, 17: 9.0e9, 19: 1.05e10, 21: 1.17e10, 23: 1.29e10}
    once = {13: 0, 15: 2.0e7, 17: 6.2e9, 19: 7.5e9, 21: 8.7e9, 23: 9.9e9}
    depth, tier, nbytes = choose(noisy, once, 120 * GB)
    assert (depth, tier) == (23, 1) and 62 * GB < nbytes < 68 * GB          # 3.0e9 entries at 5.8 per 128-byte bucket
    assert choose(noisy, once, 200 * GB)[:2] == (23, 0)                       # the complete table where it fits (185 GB)
    assert choose(noisy, once, 120 * GB, tiers=0)[0] == 0                     # complete tables only: nothing fits
    assert choose(noisy, once, 200 * GB, tiers=1)[:2] == (23, 1)
    assert choose(noisy, once, 55 * GB)[0] == 0                               # not even the two-tier form of any depth: the deep direct table
    # a chr20-sized read set with errors (config C4 as the device builder counted it): the complete depth-23 table sits at the 2^25 buckets its
    # 24-bit tags demand (4.3 GB), depth 21's needs 3.3 GB; the two-tier form of depth 23 holds a third of the entries, gets by with a probe
    # limit of 3 and half the buckets -- 2^24, 2.1 GB -- so a budget of 2.6 GB for the table keeps depth 23
    c4 = {13: 62114725, 15: 173178299, 17: 201574823, 19: 215988570, 21: 228772228, 23: 240890440}
    c4_once = {15: 30000000, 17: 120000000, 19: 145000000, 21: 161297690, 23: 173293266}
    depth, tier, nbytes = choose(c4, c4_once, 2.6 * GB)
    assert (depth, tier) == (23, 1) and 2.1 * GB < nbytes < 2.2 * GB
    assert choose(c4, c4_once, 200 * GB)[:2] == (23, 0) and choose(c4, c4_once, 2.6 * GB, tiers=0)[0] < 21
    assert choose(c4, c4_once, 1.6 * GB)[:2] == (21, 1)                       # 6.7e7 entries at 5.8 per bucket: 1.49 GB
    # the filter's load bounds the table from below: 32 once-only suffixes per bucket at most
    few_solid = {13: 1000, 15: 10**6, 17: 10**9, 19: 2 * 10**9}
    few_once = {17: 10**9 - 10**6, 19: 2 * 10**9 - 10**6}
    depth, tier, nbytes = choose(few_solid, few_once, 20 * GB, tiers=1)
    assert (depth, tier) == (19, 1) and nbytes >= ((2 * 10**9 - 10**6) // 32) * 128
    # error-free reads: nothing occurs once, the complete table always wins
    human = {13: 67108864, 15: 1006827312, 17: 2735962851, 19: 2964035516, 21: 2979123385, 23: 2980069347}
    assert choose(human, {23: 120, 21: 80}, 80 * GB)[:2] == (23, 0)
    # a declared k: the two-tier form reaches depth 29 (32-bit tags, 9 entries per bucket); depths 30..31 have no two-tier form
    deep = dict(noisy)
    deep.update({25: 1.41e10, 27: 1.53e10, 29: 1.65e10, 31: 1.77e10})
    deep_once = dict(once)
    deep_once.update({25: 1.11e10, 27: 1.23e10, 29: 1.35e10, 31: 1.47e10})
    depth, tier, nbytes = choose(deep, deep_once, 120 * GB, query_length=31)
    assert (depth, tier) == (29, 1) and nbytes <= 120 * GB
    assert choose(deep, deep_once, 80 * GB, query_length=31)[:2] == (29, 1)   # 5.0 entries per bucket: 76.8 GB + its slot counters (its tags would allow 2^28 buckets)
    assert choose(deep, deep_once, 70 * GB, query_length=31)[:2] == (23, 1)   # 5.0 entries per bucket at depths 25..29, 5.8 up to 24: 66.2 GB
    # the filter as a pure function: a word 0..7 and at most four bits, the same for the same tag
    word, mask, w2, m2 = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    seen = set()
    for tag in list(range(2000)) + [0xFFFFFF, 0xFFFFFFFF, 0x12345678]:
        assert _lib.lib().msbwt_sparse_filter_bits(tag, C.byref(word), C.byref(mask)) == 0
        assert _lib.lib().msbwt_sparse_filter_bits(tag, C.byref(w2), C.byref(m2)) == 0
        assert (word.value, mask.value) == (w2.value, m2.value) and word.value < 8 and 1 <= bin(mask.value).count("1") <= 4
        seen.add((word.value, mask.value))
    assert len(seen) > 1900 and len({w for w, _ in seen}) == 8
