"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the reference's
golden vectors.  Needs an MI355X: run with `pytest -m gpu`."""
import itertools
import os

import numpy as np
import pytest

import rust_msbwt_amd as msbwt
from rust_msbwt_amd import BWTRange, RleBWT
from oracle import oracle as orc
from rle_random import random_kmers, random_stream, raw_byte_stream, runs_to_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "groups", "lanes", "runs", "sparse"])
def search_kernel(request, monkeypatch):
    """Every test of this file runs five times: with the automatic choices (kernel, sparse suffix table when the data warrant
    one), with the 8-lanes-per-query kernel forced, with the one-query-per-lane (LDS-staged) kernel forced on the DIRECT suffix
    table only, on the run-block index format (MSBWT_BLOCKS=runs: no plane blocks, no pair index), and with the lane-per-query
    kernel on a SPARSE suffix table of the smallest depth (16: every query of 16 symbols or more is looked up there) -- a handle
    reads MSBWT_SEARCH / MSBWT_BLOCKS / MSBWT_SPARSE_TABLE when it is created."""
    mode = request.param
    monkeypatch.setenv("MSBWT_SEARCH", "lanes" if mode in ("lanes", "sparse") else ("auto" if mode == "runs" else mode))
    monkeypatch.setenv("MSBWT_BLOCKS", "runs" if mode == "runs" else "planes")
    monkeypatch.setenv("MSBWT_SPARSE_TABLE", {"lanes": "0", "sparse": "16"}.get(mode, "auto"))
    # pair blocks: the forced-lanes mode keeps the disjoint 128-position blocks, the others take the
    # automatic choice (overlapping stride-96 blocks on indexes this small)
    monkeypatch.setenv("MSBWT_PAIR_STRIDE", "128" if mode == "lanes" else "0")
    return mode


def needs_plane_blocks(mode):
    if mode == "runs":
        pytest.skip("pair index / packed table / block download exist on plane blocks only")

CODES = {"$": 0, "A": 1, "C": 2, "G": 3, "N": 4, "T": 5}


def stoi(s):
    return [CODES[c] for c in s]


def gpu_bwt(rle, bin_power=8):
    b = RleBWT.with_bin_power(bin_power)
    b.load_vector(rle)
    return b


def test_g4_totals_via_npy(golden, tmp_path):
    g = golden["G4_totals"]
    path = str(tmp_path / "g4.npy")
    msbwt.bwt_converter.save_bwt_numpy(msbwt.bwt_converter.convert_to_vec(orc.naive_bwt(g["strings"])), path)
    b = RleBWT()
    b.load_numpy_file(path)
    assert [b.get_symbol_count(s) for s in range(6)] == g["symbol_counts"]


def test_g6_constrain_range_exhaustive(golden):
    """rle_bwt.rs:603-675, through the GPU."""
    g = golden["G6_constrain_range"]
    text = g["bwt"]
    ints = stoi(text)
    rle = msbwt.bwt_converter.convert_to_vec(text)
    for bp in g["bin_powers"] + [8]:
        b = gpu_bwt(rle, bp)
        counts = [b.get_symbol_count(s) for s in range(6)]
        start = np.concatenate([[0], np.cumsum(counts)[:-1]])
        end = np.cumsum(counts)
        for sym in range(6):
            assert b.constrain_range(sym, BWTRange(0, len(text))) == BWTRange(int(start[sym]), int(end[sym]))
            cnt = 0
            for ind in range(len(text) + 1):
                assert b.constrain_range(sym, BWTRange(0, ind)) == BWTRange(int(start[sym]), int(start[sym]) + cnt)
                assert b.constrain_range(sym, BWTRange(ind, len(text))) == BWTRange(int(start[sym]) + cnt, int(end[sym]))
                if ind < len(text) and ints[ind] == sym:
                    cnt += 1


@pytest.mark.parametrize("key", ["G7_count_kmer", "G10_load_and_add"])
def test_count_kmer_literals(golden, key):
    """rle_bwt.rs:677-710 and dynamic_bwt.rs:733-773."""
    g = golden[key]
    rle = msbwt.bwt_converter.convert_to_vec(orc.naive_bwt(g["strings"]))
    for bp in g.get("bin_powers", [1, 2, 3, 4]):
        b = gpu_bwt(rle, bp)
        for c in range(6):
            assert b.count_kmer([c]) == b.get_symbol_count(c)
        for s in g["strings"]:
            assert b.count_kmer(msbwt.string_util.convert_stoi(s)) == 1
        for kmer, n in g["counts"].items():
            assert b.count_kmer(msbwt.string_util.convert_stoi(kmer)) == n


def test_g8_doc_tests(golden):
    g = golden["G8_doc_tests"]
    b = gpu_bwt(msbwt.bwt_converter.convert_to_vec(g["bwt"]))
    for codes, n in g["count_codes"]:
        assert b.count_kmer(codes) == n
    for text, n in g["count_text"]:
        assert b.count_kmer(msbwt.string_util.convert_stoi(text)) == n
    assert b.get_symbol_count(0) == g["symbol_count_0"]
    assert b.get_total_size() == g["total_size"]


def test_g9_config_c1_two_string(golden, golden_dir):
    """BASELINE.json configs[0]: two_string.npy, every 4-mer (all 1296 over the 6 symbols)."""
    g = golden["G9_two_string"]
    b = RleBWT()
    b.load_numpy_file(os.path.join(golden_dir, g["file"]))
    for kmer, n in g["counts"].items():
        assert b.count_kmer(msbwt.string_util.convert_stoi(kmer)) == n
    o = orc.OracleRleBWT()
    o.load_numpy_file(os.path.join(golden_dir, g["file"]))
    kmers = np.array(list(itertools.product(range(6), repeat=4)), dtype=np.uint8)
    got = b.count_kmers(kmers)
    assert np.array_equal(got, o.count_kmers(kmers))
    present = {"$ACG", "ACGT", "CGT$", "GT$A", "T$AC", "$TGC", "TGCA", "GCA$", "CA$T", "A$TG"}
    syms = "$ACGNT"
    for row, c in zip(kmers, got):
        assert int(c) == (1 if "".join(syms[x] for x in row) in present else 0)


def _ranges_parity(rle, seed, n=4000):
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    total = o.get_total_size()
    assert b.get_total_size() == total
    assert [b.get_symbol_count(s) for s in range(6)] == [o.get_symbol_count(s) for s in range(6)]
    rng = np.random.default_rng(seed)
    pos = np.concatenate([rng.integers(0, total + 1, size=n), [0, total, total, 0]])
    edge = np.arange(0, total + 1, 256)[:200]
    pos = np.concatenate([pos, edge, np.maximum(edge, 1) - 1, np.minimum(edge + 1, total)])
    l = rng.choice(pos, size=n)
    h = rng.choice(pos, size=n)
    l, h = np.minimum(l, h).astype(np.uint64), np.maximum(l, h).astype(np.uint64)
    syms = rng.integers(0, 6, size=n).astype(np.uint8)
    gl, gh = b.constrain_ranges(syms, l, h)
    ol, oh = o.constrain_ranges(syms, l, h)
    assert np.array_equal(gl, ol)
    assert np.array_equal(gh, oh)
    return o, b


@pytest.mark.parametrize("kind", ["ones", "short", "long", "mixed"])
def test_constrain_ranges_random_streams(kind):
    _ranges_parity(random_stream(21, 3000, kind), seed=5)


def test_constrain_ranges_edge_streams():
    for syms, lens in [([1, 2], [256, 256]), ([0, 1], [1, 255]), ([3], [100000]), ([0], [512]),
                       ([5, 0, 5], [32, 1024, 32768]), ([4, 1], [255, 1])]:
        _ranges_parity(runs_to_bytes(syms, lens), seed=1, n=500)
    for seed in range(3):
        _ranges_parity(raw_byte_stream(seed, 300), seed=seed, n=500)


def _real_bwt(seed, nreads, length):
    """A small true multi-string BWT (so that k-mers of the reads are present)."""
    rng = np.random.default_rng(seed)
    genome = "".join(rng.choice(list("ACGT"), size=600))
    reads = []
    for _ in range(nreads):
        p = int(rng.integers(0, len(genome) - length))
        r = list(genome[p:p + length])
        if rng.random() < 0.3:
            r[int(rng.integers(0, length))] = "ACGTN"[int(rng.integers(0, 5))]
        reads.append("".join(r))
    return reads, orc.convert_to_vec(orc.naive_bwt(reads))


@pytest.mark.parametrize("k", [1, 2, 5, 12, 21, 31, 32, 33, 42, 43, 50, 59, 63, 64, 65, 70])
def test_count_kmers_on_true_bwt(k):
    reads, rle = _real_bwt(2, 150, 72)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    rng = np.random.default_rng(k)
    qs = []
    for r in reads:                       # present k-mers (windows of the reads)
        if len(r) >= k:
            p = int(rng.integers(0, len(r) - k + 1))
            qs.append(orc.convert_stoi(r[p:p + k]))
    qs = np.array(qs, dtype=np.uint8)
    qs = np.concatenate([qs, random_kmers(k, 300, k), random_kmers(k + 1, 100, k, alphabet=(0, 1, 2, 3, 4, 5))])
    got = b.count_kmers(qs)
    exp = o.count_kmers(qs)
    assert np.array_equal(got, exp)
    assert got[:10].sum() > 0
    # the single-query trait call is the same path
    for i in (0, len(qs) // 2, len(qs) - 1):
        assert b.count_kmer(qs[i]) == int(exp[i])


def test_count_kmers_random_stream_any_symbols():
    rle = random_stream(4, 20000, "short")
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    for k in (3, 8, 21):
        qs = random_kmers(k, 5000, k, alphabet=(0, 1, 2, 3, 4, 5))
        assert np.array_equal(b.count_kmers(qs), o.count_kmers(qs))


def test_empty_kmer_and_empty_bwt():
    b = gpu_bwt(msbwt.bwt_converter.convert_to_vec("GTN$$ACCC$G"))
    assert b.count_kmer([]) == 11                       # msbwt_core.rs:128-131,160
    assert np.array_equal(b.count_kmers(np.zeros((3, 0), dtype=np.uint8)), [11, 11, 11])
    assert len(b.count_kmers(np.zeros((0, 5), dtype=np.uint8))) == 0
    e = gpu_bwt(np.zeros(0, dtype=np.uint8))
    assert e.get_total_size() == 0
    assert e.count_kmer([1, 2]) == 0
    assert e.count_kmer([]) == 0


def test_error_behaviour(tmp_path, golden_dir):
    b = gpu_bwt(msbwt.bwt_converter.convert_to_vec("GTN$$ACCC$G"))
    with pytest.raises(msbwt.MsbwtError) as e:          # reference: assert -> panic
        b.count_kmer([1, 6])
    assert e.value.code == msbwt._lib.ERR_INVALID_SYMBOL
    with pytest.raises(msbwt.MsbwtError) as e:
        b.count_kmers(np.array([[1, 2], [7, 1]], dtype=np.uint8))
    assert e.value.code == msbwt._lib.ERR_INVALID_SYMBOL
    with pytest.raises(msbwt.MsbwtError) as e:
        b.constrain_range(1, BWTRange(0, 12))
    assert e.value.code == msbwt._lib.ERR_INVALID_RANGE
    with pytest.raises(msbwt.MsbwtError) as e:
        b.constrain_range(1, BWTRange(5, 4))
    assert e.value.code == msbwt._lib.ERR_INVALID_RANGE
    assert b.count_kmer([1]) == 1                        # the handle stays usable
    fresh = RleBWT()
    with pytest.raises(msbwt.MsbwtError) as e:
        fresh.count_kmer([1])
    assert e.value.code == msbwt._lib.ERR_NOT_LOADED
    with pytest.raises(OSError):
        fresh.load_numpy_file(str(tmp_path / "missing.npy"))
    good = open(os.path.join(golden_dir, "two_string.npy"), "rb").read()
    for name, blob, exc in [("trunc", good[:-1], EOFError), ("short", good[:50], EOFError),
                            ("tiny", good[:7], msbwt.MsbwtError),
                            ("nojson", good[:10] + b"{'descr' '|u1'}".ljust(86) + good[96:], msbwt.MsbwtError)]:
        p = str(tmp_path / (name + ".npy"))
        open(p, "wb").write(blob)
        with pytest.raises(exc):
            fresh.load_numpy_file(p)
    with pytest.raises(msbwt.MsbwtError) as e:
        fresh.load_vector(np.array([9, 14], dtype=np.uint8))  # symbol code 6 in the stream
    assert e.value.code == msbwt._lib.ERR_INVALID_SYMBOL


def test_reload_replaces_index():
    b = gpu_bwt(msbwt.bwt_converter.convert_to_vec("GTN$$ACCC$G"))
    assert b.count_kmer(stoi("CC")) == 1
    b.load_vector(msbwt.bwt_converter.convert_to_vec("TG$$CAGCCG"))
    assert b.get_total_size() == 10
    assert b.count_kmer(stoi("CG")) == 2
    # a failed load leaves the handle unloaded and reporting nothing of the previous BWT
    with pytest.raises(msbwt.MsbwtError) as err:
        b.load_vector(np.array([9, 10, 15], dtype=np.uint8))  # 15 & 7 = 7: not a symbol code
    assert err.value.code == msbwt._lib.ERR_INVALID_SYMBOL
    assert b.get_total_size() == 0 and b.get_symbol_count(1) == 0
    with pytest.raises(msbwt.MsbwtError) as err:
        b.count_kmer(stoi("CG"))
    assert err.value.code == msbwt._lib.ERR_NOT_LOADED


def test_device_and_host_entry_points_keep_separate_status_words():
    """An invalid symbol in a *_device batch must be reported by device_status, not swallowed (or
    blamed on them) by host-pointer calls in between."""
    import torch
    reads, rle = _real_bwt(5, 100, 60)
    b = gpu_bwt(rle)
    dev = torch.device("cuda", 0)
    bad = torch.full((130, 21), 7, dtype=torch.uint8, device=dev)
    out = torch.zeros(130, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    b.count_kmers_device(bad.data_ptr(), 21, 130, out.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    good = np.array([orc.convert_stoi(reads[0][:21])], dtype=np.uint8)
    assert b.count_kmers(good)[0] >= 1           # a host call: fine, and does not clear the device word
    with pytest.raises(msbwt.MsbwtError) as err:
        b.device_status(stream)
    assert err.value.code == msbwt._lib.ERR_INVALID_SYMBOL
    b.device_status(stream)                      # cleared by the call that reported it
    assert int(out[0].item()) == -1              # u64::MAX


@pytest.mark.parametrize("depth", [0, 1, 2, 5, 9])
def test_suffix_table_never_changes_results(depth):
    """The precomputed suffix table (the reference's stubbed kmer_cache idea) must be
    invisible: same counts for every depth, for k below/at/above the depth, with $ and N."""
    reads, rle = _real_bwt(7, 200, 70)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    b.set_table_depth(depth)
    assert b.get_table_depth() == depth
    rng = np.random.default_rng(depth)
    for k in (1, 3, 5, 9, 10, 21, 31, 32):
        qs = [orc.convert_stoi(r[p:p + k]) for r in reads for p in (int(rng.integers(0, len(r) - k + 1)),)]
        qs = np.concatenate([np.array(qs, dtype=np.uint8), random_kmers(k, 200, k),
                             random_kmers(k + 7, 100, k, alphabet=(0, 1, 2, 3, 4, 5))])
        assert np.array_equal(b.count_kmers(qs), o.count_kmers(qs)), (depth, k)


def test_table_rebuilt_on_reload_and_auto_depth():
    rle = random_stream(8, 60000, "short")
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    d = b.get_table_depth()
    assert 1 <= d <= 17 and 4 ** d <= 256 * o.get_total_size()  # flat: 4^d <= T; packed: at most 256 entries per symbol
    qs = random_kmers(1, 3000, 12)
    assert np.array_equal(b.count_kmers(qs), o.count_kmers(qs))
    b.load_vector(msbwt.bwt_converter.convert_to_vec("TG$$CAGCCG"))   # tiny: table depth shrinks
    assert b.get_table_depth() <= 5                                   # flat 1 (4 <= 10) on its own, packed 5 (1024 <= 2560)
    assert b.count_kmer(stoi("CG")) == 2


def test_device_pointer_api_alignment_and_ragged_tiles():
    """Device-pointer entry point (torch only supplies HBM buffers): batch sizes around the
    64-query tile, and a misaligned query buffer (takes the generic kernel)."""
    torch = pytest.importorskip("torch")
    reads, rle = _real_bwt(9, 120, 64)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    k = 21
    base = np.concatenate([np.array([orc.convert_stoi(r[3:3 + k]) for r in reads], dtype=np.uint8), random_kmers(3, 400, k)])
    for n in (1, 7, 63, 64, 65, 127, 128, 129, 520):
        for offset in (0, 5):
            buf = torch.zeros(n * k + 64, dtype=torch.uint8, device=dev)
            buf[offset:offset + n * k] = torch.from_numpy(base[:n].reshape(-1)).to(dev)
            out = torch.full((n + 2,), -7, dtype=torch.int64, device=dev)
            b.count_kmers_device(buf.data_ptr() + offset, k, n, out.data_ptr() + 8, stream)
            b.device_status(stream)
            got = out.cpu().numpy()
            assert got[0] == -7 and got[-1] == -7          # nothing written outside the batch
            assert np.array_equal(got[1:-1].astype(np.uint64), o.count_kmers(base[:n])), (n, offset)


def _download_blocks(b):
    import ctypes as C
    L = msbwt._lib.lib()
    n = L.msbwt_rle_download_blocks(b._h, None, 0)
    assert n != msbwt._lib.SIZE_MAX
    out = np.zeros((n, 8, 4), dtype=np.uint32)
    assert L.msbwt_rle_download_blocks(b._h, out.ctypes.data_as(C.c_void_p), n) == n
    return out


def _host_blocks(rle):
    import ctypes as C
    rle = np.ascontiguousarray(rle, dtype=np.uint8)
    L = msbwt._lib.lib()
    total = C.c_uint64()
    n = L.msbwt_build_plane_blocks(rle.ctypes.data_as(C.c_void_p), rle.size, None, 0, C.byref(total))
    out = np.zeros((n, 8, 4), dtype=np.uint32)
    L.msbwt_build_plane_blocks(rle.ctypes.data_as(C.c_void_p), rle.size, out.ctypes.data_as(C.c_void_p), n, C.byref(total))
    return out


@pytest.mark.parametrize("case", ["ones", "short", "long", "mixed", "raw", "edges", "big_short"])
def test_device_built_index_equals_host_built(case, search_kernel):
    """The device-side builder (scan + atomic paint) against the host builder, word for word."""
    needs_plane_blocks(search_kernel)
    if case == "raw":
        streams = [raw_byte_stream(s, 500) for s in range(4)]
    elif case == "edges":
        streams = [runs_to_bytes(s, l) for s, l in [([1, 2], [256, 256]), ([0, 1], [1, 255]), ([3], [100000]),
                                                   ([0], [512]), ([5, 0, 5], [32, 1024, 32768]), ([4, 1], [255, 1]),
                                                   ([2], [2047]), ([2], [2048]), ([1, 2, 1], [2049, 1, 4096])]]
        streams.append(np.zeros(0, dtype=np.uint8))
        streams.append(orc.convert_to_vec("GTN$$ACCC$G"))
    elif case == "big_short":
        streams = [random_stream(12, 300000, "short")]          # many 4 KiB tiles: exercises the tile scan
    else:
        streams = [random_stream(31, 20000, case)]
    for rle in streams:
        b = gpu_bwt(rle)
        dev, host = _download_blocks(b), _host_blocks(rle)
        assert dev.shape == host.shape
        assert np.array_equal(dev, host)


@pytest.mark.parametrize("k", [1, 5, 21, 31, 32, 33, 47])
def test_fused_read_kmers_match_host_side_preparation(k):
    """count_read_kmers == convert_stoi -> windows -> (reverse_complement_i) -> count_kmer done
    on the host with the reference's string_util semantics (src/string_util.rs)."""
    reads, rle = _real_bwt(11, 90, 48)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    rng = np.random.default_rng(k)
    texts = [r for r in reads if len(r) == 48][:60]
    # sprinkle lower case, N, and bytes outside the alphabet (all map to N), and a '$'
    noisy = []
    for t in texts:
        t = list(t)
        for _ in range(3):
            p = int(rng.integers(0, len(t)))
            t[p] = rng.choice(["a", "c", "g", "t", "n", "N", "x", "-", "$", "\x04"])
        noisy.append("".join(t))
    texts = texts[:30] + noisy[30:]
    fwd, rc = b.count_read_kmers(texts, k, revcomp=True)
    w = 48 - k + 1
    assert fwd.shape == rc.shape == (len(texts), w)
    for r, t in enumerate(texts):
        codes = orc.convert_stoi(t.encode("latin1"))
        wins = np.array([codes[i:i + k] for i in range(w)], dtype=np.uint8)
        rcs = np.array([orc.reverse_complement_i(x) for x in wins], dtype=np.uint8)
        assert np.array_equal(fwd[r], o.count_kmers(wins)), (k, r)
        assert np.array_equal(rc[r], o.count_kmers(rcs)), (k, r)
    assert fwd[:30].min() >= 1                     # clean reads: every window is present
    # symbol-code input, single strand
    codes = np.array([orc.convert_stoi(t.encode("latin1")) for t in texts], dtype=np.uint8)
    f2, none = b.count_read_kmers(codes, k, ascii=False)
    assert none is None and np.array_equal(f2, fwd)
    none, r2 = b.count_read_kmers(codes, k, ascii=False, forward=False, revcomp=True)
    assert none is None and np.array_equal(r2, rc)
    with pytest.raises(msbwt.MsbwtError):
        b.count_read_kmers(codes, 49, ascii=False)


def test_fused_read_kmers_invalid_codes_are_flagged():
    b = gpu_bwt(msbwt.bwt_converter.convert_to_vec("GTN$$ACCC$G"))
    bad = np.array([[1, 2, 7, 1]], dtype=np.uint8)
    with pytest.raises(msbwt.MsbwtError) as e:
        b.count_read_kmers(bad, 2, ascii=False)
    assert e.value.code == msbwt._lib.ERR_INVALID_SYMBOL


def test_ragged_read_kmers():
    """Reads of different lengths (some shorter than k, one empty) through the fused path."""
    reads, rle = _real_bwt(13, 80, 40)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    rng = np.random.default_rng(3)
    ragged = []
    for r in reads[:50]:
        cut = int(rng.integers(0, len(r) + 1))
        ragged.append(r[:cut])
    ragged += ["", "ACG", reads[0]]
    for k in (4, 21, 31):
        fwd, rc, woff = b.count_ragged_read_kmers(ragged, k, revcomp=True)
        assert len(woff) == len(ragged) + 1 and woff[-1] == len(fwd) == len(rc)
        for i, t in enumerate(ragged):
            n = max(0, len(t) - k + 1)
            assert int(woff[i + 1] - woff[i]) == n
            if n == 0:
                continue
            codes = orc.convert_stoi(t)
            wins = np.array([codes[j:j + k] for j in range(n)], dtype=np.uint8)
            rcs = np.array([orc.reverse_complement_i(x) for x in wins], dtype=np.uint8)
            assert np.array_equal(fwd[int(woff[i]):int(woff[i + 1])], o.count_kmers(wins)), (k, i)
            assert np.array_equal(rc[int(woff[i]):int(woff[i + 1])], o.count_kmers(rcs)), (k, i)
    # nothing to count: every read shorter than k
    f, r, w = b.count_ragged_read_kmers(["AC", "G"], 5)
    assert len(f) == 0 and w.tolist() == [0, 0, 0]


@pytest.mark.parametrize("pair", [0, 1])
@pytest.mark.parametrize("depth", [0, 3, 8])
def test_pair_index_never_changes_results(pair, depth):
    """Two-symbols-per-step blocks on/off x table depths: identical counts, for k of both
    parities, with $ / N inside the k-mer (those steps fall back to single symbols)."""
    reads, rle = _real_bwt(17, 220, 64)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    b.set_table_depth(depth)
    b.set_pair_index(pair)
    assert b.get_pair_index() == (bool(pair) and b.get_block_format() == "planes")
    rng = np.random.default_rng(depth * 2 + pair)
    for k in (1, 2, 3, 4, 9, 10, 20, 21, 31, 32, 33, 48, 59, 64):
        qs = [orc.convert_stoi(r[p:p + k]) for r in reads if len(r) >= k for p in (int(rng.integers(0, len(r) - k + 1)),)]
        qs = np.concatenate([np.array(qs, dtype=np.uint8), random_kmers(k, 300, k),
                             random_kmers(k + 7, 200, k, alphabet=(0, 1, 2, 3, 4, 5))])
        assert np.array_equal(b.count_kmers(qs), o.count_kmers(qs)), (pair, depth, k)
    texts = [r for r in reads if len(r) == 64][:40]
    fwd, rc = b.count_read_kmers(texts, 31, revcomp=True)
    for r, t in enumerate(texts):
        codes = orc.convert_stoi(t)
        wins = np.array([codes[i:i + 31] for i in range(34)], dtype=np.uint8)
        assert np.array_equal(fwd[r], o.count_kmers(wins))
        assert np.array_equal(rc[r], o.count_kmers(np.array([orc.reverse_complement_i(x) for x in wins], dtype=np.uint8)))


def test_pair_index_on_random_streams_and_superblock_borders():
    """Synthetic streams (not BWTs): the pair identity must hold for any symbol string.  The
    second stream is longer than one superblock (2^17 pair blocks)."""
    for rle, nq in ((random_stream(41, 50000, "short"), 4000), (random_stream(43, 20_000, "long"), 4000)):
        o = orc.OracleRleBWT()
        o.load_vector(rle)
        b = gpu_bwt(rle)
        b.set_table_depth(2)
        b.set_pair_index(1)
        for k in (6, 7, 12):
            qs = random_kmers(k, nq, k)
            assert np.array_equal(b.count_kmers(qs), o.count_kmers(qs)), k
        b.set_pair_index(0)
        qs = random_kmers(5, nq, 8)
        assert np.array_equal(b.count_kmers(qs), o.count_kmers(qs))


@pytest.mark.parametrize("flat_depth", [1, 3, 6])
def test_packed_table_never_changes_results(flat_depth, search_kernel):
    """The packed suffix table (two levels deeper than the flat one, 30 entries per 128-byte line,
    16-bit deltas) on a true BWT: present and absent k-mers around the effective depth."""
    needs_plane_blocks(search_kernel)
    reads, rle = _real_bwt(12 + flat_depth, 220, 75)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    b.set_pair_index(1)
    b.set_table_depth(flat_depth)
    b.set_table_packed(1)
    assert b.get_table_packed() and b.get_table_depth() == flat_depth + 2
    rng = np.random.default_rng(flat_depth)
    for k in (flat_depth + 1, flat_depth + 2, flat_depth + 3, 21, 31, 40):
        qs = []
        for r in reads:
            if len(r) >= k:
                p = int(rng.integers(0, len(r) - k + 1))
                qs.append(orc.convert_stoi(r[p:p + k]))
        qs = np.concatenate([np.array(qs, dtype=np.uint8), random_kmers(k, 500, k), random_kmers(k + 1, 100, k, alphabet=(0, 1, 2, 3, 4, 5))])
        exp = o.count_kmers(qs)
        assert np.array_equal(b.count_kmers(qs), exp), k
    b.set_table_packed(0)
    assert not b.get_table_packed() and b.get_table_depth() == flat_depth
    assert np.array_equal(b.count_kmers(qs), exp)
    b.set_table_packed(1)
    b.set_pair_index(0)          # no pair index: the table falls back to its flat form
    assert not b.get_table_packed() and b.get_table_depth() == flat_depth
    assert np.array_equal(b.count_kmers(qs), exp)


def test_packed_table_escape_lines_on_long_run_streams(search_kernel):
    """Arbitrary symbol streams with runs of up to 10^5: ranges far wider than 16 bits, so most packed
    lines are ESCAPE lines and their queries must search from scratch -- same counts."""
    needs_plane_blocks(search_kernel)
    for seed, kind in ((51, "long"), (52, "mixed"), (53, "short")):
        rle = random_stream(seed, 30_000, kind)
        o = orc.OracleRleBWT()
        o.load_vector(rle)
        b = gpu_bwt(rle)
        b.set_pair_index(1)
        b.set_table_depth(3)
        b.set_table_packed(1)
        assert b.get_table_packed() and b.get_table_depth() == 5
        info = b.table_info()
        assert info["lines"] == (4 ** 5 + 29) // 30 and info["side_bytes"] == 512 * info["escape_lines"]
        if kind == "long":
            assert info["escape_lines"] > 0
        for k in (5, 6, 9, 14):
            qs = random_kmers(k + seed, 3000, k)
            assert np.array_equal(b.count_kmers(qs), o.count_kmers(qs)), (kind, k)
        rep = b.replicate(0)     # a replica carries the packed table (and its side array) along
        assert rep.get_table_packed() and rep.get_table_depth() == 5 and rep.table_info() == info
        assert np.array_equal(rep.count_kmers(qs), o.count_kmers(qs))
        # the same without the side array: queries of escape lines search from scratch (the round-3 behaviour)
        b.set_table_side(0)
        assert b.table_info()["side_bytes"] == 0 and b.table_info()["escape_lines"] == info["escape_lines"]
        assert np.array_equal(b.count_kmers(qs), o.count_kmers(qs))


def test_escape_lines_of_a_real_bwt_with_repeats_cost_one_line(search_kernel):
    """A true multi-string BWT (reads of a genome with repeat families, synth.repeat_genome) large enough that EVERY line
    of a depth-5 packed table holds more than 2^16 positions: all table lookups go through the side array.  Counts equal
    the oracle's for read-derived 31-mers (matrix and fused windows) and for k == table depth; the search counters say
    that every query took the escape route, none restarted -- and that all restart once the side array is switched off."""
    import synth
    needs_plane_blocks(search_kernel)
    genome = synth.repeat_genome(200_000, 5)
    reads = synth.reads(genome, 27_000, 150, 6, 0.005)
    rle = synth.rle_encode(synth.build_msbwt_symbols(reads))
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    b.set_sparse_table(0)  # (this test is about the DIRECT table's escape lines: 31-mers must not be served by the sparse one)
    b.set_pair_index(1)
    b.set_table_depth(3)
    b.set_table_packed(1)
    info = b.table_info()
    # (the 35th line holds the last 4 of the 1024 entries: it may fit 16 bits)
    assert b.get_table_depth() == 5 and info["lines"] == 35 and info["escape_lines"] >= 33 and info["side_bytes"] == info["escape_lines"] * 512
    q31 = synth.read_kmers(reads, 31, limit=20_000, seed=3)
    q5 = synth.read_kmers(reads, 5, limit=5_000, seed=4)
    b.set_search_counters(True)
    assert np.array_equal(b.count_kmers(q31), o.count_kmers(q31))
    c = b.search_counters()
    if b.search_kernel_for(31) == "lanes":   # present k-mers: every one passes the filter and lands on an escape line
        assert 0.9 * len(q31) <= c["escape_queries"] <= len(q31) and c["escape_restarts"] == 0 and c["searched"] == len(q31)
        assert c["lane_steps"] >= 7 * len(q31)                         # seven pair steps ...
        assert c["first_lines"] == c["lane_steps"] + c["escape_queries"]  # ... and one side-array line for the escaped
    for qs in (q5, np.concatenate([q31[:1000], random_kmers(9, 1000, 31)])):
        assert np.array_equal(b.count_kmers(qs), o.count_kmers(qs))
    fwd, rc = b.count_read_kmers(reads[:300], 31, ascii=False)
    win = np.lib.stride_tricks.sliding_window_view(reads[:300], 31, axis=1).reshape(-1, 31)
    assert np.array_equal(fwd.reshape(-1), o.count_kmers(np.ascontiguousarray(win)))
    b.search_counters()
    b.set_table_side(0)
    assert np.array_equal(b.count_kmers(q31), o.count_kmers(q31))
    c = b.search_counters()
    if b.search_kernel_for(31) == "lanes":
        assert c["escape_queries"] == c["escape_restarts"] >= 0.9 * len(q31)


def test_too_large_index_is_rejected():
    """A 9-digit run encodes 32^8 = 2^40 symbols: beyond the 40-bit block counters."""
    rle = np.array([1] * 8 + [1 | (1 << 3)], dtype=np.uint8)      # A-run of exactly 2^40
    b = RleBWT()
    with pytest.raises(msbwt.MsbwtError) as e:
        b.load_vector(rle)
    assert e.value.code == msbwt._lib.ERR_TOO_LARGE
    ok = np.array([1] * 6 + [1 | (1 << 3), 2 | (3 << 3)], dtype=np.uint8)  # 32^6 = 2^30 A's, then CCC
    b.load_vector(ok)                                              # one sub-run of 2^30 symbols
    o = orc.OracleRleBWT()
    o.load_vector(ok)
    assert b.get_total_size() == o.get_total_size() == 2 ** 30 + 3
    for kmer in ([1, 1, 1], [2], [1, 2], [2, 1], [2, 2, 2], [1, 2, 2, 2]):
        assert b.count_kmer(kmer) == o.count_kmer(kmer), kmer
    assert b.count_kmer([1, 1, 1]) == 2 ** 30
    got = b.constrain_range(1, BWTRange(5, 2 ** 30 + 1))
    assert (got.l, got.h) == o.constrain_range(1, 5, 2 ** 30 + 1) == (5, 2 ** 30)


def test_concurrent_queries_from_host_threads():
    """Query entry points take `&self` in the reference: several host threads may share one
    loaded handle."""
    import threading
    reads, rle = _real_bwt(23, 150, 60)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    qs = [np.concatenate([np.array([orc.convert_stoi(r[2:2 + 21]) for r in reads], dtype=np.uint8), random_kmers(t, 2000, 21)])
          for t in range(4)]
    exp = [o.count_kmers(q) for q in qs]
    got = [None] * 4
    errs = []

    def work(i):
        try:
            for _ in range(5):
                got[i] = b.count_kmers(qs[i])
                assert b.count_kmer(qs[i][0]) == int(exp[i][0])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(4):
        assert np.array_equal(got[i], exp[i])


def test_long_launch_on_one_stream_and_many_short_ones_on_another():
    """Every launch gets tile-ticket counters that no launch still in flight uses: one long *_device
    launch on stream A, then more short launches on stream B than any fixed pool would hold, all
    before A has finished.  (A round-robin pool of 8 counter sets let the 9th short launch zero the
    long launch's counters: tiles were lost silently.)"""
    torch = pytest.importorskip("torch")
    reads, rle = _real_bwt(31, 2000, 100)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    dev = torch.device("cuda:0")
    k = 31
    present = np.array([orc.convert_stoi(r[i:i + k]) for r in reads for i in range(0, 70, 3)], dtype=np.uint8)
    long_q = np.tile(present, (max(1, 4_000_000 // len(present)), 1))          # ~4e6 present 31-mers: a launch of several ms
    short_q = [np.concatenate([present[j::17][:300], random_kmers(100 + j, 211, k)]) for j in range(24)]
    d_long = torch.from_numpy(long_q).to(dev)
    d_short = [torch.from_numpy(q).to(dev) for q in short_q]
    out_long = torch.full((len(long_q),), -1, dtype=torch.int64, device=dev)
    out_short = [torch.full((len(q),), -1, dtype=torch.int64, device=dev) for q in short_q]
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    b.count_kmers_device(d_long.data_ptr(), k, len(long_q), out_long.data_ptr(), sa.cuda_stream)
    for d_q, d_o in zip(d_short, out_short):
        b.count_kmers_device(d_q.data_ptr(), k, d_q.shape[0], d_o.data_ptr(), sb.cuda_stream)
    b.device_status(sb.cuda_stream)
    b.device_status(sa.cuda_stream)
    exp_present = o.count_kmers(present)
    assert np.array_equal(out_long.cpu().numpy().astype(np.uint64), np.tile(exp_present, len(long_q) // len(present)))
    for q, d_o in zip(short_q, out_short):
        assert np.array_equal(d_o.cpu().numpy().astype(np.uint64), o.count_kmers(q))


def test_native_allgather_of_counts_over_rccl_one_rank(search_kernel):
    """msbwt_rle_allgather_counts on REAL RCCL with the one-rank communicator a one-GPU box allows: the id /
    init / gather / destroy calls, all three wire widths, and the overflow report of the narrow ones."""
    if search_kernel != "auto":
        pytest.skip("the gather does not depend on the search kernel")
    torch = pytest.importorskip("torch")
    reads, rle = _real_bwt(41, 200, 80)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    k = 21
    qs = np.concatenate([np.array([orc.convert_stoi(r[5:5 + k]) for r in reads], dtype=np.uint8), random_kmers(5, 1000, k)])
    exp = o.count_kmers(qs)
    comm = msbwt.RankComm(1, msbwt.RankComm.unique_id(), 0)
    d_q = torch.from_numpy(qs).to(dev)
    d_mine = torch.zeros(len(qs), dtype=torch.int64, device=dev)
    b.count_kmers_device(d_q.data_ptr(), k, len(qs), d_mine.data_ptr(), stream)
    for bits in (64, 32, 16):
        d_all = torch.full((len(qs),), -1, dtype=torch.int64, device=dev)
        b.allgather_counts(comm, d_mine.data_ptr(), len(qs), d_all.data_ptr(), bits, stream)
        b.device_status(stream)
        assert np.array_equal(d_all.cpu().numpy().astype(np.uint64), exp), bits
    # a count beyond the wire width is reported, not truncated silently
    big = torch.tensor([1, 70000, 3, 2 ** 40], dtype=torch.int64, device=dev)
    out = torch.zeros(4, dtype=torch.int64, device=dev)
    b.allgather_counts(comm, big.data_ptr(), 4, out.data_ptr(), 32, stream)
    with pytest.raises(msbwt.MsbwtError) as err:
        b.device_status(stream)
    assert err.value.code == msbwt._lib.ERR_OVERFLOW
    b.allgather_counts(comm, big.data_ptr(), 4, out.data_ptr(), 64, stream)
    b.device_status(stream)
    assert out.tolist() == [1, 70000, 3, 2 ** 40]
    comm.close()


def test_pipelined_count_and_allgather_over_rccl_one_rank(search_kernel):
    """msbwt_rle_count_kmers_allgather_device on REAL RCCL (one-rank communicator): a batch counted piece by piece while the finished
    pieces' counts are gathered on a second stream -- every piece count, every wire width, counts left narrow at the destination or
    widened, ragged last pieces; all equal to the oracle."""
    if search_kernel != "auto":
        pytest.skip("the gather does not depend on the search kernel")
    torch = pytest.importorskip("torch")
    reads, rle = _real_bwt(43, 300, 80)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    k = 25
    comm = msbwt.RankComm(1, msbwt.RankComm.unique_id(), 0)
    for n in (1, 17, 1000, 70_001):
        qs = np.ascontiguousarray(np.concatenate([np.array([orc.convert_stoi(r[3:3 + k]) for r in reads], dtype=np.uint8), random_kmers(n, n, k)])[:n])
        exp = o.count_kmers(qs)
        d_q = torch.from_numpy(qs).to(dev)
        for pieces in (1, 3, 4, 64):
            for wire, out_bits in ((64, 64), (32, 64), (16, 64), (32, 32), (16, 16)):
                d_mine = torch.full((n,), -1, dtype=torch.int64, device=dev)
                d_all = torch.full((n,), -1, dtype={64: torch.int64, 32: torch.int32, 16: torch.int16}[out_bits], device=dev)
                b.count_kmers_allgather_device(comm, d_q.data_ptr(), k, n, d_mine.data_ptr(), d_all.data_ptr(), wire, out_bits, pieces, stream)
                b.device_status(stream)
                assert np.array_equal(d_mine.cpu().numpy().astype(np.uint64), exp), (n, pieces, wire)
                assert np.array_equal(d_all.cpu().numpy().astype(np.int64).astype(np.uint64), exp), (n, pieces, wire, out_bits)
    with pytest.raises(msbwt.MsbwtError):
        b.count_kmers_allgather_device(comm, d_q.data_ptr(), k, n, d_mine.data_ptr(), d_all.data_ptr(), 32, 16, 4, stream)   # widths must agree or be 64
    comm.close()


def test_batch_order_keys_and_ordered_batches(search_kernel):
    """msbwt_kmer_order_keys / _device: same keys on host and device; a batch sorted by them gives the same counts (permuted)."""
    if search_kernel != "auto":
        pytest.skip("the keys do not depend on the search kernel")
    torch = pytest.importorskip("torch")
    reads, rle = _real_bwt(51, 300, 90)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    for k in (3, 17, 21, 31, 40):
        qs = np.concatenate([np.array([orc.convert_stoi(r[i:i + k]) for r in reads for i in (0, 11, 40)], dtype=np.uint8),
                             random_kmers(k, 3000, k), random_kmers(k + 1, 200, k, alphabet=(0, 1, 2, 3, 4, 5))])
        host_keys = msbwt.rle_bwt.kmer_order_keys(qs)
        d_q = torch.from_numpy(qs).to(dev)
        d_keys = torch.zeros(len(qs), dtype=torch.int64, device=dev)
        b.kmer_order_keys_device(d_q.data_ptr(), k, len(qs), d_keys.data_ptr(), stream)
        torch.cuda.synchronize(dev)
        assert np.array_equal(d_keys.cpu().numpy().view(np.uint64), host_keys)
        has_other = np.isin(qs[:, max(0, k - 31):], (0, 4)).any(axis=1)
        assert np.array_equal(host_keys == np.uint64(2**64 - 1), has_other)
        order = np.argsort(host_keys, kind="stable")
        exp = o.count_kmers(qs)
        assert np.array_equal(b.count_kmers(qs[order]), exp[order])


def _numpy_pack_2bit(qs):
    """the packed layout restated with numpy: the k-mer as a base-4 number, first symbol most significant, 32 symbols per word"""
    n, k = qs.shape
    code = np.where(qs == 5, 3, qs.astype(np.int64) - 1).astype(np.uint64)
    words = np.zeros((n, 2 if k > 32 else 1), dtype=np.uint64)
    for t in range(k):   # step t = symbol k-1-t
        words[:, t >> 5] |= code[:, k - 1 - t] << np.uint64(2 * (t & 31))
    return words


@pytest.mark.parametrize("k", [1, 5, 12, 17, 21, 31, 32, 33, 48, 64])
def test_packed_two_bit_queries_count_like_the_byte_form(k, search_kernel):
    """msbwt_rle_count_kmers_packed[_device]: 2-bit queries give the counts of the same k-mers as symbol codes (= the oracle's),
    with 64- and 32-bit outputs, through the host pipeline and the device entry point, at batch sizes around the tile size."""
    torch = pytest.importorskip("torch")
    reads, rle = _real_bwt(7, 200, 80)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    wins = [orc.convert_stoi(r[i:i + k]) for r in reads for i in (0, 3, 16) if len(r) >= i + k]
    qs = np.array([w for w in wins if not np.isin(w, (0, 4)).any()], dtype=np.uint8).reshape(-1, k)
    qs = np.concatenate([qs, random_kmers(k, 2000, k)])
    words = msbwt.rle_bwt.pack_2bit(qs)
    assert np.array_equal(words, _numpy_pack_2bit(qs))
    exp = o.count_kmers(qs)
    assert np.array_equal(b.count_kmers_packed(words, k), exp)
    got32 = b.count_kmers_packed(words, k, count_bits=32)
    assert got32.dtype == np.uint32 and np.array_equal(got32.astype(np.uint64), exp)
    assert exp[:50].sum() > 0
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    d_w = torch.from_numpy(words.view(np.int64)).to(dev)
    for n in (1, 63, 64, 65, 1000, len(qs)):
        out = torch.full((n + 1,), -7, dtype=torch.int64, device=dev)
        b.count_kmers_packed_device(d_w.data_ptr(), k, n, out.data_ptr(), stream)
        b.device_status(stream)
        got = out.cpu().numpy()
        assert got[n] == -7 and np.array_equal(got[:n].astype(np.uint64), exp[:n]), n
    with pytest.raises(msbwt.MsbwtError):
        msbwt.rle_bwt.pack_2bit(np.array([[1, 2, 4]], dtype=np.uint8))


@pytest.mark.parametrize("bits", [7, 10, 13, 22, 26])
def test_in_library_batch_order_never_changes_results(bits, search_kernel, monkeypatch):
    """msbwt_rle_set_batch_order(1): the library packs and bucket-orders the batch on the device (one global pass of up to 10
    key bits, then up to two passes of 12 inside the buckets), counts it in index order and writes every count to its query's own place -- the caller sees the counts of
    the unordered run (= the oracle's), for rows of symbol codes (with '$' / 'N' rows: the exception list) and for 2-bit
    queries, through host and device entry points."""
    import synth
    torch = pytest.importorskip("torch")
    needs_plane_blocks(search_kernel)
    monkeypatch.setenv("MSBWT_ORDER_BITS", str(bits))
    genome = synth.repeat_genome(40_000, 9)
    reads = synth.reads(genome, 8_000, 100, 10, 0.01)
    rle = synth.rle_encode(synth.build_msbwt_symbols(reads))
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    b.set_pair_index(1)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    rng = np.random.default_rng(bits)
    for k in (12, 21, 31, 32, 33, 64):
        qs = np.concatenate([synth.read_kmers(reads, k, limit=30_000, seed=k), random_kmers(k, 10_000, k)])
        rng.shuffle(qs)
        other = rng.integers(0, len(qs), size=500)                      # rows that two bits cannot say
        qs[other, rng.integers(0, k, size=500)] = rng.choice([0, 4], size=500)
        exp = o.count_kmers(qs, nthreads=4)
        b.set_batch_order(0)
        assert not b.batch_order_for(k, len(qs))
        plain = b.count_kmers(qs)
        b.set_batch_order(1)
        assert b.batch_order_for(k, len(qs)) == (b.search_kernel_for(k) == "lanes")
        assert np.array_equal(plain, exp) and np.array_equal(b.count_kmers(qs), exp), k
        d_q = torch.from_numpy(qs).to(dev)
        out = torch.full((len(qs) + 1,), -7, dtype=torch.int64, device=dev)
        b.count_kmers_device(d_q.data_ptr(), k, len(qs), out.data_ptr(), stream)
        b.device_status(stream)
        got = out.cpu().numpy()
        assert got[-1] == -7 and np.array_equal(got[:-1].astype(np.uint64), exp), k
        acgt = ~np.isin(qs, (0, 4)).any(axis=1)
        words = msbwt.rle_bwt.pack_2bit(qs[acgt])
        assert np.array_equal(b.count_kmers_packed(words, k), exp[acgt]), k
    # an invalid code is still reported, and only its own row is affected
    bad = qs.copy()
    bad[17, 3] = 7
    b.set_batch_order(1)
    with pytest.raises(msbwt.MsbwtError) as e:
        b.count_kmers(bad)
    assert e.value.code == msbwt._lib.ERR_INVALID_SYMBOL


def test_introspection(search_kernel):
    needs_plane_blocks(search_kernel)
    rle = random_stream(2, 30000, "short")
    b = gpu_bwt(rle)
    blocks = b.get_total_size() // 256 + 1
    b.set_pair_index(0)
    b.set_table_depth(0)
    assert b.device_bytes() == blocks * 128 and b.get_table_depth() == 0 and not b.get_pair_index()
    assert b.get_presence_filter() == 0
    b.set_table_depth(4)
    assert b.device_bytes() == blocks * 128 + 16 * 4 ** 4 and b.get_presence_filter() == 0  # table too shallow
    b.set_pair_index(1)
    assert b.get_pair_index() and b.device_bytes() > blocks * 128 * 3
    assert b.device_ordinal() == 0
    assert "gfx950" in msbwt.version()
    # which kernel a batch runs on (bench.py labels its roofline with it)
    b.set_search_kernel("auto")  # with a pair index everything but the shortest k-mers goes to the lanes kernel
    assert b.search_kernel_for(31) == "lanes" and b.search_kernel_for(8) == "lanes" and b.search_kernel_for(4) == "groups"
    assert b.search_kernel_for(70) == "generic"
    b.set_search_kernel("groups")
    assert b.search_kernel_for(31) == "groups"
    b.set_search_kernel("lanes")
    assert b.search_kernel_for(8) == "lanes" and b.search_kernel_for(65) == "generic"
    b.set_pair_index(0)
    b.set_search_kernel("auto")
    assert b.search_kernel_for(31) == "groups"


def test_sharded_counter_single_gpu_worker():
    """ShardedCounter with its default (GPU) worker, world size 1: the per-rank path of the
    multi-GPU bench."""
    torch = pytest.importorskip("torch")
    from rust_msbwt_amd.sharded import ShardedCounter, as_u64
    reads, rle = _real_bwt(29, 100, 50)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    q = np.concatenate([np.array([orc.convert_stoi(r[1:1 + 25]) for r in reads], dtype=np.uint8), random_kmers(5, 999, 25)])
    counter = ShardedCounter(bwt=b)
    got = counter.count_kmers(torch.from_numpy(q).to("cuda:0"))
    torch.cuda.synchronize()
    assert np.array_equal(as_u64(got), o.count_kmers(q))


@pytest.mark.parametrize("filt", [0, 1])
def test_presence_filter_never_changes_results(filt):
    """The L2-resident presence bitmap in front of the suffix table: on a small genome most random
    k-mers are rejected by it, and counts stay identical with and without it."""
    reads, rle = _real_bwt(31, 300, 64)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    for depth in (6, 8, 9):
        b.set_table_depth(depth)
        b.set_presence_filter(filt)
        assert (b.get_presence_filter() == depth) == bool(filt)      # sparse enough to be kept
        for k in (depth, depth + 1, 21, 31, 40):
            qs = [orc.convert_stoi(r[p:p + k]) for r in reads if len(r) >= k for p in (int(np.random.default_rng(k).integers(0, len(r) - k + 1)),)]
            qs = np.concatenate([np.array(qs, dtype=np.uint8), random_kmers(k, 3000, k),
                                 random_kmers(k + 3, 300, k, alphabet=(0, 1, 2, 3, 4, 5))])
            assert np.array_equal(b.count_kmers(qs), o.count_kmers(qs)), (filt, depth, k)
    # a saturated index (every short suffix present) drops the filter by itself
    dense = gpu_bwt(random_stream(5, 200000, "short", alphabet=(1, 2, 3, 5)))
    dense.set_table_depth(6)
    assert dense.get_presence_filter() == 0


def test_replicas_and_sharded_batches_match_a_single_handle(monkeypatch):
    monkeypatch.setenv("MSBWT_FORCE_PEER_COPIES", "1")  # shards of replicas 1.. go through hipMemcpyPeerAsync staging
    """The C ABI's multi-device entry points, rehearsed on this box's one GPU (devices = {0, 0, 0}):
    replicas are GPU -> GPU copies of the loaded index, a batch is sharded over them, and the counts
    must equal a single handle's (and the oracle's) bit for bit -- host and device forms."""
    import torch
    from rust_msbwt_amd import rle_bwt
    reads, rle = _real_bwt(77, 300, 80)
    reads = [r for r in reads if "N" not in r]
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    replicas = [b, b.replicate(0), b.replicate(0)]
    assert all(r.get_total_size() == b.get_total_size() and r.get_table_depth() == b.get_table_depth()
               and r.get_pair_index() == b.get_pair_index() for r in replicas)
    rng = np.random.default_rng(5)
    for k in (4, 21, 31, 40):
        present = np.array([orc.convert_stoi(r[i:i + k]) for r in reads for i in (0, 11, 23) if i + k <= len(r)], dtype=np.uint8)
        absent = np.array([1, 2, 3, 5], dtype=np.uint8)[rng.integers(0, 4, size=(1003, k))]
        q = np.concatenate([present, absent])
        exp = o.count_kmers(q)
        assert np.array_equal(rle_bwt.count_kmers_multi(replicas, q), exp)
        assert np.array_equal(rle_bwt.count_kmers_multi(replicas[:2], q[:5]), exp[:5])  # fewer queries than one shard unit
        dev = torch.device("cuda", 0)
        d_q = torch.from_numpy(q).to(dev)
        d_out = torch.zeros(len(q), dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        rle_bwt.count_kmers_multi_device(replicas, d_q.data_ptr(), k, len(q), d_out.data_ptr())
        assert np.array_equal(d_out.cpu().numpy().astype(np.uint64), exp)
    codes = np.array([orc.convert_stoi(r) for r in reads], dtype=np.uint8)
    f1, r1 = b.count_read_kmers(codes, 25, ascii=False, revcomp=True)
    f3, r3 = rle_bwt.count_read_kmers_multi(replicas, codes, 25, ascii=False, revcomp=True)
    assert np.array_equal(f1, f3) and np.array_equal(r1, r3)
    # an invalid symbol in one shard is reported by the whole call
    bad = np.concatenate([q[:100], np.full((1, q.shape[1]), 7, dtype=np.uint8), q[100:]])
    with pytest.raises(msbwt.MsbwtError) as err:
        rle_bwt.count_kmers_multi(replicas, bad)
    assert err.value.code == msbwt._lib.ERR_INVALID_SYMBOL


@pytest.mark.parametrize("kind", ["ones", "short", "long", "mixed", "raw", "edges"])
def test_run_blocks_built_on_the_device_equal_the_host_built_ones(kind, monkeypatch):
    """The run-block index is made on the device from plane blocks (csrc/run_build.hip; MSBWT_BUILD=host keeps the host builder):
    both answer every rank alike -- constrain_ranges at random positions, at every multiple of 256 and at T itself, on streams of
    single-symbol runs (every block overflows), long runs, zero-length runs, totals that are exact multiples of the block sizes."""
    monkeypatch.setenv("MSBWT_BLOCKS", "runs")
    streams = {"ones": [random_stream(3, 4000, "ones")], "short": [random_stream(4, 6000, "short")], "long": [random_stream(5, 60, "long")],
               "mixed": [random_stream(6, 400, "mixed")], "raw": [raw_byte_stream(7, 3000)],
               "edges": [runs_to_bytes([1, 2], [256, 256]), runs_to_bytes([3], [512]), runs_to_bytes([0, 5], [1, 1023]), runs_to_bytes([2] * 1, [1]),
                         runs_to_bytes(list(range(1, 6)) * 120, [1] * 600), np.zeros(0, dtype=np.uint8)]}[kind]
    for rle in streams:
        o = orc.OracleRleBWT()
        o.load_vector(rle)
        total = o.get_total_size()
        monkeypatch.setenv("MSBWT_BUILD", "device")
        dev_built = gpu_bwt(rle)
        monkeypatch.setenv("MSBWT_BUILD", "host")
        host_built = gpu_bwt(rle)
        assert dev_built.get_block_format() == host_built.get_block_format() == "runs" and dev_built.get_total_size() == total
        rng = np.random.default_rng(len(rle))
        pos = np.unique(np.concatenate([rng.integers(0, total + 1, size=3000), np.arange(0, total + 1, 256)[:4000], [0, total]]))
        l, h = rng.choice(pos, size=6000), rng.choice(pos, size=6000)
        l, h = np.minimum(l, h).astype(np.uint64), np.maximum(l, h).astype(np.uint64)
        sy = rng.integers(0, 6, size=6000).astype(np.uint8)
        exp = o.constrain_ranges(sy, l, h)
        for b in (dev_built, host_built):
            got = b.constrain_ranges(sy, l, h)
            assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
        if total:
            qs = random_kmers(5, 3000, 9, alphabet=(0, 1, 2, 3, 4, 5))
            assert np.array_equal(dev_built.count_kmers(qs), o.count_kmers(qs)) and np.array_equal(host_built.count_kmers(qs), o.count_kmers(qs))


def test_run_block_format_is_selectable_and_lean():
    """msbwt_rle_set_block_format: the same handle loads the same stream as plane blocks and as run
    blocks; identical answers, about 0.3 instead of 0.5 bytes per symbol, no pair index in run mode."""
    import synth
    rle, total = synth.rle_stream(30_000_000, 6.0, 5)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    q1, q2 = random_kmers(3, 20000, 25), random_kmers(4, 3000, 9, alphabet=(0, 1, 2, 3, 4, 5))
    exp = o.count_kmers(q1), o.count_kmers(q2)
    sizes = {}
    for fmt in ("planes", "runs"):
        b = RleBWT()
        b.set_block_format(fmt)
        b.set_table_depth(0)
        b.set_pair_index(0)
        b.load_vector(rle)
        assert b.get_block_format() == fmt and b.get_total_size() == total
        info = b.sparse_table_info()   # (the `sparse` pass of this file asks for a depth-16 table explicitly: run blocks get one too since round 6 -- not counted here)
        sizes[fmt] = b.device_bytes() - info["bytes"] - info["side_bytes"] - info["second_bytes"]
        assert np.array_equal(b.count_kmers(q1), exp[0])
        assert np.array_equal(b.count_kmers(q2), exp[1])
        b.set_table_depth(6)
        b.set_pair_index(1)
        assert b.get_pair_index() == (fmt == "planes")
        assert np.array_equal(b.count_kmers(q1), exp[0])
        b.set_line_streaming(1)   # (non-temporal line loads: the automatic choice of indexes of 4 GiB and more, forced here)
        assert b.get_line_streaming() and np.array_equal(b.count_kmers(q1), exp[0])
    assert 0.25 * total < sizes["runs"] < 0.36 * total and 0.49 * total < sizes["planes"] < 0.51 * total


def test_run_blocks_fall_back_to_the_host_builder_when_the_device_peak_does_not_fit(monkeypatch):
    """The device builder of the lean format holds the plane blocks beside the run blocks for a moment (0.8 byte per symbol);
    when that does not fit the free HBM -- here: MSBWT_RUN_BUILD_FREE pretends it does not -- the load builds the run blocks on
    the host instead of failing, and the index is the same one."""
    import synth
    rle, total = synth.rle_stream(20_000_000, 6.0, 9)
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    q = random_kmers(8, 30000, 21)
    exp = o.count_kmers(q)
    monkeypatch.setenv("MSBWT_BLOCKS", "runs")
    sizes = []
    for free in (None, str(int(0.6 * total))):   # all of the HBM; enough for the finished index (0.3 byte per symbol), not for its builder
        if free is None:
            monkeypatch.delenv("MSBWT_RUN_BUILD_FREE", raising=False)
        else:
            monkeypatch.setenv("MSBWT_RUN_BUILD_FREE", free)
        b = RleBWT()
        b.load_vector(rle)
        assert b.get_block_format() == "runs" and b.get_total_size() == total
        assert np.array_equal(b.count_kmers(q), exp)
        info = b.sparse_table_info()   # (a sparse table -- the `sparse` pass asks for one -- is built from the device builder's plane blocks only)
        assert free is None or info["depth"] == 0
        sizes.append(b.device_bytes() - info["bytes"] - info["side_bytes"] - info["second_bytes"])
    assert sizes[0] == sizes[1]
    assert msbwt._lib.lib().msbwt_run_build_fits_device(total, int(0.6 * total)) == 0


def test_two_host_threads_on_their_per_thread_streams(search_kernel):
    """hipStreamPerThread is ONE handle value that stands for a different queue in every host thread: launches two threads make
    'on the same stream' may run side by side, so they must not share a block of tile-ticket counters.  Two threads,
    many launches each with more tiles than resident waves, every count against the oracle."""
    import threading
    import torch
    needs_plane_blocks(search_kernel)
    rng = np.random.default_rng(12)
    genome = np.array([1, 2, 3, 5], dtype=np.uint8)[rng.integers(0, 4, size=30000)]
    reads = ["".join("$ACGNT"[c] for c in genome[s:s + 60]) for s in rng.integers(0, len(genome) - 60, size=3000)]
    rle = orc.convert_to_vec(orc.naive_bwt(reads))
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    per_thread = 2   # hipStreamPerThread
    dev = torch.device("cuda:0")
    failures = []

    def worker(seed):
        r = np.random.default_rng(seed)
        win = np.array([[("$ACGNT".index(ch)) for ch in reads[i][p:p + 31]] for i, p in zip(r.integers(0, len(reads), size=4000), r.integers(0, 30, size=4000))], dtype=np.uint8)
        q = np.ascontiguousarray(np.concatenate([win] * 60 + [random_kmers(seed, 20000, 31)]))   # 260 000 queries: 4 063 tiles > 3 072 resident waves
        exp = o.count_kmers(q)
        d_q = torch.from_numpy(q).to(dev)
        for _ in range(15):
            d_out = torch.full((len(q),), -1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            b.count_kmers_device(d_q.data_ptr(), 31, len(q), d_out.data_ptr(), per_thread)
            b.device_status(per_thread)
            if not np.array_equal(d_out.cpu().numpy().astype(np.uint64), exp):
                failures.append(seed)
                return

    threads = [threading.Thread(target=worker, args=(s,)) for s in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures


@pytest.mark.parametrize("fraction", [0.9, 0.5, 0.2])
def test_memory_budget_on_a_loaded_index(fraction, search_kernel):
    """msbwt_rle_set_memory_budget on a LOADED index: the optional structures are rebuilt under the budget (pair blocks, direct and
    sparse suffix tables), the index holds no more than the budget says (as long as that is at least the plane blocks), counts
    stay the oracle's, and budget 0 restores the default index."""
    import synth
    needs_plane_blocks(search_kernel)
    genome = synth.genome(300_000, 3)
    reads = synth.reads(genome, 60_000, 100, 4, 0.005)
    rle = synth.rle_encode(synth.build_msbwt_symbols(reads))
    o = orc.OracleRleBWT()
    o.load_vector(rle)
    b = gpu_bwt(rle)
    full = b.device_bytes()
    planes = (b.get_total_size() // 256 + 1) * 128
    shape = (b.get_pair_index(), b.get_pair_stride(), b.get_table_depth(), b.get_sparse_table())
    qs = [np.ascontiguousarray(np.concatenate([synth.read_kmers(reads, k, limit=20_000, seed=k), random_kmers(k, 5_000, k)])) for k in (12, 21, 31)]
    exp = [o.count_kmers(q) for q in qs]
    budget = max(int(full * fraction), planes + (1 << 20))
    b.set_memory_budget(budget)
    assert b.get_memory_budget() == budget
    held = b.device_bytes()
    assert held <= budget + (4 << 20), (held, budget)   # (+ the presence filter, 2 MiB at most, and the side arrays)
    assert held < full or fraction >= 0.9
    for q, e in zip(qs, exp):
        assert np.array_equal(b.count_kmers(q), e)
    twin = b.replicate(b.device_ordinal())   # a replica carries the plan
    assert twin.device_bytes() == held
    assert np.array_equal(twin.count_kmers_packed(msbwt.rle_bwt.pack_2bit(qs[2][np.isin(qs[2], [1, 2, 3, 5]).all(axis=1)]), 31),
                          o.count_kmers(qs[2][np.isin(qs[2], [1, 2, 3, 5]).all(axis=1)]))
    twin.set_batch_order(1)
    assert np.array_equal(twin.count_kmers(qs[2]), exp[2])
    b.set_memory_budget(0)
    assert b.device_bytes() == full and (b.get_pair_index(), b.get_pair_stride(), b.get_table_depth(), b.get_sparse_table()) == shape
    for q, e in zip(qs, exp):
        assert np.array_equal(b.count_kmers(q), e)
