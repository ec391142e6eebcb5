"""Synthetic workloads (test / bench infrastructure): genome, reads, MSBWT, queries.

SURVEY.md 8(d) generators, seeded and integer-only; `workload(name)` builds (and caches
under synth/cache/) the inputs of a BASELINE.json config."""
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmsbwt_synth.so")
SRC = os.path.join(HERE, "msbwt_synth.cpp")
CACHE = os.path.join(HERE, "cache")

_lib = None


def build(force=False):
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O3", "-march=x86-64-v3", "-std=c++17", "-fPIC", "-shared", "-Wall",
                               "-Wextra", "-o", SO, SRC, "-lpthread"])
    return SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(SO)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.synth_genome.argtypes = [u64, u64, vp]
        L.synth_reads.argtypes = [vp, u64, u64, u32, u64, u32, vp]
        L.synth_repeat_genome.argtypes = [u64, u64, u32, vp, vp, vp, vp, vp, u64, u32, u32, u64, vp, vp]
        L.synth_random_kmers.argtypes = [u64, u32, u64, vp]
        L.synth_build_msbwt.argtypes = [vp, vp, u64, vp, C.c_int]
        L.synth_build_msbwt.restype = C.c_int
        L.synth_rle_encode.argtypes = [vp, u64, vp, u64]
        L.synth_rle_encode.restype = u64
        L.synth_rle_stream.argtypes = [u64, C.c_double, u64, C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.synth_rle_stream.restype = vp
        L.synth_run_histogram.argtypes = [vp, u64, vp, u64]
        L.synth_rle_stream_hist.argtypes = [u64, vp, vp, C.c_double, u64, C.c_int, C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.synth_rle_stream_hist.restype = vp
        L.synth_set_threads.argtypes = [C.c_int]
        L.synth_free.argtypes = [vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def genome(length, seed):
    out = np.empty(length, dtype=np.uint8)
    lib().synth_genome(length, seed, _p(out))
    return out


# Repeat model of the `c4r` workload, quoted per 64 444 167 bases (config C4's genome; other sizes scale the copy numbers and
# the tandem-repeat bases): the human genome's repeat classes at its own proportions -- Alu-like SINEs ~10 % of the bases, L1-like
# 5'-truncated LINEs ~17 %, LTR and DNA elements ~10 %, old MIR-like copies ~4 %, recent segmental duplications ~3 %, satellite
# arrays ~3 %, microsatellites ~1 % -- with per-copy divergence between 0 and 30 %.  Copy numbers are the human ones divided by
# ~50 (the genome is 1/48 of a human one): what a 17-mer of a young Alu copy does to a 30x read set -- tens of thousands of
# occurrences -- it still does here (22 000 copies x 27 covering windows x the share of copies that kept the 17-mer intact).
REPEAT_FAMILIES = [
    # (name, consensus length, copies per 64.4 Mbp, divergence low, high, 5'-truncated)
    ("alu_young", 300, 6_000, 0.01, 0.05, False),
    ("alu_old", 300, 24_000, 0.05, 0.15, False),
    ("l1", 6000, 4_500, 0.01, 0.20, True),
    ("ltr_a", 1000, 1_500, 0.05, 0.20, False),
    ("ltr_b", 1200, 1_500, 0.05, 0.20, False),
    ("ltr_c", 800, 1_500, 0.05, 0.20, False),
    ("dna_a", 800, 1_200, 0.10, 0.20, False),
    ("dna_b", 900, 1_200, 0.10, 0.20, False),
    ("mir_old", 260, 10_000, 0.20, 0.30, False),
    ("segdup", 20_000, 100, 0.00, 0.02, False),
]
REPEAT_SATELLITE = dict(bases=2_000_000, monomer=171, div=0.02)
REPEAT_MICRO_BASES = 640_000
REPEAT_REF_GENOME = 64_444_167


def repeat_genome(length, seed, with_classes=False):
    """A genome with interspersed repeat families, satellite arrays and microsatellites (model above, scaled to `length`)."""
    scale = length / REPEAT_REF_GENOME
    fams = [f for f in REPEAT_FAMILIES if f[1] < length // 4]
    n = len(fams)
    lens = np.array([f[1] for f in fams], dtype=np.uint32)
    copies = np.array([max(1, int(round(f[2] * scale))) for f in fams], dtype=np.uint32)
    dlo = np.array([int(f[3] * 1e6) for f in fams], dtype=np.uint32)
    dhi = np.array([int(f[4] * 1e6) for f in fams], dtype=np.uint32)
    trunc = np.array([1 if f[5] else 0 for f in fams], dtype=np.uint8)
    out = np.empty(length, dtype=np.uint8)
    cls = np.empty(length, dtype=np.uint8) if with_classes else None
    lib().synth_repeat_genome(length, seed, n, _p(lens), _p(copies), _p(dlo), _p(dhi), _p(trunc),
                              int(REPEAT_SATELLITE["bases"] * scale), REPEAT_SATELLITE["monomer"] if length > 4 * REPEAT_SATELLITE["monomer"] else 0,
                              int(REPEAT_SATELLITE["div"] * 1e6), int(REPEAT_MICRO_BASES * scale), _p(out), _p(cls) if with_classes else None)
    return (out, cls) if with_classes else out


def reads(genome_codes, n, length, seed, err=0.005):
    out = np.empty((n, length), dtype=np.uint8)
    g = np.ascontiguousarray(genome_codes, dtype=np.uint8)
    lib().synth_reads(_p(g), g.size, n, length, seed, int(round(err * 1e6)), _p(out))
    return out


def random_kmers(n, k, seed):
    out = np.empty((n, k), dtype=np.uint8)
    lib().synth_random_kmers(n, k, seed, _p(out))
    return out


def read_kmers(read_codes, k, limit=None, seed=0):
    """k-mer windows of the reads: all of them, or `limit` sampled uniformly (seeded)."""
    n, length = read_codes.shape
    per = length - k + 1
    if limit is None or limit >= n * per:
        win = np.lib.stride_tricks.sliding_window_view(read_codes, k, axis=1)
        return np.ascontiguousarray(win.reshape(-1, k))
    rng = np.random.default_rng(seed)
    r = rng.integers(0, n, size=limit)
    p = rng.integers(0, per, size=limit)
    idx = p[:, None] + np.arange(k)[None, :]
    return np.ascontiguousarray(read_codes[r[:, None], idx])


def build_msbwt_symbols(read_list_or_array, threads=0):
    """BWT symbol codes (one per position) of a read set, in the reference's ordering."""
    if isinstance(read_list_or_array, np.ndarray) and read_list_or_array.ndim == 2:
        n, length = read_list_or_array.shape
        flat = np.ascontiguousarray(read_list_or_array, dtype=np.uint8).reshape(-1)
        offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(length))
    else:
        arrs = [np.asarray(r, dtype=np.uint8) for r in read_list_or_array]
        n = len(arrs)
        flat = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.uint8)
        offsets = np.concatenate([[0], np.cumsum([len(a) for a in arrs])]).astype(np.uint64)
    out = np.empty(int(offsets[-1]) + n, dtype=np.uint8)
    rc = lib().synth_build_msbwt(_p(flat), _p(offsets), n, _p(out), threads)
    if rc:
        raise ValueError("synth_build_msbwt failed (symbols must be codes 1..5)")
    return out


def rle_encode(symbols):
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    need = lib().synth_rle_encode(_p(s), s.size, None, 0)
    out = np.empty(need, dtype=np.uint8)
    lib().synth_rle_encode(_p(s), s.size, _p(out), need)
    return out


def set_threads(threads):
    """Host threads the generators use (0 = default: min(16, hardware)); bench.py bounds them per rank."""
    lib().synth_set_threads(int(threads))


HISTOGRAM_FILE = os.path.join(HERE, "c4_run_histogram.json")
RUN_HIST_CAP = 65536


def run_histogram(rle_bytes, cap=RUN_HIST_CAP):
    """Per-symbol run-length histogram of an RLE stream: (6, cap) uint64, lengths >= cap in the last bin."""
    a = np.ascontiguousarray(rle_bytes, dtype=np.uint8)
    hist = np.zeros((6, cap), dtype=np.uint64)
    lib().synth_run_histogram(_p(a), a.size, _p(hist), cap)
    return hist


def load_histogram(path=HISTOGRAM_FILE):
    """The committed measurement (tools/measure_run_histogram.py): {"lengths": {symbol: {run length: runs}}} ->
    (len_table uint16[6, 65536], sym_cdf uint32[6], mean run)."""
    import json
    with open(path) as f:
        data = json.load(f)
    table = np.ones((6, 65536), dtype=np.uint16)
    runs_of = np.zeros(6, dtype=np.float64)
    symbols = 0.0
    for s in range(6):
        items = sorted((int(l), int(c)) for l, c in data["lengths"].get(str(s), {}).items())
        if not items:
            continue
        lens = np.array([l for l, _ in items], dtype=np.int64)
        cnts = np.array([c for _, c in items], dtype=np.float64)
        runs_of[s] = cnts.sum()
        symbols += float((lens * cnts).sum())
        cdf = np.cumsum(cnts) / cnts.sum()
        # inverse CDF at (u + 0.5) / 65536
        table[s] = np.minimum(lens[np.searchsorted(cdf, (np.arange(65536) + 0.5) / 65536.0, side="left").clip(0, len(lens) - 1)], 65535).astype(np.uint16)
    cdf = np.cumsum(runs_of) / runs_of.sum()
    sym_cdf = np.minimum(np.floor(cdf * 4294967296.0), 4294967295.0).astype(np.uint32)
    sym_cdf[5] = 4294967295
    return np.ascontiguousarray(table), sym_cdf, symbols / runs_of.sum()


def rle_stream(target_symbols, mean_run, seed, chunks=0, histogram=None):
    """Structure-equivalent synthetic RLE stream (NOT a real BWT). Returns (bytes, total).
    Deterministic for a given (target, mean_run, seed, chunks).  histogram = path of a measured run-length
    histogram (load_histogram): run lengths and symbols are drawn from it instead of a geometric law."""
    if chunks <= 0:  # ~64 Mi symbols per chunk, at least 64 chunks (fixed rule => reproducible)
        chunks = max(64, int(target_symbols) >> 26)
    nbytes, total = C.c_uint64(), C.c_uint64()
    if histogram is not None:
        table, sym_cdf, mean = load_histogram(histogram)
        ptr = lib().synth_rle_stream_hist(target_symbols, _p(table), _p(sym_cdf), mean, seed, chunks, 0, C.byref(nbytes), C.byref(total))
    else:
        ptr = lib().synth_rle_stream(target_symbols, mean_run, seed, chunks, C.byref(nbytes), C.byref(total))
    if not ptr:
        raise MemoryError("synth_rle_stream")
    # wrap the malloc'ed buffer without copying it (it is 15 GB at human scale)
    out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(int(nbytes.value), 1),))[:int(nbytes.value)]
    weakref.finalize(out, lib().synth_free, ptr)
    return out, int(total.value)


def write_npy(path, rle_bytes):
    """The crate's 96-byte NumPy v1.0 header + payload (bwt_converter.rs:102-130 format)."""
    head = b"\x93NUMPY\x01\x00\x56\x00" + ("{'descr': '|u1', 'fortran_order': False, 'shape': (%d, ), }" % len(rle_bytes)).encode()
    head = head + b" " * (95 - len(head)) + b"\n"
    with open(path, "wb") as f:
        f.write(head)
        f.write(np.ascontiguousarray(rle_bytes, dtype=np.uint8).tobytes())


# BASELINE.json configs made concrete.  scale < 1 shrinks a config for tests.
CONFIGS = {
    "c2": dict(genome=3_333_333, gseed=1, nreads=1_000_000, rlen=100, rseed=2, err=0.005, k=21,
               nq=10_000_000, qseed=3, queries="random"),
    "c3": dict(genome=4_641_652, gseed=11, nreads=1_547_217, rlen=150, rseed=12, err=0.005, k=31,
               nq=None, qseed=13, queries="reads"),
    "c4": dict(genome=64_444_167, gseed=21, nreads=12_888_833, rlen=150, rseed=22, err=0.005, k=31,
               nq=100_000_000, qseed=23, queries="random"),
    # not a BASELINE config: three times C4 (5.84e9 symbols), the largest real MSBWT (reads with substitutions, suffix-sorted)
    # that the GPU box's host share builds within its memory cap (~205 GB) -- a size point between C4 and human scale
    # not a BASELINE config either: C4's size and read set on a genome WITH REPEATS (repeat_genome above) -- what a human-like
    # read set does to the search: wide ranges, high-copy 17-mers (the packed table's escape lines), early exits
    "c4r": dict(genome=64_444_167, gseed=41, nreads=12_888_833, rlen=150, rseed=42, err=0.005, k=31,
                nq=100_000_000, qseed=43, queries="reads", repeats=True),
    # ... and three times that (the size of c4x3): copy numbers three times higher -- more of the packed table's lines escape
    "c4x3r": dict(genome=193_332_501, gseed=51, nreads=38_666_499, rlen=150, rseed=52, err=0.005, k=31,
                  nq=100_000_000, qseed=53, queries="reads", repeats=True),
    "c4x3": dict(genome=193_332_501, gseed=31, nreads=38_666_499, rlen=150, rseed=32, err=0.005, k=31,
                 nq=100_000_000, qseed=33, queries="reads"),
}


def workload_index(name, scale=1.0, cache=True, threads=0):
    """Returns (path of comp_msbwt.npy, reads array) for a config, building it if needed."""
    cfg = dict(CONFIGS[name])
    g = max(int(cfg["genome"] * scale), 4 * cfg["rlen"])
    n = max(int(cfg["nreads"] * scale), 4)
    tag = "%s_g%d_n%d_l%d" % (name, g, n, cfg["rlen"])
    os.makedirs(CACHE, exist_ok=True)
    npy = os.path.join(CACHE, tag + "_comp_msbwt.npy")
    rd = reads(repeat_genome(g, cfg["gseed"]) if cfg.get("repeats") else genome(g, cfg["gseed"]), n, cfg["rlen"], cfg["rseed"], cfg["err"])
    if not (cache and os.path.exists(npy)):
        sym = build_msbwt_symbols(rd, threads)
        write_npy(npy, rle_encode(sym))
    return npy, rd
