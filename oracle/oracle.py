"""ctypes view of oracle/libmsbwt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (rust-msbwt_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmsbwt_oracle.so")

OK, ERR_IO, ERR_EOF, ERR_HEADER, ERR_SYMBOL, ERR_RANGE = 0, -1, -2, -3, -4, -5


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = [os.path.join(_HERE, f) for f in ("msbwt_oracle.c", "msbwt_oracle.h")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmsbwt_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class Stats(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("steps", C.c_uint64), ("visits", C.c_uint64),
                ("scan_bytes", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}

    def algorithmic_bytes(self, k):
        """sum over visits of (56 + scanned RLE bytes) + k + 8 per query."""
        return 56 * self.visits + self.scan_bytes + (k + 8) * self.queries


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    u8p, u64p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.c_void_p
    sig = {
        "orc_rle_new": (vp, [C.c_uint8]),
        "orc_rle_free": (None, [vp]),
        "orc_rle_load_vector": (C.c_int, [vp, vp, C.c_size_t]),
        "orc_rle_load_numpy_file": (C.c_int, [vp, C.c_char_p]),
        "orc_rle_get_symbol_count": (C.c_uint64, [vp, C.c_uint8]),
        "orc_rle_get_total_size": (C.c_uint64, [vp]),
        "orc_rle_constrain_range": (C.c_int, [vp, C.c_uint8, C.c_uint64, C.c_uint64, u64p, u64p, vp]),
        "orc_rle_count_kmer": (C.c_int, [vp, vp, C.c_size_t, u64p, vp]),
        "orc_rle_count_kmers": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_int, vp]),
        "orc_rle_constrain_ranges": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, vp]),
        "orc_rle_index_length": (C.c_size_t, [vp]),
        "orc_rle_ref_index": (u64p, [vp]),
        "orc_rle_fm_index": (u64p, [vp, C.c_int]),
        "orc_rle_start_index": (u64p, [vp]),
        "orc_rle_end_index": (u64p, [vp]),
        "orc_convert_to_vec": (C.c_size_t, [vp, C.c_size_t, vp, C.c_size_t]),
        "orc_save_bwt_numpy": (C.c_int, [vp, C.c_size_t, C.c_char_p]),
        "orc_save_bwt_runs_numpy": (C.c_int, [vp, vp, C.c_size_t, C.c_char_p]),
        "orc_convert_stoi": (None, [vp, C.c_size_t, vp]),
        "orc_convert_itos": (None, [vp, C.c_size_t, vp]),
        "orc_reverse_complement_i": (None, [vp, C.c_size_t, vp]),
        "orc_naive_bwt": (C.c_size_t, [C.POINTER(C.c_char_p), C.c_size_t, vp]),
        "orc_decompress": (C.c_uint64, [vp, C.c_size_t, vp, C.c_uint64]),
        "orc_rank_bruteforce": (C.c_uint64, [vp, C.c_uint64, C.c_uint8, C.c_uint64]),
        "orc_runblock_count": (C.c_uint64, [vp, C.c_size_t, C.c_uint64, C.c_uint8]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d" % code)
        self.code = code


class OracleRleBWT:
    """Mirror of the reference's RleBWT (src/rle_bwt.rs) over the C restatement."""

    def __init__(self, bin_power=8):
        self._h = lib().orc_rle_new(bin_power)
        self.bin_power = bin_power

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().orc_rle_free(h)
            except (TypeError, AttributeError):  # interpreter shutdown
                pass

    def load_vector(self, bwt):
        a, p = _u8(bwt)
        rc = lib().orc_rle_load_vector(self._h, p, a.size)
        if rc:
            raise OracleError(rc)

    def load_numpy_file(self, filename):
        rc = lib().orc_rle_load_numpy_file(self._h, os.fsencode(filename))
        if rc:
            raise OracleError(rc)

    def get_symbol_count(self, sym):
        return int(lib().orc_rle_get_symbol_count(self._h, sym))

    def get_total_size(self):
        return int(lib().orc_rle_get_total_size(self._h))

    def constrain_range(self, sym, l, h):
        ol, oh = C.c_uint64(), C.c_uint64()
        rc = lib().orc_rle_constrain_range(self._h, sym, l, h, C.byref(ol), C.byref(oh), None)
        if rc:
            raise OracleError(rc)
        return int(ol.value), int(oh.value)

    def count_kmer(self, kmer):
        a, p = _u8(kmer)
        out = C.c_uint64()
        rc = lib().orc_rle_count_kmer(self._h, p, a.size, C.byref(out), None)
        if rc:
            raise OracleError(rc)
        return int(out.value)

    def count_kmers(self, kmers, nthreads=1, stats=None):
        """kmers: (n, k) uint8 codes. Returns uint64[n]; fills `stats` (Stats) if given."""
        a = np.ascontiguousarray(kmers, dtype=np.uint8)
        assert a.ndim == 2
        n, k = a.shape
        out = np.zeros(n, dtype=np.uint64)
        rc = lib().orc_rle_count_kmers(self._h, a.ctypes.data_as(C.c_void_p), k, n,
                                       out.ctypes.data_as(C.c_void_p), nthreads,
                                       C.byref(stats) if stats is not None else None)
        if rc:
            raise OracleError(rc)
        return out

    def constrain_ranges(self, syms, l, h):
        s, sp = _u8(syms)
        l = np.ascontiguousarray(l, dtype=np.uint64)
        h = np.ascontiguousarray(h, dtype=np.uint64)
        ol = np.zeros(s.size, dtype=np.uint64)
        oh = np.zeros(s.size, dtype=np.uint64)
        rc = lib().orc_rle_constrain_ranges(self._h, sp, l.ctypes.data_as(C.c_void_p),
                                            h.ctypes.data_as(C.c_void_p), s.size,
                                            ol.ctypes.data_as(C.c_void_p), oh.ctypes.data_as(C.c_void_p))
        if rc:
            raise OracleError(rc)
        return ol, oh

    # sampled index, for the G5 literals
    def ref_index(self):
        n = lib().orc_rle_index_length(self._h)
        return [int(lib().orc_rle_ref_index(self._h)[i]) for i in range(n)]

    def fm_index(self, sym):
        n = lib().orc_rle_index_length(self._h)
        return [int(lib().orc_rle_fm_index(self._h, sym)[i]) for i in range(n)]

    def start_index(self):
        return [int(lib().orc_rle_start_index(self._h)[i]) for i in range(6)]

    def end_index(self):
        return [int(lib().orc_rle_end_index(self._h)[i]) for i in range(6)]


def convert_to_vec(text):
    """src/bwt_converter.rs:26-80. text: bytes/str of "$ACGNT" (+ newlines)."""
    if isinstance(text, str):
        text = text.encode()
    a, p = _u8(np.frombuffer(text, dtype=np.uint8))
    need = lib().orc_convert_to_vec(p, a.size, None, 0)
    if need == C.c_size_t(-1).value:
        raise OracleError(ERR_SYMBOL)
    out = np.zeros(need, dtype=np.uint8)
    lib().orc_convert_to_vec(p, a.size, out.ctypes.data_as(C.c_void_p), need)
    return out


def save_bwt_numpy(bwt_bytes, filename):
    a, p = _u8(bwt_bytes)
    rc = lib().orc_save_bwt_numpy(p, a.size, os.fsencode(filename))
    if rc:
        raise OracleError(rc)


def save_bwt_runs_numpy(runs, filename):
    syms = np.array([r[0] for r in runs], dtype=np.uint8)
    cnts = np.array([r[1] for r in runs], dtype=np.uint64)
    rc = lib().orc_save_bwt_runs_numpy(syms.ctypes.data_as(C.c_void_p), cnts.ctypes.data_as(C.c_void_p),
                                       syms.size, os.fsencode(filename))
    if rc:
        raise OracleError(rc)


def convert_stoi(seq):
    if isinstance(seq, str):
        seq = seq.encode()
    a, p = _u8(np.frombuffer(seq, dtype=np.uint8))
    out = np.zeros(a.size, dtype=np.uint8)
    lib().orc_convert_stoi(p, a.size, out.ctypes.data_as(C.c_void_p))
    return out


def convert_itos(codes):
    a, p = _u8(codes)
    out = np.zeros(a.size, dtype=np.uint8)
    lib().orc_convert_itos(p, a.size, out.ctypes.data_as(C.c_void_p))
    return out.tobytes().decode()


def reverse_complement_i(codes):
    a, p = _u8(codes)
    out = np.zeros(a.size, dtype=np.uint8)
    lib().orc_reverse_complement_i(p, a.size, out.ctypes.data_as(C.c_void_p))
    return out


def naive_bwt(strings):
    arr = (C.c_char_p * len(strings))(*[s.encode() for s in strings])
    total = sum(len(s) + 1 for s in strings)
    out = np.zeros(max(total, 1), dtype=np.uint8)
    n = lib().orc_naive_bwt(arr, len(strings), out.ctypes.data_as(C.c_void_p))
    return out[:n].tobytes().decode()


def decompress(bwt_bytes):
    a, p = _u8(bwt_bytes)
    n = lib().orc_decompress(p, a.size, None, 0)
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_decompress(p, a.size, out.ctypes.data_as(C.c_void_p), n)
    return out


def rank_bruteforce(symbols, sym, pos):
    a, p = _u8(symbols)
    return int(lib().orc_rank_bruteforce(p, a.size, sym, pos))


def runblock_count(runs_u16, position, symbol):
    a = np.ascontiguousarray(runs_u16, dtype=np.uint16)
    return int(lib().orc_runblock_count(a.ctypes.data_as(C.c_void_p), a.size, position, symbol))
