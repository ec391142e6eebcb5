"""`RleBWT` -- host-side mirror of the reference's `msbwt2::rle_bwt::RleBWT`
(src/rle_bwt.rs) whose queries run on the MI355X through the C ABI (include/msbwt_hip.h).

Same method names, argument meaning and error behaviour as the reference's `impl BWT for
RleBWT`; `count_kmers` / `constrain_ranges` are the batch forms (host arrays), and the
`*_device` forms take device pointers (e.g. `torch.Tensor.data_ptr()`) and a HIP stream.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .msbwt_core import BWT, BWTRange, VC_LEN


class MsbwtError(Exception):
    def __init__(self, code, message=""):
        super().__init__("msbwt error %d: %s" % (code, message))
        self.code = code


def _raise(code, handle):
    msg = _lib.lib().msbwt_rle_last_error(handle)
    msg = msg.decode(errors="replace") if msg else ""
    if code in (_lib.ERR_IO, _lib.ERR_UNEXPECTED_EOF):
        # the reference returns io::Error for these (rle_bwt.rs:84-148)
        err = EOFError(msg) if code == _lib.ERR_UNEXPECTED_EOF else OSError(msg)
        err.code = code
        raise err
    raise MsbwtError(code, msg)


class RleBWT(BWT):
    def __init__(self, bin_power=8, device=-1):
        """RleBWT::new() / with_bin_power (rle_bwt.rs:297-322). `device` = HIP ordinal."""
        self._h = None
        self._h = _lib.lib().msbwt_rle_new_on_device(bin_power, device)
        if not self._h:
            raise MemoryError("msbwt_rle_new failed")
        self.bin_power = bin_power

    @classmethod
    def with_bin_power(cls, bin_power, device=-1):
        return cls(bin_power, device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().msbwt_rle_free(h)
            except (TypeError, AttributeError):  # interpreter shutdown: module globals already gone
                pass

    # ---- trait BWT -------------------------------------------------------------------
    def load_vector(self, bwt):
        a = np.ascontiguousarray(bwt, dtype=np.uint8)
        rc = _lib.lib().msbwt_rle_load_vector(self._h, a.ctypes.data_as(C.c_void_p), a.size)
        if rc:
            _raise(rc, self._h)

    def load_numpy_file(self, filename):
        rc = _lib.lib().msbwt_rle_load_numpy_file(self._h, os.fsencode(filename))
        if rc:
            _raise(rc, self._h)

    def get_symbol_count(self, symbol):
        if not 0 <= symbol < VC_LEN:
            raise IndexError("symbol out of range")  # the reference panics (array index)
        return int(_lib.lib().msbwt_rle_get_symbol_count(self._h, symbol))

    def get_total_size(self):
        return int(_lib.lib().msbwt_rle_get_total_size(self._h))

    def constrain_range(self, sym, input_range):
        ol, oh = C.c_uint64(), C.c_uint64()
        rc = _lib.lib().msbwt_rle_constrain_range(self._h, sym, input_range.l, input_range.h,
                                                  C.byref(ol), C.byref(oh))
        if rc:
            _raise(rc, self._h)
        return BWTRange(int(ol.value), int(oh.value))

    def count_kmer(self, kmer):
        a = np.ascontiguousarray(kmer, dtype=np.uint8)
        out = C.c_uint64()
        rc = _lib.lib().msbwt_rle_count_kmer(self._h, a.ctypes.data_as(C.c_void_p), a.size, C.byref(out))
        if rc:
            _raise(rc, self._h)
        return int(out.value)

    # ---- batch forms -----------------------------------------------------------------
    def count_kmers(self, kmers, out=None):
        """kmers: (n, k) uint8 symbol codes -> uint64[n] (`out`: optional preallocated result array)."""
        a = np.ascontiguousarray(kmers, dtype=np.uint8)
        if a.ndim != 2:
            raise ValueError("kmers must be (n, k)")
        n, k = a.shape
        if out is None:
            out = np.empty(n, dtype=np.uint64)
        elif out.dtype != np.uint64 or out.shape != (n,) or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous uint64 array of length n")
        rc = _lib.lib().msbwt_rle_count_kmers(self._h, a.ctypes.data_as(C.c_void_p), k, n,
                                              out.ctypes.data_as(C.c_void_p))
        if rc:
            _raise(rc, self._h)
        return out

    def constrain_ranges(self, syms, l, h):
        s = np.ascontiguousarray(syms, dtype=np.uint8)
        l = np.ascontiguousarray(l, dtype=np.uint64)
        h = np.ascontiguousarray(h, dtype=np.uint64)
        if not (s.shape == l.shape == h.shape and s.ndim == 1):
            raise ValueError("syms, l, h must be 1-D and equally long")
        ol = np.empty(s.size, dtype=np.uint64)
        oh = np.empty(s.size, dtype=np.uint64)
        rc = _lib.lib().msbwt_rle_constrain_ranges(self._h, s.ctypes.data_as(C.c_void_p),
                                                   l.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p),
                                                   s.size, ol.ctypes.data_as(C.c_void_p), oh.ctypes.data_as(C.c_void_p))
        if rc:
            _raise(rc, self._h)
        return ol, oh

    def count_read_kmers(self, reads, k, ascii=None, forward=True, revcomp=False, out_fwd=None, out_rc=None):
        """Counts every k-mer window of every read, fused on the GPU (no n x k query matrix).

        reads: (n_reads, read_len) uint8 -- symbol codes, or ASCII bytes (`ascii=True`; a list of
        equal-length str/bytes is accepted and implies ASCII).  Returns (fwd, rc): uint64 arrays
        of shape (n_reads, read_len - k + 1), `None` for a strand that was not requested.
        rc[r, w] = count_kmer(reverse_complement_i(window))."""
        if not isinstance(reads, np.ndarray):
            reads = np.array([np.frombuffer(r.encode() if isinstance(r, str) else bytes(r), dtype=np.uint8) for r in reads])
            ascii = True if ascii is None else ascii
        a = np.ascontiguousarray(reads, dtype=np.uint8)
        if a.ndim != 2:
            raise ValueError("reads must be (n_reads, read_len)")
        n, length = a.shape
        w = length - k + 1
        fwd = (out_fwd if out_fwd is not None else np.empty((n, max(w, 0)), dtype=np.uint64)) if forward else None
        rc = (out_rc if out_rc is not None else np.empty((n, max(w, 0)), dtype=np.uint64)) if revcomp else None
        for o in (fwd, rc):
            if o is not None and (o.dtype != np.uint64 or o.shape != (n, max(w, 0)) or not o.flags.c_contiguous):
                raise ValueError("output arrays must be contiguous uint64 of shape (n_reads, read_len - k + 1)")
        code = _lib.lib().msbwt_rle_count_read_kmers(
            self._h, a.ctypes.data_as(C.c_void_p), length, n, k, 1 if ascii else 0,
            fwd.ctypes.data_as(C.c_void_p) if fwd is not None else None,
            rc.ctypes.data_as(C.c_void_p) if rc is not None else None)
        if code:
            _raise(code, self._h)
        return fwd, rc

    def count_ragged_read_kmers(self, reads, k, ascii=True, forward=True, revcomp=False):
        """Like count_read_kmers for reads of different lengths.  `reads`: list of str/bytes (ASCII)
        or of uint8 arrays (codes when ascii=False).  Returns (fwd, rc, window_offsets): flat
        uint64 arrays in read order, read r owning [window_offsets[r], window_offsets[r+1])."""
        arrs = [np.frombuffer(r.encode() if isinstance(r, str) else bytes(r), dtype=np.uint8)
                if isinstance(r, (str, bytes, bytearray)) else np.asarray(r, dtype=np.uint8) for r in reads]
        lens = np.array([len(a) for a in arrs], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        flat = np.ascontiguousarray(np.concatenate(arrs)) if arrs else np.zeros(0, dtype=np.uint8)
        woff = np.concatenate([[0], np.cumsum(np.where(lens >= k, lens - np.uint64(k) + np.uint64(1), 0))]).astype(np.uint64)
        total = int(woff[-1])
        fwd = np.empty(total, dtype=np.uint64) if forward else None
        rc = np.empty(total, dtype=np.uint64) if revcomp else None
        got = C.c_uint64()
        code = _lib.lib().msbwt_rle_count_ragged_read_kmers(
            self._h, flat.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), len(arrs), k,
            1 if ascii else 0, fwd.ctypes.data_as(C.c_void_p) if fwd is not None else None,
            rc.ctypes.data_as(C.c_void_p) if rc is not None else None, C.byref(got))
        if code:
            _raise(code, self._h)
        assert int(got.value) == total
        return fwd, rc, woff

    def count_read_kmers_device(self, d_reads, read_len, n_reads, k, ascii, d_out_fwd, d_out_rc, stream=0):
        code = _lib.lib().msbwt_rle_count_read_kmers_device(self._h, d_reads, read_len, n_reads, k,
                                                            1 if ascii else 0, d_out_fwd, d_out_rc, stream)
        if code:
            _raise(code, self._h)

    def count_kmers_device(self, d_kmers, k, n, d_out, stream=0):
        """Device pointers (ints); asynchronous on `stream` (a hipStream_t as int)."""
        rc = _lib.lib().msbwt_rle_count_kmers_device(self._h, d_kmers, k, n, d_out, stream)
        if rc:
            _raise(rc, self._h)

    def constrain_ranges_device(self, d_syms, d_l, d_h, n, d_out_l, d_out_h, stream=0):
        rc = _lib.lib().msbwt_rle_constrain_ranges_device(self._h, d_syms, d_l, d_h, n, d_out_l, d_out_h, stream)
        if rc:
            _raise(rc, self._h)

    def device_status(self, stream=0):
        """Synchronises `stream` and raises if a device batch saw invalid input."""
        rc = _lib.lib().msbwt_rle_device_status(self._h, stream)
        if rc:
            _raise(rc, self._h)

    def count_kmers_packed(self, words, k, count_bits=64, out=None):
        """msbwt_rle_count_kmers_packed: `words` = (n, ceil(k / 32)) uint64, two bits per symbol (pack_2bit); returns uint64[n]
        or, with count_bits=32, uint32[n] (`out`: optional preallocated result array of that type)."""
        w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, 2 if k > 32 else 1)
        want = np.uint64 if count_bits == 64 else np.uint32
        if out is None:
            out = np.empty(len(w), dtype=want)
        elif out.dtype != want or out.shape != (len(w),) or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous array of n counts of the requested width")
        rc = _lib.lib().msbwt_rle_count_kmers_packed(self._h, w.ctypes.data_as(C.c_void_p), k, len(w), out.ctypes.data_as(C.c_void_p), count_bits)
        if rc:
            _raise(rc, self._h)
        return out

    def count_kmers_packed_device(self, d_words, k, n, d_out, stream=0):
        rc = _lib.lib().msbwt_rle_count_kmers_packed_device(self._h, d_words, k, n, d_out, stream)
        if rc:
            _raise(rc, self._h)

    def set_batch_order(self, mode):
        """-1 = automatic (default; today that means never, include/msbwt_hip.h), 0 = never, 1 = whenever the pass applies."""
        rc = _lib.lib().msbwt_rle_set_batch_order(self._h, int(mode))
        if rc:
            _raise(rc, self._h)

    def get_batch_order(self):
        return int(_lib.lib().msbwt_rle_get_batch_order(self._h))

    def batch_order_for(self, k, n):
        rc = int(_lib.lib().msbwt_rle_batch_order_for(self._h, int(k), int(n)))
        if rc < 0:
            _raise(rc, self._h)
        return bool(rc)

    def kmer_order_keys_device(self, d_kmers, k, n, d_out_keys, stream=0):
        """msbwt_rle_kmer_order_keys_device: u64 keys (device pointers); a batch sorted by them ascending walks the index in order."""
        rc = _lib.lib().msbwt_rle_kmer_order_keys_device(self._h, d_kmers, k, n, d_out_keys, stream)
        if rc:
            _raise(rc, self._h)

    def allgather_counts(self, comm, d_mine, n_mine, d_all, wire_bits=64, stream=0):
        """msbwt_rle_allgather_counts: d_all[r * n_mine + i] = rank r's d_mine[i] on every rank (device pointers,
        u64 counts), asynchronous on `stream`; wire_bits 16 / 32 narrows the payload (overflow -> device_status)."""
        rc = _lib.lib().msbwt_rle_allgather_counts(self._h, comm._c, d_mine, n_mine, d_all, wire_bits, stream)
        if rc:
            _raise(rc, self._h)

    def count_kmers_allgather_device(self, comm, d_kmers, k, n_mine, d_mine_counts, d_all, wire_bits=16, out_bits=64, pieces=4, stream=0):
        """msbwt_rle_count_kmers_allgather_device: this rank's shard counted piece by piece on `stream` while the finished pieces' counts
        are all-gathered on a second stream; d_all[r * n_mine + i] as out_bits-wide integers (64 or the wire width)."""
        rc = _lib.lib().msbwt_rle_count_kmers_allgather_device(self._h, comm._c, d_kmers, k, n_mine, d_mine_counts, d_all, wire_bits, out_bits, pieces, stream)
        if rc:
            _raise(rc, self._h)

    # ---- several GPUs of one node ---------------------------------------------------------
    def replicate(self, device):
        """A new RleBWT on `device` holding a GPU -> GPU copy of this index (no rebuild, no upload)."""
        h = _lib.lib().msbwt_rle_replicate(self._h, device)
        if not h:
            _raise(_lib.ERR_HIP, self._h)
        other = object.__new__(RleBWT)
        other._h = h
        other.bin_power = self.bin_power
        return other

    # ---- tuning / introspection --------------------------------------------------------
    def set_table_depth(self, depth):
        rc = _lib.lib().msbwt_rle_set_table_depth(self._h, depth)
        if rc:
            _raise(rc, self._h)

    def get_table_depth(self):
        return int(_lib.lib().msbwt_rle_get_table_depth(self._h))

    def set_table_packed(self, mode):
        """1 = packed table two levels deeper whenever a pair index exists, 0 = flat table only, -1 = automatic."""
        rc = _lib.lib().msbwt_rle_set_table_packed(self._h, mode)
        if rc:
            _raise(rc, self._h)

    def get_table_packed(self):
        return bool(_lib.lib().msbwt_rle_get_table_packed(self._h))

    def set_memory_budget(self, nbytes):
        """HBM the loaded index may hold (0 = no budget); rebuilds the optional structures of a loaded index under it."""
        rc = _lib.lib().msbwt_rle_set_memory_budget(self._h, int(nbytes))
        if rc:
            _raise(rc, self._h)

    def get_memory_budget(self):
        return int(_lib.lib().msbwt_rle_get_memory_budget(self._h))

    def set_table_side(self, mode):
        """1 = escape lines of the packed table keep flat entries in a side array (default), 0 = their queries search from scratch."""
        rc = _lib.lib().msbwt_rle_set_table_side(self._h, int(mode))
        if rc:
            _raise(rc, self._h)

    def table_info(self):
        """{"lines", "escape_lines", "side_bytes"} of the packed table in HBM (zeros without one)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = _lib.lib().msbwt_rle_table_info(self._h, C.byref(a), C.byref(b), C.byref(c))
        if rc:
            _raise(rc, self._h)
        return {"lines": a.value, "escape_lines": b.value, "side_bytes": c.value}

    COUNTER_NAMES = ("wave_steps", "lane_steps", "pair_steps", "second_lines", "sat_out", "escape_queries", "escape_restarts",
                     "table_decided", "searched", "first_lines", "table_steps", "table_displaced", "table_rides", "waves_worked", "tier_fallbacks")

    # ---- sparse suffix table (include/msbwt_hip.h, msbwt_rle_set_sparse_table) ----
    def set_sparse_table(self, depth):
        """-1 = automatic (default), 0 = off, 16..31 = exactly that depth (29: at least 2^29 buckets, 69 GB)."""
        rc = _lib.lib().msbwt_rle_set_sparse_table(self._h, int(depth))
        if rc:
            _raise(rc, self._h)

    def set_query_length(self, k):
        """The k this index will mostly be asked about (0 = unknown): the AUTOMATIC sparse table goes as deep as min(k, 31) -- where its table fits -- instead of 23
        -- a table of d-mers serves k >= d only.  Results never depend on it."""
        rc = _lib.lib().msbwt_rle_set_query_length(self._h, int(k))
        if rc:
            _raise(rc, self._h)

    def get_query_length(self):
        return int(_lib.lib().msbwt_rle_get_query_length(self._h))

    def set_sparse_tiers(self, mode):
        """Two-tier form of the sparse table (entries for the suffixes that occur at least twice, filter bits for the rest -- read sets with
        errors): -1 = where the complete table of a depth does not fit (default), 0 = never, 1 = always.  Results never depend on it."""
        rc = _lib.lib().msbwt_rle_set_sparse_tiers(self._h, int(mode))
        if rc:
            _raise(rc, self._h)

    def set_sparse_second(self, mode):
        """The second, shallower sparse level (17-symbol suffixes, for k undeclared): -1 = automatic (default), 0 = never."""
        rc = _lib.lib().msbwt_rle_set_sparse_second(self._h, int(mode))
        if rc:
            _raise(rc, self._h)

    def get_sparse_tiers(self):
        """True when the sparse table in HBM is of the two-tier form."""
        return bool(_lib.lib().msbwt_rle_get_sparse_tiers(self._h))

    def get_sparse_table(self):
        """Depth of the sparse suffix table in HBM, 0 = none."""
        return int(_lib.lib().msbwt_rle_get_sparse_table(self._h))

    def sparse_table_info(self):
        """What msbwt_rle_sparse_table_info reports, by name; "distinct" / "wide": {depth: count} for the depths the build passed."""
        out = (C.c_uint64 * _lib.SPARSE_INFO_WORDS)()
        rc = _lib.lib().msbwt_rle_sparse_table_info(self._h, out)
        if rc:
            _raise(rc, self._h)
        info = {"depth": int(out[0]), "entries": int(out[1]), "buckets": int(out[2]), "bytes": int(out[3]), "side_entries": int(out[4]),
                "side_bytes": int(out[5]), "displaced": int(out[6]), "parent_depth": int(out[7]), "two_tier": bool(out[8]), "probe": int(out[9]),
                "filtered": int(out[42]), "second_depth": int(out[43]), "second_bytes": int(out[44])}
        info["distinct"] = {d: int(out[10 + d]) for d in range(32) if out[10 + d]}
        info["wide"] = {d: int(out[45 + d]) for d in range(32) if out[10 + d]}
        info["once"] = {d: int(out[80 + d]) for d in range(32) if out[10 + d]}
        return info

    def download_sparse_table(self):
        """(lines, side): the bucket lines as a (nlines, 32) uint32 array and the side array as (n, 2) uint64 -- for tests."""
        info = self.sparse_table_info()
        if not info["depth"]:
            return None, None
        lines = np.empty((info["bytes"] // 128, 32), dtype=np.uint32)
        side = np.empty((info["side_bytes"] // 16, 2), dtype=np.uint64)
        got = _lib.lib().msbwt_rle_download_sparse_table(self._h, lines.ctypes.data_as(C.c_void_p), lines.nbytes,
                                                         side.ctypes.data_as(C.c_void_p) if side.size else None, side.nbytes)
        if got == _lib.SIZE_MAX:
            _raise(_lib.ERR_HIP, self._h)
        return lines, side

    def set_search_counters(self, enabled):
        rc = _lib.lib().msbwt_rle_set_search_counters(self._h, 1 if enabled else 0)
        if rc:
            _raise(rc, self._h)

    def search_counters(self, stream=0):
        """The counters of the launches since the last call, by name (include/msbwt_hip.h); reads and zeroes them."""
        out = (C.c_uint64 * 16)()
        rc = _lib.lib().msbwt_rle_search_counters(self._h, out, stream)
        if rc:
            _raise(rc, self._h)
        return {name: int(out[i]) for i, name in enumerate(self.COUNTER_NAMES)}

    def set_presence_filter(self, mode):
        """0 = no presence filter, anything else = automatic (kept when it can reject something)."""
        rc = _lib.lib().msbwt_rle_set_presence_filter(self._h, mode)
        if rc:
            _raise(rc, self._h)

    def get_presence_filter(self):
        """Depth of the L2-resident presence filter in front of the suffix table, 0 if none."""
        return int(_lib.lib().msbwt_rle_get_presence_filter(self._h))

    def set_pair_index(self, mode):
        """1 = build the two-symbols-per-step index, 0 = drop it, -1 = automatic (default)."""
        rc = _lib.lib().msbwt_rle_set_pair_index(self._h, mode)
        if rc:
            _raise(rc, self._h)

    def get_pair_index(self):
        return bool(_lib.lib().msbwt_rle_get_pair_index(self._h))

    def set_pair_stride(self, stride):
        """128 = disjoint pair blocks, 96 = overlapping (one line for ranges up to 32 wide), 0 = automatic."""
        rc = _lib.lib().msbwt_rle_set_pair_stride(self._h, stride)
        if rc:
            _raise(rc, self._h)

    def get_pair_stride(self):
        return int(_lib.lib().msbwt_rle_get_pair_stride(self._h))

    def get_typical_range_width(self):
        """Median number of occurrences of a present 24-mer, probed at load time (-1.0: not probed)."""
        return float(_lib.lib().msbwt_rle_get_typical_range_width(self._h))

    BLOCK_FORMATS = {"planes": 0, "runs": 1}

    def set_block_format(self, fmt):
        """"planes" (default) or "runs" (memory-lean run blocks, no pair index); takes effect at the next load."""
        rc = _lib.lib().msbwt_rle_set_block_format(self._h, self.BLOCK_FORMATS.get(fmt, fmt))
        if rc:
            _raise(rc, self._h)

    def get_block_format(self):
        return {v: k for k, v in self.BLOCK_FORMATS.items()}[int(_lib.lib().msbwt_rle_get_block_format(self._h))]

    SEARCH_KERNELS = {"auto": 0, "groups": 1, "lanes": 2}

    def set_search_kernel(self, mode):
        """"auto" (default), "groups" (8 lanes per query) or "lanes" (one query per lane, LDS-staged lines)."""
        rc = _lib.lib().msbwt_rle_set_search_kernel(self._h, self.SEARCH_KERNELS.get(mode, mode))
        if rc:
            _raise(rc, self._h)

    def get_search_kernel(self):
        return {v: k for k, v in self.SEARCH_KERNELS.items()}[int(_lib.lib().msbwt_rle_get_search_kernel(self._h))]

    def search_kernel_for(self, k):
        """The kernel a batch of k-symbol queries runs on right now: "lanes", "groups" or "generic" (k > 64)."""
        rc = int(_lib.lib().msbwt_rle_search_kernel_for(self._h, int(k)))
        if rc < 0:
            _raise(rc, self._h)
        return {0: "generic", 1: "groups", 2: "lanes"}[rc]

    def set_line_streaming(self, mode):
        """-1 = automatic (index lines are fetched non-temporally once the random-access arrays reach 4 GiB), 0 = never, 1 = always."""
        rc = _lib.lib().msbwt_rle_set_line_streaming(self._h, int(mode))
        if rc:
            _raise(rc, self._h)

    def get_line_streaming(self):
        return bool(_lib.lib().msbwt_rle_get_line_streaming(self._h))

    PROBE_ARRAYS = {"blocks": 0, "pair_blocks": 1, "sparse_table": 2, "table": 3}

    def probe_line_rate(self, which):
        """Random 128-byte lines per second served right now from one of the index's arrays ("blocks", "pair_blocks",
        "sparse_table", "table"); 0.0 if there is no such array.  A placement diagnostic (msbwt_rle_probe_line_rate)."""
        out = C.c_double()
        rc = _lib.lib().msbwt_rle_probe_line_rate(self._h, self.PROBE_ARRAYS.get(which, which), C.byref(out))
        if rc:
            _raise(rc, self._h)
        return float(out.value)

    def device_bytes(self):
        return int(_lib.lib().msbwt_rle_device_bytes(self._h))

    def set_kernel_timing(self, enabled):
        _lib.lib().msbwt_rle_set_kernel_timing(self._h, 1 if enabled else 0)

    def kernel_time_ms(self):
        """(average ms per count-kernel launch, launches) since the last call; HIP events."""
        avg, cnt = C.c_double(), C.c_uint64()
        rc = _lib.lib().msbwt_rle_kernel_time_ms(self._h, C.byref(avg), C.byref(cnt))
        if rc:
            _raise(rc, self._h)
        return float(avg.value), int(cnt.value)

    def device_ordinal(self):
        return int(_lib.lib().msbwt_rle_device_ordinal(self._h))


class RankComm:
    """An RCCL communicator over the GPUs of a one-process-per-GPU job, made through the library
    (msbwt_comm_*): rank 0 calls RankComm.unique_id(), the bytes reach the other ranks by any channel
    (torch.distributed.broadcast in bench.py), every rank calls RankComm(nranks, id, rank) with its GPU current."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        rc = _lib.lib().msbwt_comm_get_unique_id(buf)
        if rc:
            raise MsbwtError(rc, "msbwt_comm_get_unique_id (is librccl.so there?)")
        return bytes(buf)

    def __init__(self, nranks, unique_id, rank):
        self._c = C.c_void_p()
        self.nranks, self.rank = nranks, rank
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(unique_id)
        rc = _lib.lib().msbwt_comm_init_rank(C.byref(self._c), nranks, buf, rank)
        if rc:
            raise MsbwtError(rc, "msbwt_comm_init_rank")

    def close(self):
        c, self._c = self._c, None
        if c:
            _lib.lib().msbwt_comm_destroy(c)

    def __del__(self):
        try:
            self.close()
        except (TypeError, AttributeError):
            pass


def kmer_order_keys(kmers):
    """msbwt_kmer_order_keys (host): the u64 key of every row of an (n, k) matrix of symbol codes; sort a batch by it
    (ascending) and its queries read the suffix table, and then the index, in ascending order."""
    a = np.ascontiguousarray(kmers, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("kmers must be (n, k)")
    n, k = a.shape
    out = np.empty(n, dtype=np.uint64)
    rc = _lib.lib().msbwt_kmer_order_keys(a.ctypes.data_as(C.c_void_p), k, n, out.ctypes.data_as(C.c_void_p))
    if rc:
        raise MsbwtError(rc, "msbwt_kmer_order_keys")
    return out


def _handles(replicas):
    arr = (C.c_void_p * len(replicas))(*[r._h for r in replicas])
    return arr


def count_kmers_multi(replicas, kmers, out=None):
    """msbwt_rle_count_kmers_multi: a host batch sharded over the replicas (one per GPU)."""
    a = np.ascontiguousarray(kmers, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("kmers must be (n, k)")
    n, k = a.shape
    if out is None:
        out = np.empty(n, dtype=np.uint64)
    rc = _lib.lib().msbwt_rle_count_kmers_multi(_handles(replicas), len(replicas), a.ctypes.data_as(C.c_void_p), k, n,
                                                out.ctypes.data_as(C.c_void_p))
    if rc:
        raise MsbwtError(rc, "; ".join(_lib.lib().msbwt_rle_last_error(r._h).decode(errors="replace") for r in replicas))
    return out


def pack_2bit(kmers):
    """(n, k) symbol codes over ACGT -> (n, ceil(k / 32)) uint64 words (msbwt_kmers_pack_2bit; include/msbwt_hip.h has the layout)."""
    a = np.ascontiguousarray(kmers, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("kmers must be (n, k)")
    n, k = a.shape
    out = np.empty((n, 2 if k > 32 else 1), dtype=np.uint64)
    rc = _lib.lib().msbwt_kmers_pack_2bit(a.ctypes.data_as(C.c_void_p), k, n, out.ctypes.data_as(C.c_void_p))
    if rc:
        raise MsbwtError(rc, "a symbol outside A C G T cannot be packed into two bits")
    return out


def count_read_kmers_multi(replicas, reads, k, ascii=False, forward=True, revcomp=False):
    """msbwt_rle_count_read_kmers_multi: reads sharded over the replicas, every k-mer window counted."""
    a = np.ascontiguousarray(reads, dtype=np.uint8)
    n, length = a.shape
    w = length - k + 1
    fwd = np.empty((n, w), dtype=np.uint64) if forward else None
    rc_arr = np.empty((n, w), dtype=np.uint64) if revcomp else None
    rc = _lib.lib().msbwt_rle_count_read_kmers_multi(
        _handles(replicas), len(replicas), a.ctypes.data_as(C.c_void_p), length, n, k, 1 if ascii else 0,
        fwd.ctypes.data_as(C.c_void_p) if fwd is not None else None, rc_arr.ctypes.data_as(C.c_void_p) if rc_arr is not None else None)
    if rc:
        raise MsbwtError(rc, "; ".join(_lib.lib().msbwt_rle_last_error(r._h).decode(errors="replace") for r in replicas))
    return fwd, rc_arr


def count_kmers_multi_device(replicas, d_kmers, k, n, d_out):
    """msbwt_rle_count_kmers_multi_device: device pointers on replicas[0]'s device; synchronous."""
    rc = _lib.lib().msbwt_rle_count_kmers_multi_device(_handles(replicas), len(replicas), d_kmers, k, n, d_out)
    if rc:
        raise MsbwtError(rc, "; ".join(_lib.lib().msbwt_rle_last_error(r._h).decode(errors="replace") for r in replicas))
