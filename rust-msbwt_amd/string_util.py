"""Mirror of the reference's `string_util` (src/string_util.rs) over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib


def _as_u8(seq):
    if isinstance(seq, str):
        seq = seq.encode()
    if isinstance(seq, (bytes, bytearray)):
        return np.frombuffer(bytes(seq), dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


def convert_stoi(seq):
    """ASCII -> symbol codes; anything outside $ACGTacgt (N included) is 4 (string_util.rs:15-32,63-67)."""
    a = _as_u8(seq)
    out = np.empty(a.size, dtype=np.uint8)
    _lib.lib().msbwt_convert_stoi(a.ctypes.data_as(C.c_void_p), a.size, out.ctypes.data_as(C.c_void_p))
    return out


def convert_itos(iseq):
    """Symbol codes -> "$ACGNT" text (string_util.rs:80-88)."""
    a = _as_u8(iseq)
    out = np.empty(a.size, dtype=np.uint8)
    _lib.lib().msbwt_convert_itos(a.ctypes.data_as(C.c_void_p), a.size, out.ctypes.data_as(C.c_void_p))
    return out.tobytes().decode()


def reverse_complement_i(seq):
    """Reverse complement in code space, $ and N map to themselves (string_util.rs:12,45-50)."""
    a = _as_u8(seq)
    out = np.empty(a.size, dtype=np.uint8)
    _lib.lib().msbwt_reverse_complement_i(a.ctypes.data_as(C.c_void_p), a.size, out.ctypes.data_as(C.c_void_p))
    return out
