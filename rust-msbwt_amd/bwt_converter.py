"""Mirror of the reference's `bwt_converter` (src/bwt_converter.rs) over the C ABI."""
import ctypes as C
import os

import numpy as np

from . import _lib
from .rle_bwt import MsbwtError


def convert_to_vec(bwt):
    """"$ACGNT" text (newlines ignored) -> RLE bytes (bwt_converter.rs:26-80)."""
    if isinstance(bwt, str):
        bwt = bwt.encode()
    a = np.frombuffer(bytes(bwt), dtype=np.uint8) if isinstance(bwt, (bytes, bytearray)) else np.ascontiguousarray(bwt, dtype=np.uint8)
    L = _lib.lib()
    need = L.msbwt_convert_to_vec(a.ctypes.data_as(C.c_void_p), a.size, None, 0)
    if need == _lib.SIZE_MAX:
        raise ValueError("Unexpected symbol in input")  # the reference panics
    out = np.empty(need, dtype=np.uint8)
    L.msbwt_convert_to_vec(a.ctypes.data_as(C.c_void_p), a.size, out.ctypes.data_as(C.c_void_p), need)
    return out


def save_bwt_numpy(bwt, filename):
    """RLE bytes -> .npy with the crate's 96-byte header (bwt_converter.rs:102-130)."""
    a = np.ascontiguousarray(bwt, dtype=np.uint8)
    rc = _lib.lib().msbwt_save_bwt_numpy(a.ctypes.data_as(C.c_void_p), a.size, os.fsencode(filename))
    if rc:
        raise MsbwtError(rc, "cannot write %s" % filename)


def save_bwt_runs_numpy(runs, filename):
    """(symbol, count) runs -> .npy (bwt_converter.rs:151-184)."""
    runs = list(runs)
    syms = np.array([r[0] for r in runs], dtype=np.uint8)
    cnts = np.array([r[1] for r in runs], dtype=np.uint64)
    rc = _lib.lib().msbwt_save_bwt_runs_numpy(syms.ctypes.data_as(C.c_void_p), cnts.ctypes.data_as(C.c_void_p),
                                              syms.size, os.fsencode(filename))
    if rc:
        raise MsbwtError(rc, "cannot write %s" % filename)
