"""rust-msbwt_amd -- MI355X-native batched k-mer counting over a run-length-encoded
multi-string BWT: the `RleBWT::count_kmer` path of HudsonAlpha/rust-msbwt, rebuilt for
gfx950 behind the reference's own interface.

The directory name carries a hyphen (it is fixed by the project layout), so import it with
    import importlib; msbwt = importlib.import_module("rust-msbwt_amd")
or through the `rust_msbwt_amd` alias module at the repo root.

Modules mirror the reference crate: msbwt_core (BWTRange, constants, BWT), rle_bwt (RleBWT),
string_util, bwt_converter.  Everything that computes runs in libmsbwt_hip.so.
"""
from . import _lib
from .msbwt_core import BWT, BWTRange, VC_LEN, LETTER_BITS, NUMBER_BITS, NUM_POWER, MASK, COUNT_MASK
from .rle_bwt import RleBWT, MsbwtError, RankComm
from . import string_util, bwt_converter, msbwt_core, rle_bwt, sharded

__all__ = ["BWT", "BWTRange", "RleBWT", "MsbwtError", "RankComm", "string_util", "bwt_converter", "msbwt_core",
           "rle_bwt", "sharded", "VC_LEN", "LETTER_BITS", "NUMBER_BITS", "NUM_POWER", "MASK", "COUNT_MASK"]


def version():
    return _lib.lib().msbwt_version().decode()


def auto_table_depths(total_symbols, free_hbm_bytes, pair_index=True):
    """(flat, packed) levels of the suffix table the library builds by default for an index of that size
    with that much HBM free after plane and pair blocks (packed 0 = the table stays flat).  Pure host logic."""
    import ctypes
    flat, packed = ctypes.c_int(0), ctypes.c_int(0)
    rc = _lib.lib().msbwt_auto_table_depths(int(total_symbols), int(free_hbm_bytes), 1 if pair_index else 0,
                                            ctypes.byref(flat), ctypes.byref(packed))
    if rc:
        raise MsbwtError(rc, "msbwt_auto_table_depths")
    return flat.value, packed.value


def auto_pair_stride(total_symbols, free_hbm_bytes, hbm_total_bytes, typical_width=-1.0):
    """Spacing of the pair blocks (96 or 128) the loader picks for an index of that size: `free_hbm_bytes` free once
    the plane blocks are in place, `typical_width` what the load-time probe reports about the data
    (RleBWT.get_typical_range_width; negative = unknown).  Pure host logic (csrc/table_policy.hpp)."""
    import ctypes
    stride = ctypes.c_int(0)
    rc = _lib.lib().msbwt_auto_pair_stride(int(total_symbols), int(free_hbm_bytes), int(hbm_total_bytes), float(typical_width), ctypes.byref(stride))
    if rc:
        raise MsbwtError(rc, "msbwt_auto_pair_stride")
    return stride.value


def auto_index_plan(total_symbols, free_hbm_bytes, hbm_total_bytes, typical_width=-1.0, budget_bytes=0):
    """What the loader builds under a memory budget (msbwt_rle_set_memory_budget; 0 = none) for an index of that size:
    {"pair_index", "pair_stride", "flat_depth", "packed_depth", "index_bytes"}.  Pure host logic (csrc/table_policy.hpp, plan_index)."""
    import ctypes
    pair, stride, flat, packed = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    size = ctypes.c_uint64(0)
    rc = _lib.lib().msbwt_auto_index_plan(int(total_symbols), int(free_hbm_bytes), int(hbm_total_bytes), float(typical_width), int(budget_bytes),
                                          ctypes.byref(pair), ctypes.byref(stride), ctypes.byref(flat), ctypes.byref(packed), ctypes.byref(size))
    if rc:
        raise MsbwtError(rc, "msbwt_auto_index_plan")
    return {"pair_index": bool(pair.value), "pair_stride": stride.value, "flat_depth": flat.value, "packed_depth": packed.value, "index_bytes": size.value}

