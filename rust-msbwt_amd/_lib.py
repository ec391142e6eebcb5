"""ctypes binding of libmsbwt_hip.so (include/msbwt_hip.h).  No fallback: if the library is
missing this raises, telling the caller to build it."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmsbwt_hip.so")

OK = 0
ERR_IO, ERR_UNEXPECTED_EOF, ERR_BAD_HEADER, ERR_INVALID_SYMBOL = -1, -2, -3, -4
ERR_INVALID_RANGE, ERR_HIP, ERR_NOT_LOADED, ERR_TOO_LARGE, ERR_INVALID_ARG, ERR_INTERNAL = -5, -6, -7, -8, -9, -10
ERR_OVERFLOW, ERR_RCCL = -11, -12
COMM_ID_BYTES = 128
SPARSE_INFO_WORDS = 120  # MSBWT_SPARSE_INFO_WORDS

SIZE_MAX = C.c_size_t(-1).value

# every symbol include/msbwt_hip.h declares: name -> (restype, argtypes)
_vp, _u8, _u64, _sz, _int = C.c_void_p, C.c_uint8, C.c_uint64, C.c_size_t, C.c_int
_pu64 = C.POINTER(C.c_uint64)
SIGNATURES = {
    "msbwt_rle_new": (_vp, [_u8]),
    "msbwt_rle_new_on_device": (_vp, [_u8, _int]),
    "msbwt_rle_free": (None, [_vp]),
    "msbwt_rle_load_vector": (_int, [_vp, _vp, _sz]),
    "msbwt_rle_load_numpy_file": (_int, [_vp, C.c_char_p]),
    "msbwt_rle_get_symbol_count": (_u64, [_vp, _u8]),
    "msbwt_rle_get_total_size": (_u64, [_vp]),
    "msbwt_rle_constrain_range": (_int, [_vp, _u8, _u64, _u64, _pu64, _pu64]),
    "msbwt_rle_count_kmer": (_int, [_vp, _vp, _sz, _pu64]),
    "msbwt_rle_count_kmers": (_int, [_vp, _vp, _sz, _sz, _vp]),
    "msbwt_rle_constrain_ranges": (_int, [_vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "msbwt_rle_count_kmers_device": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "msbwt_rle_constrain_ranges_device": (_int, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "msbwt_rle_device_status": (_int, [_vp, _vp]),
    "msbwt_rle_count_read_kmers": (_int, [_vp, _vp, _sz, _sz, _sz, _int, _vp, _vp]),
    "msbwt_rle_count_ragged_read_kmers": (_int, [_vp, _vp, _vp, _sz, _sz, _int, _vp, _vp, _pu64]),
    "msbwt_rle_count_read_kmers_device": (_int, [_vp, _vp, _sz, _sz, _sz, _int, _vp, _vp, _vp]),
    "msbwt_rle_replicate": (_vp, [_vp, _int]),
    "msbwt_rle_count_kmers_multi": (_int, [_vp, _sz, _vp, _sz, _sz, _vp]),
    "msbwt_rle_count_read_kmers_multi": (_int, [_vp, _sz, _vp, _sz, _sz, _sz, _int, _vp, _vp]),
    "msbwt_rle_count_kmers_multi_device": (_int, [_vp, _sz, _vp, _sz, _sz, _vp]),
    "msbwt_comm_get_unique_id": (_int, [_vp]),
    "msbwt_comm_init_rank": (_int, [C.POINTER(C.c_void_p), _int, _vp, _int]),
    "msbwt_comm_destroy": (_int, [_vp]),
    "msbwt_rle_allgather_counts": (_int, [_vp, _vp, _vp, _sz, _vp, _int, _vp]),
    "msbwt_rle_count_kmers_allgather_device": (_int, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _int, _int, _int, _vp]),
    "msbwt_kmers_pack_2bit": (_int, [_vp, _sz, _sz, _vp]),
    "msbwt_rle_count_kmers_packed": (_int, [_vp, _vp, _sz, _sz, _vp, _int]),
    "msbwt_rle_count_kmers_packed_device": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "msbwt_rle_set_batch_order": (_int, [_vp, _int]),
    "msbwt_rle_get_batch_order": (_int, [_vp]),
    "msbwt_rle_batch_order_for": (_int, [_vp, _sz, _sz]),
    "msbwt_kmer_order_keys": (_int, [_vp, _sz, _sz, _vp]),
    "msbwt_rle_kmer_order_keys_device": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "msbwt_rle_set_table_depth": (_int, [_vp, _int]),
    "msbwt_rle_get_table_depth": (_int, [_vp]),
    "msbwt_rle_set_table_packed": (_int, [_vp, _int]),
    "msbwt_rle_get_table_packed": (_int, [_vp]),
    "msbwt_rle_set_memory_budget": (_int, [_vp, _u64]),
    "msbwt_rle_get_memory_budget": (_u64, [_vp]),
    "msbwt_auto_index_plan": (_int, [_u64, _u64, _u64, C.c_double, _u64, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _pu64]),
    "msbwt_rle_set_table_side": (_int, [_vp, _int]),
    "msbwt_rle_table_info": (_int, [_vp, _pu64, _pu64, _pu64]),
    "msbwt_rle_set_sparse_table": (_int, [_vp, _int]),
    "msbwt_rle_get_sparse_table": (_int, [_vp]),
    "msbwt_rle_sparse_table_info": (_int, [_vp, _pu64]),
    "msbwt_sparse_hash": (_int, [_u64, _int, _u64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "msbwt_sparse_table_shape": (_int, [_int, _u64, _pu64, C.POINTER(C.c_int)]),
    "msbwt_sparse_hash64": (_int, [_u64, _int, _u64, C.POINTER(C.c_uint32), _pu64]),
    "msbwt_auto_sparse_depth": (_int, [_pu64, _pu64, _int, _u64, _int, C.POINTER(C.c_int), _pu64]),
    "msbwt_rle_download_sparse_table": (_sz, [_vp, _vp, _sz, _vp, _sz]),
    "msbwt_allgather_piece_queries": (_sz, [_sz, _int]),
    "msbwt_rle_set_search_counters": (_int, [_vp, _int]),
    "msbwt_rle_search_counters": (_int, [_vp, _vp, _vp]),
    "msbwt_rle_set_presence_filter": (_int, [_vp, _int]),
    "msbwt_rle_get_presence_filter": (_int, [_vp]),
    "msbwt_run_build_fits_device": (_int, [_u64, _u64]),
    "msbwt_rle_set_block_format": (_int, [_vp, _int]),
    "msbwt_rle_get_block_format": (_int, [_vp]),
    "msbwt_rle_set_search_kernel": (_int, [_vp, _int]),
    "msbwt_rle_get_search_kernel": (_int, [_vp]),
    "msbwt_rle_search_kernel_for": (_int, [_vp, _sz]),
    "msbwt_rle_set_pair_index": (_int, [_vp, _int]),
    "msbwt_rle_get_pair_index": (_int, [_vp]),
    "msbwt_rle_set_pair_stride": (_int, [_vp, _int]),
    "msbwt_rle_get_pair_stride": (_int, [_vp]),
    "msbwt_rle_get_typical_range_width": (C.c_double, [_vp]),
    "msbwt_auto_pair_stride": (_int, [_u64, _u64, _u64, C.c_double, C.POINTER(C.c_int)]),
    "msbwt_rle_set_sparse_tiers": (_int, [_vp, _int]),
    "msbwt_rle_get_sparse_tiers": (_int, [_vp]),
    "msbwt_rle_set_sparse_second": (_int, [_vp, _int]),
    "msbwt_auto_sparse_choice": (_int, [_pu64, _pu64, _pu64, _int, _u64, _int, _int, C.POINTER(C.c_int), C.POINTER(C.c_int), _pu64]),
    "msbwt_sparse_filter_bits": (_int, [_u64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "msbwt_rle_set_query_length": (_int, [_vp, _int]),
    "msbwt_rle_get_query_length": (_int, [_vp]),
    "msbwt_auto_sparse_max_depth": (_int, [_int]),
    "msbwt_rle_set_line_streaming": (_int, [_vp, _int]),
    "msbwt_rle_get_line_streaming": (_int, [_vp]),
    "msbwt_rle_probe_line_rate": (_int, [_vp, _int, C.POINTER(C.c_double)]),
    "msbwt_rle_device_bytes": (_u64, [_vp]),
    "msbwt_rle_kernel_time_ms": (_int, [_vp, C.POINTER(C.c_double), _pu64]),
    "msbwt_rle_set_kernel_timing": (_int, [_vp, _int]),
    "msbwt_rle_device_ordinal": (_int, [_vp]),
    "msbwt_rle_last_error": (C.c_char_p, [_vp]),
    "msbwt_version": (C.c_char_p, []),
    "msbwt_auto_table_depths": (_int, [_u64, _u64, _int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "msbwt_build_plane_blocks": (_sz, [_vp, _sz, _vp, _sz, _pu64]),
    "msbwt_build_run_blocks": (_sz, [_vp, _sz, _vp, _sz, _vp, _sz, _pu64, _pu64]),
    "msbwt_rle_download_blocks": (_sz, [_vp, _vp, _sz]),
    "msbwt_convert_to_vec": (_sz, [_vp, _sz, _vp, _sz]),
    "msbwt_save_bwt_numpy": (_int, [_vp, _sz, C.c_char_p]),
    "msbwt_save_bwt_runs_numpy": (_int, [_vp, _vp, _sz, C.c_char_p]),
    "msbwt_convert_stoi": (None, [_vp, _sz, _vp]),
    "msbwt_convert_itos": (None, [_vp, _sz, _vp]),
    "msbwt_reverse_complement_i": (None, [_vp, _sz, _vp]),
}

_lib = None


class MsbwtLibraryMissing(ImportError):
    pass


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (soname libamdhip64.so.7, loaded through an RPATH); if this library pulled in
    /opt/rocm's copy first, a later `import torch` would map a second runtime and one of the
    two would see no device.  So, when torch is installed, map its copy first: the loader
    then binds libmsbwt_hip.so's NEEDED libamdhip64.so.7 to it.  MSBWT_HIP_RUNTIME=system
    skips this (pure C/Rust hosts never come through here anyway)."""
    if os.environ.get("MSBWT_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        _share_hip_runtime_with_torch()
        if not os.path.exists(LIB_PATH):
            raise MsbwtLibraryMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
