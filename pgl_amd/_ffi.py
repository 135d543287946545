"""ctypes binding of libpglamd.so (the C ABI declared in include/pgl_amd.h).

The product path has NO CPU fallback: if the shared library is missing or fails to load, every
op raises.  `lib()` is the only place the .so is opened, so "which native code ran" is
unambiguous for the driver's loaded-.so check.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGLAMD_LIB: load another build of the same ABI (kernel experiments: scripts/prof.py variant)
LIB_PATH = os.environ.get("PGLAMD_LIB") or os.path.join(_HERE, "csrc", "libpglamd.so")
_lib = None

c_i32, c_i64, c_sz, c_vp, c_u64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint64

# name -> (restype, argtypes) ; mirrors include/pgl_amd.h one to one
_SIGNATURES = {
    "pglamd_abi_version": (c_i32, []),
    "pglamd_set_option": (c_i32, [ctypes.c_char_p, c_i64]),
    "pglamd_last_error": (ctypes.c_char_p, []),
    "pglamd_device_arch": (ctypes.c_char_p, []),
    "pglamd_csr_build_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "pglamd_csr_build": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                  c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_unique_segment_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "pglamd_unique_segment": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_narrow_i64": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "pglamd_exclusive_scan_i64_workspace_bytes": (c_sz, [c_i64]),
    "pglamd_exclusive_scan_i64": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_aggregate_workspace_bytes": (c_sz, [c_i64, c_i64, c_i32]),
    "pglamd_aggregate": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64,
                                  c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_aggregate_ext": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64,
                                      c_i64, c_i64, c_i64, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_sz, c_i32, c_vp]),
    "pglamd_aggregate_dense_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "pglamd_aggregate_dense": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64,
                                        c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_winner_grad_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "pglamd_winner_grad": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_edge_operand_grad": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "pglamd_profile_begin": (c_i32, []),
    "pglamd_profile_end": (c_i32, [c_vp, c_vp]),
    "pglamd_profile_last_kernel": (ctypes.c_char_p, []),
    "pglamd_scatter_add_coo": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "pglamd_send_uv": (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "pglamd_segment_reduce_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i32]),
    "pglamd_segment_reduce": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_segment_softmax_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i32]),
    "pglamd_segment_softmax": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_gat_aggregate_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "pglamd_gat_aggregate": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_uint32,
                                      c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_gat_backward_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "pglamd_gat_backward": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_uint32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp,
                                     c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_add_score": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, ctypes.c_float, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "pglamd_add_score_chunks": (c_i64, [c_i64]),
    "pglamd_add_score_backward": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64,
                                           c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_sddmm": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "pglamd_seg_ptr_from_ids": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_vp, c_vp]),
    "pglamd_gather_rows": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i64, c_vp, c_vp]),
    "pglamd_gather_rows_cast": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp]),
    "pglamd_scatter_rows": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i64, c_vp, c_vp]),
    "pglamd_degree_norm": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp]),
    "pglamd_sample_neighbors_count": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "pglamd_sample_neighbors_fill": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_u64, c_vp, c_vp, c_vp, c_vp]),
    "pglamd_reindex_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "pglamd_reindex": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pglamd_build_index_host": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pglamd_map_ids": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "pglamd_partition_kway": (c_i32, [c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_u64, c_vp, c_vp]),
    "pglamd_partition_kway2": (c_i32, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_double, ctypes.c_double, c_u64, c_i32,
                                        c_vp, c_vp]),
    "pglamd_partition_edges": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, ctypes.c_double, ctypes.c_double,
                                        c_u64, c_i32, c_vp, c_vp]),
    "pglamd_row_epilogue_partials": (c_i64, [c_i64]),
    "pglamd_row_epilogue": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, ctypes.c_float, c_vp, c_vp, c_vp]),
    "pglamd_row_epilogue_backward": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "pglamd_comm_unique_id": (c_i32, [c_vp]),
    "pglamd_comm_init": (c_i32, [c_i32, c_i32, c_vp, c_vp]),
    "pglamd_comm_destroy": (c_i32, [c_vp]),
    "pglamd_halo_exchange_start": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "pglamd_halo_exchange_wait": (c_i32, [c_vp, c_vp]),
    "pglamd_halo_exchange_start_ranges": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "pglamd_halo_plan_sizes": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_i32, c_i32, c_vp]),
    "pglamd_halo_plan_fill": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_i32, c_i32] + [c_vp] * 13),
}

ABI_VERSION = 4
_ERRORS = {-1: ValueError, -2: TypeError, -3: OverflowError, -4: RuntimeError, -5: RuntimeError, -6: ValueError, -7: RuntimeError, -8: RuntimeError}


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    """Opens libpglamd.so once.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            "pgl_amd: native library %s not found. Build it with `python -m pgl_amd._build` "
            "(hipcc, gfx950). There is no CPU fallback for the message-passing path." % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)      # AttributeError here == header/.so mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if handle.pglamd_abi_version() != ABI_VERSION:
        raise NativeLibraryMissing("pgl_amd: ABI version mismatch (lib %d, python %d)" %
                                   (handle.pglamd_abi_version(), ABI_VERSION))
    _lib = handle
    return _lib


def check(rc, what=""):
    if rc == 0:
        return
    msg = lib().pglamd_last_error().decode("utf-8", "replace")
    raise _ERRORS.get(rc, RuntimeError)("pgl_amd%s: %s (code %d)" % (" " + what if what else "", msg, rc))


def exported_symbols():
    return sorted(_SIGNATURES)
