"""ctypes binding of libbsmrocm.so -- the C ABI declared in include/bsm_rocm.h.

There is NO CPU fallback: if the HIP library is missing or a call fails, an exception is
raised.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BSM_LIB", os.path.join(_HERE, "libbsmrocm.so"))  # BSM_LIB: developer A/B builds

BSM_F32, BSM_F64, BSM_C64, BSM_C128 = 0, 1, 2, 3
BSM_OP_N, BSM_OP_T, BSM_OP_C = 0, 1, 2
BSM_MEM_HOST, BSM_MEM_DEVICE = 0, 1
BSM_SCHED_SERIAL, BSM_SCHED_DYNAMIC = 0, 1
BSM_ACC_AUTO, BSM_ACC_ATOMIC, BSM_ACC_COLORED, BSM_ACC_GATHER, BSM_ACC_DIRECT = 0, 1, 2, 3, 4
BSM_DEVICE_CURRENT, BSM_DEVICE_NONE = -1, -2
BSM_COLOR_WORKSTREAM_DSATUR, BSM_COLOR_DSATUR = 0, 1
(BSM_BK_VBCRS_PERM, BSM_BK_VBCRS_ROWPTR, BSM_BK_VBCRS_COLINDICES, BSM_BK_VBCRS_ROWINDICES,
 BSM_BK_COLORS, BSM_BK_TRANSPOSECOLORS, BSM_BK_DIAGONALCOLORS) = range(7)


class BsmOptions(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("scheduler", C.c_int32),
                ("accumulate", C.c_int32), ("validate", C.c_int32), ("transpose_image", C.c_int32),
                ("own_lo", C.c_int64), ("own_hi", C.c_int64), ("ctx", C.c_void_p),
                ("blocks_memspace", C.c_int64), ("coloring", C.c_int64), ("reserved", C.c_int64 * 1)]


class BsmPartInfo(C.Structure):
    _fields_ = [("device", C.c_int32), ("reserved32", C.c_int32), ("own_lo", C.c_int64), ("own_hi", C.c_int64),
                ("touched_lo", C.c_int64), ("touched_hi", C.c_int64), ("device_bytes", C.c_int64),
                ("nblocks", C.c_int64), ("col_lo", C.c_int64), ("col_hi", C.c_int64), ("reserved", C.c_int64 * 2)]


class BsmStats(C.Structure):
    _fields_ = [("nnz", C.c_int64), ("stored_entries", C.c_int64), ("alg_bytes", C.c_int64),
                ("device_bytes", C.c_int64), ("npanels", C.c_int64), ("ntasks", C.c_int64),
                ("nworkgroups", C.c_int64), ("exclusive", C.c_int64), ("win_emissions", C.c_int64),
                ("win_inside", C.c_int64), ("win_flushed", C.c_int64), ("reserved", C.c_int64 * 5)]


class BsmError(RuntimeError):
    pass


_PP = C.POINTER(C.c_void_p)
_I64P = C.POINTER(C.c_int64)
_lib = None

# every symbol include/bsm_rocm.h declares
EXPORTS = ["bsm_options_default", "bsm_vbcrs_create", "bsm_vbcrs_create_from_symmetric",
           "bsm_vbcrs_create_from_blocksparse", "bsm_ctx_create", "bsm_ctx_destroy", "bsm_ctx_devices",
           "bsm_partition_rows", "bsm_part_info", "bsm_host_register", "bsm_host_unregister", "bsm_rowcolvals",
           "bsm_blocksparse_create",
           "bsm_symmetric_create", "bsm_mul", "bsm_mul_multi", "bsm_mul_parts", "bsm_get_bookkeeping", "bsm_get_image", "bsm_stats",
           "bsm_color", "bsm_destroy", "bsm_last_error", "bsm_version",
           "bsm_vec_add_segments", "bsm_stream_create_reserved", "bsm_stream_destroy"]


# include/bsm_synth.h (bench / test utility: synthetic operators generated in HBM)
SYNTH_EXPORTS = ["bsm_synth_blocks", "bsm_synth_vector", "bsm_bench_stream"]


def lib():
    """Loads libbsmrocm.so once; raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # compile on demand (hipcc --offload-arch=gfx950, in-tree); this is a build step, not a
        # fallback: without the HIP library nothing below works
        import fcntl
        import subprocess
        try:
            # one builder at a time: the ranks of a multi-GPU job may all arrive here together
            with open(os.path.join(_HERE, "csrc", ".build.lock"), "w") as lk:
                fcntl.flock(lk, fcntl.LOCK_EX)
                if not os.path.exists(LIB_PATH):
                    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the product path.")
    L = C.CDLL(LIB_PATH)
    L.bsm_options_default.argtypes = [C.POINTER(BsmOptions)]
    L.bsm_options_default.restype = None
    L.bsm_vbcrs_create.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, _PP, _I64P, _I64P,
                                   _I64P, _I64P, _I64P, C.POINTER(BsmOptions),
                                   C.POINTER(C.c_void_p)]
    L.bsm_blocksparse_create.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, _PP, _I64P,
                                         _I64P, _I64P, _PP, _PP, C.POINTER(BsmOptions),
                                         C.POINTER(C.c_void_p)]
    L.bsm_symmetric_create.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, _PP, _I64P,
                                       _I64P, _PP, C.c_int64, _PP, _I64P, _I64P, _I64P, _PP,
                                       _PP, C.POINTER(BsmOptions), C.POINTER(C.c_void_p)]
    L.bsm_vbcrs_create_from_symmetric.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, _PP, _I64P,
                                                  _I64P, _I64P, C.c_int64, _PP, _I64P, _I64P, _I64P,
                                                  _I64P, _I64P, C.POINTER(BsmOptions),
                                                  C.POINTER(C.c_void_p)]
    L.bsm_vbcrs_create_from_blocksparse.argtypes = L.bsm_blocksparse_create.argtypes
    _I32P = C.POINTER(C.c_int32)
    L.bsm_ctx_create.argtypes = [_I32P, C.c_int32, C.POINTER(C.c_void_p)]
    L.bsm_ctx_destroy.argtypes = [C.c_void_p]
    L.bsm_ctx_devices.argtypes = [C.c_void_p, _I32P, _I32P, C.c_int32]
    L.bsm_partition_rows.argtypes = [C.c_int64, C.c_int64, _I64P, _I64P, C.c_int32, _I32P, _I64P, _I64P]
    L.bsm_part_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(BsmPartInfo)]
    for name in ("bsm_vbcrs_create_from_blocksparse", "bsm_ctx_create", "bsm_ctx_destroy", "bsm_ctx_devices",
                 "bsm_partition_rows", "bsm_part_info"):
        getattr(L, name).restype = C.c_int
    L.bsm_rowcolvals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _I64P, C.c_int, C.c_void_p]
    L.bsm_rowcolvals.restype = C.c_int
    L.bsm_stream_create_reserved.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.bsm_stream_destroy.argtypes = [C.c_void_p]
    L.bsm_stream_create_reserved.restype = L.bsm_stream_destroy.restype = C.c_int
    L.bsm_vec_add_segments.argtypes = [C.c_int, C.c_void_p, C.c_int32, _I64P, _PP, _I64P, C.c_void_p]
    L.bsm_vec_add_segments.restype = C.c_int
    L.bsm_host_register.argtypes = [C.c_void_p, C.c_int64]
    L.bsm_host_unregister.argtypes = [C.c_void_p]
    L.bsm_host_register.restype = L.bsm_host_unregister.restype = C.c_int
    L.bsm_synth_blocks.argtypes = [C.c_int, C.c_uint64, C.c_int64, _I64P, _I64P, _I64P, _I32P, _PP, C.c_void_p]
    L.bsm_synth_vector.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.bsm_synth_blocks.restype = L.bsm_synth_vector.restype = C.c_int
    if hasattr(L, "bsm_bench_stream") or "BSM_LIB" not in os.environ:  # (developer A/B builds may predate it)
        L.bsm_bench_stream.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.bsm_bench_stream.restype = C.c_int
    L.bsm_mul.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.c_int, C.c_int, C.c_void_p]
    L.bsm_mul_multi.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.bsm_mul_multi.restype = C.c_int
    if hasattr(L, "bsm_mul_parts") or "BSM_LIB" not in os.environ:
        L.bsm_mul_parts.argtypes = [C.c_void_p, C.c_int, _PP, _PP, C.c_void_p, C.c_void_p, C.c_int, _PP]
        L.bsm_mul_parts.restype = C.c_int
    L.bsm_get_bookkeeping.argtypes = [C.c_void_p, C.c_int, _I64P, _I64P]
    L.bsm_get_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p, _I64P]
    L.bsm_color.argtypes = [C.c_int64, _PP, _I64P, C.c_int, _I64P, _I64P]
    L.bsm_color.restype = C.c_int
    L.bsm_stats.argtypes = [C.c_void_p, C.POINTER(BsmStats)]
    L.bsm_destroy.argtypes = [C.c_void_p]
    L.bsm_last_error.restype = C.c_char_p
    L.bsm_version.restype = C.c_char_p
    for name in ("bsm_vbcrs_create", "bsm_vbcrs_create_from_symmetric", "bsm_blocksparse_create", "bsm_symmetric_create", "bsm_mul",
                 "bsm_get_bookkeeping", "bsm_get_image", "bsm_stats", "bsm_destroy"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise BsmError(f"libbsmrocm error {rc}: {lib().bsm_last_error().decode()}")
