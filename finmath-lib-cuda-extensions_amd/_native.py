"""ctypes binding of libfmhip.so — the C-ABI declared in include/fmhip.h, nothing else.

The product path has NO fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfmhip.so")
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/fmhip.h declares (tests/test_abi_symbols.py checks the header against this list)
SYMBOLS = [
    "fmhip_init", "fmhip_init_devices", "fmhip_device_count", "fmhip_set_thread_engines", "fmhip_shutdown", "fmhip_is_initialized", "fmhip_abi_version", "fmhip_last_error",
    "fmhip_device_info", "fmhip_synchronize", "fmhip_get_stream",
    "fmhip_vec_create_from_double", "fmhip_vec_create_from_float", "fmhip_vec_create_filled",
    "fmhip_vec_create_uninitialized", "fmhip_vec_retain", "fmhip_vec_release", "fmhip_vec_size", "fmhip_vec_give_up_values",
    "fmhip_vec_read_double", "fmhip_vec_read_float", "fmhip_vec_device_ptr",
    "fmhip_call_v1s0", "fmhip_call_v1s1", "fmhip_call_v2s0", "fmhip_call_v2s1", "fmhip_call_v3s0",
    "fmhip_set_fusion", "fmhip_flush", "fmhip_fusion_hold", "fmhip_set_step_grouping", "fmhip_graph_clone", "fmhip_graph_scalars", "fmhip_set_math_mode",
    "fmhip_reduce_moments", "fmhip_reduce_moments_batch", "fmhip_reduce_moments_batch_device", "fmhip_reduce_moments_device",
    "fmhip_reduce_moments_batch_begin", "fmhip_reduce_moments_batch_end", "fmhip_reduce_moments_batch_devices", "fmhip_get_stream_of", "fmhip_expectation_collective",
    "fmhip_set_expectation_comm", "fmhip_expectation_world", "fmhip_expectation_combine",
    "fmhip_program_create", "fmhip_program_release", "fmhip_program_launch_count", "fmhip_program_shape",
    "fmhip_program_run", "fmhip_program_run_into",
    "fmhip_set_jit", "fmhip_jit_wait", "fmhip_jit_stats", "fmhip_program_tier", "fmhip_program_source",
    "fmhip_bm_generate", "fmhip_mersenne_increments", "fmhip_bm_generate_mersenne", "fmhip_inverse_normal_cdf",
    "fmhip_pool_clean", "fmhip_pool_purge", "fmhip_pool_stats",
    "fmhip_profile_enable", "fmhip_profile_read", "fmhip_traffic_stats", "fmhip_engine_stats",
]

OK = 0
ERR_INVALID_HANDLE, ERR_SIZE_MISMATCH, ERR_OUT_OF_MEMORY, ERR_HIP = -1, -2, -3, -4
ERR_INVALID_ARGUMENT, ERR_NOT_INITIALIZED, ERR_UNSUPPORTED, ERR_PROGRAM_LIMIT = -5, -6, -7, -8


# fmhip_gather_fn: int (*)(void* context, const double* local, int count_doubles, double* gathered)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double))


class Moments(C.Structure):
    _fields_ = [("sum", C.c_double), ("sumsq", C.c_double), ("min", C.c_double), ("max", C.c_double)]


class ProgOp(C.Structure):
    _fields_ = [("opcode", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("scalar", C.c_double)]


class PoolStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "bytes_reserved", "bytes_in_use", "bytes_cached", "device_bytes_free", "device_bytes_total",
        "n_alloc_hits", "n_alloc_misses", "n_live_vectors", "n_kernel_launches", "n_ops_executed")]


class EngineStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "size", "kernel_launches", "specialised_launches", "interpreter_launches", "algorithmic_bytes", "algorithmic_bytes_written",
        "values_deferred", "values_deferred_now", "values_demanded", "pending_operations", "peak_bytes_reserved",
        "late_releases_while_waiting", "late_releases_at_once", "late_release_nanoseconds", "merged_launches", "merged_chains", "common_rows")]


class FmhipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"fmhip error {code}: {message}")
        self.code = code


def _source_hash() -> str:
    """Content hash of everything libfmhip.so is built from (content, not mtimes: a snapshot copy of the tree keeps the former)."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    files += [os.path.join(_HERE, "host", f) for f in sorted(os.listdir(os.path.join(_HERE, "host")))]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "fmhip.h"))
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


_HASH_PATH = LIB_PATH + ".srchash"


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 … → lib/libfmhip.so (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    with open(_HASH_PATH, "w") as fh:
        fh.write(_source_hash())
    return LIB_PATH


def _stale() -> bool:
    """True when the library on disk was not built from the sources on disk (edited csrc, or never built)."""
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(_HASH_PATH) as fh:
            return fh.read().strip() != _source_hash()
    except OSError:
        return True


_lib = None


def lib():
    """The loaded library; builds it if the source tree is newer (never silently substitutes anything)."""
    global _lib
    if _lib is not None:
        return _lib
    # NOTE: PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7).  A process that uses both
    # must `import torch` BEFORE this library is loaded, so that the loader resolves our NEEDED
    # libamdhip64.so.7 to the runtime already in the process (one HIP runtime, shared device pointers).
    if _stale():
        # One builder at a time (every rank of a torchrun launch imports this module at once): the others wait for the lock and then find
        # the library current.  A box with a prebuilt library but no compiler keeps working: the existing library is loaded, loudly.
        import fcntl
        os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
        with open(LIB_PATH + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if _stale():
                    try:
                        build()
                    except Exception as e:
                        if not os.path.exists(LIB_PATH):
                            raise
                        import warnings
                        warnings.warn(f"libfmhip.so is older than its sources and rebuilding it failed ({e}); loading the existing library", RuntimeWarning)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing — the HIP extension is required (no CPU fallback exists)")
    L = C.CDLL(LIB_PATH)
    vec, i64, i32, dbl, vp = C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_void_p
    pv = C.POINTER(C.c_int64)
    sig = {
        "fmhip_init": [i32], "fmhip_init_devices": [C.POINTER(C.c_int), i32], "fmhip_device_count": [C.POINTER(C.c_int)], "fmhip_set_thread_engines": [i32, C.POINTER(C.c_int)], "fmhip_shutdown": [], "fmhip_is_initialized": [], "fmhip_abi_version": [],
        "fmhip_device_info": [C.c_char_p, i32, C.POINTER(i32), C.POINTER(i64)],
        "fmhip_synchronize": [], "fmhip_get_stream": [C.POINTER(vp)],
        "fmhip_vec_create_from_double": [C.POINTER(dbl), i64, pv],
        "fmhip_vec_create_from_float": [C.POINTER(C.c_float), i64, pv],
        "fmhip_vec_create_filled": [i64, dbl, pv], "fmhip_vec_create_uninitialized": [i64, pv],
        "fmhip_vec_retain": [vec], "fmhip_vec_release": [vec], "fmhip_vec_size": [vec, C.POINTER(i64)],
        "fmhip_vec_read_double": [vec, C.POINTER(dbl), i64], "fmhip_vec_read_float": [vec, C.POINTER(C.c_float), i64],
        "fmhip_vec_device_ptr": [vec, C.POINTER(vp)],
        "fmhip_call_v1s0": [i32, vec, pv], "fmhip_call_v1s1": [i32, vec, dbl, pv],
        "fmhip_call_v2s0": [i32, vec, vec, pv], "fmhip_call_v2s1": [i32, vec, vec, dbl, pv],
        "fmhip_call_v3s0": [i32, vec, vec, vec, pv],
        "fmhip_set_fusion": [i32, C.POINTER(i32)], "fmhip_flush": [], "fmhip_fusion_hold": [i32, C.POINTER(i32)], "fmhip_set_step_grouping": [i32, C.POINTER(i32)], "fmhip_set_math_mode": [i32, C.POINTER(i32)],
        "fmhip_graph_clone": [pv, i32, i32, pv, pv, i32, C.POINTER(dbl), i32, pv], "fmhip_graph_scalars": [pv, i32, C.POINTER(dbl), i32, C.POINTER(i32)],
        "fmhip_reduce_moments": [vec, dbl, C.POINTER(Moments)], "fmhip_reduce_moments_batch": [pv, i32, C.POINTER(dbl), C.POINTER(Moments)], "fmhip_reduce_moments_batch_device": [pv, i32, C.POINTER(dbl), vp], "fmhip_reduce_moments_device": [vec, dbl, vp],
        "fmhip_reduce_moments_batch_devices": [pv, i32, C.POINTER(dbl), C.POINTER(vp), i32], "fmhip_get_stream_of": [i32, C.POINTER(vp)], "fmhip_expectation_collective": [C.POINTER(i32), C.c_char_p, i32],
        "fmhip_reduce_moments_batch_begin": [pv, i32, C.POINTER(dbl), pv], "fmhip_vec_give_up_values": [pv, i32], "fmhip_reduce_moments_batch_end": [i64, C.POINTER(Moments), i32],
        "fmhip_program_create": [C.POINTER(ProgOp), i32, i32, C.POINTER(C.c_int32), i32, C.POINTER(C.c_int32), i32, pv],
        "fmhip_program_release": [i64], "fmhip_program_launch_count": [i64, C.POINTER(i32)],
        "fmhip_program_shape": [i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
        "fmhip_set_expectation_comm": [i32, i32, GATHER_FN, vp], "fmhip_expectation_world": [C.POINTER(i32), C.POINTER(i32)],
        "fmhip_expectation_combine": [C.POINTER(Moments), i32, i32, C.POINTER(Moments)],
        "fmhip_program_run": [i64, i32, pv, pv, C.POINTER(dbl), C.POINTER(Moments), vp],
        "fmhip_program_run_into": [i64, i32, pv, pv, C.POINTER(dbl), C.POINTER(Moments), vp],
        "fmhip_bm_generate": [i64, i32, i32, i64, i64, C.POINTER(dbl), pv],
        "fmhip_mersenne_increments": [C.c_int32, i32, i32, i64, C.POINTER(dbl), C.POINTER(dbl)],
        "fmhip_bm_generate_mersenne": [C.c_int32, i32, i32, i64, C.POINTER(dbl), pv],
        "fmhip_pool_clean": [], "fmhip_pool_purge": [], "fmhip_pool_stats": [C.POINTER(PoolStats)],
        "fmhip_set_jit": [i32, C.POINTER(i32)], "fmhip_jit_wait": [],
        "fmhip_jit_stats": [C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(dbl), C.POINTER(i64)],
        "fmhip_program_tier": [i64, C.POINTER(i32), C.POINTER(i32)],
        "fmhip_program_source": [C.POINTER(ProgOp), i32, i32, C.POINTER(i32), i32, C.POINTER(i32), i32, C.c_char_p, i64, C.POINTER(i64)],
        "fmhip_traffic_stats": [C.POINTER(i64), C.POINTER(i64)],
        "fmhip_engine_stats": [C.POINTER(EngineStats)],
        "fmhip_profile_enable": [i32], "fmhip_profile_read": [C.POINTER(dbl), C.POINTER(i64)],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.fmhip_inverse_normal_cdf.argtypes = [dbl]
    L.fmhip_inverse_normal_cdf.restype = dbl
    L.fmhip_last_error.argtypes = []
    L.fmhip_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(status: int) -> None:
    if status != OK:
        raise FmhipError(status, lib().fmhip_last_error().decode("utf-8", "replace"))
