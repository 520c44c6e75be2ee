"""finmath-hip: MI355X-native RandomVariable / RandomVariableFactory / BrownianMotion engine.

Layers (bottom-up): csrc/kernels.hip (hand-written gfx950 kernels) → csrc/runtime.cpp (pool, handles,
program compiler, lazy fusion) → csrc/abi.cpp = the C-ABI of include/fmhip.h (libfmhip.so) →
this package: ctypes binding (_native) and the host-side mirror of the reference's Java classes.

There is NO CPU fallback: every stochastic operation runs in libfmhip.so on the GPU or raises.
The directory name contains hyphens; import it with
``importlib.import_module("finmath-lib-cuda-extensions_amd")``.
"""
from __future__ import annotations

import ctypes as _C

from . import _native
from ._native import FmhipError, Moments, PoolStats, ProgOp, build
from .random_variable import OP, DeviceVector, RandomVariableHip, RandomVariableHipFactory
from .brownian_motion import BrownianMotionHip, BrownianMotionFromMersenneRandomNumbers, TimeDiscretization, mersenne_increments
from .program import Program
from .differentiable import RandomVariableDifferentiableAAD, RandomVariableDifferentiableAADFactory


def lib():
    return _native.lib()


_atexit_registered = False


def _shutdown_at_exit() -> None:
    # interpreter exit: stop the background compiler thread and release the device before the C runtime's exit handlers run
    try:
        if _native._lib is not None and _native._lib.fmhip_is_initialized():
            _native._lib.fmhip_shutdown()
    except Exception:
        pass


def init(device_index: int = -1) -> None:
    """Bind this process to one GPU (one rank per GPU). -1: FMHIP_DEVICE_INDEX, LOCAL_RANK, else 0."""
    global _atexit_registered
    _native.check(lib().fmhip_init(int(device_index)))
    if not _atexit_registered:
        import atexit
        atexit.register(_shutdown_at_exit)
        _atexit_registered = True


def init_devices(devices) -> None:
    """ONE process, several devices (include/fmhip.h: fmhip_init_devices): every vector is cut into blocks of paths, block d on devices[d];
    everything else of this package works unchanged.  An index may repeat (shards on separate streams of one device)."""
    global _atexit_registered
    arr = (_C.c_int * len(devices))(*[int(d) for d in devices])
    _native.check(lib().fmhip_init_devices(arr, len(devices)))
    if not _atexit_registered:
        import atexit
        atexit.register(_shutdown_at_exit)
        _atexit_registered = True


def device_count() -> int:
    c = _C.c_int(0)
    _native.check(lib().fmhip_device_count(_C.byref(c)))
    return c.value


def set_thread_engines(enabled: bool = True) -> bool:
    """An engine per caller thread on the one device (fmhip_set_thread_engines): threads that simulate side by side record without meeting
    each other.  Returns the previous setting.  Ends with shutdown()."""
    prev = _C.c_int(0)
    _native.check(lib().fmhip_set_thread_engines(1 if enabled else 0, _C.byref(prev)))
    return prev.value != 0


def shutdown() -> None:
    _native.check(lib().fmhip_shutdown())


def synchronize() -> None:
    _native.check(lib().fmhip_synchronize())


def clean() -> None:
    """RandomVariableCuda.clean() (RandomVariableCuda.java:750-752)."""
    _native.check(lib().fmhip_pool_clean())


def purge() -> None:
    """RandomVariableCuda.purge() (RandomVariableCuda.java:754-756) — the reference's tests call it in @After."""
    import gc
    gc.collect()
    _native.check(lib().fmhip_pool_purge())


def set_fusion(enabled: bool) -> bool:
    prev = _C.c_int(0)
    _native.check(lib().fmhip_set_fusion(1 if enabled else 0, _C.byref(prev)))
    return bool(prev.value)


def fusion_hold(hold) -> int:
    """While held, pending chains run only when a value is needed or at flush() — many independent chains of identical
    structure recorded under a hold are batched as rows of the same launches (include/fmhip.h: fmhip_fusion_hold).
    `hold`: False / 0 = off, True / 1 = hard hold, 2 = soft hold (lifted by the engine beyond 32 k pending methods).
    Returns the previous setting as that integer, so that restoring it never turns a soft hold into a hard one."""
    prev = _C.c_int(0)
    _native.check(lib().fmhip_fusion_hold(int(hold), _C.byref(prev)))
    return prev.value


def reduce_moments_batch_begin(vectors, shifts=None) -> int:
    """Enqueues ONE launch that takes {Σ, Σ², min, max} of every vector and returns a ticket at once (include/fmhip.h:
    fmhip_reduce_moments_batch_begin); reduce_moments_batch_end(ticket, len(vectors)) waits for that launch only."""
    k = len(vectors)
    handles = (_C.c_int64 * k)(*[getattr(v, "handle", v) for v in vectors])
    sh = (_C.c_double * k)(*[float(x) for x in shifts]) if shifts is not None else None
    ticket = _C.c_int64(0)
    _native.check(lib().fmhip_reduce_moments_batch_begin(handles, k, sh, _C.byref(ticket)))
    return ticket.value


def give_up_values(vectors) -> None:
    """The caller wants only the EXPECTATIONS of these vectors: a pending vector that nobody else references is then computed by a
    launch that takes its moments and does not store it (include/fmhip.h: fmhip_vec_give_up_values)."""
    k = len(vectors)
    handles = (_C.c_int64 * k)(*[getattr(v, "handle", v) for v in vectors])
    _native.check(lib().fmhip_vec_give_up_values(handles, k))


def reduce_moments_batch_end(ticket: int, count: int):
    out = (_native.Moments * count)()
    _native.check(lib().fmhip_reduce_moments_batch_end(int(ticket), out, count))
    return list(out)


_expectation_gather = None          # keeps the ctypes callback of the expectation communicator alive


def set_expectation_comm(world: int, rank: int, gather=None) -> None:
    """fmhip_set_expectation_comm: Monte-Carlo paths sharded over `world` processes (one GPU each; this process holds the paths
    [rank·n, (rank+1)·n) of every vector).  `gather(local)` receives this rank's moments as a float64 array and returns every
    rank's, shape (world, len(local)), in rank order (an all-gather: torch.distributed, mpi4py …).  From then on getAverage(),
    getVariance(), getMin(), getMax() and fm.reduce_moments* return the moments of the GLOBAL vectors on every rank — sums added
    in rank order, so every rank sees the same bits.  world = 1 (or gather None) removes the communicator."""
    import numpy as _np
    global _expectation_gather
    if gather is None or world <= 1:
        _native.check(lib().fmhip_set_expectation_comm(1, 0, _native.GATHER_FN(0), None))
        _expectation_gather = None
        return

    def _thunk(_ctx, local, count, gathered):
        try:
            mine = _np.ctypeslib.as_array(local, shape=(count,)).copy()
            everyone = _np.ascontiguousarray(gather(mine), dtype=_np.float64).reshape(world, count)
            _np.ctypeslib.as_array(gathered, shape=(world * count,))[:] = everyone.ravel()
            return 0
        except Exception:               # an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1
    cb = _native.GATHER_FN(_thunk)
    _native.check(lib().fmhip_set_expectation_comm(int(world), int(rank), cb, None))
    _expectation_gather = cb


def expectation_world():
    """(world, rank) of the expectation communicator; (1, 0) without one."""
    w, r = _C.c_int(1), _C.c_int(0)
    _native.check(lib().fmhip_expectation_world(_C.byref(w), _C.byref(r)))
    return w.value, r.value


def expectation_combine(gathered):
    """fmhip_expectation_combine (host only): gathered[rank][vector] = (sum, sumsq, min, max) → the global moments per vector."""
    import numpy as _np
    g = _np.ascontiguousarray(gathered, dtype=_np.float64)
    world, count = g.shape[0], g.shape[1]
    out = _np.zeros((count, 4), dtype=_np.float64)
    _native.check(lib().fmhip_expectation_combine(g.ctypes.data_as(_C.POINTER(_native.Moments)), world, count, out.ctypes.data_as(_C.POINTER(_native.Moments))))
    return out


def set_step_grouping(steps: int) -> int:
    """fmhip_set_step_grouping: the engine keeps the methods recorded between `steps` time-step boundaries (first use of a Brownian
    increment with a new time index) pending and executes them together; 0 = off.  Returns the previous setting."""
    prev = _C.c_int(0)
    _native.check(lib().fmhip_set_step_grouping(int(steps), _C.byref(prev)))
    return prev.value


class holding:
    """`with fm.holding(): …` — record under fusion_hold(True), restore the previous setting on exit (no flush)."""

    def __enter__(self):
        self._prev = fusion_hold(True)
        return self

    def __exit__(self, *exc):
        fusion_hold(self._prev)
        return False


MATH_EXACT, MATH_FAST = 0, 1


def set_math_mode(mode: int) -> int:
    """MATH_EXACT (default): exp/log in fp64, narrowed once; MATH_FAST: hardware exp/log, within 2 fp32 ulp."""
    prev = _C.c_int(0)
    _native.check(lib().fmhip_set_math_mode(int(mode), _C.byref(prev)))
    return prev.value


JIT_OFF, JIT_AUTO, JIT_SYNC = 0, 1, 2


def set_jit(mode: int) -> int:
    """Execution tier policy: JIT_OFF = interpreter kernel only; JIT_AUTO (default) = explicit programs and hot lazy programs
    are compiled to specialised kernels in the background; JIT_SYNC = compiled before their first launch.  Returns the
    previous mode.  Both tiers are bit-identical."""
    prev = _C.c_int(0)
    _native.check(lib().fmhip_set_jit(int(mode), _C.byref(prev)))
    return prev.value


def jit_wait() -> None:
    """Block until the background compiler is idle."""
    _native.check(lib().fmhip_jit_wait())


def traffic_stats() -> tuple:
    """(algorithmic bytes of all launches so far: 4 B x paths x (vectors read + vectors stored) per launch, launches on specialised kernels)."""
    b, l = _C.c_int64(0), _C.c_int64(0)
    _native.check(lib().fmhip_traffic_stats(_C.byref(b), _C.byref(l)))
    return b.value, l.value


def engine_stats() -> dict:
    """fmhip_engine_stats: launches by tier, algorithmic bytes read + written and written alone, values left unstored (deferred) / wanted
    after all (demanded), pending operations, the pool's peak."""
    st = _native.EngineStats()
    _native.check(lib().fmhip_engine_stats(_C.byref(st)))
    return {n: getattr(st, n) for n, _ in _native.EngineStats._fields_ if n != "size"}


def jit_stats() -> dict:
    c, f, p, s, d = _C.c_int64(0), _C.c_int64(0), _C.c_int64(0), _C.c_double(0), _C.c_int64(0)
    _native.check(lib().fmhip_jit_stats(_C.byref(c), _C.byref(f), _C.byref(p), _C.byref(s), _C.byref(d)))
    return {"compiled": c.value, "failed": f.value, "pending": p.value, "compile_seconds": s.value, "disk_cache_hits": d.value}


def graph_clone(roots, n_copies: int, leaf_from=(), leaf_to=(), scalars=None):
    """fmhip_graph_clone: `n_copies` copies of the pending expressions below `roots` (DeviceVector objects).  Copy j reads
    leaf_to[j][i] wherever the original reads leaf_from[i] and takes its scalar operands, in recording order, from scalars[j]
    (None: the original's).  Returns a list of n_copies lists of DeviceVector, one per root."""
    import numpy as _np
    n_roots, n_map = len(roots), len(leaf_from)
    r = (_C.c_int64 * n_roots)(*[v.handle for v in roots])
    lf = (_C.c_int64 * max(1, n_map))(*[v.handle for v in leaf_from])
    lt = (_C.c_int64 * max(1, n_map * n_copies))(*[v.handle for row in leaf_to for v in row])
    sc, n_sc = None, 0
    if scalars is not None:
        a = _np.ascontiguousarray(scalars, dtype=_np.float64).reshape(n_copies, -1)
        n_sc = a.shape[1]
        sc = a.ctypes.data_as(_C.POINTER(_C.c_double))
    out = (_C.c_int64 * max(1, n_roots * n_copies))()
    _native.check(lib().fmhip_graph_clone(r, n_roots, n_copies, lf, lt, n_map, sc, n_sc, out))
    return [[DeviceVector(out[j * n_roots + k], roots[k].n) for k in range(n_roots)] for j in range(n_copies)]


def graph_scalars(roots):
    """fmhip_graph_scalars: the scalar operands of the pending graph below `roots`, in recording order."""
    import numpy as _np
    r = (_C.c_int64 * len(roots))(*[v.handle for v in roots])
    n = _C.c_int(0)
    _native.check(lib().fmhip_graph_scalars(r, len(roots), None, 0, _C.byref(n)))
    out = _np.zeros(max(1, n.value), dtype=_np.float64)
    _native.check(lib().fmhip_graph_scalars(r, len(roots), out.ctypes.data_as(_C.POINTER(_C.c_double)), n.value, _C.byref(n)))
    return out[:n.value]


def flush() -> None:
    _native.check(lib().fmhip_flush())


def pool_stats() -> PoolStats:
    s = PoolStats()
    _native.check(lib().fmhip_pool_stats(_C.byref(s)))
    return s


def device_info():
    name = _C.create_string_buffer(256)
    cus, hbm = _C.c_int(0), _C.c_int64(0)
    _native.check(lib().fmhip_device_info(name, 256, _C.byref(cus), _C.byref(hbm)))
    return name.value.decode(), cus.value, hbm.value


def stream_ptr() -> int:
    p = _C.c_void_p(0)
    _native.check(lib().fmhip_get_stream(_C.byref(p)))
    return p.value or 0


def profile_enable(enabled: bool) -> None:
    _native.check(lib().fmhip_profile_enable(1 if enabled else 0))


def profile_read():
    """(sum of fused-program kernel durations in ms, number of launches) since the last read."""
    ms, n = _C.c_double(0.0), _C.c_int64(0)
    _native.check(lib().fmhip_profile_read(_C.byref(ms), _C.byref(n)))
    return ms.value, n.value
