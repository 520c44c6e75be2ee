"""CPU oracle for the RandomVariable / BrownianMotion hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package.  The product package never does; it fails loudly if its HIP library is missing.

Two layers:

* ``lib`` — ctypes binding of ``libfm_oracle.so`` (plain C, ``oracle/*.c``): array-level restatement of
  the reference's fp32 CPU twin ``RandomVariableFromFloatArray.java`` (file:line cited in the C sources).
* ``RandomVariableFromFloatArray`` / ``RandomVariableFloatFactory`` — object-level restatement of the same
  Java class (deterministic/stochastic dispatch, filtration-time propagation, type priority), so parity
  tests can run one lambda against both factories exactly as the reference's
  ``RandomVariableGPUTest.testRandomVariableOperators`` does (RandomVariableGPUTest.java:191-360).

Parity status: pinned by the known-answer values of the reference's own tests
(tests/test_oracle_known_answers.py); there are no golden files in the reference.  Parity with
finmath-lib's double class (not vendored) is unpinned.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfm_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("rv_float.c", "rv_double.c", "java_random.c", "philox_normal.c", "fm_oracle.h", "normal_table.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def _load():
    build()
    lib = C.CDLL(_LIB_PATH)
    f32p, f64p, i64, dbl, i32 = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_int64, C.c_double, C.c_int
    lib.orc_f_v1s0.argtypes = [i32, f32p, i64, f32p]
    lib.orc_f_v1s1.argtypes = [i32, f32p, dbl, i64, f32p]
    lib.orc_f_v2s0.argtypes = [i32, f32p, f32p, i64, f32p]
    lib.orc_f_v2s1.argtypes = [i32, f32p, f32p, dbl, i64, f32p]
    lib.orc_f_v3s0.argtypes = [i32, f32p, f32p, f32p, i64, f32p]
    lib.orc_d_v1s0.argtypes = [i32, f64p, i64, f64p]
    lib.orc_d_v1s1.argtypes = [i32, f64p, dbl, i64, f64p]
    lib.orc_d_v2s0.argtypes = [i32, f64p, f64p, i64, f64p]
    lib.orc_d_v2s1.argtypes = [i32, f64p, f64p, dbl, i64, f64p]
    lib.orc_d_v3s0.argtypes = [i32, f64p, f64p, f64p, i64, f64p]
    for name in ("orc_f_average", "orc_f_variance", "orc_f_min", "orc_f_max"):
        getattr(lib, name).argtypes = [f32p, i64]
        getattr(lib, name).restype = dbl
    for name in ("orc_f_average_weighted", "orc_f_variance_weighted"):
        getattr(lib, name).argtypes = [f32p, f32p, i64]
        getattr(lib, name).restype = dbl
    for name in ("orc_d_average", "orc_d_variance", "orc_d_min", "orc_d_max"):
        getattr(lib, name).argtypes = [f64p, i64]
        getattr(lib, name).restype = dbl
    lib.orc_f_moments.argtypes = [f32p, i64, dbl, f64p]
    lib.orc_f_moments.restype = None
    lib.orc_f_quantile.argtypes = [f32p, i64, dbl]
    lib.orc_f_quantile.restype = dbl
    lib.orc_f_from_double.argtypes = [f64p, i64, f32p]
    lib.orc_f_from_double.restype = None
    lib.orc_java_random_doubles.argtypes = [i64, i64, f64p]
    lib.orc_java_random_doubles.restype = None
    lib.orc_java_random_next_int.argtypes = [i64, i32]
    lib.orc_java_random_next_int.restype = C.c_int32
    u32p = C.POINTER(C.c_uint32)
    lib.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
    lib.orc_philox4x32_10.restype = None
    lib.orc_normal4.argtypes = [i64, C.c_uint64, C.c_uint32, f32p]
    lib.orc_normal4.restype = None
    lib.orc_bm_increment.argtypes = [i64, C.c_uint32, i64, i64, C.c_float, f32p]
    lib.orc_bm_increment.restype = None
    return lib


lib = _load()

# opcode numbering is the public one of include/fmhip.h
OP = dict(CAP_S=1, FLOOR_S=2, ADD_S=3, SUB_S=4, BUS_S=5, MULT_S=6, DIV_S=7, VID_S=8, POW_S=9,
          SQUARED=10, SQRT=11, EXP=12, LOG=13, INVERT=14, ABS=15, SIN=16, COS=17, ISNAN=18,
          CAP=19, FLOOR=20, ADD=21, SUB=22, MULT=23, DIV=24,
          ACCRUE=25, DISCOUNT=26, ADDPRODUCT_VS=27,
          ADDPRODUCT=28, ADDRATIO=29, SUBRATIO=30, CHOOSE=31)
V1S0 = ("SQUARED", "SQRT", "EXP", "LOG", "INVERT", "ABS", "SIN", "COS", "ISNAN")
V1S1 = ("CAP_S", "FLOOR_S", "ADD_S", "SUB_S", "BUS_S", "MULT_S", "DIV_S", "VID_S", "POW_S")
V2S0 = ("CAP", "FLOOR", "ADD", "SUB", "MULT", "DIV")
V2S1 = ("ACCRUE", "DISCOUNT", "ADDPRODUCT_VS")
V3S0 = ("ADDPRODUCT", "ADDRATIO", "SUBRATIO", "CHOOSE")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _code(op):
    return OP[op] if isinstance(op, str) else int(op)


# ------------------------------------------------------------------ array level, float twin
def f_v1s0(op, a):
    a = _f32(a); out = np.empty_like(a)
    assert lib.orc_f_v1s0(_code(op), _p32(a), a.size, _p32(out)) == 0
    return out


def f_v1s1(op, a, s):
    a = _f32(a); out = np.empty_like(a)
    assert lib.orc_f_v1s1(_code(op), _p32(a), float(s), a.size, _p32(out)) == 0
    return out


def f_v2s0(op, a, b):
    a, b = _f32(a), _f32(b); out = np.empty_like(a)
    assert a.size == b.size
    assert lib.orc_f_v2s0(_code(op), _p32(a), _p32(b), a.size, _p32(out)) == 0
    return out


def f_v2s1(op, a, b, s):
    a, b = _f32(a), _f32(b); out = np.empty_like(a)
    assert a.size == b.size
    assert lib.orc_f_v2s1(_code(op), _p32(a), _p32(b), float(s), a.size, _p32(out)) == 0
    return out


def f_v3s0(op, a, b, c):
    a, b, c = _f32(a), _f32(b), _f32(c); out = np.empty_like(a)
    assert a.size == b.size == c.size
    assert lib.orc_f_v3s0(_code(op), _p32(a), _p32(b), _p32(c), a.size, _p32(out)) == 0
    return out


def f_apply(op, *args):
    """Dispatch on the opcode's call shape; the scalar (if any) is the last argument."""
    name = op if isinstance(op, str) else {v: k for k, v in OP.items()}[int(op)]
    if name in V1S0: return f_v1s0(name, *args)
    if name in V1S1: return f_v1s1(name, *args)
    if name in V2S0: return f_v2s0(name, *args)
    if name in V2S1: return f_v2s1(name, *args)
    if name in V3S0: return f_v3s0(name, *args)
    raise KeyError(name)


def f_average(x): x = _f32(x); return lib.orc_f_average(_p32(x), x.size)
def f_variance(x): x = _f32(x); return lib.orc_f_variance(_p32(x), x.size)
def f_min(x): x = _f32(x); return lib.orc_f_min(_p32(x), x.size)
def f_max(x): x = _f32(x); return lib.orc_f_max(_p32(x), x.size)
def f_average_weighted(x, w): x, w = _f32(x), _f32(w); return lib.orc_f_average_weighted(_p32(x), _p32(w), x.size)
def f_variance_weighted(x, w): x, w = _f32(x), _f32(w); return lib.orc_f_variance_weighted(_p32(x), _p32(w), x.size)
def f_quantile(x, q): x = _f32(x); return lib.orc_f_quantile(_p32(x), x.size, float(q))


def f_moments(x, shift=0.0):
    x = _f32(x); out = np.zeros(4, dtype=np.float64)
    lib.orc_f_moments(_p32(x), x.size, float(shift), _p64(out))
    return out


def f_from_double(x):
    x = _f64(x); out = np.empty(x.size, dtype=np.float32)
    lib.orc_f_from_double(_p64(x), x.size, _p32(out))
    return out


# ------------------------------------------------------------------ array level, double stand-in
def d_apply(op, *args):
    name = op if isinstance(op, str) else {v: k for k, v in OP.items()}[int(op)]
    code = OP[name]
    if name in V1S0:
        a = _f64(args[0]); out = np.empty_like(a)
        assert lib.orc_d_v1s0(code, _p64(a), a.size, _p64(out)) == 0
    elif name in V1S1:
        a = _f64(args[0]); out = np.empty_like(a)
        assert lib.orc_d_v1s1(code, _p64(a), float(args[1]), a.size, _p64(out)) == 0
    elif name in V2S0:
        a, b = _f64(args[0]), _f64(args[1]); out = np.empty_like(a)
        assert lib.orc_d_v2s0(code, _p64(a), _p64(b), a.size, _p64(out)) == 0
    elif name in V2S1:
        a, b = _f64(args[0]), _f64(args[1]); out = np.empty_like(a)
        assert lib.orc_d_v2s1(code, _p64(a), _p64(b), float(args[2]), a.size, _p64(out)) == 0
    else:
        a, b, c = _f64(args[0]), _f64(args[1]), _f64(args[2]); out = np.empty_like(a)
        assert lib.orc_d_v3s0(code, _p64(a), _p64(b), _p64(c), a.size, _p64(out)) == 0
    return out


def d_average(x): x = _f64(x); return lib.orc_d_average(_p64(x), x.size)
def d_variance(x): x = _f64(x); return lib.orc_d_variance(_p64(x), x.size)
def d_min(x): x = _f64(x); return lib.orc_d_min(_p64(x), x.size)
def d_max(x): x = _f64(x); return lib.orc_d_max(_p64(x), x.size)


# ------------------------------------------------------------------ generators
def java_random_doubles(seed: int, n: int) -> np.ndarray:
    """`new java.util.Random(seed)`; n calls of nextDouble()."""
    out = np.empty(n, dtype=np.float64)
    lib.orc_java_random_doubles(int(seed), int(n), _p64(out))
    return out


def java_random_next_int(seed: int, skip: int = 0) -> int:
    return int(lib.orc_java_random_next_int(int(seed), int(skip)))


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32); k = np.asarray(key, dtype=np.uint32); o = np.zeros(4, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    lib.orc_philox4x32_10(c.ctypes.data_as(u32p), k.ctypes.data_as(u32p), o.ctypes.data_as(u32p))
    return o


def bm_increment(seed: int, stream: int, path_offset: int, n: int, sqrt_dt: float) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    lib.orc_bm_increment(int(seed), int(stream), int(path_offset), int(n), float(np.float32(sqrt_dt)), _p32(out))
    return out


def bm_generate(seed: int, dt, n_factors: int, n_paths: int, path_offset: int = 0):
    """[step][factor] -> float32[n_paths]; scaling (float)sqrt(dt) as
    BrownianMotionCudaWithRandomVariableCuda.java:170,175."""
    res = []
    for step, d in enumerate(dt):
        sq = np.float32(math.sqrt(d))
        res.append([bm_increment(seed, step * n_factors + f, path_offset, n_paths, sq) for f in range(n_factors)])
    return res


from .random_variable_float import RandomVariableFromFloatArray, RandomVariableFloatFactory  # noqa: E402,F401
from .random_variable_double import RandomVariableFromDoubleArray, RandomVariableFromArrayFactory  # noqa: E402,F401
