"""Host-side mirror of the reference's drop-in classes, above the C-ABI (Python flavour; the C++ flavour
with the same method set lives in host/).

    reference (Java)                                           here
    ---------------------------------------------------------------------------------------------
    RandomVariableCuda        (RandomVariableCuda.java)        RandomVariableHip
    RandomVariableCudaFactory (RandomVariableCudaFactory.java) RandomVariableHipFactory

Same method names, argument meaning, dispatch order and error behaviour as RandomVariableCuda
(``:line`` cites that file): a random variable is either a constant kept on the host as a double
(no device memory) or an fp32 device vector; binary methods check the type priority first, then take the
maximum of the filtration times, then dispatch on which operands are deterministic.

Deliberate deviations from RandomVariableCuda, each following the reference's own CPU twin
(RandomVariableFromFloatArray.java, ``twin:line``) where the GPU class is broken or unimplemented
(SURVEY.md Appendix A):
  * add/sub/bus(RandomVariable) stochastic branch returns newTime (the GPU class drops it, :1410,:1434,:1459); so do
    accrue/discount of a deterministic receiver with a stochastic rate (:1595-1596, :1615-1619; twin:1214-1219, :1242-1247),
    and discount does not short-cut a zero receiver to a constant (the twin returns the vector 0/(1+r·Δ), NaNs included);
    addRatio/subRatio carry the maximum of all three filtration times (twin:1395-1438; add(div(·)) drops a constant denominator's);
    vid(RandomVariable) with a constant argument takes the twin's newTime; its VALUE follows the GPU class (constant narrowed
    to fp32 like every scalar operand, :1528 — the twin divides the double constant, twin:1138; different only for constants
    that are not fp32 values);
  * cap(RandomVariable) handles "argument deterministic, receiver stochastic" (null dereference at :1546-1555);
  * vid(RandomVariable) priority branch calls div (twin:1116-1119; the GPU class calls vid, :1513-1516);
  * choose, isNaN, sin, cos are implemented (GPU class returns null / throws; twin:1264,1441,927,942);
  * getVariance() is the twin's two-pass Σ(x-mean)²/n (twin:360-382), evaluated on the device, instead of
    E[X²]-E[X]² on fp32-rounded squares (:891-901);
  * reductions run on the device and return 32 bytes instead of copying the vector to the host (:830-878).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _native as N

OP = dict(CAP_S=1, FLOOR_S=2, ADD_S=3, SUB_S=4, BUS_S=5, MULT_S=6, DIV_S=7, VID_S=8, POW_S=9,
          SQUARED=10, SQRT=11, EXP=12, LOG=13, INVERT=14, ABS=15, SIN=16, COS=17, ISNAN=18,
          CAP=19, FLOOR=20, ADD=21, SUB=22, MULT=23, DIV=24,
          ACCRUE=25, DISCOUNT=26, ADDPRODUCT_VS=27,
          ADDPRODUCT=28, ADDRATIO=29, SUBRATIO=30, CHOOSE=31)

TYPE_PRIORITY_DEFAULT = 20          # RandomVariableCuda.java:568


# ------------------------------------------------------------------ thin RAII wrapper of a vector handle
class DeviceVector:
    """Owns one fmhip_vec handle (released on garbage collection — explicit, never via device-memory polling)."""
    __slots__ = ("handle", "n")

    def __init__(self, handle: int, n: int):
        self.handle = handle
        self.n = n

    def __del__(self):
        h, self.handle = self.handle, 0
        if h and N is not None and N._lib is not None:
            try:
                N._lib.fmhip_vec_release(h)
            except Exception:
                pass

    # -- creation
    @staticmethod
    def from_host(values) -> "DeviceVector":
        a = np.asarray(values)
        out = C.c_int64(0)
        if a.dtype == np.float32:
            a = np.ascontiguousarray(a)
            N.check(N.lib().fmhip_vec_create_from_float(a.ctypes.data_as(C.POINTER(C.c_float)), a.size, C.byref(out)))
        else:
            a = np.ascontiguousarray(a, dtype=np.float64)
            N.check(N.lib().fmhip_vec_create_from_double(a.ctypes.data_as(C.POINTER(C.c_double)), a.size, C.byref(out)))
        return DeviceVector(out.value, a.size)

    @staticmethod
    def filled(n: int, value: float) -> "DeviceVector":
        out = C.c_int64(0)
        N.check(N.lib().fmhip_vec_create_filled(n, float(value), C.byref(out)))
        return DeviceVector(out.value, n)

    # -- the five launch helpers of the reference (callFunctionv1s0 … v3s0, RandomVariableCuda.java:483-557)
    def v1s0(self, op) -> "DeviceVector":
        out = C.c_int64(0)
        N.check(N.lib().fmhip_call_v1s0(OP[op], self.handle, C.byref(out)))
        return DeviceVector(out.value, self.n)

    def v1s1(self, op, s: float) -> "DeviceVector":
        out = C.c_int64(0)
        N.check(N.lib().fmhip_call_v1s1(OP[op], self.handle, float(s), C.byref(out)))
        return DeviceVector(out.value, self.n)

    def v2s0(self, op, b: "DeviceVector") -> "DeviceVector":
        out = C.c_int64(0)
        N.check(N.lib().fmhip_call_v2s0(OP[op], self.handle, b.handle, C.byref(out)))
        return DeviceVector(out.value, self.n)

    def v2s1(self, op, b: "DeviceVector", s: float) -> "DeviceVector":
        out = C.c_int64(0)
        N.check(N.lib().fmhip_call_v2s1(OP[op], self.handle, b.handle, float(s), C.byref(out)))
        return DeviceVector(out.value, self.n)

    def v3s0(self, op, b: "DeviceVector", c: "DeviceVector") -> "DeviceVector":
        out = C.c_int64(0)
        N.check(N.lib().fmhip_call_v3s0(OP[op], self.handle, b.handle, c.handle, C.byref(out)))
        return DeviceVector(out.value, self.n)

    # -- host access
    def to_float32(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float32)
        N.check(N.lib().fmhip_vec_read_float(self.handle, out.ctypes.data_as(C.POINTER(C.c_float)), self.n))
        return out

    def to_float64(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        N.check(N.lib().fmhip_vec_read_double(self.handle, out.ctypes.data_as(C.POINTER(C.c_double)), self.n))
        return out

    def moments(self, shift: float = 0.0) -> N.Moments:
        m = N.Moments()
        N.check(N.lib().fmhip_reduce_moments(self.handle, float(shift), C.byref(m)))
        return m

    def device_ptr(self) -> int:
        p = C.c_void_p(0)
        N.check(N.lib().fmhip_vec_device_ptr(self.handle, C.byref(p)))
        return p.value or 0


def _jmin(a, b):                    # java.lang.Math.min(double,double)
    if a != a: return a
    if a == 0.0 and b == 0.0 and math.copysign(1.0, b) < 0: return b
    return a if a <= b else b


def _jmax(a, b):
    if a != a: return a
    if a == 0.0 and b == 0.0 and math.copysign(1.0, a) < 0: return b
    return a if a >= b else b


def _f64(fn, *a):
    with np.errstate(all="ignore"):
        return float(fn(*[np.float64(x) for x in a]))


def _jpow(x, y):
    if y == 0.0: return 1.0
    if y != y: return y
    if math.isinf(y) and abs(x) == 1.0: return math.nan
    return _f64(np.power, x, y)


class RandomVariableHip:
    """Mirror of net.finmath.cuda.montecarlo.RandomVariableCuda."""

    __slots__ = ("time", "type_priority", "realizations", "value")

    def __init__(self, time, value, type_priority=TYPE_PRIORITY_DEFAULT):
        """(time, double) → constant (:683-689); (time, array) → stochastic, narrowed to fp32 and uploaded (:696-723);
        (time, DeviceVector) → wrap device memory (RandomVariableCuda.of, :618-646)."""
        self.time = float(time)
        self.type_priority = type_priority
        if isinstance(value, DeviceVector):
            self.realizations, self.value = value, math.nan
        elif np.isscalar(value):
            self.realizations, self.value = None, float(value)
        else:
            self.realizations, self.value = DeviceVector.from_host(value), math.nan

    @staticmethod
    def of(time, value, type_priority=TYPE_PRIORITY_DEFAULT):
        return RandomVariableHip(time, value, type_priority)

    # ---- accessors (:785-827, :1094-1131)
    def getFiltrationTime(self): return self.time
    def getTypePriority(self): return self.type_priority
    def isDeterministic(self): return self.realizations is None
    def size(self): return 1 if self.isDeterministic() else self.realizations.n
    def cache(self): return self

    def get(self, path_or_state):
        if self.isDeterministic(): return self.value
        raise NotImplementedError("UnsupportedOperationException: get(i) on a stochastic RandomVariableHip (:812-818)")

    def doubleValue(self):
        if self.isDeterministic(): return self.value
        raise NotImplementedError("UnsupportedOperationException: The random variable is non-deterministic (:1125-1131)")

    def getRealizations(self):
        if self.isDeterministic(): return np.array([self.value], dtype=np.float64)
        return self.realizations.to_float64()

    def equals(self, other):
        raise NotImplementedError("UnsupportedOperationException (:785)")

    # ---- reductions, on the device
    def _sample_size(self):
        """Paths behind an expectation: size() of this process's shard times the ranks of the expectation communicator
        (fmhip_set_expectation_comm; 1 without one) — the moments the engine returns are then those of the global vector."""
        w = C.c_int(1)
        N.check(N.lib().fmhip_expectation_world(C.byref(w), None))
        return self.size() * w.value

    def getMin(self):
        if self.isDeterministic(): return self.value
        return self.realizations.moments().min

    def getMax(self):
        if self.isDeterministic(): return self.value
        return self.realizations.moments().max

    def getAverage(self, probabilities=None):
        if probabilities is not None:
            return self.mult(probabilities).getAverage()            # :886-888
        if self.isDeterministic(): return self.value
        if self.size() == 0: return math.nan
        return self.realizations.moments().sum / self._sample_size()

    def getVariance(self, probabilities=None):
        if probabilities is not None:                                # twin:385-407  Σ (x-avg)² p  (no division)
            if self.isDeterministic(): return 0.0
            if self.size() == 0: return math.nan
            average = self.getAverage(probabilities)
            d = self.sub(average).squared().mult(probabilities)
            if d.isDeterministic(): return d.value
            return d.realizations.moments().sum
        if self.isDeterministic() or self.size() == 1: return 0.0
        if self.size() == 0: return math.nan
        average = self.getAverage()
        return self.realizations.moments(shift=average).sumsq / self._sample_size()     # twin:368-381

    def getSampleVariance(self):                                    # :904-913
        if self.isDeterministic() or self.size() == 1: return 0.0
        if self.size() == 0: return math.nan
        n = self._sample_size()
        return self.getVariance() * n / (n - 1)

    def getStandardDeviation(self, probabilities=None):             # :916-937
        if self.isDeterministic(): return 0.0
        if self.size() == 0: return math.nan
        return math.sqrt(self.getVariance(probabilities))

    def getStandardError(self, probabilities=None):                 # :940-967
        if self.isDeterministic(): return 0.0
        if self.size() == 0: return math.nan
        return self.getStandardDeviation(probabilities) / math.sqrt(self._sample_size())

    def average(self):                                               # :1280
        return RandomVariableHip(-math.inf, self.getAverage())

    # ---- host-side cold paths (sort on the host, as the reference does: :970-1091)
    def getQuantile(self, quantile):
        if self.isDeterministic(): return self.value
        if self.size() == 0: return math.nan
        s = np.sort(self.getRealizations())
        n = self.size()
        idx = min(max(int(math.floor((n + 1) * (1 - quantile) - 1 + 0.5)), 0), n - 1)      # :983
        return float(s[idx])

    def getQuantileExpectation(self, quantile_start, quantile_end):
        if self.isDeterministic(): return self.value
        if self.size() == 0: return math.nan
        if quantile_start > quantile_end: return self.getQuantileExpectation(quantile_end, quantile_start)
        s = np.sort(self.getRealizations())
        n = self.size()
        i0 = min(max(int(math.floor((n + 1) * quantile_start - 1 + 0.5)), 0), n - 1)
        i1 = min(max(int(math.floor((n + 1) * quantile_end - 1 + 0.5)), 0), n - 1)
        return float(s[i0:i1 + 1].sum() / (i1 - i0 + 1))

    def getHistogram(self, interval_points, standard_deviations=None):
        if standard_deviations is not None:                          # getHistogram(int, double) :1070-1091
            number_of_points = int(interval_points)
            center = self.getAverage()
            radius = standard_deviations * self.getStandardDeviation()
            step = (number_of_points - 1) / 2.0
            pts = np.empty(number_of_points); anchors = np.empty(number_of_points + 1)
            for i in range(number_of_points):
                alpha = (-(number_of_points - 1) / 2.0 + i) / step
                pts[i] = center + alpha * radius
                anchors[i] = center + alpha * radius - radius / (2 * step)
            anchors[number_of_points] = center + radius + radius / (2 * step)
            return [anchors, self.getHistogram(pts)]
        pts = np.asarray(interval_points, dtype=np.float64)           # :1026-1068
        hist = np.zeros(pts.size + 1)
        if self.isDeterministic():
            for k in range(pts.size):
                if self.value > pts[k]:
                    hist[k] = 1.0
                    break
            hist[pts.size] = 1.0
            return hist
        s = np.sort(self.getRealizations())
        idx = np.searchsorted(s, pts, side="right")
        prev = 0
        for k in range(pts.size):
            cur = max(int(idx[k]), prev)
            hist[k] = cur - prev
            prev = cur
        hist[pts.size] = s.size - prev
        if s.size > 0: hist /= s.size
        return hist

    # ---- helpers
    @staticmethod
    def _vec(rv) -> DeviceVector:
        """getRandomVariableCuda(rv).realizations (:759-766): foreign types are uploaded via getRealizations()."""
        if isinstance(rv, RandomVariableHip): return rv.realizations
        return DeviceVector.from_host(rv.getRealizations())

    def _det(self, value, time=None):
        return RandomVariableHip(self.time if time is None else time, value)

    def _sto(self, vec, time=None):
        return RandomVariableHip(self.time if time is None else time, vec)

    def _scalar(self, det, op, s):
        if self.isDeterministic(): return self._det(det(self.value, s))
        return self._sto(self.realizations.v1s1(op, s))

    def _unary(self, det, op):
        if self.isDeterministic(): return self._det(det(self.value))
        return self._sto(self.realizations.v1s0(op))

    # ---- scalar-operand and unary methods (:1172-1352)
    def cap(self, x):
        if not np.isscalar(x): return self._cap_rv(x)
        return self._scalar(_jmin, "CAP_S", x)
    def floor(self, x):
        if not np.isscalar(x): return self._floor_rv(x)
        return self._scalar(_jmax, "FLOOR_S", x)
    def add(self, x):
        if not np.isscalar(x): return self._add_rv(x)
        return self._scalar(lambda a, b: a + b, "ADD_S", x)
    def sub(self, x):
        if not np.isscalar(x): return self._sub_rv(x)
        return self._scalar(lambda a, b: a - b, "SUB_S", x)
    def bus(self, x):
        if not np.isscalar(x): return self._bus_rv(x)
        return self._scalar(lambda a, b: -a + b, "BUS_S", x)
    def mult(self, x):
        if not np.isscalar(x): return self._mult_rv(x)
        return self._scalar(lambda a, b: a * b, "MULT_S", x)
    def div(self, x):
        if not np.isscalar(x): return self._div_rv(x)
        return self._scalar(lambda a, b: _f64(np.divide, a, b), "DIV_S", x)
    def vid(self, x):
        if not np.isscalar(x): return self._vid_rv(x)
        return self._scalar(lambda a, b: _f64(np.divide, b, a), "VID_S", x)
    def pow(self, exponent): return self._scalar(_jpow, "POW_S", exponent)

    def squared(self):                                               # :1285-1292 (mult kernel with itself)
        if self.isDeterministic(): return self._det(self.value * self.value)
        return self._sto(self.realizations.v1s0("SQUARED"))
    def sqrt(self): return self._unary(lambda a: _f64(np.sqrt, a), "SQRT")
    def invert(self): return self._unary(lambda a: _f64(np.divide, 1.0, a), "INVERT")
    def abs(self): return self._unary(abs, "ABS")
    def exp(self): return self._unary(lambda a: _f64(np.exp, a), "EXP")
    def log(self): return self._unary(lambda a: _f64(np.log, a), "LOG")
    def sin(self): return self._unary(math.sin, "SIN")               # twin:927 (GPU class throws, :1355)
    def cos(self): return self._unary(math.cos, "COS")               # twin:942
    def isNaN(self): return self._unary(lambda a: 1.0 if a != a else 0.0, "ISNAN")   # twin:1441

    # ---- binary methods (:1391-1580)
    def _add_rv(self, rv):
        if rv.getTypePriority() > self.getTypePriority(): return rv.add(self)
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic(): return self._det(self.value + rv.doubleValue(), new_time)
        if self.isDeterministic(): return self._sto(self._vec(rv).v1s1("ADD_S", self.value), new_time)
        if rv.isDeterministic(): return self._sto(self.realizations.v1s1("ADD_S", rv.doubleValue()), new_time)
        return self._sto(self.realizations.v2s0("ADD", self._vec(rv)), new_time)

    def _sub_rv(self, rv):
        if rv.getTypePriority() > self.getTypePriority(): return rv.bus(self)
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic(): return self._det(self.value - rv.doubleValue(), new_time)
        if self.isDeterministic(): return self._sto(self._vec(rv).v1s1("BUS_S", self.value), new_time)
        if rv.isDeterministic(): return self._sto(self.realizations.v1s1("SUB_S", rv.doubleValue()), new_time)
        return self._sto(self.realizations.v2s0("SUB", self._vec(rv)), new_time)

    def _bus_rv(self, rv):
        if rv.getTypePriority() > self.getTypePriority(): return rv.sub(self)
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic(): return self._det(-self.value + rv.doubleValue(), new_time)
        if self.isDeterministic(): return self._sto(self._vec(rv).v1s1("SUB_S", self.value), new_time)
        if rv.isDeterministic(): return self._sto(self.realizations.v1s1("BUS_S", rv.doubleValue()), new_time)
        return self._sto(self._vec(rv).v2s0("SUB", self.realizations), new_time)       # flipped arguments, :1458

    def _mult_rv(self, rv):
        if rv.getTypePriority() > self.getTypePriority(): return rv.mult(self)
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic(): return self._det(self.value * rv.doubleValue(), new_time)
        if rv.isDeterministic(): return self.mult(rv.doubleValue())
        if self.isDeterministic(): return self._sto(self._vec(rv).v1s1("MULT_S", self.value), new_time)
        return self._sto(self.realizations.v2s0("MULT", self._vec(rv)), new_time)

    def _div_rv(self, rv):
        if rv.getTypePriority() > self.getTypePriority(): return rv.vid(self)
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic():
            return self._det(_f64(np.divide, self.value, rv.doubleValue()), new_time)
        if self.isDeterministic(): return self._sto(self._vec(rv).v1s1("VID_S", self.value), new_time)
        if rv.isDeterministic(): return self.div(rv.doubleValue())
        return self._sto(self.realizations.v2s0("DIV", self._vec(rv)), new_time)

    def _vid_rv(self, rv):
        if rv.getTypePriority() > self.getTypePriority(): return rv.div(self)          # twin:1116-1119
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic():
            return self._det(_f64(np.divide, rv.doubleValue(), self.value), new_time)
        if self.isDeterministic(): return self._sto(self._vec(rv).v1s1("DIV_S", self.value), new_time)
        if rv.isDeterministic():        # value as :1528 (scalar narrowed like every scalar operand), time as the twin (twin:1135-1140)
            return self._restamp(self.vid(rv.doubleValue()), new_time)
        return self._sto(self._vec(rv).v2s0("DIV", self.realizations), new_time)       # flipped arguments, :1531

    def _cap_rv(self, rv):
        if rv.getTypePriority() > self.getTypePriority(): return rv.cap(self)
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic(): return self._det(_jmin(self.value, rv.doubleValue()), new_time)
        if self.isDeterministic(): return self._sto(self._vec(rv).v1s1("CAP_S", self.value), new_time)
        if rv.isDeterministic(): return self._sto(self.realizations.v1s1("CAP_S", rv.doubleValue()), new_time)
        return self._sto(self.realizations.v2s0("CAP", self._vec(rv)), new_time)

    def _floor_rv(self, rv):
        if rv.getTypePriority() > self.getTypePriority(): return rv.floor(self)
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic(): return self._det(_jmax(self.value, rv.doubleValue()), new_time)
        if self.isDeterministic(): return self._sto(self._vec(rv).v1s1("FLOOR_S", self.value), new_time)
        if rv.isDeterministic(): return self._sto(self.realizations.v1s1("FLOOR_S", rv.doubleValue()), new_time)
        return self._sto(self.realizations.v2s0("FLOOR", self._vec(rv)), new_time)

    def accrue(self, rate, period_length):                            # :1583-1601
        if rate.getTypePriority() > self.getTypePriority(): return rate.mult(period_length).add(1.0).mult(self)
        new_time = max(self.time, rate.getFiltrationTime())
        if rate.isDeterministic(): return self.mult(1.0 + rate.doubleValue() * period_length)
        if self.isDeterministic():       # same rounding sequence as :1595-1596, but newTime kept as the twin does (twin:1214-1219)
            return self._sto(rate.mult(period_length).add(1.0).mult(self.value).realizations, new_time)
        return self._sto(self.realizations.v2s1("ACCRUE", self._vec(rate), period_length), new_time)

    def discount(self, rate, period_length):                          # :1604-1624
        if rate.getTypePriority() > self.getTypePriority(): return rate.mult(period_length).add(1.0).invert().mult(self)
        new_time = max(self.time, rate.getFiltrationTime())
        if rate.isDeterministic(): return self.div(1.0 + rate.doubleValue() * period_length)
        if self.isDeterministic():       # twin:1242-1247 (the GPU class short-cuts value == 0 to `this` and drops newTime, :1615-1619)
            return self._sto(rate.mult(period_length).add(1.0).vid(self.value).realizations, new_time)
        return self._sto(self.realizations.v2s1("DISCOUNT", self._vec(rate), period_length), new_time)

    def choose(self, value_if_trigger_non_negative, value_if_trigger_negative):        # twin:1264-1285
        a, b = value_if_trigger_non_negative, value_if_trigger_negative
        new_time = max(self.time, a.getFiltrationTime(), b.getFiltrationTime())
        if self.isDeterministic():
            return a if self.value >= 0 else b
        n = self.size()
        va = DeviceVector.filled(n, a.doubleValue()) if a.isDeterministic() else self._vec(a)
        vb = DeviceVector.filled(n, b.doubleValue()) if b.isDeterministic() else self._vec(b)
        return self._sto(self.realizations.v3s0("CHOOSE", va, vb), new_time)

    def addProduct(self, factor1, factor2):
        if not np.isscalar(factor2): return self._add_product_rv(factor1, factor2)
        # addProduct(RandomVariable, double)  :1638-1656
        if factor1.getTypePriority() > self.getTypePriority(): return factor1.mult(factor2).add(self)
        new_time = max(self.time, factor1.getFiltrationTime())
        if factor1.isDeterministic(): return self.add(factor1.doubleValue() * factor2)
        if not self.isDeterministic():
            return self._sto(self.realizations.v2s1("ADDPRODUCT_VS", self._vec(factor1), factor2), new_time)
        return self.add(factor1.mult(factor2))

    def _add_product_rv(self, factor1, factor2):                      # :1658-1683
        if factor1.getTypePriority() > self.getTypePriority() or factor2.getTypePriority() > self.getTypePriority():
            return factor1.mult(factor2).add(self)
        new_time = max(self.time, factor1.getFiltrationTime(), factor2.getFiltrationTime())
        if self.isDeterministic() and factor1.isDeterministic() and factor2.isDeterministic():
            return self._det(self.value + factor1.doubleValue() * factor2.doubleValue(), new_time)
        if factor1.isDeterministic() and factor2.isDeterministic():
            return self.add(factor1.doubleValue() * factor2.doubleValue())
        if factor2.isDeterministic(): return self.addProduct(factor1, factor2.doubleValue())
        if factor1.isDeterministic(): return self.addProduct(factor2, factor1.doubleValue())
        if not self.isDeterministic():
            return self._sto(self.realizations.v3s0("ADDPRODUCT", self._vec(factor1), self._vec(factor2)), new_time)
        return self.add(factor1.mult(factor2))

    def addSumProduct(self, factor1, factor2):                        # interface default, twin:1385-1392
        result = self
        for f1, f2 in zip(factor1, factor2):
            result = result.addProduct(f1, f2)
        return result

    def _restamp(self, rv, time):
        """`rv` with filtration time `time` (same value / same device vector)."""
        if not isinstance(rv, RandomVariableHip) or rv.time == time: return rv
        return RandomVariableHip(time, rv.value if rv.isDeterministic() else rv.realizations, rv.type_priority)

    def addRatio(self, numerator, denominator):                       # :1686-1689 composes add(div); the filtration time is the
        new_time = max(self.time, numerator.getFiltrationTime(), denominator.getFiltrationTime())   # twin's (twin:1395-1438)
        return self._restamp(self.add(numerator.div(denominator)), new_time)

    def subRatio(self, numerator, denominator):                       # :1692-1695
        new_time = max(self.time, numerator.getFiltrationTime(), denominator.getFiltrationTime())
        return self._restamp(self.sub(numerator.div(denominator)), new_time)

    def apply(self, *args):
        raise NotImplementedError("UnsupportedOperationException: apply(lambda) cannot run on the device (:1146-1169)")

    def __repr__(self):
        if self.isDeterministic(): return f"RandomVariableHip(time={self.time}, value={self.value})"
        return f"RandomVariableHip(time={self.time}, size={self.size()}, handle={self.realizations.handle})"


class RandomVariableHipFactory:
    """Mirror of RandomVariableCudaFactory (RandomVariableCudaFactory.java:27-34)."""

    def createRandomVariable(self, *args):
        if len(args) == 1:                                           # AbstractRandomVariableFactory: time = -inf
            return RandomVariableHip(-math.inf, args[0])
        time, value = args
        return RandomVariableHip(time, value)

    def createRandomVariableArray(self, values):
        return [self.createRandomVariable(v) for v in values]
