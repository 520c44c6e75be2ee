"""Object-level restatement of the reference's CPU twin (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/main/java/net/finmath/cuda/cpu/montecarlo/RandomVariableFromFloatArray.java
(cited as ``:line``) method by method: a random variable is EITHER a constant held as a double
(``realizations is None``) or a float32 vector; every binary method first checks the type priority,
then takes the maximum of the filtration times, then dispatches on which operands are deterministic.
Array arithmetic is done by the C oracle (oracle/rv_float.c) so each operation rounds to fp32 exactly
as the Java loop bodies do.
"""
from __future__ import annotations

import math

import numpy as np

from . import (f_v1s0, f_v1s1, f_v2s0, f_v2s1, f_v3s0, f_average, f_variance, f_min, f_max,
               f_average_weighted, f_variance_weighted, f_quantile, f_from_double)

TYPE_PRIORITY_DEFAULT = 1           # :47


def _jmin(a, b):                    # java.lang.Math.min(double,double)
    if a != a: return a
    if a == 0.0 and b == 0.0 and math.copysign(1.0, b) < 0: return b
    return a if a <= b else b


def _jmax(a, b):
    if a != a: return a
    if a == 0.0 and b == 0.0 and math.copysign(1.0, a) < 0: return b
    return a if a >= b else b


def _jpow(x, y):
    if y == 0.0: return 1.0
    if y != y: return y
    if math.isinf(y) and abs(x) == 1.0: return math.nan
    try:
        return math.pow(x, y)
    except (OverflowError, ValueError):
        return float(np.power(np.float64(x), np.float64(y)))


def _jdiv(a, b):
    return float(np.float64(a) / np.float64(b))


def _jexp(x):
    return float(np.exp(np.float64(x)))


def _jlog(x):
    with np.errstate(all="ignore"):
        return float(np.log(np.float64(x)))


def _jsqrt(x):
    with np.errstate(all="ignore"):
        return float(np.sqrt(np.float64(x)))


class RandomVariableFromFloatArray:
    """``time`` (filtration), then either ``value`` (double) or ``realizations`` (float32[n])."""

    def __init__(self, time, value, type_priority=TYPE_PRIORITY_DEFAULT):
        self.time = float(time)
        self.type_priority = type_priority
        if np.isscalar(value):
            self.realizations = None                    # :103-109
            self.value = float(value)
        else:
            arr = np.asarray(value)
            self.realizations = arr if arr.dtype == np.float32 else f_from_double(arr)   # :177-179, :217-223
            self.value = math.nan

    # ---- accessors ------------------------------------------------------------------------
    def getFiltrationTime(self): return self.time                      # :256
    def getTypePriority(self): return self.type_priority               # :261
    def isDeterministic(self): return self.realizations is None        # :605
    def size(self): return 1 if self.isDeterministic() else self.realizations.size   # :275
    def get(self, i): return self.value if self.isDeterministic() else float(self.realizations[i])  # :266

    def doubleValue(self):                                             # :639
        if self.isDeterministic(): return self.value
        raise NotImplementedError("The random variable is non-deterministic")

    def getRealizations(self):                                         # :629
        if self.isDeterministic(): return np.array([self.value], dtype=np.float64)
        return self.realizations.astype(np.float64)

    def cache(self): return self                                       # :610

    def _f(self, n=None):
        """float32 view of this operand as the Java loops see it: realizations[i] or (float)value."""
        if not self.isDeterministic(): return self.realizations
        return np.full(n, np.float32(self.value), dtype=np.float32)

    # ---- reductions -------------------------------------------------------------------------
    def getMin(self): return self.value if self.isDeterministic() else f_min(self.realizations)     # :284
    def getMax(self): return self.value if self.isDeterministic() else f_max(self.realizations)     # :299

    def getAverage(self, probabilities=None):
        if probabilities is None:                                      # :314-334
            if self.isDeterministic(): return self.value
            if self.size() == 0: return math.nan
            return f_average(self.realizations)
        if self.isDeterministic(): return self.value * probabilities.getAverage()     # :338-340
        if self.size() == 0: return math.nan
        return f_average_weighted(self.realizations, probabilities._f(self.size()))   # :345-356

    def getVariance(self, probabilities=None):
        if probabilities is None:                                      # :360-382
            if self.isDeterministic() or self.size() == 1: return 0.0
            if self.size() == 0: return math.nan
            return f_variance(self.realizations)
        if self.isDeterministic(): return 0.0                          # :385-407
        if self.size() == 0: return math.nan
        return f_variance_weighted(self.realizations, probabilities._f(self.size()))

    def getSampleVariance(self):                                       # :410-419
        if self.isDeterministic() or self.size() == 1: return 0.0
        if self.size() == 0: return math.nan
        return self.getVariance() * self.size() / (self.size() - 1)

    def getStandardDeviation(self, probabilities=None):                # :422-443
        if self.isDeterministic(): return 0.0
        if self.size() == 0: return math.nan
        return math.sqrt(self.getVariance(probabilities))

    def getStandardError(self, probabilities=None):                    # :446-470
        if self.isDeterministic(): return 0.0
        if self.size() == 0: return math.nan
        return self.getStandardDeviation(probabilities) / math.sqrt(self.size())

    def getQuantile(self, quantile):                                   # :473-487
        if self.isDeterministic(): return self.value
        if self.size() == 0: return math.nan
        return f_quantile(self.realizations, quantile)

    def average(self):                                                 # :856
        return RandomVariableFromFloatArray(-math.inf, self.getAverage())

    # ---- scalar operand / unary ---------------------------------------------------------------
    def _unary(self, det, op):
        if self.isDeterministic(): return RandomVariableFromFloatArray(self.time, det(self.value))
        return RandomVariableFromFloatArray(self.time, f_v1s0(op, self.realizations))

    def _scalar(self, det, op, s):
        if self.isDeterministic(): return RandomVariableFromFloatArray(self.time, det(self.value, s))
        return RandomVariableFromFloatArray(self.time, f_v1s1(op, self.realizations, s))

    def cap(self, x):
        if not np.isscalar(x): return self._cap_rv(x)
        return self._scalar(_jmin, "CAP_S", x)                         # :751
    def floor(self, x):
        if not np.isscalar(x): return self._floor_rv(x)
        return self._scalar(_jmax, "FLOOR_S", x)                       # :766
    def add(self, x):
        if not np.isscalar(x): return self._add_rv(x)
        return self._scalar(lambda a, b: a + b, "ADD_S", x)            # :781
    def sub(self, x):
        if not np.isscalar(x): return self._sub_rv(x)
        return self._scalar(lambda a, b: a - b, "SUB_S", x)            # :796
    def bus(self, x):
        if not np.isscalar(x): return self._bus_rv(x)
        return self._scalar(lambda a, b: -a + b, "BUS_S", x)           # RandomVariableCuda.java:1220
    def mult(self, x):
        if not np.isscalar(x): return self._mult_rv(x)
        return self._scalar(lambda a, b: a * b, "MULT_S", x)           # :811
    def div(self, x):
        if not np.isscalar(x): return self._div_rv(x)
        return self._scalar(_jdiv, "DIV_S", x)                         # :826
    def vid(self, x):
        if not np.isscalar(x): return self._vid_rv(x)
        return self._scalar(lambda a, b: _jdiv(b, a), "VID_S", x)      # RandomVariableCuda.java:1256
    def pow(self, exponent): return self._scalar(_jpow, "POW_S", exponent)   # :841

    def squared(self): return self._unary(lambda a: a * a, "SQUARED")  # :867
    def sqrt(self): return self._unary(_jsqrt, "SQRT")                 # :882
    def exp(self): return self._unary(_jexp, "EXP")                    # :897
    def log(self): return self._unary(_jlog, "LOG")                    # :912
    def sin(self): return self._unary(math.sin, "SIN")                 # :927
    def cos(self): return self._unary(math.cos, "COS")                 # :942
    def invert(self): return self._unary(lambda a: _jdiv(1.0, a), "INVERT")   # :1288
    def abs(self): return self._unary(abs, "ABS")                      # :1303
    def isNaN(self): return self._unary(lambda a: 1.0 if a != a else 0.0, "ISNAN")   # :1441

    # ---- binary: priority → newTime → dispatch ---------------------------------------------------
    def _binary(self, rv, swapped, det, op_vv, op_det_receiver, scalar_shortcut=None):
        if rv.getTypePriority() > self.getTypePriority():
            return swapped(rv)
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic():
            return RandomVariableFromFloatArray(new_time, det(self.value, rv.get(0)))
        if scalar_shortcut is not None and rv.isDeterministic():
            return scalar_shortcut(rv.get(0))                           # e.g. :1063-1064 "return this.mult(rv.get(0))"
        n = max(self.size(), rv.size())
        if self.isDeterministic():
            return RandomVariableFromFloatArray(new_time, f_v1s1(op_det_receiver, rv._f(n), self.value))
        return RandomVariableFromFloatArray(new_time, f_v2s0(op_vv, self.realizations, rv._f(n)))

    def _add_rv(self, rv):      # :961-987   det receiver: (float)v + b[i]
        return self._binary(rv, lambda r: r.add(self), lambda a, b: a + b, "ADD", "ADD_S")
    def _sub_rv(self, rv):      # :990-1017  det receiver: (float)v - b[i]  ==  -b[i] + (float)v
        return self._binary(rv, lambda r: r.bus(self), lambda a, b: a - b, "SUB", "BUS_S")
    def _bus_rv(self, rv):      # :1020-1047 b[i] - a[i]
        if rv.getTypePriority() > self.getTypePriority(): return rv.sub(self)
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic():
            return RandomVariableFromFloatArray(new_time, rv.get(0) - self.value)
        n = max(self.size(), rv.size())
        if self.isDeterministic():
            return RandomVariableFromFloatArray(new_time, f_v1s1("SUB_S", rv._f(n), self.value))
        return RandomVariableFromFloatArray(new_time, f_v2s0("SUB", rv._f(n), self.realizations))
    def _mult_rv(self, rv):     # :1050-1079
        return self._binary(rv, lambda r: r.mult(self), lambda a, b: a * b, "MULT", "MULT_S",
                            scalar_shortcut=lambda s: self.mult(s))
    def _div_rv(self, rv):      # :1082-1112 det receiver: (float)v / b[i]
        return self._binary(rv, lambda r: r.vid(self), _jdiv, "DIV", "VID_S",
                            scalar_shortcut=lambda s: self.div(s))
    def _vid_rv(self, rv):      # :1115-1142 b[i] / a[i]
        if rv.getTypePriority() > self.getTypePriority(): return rv.div(self)
        new_time = max(self.time, rv.getFiltrationTime())
        if self.isDeterministic() and rv.isDeterministic():
            return RandomVariableFromFloatArray(new_time, _jdiv(rv.get(0), self.value))
        n = max(self.size(), rv.size())
        if self.isDeterministic():
            return RandomVariableFromFloatArray(new_time, f_v1s1("DIV_S", rv._f(n), self.value))
        if rv.isDeterministic():
            # :1138 `(float)(randomVariable.get(i) / realizations[i])`: get(i) of a constant is its DOUBLE value — the quotient
            # is formed in double and narrowed once (differs from the fp32 quotient only for constants that are not fp32 values;
            # RandomVariableCuda narrows the constant first, :1528, and so does the HIP mirror)
            with np.errstate(all="ignore"):
                q = np.float64(rv.get(0)) / self.realizations.astype(np.float64)
            return RandomVariableFromFloatArray(new_time, q.astype(np.float32))
        return RandomVariableFromFloatArray(new_time, f_v2s0("DIV", rv._f(n), self.realizations))   # double quotient of two fp32 values: same bits
    def _cap_rv(self, rv):      # :1145-1171
        return self._binary(rv, lambda r: r.cap(self), _jmin, "CAP", "CAP_S")
    def _floor_rv(self, rv):    # :1174-1200
        return self._binary(rv, lambda r: r.floor(self), _jmax, "FLOOR", "FLOOR_S")

    def accrue(self, rate, period_length):                             # :1203-1228
        if rate.getTypePriority() > self.getTypePriority():
            return rate.mult(period_length).add(1.0).mult(self)
        new_time = max(self.time, rate.getFiltrationTime())
        if rate.isDeterministic():
            return self.mult(1.0 + rate.get(0) * period_length)
        n = max(self.size(), rate.size())
        if self.isDeterministic():      # (float)v * (1 + b[i]*(float)p)
            t = f_v1s1("ADD_S", f_v1s1("MULT_S", rate._f(n), period_length), 1.0)
            return RandomVariableFromFloatArray(new_time, f_v1s1("MULT_S", t, self.value))
        return RandomVariableFromFloatArray(new_time, f_v2s1("ACCRUE", self.realizations, rate._f(n), period_length))

    def discount(self, rate, period_length):                           # :1231-1256
        if rate.getTypePriority() > self.getTypePriority():
            return rate.mult(period_length).add(1.0).vid(self)
        new_time = max(self.time, rate.getFiltrationTime())
        if rate.isDeterministic():
            return self.div(1.0 + rate.doubleValue() * period_length)
        n = max(self.size(), rate.size())
        if self.isDeterministic():      # (float)v / (1.0f + b[i]*(float)p)
            t = f_v1s1("ADD_S", f_v1s1("MULT_S", rate._f(n), period_length), 1.0)
            return RandomVariableFromFloatArray(new_time, f_v1s1("VID_S", t, self.value))
        return RandomVariableFromFloatArray(new_time, f_v2s1("DISCOUNT", self.realizations, rate._f(n), period_length))

    def choose(self, if_non_negative, if_negative):                    # :1264-1285
        new_time = max(self.time, if_non_negative.getFiltrationTime(), if_negative.getFiltrationTime())
        if self.isDeterministic():
            return if_non_negative if self.value >= 0 else if_negative
        n = self.size()
        return RandomVariableFromFloatArray(
            new_time, f_v3s0("CHOOSE", self.realizations, if_non_negative._f(n), if_negative._f(n)))

    def addProduct(self, factor1, factor2):
        if not np.isscalar(factor2):
            return self._add_product_rv(factor1, factor2)
        # addProduct(RandomVariable, double)  :1318-1351
        if factor1.getTypePriority() > self.getTypePriority():
            return factor1.mult(factor2).add(self)
        new_time = max(self.time, factor1.getFiltrationTime())
        if factor1.isDeterministic():
            return self.add(factor1.get(0) * factor2)
        n = max(self.size(), factor1.size())
        if self.isDeterministic():      # (float)v + b[i]*(float)f2
            return RandomVariableFromFloatArray(
                new_time, f_v1s1("ADD_S", f_v1s1("MULT_S", factor1._f(n), factor2), self.value))
        return RandomVariableFromFloatArray(
            new_time, f_v2s1("ADDPRODUCT_VS", self.realizations, factor1._f(n), factor2))

    def _add_product_rv(self, factor1, factor2):                       # :1354-1382
        if factor1.getTypePriority() > self.getTypePriority() or factor2.getTypePriority() > self.getTypePriority():
            return factor1.mult(factor2).add(self)
        new_time = max(self.time, factor1.getFiltrationTime(), factor2.getFiltrationTime())
        if self.isDeterministic() and factor1.isDeterministic() and factor2.isDeterministic():
            return RandomVariableFromFloatArray(new_time, self.value + factor1.doubleValue() * factor2.doubleValue())
        if factor1.isDeterministic() and factor2.isDeterministic():
            return self.add(factor1.doubleValue() * factor2.doubleValue())
        if factor2.isDeterministic():
            return self.addProduct(factor1, factor2.doubleValue())
        if factor1.isDeterministic():
            return self.addProduct(factor2, factor1.doubleValue())
        if not self.isDeterministic():
            return RandomVariableFromFloatArray(
                new_time, f_v3s0("ADDPRODUCT", self.realizations, factor1.realizations, factor2.realizations))
        return self.add(factor1.mult(factor2))

    def addSumProduct(self, factor1, factor2):                         # :1385-1392
        result = self
        for f1, f2 in zip(factor1, factor2):
            result = result.addProduct(f1, f2)
        return result

    def _ratio(self, numerator, denominator, op, det):                 # :1395-1438
        if numerator.getTypePriority() > self.getTypePriority() or denominator.getTypePriority() > self.getTypePriority():
            q = numerator.div(denominator)
            return q.add(self) if op == "ADDRATIO" else q.mult(-1).add(self)
        new_time = max(self.time, numerator.getFiltrationTime(), denominator.getFiltrationTime())
        if self.isDeterministic() and numerator.isDeterministic() and denominator.isDeterministic():
            return RandomVariableFromFloatArray(new_time, det(self.value, _jdiv(numerator.get(0), denominator.get(0))))
        n = max(self.size(), numerator.size(), denominator.size())
        return RandomVariableFromFloatArray(new_time, f_v3s0(op, self._f(n), numerator._f(n), denominator._f(n)))

    def addRatio(self, numerator, denominator):
        return self._ratio(numerator, denominator, "ADDRATIO", lambda a, b: a + b)
    def subRatio(self, numerator, denominator):
        return self._ratio(numerator, denominator, "SUBRATIO", lambda a, b: a - b)


class RandomVariableFloatFactory:
    """RandomVariableFloatFactory.java:24-35."""

    def createRandomVariable(self, *args):
        if len(args) == 1:
            return RandomVariableFromFloatArray(-math.inf, args[0])     # AbstractRandomVariableFactory default time
        time, value = args
        return RandomVariableFromFloatArray(time, value)
