"""Object-level DOUBLE-precision stand-in for finmath-lib's RandomVariableFromDoubleArray / RandomVariableFromArrayFactory
(TEST INFRASTRUCTURE ONLY — the comparison class BASELINE.json's north star names).

The real class lives in net.finmath:finmath-lib:5.1.3 (pom.xml:29 of the reference), which is NOT vendored under
/root/reference: bit-level parity with it is "parity unpinned".  What is restated is the contract of the
net.finmath.stochastic.RandomVariable interface as the in-tree CPU twin implements it
(/root/reference/src/main/java/net/finmath/cuda/cpu/montecarlo/RandomVariableFromFloatArray.java, cited as ``:line``)
with every operation carried out in double and no narrowing anywhere: a random variable is either a constant or a
float64 vector, binary methods take the maximum of the filtration times, Math.min/max semantics for cap/floor, Kahan
average (:314-334), two-pass variance (:360-382).  Array arithmetic goes through the C oracle (oracle/rv_double.c).
The HIP path is compared with this class only within a stated fp32 tolerance (tests/test_gpu_double_class.py).
"""
from __future__ import annotations

import math

import numpy as np

from . import d_apply, d_average, d_variance, d_min, d_max


class RandomVariableFromDoubleArray:
    def __init__(self, time, value):
        self.time = float(time)
        if np.isscalar(value):
            self.realizations, self.value = None, float(value)
        else:
            self.realizations, self.value = np.ascontiguousarray(value, dtype=np.float64), math.nan

    # ---- accessors
    def getFiltrationTime(self): return self.time
    def getTypePriority(self): return 0
    def isDeterministic(self): return self.realizations is None
    def size(self): return 1 if self.isDeterministic() else self.realizations.size
    def get(self, i): return self.value if self.isDeterministic() else float(self.realizations[i])
    def doubleValue(self):
        if self.isDeterministic(): return self.value
        raise NotImplementedError("The random variable is non-deterministic")
    def getRealizations(self):
        return np.array([self.value]) if self.isDeterministic() else self.realizations.copy()
    def cache(self): return self

    def _d(self, n):
        return self.realizations if not self.isDeterministic() else np.full(n, self.value, dtype=np.float64)

    # ---- reductions (:284-487)
    def getMin(self): return self.value if self.isDeterministic() else d_min(self.realizations)
    def getMax(self): return self.value if self.isDeterministic() else d_max(self.realizations)
    def getAverage(self):
        if self.isDeterministic(): return self.value
        return d_average(self.realizations) if self.size() else math.nan
    def getVariance(self):
        if self.isDeterministic() or self.size() == 1: return 0.0
        return d_variance(self.realizations) if self.size() else math.nan
    def getSampleVariance(self):
        if self.isDeterministic() or self.size() == 1: return 0.0
        return self.getVariance() * self.size() / (self.size() - 1)
    def getStandardDeviation(self): return 0.0 if self.isDeterministic() else math.sqrt(self.getVariance())
    def getStandardError(self): return 0.0 if self.isDeterministic() else self.getStandardDeviation() / math.sqrt(self.size())
    def average(self): return RandomVariableFromDoubleArray(-math.inf, self.getAverage())

    # ---- element-wise: everything through d_apply on broadcast operands
    def _n(self, *others):
        return max([self.size()] + [o.size() for o in others if isinstance(o, RandomVariableFromDoubleArray)])

    def _apply(self, op, operands, scalar=None):
        """operands: RandomVariable operands in the opcode's (a, b, c) order."""
        time = max(o.time for o in operands)
        if all(o.isDeterministic() for o in operands):
            args = [np.array([o.value]) for o in operands] + ([scalar] if scalar is not None else [])
            with np.errstate(all="ignore"):
                return RandomVariableFromDoubleArray(time, float(d_apply(op, *args)[0]))
        n = max(o.size() for o in operands)
        args = [o._d(n) for o in operands] + ([scalar] if scalar is not None else [])
        with np.errstate(all="ignore"):
            return RandomVariableFromDoubleArray(time, d_apply(op, *args))

    def _rv_or_scalar(self, x, op_s, op_v, swap=False):
        if np.isscalar(x): return self._apply(op_s, [self], float(x))
        return self._apply(op_v, [x, self] if swap else [self, x])

    def cap(self, x): return self._rv_or_scalar(x, "CAP_S", "CAP")
    def floor(self, x): return self._rv_or_scalar(x, "FLOOR_S", "FLOOR")
    def add(self, x): return self._rv_or_scalar(x, "ADD_S", "ADD")
    def sub(self, x): return self._rv_or_scalar(x, "SUB_S", "SUB")
    def bus(self, x): return self._rv_or_scalar(x, "BUS_S", "SUB", swap=True)
    def mult(self, x): return self._rv_or_scalar(x, "MULT_S", "MULT")
    def div(self, x): return self._rv_or_scalar(x, "DIV_S", "DIV")
    def vid(self, x): return self._rv_or_scalar(x, "VID_S", "DIV", swap=True)
    def pow(self, e): return self._apply("POW_S", [self], float(e))
    def squared(self): return self._apply("SQUARED", [self])
    def sqrt(self): return self._apply("SQRT", [self])
    def exp(self): return self._apply("EXP", [self])
    def log(self): return self._apply("LOG", [self])
    def sin(self): return self._apply("SIN", [self])
    def cos(self): return self._apply("COS", [self])
    def invert(self): return self._apply("INVERT", [self])
    def abs(self): return self._apply("ABS", [self])
    def isNaN(self): return self._apply("ISNAN", [self])
    def accrue(self, rate, period): return self._apply("ACCRUE", [self, rate], float(period))           # :1203
    def discount(self, rate, period): return self._apply("DISCOUNT", [self, rate], float(period))       # :1231
    def choose(self, a, b):                                                                            # :1264-1285
        if self.isDeterministic(): return a if self.value >= 0 else b
        return self._apply("CHOOSE", [self, a, b])
    def addProduct(self, f1, f2):                                                                      # :1318-1382
        if np.isscalar(f2): return self._apply("ADDPRODUCT_VS", [self, f1], float(f2))
        return self._apply("ADDPRODUCT", [self, f1, f2])
    def addSumProduct(self, f1, f2):                                                                   # :1385-1392
        r = self
        for a, b in zip(f1, f2): r = r.addProduct(a, b)
        return r
    def addRatio(self, num, den): return self._apply("ADDRATIO", [self, num, den])                      # :1395
    def subRatio(self, num, den): return self._apply("SUBRATIO", [self, num, den])                      # :1418


class RandomVariableFromArrayFactory:
    """Stand-in for net.finmath.montecarlo.RandomVariableFromArrayFactory (finmath-lib 5.1.3, not vendored)."""

    def createRandomVariable(self, *args):
        if len(args) == 1: return RandomVariableFromDoubleArray(-math.inf, args[0])
        return RandomVariableFromDoubleArray(*args)
