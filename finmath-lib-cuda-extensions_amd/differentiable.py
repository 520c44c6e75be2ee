"""Adjoint algorithmic differentiation (AAD) on top of ANY RandomVariable implementation — SURVEY.md §8f row f4.

The reference's README (README.md:50-52, :119): "The RandomVariableCudaFactory can be combined with algorithmic
differentiation AAD wrappers, for example RandomVariableDifferentiableAAD, to allow algorithmic differentiation together
with calculations performed on the GPU.  For the type priority: objects allowing for algorithmic differentiation (AAD)
have higher priority, AAD on GPU has higher priority than AAD on CPU."  The wrapper classes themselves live in
finmath-lib (net.finmath.montecarlo.automaticdifferentiation.backward, not vendored): this module restates their
published design — an operator tree recorded while the wrapped values are computed by the inner factory, and a reverse
sweep (`getGradient`) that is itself written in RandomVariable operations, so that EVERY derivative is computed by the
inner implementation: with `RandomVariableHipFactory` inside, values and adjoints live in HBM and run through the same
fused launches as the valuation.  Parity is pinned only by mathematics (finite differences, closed forms) and by
bit-equality between the GPU engine and the CPU twin fed the same operations (tests/test_aad_cpu.py, tests/test_gpu_aad.py).

Conventions (as in finmath-lib):
  * `getGradient()` returns {id of independent: RandomVariable}; ids grow with creation, so a descending-id sweep is a
    reverse topological order.
  * the expectation operator `average()` propagates `derivative.average()`; the sensitivity of a Monte-Carlo value
    V = E[f] with respect to a deterministic parameter θ is `gradient[θ.getID()].getAverage()`.
  * non-differentiable ops use the almost-everywhere derivative: cap/floor/abs/choose via indicator functions.
"""
from __future__ import annotations

import itertools
import math

import numpy as np

AAD_TYPE_PRIORITY_OFFSET = 1000        # AAD above every plain type; AAD(GPU) = 1020 above AAD(CPU twin) = 1001 (README.md:52)

_ids = itertools.count(1)


class _Node:
    """One vertex of the operator tree: which operation produced a value, from which argument vertices (None for
    constants), together with the argument VALUES the partial derivatives need."""
    __slots__ = ("id", "op", "args", "arg_values", "scalar")

    def __init__(self, op, args, arg_values, scalar):
        self.id = next(_ids)
        self.op = op
        self.args = args                # tuple of _Node | None
        self.arg_values = arg_values    # tuple of inner RandomVariables (or None where not needed)
        self.scalar = scalar


class RandomVariableDifferentiableAAD:
    """A RandomVariable that records how it was computed.  `values` is a random variable of the inner factory."""

    __slots__ = ("values", "node", "factory")

    def __init__(self, values, node=None, factory=None):
        self.values = values
        self.node = node if node is not None else _Node("LEAF", (), (), 0.0)
        self.factory = factory

    # ---- identity / accessors: delegate to the wrapped value
    def getID(self): return self.node.id
    def getValues(self): return self.values
    def getFiltrationTime(self): return self.values.getFiltrationTime()
    def getTypePriority(self): return AAD_TYPE_PRIORITY_OFFSET + self.values.getTypePriority()
    def isDeterministic(self): return self.values.isDeterministic()
    def size(self): return self.values.size()
    def cache(self): return self
    def get(self, i): return self.values.get(i)
    def doubleValue(self): return self.values.doubleValue()
    def getRealizations(self): return self.values.getRealizations()
    def getMin(self): return self.values.getMin()
    def getMax(self): return self.values.getMax()
    def getAverage(self, probabilities=None): return self.values.getAverage(*( [_val(probabilities)] if probabilities is not None else [] ))
    def getVariance(self, probabilities=None): return self.values.getVariance(*( [_val(probabilities)] if probabilities is not None else [] ))
    def getSampleVariance(self): return self.values.getSampleVariance()
    def getStandardDeviation(self, probabilities=None): return self.values.getStandardDeviation(*( [_val(probabilities)] if probabilities is not None else [] ))
    def getStandardError(self, probabilities=None): return self.values.getStandardError(*( [_val(probabilities)] if probabilities is not None else [] ))
    def getQuantile(self, q): return self.values.getQuantile(q)
    def getQuantileExpectation(self, a, b): return self.values.getQuantileExpectation(a, b)
    def getHistogram(self, *a): return self.values.getHistogram(*a)

    # ---- recording
    def _new(self, op, values, operands, scalar=0.0):
        nodes = tuple(o.node if isinstance(o, RandomVariableDifferentiableAAD) else None for o in operands)
        vals = tuple(_val(o) for o in operands)
        return RandomVariableDifferentiableAAD(values, _Node(op, nodes, vals, float(scalar)), self.factory)

    def _unary(self, op):
        return self._new(op, getattr(self.values, _METHOD[op])(), (self,))

    def _scalar(self, op, s):
        return self._new(op, getattr(self.values, _METHOD[op])(s), (self,), s)

    def _binary(self, op, other):
        return self._new(op, getattr(self.values, _METHOD[op])(_val(other)), (self, other))

    # ---- the RandomVariable method set (RandomVariableCuda.java:1172-1695)
    def squared(self): return self._unary("SQUARED")
    def sqrt(self): return self._unary("SQRT")
    def exp(self): return self._unary("EXP")
    def log(self): return self._unary("LOG")
    def invert(self): return self._unary("INVERT")
    def abs(self): return self._unary("ABS")
    def sin(self): return self._unary("SIN")
    def cos(self): return self._unary("COS")
    def isNaN(self): return RandomVariableDifferentiableAAD(self.values.isNaN(), None, self.factory)      # piecewise constant

    def add(self, x): return self._scalar("ADD_S", x) if np.isscalar(x) else self._binary("ADD", x)
    def sub(self, x): return self._scalar("SUB_S", x) if np.isscalar(x) else self._binary("SUB", x)
    def bus(self, x): return self._scalar("BUS_S", x) if np.isscalar(x) else self._binary("BUS", x)
    def mult(self, x): return self._scalar("MULT_S", x) if np.isscalar(x) else self._binary("MULT", x)
    def div(self, x): return self._scalar("DIV_S", x) if np.isscalar(x) else self._binary("DIV", x)
    def vid(self, x): return self._scalar("VID_S", x) if np.isscalar(x) else self._binary("VID", x)
    def cap(self, x): return self._scalar("CAP_S", x) if np.isscalar(x) else self._binary("CAP", x)
    def floor(self, x): return self._scalar("FLOOR_S", x) if np.isscalar(x) else self._binary("FLOOR", x)
    def pow(self, e): return self._scalar("POW_S", e)

    def accrue(self, rate, period_length):
        return self._new("ACCRUE", self.values.accrue(_val(rate), period_length), (self, rate), period_length)

    def discount(self, rate, period_length):
        return self._new("DISCOUNT", self.values.discount(_val(rate), period_length), (self, rate), period_length)

    def choose(self, value_if_non_negative, value_if_negative):
        return self._new("CHOOSE", self.values.choose(_val(value_if_non_negative), _val(value_if_negative)),
                         (self, value_if_non_negative, value_if_negative))

    def addProduct(self, factor1, factor2):
        if np.isscalar(factor2):
            return self._new("ADDPRODUCT_VS", self.values.addProduct(_val(factor1), factor2), (self, factor1), factor2)
        return self._new("ADDPRODUCT", self.values.addProduct(_val(factor1), _val(factor2)), (self, factor1, factor2))

    def addSumProduct(self, factor1, factor2):
        result = self
        for a, b in zip(factor1, factor2):
            result = result.addProduct(a, b)
        return result

    def addRatio(self, numerator, denominator):
        return self._new("ADDRATIO", self.values.addRatio(_val(numerator), _val(denominator)), (self, numerator, denominator))

    def subRatio(self, numerator, denominator):
        return self._new("SUBRATIO", self.values.subRatio(_val(numerator), _val(denominator)), (self, numerator, denominator))

    def average(self):
        return self._new("AVERAGE", self.values.average(), (self,))

    def apply(self, *a, **k):
        raise NotImplementedError("UnsupportedOperationException: apply(lambda) has no derivative")

    # ---- the adjoint sweep
    def getGradient(self, independent_ids=None):
        """{id: d(this)/d(independent)} for every leaf below this value (or the requested ids).  Every operation of the
        sweep is a RandomVariable method of the inner implementation."""
        one = self._constant(1.0)
        adjoint = {self.node.id: one}
        # collect the sub-tree
        by_id, stack = {}, [self.node]
        while stack:
            nd = stack.pop()
            if nd.id in by_id:
                continue
            by_id[nd.id] = nd
            stack.extend(a for a in nd.args if a is not None)
        gradient = {}
        for nid in sorted(by_id, reverse=True):            # reverse topological order
            nd = by_id[nid]
            d = adjoint.pop(nid, None)
            if d is None:
                continue
            if nd.op == "LEAF":
                gradient[nid] = d
                continue
            for k, arg in enumerate(nd.args):
                if arg is None:
                    continue
                contribution = _partial_times(nd, k, d, self._constant)
                if contribution is None:
                    continue
                prev = adjoint.get(arg.id)
                adjoint[arg.id] = contribution if prev is None else prev.add(contribution)
        if independent_ids is not None:
            zero = self._constant(0.0)
            return {i: gradient.get(i, zero) for i in independent_ids}
        return gradient

    def _constant(self, v):
        if self.factory is not None:
            return self.factory.inner.createRandomVariable(v)
        # a constant of the inner type: 0·x + v keeps the implementation (and stays deterministic where x is)
        return self.values.average().mult(0.0).add(v) if not self.values.isDeterministic() else self.values.mult(0.0).add(v)

    def __repr__(self):
        return f"RandomVariableDifferentiableAAD(id={self.node.id}, op={self.node.op}, values={self.values!r})"


def _val(x):
    return x.values if isinstance(x, RandomVariableDifferentiableAAD) else x


_METHOD = dict(SQUARED="squared", SQRT="sqrt", EXP="exp", LOG="log", INVERT="invert", ABS="abs", SIN="sin", COS="cos",
               ADD_S="add", SUB_S="sub", BUS_S="bus", MULT_S="mult", DIV_S="div", VID_S="vid", CAP_S="cap", FLOOR_S="floor", POW_S="pow",
               ADD="add", SUB="sub", BUS="bus", MULT="mult", DIV="div", VID="vid", CAP="cap", FLOOR="floor")


def _partial_times(nd, k, d, const):
    """d · ∂(node)/∂(argument k), written in RandomVariable operations of the inner implementation."""
    op, s = nd.op, nd.scalar
    X = nd.arg_values[0]
    Y = nd.arg_values[1] if len(nd.arg_values) > 1 else None
    Z = nd.arg_values[2] if len(nd.arg_values) > 2 else None
    one, zero = const(1.0), const(0.0)
    # unary
    if op == "SQUARED": return d.mult(X.mult(2.0))
    if op == "SQRT": return d.div(X.sqrt().mult(2.0))
    if op == "EXP": return d.mult(X.exp())
    if op == "LOG": return d.div(X)
    if op == "INVERT": return d.div(X.squared()).mult(-1.0)
    if op == "ABS": return d.mult(X.choose(one, const(-1.0)))
    if op == "SIN": return d.mult(X.cos())
    if op == "COS": return d.mult(X.sin()).mult(-1.0)
    # scalar operand
    if op in ("ADD_S", "SUB_S"): return d
    if op == "BUS_S": return d.mult(-1.0)
    if op == "MULT_S": return d.mult(s)
    if op == "DIV_S": return d.div(s)
    if op == "VID_S": return d.div(X.squared()).mult(-s)
    if op == "POW_S": return d.mult(X.pow(s - 1.0).mult(s))
    if op == "CAP_S": return d.mult(X.sub(s).choose(zero, one))          # 1 where X < s
    if op == "FLOOR_S": return d.mult(X.sub(s).choose(one, zero))        # 1 where X >= s
    # two vector operands
    if op == "ADD": return d
    if op == "SUB": return d if k == 0 else d.mult(-1.0)
    if op == "BUS": return d.mult(-1.0) if k == 0 else d
    if op == "MULT": return d.mult(Y if k == 0 else X)
    if op == "DIV": return d.div(Y) if k == 0 else d.mult(X).div(Y.squared()).mult(-1.0)
    if op == "VID": return d.mult(Y).div(X.squared()).mult(-1.0) if k == 0 else d.div(X)
    if op == "CAP":
        ind = X.sub(Y).choose(zero, one)                                    # 1 where X < Y
        return d.mult(ind) if k == 0 else d.mult(ind.bus(1.0))
    if op == "FLOOR":
        ind = X.sub(Y).choose(one, zero)                                    # 1 where X >= Y
        return d.mult(ind) if k == 0 else d.mult(ind.bus(1.0))
    if op == "ACCRUE":                                                      # X (1 + Y s)
        return d.mult(Y.mult(s).add(1.0)) if k == 0 else d.mult(X.mult(s))
    if op == "DISCOUNT":                                                    # X / (1 + Y s)
        den = Y.mult(s).add(1.0)
        return d.div(den) if k == 0 else d.mult(X.mult(s)).div(den.squared()).mult(-1.0)
    if op == "ADDPRODUCT_VS": return d if k == 0 else d.mult(s)             # X + Y s
    if op == "ADDPRODUCT": return d if k == 0 else d.mult(Z if k == 1 else Y)      # X + Y Z
    if op == "ADDRATIO":                                                    # X + Y / Z
        if k == 0: return d
        return d.div(Z) if k == 1 else d.mult(Y).div(Z.squared()).mult(-1.0)
    if op == "SUBRATIO":                                                    # X - Y / Z
        if k == 0: return d
        return d.div(Z).mult(-1.0) if k == 1 else d.mult(Y).div(Z.squared())
    if op == "CHOOSE":                                                      # X >= 0 ? Y : Z
        if k == 0: return None
        return d.mult(X.choose(one, zero)) if k == 1 else d.mult(X.choose(zero, one))
    if op == "AVERAGE": return d.average()
    raise NotImplementedError(op)


class RandomVariableDifferentiableAADFactory:
    """net.finmath.montecarlo.automaticdifferentiation.backward.RandomVariableDifferentiableAADFactory: endows the random
    variables of any inner factory (README.md:119) with adjoint differentiation."""

    def __init__(self, inner_factory):
        self.inner = inner_factory

    def createRandomVariable(self, *args):
        return RandomVariableDifferentiableAAD(self.inner.createRandomVariable(*args), None, self)

    def createRandomVariableNonDifferentiable(self, *args):
        return self.inner.createRandomVariable(*args)

    def createRandomVariableArray(self, values):
        return [self.createRandomVariable(v) for v in values]

    def wrap(self, random_variable):
        """An independent (leaf) differentiable variable around an existing value, e.g. a Brownian-driven state."""
        return RandomVariableDifferentiableAAD(random_variable, None, self)
