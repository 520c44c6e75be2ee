"""CPU-only: host-side logic of the mirror classes that never touches the device — deterministic
(constant) random variables are kept on the host as doubles (RandomVariableCuda.java:683-689) and every
method has a host fast path for them.  Compared with the twin restatement (oracle) value for value."""
import math

import pytest


CASES = {
    "chain": lambda y, z: y.mult(2.0).add(1.0).squared().sub(4.0).div(7.0),
    "cap_floor": lambda y, z: y.cap(0.3).floor(z).cap(z.add(0.1)),
    "bus_vid": lambda y, z: y.bus(z).vid(3.0).bus(1.0).vid(z),
    "pow_sqrt_exp_log": lambda y, z: y.pow(1.5).sqrt().exp().log().abs().invert(),
    "accrue_discount": lambda y, z: y.accrue(z, 0.5).discount(z, 0.25),
    "addProduct": lambda y, z: y.addProduct(z, 2.0).addProduct(z, y).addSumProduct([y, z], [z, z]),
    "choose": lambda y, z: y.sub(1.0).choose(y, z),
    "ratio": lambda y, z: y.addRatio(z, y).subRatio(y, z),
    "sincos_isnan": lambda y, z: y.sin().add(z.cos()).add(y.log().sub(10.0).sqrt().isNaN()),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_deterministic_fast_paths_match_twin(fm, oracle, name):
    f = CASES[name]
    g = f(fm.RandomVariableHip(1.0, 0.7275636800328681), fm.RandomVariableHip(2.0, 0.25))
    o = f(oracle.RandomVariableFromFloatArray(1.0, 0.7275636800328681), oracle.RandomVariableFromFloatArray(2.0, 0.25))
    assert g.isDeterministic() and o.isDeterministic()
    assert g.doubleValue() == o.doubleValue() or (math.isnan(g.doubleValue()) and math.isnan(o.doubleValue()))
    assert g.getFiltrationTime() == o.getFiltrationTime()


def test_deterministic_reductions(fm):
    y = fm.RandomVariableHipFactory().createRandomVariable(2.5)
    assert y.getFiltrationTime() == -math.inf and y.size() == 1 and y.getTypePriority() == 20
    assert y.getAverage() == 2.5 and y.getVariance() == 0.0 and y.getMin() == 2.5 and y.getMax() == 2.5
    assert y.getStandardDeviation() == 0.0 and y.getStandardError() == 0.0 and y.getSampleVariance() == 0.0
    assert y.getQuantile(0.3) == 2.5 and y.getQuantileExpectation(0.1, 0.9) == 2.5
    assert list(y.getRealizations()) == [2.5] and y.get(0) == 2.5 and y.doubleValue() == 2.5
    assert list(y.getHistogram([1.0, 2.0, 3.0])) == [1.0, 0.0, 0.0, 1.0]
    assert y.average().doubleValue() == 2.5 and y.cache() is y


def test_time_discretization_and_bm_metadata(fm):
    td = fm.TimeDiscretization(0.0, 10, 0.1)
    assert td.getNumberOfTimeSteps() == 10 and abs(td.getTimeStep(3) - 0.1) < 1e-15 and td.getTime(10) == 1.0
    bm = fm.BrownianMotionHip(td, 2, 1000, 1234)
    assert (bm.getNumberOfFactors(), bm.getNumberOfPaths(), bm.getSeed()) == (2, 1000, 1234)
    assert bm == fm.BrownianMotionHip(td, 2, 1000, 1234) and hash(bm) == hash(fm.BrownianMotionHip(td, 2, 1000, 1234))
    assert bm.getCloneWithModifiedSeed(5).getSeed() == 5
    assert bm.getCloneWithModifiedTimeDiscretization(fm.TimeDiscretization(0.0, 5, 0.2)).getTimeDiscretization().getNumberOfTimeSteps() == 5
    assert bm.getRandomVariableForConstant(3.0).doubleValue() == 3.0
