"""SURVEY.md §8f row f4 — the AAD wrapper (differentiable.py) on the CPU twin: every partial derivative against closed
forms, the adjoint sweep with shared sub-expressions, the expectation operator, type-priority interplay (README.md:50-52),
and a Monte-Carlo Black–Scholes delta / vega by pathwise adjoint differentiation against the analytic greeks."""
import math

import numpy as np
import pytest

from aad_cases import black_scholes, expressions, finance_ops_f64

N = 4001


@pytest.fixture(scope="module")
def xy(oracle):
    x = oracle.f_from_double(oracle.java_random_doubles(31415, N) * 0.5 + 0.5)       # [0.5, 1)
    y = oracle.f_from_double(oracle.java_random_doubles(27182, N) * 0.5 + 0.6)       # [0.6, 1.1)
    return x, y


def wrap(fm, oracle, x, y):
    f = fm.RandomVariableDifferentiableAADFactory(oracle.RandomVariableFloatFactory())
    return f, f.createRandomVariable(0.0, x), f.createRandomVariable(0.0, y)


@pytest.mark.parametrize("name", ["poly", "ratio", "transcendental", "trig", "kinks", "choose"])
def test_gradient_matches_closed_form(fm, oracle, xy, name):
    x, y = xy
    f, dfdx, dfdy = expressions()[name]
    factory, X, Y = wrap(fm, oracle, x, y)
    g = f(X, Y).getGradient()
    xd, yd = x.astype(np.float64), y.astype(np.float64)
    for leaf, want in ((X, dfdx(xd, yd)), (Y, dfdy(xd, yd))):
        got = g[leaf.getID()].getRealizations()
        got = np.broadcast_to(got, want.shape)
        assert np.allclose(got, want, rtol=2e-5, atol=2e-5), name      # fp32 arithmetic of the twin


def test_finance_ops_by_finite_differences(fm, oracle, xy):
    x, y = xy
    f = expressions()["finance_ops"][0]
    factory, X, Y = wrap(fm, oracle, x, y)
    g = f(X, Y).getGradient()
    xd, yd, h = x.astype(np.float64), y.astype(np.float64), 1e-6
    ddx = (finance_ops_f64(xd + h, yd) - finance_ops_f64(xd - h, yd)) / (2 * h)
    ddy = (finance_ops_f64(xd, yd + h) - finance_ops_f64(xd, yd - h)) / (2 * h)
    assert np.allclose(g[X.getID()].getRealizations(), ddx, rtol=5e-5, atol=5e-5)
    assert np.allclose(g[Y.getID()].getRealizations(), ddy, rtol=5e-5, atol=5e-5)


def test_expectation_operator(fm, oracle, xy):
    """z = E[x y] · x: dz/dx collects the direct term E[xy] and, through average(), E[x]·y (finmath's convention:
    the adjoint of average() is the average of the adjoint)."""
    x, y = xy
    factory, X, Y = wrap(fm, oracle, x, y)
    z = expressions()["expectation"][0](X, Y)
    g = z.getGradient()
    xd, yd = x.astype(np.float64), y.astype(np.float64)
    want_dx = np.mean(xd * yd) + np.mean(xd) * yd
    assert np.allclose(g[X.getID()].getRealizations(), want_dx, rtol=1e-5)
    assert np.allclose(g[Y.getID()].getRealizations(), np.mean(xd) * xd, rtol=1e-5)


def test_type_priority_and_mixed_operands(fm, oracle, xy):
    x, y = xy
    factory, X, Y = wrap(fm, oracle, x, y)
    plain = oracle.RandomVariableFromFloatArray(0.0, y)
    assert X.getTypePriority() > plain.getTypePriority()
    for z in (plain.add(X), plain.mult(X), plain.sub(X), plain.div(X), plain.addProduct(X, 2.0), plain.addProduct(X, plain)):
        assert isinstance(z, fm.RandomVariableDifferentiableAAD)                   # the plain class hands over (priority)
        assert X.getID() in z.getGradient()
    g = plain.div(X).getGradient()                                                 # y / x
    assert np.allclose(g[X.getID()].getRealizations(), -y.astype(np.float64) / x.astype(np.float64) ** 2, rtol=1e-5)
    # constants and non-differentiable values contribute nothing
    c = factory.createRandomVariableNonDifferentiable(2.0)
    assert list(X.mult(c).getGradient()) == [X.getID()]
    assert X.getGradient([X.getID(), Y.getID()])[Y.getID()].doubleValue() == 0.0


def test_shared_subexpressions_accumulate(fm, oracle, xy):
    x, y = xy
    factory, X, Y = wrap(fm, oracle, x, y)
    t = X.mult(Y)                               # used three times
    z = t.add(t).mult(t)                        # 2 t²  → dz/dx = 4 t y
    g = z.getGradient()
    xd, yd = x.astype(np.float64), y.astype(np.float64)
    assert np.allclose(g[X.getID()].getRealizations(), 4 * xd * yd * yd, rtol=1e-5)


def test_black_scholes_delta_and_vega(fm, oracle):
    """Pathwise adjoint greeks of MonteCarloBlackScholesModelTest's product (S0 = 1, r = 5 %, σ = 30 %, T = 2, K = 1.05):
    the model parameters are deterministic leaves, the Brownian increment a plain random variable."""
    S0, r, sigma, T, K, n = 1.0, 0.05, 0.30, 2.0, 1.05, 200000
    factory = fm.RandomVariableDifferentiableAADFactory(oracle.RandomVariableFloatFactory())
    W = oracle.RandomVariableFromFloatArray(T, oracle.bm_increment(31415, 0, 0, n, math.sqrt(T)))
    s0, vol = factory.createRandomVariable(S0), factory.createRandomVariable(sigma)
    drift = vol.squared().mult(-0.5 * T).add(r * T)
    ST = drift.add(vol.mult(W)).exp().mult(s0)
    value = ST.sub(K).floor(0.0).mult(math.exp(-r * T)).average()
    price, delta, vega = black_scholes(S0, r, sigma, T, K)
    g = value.getGradient()
    assert abs(value.getAverage() - price) < 0.005                         # MonteCarloBlackScholesModelTest.java:156
    assert abs(g[s0.getID()].getAverage() - delta) < 0.01
    assert abs(g[vol.getID()].getAverage() - vega) < 0.02
