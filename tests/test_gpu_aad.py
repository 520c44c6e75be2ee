"""SURVEY.md §8f row f4 on the device: RandomVariableDifferentiableAADFactory(RandomVariableHipFactory()) — values AND
adjoints are computed by the HIP engine.  The same operations on the CPU twin give the same bits (arithmetic ops) or
≤ 1 fp32 ulp (fp64-evaluated exp/log/pow/sin/cos feeding further arithmetic: small relative tolerance)."""
import math

import numpy as np
import pytest

from aad_cases import black_scholes, expressions
from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu
N = 50021


@pytest.fixture(scope="module")
def xy(oracle):
    x = oracle.f_from_double(oracle.java_random_doubles(31415, N) * 0.5 + 0.5)
    y = oracle.f_from_double(oracle.java_random_doubles(27182, N) * 0.5 + 0.6)
    return x, y


def gradients(factory, f, x, y):
    X, Y = factory.createRandomVariable(0.0, x), factory.createRandomVariable(0.0, y)
    z = f(X, Y)
    g = z.getGradient()
    return z, g[X.getID()], g[Y.getID()]


@pytest.mark.parametrize("fusion", [False, True])
@pytest.mark.parametrize("name", ["poly", "ratio", "kinks", "choose", "finance_ops", "expectation", "transcendental", "trig"])
def test_gradient_on_device_equals_cpu_twin(gpu, oracle, xy, name, fusion):
    x, y = xy
    f = expressions()[name][0]
    cpu = gradients(gpu.RandomVariableDifferentiableAADFactory(oracle.RandomVariableFloatFactory()), f, x, y)
    prev = gpu.set_fusion(fusion)
    try:
        dev = gradients(gpu.RandomVariableDifferentiableAADFactory(gpu.RandomVariableHipFactory()), f, x, y)
        got = [np.asarray(v.getRealizations()) for v in dev]
    finally:
        gpu.set_fusion(prev)
    assert dev[0].getTypePriority() > cpu[0].getTypePriority() > 20          # AAD(GPU) > AAD(CPU) > GPU (README.md:52)
    for a, b in zip(got, cpu):
        want = np.asarray(b.getRealizations())
        if name in ("transcendental", "trig"):
            assert np.allclose(a, want, rtol=3e-6, atol=1e-7), name
        elif name == "expectation":                                            # fp64 sums differ in the last bits → 1 fp32 ulp
            assert np.allclose(a, want, rtol=2e-7), name
        else:
            assert_bits_equal(a.astype(np.float32), want.astype(np.float32), name)


def test_forward_and_adjoint_sweep_are_fused_launches(gpu, oracle, xy):
    x, y = xy
    f = expressions()["finance_ops"][0]
    factory = gpu.RandomVariableDifferentiableAADFactory(gpu.RandomVariableHipFactory())
    prev = gpu.set_fusion(True)
    try:
        X, Y = factory.createRandomVariable(0.0, x), factory.createRandomVariable(0.0, y)
        gpu.flush()
        before = gpu.pool_stats().n_kernel_launches
        z = f(X, Y)
        g = z.getGradient()
        total = z.getAverage() + g[X.getID()].getAverage() + g[Y.getID()].getAverage()
        launches = gpu.pool_stats().n_kernel_launches - before
    finally:
        gpu.set_fusion(prev)
    assert math.isfinite(total)
    # 6 forward methods + ≈ 60 adjoint methods: a handful of fused launches + 3 reductions (2 launches each)
    assert launches <= 16, launches


def test_black_scholes_greeks_on_device(gpu, oracle):
    """1 M paths, BrownianMotionHip increments, pathwise adjoint delta / vega against the closed form."""
    S0, r, sigma, T, K, n = 1.0, 0.05, 0.30, 2.0, 1.05, 1_000_000
    factory = gpu.RandomVariableDifferentiableAADFactory(gpu.RandomVariableHipFactory())
    bm = gpu.BrownianMotionHip(gpu.TimeDiscretization(0.0, 1, T), 1, n, 31415)
    W = bm.getBrownianIncrement(0, 0)
    prev = gpu.set_fusion(True)
    try:
        s0, vol = factory.createRandomVariable(S0), factory.createRandomVariable(sigma)
        drift = vol.squared().mult(-0.5 * T).add(r * T)
        ST = drift.add(vol.mult(W)).exp().mult(s0)
        value = ST.sub(K).floor(0.0).mult(math.exp(-r * T)).average()
        g = value.getGradient()
        got = value.getAverage(), g[s0.getID()].getAverage(), g[vol.getID()].getAverage()
    finally:
        gpu.set_fusion(prev)
    price, delta, vega = black_scholes(S0, r, sigma, T, K)
    assert abs(got[0] - price) < 0.005 and abs(got[1] - delta) < 0.005 and abs(got[2] - vega) < 0.01
