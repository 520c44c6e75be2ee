"""Pins the CPU oracle against every known-answer value the reference's own tests state for this path
 and against published vectors for the two restated generators.
The reference has no golden files (no src/test/resources); these values ARE its fixtures."""
import math

import numpy as np
import pytest

from oracle import (RandomVariableFloatFactory, java_random_doubles, java_random_next_int, philox4x32_10,
                    bm_increment, f_average, f_variance)

ERROR_TOLERANCE = 1e-7      # RandomVariableGPUTest.java:57 (errorTolerance)


@pytest.fixture
def factory():
    return RandomVariableFloatFactory()


def test_deterministic(factory):
    """RandomVariableGPUTest.java:69-86 — avg == 3.0, var == 0.0 exactly."""
    rv = factory.createRandomVariable(2.0)
    rv = rv.mult(2.0).add(1.0).squared().sub(4.0).div(7.0)
    assert rv.getAverage() == 3.0
    assert rv.getVariance() == 0.0


def test_stochastic(factory):
    """RandomVariableGPUTest.java:89-122."""
    rv = factory.createRandomVariable(0.0, np.array([-4.0, -2.0, 0.0, 2.0, 4.0]))
    rv = rv.add(4.0).div(2.0).mult(2.0).div(2.0)
    assert abs(rv.getAverage() - 2.0) <= 1e-7
    assert rv.getVariance() == 2.0
    rv2 = factory.createRandomVariable(3.0).mult(rv)
    assert rv2.getAverage() == 6.0
    assert rv2.getVariance() == 2.0 * 9.0


SIZES = [2, 2, 3, 4, 5, 7, 10, 13, 99, 100, 1000, 1024, 2047, 2048, 2049, 20000, 200000]


@pytest.mark.parametrize("size", SIZES)
def test_average_closed_form(factory, size):
    """RandomVariableGPUTest.java:125-153."""
    rv = factory.createRandomVariable(0.0, np.arange(size, dtype=np.float64))
    want = size * (size - 1.0) / 2.0 / size
    assert abs(rv.getAverage() - want) <= want * 1e-6
    rv = factory.createRandomVariable(0.0, (np.arange(size) % 2).astype(np.float64))
    want = (size / 2.0) / size if size % 2 == 0 else float(size // 2) / size
    assert abs(rv.getAverage() - want) <= size / 2.0 * 1e-7


def test_sqrt_pow_squared_stddev(factory):
    """RandomVariableGPUTest.java:156-188."""
    rv = factory.createRandomVariable(0.0, np.array([3.0, 1.0, 0.0, 2.0, 4.0, 1.0 / 3.0]))
    check = rv.sqrt().sub(rv.pow(0.5))
    assert abs(check.getAverage()) <= ERROR_TOLERANCE and abs(check.getVariance()) <= ERROR_TOLERANCE
    check = rv.squared().sub(rv.pow(2.0))
    assert abs(check.getAverage()) <= ERROR_TOLERANCE and abs(check.getVariance()) <= ERROR_TOLERANCE
    assert abs(math.sqrt(rv.getVariance()) - rv.getStandardDeviation()) <= ERROR_TOLERANCE


def test_java_random_published_values():
    """java.util.Random: widely published first outputs (JDK semantics are a stable spec)."""
    assert java_random_next_int(42) == -1170105035
    assert java_random_next_int(0) == -1155484576
    d = java_random_doubles(42, 2)
    assert d[0] == 0.7275636800328681
    x = java_random_doubles(31415, 100000)     # RandomVariableGPUTest.java:194-201
    assert ((x >= 0.0) & (x < 1.0)).all()
    assert abs(x.mean() - 0.5) < 5e-3 and abs(x.var() - 1.0 / 12.0) < 2e-3


def test_philox4x32_10_random123_kat():
    """Random123 kat_vectors, philox4x32 10 rounds."""
    def h(s): return [int(t, 16) for t in s.split()]
    assert list(philox4x32_10([0, 0, 0, 0], [0, 0])) == h("6627e8d5 e169c58d bc57ac4c 9b00dbd8")
    assert list(philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)) == h("408f276d 41c83b0e a20bc7c6 6d5451fd")
    assert list(philox4x32_10(h("243f6a88 85a308d3 13198a2e 03707344"), h("a4093822 299f31d0"))) == \
        h("d16cfe09 94fdcceb 5001e420 24126ea1")


def test_bm_increment_moments_reference_bounds():
    """BrownianMotionTest.java:66-127: N = 1e6, dt = 0.1, mean 0 ± 3·sqrt(dt)/sqrt(N), var dt ± 3·dt/sqrt(N)."""
    n, dt = 1_000_000, 0.1
    z = bm_increment(1234, 0, 0, n, math.sqrt(dt))
    mean, var = f_average(z), f_variance(z)
    assert abs(mean) < 3.0 * math.sqrt(dt) / math.sqrt(n)
    assert abs(var - dt) < 3.0 * dt / math.sqrt(n)
    # shard invariance: any window of the stream equals the same window of the whole
    w = bm_increment(1234, 0, 12345, 1001, math.sqrt(dt))
    assert (w.view(np.uint32) == z[12345:12345 + 1001].view(np.uint32)).all()
    # distribution shape: 4th moment of N(0,1) is 3, tails exist
    zs = z.astype(np.float64) / math.sqrt(dt)
    assert abs((zs ** 4).mean() - 3.0) < 0.05
    assert zs.max() > 4.0 and zs.min() < -4.0


def test_normal_generator_z_scores_over_many_streams():
    """Quality check of the restated generator: z-scores of sample mean and variance over 200 independent streams
    behave like N(0,1) draws (no stream beyond 4.5σ, mean square ≈ 1)."""
    n = 50_000
    zm, zv = [], []
    for stream in range(200):
        z = bm_increment(987654321, stream, 0, n, 1.0).astype(np.float64)
        zm.append(z.mean() * math.sqrt(n))
        zv.append((z.var() - 1.0) / math.sqrt(2.0 / n))
    zm, zv = np.array(zm), np.array(zv)
    assert np.abs(zm).max() < 4.5 and np.abs(zv).max() < 4.5
    assert 0.7 < (zm ** 2).mean() < 1.3 and 0.7 < (zv ** 2).mean() < 1.3
    assert abs(zm.mean()) < 0.3 and abs(zv.mean()) < 0.3


def test_bench_input_generator_is_java_util_random():
    """bench.py draws its inputs itself (the product side may not use the oracle): its vectorised java.util.Random must be the
    stream the oracle's sequential restatement gives, at the start and at a block offset (SURVEY.md §8d config 2 seeds)."""
    import importlib.util
    import os
    import numpy as np
    import oracle
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for seed in (31415, 27182, 16180):
        want = oracle.java_random_doubles(seed, 30000)
        assert np.array_equal(bench.java_random_doubles(seed, 10000), want[:10000])
        assert np.array_equal(bench.java_random_doubles(seed, 10000, skip=20000), want[20000:])
        assert np.array_equal(bench.java_random_doubles(seed, 7, skip=12345), want[12345:12352])
    assert abs(bench.java_random_doubles(31415, 100000).mean() - 0.5) < 0.005
