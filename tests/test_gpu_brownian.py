"""BrownianMotionHip: bit-exact against the generator's CPU specification (oracle/philox_normal.c), the
reference's moment bounds (BrownianMotionTest.java:120-121, BrownianMotionMemoryTest.java:73-74), shard
invariance, filtration times and scaling (BrownianMotionCudaWithRandomVariableCuda.java:169-176)."""
import math

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


def test_bit_exact_vs_spec(gpu, oracle):
    td = gpu.TimeDiscretization(0.0, 3, 0.1)
    for n_paths in (1, 4, 1023, 10001):
        bm = gpu.BrownianMotionHip(td, 2, n_paths, 1234)
        want = oracle.bm_generate(1234, [td.getTimeStep(i) for i in range(3)], 2, n_paths)
        for t in range(3):
            for f in range(2):
                rv = bm.getBrownianIncrement(t, f)
                assert rv.getFiltrationTime() == td.getTime(t + 1)
                assert_bits_equal(rv.realizations.to_float32(), want[t][f], f"n={n_paths} t={t} f={f}")


def test_non_uniform_steps_and_seed(gpu, oracle):
    td = gpu.TimeDiscretization([0.0, 0.25, 0.3, 1.3])
    bm = gpu.BrownianMotionHip(td, 1, 4097, -77)
    want = oracle.bm_generate(-77, [td.getTimeStep(i) for i in range(3)], 1, 4097)
    for t in range(3):
        assert_bits_equal(bm.getBrownianIncrement(t, 0).realizations.to_float32(), want[t][0], f"t={t}")
    other = bm.getCloneWithModifiedSeed(78).getBrownianIncrement(0, 0).realizations.to_float32()
    assert (other != want[0][0]).mean() > 0.99
    assert bm == gpu.BrownianMotionHip(td, 1, 4097, -77) and bm != bm.getCloneWithModifiedSeed(78)


def test_shard_invariance(gpu):
    """Union of path shards == single-device stream, for aligned and unaligned shard boundaries."""
    td = gpu.TimeDiscretization(0.0, 2, 0.5)
    n = 10000
    whole = gpu.BrownianMotionHip(td, 2, n, 31415)
    for cuts in ([0, 2500, 5000, 7500, n], [0, 3, 4098, 9999, n]):
        for t in range(2):
            for f in range(2):
                parts = [gpu.BrownianMotionHip(td, 2, cuts[i + 1] - cuts[i], 31415, path_offset=cuts[i])
                         .getBrownianIncrement(t, f).realizations.to_float32() for i in range(len(cuts) - 1)]
                assert_bits_equal(np.concatenate(parts), whole.getBrownianIncrement(t, f).realizations.to_float32(), f"{cuts} t={t} f={f}")


def test_moments_reference_bounds(gpu):
    """BrownianMotionTest.java:66-127 with N = 1 000 000, dt = 0.1, seed 1234."""
    n, dt = 1_000_000, 0.1
    bm = gpu.BrownianMotionHip(gpu.TimeDiscretization(0.0, 10, dt), 1, n, 1234)
    # The reference asserts |mean| < 3·sqrt(dt)/sqrt(N) (a 3σ bound) and |var - dt| < 3·dt/sqrt(N), which is only a
    # 2.1σ bound (σ of the variance estimator is dt·sqrt(2/N)); over 10 steps the latter fails by chance 30 % of the
    # time for ANY correct generator.  Asserted here: the reference's bound on its own configuration (first increment),
    # and 4σ bounds on every step.
    rv0 = bm.getBrownianIncrement(0, 0)
    assert abs(rv0.getAverage()) < 3.0 * math.sqrt(dt) / math.sqrt(n)
    assert abs(rv0.getVariance() - dt) < 3.0 * dt / math.sqrt(n)
    for t in range(10):
        rv = bm.getBrownianIncrement(t, 0)
        d = bm.getTimeDiscretization().getTimeStep(t)
        assert abs(rv.getAverage()) < 4.0 * math.sqrt(d) / math.sqrt(n)
        assert abs(rv.getVariance() - d) < 4.0 * d * math.sqrt(2.0 / n)
    a, b = bm.getBrownianIncrement(0, 0), bm.getBrownianIncrement(1, 0)
    assert abs(a.mult(b).getAverage()) < 4.0 * dt / math.sqrt(n)        # independent across steps


def test_memory_soak(gpu):
    """BrownianMotionMemoryTest.java:41-80 (shortened): growing path counts, pool reuse, no leak."""
    dt = 0.1
    for n in range(100_000, 400_001, 50_000):
        bm = gpu.BrownianMotionHip(gpu.TimeDiscretization(0.0, 4, dt), 2, n, 1234)
        rv = bm.getBrownianIncrement(3, 1)
        assert abs(rv.getAverage()) < 4.0 * math.sqrt(dt) / math.sqrt(n)
        assert abs(rv.getVariance() - dt) < 5.0 * dt / math.sqrt(n)
        del bm, rv
        gpu.purge()
    s = gpu.pool_stats()
    assert s.bytes_cached == 0
