"""CPU-only: the host-side Mersenne-Twister Brownian motion (csrc/mersenne.cpp) against published
known answers — MT19937 reference outputs (numpy's MT19937 with the same init_by_array seeding), AS 241 against scipy's
normal quantile — and the statistical bounds the reference asserts for Brownian increments (BrownianMotionTest.java:120-121)."""
import math

import numpy as np


def test_inverse_normal_cdf_as241(fm):
    from scipy.stats import norm
    inv = fm.lib().fmhip_inverse_normal_cdf
    ps = np.concatenate([np.linspace(1e-12, 1 - 1e-12, 20001), 10.0 ** -np.arange(1, 300, 7.0), 1 - 10.0 ** -np.arange(1, 16.0)])
    got = np.array([inv(float(p)) for p in ps])
    want = norm.ppf(ps)
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 5e-15
    assert inv(0.5) == 0.0 and inv(0.0) == -math.inf and inv(1.0) == math.inf and math.isnan(inv(1.5))
    assert abs(inv(0.975) - 1.959963984540054) < 1e-15


def test_mt19937_stream_matches_reference_generator(fm):
    """uniform = ((next32()>>6) << 26 | (next32()>>6)) * 2^-52 from MT19937 seeded by init_by_array({hi, lo}) of the widened
    seed (commons-math3 MersenneTwister(long)): reproduce with numpy's legacy array seeding (= init_by_array) and invert our
    increments back to uniforms; positive and negative (sign-extended) seeds."""
    from scipy.stats import norm
    for seed in (31415, -7, 1):
        n_paths = 50
        inc = fm.mersenne_increments(seed, [1.0], 1, n_paths)[0, 0]
        wide = seed & 0xffffffffffffffff
        rs = np.random.RandomState(np.array([wide >> 32, wide & 0xffffffff], dtype=np.uint32))
        raw = rs._bit_generator.random_raw(2 * n_paths).astype(np.uint64)
        u = (((raw[0::2] >> np.uint64(6)) << np.uint64(26)) | (raw[1::2] >> np.uint64(6))).astype(np.float64) * 2.0 ** -52
        assert np.max(np.abs(norm.cdf(inc) - u)) < 1e-15
    # published known answers of mt19937ar.c: init_genrand(5489) -> 3499211612; init_by_array({0x123,0x234,0x345,0x456}) -> 1067595299
    assert int(np.random.RandomState(5489)._bit_generator.random_raw(1)[0]) == 3499211612
    assert int(np.random.RandomState(np.array([0x123, 0x234, 0x345, 0x456], dtype=np.uint32))._bit_generator.random_raw(1)[0]) == 1067595299


def test_draw_order_and_scaling(fm):
    dt = [0.25, 1.0]
    a = fm.mersenne_increments(7, dt, 2, 3)             # [step][factor][path]
    flat = fm.mersenne_increments(7, [1.0], 1, 12)[0, 0]   # 12 consecutive draws
    k = 0
    for path in range(3):
        for step in range(2):
            for factor in range(2):
                assert a[step, factor, path] == flat[k] * math.sqrt(dt[step])
                k += 1


def test_moments_reference_bounds(fm):
    n, dt = 200_000, 0.1
    z = fm.mersenne_increments(1234, [dt], 1, n)[0, 0]
    assert abs(z.mean()) < 3.0 * math.sqrt(dt) / math.sqrt(n)
    assert abs(z.var() - dt) < 4.0 * dt * math.sqrt(2.0 / n)
