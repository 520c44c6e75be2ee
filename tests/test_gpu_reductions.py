"""GPU parity, reductions: fmhip_reduce_moments / getAverage / getVariance / getMin / getMax vs the twin's
Kahan-in-double loops (RandomVariableFromFloatArray.java:284-382).

Tolerance: the device sums fp32 values in fp64 in a tree; the twin sums sequentially with Kahan
compensation.  Both carry O(1e-16) relative error w.r.t. Σ|x|, so |gpu - twin| ≤ 1e-13 · Σ|x| / n is asserted
(min/max: exact, including NaN propagation and -0.0 < +0.0 of java.lang.Math.min/max)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SIZES = [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2049, 100000, 1000003]


def dv(gpu, a):
    return gpu.DeviceVector.from_host(np.asarray(a, dtype=np.float32))


@pytest.mark.parametrize("n", SIZES)
def test_moments_vs_oracle(gpu, oracle, n):
    x = oracle.f_from_double(oracle.java_random_doubles(100 + n, n) * 10.0 - 3.0)
    v = dv(gpu, x)
    scale = float(np.abs(x.astype(np.float64)).sum())
    for shift in (0.0, float(oracle.f_average(x))):
        m = v.moments(shift)
        want = oracle.f_moments(x, shift)
        assert abs(m.sum - want[0]) <= 1e-13 * scale + 1e-300
        assert abs(m.sumsq - want[1]) <= 1e-13 * float(((x.astype(np.float64) - shift) ** 2).sum()) + 1e-300
        assert m.min == want[2] and m.max == want[3]


@pytest.mark.parametrize("n", SIZES)
def test_rv_reductions_vs_twin(gpu, oracle, n):
    d = oracle.java_random_doubles(200 + n, n) * 2.0 - 0.5
    xo = oracle.RandomVariableFloatFactory().createRandomVariable(0.0, d) if n > 0 else None
    xg = gpu.RandomVariableHipFactory().createRandomVariable(0.0, d)
    assert abs(xg.getAverage() - xo.getAverage()) <= 1e-13
    assert abs(xg.getVariance() - xo.getVariance()) <= 1e-13
    assert abs(xg.getSampleVariance() - xo.getSampleVariance()) <= 1e-13
    assert abs(xg.getStandardDeviation() - xo.getStandardDeviation()) <= 1e-13
    assert abs(xg.getStandardError() - xo.getStandardError()) <= 1e-13
    assert xg.getMin() == xo.getMin() and xg.getMax() == xo.getMax()


def test_min_max_special_values(gpu, oracle):
    cases = [
        [1.0, float("nan"), 3.0], [float("nan")], [0.0, -0.0], [-0.0, 0.0], [float("inf"), -float("inf"), 1.0],
        [5.0] * 1000 + [float("nan")] + [7.0] * 1000, [-0.0] * 300 + [0.0] + [-0.0] * 300,
    ]
    for c in cases:
        x = np.array(c, dtype=np.float32)
        m = dv(gpu, x).moments()
        wmin, wmax = oracle.f_min(x), oracle.f_max(x)
        for got, want in ((m.min, wmin), (m.max, wmax)):
            assert (math.isnan(got) and math.isnan(want)) or (got == want and math.copysign(1, got) == math.copysign(1, want)), (c[:4], got, want)


def test_empty_vector(gpu):
    v = dv(gpu, np.zeros(0))
    m = v.moments()
    assert m.sum == 0.0 and m.sumsq == 0.0 and m.min == 1.7976931348623157e308 and m.max == -1.7976931348623157e308
    rv = gpu.RandomVariableHip(0.0, v)
    assert math.isnan(rv.getAverage()) and math.isnan(rv.getVariance())     # twin:318-320, :364-366


def test_weighted_and_quantiles(gpu, oracle):
    d = oracle.java_random_doubles(77, 50001)
    w = oracle.java_random_doubles(78, 50001)
    xo, wo = [oracle.RandomVariableFloatFactory().createRandomVariable(0.0, a) for a in (d, w)]
    xg, wg = [gpu.RandomVariableHipFactory().createRandomVariable(0.0, a) for a in (d, w)]
    # fp32 product rounding (GPU class) vs exact product (twin): fp32 tolerance
    assert abs(xg.getAverage(wg) - xo.getAverage(wo)) <= 1e-7
    assert abs(xg.getVariance(wg) - xo.getVariance(wo)) <= 1e-6 * xo.getVariance(wo)
    # RandomVariableCuda.getQuantile uses (1 - quantile) (:983); the twin uses quantile (twin:484)
    for q in (0.0, 0.05, 0.5, 0.95, 1.0):
        assert xg.getQuantile(q) == xo.getQuantile(1.0 - q)
    assert abs(xg.getQuantileExpectation(0.1, 0.9) - np.sort(xo.getRealizations())[
        max(int(math.floor(50002 * 0.1 - 1 + 0.5)), 0):int(math.floor(50002 * 0.9 - 1 + 0.5)) + 1].mean()) <= 1e-12
    h = xg.getHistogram([0.25, 0.5, 0.75])
    assert abs(h.sum() - 1.0) < 1e-12 and all(abs(v - 0.25) < 0.01 for v in h)
    anchors, hist = xg.getHistogram(5, 2.0)
    assert len(anchors) == 6 and len(hist) == 6


@pytest.mark.parametrize("n", [(1 << 24) + 3, 4_194_304, 4_194_305, 5_000_003, 9_000_001, 12_000_001, 14_680_064, 14_680_065])
def test_large_vector_closed_form(gpu, n):
    """Full-size property check (no oracle pass needed): values k mod 7 → exact sums.  The sizes cover every layout of the
    arrival counting of the fused final combine (fm_kernel_parts.hpp block_combine: 1 group below 512 workgroups of
    8192 elements, 2…6 groups up to 1791, 7 above) and its boundaries; run twice, the counters must be back at zero."""
    x = (np.arange(n, dtype=np.int64) % 7).astype(np.float32)
    x[n // 3] = -2.0
    xs = x.astype(np.float64)
    v = dv(gpu, x)
    for _ in range(2):
        m = v.moments()
        assert m.sum == xs.sum() and m.sumsq == (xs * xs).sum() and m.min == -2.0 and m.max == 6.0


@pytest.mark.parametrize("n", [520_000, 524_288, 524_289, 600_000, 1_000_000, 2_097_152, 2_200_000, 3_999_999, 4_194_303])
def test_one_row_that_counts_its_arrivals_in_groups(gpu, n):
    """A row of ONE arithmetic group whose launch has 256 or more workgroups (round 4: fm_kernel_parts.hpp block_combine, FM_COUNT_IN_GROUPS_FROM)
    counts its arrivals in seven groups and a second level; the sums are still those of one group.  The sizes sit on both sides of that
    threshold for a unit per workgroup (one row of up to 128 spans: 254, 256, 257, 293, 489 workgroups) and for a span per workgroup
    (256 … 511 spans); each vector alone, twice (the counters of both levels must be back at zero), then as one of three rows of a launch
    (another launch shape: a span per workgroup, fewer than 256 of them per row for the smaller sizes) — exact sums, the same bits."""
    x = (np.arange(n, dtype=np.int64) % 5).astype(np.float32)
    x[n // 7] = -3.0
    x[n - 1] = 9.0
    xs = x.astype(np.float64)
    v = dv(gpu, x)
    alone = []
    for _ in range(2):
        m = v.moments()
        assert m.sum == xs.sum() and m.sumsq == (xs * xs).sum() and m.min == -3.0 and m.max == 9.0, n
        alone.append((m.sum, m.sumsq, m.min, m.max))
    others = [dv(gpu, np.full(n, 0.5, dtype=np.float32)), dv(gpu, (np.arange(n) % 3).astype(np.float32))]
    p = gpu.Program(1); p.reduce(0); p.compile()
    for _ in range(2):
        m = np.asarray(p.run([[others[0]], [v], [others[1]]])[1]).reshape(3, 4)
        assert tuple(m[1]) == alone[0], n
        assert m[0][0] == 0.5 * n and m[2][3] == 2.0


def test_hand_off_with_warm_caches_and_many_workgroups_per_cu(gpu, oracle):
    """The cross-workgroup hand-off of the partials (fm_kernel_parts.hpp: block_combine) where a stale cache line would show: the SAME
    output buffers, partial slots and arrival counters are reused launch after launch with DIFFERENT data (A, B, A, C, …), many rows
    per launch and several workgroups per CU, a light program (the workgroups arrive in a burst), on both execution tiers.  Every
    moment of every row of every launch against the oracle on that launch's data: a partial read from a previous launch's line
    (the consumer's L1, or the L2 of another XCD) would be the other data set's sum."""
    n, rows, sets = 300_007, 48, 3
    data = [[oracle.f_from_double(oracle.java_random_doubles(9000 + 100 * s + r, n) * (s + 1.5) - 0.25 * r) for r in range(rows)] for s in range(sets)]
    want = [[oracle.f_moments(oracle.f_v1s1("MULT_S", x, 1.25), 0.0) for x in row] for row in data]
    for tier in (gpu.JIT_OFF, gpu.JIT_SYNC):
        prev = gpu.set_jit(tier)
        try:
            p = gpu.Program(1)
            w = p.op("MULT_S", 0, s=1.25)
            p.output(w); p.reduce(w); p.compile()
            dev = [[[gpu.DeviceVector.from_host(x)] for x in row] for row in data]
            outs = [[gpu.DeviceVector.filled(n, 0.0)] for _ in range(rows)]
            for s in (0, 1, 0, 2, 1, 1, 0, 2, 2, 0):
                m = np.asarray(p.run_into(dev[s], outs)).reshape(rows, 4)
                for r in range(rows):
                    scale = float(np.abs(data[s][r].astype(np.float64)).sum()) * 1.25
                    assert abs(m[r][0] - want[s][r][0]) <= 1e-13 * scale, (tier, s, r)
                    assert m[r][2] == want[s][r][2] and m[r][3] == want[s][r][3], (tier, s, r)
        finally:
            gpu.set_jit(prev)


def test_special_values_in_both_launch_shapes_and_inside_a_loop_kernel(gpu, oracle):
    """NaN, ±inf, −0 anywhere in a vector: the moments of the twin's loops whatever computes them — a unit-shaped launch (few rows), a
    span-shaped one (many rows: 600 spans > 128), shifted or not, both tiers, and the fused expectation of a peeled chain."""
    import importlib
    rolled = importlib.import_module("test_gpu_rolled")
    n = 70_001                                              # 9 spans; the last unit is ragged
    base = oracle.f_from_double(oracle.java_random_doubles(5, n) - 0.5)
    specials = {"nan in the last unit": (n - 3, np.nan), "nan in lane 0": (0, np.nan), "+inf": (2048 * 5 + 17, np.inf), "-inf": (8191, -np.inf),
                "-0 among zeros": None}
    for name, where in specials.items():
        x = base.copy()
        if where is None:
            x[:] = 0.0; x[4097] = -0.0
        else:
            x[where[0]] = where[1]
        want_min, want_max = oracle.f_min(x), oracle.f_max(x)
        v = dv(gpu, x)
        for tier in (gpu.JIT_OFF, gpu.JIT_SYNC):
            prev = gpu.set_jit(tier)
            try:
                few = v.moments()
                import ctypes as C
                handles = (C.c_int64 * 70)(*[v.handle] * 70)
                out = (gpu.Moments * 70)()
                gpu._native.check(gpu.lib().fmhip_reduce_moments_batch(handles, 70, None, out))
                many = out[33]
                with np.errstate(invalid="ignore"):
                    want_sum = np.sum(x.astype(np.float64))
                for m in (few, many):
                    assert np.float64(m.sum).tobytes() == np.float64(few.sum).tobytes() and np.float64(m.sumsq).tobytes() == np.float64(few.sumsq).tobytes(), (name, tier)
                    assert (math.isnan(m.sum) and math.isnan(want_sum)) or m.sum == want_sum or abs(m.sum - want_sum) <= 1e-13 * np.sum(np.abs(x.astype(np.float64))), (name, tier)
                    for got, want in ((m.min, want_min), (m.max, want_max)):
                        assert (math.isnan(got) and math.isnan(want)) or (got == want and math.copysign(1, got) == math.copysign(1, want)), (name, tier, got, want)
            finally:
                gpu.set_jit(prev)
    # … and through the fused expectation of a peeled chain: a NaN in one period's rate makes the product's value NaN on that path
    periods, strike, delta = 24, 0.02, 0.5
    rng = np.random.default_rng(9)
    libors = [oracle.f_from_double(rng.uniform(0.005, 0.04, n)) for _ in range(periods)]
    libors[7][n - 1] = np.nan
    num = oracle.f_from_double(rng.uniform(1.0, 1.3, n))
    prev_fusion, prev_jit = gpu.set_fusion(True), gpu.set_jit(gpu.JIT_SYNC)
    try:
        dev = [gpu.DeviceVector.from_host(a) for a in libors]
        dnum = gpu.DeviceVector.from_host(num)
        results = []
        for _ in range(3):                                   # discovery, then the peeled kernel with its reduction
            with gpu.holding():
                c = rolled.swaption_like_chain(lambda p: dev[p], periods, dnum, strike, delta)
            results.append((c.moments(), c.to_float32()))
        for m, values in results:
            assert math.isnan(m.sum) and math.isnan(m.sumsq) and math.isnan(m.min) and math.isnan(m.max)
            assert np.isnan(values[n - 1]) and np.isfinite(values[: n - 1]).all()
    finally:
        gpu.set_jit(prev_jit)
        gpu.set_fusion(prev_fusion)


def test_expectations_in_two_halves(gpu, oracle):
    """fmhip_reduce_moments_batch_begin / _end: the reduction is enqueued at once, the second half waits for THAT reduction (work enqueued
    in between does not hold it up, and is not lost), the moments are those of the blocking call; a ticket ends once, with its count."""
    import ctypes as C
    n, k = 300_007, 9
    xs = [oracle.f_from_double(oracle.java_random_doubles(900 + i, n) - 0.4) for i in range(k)]
    vecs = [dv(gpu, x) for x in xs]
    prev = gpu.set_fusion(True)
    try:
        pending = [v.v1s1("MULT_S", 1.5).v1s1("ADD_S", 0.25) for v in vecs]                 # lazily recorded inputs: the first half runs them
        first = gpu.reduce_moments_batch_begin(pending, shifts=[0.01 * i for i in range(k)])
        later = [p.v1s0("EXP").v2s0("MULT", p) for p in pending]                              # more work behind the first reduction …
        second = gpu.reduce_moments_batch_begin(later)                                        # … and a second ticket in flight
        got_second = gpu.reduce_moments_batch_end(second, k)                                  # ended out of order
        got_first = gpu.reduce_moments_batch_end(first, k)
        for i in range(k):
            for got, vec, shift in ((got_first[i], pending[i], 0.01 * i), (got_second[i], later[i], 0.0)):
                m = vec.moments(shift)
                assert (got.sum, got.sumsq, got.min, got.max) == (m.sum, m.sumsq, m.min, m.max), i
        lib, out = gpu.lib(), (gpu.Moments * k)()
        assert lib.fmhip_reduce_moments_batch_end(first, out, k) == gpu._native.ERR_INVALID_HANDLE            # ended already
        third = gpu.reduce_moments_batch_begin(vecs)
        assert lib.fmhip_reduce_moments_batch_end(third, out, k - 1) == gpu._native.ERR_SIZE_MISMATCH         # (and the ticket is gone)
        assert lib.fmhip_reduce_moments_batch_end(third, out, k) == gpu._native.ERR_INVALID_HANDLE
        ticket = C.c_int64(0)
        assert lib.fmhip_reduce_moments_batch_begin(None, 1, None, C.byref(ticket)) == gpu._native.ERR_INVALID_ARGUMENT
    finally:
        gpu.set_fusion(prev)
