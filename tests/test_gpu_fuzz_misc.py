"""Randomised checks of the two remaining kernels against the oracle:
  * fm_bm_kernel: random (steps, factors, paths, path offset, seed, non-uniform Δt) — every increment bit for bit;
  * reductions: random lengths and contents (±0, ±inf, NaN, denormals, huge values), random shifts — Σ, Σ² to fp64 summation
    tolerance, min/max and the NaN rule exactly; single, batched and device-side entry points agree bit for bit."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SCALE = int(__import__("os").environ.get("FMHIP_FUZZ_SCALE", "1"))      # one-off deep runs: FMHIP_FUZZ_SCALE=25


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_random_brownian_motion_configurations(gpu, oracle, seed):
    rng = np.random.default_rng(31000 + seed)
    steps, factors = int(rng.integers(1, 6)), int(rng.integers(1, 5))
    paths = int(rng.choice([1, 3, 4, 5, 255, 1021, 4096, 10007]))
    offset = int(rng.choice([0, 1, 2, 3, 4, 1000003, 2 ** 33 + 1]))
    times = np.concatenate([[0.0], np.cumsum(rng.uniform(0.01, 0.7, steps))])
    s = int(rng.integers(-2 ** 40, 2 ** 40))
    td = gpu.TimeDiscretization(times.tolist())
    bm = gpu.BrownianMotionHip(td, factors, paths, s, path_offset=offset)
    want = oracle.bm_generate(s, [td.getTimeStep(i) for i in range(steps)], factors, paths, path_offset=offset)
    for i in range(steps):
        for f in range(factors):
            inc = bm.getBrownianIncrement(i, f)
            assert inc.getFiltrationTime() == td.getTime(i + 1)
            got = inc.realizations.to_float32()
            assert (got.view(np.uint32) == want[i][f].view(np.uint32)).all(), (i, f)


def special_vector(oracle, rng, n):
    x = oracle.f_from_double(oracle.java_random_doubles(int(rng.integers(1, 1 << 30)), n) * 200.0 - 100.0)
    kinds = [np.float32(0.0), np.float32(-0.0), np.float32(np.inf), np.float32(-np.inf), np.float32(np.nan), np.float32(1e-45),
             np.float32(-1e-45), np.float32(3.4028235e38), np.float32(-3.4028235e38)]
    for _ in range(int(rng.integers(0, 4))):
        if n:
            x[rng.integers(n)] = kinds[rng.integers(len(kinds))]
    return x


@pytest.mark.parametrize("seed", range(40 * SCALE))
def test_random_reductions(gpu, oracle, seed):
    rng = np.random.default_rng(52000 + seed)
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 1023, 2047, 2048, 2049, 8191, 8192, 8193, 65537, 300001]))
    x = special_vector(oracle, rng, n)
    shift = float(rng.choice([0.0, 0.0, 1.5, -7.25, 100.0]))
    v = gpu.DeviceVector.from_host(x)
    with np.errstate(all="ignore"):
        m = v.moments(shift)
        want = oracle.f_moments(x, shift)
        xd = x.astype(np.float64) - shift
        scale = np.abs(xd[np.isfinite(xd)]).sum() + 1.0
        if np.isfinite(xd).all():
            refs = (want[0], want[1])
        else:
            # DELIBERATE DEVIATION: the twin's Kahan loop turns every infinity that is not the LAST element into
            # NaN (error = (newSum - sum) - value = inf - inf); the device returns the IEEE sum: ±inf, or NaN for inf - inf / NaN.
            refs = (float(np.sum(xd)), float(np.sum(xd * xd)))
        for got, ref, tol in ((m.sum, refs[0], 1e-13 * scale), (m.sumsq, refs[1], 1e-13 * (xd[np.isfinite(xd)] ** 2).sum() + 1e-300)):
            if math.isnan(ref):
                assert math.isnan(got)
            elif math.isinf(ref):
                assert got == ref
            else:
                assert abs(got - ref) <= tol, (got, ref)
        for got, ref in ((m.min, want[2]), (m.max, want[3])):
            assert (math.isnan(got) and math.isnan(ref)) or (got == ref and math.copysign(1.0, got) == math.copysign(1.0, ref)), (got, ref)
    # batched and device-side entry points: same bits as the single reduction
    k = int(rng.integers(1, 6))
    vecs = [v] + [gpu.DeviceVector.from_host(special_vector(oracle, rng, n)) for _ in range(k - 1)]
    handles = (C.c_int64 * k)(*[w.handle for w in vecs])
    shifts = (C.c_double * k)(*([shift] * k))
    out = (gpu.Moments * k)()
    gpu._native.check(gpu.lib().fmhip_reduce_moments_batch(handles, k, shifts, out))
    dev = gpu.DeviceVector.filled(8 * k, 0.0)
    gpu._native.check(gpu.lib().fmhip_reduce_moments_batch_device(handles, k, shifts, C.c_void_p(dev.device_ptr())))
    raw = dev.to_float32().view(np.float64).reshape(k, 4)
    for i, w in enumerate(vecs):
        one = w.moments(shift)
        a = np.array([one.sum, one.sumsq, one.min, one.max])
        b = np.array([out[i].sum, out[i].sumsq, out[i].min, out[i].max])
        assert a.tobytes() == b.tobytes() == raw[i].tobytes()
