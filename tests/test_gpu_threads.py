"""Threading contract of the boundary: any thread may call, concurrently; finmath's optimiser may be
multi-threaded (LIBORMarketModelCalibrationATMTest.java:319 numberOfThreads).  The reference funnels every CUDA call
through one executor thread (RandomVariableCuda.java:155); here the entry points are re-entrant."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fusion", [False, True])
def test_concurrent_callers(gpu, oracle, fusion):
    n, n_threads, reps = 20011, 6, 25
    data = [oracle.java_random_doubles(4000 + t, n) for t in range(n_threads)]
    want = []
    of = oracle.RandomVariableFloatFactory()
    for t in range(n_threads):
        x = of.createRandomVariable(0.0, data[t])
        r = x.add(1.0 + t).mult(x).sub(0.25).squared().cap(50.0).addProduct(x, 0.5 + t).discount(x, 0.5)
        want.append((r.getRealizations(), r.getAverage(), r.getMin(), r.getMax()))
    errors = []
    prev = gpu.set_fusion(fusion)

    def worker(t):
        try:
            f = gpu.RandomVariableHipFactory()
            for _ in range(reps):
                x = f.createRandomVariable(0.0, data[t])
                r = x.add(1.0 + t).mult(x).sub(0.25).squared().cap(50.0).addProduct(x, 0.5 + t).discount(x, 0.5)
                got = r.getRealizations()
                assert (got == want[t][0]).all()
                assert abs(r.getAverage() - want[t][1]) <= 1e-12 and r.getMin() == want[t][2] and r.getMax() == want[t][3]
        except Exception as e:          # noqa: BLE001
            errors.append((t, repr(e)))

    try:
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
        for th in threads: th.start()
        for th in threads: th.join(timeout=300)
    finally:
        gpu.set_fusion(prev)
    assert not errors, errors
    assert all(not th.is_alive() for th in threads)


def test_expectations_from_several_threads_while_the_device_works(gpu, oracle):
    """`getAverage()` of a pending chain holds the engine lock for its bookkeeping only; the wait for the device happens without it
    (abi.cpp: fmhip_reduce_moments — the moments arrive in a slot of pinned memory of their own).  Several threads value long chains
    at the same time: every thread gets the moments of ITS chain (bits equal to the single-threaded ones), whoever's launch finishes
    first, and more launches are in flight than there are result slots only if the engine falls back to waiting under the lock."""
    import importlib
    rolled = importlib.import_module("test_gpu_rolled")
    n, periods, n_threads, reps = 200_003, 30, 8, 40
    rng = np.random.default_rng(2024)
    libors = [oracle.f_from_double(rng.uniform(0.005, 0.04, n)) for _ in range(periods)]
    num = oracle.f_from_double(rng.uniform(1.0, 1.3, n))
    prev_fusion, prev_jit = gpu.set_fusion(True), gpu.set_jit(gpu.JIT_SYNC)
    prev_hold = 0
    try:
        dev = [gpu.DeviceVector.from_host(x) for x in libors]
        dnum = gpu.DeviceVector.from_host(num)

        prev_hold = gpu.fusion_hold(1)                       # (one hold for all threads: nothing runs before a value is asked for)

        def value(t, shift=0.0):
            c = rolled.swaption_like_chain(lambda p: dev[(p + t) % periods], periods - t, dnum, 0.02 + 0.001 * t, 0.5)
            m = c.moments(shift)
            return (m.sum, m.sumsq, m.min, m.max)
        want = []
        for t in range(n_threads):
            value(t); value(t)                               # discovery, kernels
            want.append(value(t))
        errors = []

        def worker(t):
            try:
                for _ in range(reps):
                    got = value(t)
                    assert got == want[t], (t, got, want[t])
            except Exception as e:          # noqa: BLE001
                errors.append((t, repr(e)))
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
        for th in threads: th.start()
        for th in threads: th.join(timeout=300)
        assert not errors, errors[:3]
        assert all(not th.is_alive() for th in threads)
    finally:
        gpu.fusion_hold(prev_hold)
        gpu.flush()
        gpu.set_jit(prev_jit)
        gpu.set_fusion(prev_fusion)
