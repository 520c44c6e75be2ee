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
