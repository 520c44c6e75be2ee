"""The table-driven log (csrc/fm_device_math.hpp: log_f, table csrc/fm_log_table.hpp) on the arguments where its pieces
meet: both sides of every grid interval (257 grid points over the mantissa range), the neighbourhood of 1 (where the
result is tiny and the table entry must vanish exactly), the seam at sqrt(1/2) (entries below it carry log(2c) - ln2),
powers of two, the denormals, the extremes and the special values — bit for bit against the oracle
(`(float)Math.log(x)`, RandomVariableFromFloatArray.java:920), on both execution tiers and both kernel variants.
The full sweep over all 2^32 arguments is benchmarks/exhaustive_unary.py (not a pytest: minutes)."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


def critical_arguments():
    parts = []
    k = np.arange(-40, 41, dtype=np.int64)
    for e in (-149, -140, -127, -126, -64, -2, -1, 0, 1, 2, 63, 126, 127):        # x = 2^e * m, incl. denormal scales
        base = np.float64(2.0) ** e
        # grid points c = 0.5 + i/512 and the interval mid-points (where the nearest grid point changes), +- 40 ulp each
        pts = np.concatenate([0.5 + np.arange(0, 257) / 512.0, 0.5 + (np.arange(0, 256) + 0.5) / 512.0])
        m = (pts[:, None] + k[None, :] * 2.0 ** -24).ravel()
        parts.append((m * base).astype(np.float32))
    one = np.float32(1.0).view(np.uint32)
    parts.append((one + np.arange(-5000, 5001, dtype=np.int64)).astype(np.uint32).view(np.float32))       # around 1
    seam = np.float32(0.70710678).view(np.uint32)
    parts.append((seam + np.arange(-5000, 5001, dtype=np.int64)).astype(np.uint32).view(np.float32))      # around sqrt(1/2)
    parts.append(np.arange(1, 20001, dtype=np.uint32).view(np.float32))                                     # smallest denormals
    parts.append((np.uint32(0x7f800000) - np.arange(0, 20001, dtype=np.uint32)).view(np.float32))           # largest finite, +inf
    parts.append(np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -1.0, -1e-40, -3.4e38, 1.0, 2.0, 0.5], dtype=np.float32))
    rng = np.random.default_rng(2718)
    parts.append(rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32))      # any bit pattern
    x = np.concatenate(parts)
    return x


def run_log(gpu, x, wide):
    p = gpu.Program(1)
    v = p.op("LOG", 0)
    if wide:                                # more than 8 live values: the 4-elements-per-lane variant of both tiers
        keep = [p.op("ADD_S", 0, s=float(i)) for i in range(10)]
        acc = v
        for t in keep:
            acc = p.op("CHOOSE", 0, acc, t)            # x >= 0 ? acc : t — keeps every t alive, result is log(x) where x >= 0
        p.output(v)
        p.output(acc)
    else:
        p.output(v)
    p.compile()
    outs, _ = p.run([[gpu.DeviceVector.from_host(x)]])
    return outs[0][0].to_float32()


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("jit", ["off", "sync"])
def test_log_critical_arguments(gpu, oracle, jit, wide):
    x = critical_arguments()
    with np.errstate(all="ignore"):
        want = oracle.f_v1s0("LOG", x)
    prev = gpu.set_jit(gpu.JIT_OFF if jit == "off" else gpu.JIT_SYNC)
    try:
        got = run_log(gpu, x, wide)
    finally:
        gpu.set_jit(prev)
    assert_bits_equal(got, want, f"log, jit={jit}, wide={wide}")


def test_log_then_more_logs_in_one_program(gpu, oracle):
    """Several log micro-ops in one program share one table copy; the table must survive the passes of a multi-pass
    workgroup (fused reduction: 4 passes per workgroup)."""
    n = 300_001
    x = oracle.f_from_double(oracle.java_random_doubles(5, n) * 50.0 + 1e-3)
    with np.errstate(all="ignore"):
        a = oracle.f_v1s0("LOG", x)
        b = oracle.f_v1s0("LOG", oracle.f_v1s0("ABS", a))
        want = oracle.f_v2s0("ADD", a, b)
    for mode in (gpu.JIT_OFF, gpu.JIT_SYNC):
        prev = gpu.set_jit(mode)
        try:
            p = gpu.Program(1)
            la = p.op("LOG", 0)
            lb = p.op("LOG", p.op("ABS", la))
            w = p.op("ADD", la, lb)
            p.output(w)
            p.reduce(w)
            p.compile()
            outs, m = p.run([[gpu.DeviceVector.from_host(x)]])
        finally:
            gpu.set_jit(prev)
        assert_bits_equal(outs[0][0].to_float32(), want, f"log chain, mode {mode}")
        fin = want[np.isfinite(want)].astype(np.float64)
        if fin.size == want.size:
            assert abs(m[0][0][0] - fin.sum()) <= 1e-9 * np.abs(fin).sum()
