"""GPU parity, element-wise: every opcode of include/fmhip.h through the C-ABI vs the CPU oracle
(restatement of RandomVariableFromFloatArray.java) on the same seeded inputs.

Tolerances (written here, as the task demands):
  * + - * / min max abs sqrt squared invert choose isNaN accrue discount addProduct addRatio subRatio:
    BIT-EXACT (NaN == NaN).
  * exp log pow sin cos: evaluated in fp64 on both sides and narrowed once; two fp64 libms may differ in
    the last fp64 ulp, which changes the fp32 result only when it straddles a rounding boundary
    (p ≈ 2^-28 per element).  Asserted: at most 1 fp32 ulp, on at most 1e-5 of the elements.
"""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

EXACT_V1S0 = ["SQUARED", "SQRT", "INVERT", "ABS", "ISNAN"]
LIBM_V1S0 = ["EXP", "LOG", "SIN", "COS"]
V1S1 = ["CAP_S", "FLOOR_S", "ADD_S", "SUB_S", "BUS_S", "MULT_S", "DIV_S", "VID_S"]
V2S0 = ["CAP", "FLOOR", "ADD", "SUB", "MULT", "DIV"]
V2S1 = ["ACCRUE", "DISCOUNT", "ADDPRODUCT_VS"]
V3S0 = ["ADDPRODUCT", "ADDRATIO", "SUBRATIO", "CHOOSE"]
SCALARS = [1.0 / 3.0, 3.1415, -2.0, 0.0, 1e-3]

RAGGED_SIZES = [0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4097, 100000]


def ulp_diff(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    d = np.abs(ia - ib)
    d[np.isnan(a) & np.isnan(b)] = 0
    return d


def assert_libm_close(got, want, what):
    d = ulp_diff(got, want)
    assert d.max(initial=0) <= 1, f"{what}: max ulp diff {d.max()}"
    assert (d > 0).mean() <= 1e-5 if d.size else True, f"{what}: {(d > 0).sum()} of {d.size} differ by 1 ulp"
    assert (np.isnan(got) == np.isnan(want)).all(), what


@pytest.fixture(scope="module")
def xyz(oracle):
    """x = the reference test's stream `new Random(31415).nextDouble()` (RandomVariableGPUTest.java:194-201);
    y, z from seeds 27182 / 16180."""
    n = 100000
    x = oracle.f_from_double(oracle.java_random_doubles(31415, n))
    y = oracle.f_from_double(oracle.java_random_doubles(27182, n) + 0.5)
    z = oracle.f_from_double(oracle.java_random_doubles(16180, n) - 0.5)
    return x, y, z


def edge_values():
    e = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754942e-38, 3.4028235e38,
                  -3.4028235e38, 1e-20, 1e20, 0.5, 2.0, 88.0, -88.0, 89.0, -104.0, 1.0000001, 0.99999994,
                  3.1415927, 1e-7, 16777216.0, -16777217.0, 0.1, 1.0 / 3.0], dtype=np.float32)
    a, b = np.meshgrid(e, e)
    return a.ravel().copy(), b.ravel().copy()


def dv(gpu, arr):
    return gpu.DeviceVector.from_host(np.asarray(arr, dtype=np.float32))


@pytest.mark.parametrize("op", EXACT_V1S0)
def test_unary_exact(gpu, oracle, xyz, op):
    x, y, z = xyz
    for v in (x, y, z, edge_values()[0]):
        assert_bits_equal(dv(gpu, v).v1s0(op).to_float32(), oracle.f_v1s0(op, v), op)


@pytest.mark.parametrize("op", LIBM_V1S0)
def test_unary_libm(gpu, oracle, xyz, op):
    x, y, z = xyz
    with np.errstate(all="ignore"):
        for v in (x, y, z * 20.0, edge_values()[0]):
            assert_libm_close(dv(gpu, v).v1s0(op).to_float32(), oracle.f_v1s0(op, v), op)


@pytest.mark.parametrize("op", V1S1)
def test_scalar_ops_exact(gpu, oracle, xyz, op):
    x, y, z = xyz
    for v in (x, z, edge_values()[0]):
        for s in SCALARS + [float("nan"), float("inf"), -0.0]:
            assert_bits_equal(dv(gpu, v).v1s1(op, s).to_float32(), oracle.f_v1s1(op, v, s), f"{op}({s})")


def test_pow(gpu, oracle, xyz):
    x, y, z = xyz
    for v in (x, y, edge_values()[0]):
        for s in (0.5, 2.0, -1.0, 0.0, 3.0, 1.0 / 3.0, float("nan"), float("inf")):
            assert_libm_close(dv(gpu, v).v1s1("POW_S", s).to_float32(), oracle.f_v1s1("POW_S", v, s), f"POW_S({s})")


@pytest.mark.parametrize("op", V2S0)
def test_binary_exact(gpu, oracle, xyz, op):
    x, y, z = xyz
    ea, eb = edge_values()
    for a, b in ((x, y), (z, x), (y, z), (ea, eb)):
        assert_bits_equal(dv(gpu, a).v2s0(op, dv(gpu, b)).to_float32(), oracle.f_v2s0(op, a, b), op)


@pytest.mark.parametrize("op", V2S1)
def test_binary_scalar_exact(gpu, oracle, xyz, op):
    x, y, z = xyz
    ea, eb = edge_values()
    for a, b in ((x, y), (z, x), (ea, eb)):
        for s in (2.0, 1.0 / 3.0, 0.5, -1.0):
            assert_bits_equal(dv(gpu, a).v2s1(op, dv(gpu, b), s).to_float32(), oracle.f_v2s1(op, a, b, s), f"{op}({s})")


@pytest.mark.parametrize("op", V3S0)
def test_ternary_exact(gpu, oracle, xyz, op):
    x, y, z = xyz
    ea, eb = edge_values()
    for a, b, c in ((x, y, z), (z, x, y), (ea, eb, ea[::-1].copy())):
        got = dv(gpu, a).v3s0(op, dv(gpu, b), dv(gpu, c)).to_float32()
        assert_bits_equal(got, oracle.f_v3s0(op, a, b, c), op)


@pytest.mark.parametrize("n", RAGGED_SIZES)
def test_ragged_sizes(gpu, oracle, n):
    """Empty, tiny and non-multiple-of-4/256 sizes: every lane of the 128-bit tail access must be right."""
    a = oracle.f_from_double(oracle.java_random_doubles(7 + n, n) - 0.5)
    b = oracle.f_from_double(oracle.java_random_doubles(11 + n, n) + 0.5)
    va, vb = dv(gpu, a), dv(gpu, b)
    assert_bits_equal(va.v2s0("DIV", vb).to_float32(), oracle.f_v2s0("DIV", a, b), f"DIV n={n}")
    assert_bits_equal(va.v3s0("CHOOSE", vb, va).to_float32(), oracle.f_v3s0("CHOOSE", a, b, a), f"CHOOSE n={n}")
    assert_bits_equal(va.v1s1("ADD_S", 4.0).to_float32(), oracle.f_v1s1("ADD_S", a, 4.0), f"ADD_S n={n}")
    assert va.to_float64().dtype == np.float64 and (va.to_float64() == a.astype(np.float64)).all()


def test_upload_narrows_like_reference(gpu, oracle):
    """double[] → float[] narrowing at the boundary (RandomVariableCuda.java:768-774)."""
    d = oracle.java_random_doubles(5, 10001) * 1e3 - 500.0
    v = gpu.DeviceVector.from_host(d)
    assert_bits_equal(v.to_float32(), oracle.f_from_double(d), "upload")


def test_errors(gpu):
    a = dv(gpu, np.arange(8)); b = dv(gpu, np.arange(9))
    with pytest.raises(gpu.FmhipError) as e:
        a.v2s0("ADD", b)
    assert e.value.code == -2           # FMHIP_ERR_SIZE_MISMATCH
    with pytest.raises(gpu.FmhipError) as e:
        gpu.DeviceVector(987654321, 8).to_float32()
    assert e.value.code == -1           # FMHIP_ERR_INVALID_HANDLE
    with pytest.raises(gpu.FmhipError) as e:
        a.v1s0("ADD")                   # wrong call shape for the opcode
    assert e.value.code == -5


@pytest.mark.parametrize("op", ["EXP", "LOG"])
def test_fast_math_mode_within_2_ulp(gpu, oracle, xyz, op):
    """FMHIP_MATH_FAST: hardware v_exp_f32 / v_log_f32 with fp32 range reduction.  Stated tolerance: 2 fp32 ulp of the
    twin's `(float)Math.exp/log((double)x)` (the accuracy class of the CUDA expf/logf the reference kernels call);
    special values (0, negatives, ±inf, NaN, denormals, overflow/underflow) exactly as in exact mode."""
    x, y, z = xyz
    prev = gpu.set_math_mode(gpu.MATH_FAST)
    try:
        with np.errstate(all="ignore"):
            worst = 0
            wide = np.concatenate([x * 170.0 - 85.0, np.logspace(-44, 38, 20001).astype(np.float32), 1.0 + (x - 0.5) * 1e-3])
            for v in (x, y, z * 20.0, wide, edge_values()[0]):
                got = dv(gpu, v).v1s0(op).to_float32()
                want = oracle.f_v1s0(op, v)
                d = ulp_diff(got, want)
                assert (np.isnan(got) == np.isnan(want)).all(), op
                assert (np.isinf(got) == np.isinf(want)).all() or op == "EXP", op
                worst = max(worst, int(d[np.isfinite(want) & np.isfinite(got)].max(initial=0)))
            assert worst <= 2, f"{op}: max ulp error {worst}"
    finally:
        gpu.set_math_mode(prev)
    # exact mode is restored: bit-level agreement again
    assert_libm_close(dv(gpu, x).v1s0(op).to_float32(), oracle.f_v1s0(op, x), op)
