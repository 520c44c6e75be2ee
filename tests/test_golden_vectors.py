"""Frozen vectors (tests/golden/rv_float_golden.json, self-generated — see make_golden.py): the oracle must keep
reproducing them (CPU), and the HIP path must hit them through the C-ABI (GPU)."""
import json
import os

import numpy as np
import pytest

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rv_float_golden.json")))
LIBM = {"EXP", "LOG", "SIN", "COS", "POW_S"}


def inputs(o):
    n = G["n"]
    return (o.f_from_double(o.java_random_doubles(31415, n)), o.f_from_double(o.java_random_doubles(27182, n) + 0.5),
            o.f_from_double(o.java_random_doubles(16180, n) - 0.5))


def expected(op):
    return np.array(G["ops"][op], dtype=np.uint32).view(np.float32)


def test_oracle_reproduces_golden(oracle):
    o = oracle
    x, y, z = inputs(o)
    k = G["keep"]
    for op in o.V1S0: assert (o.f_v1s0(op, y)[:k].view(np.uint32) == np.array(G["ops"][op], dtype=np.uint32)).all(), op
    for op in o.V1S1: assert (o.f_v1s1(op, z, 1.0 / 3.0)[:k].view(np.uint32) == np.array(G["ops"][op], dtype=np.uint32)).all(), op
    for op in o.V2S0: assert (o.f_v2s0(op, x, y)[:k].view(np.uint32) == np.array(G["ops"][op], dtype=np.uint32)).all(), op
    for op in o.V2S1: assert (o.f_v2s1(op, x, y, 1.0 / 3.0)[:k].view(np.uint32) == np.array(G["ops"][op], dtype=np.uint32)).all(), op
    for op in o.V3S0: assert (o.f_v3s0(op, z, x, y)[:k].view(np.uint32) == np.array(G["ops"][op], dtype=np.uint32)).all(), op
    r = G["reductions"]
    assert o.f_average(x) == float.fromhex(r["average_x"]) and o.f_variance(x) == float.fromhex(r["variance_x"])
    assert o.f_min(z) == float.fromhex(r["min_z"]) and o.f_max(z) == float.fromhex(r["max_z"])
    assert list(o.philox4x32_10([1, 2, 3, 4], [5, 6])) == G["philox4x32_10"]["ctr_1_2_3_4_key_5_6"]
    assert (o.bm_increment(1234, 3, 10, k, 0.5).view(np.uint32) == np.array(G["bm_increment_seed1234_stream3_offset10"], dtype=np.uint32)).all()


@pytest.mark.gpu
def test_hip_path_hits_golden(gpu, oracle):
    x, y, z = inputs(oracle)
    k = G["keep"]
    dx, dy, dz = (gpu.DeviceVector.from_host(a) for a in (x, y, z))
    def check(op, got):
        got, want = got[:k], expected(op)
        if op in LIBM:   # fp64-evaluated on both sides: at most 1 ulp where two libms disagree (none observed)
            assert (np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))[~np.isnan(want)] <= 1).all(), op
        else:
            assert ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all(), op
    for op in oracle.V1S0: check(op, dy.v1s0(op).to_float32())
    for op in oracle.V1S1: check(op, dz.v1s1(op, 1.0 / 3.0).to_float32())
    for op in oracle.V2S0: check(op, dx.v2s0(op, dy).to_float32())
    for op in oracle.V2S1: check(op, dx.v2s1(op, dy, 1.0 / 3.0).to_float32())
    for op in oracle.V3S0: check(op, dz.v3s0(op, dx, dy).to_float32())
    r = G["reductions"]
    m = dx.moments()
    assert abs(m.sum / G["n"] - float.fromhex(r["average_x"])) <= 1e-14
    mz = dz.moments()
    assert mz.min == float.fromhex(r["min_z"]) and mz.max == float.fromhex(r["max_z"])
    bm = gpu.BrownianMotionHip(gpu.TimeDiscretization(0.0, 4, 0.25), 1, k, 1234, path_offset=10)
    got = bm.getBrownianIncrement(3, 0).realizations.to_float32()
    assert (got.view(np.uint32) == np.array(G["bm_increment_seed1234_stream3_offset10"], dtype=np.uint32)).all()
