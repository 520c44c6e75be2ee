"""Merged chains (runtime.cpp: merge_families; jit.cpp: jit_generate_merged_source): components of ONE loop shape whose vectors are the
same sequence — the shorter ones reading a suffix of the longest one's, as the swaptions of one exercise date read the forward rates of
their tenor (SwaptionSimple's backward induction: LIBORMarketModelCalibrationATMTest.java:475-520 builds 14 tenors per exercise date) —
run as ONE launch that loads every vector once.  Per chain it is the same operations on the same operands in the same order and the
same reduction tree: values and moments must be the eager path's bits (which are the oracle's), whatever was merged with what."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

PERIOD = 0.5


def swaption(L, numeraire, periods, swaprate):
    """lmm.hpp: swaptionValue — backward induction over the periods of the swap, floored at 0, numeraire-relative."""
    value = None
    for p in range(periods - 1, -1, -1):
        payoff = L[p].v1s1("SUB_S", swaprate).v1s1("MULT_S", PERIOD)
        value = (payoff if value is None else value.v2s0("ADD", payoff)).v2s1("DISCOUNT", L[p], PERIOD)
    return value.v1s1("FLOOR_S", 0.0).v2s0("DIV", numeraire)


def swaption_oracle(o, L, numeraire, periods, swaprate):
    value = None
    for p in range(periods - 1, -1, -1):
        payoff = o.f_v1s1("MULT_S", o.f_v1s1("SUB_S", L[p], swaprate), PERIOD)
        value = o.f_v2s1("DISCOUNT", payoff if value is None else o.f_v2s0("ADD", value, payoff), L[p], PERIOD)
    return o.f_v2s0("DIV", o.f_v1s1("FLOOR_S", value, 0.0), numeraire)


def moments_tuple(m):
    return (m.sum, m.sumsq, m.min, m.max)


def products(dates, tenors):
    """(date, periods, swap rate) of every product: every date has every tenor that fits its vectors"""
    out = []
    for d, n_vec in enumerate(dates):
        for k, periods in enumerate(tenors):
            if periods <= n_vec:
                out.append((d, periods, 0.01 + 0.002 * k + 0.0005 * d))
    return out


@pytest.mark.parametrize("n,dates,tenors,give_up", [
    (30011, (60, 60, 40), (60, 40, 30, 20, 14, 12, 4), False),       # ragged length; the third date has fewer tenors: a family size of its own
    (4096, (60, 60), (60, 50, 40, 30, 20, 18, 16, 14, 12), True),    # values given up: nothing is stored, the moments are all there is
    (1, (30, 30), (30, 24, 20, 12), False),                          # one path
])
def test_merged_chains_equal_eager_bits(gpu, oracle, n, dates, tenors, give_up):
    rng = np.random.default_rng(n + len(tenors))
    L_h = [[oracle.f_from_double(rng.uniform(-0.01, 0.05, n)) for _ in range(n_vec)] for n_vec in dates]
    num_h = [oracle.f_from_double(rng.uniform(0.9, 1.4, n)) for _ in dates]
    prods = products(dates, tenors)
    # eager: one launch per method, every value stored — the reference bits (and the oracle's, checked on the longest product of date 0)
    prev_fusion = gpu.set_fusion(False)
    try:
        L = [[gpu.DeviceVector.from_host(x) for x in row] for row in L_h]
        num = [gpu.DeviceVector.from_host(x) for x in num_h]
        eager = [swaption(L[d], num[d], periods, rate) for d, periods, rate in prods]
        want_values = [v.to_float32() for v in eager]
        want_moments = [moments_tuple(v.moments()) for v in eager]
        del eager
        assert_bits_equal(want_values[0], swaption_oracle(oracle, L_h[0], num_h[0], prods[0][1], prods[0][2]), "eager vs oracle")
        gpu.set_fusion(True)
        prev_jit = gpu.set_jit(gpu.JIT_SYNC)
        prev_hold = gpu.fusion_hold(2)          # nothing runs on the engine's own accord; an expectation asked with much pending runs everything
        try:
            gpu.purge()
            merged_before = gpu.engine_stats()["merged_launches"]
            for round_ in range(3):             # round 0 plans the shapes (one launch per shape), from round 1 on the families are found
                values = [swaption(L[d], num[d], periods, rate) for d, periods, rate in prods]
                if give_up:
                    gpu.give_up_values(values)
                before = gpu.engine_stats()
                got_moments = [moments_tuple(v.moments()) for v in values]        # the first call runs everything pending, moments taken along
                after = gpu.engine_stats()
                for i, (d, periods, rate) in enumerate(prods):
                    assert got_moments[i] == want_moments[i], f"round {round_}: moments of product {i} (date {d}, {periods} periods): {got_moments[i]} vs {want_moments[i]}"
                    if not give_up:
                        assert_bits_equal(values[i].to_float32(), want_values[i], f"round {round_}: product {i} (date {d}, {periods} periods)")
                if round_ >= 1:
                    assert after["merged_launches"] > before["merged_launches"], "no merged launch from the second occurrence of the shapes on"
                    # every product is a chain of a merged launch — the four-period ones too, which fit a launch of their own (match_small)
                    assert after["merged_chains"] - before["merged_chains"] == len(prods)
                    assert after["merged_launches"] - before["merged_launches"] <= len(set(sum(1 for q in prods if q[0] == d) for d in range(len(dates))))
                    # fewer bytes than one launch per shape: every vector of a date is read once for all its merged tenors
                    assert after["algorithmic_bytes"] - before["algorithmic_bytes"] < first_bytes
                else:
                    first_bytes = after["algorithmic_bytes"] - before["algorithmic_bytes"]
                del values
            assert gpu.engine_stats()["merged_launches"] > merged_before
        finally:
            gpu.fusion_hold(prev_hold)
            gpu.set_jit(prev_jit)
    finally:
        gpu.set_fusion(prev_fusion)


def test_chains_that_do_not_share_their_vectors_are_left_alone(gpu, oracle):
    """Same shapes, but every product reads vectors of its own (or a sequence that is not a suffix of the other's): nothing to merge, same bits."""
    n = 5000
    rng = np.random.default_rng(7)
    tenors = (40, 30, 20)
    L_h = [[oracle.f_from_double(rng.uniform(-0.01, 0.05, n)) for _ in range(t)] for t in tenors]
    num_h = oracle.f_from_double(rng.uniform(0.9, 1.4, n))
    prev_fusion = gpu.set_fusion(False)
    try:
        L = [[gpu.DeviceVector.from_host(x) for x in row] for row in L_h]
        num = gpu.DeviceVector.from_host(num_h)
        # product 2 reads the FIRST twenty vectors of product 0's sequence (a prefix, not a suffix): its last vector differs
        def build():
            return [swaption(L[0], num, 40, 0.02), swaption(L[1], num, 30, 0.021), swaption(L[0][20:], num, 20, 0.022)]
        want = [(v.to_float32(), moments_tuple(v.moments())) for v in build()]
        gpu.set_fusion(True)
        prev_jit = gpu.set_jit(gpu.JIT_SYNC)
        prev_hold = gpu.fusion_hold(2)
        try:
            for round_ in range(2):
                values = build()
                before = gpu.engine_stats()["merged_launches"]
                got = [moments_tuple(v.moments()) for v in values]
                assert gpu.engine_stats()["merged_launches"] == before
                for i, v in enumerate(values):
                    assert got[i] == want[i][1]
                    assert_bits_equal(v.to_float32(), want[i][0], f"round {round_}, product {i}")
                del values
        finally:
            gpu.fusion_hold(prev_hold)
            gpu.set_jit(prev_jit)
    finally:
        gpu.set_fusion(prev_fusion)


@pytest.mark.parametrize("with_moments", [False, True])
def test_common_rows_are_computed_once_and_share_their_vectors(gpu, oracle, with_moments):
    """Rows of a batched launch that read the SAME vectors with the SAME scalars — the bumped parameter sets of a finite-difference
    Jacobian before the time step at which their bump first matters (LIBORMarketModelCalibrationATMTest.java:314-340: one re-simulation
    per parameter) — are computed once (runtime.cpp: run_peeled / merge_families, common rows); the other members receive the same
    vectors.  Storage that is shared is copied before anybody writes into it in place (make_private): the two doors through which a
    vector can be written, fmhip_program_run_into and a raw device pointer, leave the other holders' values alone."""
    n = 20011
    rng = np.random.default_rng(5)
    L_h = [oracle.f_from_double(rng.uniform(-0.01, 0.05, n)) for _ in range(40)]
    num_h = oracle.f_from_double(rng.uniform(0.9, 1.4, n))
    prev_fusion = gpu.set_fusion(False)
    try:
        L = [gpu.DeviceVector.from_host(x) for x in L_h]
        num = gpu.DeviceVector.from_host(num_h)
        # three "parameter sets": two of them identical in everything they read, the third with other strikes; two tenors each
        def build():
            return [swaption(L, num, periods, rate) for rates in ((0.02, 0.021), (0.02, 0.021), (0.03, 0.031)) for periods, rate in zip((40, 24), rates)]
        want = [(v.to_float32(), moments_tuple(v.moments())) for v in build()]
        gpu.set_fusion(True)
        prev_jit = gpu.set_jit(gpu.JIT_SYNC)
        prev_hold = gpu.fusion_hold(2)
        try:
            for round_ in range(3):
                values = build()
                before = gpu.engine_stats()
                if with_moments:
                    got = [moments_tuple(v.moments()) for v in values]         # everything pending runs, moments taken along (merged families from round 1 on)
                    for i in range(len(values)):
                        assert got[i] == want[i][1], (round_, i)
                else:
                    gpu.flush()                                                  # one launch per shape, rows = the three sets
                after = gpu.engine_stats()
                assert after["common_rows"] > before["common_rows"], "the two identical parameter sets were not recognised as common rows"
                for i, v in enumerate(values):
                    assert_bits_equal(v.to_float32(), want[i][0], f"round {round_}, value {i}")
                # values 0 and 2 (sets 0 and 1, 40 periods) are the same numbers and, computed once, the same storage: overwrite one in place
                p = gpu.Program(1)
                p.output(p.op("MULT_S", 0, s=3.0))
                p.compile()
                p.run_into([[values[2]]], [[values[2]]])
                assert_bits_equal(values[2].to_float32(), oracle.f_v1s1("MULT_S", want[2][0], 3.0), "the vector written in place")
                assert_bits_equal(values[0].to_float32(), want[0][0], "its former twin")
                values[1].device_ptr()                                           # a raw pointer handed out: storage of its own from here on
                assert_bits_equal(values[1].to_float32(), want[1][0], "after a device pointer was handed out")
                assert_bits_equal(values[3].to_float32(), want[3][0], "its twin")
                del values, p
        finally:
            gpu.fusion_hold(prev_hold)
            gpu.set_jit(prev_jit)
    finally:
        gpu.set_fusion(prev_fusion)
