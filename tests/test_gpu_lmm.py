"""the native LMM Monte-Carlo calibration driver (host/lmm.hpp) on the MI355X engine vs the same
driver on the CPU twin — the whole simulation (Euler LMM, 80 forward rates, spot measure) and all 144 swaption valuations
must agree; a short Levenberg–Marquardt run must reduce the calibration error towards the reference's acceptance
threshold |mean deviation| < 2e-4 (LIBORMarketModelCalibrationATMTest.java:466)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LMM_HIP = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "bin", "lmm_hip")
LMM_CPU = os.path.join(ROOT, "oracle", "host", "lmm_cpu")


def ensure_built():
    """The driver binaries are build products (csrc/Makefile, oracle/Makefile): build them if this tree has not been built."""
    if not os.path.exists(LMM_HIP):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "csrc")], stdout=subprocess.DEVNULL)
    if not os.path.exists(LMM_CPU):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def run(binary, *args):
    ensure_built()
    out = subprocess.run([binary, *map(str, args)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_objective_evaluation_identical_to_cpu_twin():
    cpu = run(LMM_CPU, "--paths", 5000, "--mode", "evaluate")
    hip = run(LMM_HIP, "--paths", 5000, "--mode", "evaluate")
    a, b = np.array(cpu["model_volatility"]), np.array(hip["model_volatility"])
    assert a.shape == b.shape == (144,)
    # every op of this model is bit-exact arithmetic (+ - * / max); only the fp64 summation order of getAverage differs
    assert np.max(np.abs(a - b) / a) <= 1e-12
    assert hip["launches_simulation"] < 800 and hip["launches_valuation"] < 400      # fused: ~30 k method calls per evaluation
    assert 0.004 < a.mean() < 0.006                                                   # 0.5 % initial LIBOR volatility


def test_short_calibration_converges_and_matches_cpu_twin():
    cpu = run(LMM_CPU, "--paths", 2000, "--mode", "calibrate", "--max-iterations", 2)
    hip = run(LMM_HIP, "--paths", 2000, "--mode", "calibrate", "--max-iterations", 2)
    assert hip["active_parameters"] == 50 and hip["swaptions"] == 144
    assert hip["rms_deviation"] < 0.1 * hip["initial_rms"]
    assert abs(hip["mean_deviation"]) < 2e-4                                           # ...ATMTest.java:466
    # same optimiser path on both back ends
    assert abs(hip["rms_deviation"] - cpu["rms_deviation"]) <= 1e-9
    assert np.allclose(hip["parameters"], cpu["parameters"], rtol=0, atol=1e-9)


def test_reduce_moments_batch(gpu, oracle):
    n, k = 10007, 37
    xs = [oracle.f_from_double(oracle.java_random_doubles(500 + i, n) - 0.3) for i in range(k)]
    vecs = [gpu.DeviceVector.from_host(x) for x in xs]
    handles = (C.c_int64 * k)(*[v.handle for v in vecs])
    shifts = (C.c_double * k)(*[0.01 * i for i in range(k)])
    out = (gpu.Moments * k)()
    before = gpu.pool_stats().n_kernel_launches
    gpu._native.check(gpu.lib().fmhip_reduce_moments_batch(handles, k, shifts, out))
    assert gpu.pool_stats().n_kernel_launches - before == 1          # one batched launch (the final combine is fused into it)
    for i in range(k):
        m = vecs[i].moments(0.01 * i)
        assert (out[i].sum, out[i].sumsq, out[i].min, out[i].max) == (m.sum, m.sumsq, m.min, m.max)


def test_sharded_driver_path_with_rccl_world_of_one(tmp_path):
    """The path-sharded driver (BASELINE.json configs[4]) with a world of ONE rank: RCCL communicator bootstrap through the
    id file (nonce-checked), device-side batched expectation partials, ncclAllGather on the runtime stream.  The result must equal the
    unsharded run; and a 'rank 1 of 2'-style path offset must equal --path-offset (shard invariance of the generator)."""
    plain = run(LMM_HIP, "--paths", 4000, "--mode", "evaluate")
    (tmp_path / "id").write_bytes(b"stale id file of an earlier launch" * 8)          # rank 0 must replace it, not trip over it
    dist = run(LMM_HIP, "--paths", 4000, "--mode", "evaluate", "--world", 1, "--rank", 0, "--nccl-id-file", tmp_path / "id", "--nccl-nonce", 12345)
    assert dist["rccl_collectives"] >= 1 and dist["world"] == 1 and dist["rccl_collective_seconds"] > 0
    assert not (tmp_path / "id").exists()                                              # removed once the communicator exists
    a, b = np.array(plain["model_volatility"]), np.array(dist["model_volatility"])
    assert np.max(np.abs(a - b) / a) <= 1e-12
    # … and a calibration: the Jacobian batches in flight one behind the other (reduce → all-gather → pinned copy → event per parameter
    # set, read when the next sets are enqueued): the same optimiser path as the unsharded run
    plain = run(LMM_HIP, "--paths", 4000, "--mode", "calibrate", "--max-iterations", 2, "--jacobian-batch", 4)
    dist = run(LMM_HIP, "--paths", 4000, "--mode", "calibrate", "--max-iterations", 2, "--jacobian-batch", 4, "--world", 1, "--rank", 0, "--nccl-id-file", tmp_path / "id2", "--nccl-nonce", 777)
    assert dist["rccl_collectives"] == dist["evaluations"] and dist["parameters"] == plain["parameters"] and dist["evaluations"] == plain["evaluations"]


def test_lock_step_jacobian_batches_give_the_same_calibration():
    """Finite-difference bumps simulated in lock-step (rows of the same launches) change the launch count, not one bit of
    the result."""
    one = run(LMM_HIP, "--paths", 3000, "--mode", "calibrate", "--max-iterations", 2, "--jacobian-batch", 1)
    many = run(LMM_HIP, "--paths", 3000, "--mode", "calibrate", "--max-iterations", 2, "--jacobian-batch", 7)
    assert one["parameters"] == many["parameters"] and one["rms_deviation"] == many["rms_deviation"]
    assert one["evaluations"] == many["evaluations"]
    assert many["kernel_launches"] < 0.5 * one["kernel_launches"]


def test_mersenne_brownian_motion_through_the_factory():
    """--brownian mersenne: finmath's CPU generator (the one the reference's test injects, …ATMTest.java:283) creates its
    increments through the back end's factory — the HIP engine and the CPU twin then see the very same numbers and the
    objective evaluation agrees like with the device generator."""
    cpu = run(LMM_CPU, "--paths", 4000, "--mode", "evaluate", "--brownian", "mersenne")
    hip = run(LMM_HIP, "--paths", 4000, "--mode", "evaluate", "--brownian", "mersenne")
    philox = run(LMM_HIP, "--paths", 4000, "--mode", "evaluate")
    a, b, c = np.array(cpu["model_volatility"]), np.array(hip["model_volatility"]), np.array(philox["model_volatility"])
    assert np.max(np.abs(a - b) / a) <= 1e-12
    assert np.max(np.abs(b - c) / c) > 1e-6 and np.max(np.abs(b - c) / c) < 0.2       # another stream of random numbers, same model


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3] at FULL size: 1 M paths on one MI355X (80 x 80 x 4 MB of LIBOR state per parameter set).


def test_full_size_objective_evaluation_equals_its_two_path_shards_and_the_cpu_twin():
    """1 M-path objective evaluation (80-step simulation + 144 Monte-Carlo valuations).
    (i) size-independent property: every expectation is linear in the paths, and the counter-based generator makes the
        paths [0, 500k) and [500k, 1M) of the big run identical to two shard runs (--path-offset) — so each of the 144 model
        volatilities (= expectation x a constant) must equal the mean of the two shard results to fp64 summation order;
    (ii) the same evaluation on the CPU twin (oracle/host/lmm_cpu, ≈ 15 s on one core): every op of the model is bit-exact
        fp32 arithmetic, only the fp64 summation order of getAverage differs."""
    full = run(LMM_HIP, "--paths", 1000000, "--mode", "evaluate")
    lo = run(LMM_HIP, "--paths", 500000, "--mode", "evaluate")
    hi = run(LMM_HIP, "--paths", 500000, "--mode", "evaluate", "--path-offset", 500000)
    a = np.array(full["model_volatility"])
    b = 0.5 * (np.array(lo["model_volatility"]) + np.array(hi["model_volatility"]))
    assert a.shape == (144,) and np.all(a > 0)
    assert np.max(np.abs(a - b) / a) <= 1e-12
    assert np.max(np.abs(np.array(lo["model_volatility"]) - np.array(hi["model_volatility"])) / a) > 1e-6      # the shards ARE different paths
    cpu = run(LMM_CPU, "--paths", 1000000, "--mode", "evaluate")
    c = np.array(cpu["model_volatility"])
    assert np.max(np.abs(a - c) / c) <= 1e-12


def test_full_size_calibration_meets_the_reference_acceptance():
    """LIBORMarketModelCalibrationATMTest at 1 M paths (the reference runs 10 000 on the CPU): Levenberg-Marquardt over 50
    volatility parameters, every objective evaluation a fresh 1 M-path simulation; acceptance as the reference asserts it,
    `Math.abs(averageDeviation) < 2E-4` (LIBORMarketModelCalibrationATMTest.java:466)."""
    r = run(LMM_HIP, "--paths", 1000000, "--mode", "calibrate", "--max-iterations", 12)
    assert r["paths"] == 1000000 and r["swaptions"] == 144 and r["active_parameters"] == 50
    assert abs(r["mean_deviation"]) < 2e-4
    assert r["rms_deviation"] < 0.05 * r["initial_rms"]                 # the fit, not just the mean, has converged
    assert r["evaluations"] >= 50 * r["iterations"]                      # finite-difference Jacobian: one re-simulation per parameter
    assert r["seconds"] < 60.0                                           # the CPU twin needs ≈ 15 s per EVALUATION at this size


def test_caller_without_hints_gets_the_same_numbers():
    """lmm_hip --finmath-like: the driver as code that knows nothing about the engine (no hold / flush / replication / lock-step
    batches, every state kept, one getAverage per product — finmath-lib's Euler scheme and optimizer through the Java interface).
    Same model volatilities as the driver with all its hints — with the engine grouping the time steps on the caller's behalf
    (fmhip_set_step_grouping, default 4; FMHIP_GROUP_STEPS=0 switches it off) and without, the grouped run in fewer launches."""
    ensure_built()
    def evaluate(*extra, env=None):
        out = subprocess.run([LMM_HIP, "--paths", "20000", "--mode", "evaluate", "--evaluations", "2", *map(str, extra)], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, **(env or {})))
        assert out.returncode == 0, out.stderr
        return json.loads(out.stdout.strip().splitlines()[-1])
    hinted = evaluate()
    plain = evaluate("--finmath-like", env={"FMHIP_GROUP_STEPS": "0"})
    grouped = evaluate("--finmath-like")
    assert plain["model_volatility"] == hinted["model_volatility"] == grouped["model_volatility"]
    # … and with the expectations of all pending products taken by the flushes that compute them (while the caller is still recording:
    # FMHIP_SPECULATE_PENDING; when it first asks: FMHIP_BATCH_EXPECTATIONS) against one launch per product
    one_by_one = evaluate("--finmath-like", env={"FMHIP_SPECULATE_PENDING": "0", "FMHIP_BATCH_EXPECTATIONS": "0"})
    at_first_ask = evaluate("--finmath-like", env={"FMHIP_SPECULATE_PENDING": "0"})
    assert one_by_one["model_volatility"] == at_first_ask["model_volatility"] == grouped["model_volatility"]
    if os.environ.get("FMHIP_JIT", "") != "off":              # (loop kernels exist on the specialised tier only; how many launches a run takes
        for run in (grouped, at_first_ask, one_by_one):      #  also depends on which kernels are compiled yet)
            assert run["kernel_launches"] < 0.6 * plain["kernel_launches"]
        assert hinted["kernel_launches"] < 0.6 * min(grouped["kernel_launches"], at_first_ask["kernel_launches"])


def test_row_table_ring_smaller_than_a_rolled_row_table():
    """FMHIP_RING_BYTES accepts values down to 4 KB; the row table of a rolled launch (iterations x pointers, ≈ 5 KB for the LMM's Euler
    steps) does not fit such a ring, batched launches need several trips through it: rolled stretches fall back to their segments or
    take fewer members per launch, ordinary launches split their batch — same numbers, no error."""
    ensure_built()
    def evaluate(env):
        out = subprocess.run([LMM_HIP, "--paths", "20000", "--mode", "evaluate", "--evaluations", "8", "--jacobian-batch", "8"], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, FMHIP_JIT="sync", **env))
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    roomy, tight = evaluate({}), evaluate({"FMHIP_RING_BYTES": "4096"})
    assert tight["model_volatility"] == roomy["model_volatility"]
    assert tight["kernel_launches"] > roomy["kernel_launches"]


def test_brownian_motion_groups_time_steps_for_a_plain_scheme(gpu):
    """Time-step grouping by the engine (fmhip_set_step_grouping; BrownianMotionHip.setGroupSteps forwards to it): an Euler scheme
    written against the interfaces only (montecarlo.py) runs in groups of time steps — same price bit for bit, fewer launches."""
    import importlib
    mc = importlib.import_module("finmath-lib-cuda-extensions_amd.montecarlo")
    bm = gpu.BrownianMotionHip(gpu.TimeDiscretization(0.0, 100, 0.02), 2, 100_000, 31415)
    bm.getBrownianIncrement(0, 0)
    prev = gpu.set_fusion(True)
    try:
        default_steps = gpu.set_step_grouping(0)
        assert default_steps == 4
        before = gpu.pool_stats().n_kernel_launches
        plain, _ = mc.heston_call_mc(bm, 1.0, 0.05, 0.09, 1.0, 0.09, 0.3, -0.5, 2.0, 1.05)
        mid = gpu.pool_stats().n_kernel_launches
        bm.setGroupSteps(10)
        grouped, _ = mc.heston_call_mc(bm, 1.0, 0.05, 0.09, 1.0, 0.09, 0.3, -0.5, 2.0, 1.05)
        after = gpu.pool_stats().n_kernel_launches
        assert gpu.set_step_grouping(default_steps) == 10
        by_default, _ = mc.heston_call_mc(bm, 1.0, 0.05, 0.09, 1.0, 0.09, 0.3, -0.5, 2.0, 1.05)
        assert by_default == plain and gpu.pool_stats().n_kernel_launches - after < mid - before
    finally:
        gpu.set_step_grouping(4)
        gpu.fusion_hold(False)
        gpu.set_fusion(prev)
    assert grouped == plain
    assert after - mid < mid - before
