#!/usr/bin/env python3
"""BASELINE.json configs[2]: BrownianMotionHip 1M paths x 200 steps x 5 factors (4.0 GB of N(0,dt) increments, seed 31415)
driving a Heston Monte-Carlo (Euler full truncation, factors 0 and 1) on one MI355X.  Prints one JSON line: generation
rate (GB/s written, the kernel is write-bound: 4 B per normal), Heston wall time, price vs the Black-Scholes limit."""
import importlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
mc = importlib.import_module("finmath-lib-cuda-extensions_amd.montecarlo")

fm.init(0)
n, steps, factors, dt = 1_000_000, 200, 5, 0.01
S0, R, T, K = 1.0, 0.05, 2.0, 1.05
td = fm.TimeDiscretization(0.0, steps, dt)
# Generation: the device time of fm_bm_kernel (HIP events around the launch, fmhip_profile_*) is the kernel's figure; the wall time
# around getBrownianIncrement also holds the pool's allocation of the 4 GB slab (first touch when the pool has none to hand back),
# the upload of the step widths and the launch latency.  Round 2 reported only the latter (0.449 of peak where the kernel alone
# runs at 0.73): both are in the line now, named for what they are.
times, kernel_us = [], []
fm.profile_enable(True)
for rep in range(6):
    bm = fm.BrownianMotionHip(td, factors, n, 31415 + rep)
    fm.synchronize(); t0 = time.perf_counter()
    bm.getBrownianIncrement(0, 0)                 # generates all steps x factors vectors in one launch
    fm.synchronize(); times.append(time.perf_counter() - t0)
    ms, count = fm.profile_read()
    kernel_us.append(ms * 1e3 / max(1, count))
    if rep < 5: del bm                            # the slab goes back to the pool and serves the next generation
fm.profile_enable(False)
times, kernel_us = times[1:], kernel_us[1:]       # the first generation allocates the slab
gen_s = sum(kernel_us) / len(kernel_us) * 1e-6
nbytes = 4.0 * n * steps * factors
fm.set_fusion(True)
res = {}
for xi in (0.0, 0.3):
    fm.synchronize(); t0 = time.perf_counter()
    before = fm.pool_stats()
    value, _ = mc.heston_call_mc(bm, S0, R, 0.09, 1.0, 0.09, xi, -0.5, T, K)
    fm.synchronize(); wall = time.perf_counter() - t0
    after = fm.pool_stats()
    res[xi] = {"price": value, "wall_s": wall, "launches": after.n_kernel_launches - before.n_kernel_launches,
               "path_ops_per_s": (after.n_ops_executed - before.n_ops_executed) * n / wall}
# The same simulations recorded under fusion hold: the engine sees the whole time loop at once, finds it periodic in the time
# index (state carried in registers, two increments read per step) and runs it as ONE rolled-loop launch — 4 launches in all
# (first steps, loop, payoff, expectation).  The wall time is then this script's own recording (≈ 2 µs per method in Python).
held = {}
for rep in range(2):                              # the first pass compiles the rolled kernels
    for xi in (0.0, 0.3):
        fm.synchronize(); t0 = time.perf_counter()
        before = fm.pool_stats()
        with fm.holding():
            value, _ = mc.heston_call_mc(bm, S0, R, 0.09, 1.0, 0.09, xi, -0.5, T, K)
        fm.synchronize(); wall = time.perf_counter() - t0
        after = fm.pool_stats()
        held[xi] = {"price": value, "wall_s": wall, "launches": after.n_kernel_launches - before.n_kernel_launches, "same_price": value == res[xi]["price"]}
    fm.jit_wait()                                 # the rolled kernels compile in the background; the measured pass finds them ready
print(json.dumps({
    "workload": "BrownianMotionHip 1M paths x 200 steps x 5 factors + Heston MC (configs[2])",
    "generation": {"bytes": nbytes, "kernel_seconds_avg": gen_s, "kernel_seconds_min": min(kernel_us) * 1e-6, "launches": len(kernel_us),
                   "GBps_written": nbytes / gen_s / 1e9, "frac_of_8TBps": nbytes / gen_s / 8e12, "frac_of_8TBps_best_launch": nbytes / (min(kernel_us) * 1e-6) / 8e12,
                   "normals_per_s": n * steps * factors / gen_s,
                   "wall_seconds_around_the_call_min": min(times), "frac_of_8TBps_by_wall_clock": nbytes / min(times) / 8e12,
                   "timing": "kernel: HIP events around fm_bm_kernel on the runtime stream; wall: host clock around getBrownianIncrement (slab from the pool, step widths uploaded, launch, wait)"},
    "heston_xi0": res[0.0], "heston_xi03": res[0.3], "heston_xi0_time_loop_rolled": held[0.0], "heston_xi03_time_loop_rolled": held[0.3],
    "black_scholes_analytic": mc.black_scholes_call_analytic(S0, R, 0.30, T, K),
    "abs_error_xi0": abs(res[0.0]["price"] - mc.black_scholes_call_analytic(S0, R, 0.30, T, K)),
    "acceptance": "abs error < 0.005 (MonteCarloBlackScholesModelTest.java:156)"}))
