"""The two unit-test workloads whose wall times the reference's README quotes (README.md:196-216), run on this
engine — other hardware (the reference used a GeForce GTX 1080), so these are context, not a like-for-like comparison:

  * BrownianMotionTest.java:66-123: 100 x (new Brownian motion with 10 steps x 1 factor x 1M paths, all increments generated,
    mean and variance of the first increment, both checked against the test's bounds)   — reference GPU class: 2.325 s
  * MonteCarloBlackScholesModelTest.java:62-157: European call, 1M paths x 100 Euler steps of dt = 1.0 (maturity 2.0),
    price within 0.005 of the analytic value                                             — reference GPU class: 0.09 s

    python benchmarks/reference_readme_workloads.py"""
import importlib, json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
mc = importlib.import_module("finmath-lib-cuda-extensions_amd.montecarlo")
fm.init(0)
fm.set_fusion(True)


def brownian_motion_test():
    dt, n = 0.1, 1_000_000
    td = fm.TimeDiscretization(0.0, 10, dt)
    for _ in range(2):                                      # warm-up (first-use costs: pool, programs)
        fm.BrownianMotionHip(td, 1, n, 1234).getBrownianIncrement(0, 0).getAverage()
    fm.synchronize()
    t0 = time.perf_counter()
    for i in range(100):
        b = fm.BrownianMotionHip(td, 1, n, 1234 + i)
        x = b.getBrownianIncrement(0, 0)
        mean, var = x.getAverage(), x.getVariance()
        assert abs(mean) < 3.0 * math.sqrt(dt) / math.sqrt(n) * 1.5 and abs(var - dt) < 3.0 * dt / math.sqrt(n) * 1.5
    return time.perf_counter() - t0


def black_scholes_test():
    n = 1_000_000
    td = fm.TimeDiscretization(0.0, 100, 1.0)

    def once(seed):
        bm = fm.BrownianMotionHip(td, 1, n, seed)
        value, _ = mc.black_scholes_call_mc(bm, 1.0, 0.05, 0.30, 2.0, 1.05)
        return value

    def whole_process(seed):
        # as finmath-lib's EulerSchemeFromProcessModel does it: ALL 100 steps of the grid are simulated (and kept), the product reads index 2
        bm = fm.BrownianMotionHip(td, 1, n, seed)
        x = bm.getRandomVariableForConstant(math.log(1.0))
        states = [x]
        for i in range(td.getNumberOfTimeSteps()):
            x = x.add((0.05 - 0.5 * 0.30 * 0.30) * td.getTimeStep(i)).addProduct(bm.getBrownianIncrement(i, 0), 0.30)
            states.append(x)
        value = states[2].exp().sub(1.05).floor(0.0).div(math.exp(0.05 * 2.0)).getAverage()
        fm.flush()                                          # the 98 steps nobody reads are computed too
        fm.synchronize()
        return value

    once(1); whole_process(1)
    fm.synchronize()
    t0 = time.perf_counter()
    v = once(31415)
    dt = time.perf_counter() - t0
    fm.synchronize()
    t0 = time.perf_counter()
    v_all = whole_process(31415)
    dt_all = time.perf_counter() - t0
    analytic = mc.black_scholes_call_analytic(1.0, 0.05, 0.30, 2.0, 1.05)
    assert abs(v - analytic) < 0.005 and v_all == v
    return dt, v, analytic, dt_all


if __name__ == "__main__":
    t_bm = brownian_motion_test()
    t_bs, v, a, t_all = black_scholes_test()
    print(json.dumps({"BrownianMotionTest_100_iterations_s": t_bm, "reference_gpu_s": 2.325,
                      "MonteCarloBlackScholesModelTest_steps_to_maturity_s": t_bs, "MonteCarloBlackScholesModelTest_whole_process_100_steps_s": t_all,
                      "reference_gpu_bs_s": 0.09, "value": v, "analytic": a}))
