"""Rolled loops (runtime.cpp): a periodic stretch of a large pending graph — the same operations over one vector after
another, each iteration feeding the next — runs as ONE launch of a kernel that loops over the iterations, once that kernel is
compiled; until then (and with the JIT tier off) the segmented launches run.  Both forms must give the same bits, which must
be the oracle's."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LMM_HIP = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "bin", "lmm_hip")


def scan_chain(factory_like, xs, shared, a, b):
    """y_j = x_j + (running sum of x_k·a_k / (1 + shared·0.5))·b_j — a running sum carried from one iteration to the next."""
    run = None
    ys = []
    for j, x in enumerate(xs):
        t = x.v1s1("MULT_S", a[j]).v2s1("DISCOUNT", shared, 0.5)
        run = t if run is None else run.v2s0("ADD", t)
        ys.append(x.v2s1("ADDPRODUCT_VS", run, b[j]))
    return ys


def scan_chain_oracle(o, xs, shared, a, b):
    run, ys = None, []
    for j, x in enumerate(xs):
        t = o.f_v2s1("DISCOUNT", o.f_v1s1("MULT_S", x, a[j]), shared, 0.5)
        run = t if run is None else o.f_v2s0("ADD", run, t)
        ys.append(o.f_v2s1("ADDPRODUCT_VS", x, run, b[j]))
    return ys


@pytest.mark.parametrize("n,iterations,members", [(30011, 70, 1), (4096, 90, 3), (1, 64, 2)])
def test_rolled_loop_equals_segments_and_oracle(gpu, oracle, n, iterations, members):
    rng = np.random.default_rng(n + iterations)
    hosts = [[oracle.f_from_double(rng.uniform(0.5, 1.5, n)) for _ in range(iterations)] for _ in range(members)]
    shared_h = oracle.f_from_double(rng.uniform(0.0, 1.0, n))
    a = [[0.3 + 0.01 * j + 0.1 * m for j in range(iterations)] for m in range(members)]
    b = [[1.0 - 0.005 * j for j in range(iterations)] for m in range(members)]
    want = [scan_chain_oracle(oracle, hosts[m], shared_h, a[m], b[m]) for m in range(members)]
    prev_fusion = gpu.set_fusion(True)
    results, launches = {}, {}
    try:
        shared = gpu.DeviceVector.from_host(shared_h)
        dev = [[gpu.DeviceVector.from_host(x) for x in row] for row in hosts]
        for mode, name in ((gpu.JIT_OFF, "segments"), (gpu.JIT_SYNC, "discovery"), (gpu.JIT_SYNC, "rolled")):
            prev_jit = gpu.set_jit(mode)
            if name == "segments":
                gpu.purge()                                           # forget plans made with another tier setting
            try:
                with gpu.holding():
                    ys = [scan_chain(None, dev[m], shared, a[m], b[m]) for m in range(members)]
                before = gpu.pool_stats().n_kernel_launches
                gpu.flush()
                launches[name] = gpu.pool_stats().n_kernel_launches - before
                results[name] = [[y.to_float32() for y in row] for row in ys]
                del ys
            finally:
                gpu.set_jit(prev_jit)
    finally:
        gpu.set_fusion(prev_fusion)
    for name, res in results.items():
        for m in range(members):
            for j in range(iterations):
                assert_bits_equal(res[m][j], want[m][j], f"{name}: member {m}, iteration {j}")
    # the stretch of ≈ iterations·4 operations took a dozen launches as segments and takes a handful rolled
    assert launches["rolled"] < launches["segments"] and launches["rolled"] <= 6, launches


def run(*args, env=None):
    out = subprocess.run([LMM_HIP, *map(str, args)], capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_lmm_objective_rolled_equals_segmented():
    """The Euler steps of the LMM simulation as rolled loops (FMHIP_JIT=sync: compiled before first use) against the segmented
    launches (FMHIP_ROLL=0): the 144 model volatilities must be the same doubles.  Three batches of 4 evaluations in lock-step:
    the first meets every group shape for the first time (general path, the plan is written down), the others run from the plans."""
    seg = run("--paths", 20000, "--mode", "evaluate", "--evaluations", 12, "--jacobian-batch", 4, env={"FMHIP_JIT": "sync", "FMHIP_ROLL": "0"})
    rol = run("--paths", 20000, "--mode", "evaluate", "--evaluations", 12, "--jacobian-batch", 4, env={"FMHIP_JIT": "sync"})
    assert seg["model_volatility"] == rol["model_volatility"]
    assert rol["kernel_launches"] < 0.7 * seg["kernel_launches"]


def test_monte_carlo_time_loop_rolls(gpu):
    """A Monte-Carlo TIME loop is periodic too: recorded under fusion hold, the 200 Euler steps of the Heston simulation of
    BASELINE configs[2] (state carried from step to step, two Brownian increments read per step) run as ONE rolled-loop launch —
    and give the price of the step-by-step execution bit for bit."""
    import importlib
    mc = importlib.import_module("finmath-lib-cuda-extensions_amd.montecarlo")
    n, steps = 50_000, 200
    bm = gpu.BrownianMotionHip(gpu.TimeDiscretization(0.0, steps, 0.01), 2, n, 31415)
    bm.getBrownianIncrement(0, 0)
    prev_fusion = gpu.set_fusion(True)
    prev_jit = gpu.set_jit(gpu.JIT_SYNC)
    try:
        for xi in (0.0, 0.3):
            step_by_step, _ = mc.heston_call_mc(bm, 1.0, 0.05, 0.09, 1.0, 0.09, xi, -0.5, 2.0, 1.05)
            for attempt in range(2):                                  # the first held run meets the graph (plan, kernel), the second runs rolled
                before = gpu.pool_stats().n_kernel_launches
                with gpu.holding():
                    held, _ = mc.heston_call_mc(bm, 1.0, 0.05, 0.09, 1.0, 0.09, xi, -0.5, 2.0, 1.05)
                launches = gpu.pool_stats().n_kernel_launches - before
                assert held == step_by_step, (xi, attempt)
            assert launches <= 6, (xi, launches)
        assert abs(step_by_step - 0.1847) < 0.01                     # ξ = 0.3 price of the configuration, Monte-Carlo error of 50 k paths
    finally:
        gpu.set_jit(prev_jit)
        gpu.set_fusion(prev_fusion)


def swaption_like_chain(x_of, periods, numeraire, strike, delta):
    """SwaptionSimple's backward induction (host/lmm.hpp: swaptionValue): a short head (the last period has no running value yet), a
    periodic stretch, a short tail (floor at 0, division by the numeraire)."""
    value = None
    for p in range(periods - 1, -1, -1):
        libor = x_of(p)
        payoff = libor.v1s1("SUB_S", strike).v1s1("MULT_S", delta)
        value = (payoff if value is None else value.v2s0("ADD", payoff)).v2s1("DISCOUNT", libor, delta)
    return value.v1s1("FLOOR_S", 0.0).v2s0("DIV", numeraire)


def test_head_loop_and_tail_in_one_launch(gpu, oracle):
    """The PEELED form of a rolled component (runtime.cpp: plan_peel; jit.hpp: RolledBody::Peel): when the operations in front of the
    loop and behind it are few, a launch of few workgroups runs the whole component — head, loop, tail — as ONE kernel; the values
    between the parts stay in registers.  Same bits as the segmented launches (interpreter tier) and as the oracle."""
    n, periods = 50_021, 40
    rng = np.random.default_rng(77)
    libors = [oracle.f_from_double(rng.uniform(0.005, 0.04, n)) for _ in range(periods)]
    num = oracle.f_from_double(rng.uniform(1.0, 1.3, n))
    strike, delta = 0.02, 0.5
    value = None
    for p in range(periods - 1, -1, -1):
        payoff = oracle.f_v1s1("MULT_S", oracle.f_v1s1("SUB_S", libors[p], strike), delta)
        value = oracle.f_v2s1("DISCOUNT", payoff if value is None else oracle.f_v2s0("ADD", value, payoff), libors[p], delta)
    want = oracle.f_v2s0("DIV", oracle.f_v1s1("FLOOR_S", value, 0.0), num)
    prev_fusion = gpu.set_fusion(True)
    try:
        dev = [gpu.DeviceVector.from_host(x) for x in libors]
        dnum = gpu.DeviceVector.from_host(num)
        launches = {}
        for mode, name in ((gpu.JIT_OFF, "segments"), (gpu.JIT_SYNC, "discovery"), (gpu.JIT_SYNC, "peeled"), (gpu.JIT_SYNC, "peeled again")):
            prev_jit = gpu.set_jit(mode)
            if name == "segments":
                gpu.purge()
            try:
                with gpu.holding():
                    got = swaption_like_chain(lambda p: dev[p], periods, dnum, strike, delta)
                before = gpu.pool_stats().n_kernel_launches
                gpu.flush()
                launches[name] = gpu.pool_stats().n_kernel_launches - before
                assert_bits_equal(got.to_float32(), want, name)
            finally:
                gpu.set_jit(prev_jit)
        assert launches["segments"] >= 4 and launches["peeled"] == 1 and launches["peeled again"] == 1, launches
        # … and as rows of one launch for several chains at once, their expectations included
        with gpu.holding():
            chains = [swaption_like_chain(lambda p, k=k: dev[(p + k) % periods], periods, dnum, strike + 0.001 * k, delta) for k in range(3)]
        before = gpu.pool_stats().n_kernel_launches
        gpu.flush()
        assert gpu.pool_stats().n_kernel_launches - before == 1
        assert np.isfinite([c.moments().sum for c in chains]).all()
    finally:
        gpu.set_fusion(prev_fusion)


@pytest.mark.parametrize("n", [50_021, 1_000_000, 2049])
def test_expectation_in_the_launch_of_the_peeled_chain(gpu, oracle, n):
    """`chain.getAverage()` on a pending component with a peeled form: the launch that computes the chain also takes its moments (jit.hpp:
    RolledBody::Peel::reduce) — ONE launch per product for a caller that values one after the other.  A workgroup of that kernel holds
    one unit of the reduction tree (fm_kernel_parts.hpp), the stand-alone reduction a span of four: the tree is the same, so the moments
    are the same to the last bit, shifted (getVariance's second pass) or not; and so is the stored value."""
    periods = 24
    rng = np.random.default_rng(n)
    libors = [oracle.f_from_double(rng.uniform(0.005, 0.04, n)) for _ in range(periods)]
    num = oracle.f_from_double(rng.uniform(1.0, 1.3, n))
    strike, delta = 0.02, 0.5
    prev_fusion, prev_jit = gpu.set_fusion(True), gpu.set_jit(gpu.JIT_SYNC)
    try:
        dev = [gpu.DeviceVector.from_host(x) for x in libors]
        dnum = gpu.DeviceVector.from_host(num)

        def chain():
            with gpu.holding():
                return swaption_like_chain(lambda p: dev[p], periods, dnum, strike, delta)
        first = chain()
        gpu.flush()                                       # discovery: the plan and its kernels exist from here on
        plain = chain()
        gpu.flush()
        want = plain.moments()                            # the stand-alone reduction of the materialised value
        want_shifted = plain.moments(shift=want.sum / n)
        for shift, ref in ((0.0, want), (want.sum / n, want_shifted)):
            c = chain()
            before = gpu.pool_stats().n_kernel_launches
            got = c.moments(shift=shift)
            assert gpu.pool_stats().n_kernel_launches - before == 1, "chain and expectation are one launch"
            for f in ("sum", "sumsq", "min", "max"):
                assert np.float64(getattr(got, f)).tobytes() == np.float64(getattr(ref, f)).tobytes(), (f, shift, getattr(got, f), getattr(ref, f))
            assert_bits_equal(c.to_float32(), plain.to_float32(), "the value the fused launch stored")
        del first
    finally:
        gpu.set_jit(prev_jit)
        gpu.set_fusion(prev_fusion)


def test_one_expectation_asked_many_pending(gpu, oracle):
    """A caller records the payoffs of many products and then takes their averages one by one (what an optimiser over calibration products
    does).  At the first getAverage() the engine runs everything pending — components of equal shape as rows of the same launches — and
    those launches take the moments of their roots along (runtime.cpp: Engine::reduce, want_root_moments_); the other averages are
    answered from what was left with the nodes: no further launch.  Same bits as asking one at a time; nothing of the kind under a
    caller's own hold; a vector that is written into afterwards forgets its moments."""
    n, periods, products = 40_009, 24, 12
    rng = np.random.default_rng(31)
    libors = [oracle.f_from_double(rng.uniform(0.005, 0.04, n)) for _ in range(periods + products)]
    num = oracle.f_from_double(rng.uniform(1.0, 1.3, n))
    prev_fusion, prev_jit = gpu.set_fusion(True), gpu.set_jit(gpu.JIT_SYNC)
    try:
        dev = [gpu.DeviceVector.from_host(x) for x in libors]
        dnum = gpu.DeviceVector.from_host(num)

        def record(k, short=False):
            return swaption_like_chain(lambda p: dev[p + k], 3 if short else periods, dnum, 0.02 + 0.001 * k, 0.5)
        # reference: one product at a time
        want = []
        for k in range(products):
            for _ in range(2 if k == 0 else 1):              # (the first one twice: discovery, then the kernels)
                c = record(k)
                m = c.moments()
            want.append((m.sum, m.sumsq, m.min, m.max, c.to_float32()))
        short_want = []
        for k in range(products):
            c = record(k, short=True); m = c.moments(); short_want.append((m.sum, m.sumsq, m.min, m.max))
        # all recorded first (long chains: loop kernels; short ones: one launch each), then asked one by one
        for attempt in range(2):
            soft = gpu.fusion_hold(2)                        # (a soft hold: chains stay pending as they do while the engine groups time steps)
            chains = [record(k) for k in range(products)] + [record(k, short=True) for k in range(products)]
            gpu.fusion_hold(soft)
            before = gpu.pool_stats().n_kernel_launches
            first = chains[0].moments()
            after_first = gpu.pool_stats().n_kernel_launches
            got = [first] + [c.moments() for c in chains[1:]]
            after_all = gpu.pool_stats().n_kernel_launches
            for k in range(products):
                assert (got[k].sum, got[k].sumsq, got[k].min, got[k].max) == want[k][:4], (attempt, k)
                assert_bits_equal(chains[k].to_float32(), want[k][4], "the stored value")
                g = got[products + k]
                assert (g.sum, g.sumsq, g.min, g.max) == short_want[k], (attempt, "short", k)
            if attempt == 1:                                 # (attempt 0 meets the batched shapes for the first time)
                assert after_first - before <= 4, "everything pending ran as rows of a few launches"
                assert after_all == after_first, "the other expectations came from the nodes"
        # a value somebody writes into forgets what was known about it
        assert chains[1].moments().sum == want[1][0]
        p = gpu.Program(1)
        v = p.op("MULT_S", 0, s=2.0)
        p.output(v)
        prog = p.compile()
        prog.run_into([[dev[0]]], [[chains[1]]], want_moments=False)
        m = chains[1].moments()
        twice = dev[0].v1s1("MULT_S", 2.0).moments()
        assert (m.sum, m.min, m.max) == (twice.sum, twice.min, twice.max)
        # under the caller's own hold only what is asked for runs
        with gpu.holding():
            held = [record(k) for k in range(products)]
            before = gpu.pool_stats().n_kernel_launches
            m0 = held[0].moments()
            assert gpu.pool_stats().n_kernel_launches - before == 1
            assert (m0.sum, m0.sumsq) == want[0][:2]
        gpu.flush()
    finally:
        gpu.set_jit(prev_jit)
        gpu.set_fusion(prev_fusion)


def test_expectations_taken_while_the_caller_is_still_recording(gpu, oracle):
    """Behind a simulation (the engine holds the methods of the time steps it is grouping) a caller records payoff after payoff without
    asking for anything: after 5000 methods without a new time step the engine runs what is pending WITHOUT waiting, and those launches
    write the moments of their roots into slots of a pinned arena (runtime.cpp: call, arena_alloc, slot_wait).  Asked later, every
    expectation has the bits of the stand-alone reduction — also when more roots were outstanding than the arena has slots (it is
    collected and reused), and for vectors released or overwritten in between."""
    n, count = 67, 530_000                               # (more payoffs than the arena has slots: 2^19)
    rng = np.random.default_rng(5)
    base = [oracle.f_from_double(rng.uniform(0.5, 1.5, n)) for _ in range(7)]
    prev_fusion, prev_jit = gpu.set_fusion(True), gpu.set_jit(gpu.JIT_SYNC)
    try:
        bm = gpu.BrownianMotionHip(gpu.TimeDiscretization(0.0, 8, 0.25), 1, n, 99)
        dev = [gpu.DeviceVector.from_host(x) for x in base]
        state = dev[0]
        for step in range(3):                                 # "the simulation": three time steps of a scheme; the engine's hold is on afterwards
            state = state.v2s1("ADDPRODUCT_VS", bm.getBrownianIncrement(step, 0).realizations, 0.1)
        # one-method payoffs, more than the arena has slots, each with a handle; nothing is asked for
        launches_start = gpu.pool_stats().n_kernel_launches
        payoffs = [dev[k % 7].v1s1("MULT_S", 1.0 + 1e-6 * k) for k in range(count)]
        launches_before = gpu.pool_stats().n_kernel_launches
        assert launches_before - launches_start >= 7, "the engine ran the pending payoffs on its own while they were being recorded"
        sample = list(range(0, count, 9_973)) + [count - 1, 5_001, (1 << 19) - 1, 1 << 19, (1 << 19) + 1]
        del payoffs[123]                                      # a vector released before anybody asks
        sample = [k if k < 123 else k - 1 for k in sample if k != 123]
        for k in sample:
            m = payoffs[k].moments()
            kk = k if k < 123 else k + 1
            want = dev[kk % 7].v1s1("MULT_S", 1.0 + 1e-6 * kk)
            gpu.flush()
            w = want.moments()
            assert (m.sum, m.sumsq, m.min, m.max) == (w.sum, w.sumsq, w.min, w.max), k
        assert gpu.pool_stats().n_kernel_launches > launches_before
    finally:
        gpu.fusion_hold(0)
        gpu.flush()
        gpu.set_jit(prev_jit)
        gpu.set_fusion(prev_fusion)


def test_values_given_up_are_not_stored_and_their_moments_are_the_same(gpu, oracle):
    """fmhip_vec_give_up_values: a caller that wants only the EXPECTATIONS of its products' payoffs (getValue() = getAverage()) says so; the
    launch that computes a payoff then takes its moments and stores nothing — long chains through the peeled kernel whose root is not an
    output, short ones through a program without outputs.  Same moments to the last bit as the stored-and-reduced value; reading a
    given-up value is an error, asking for its moments again is not; a vector somebody else still uses is kept as it is."""
    n, periods, products = 30_011, 24, 10
    rng = np.random.default_rng(77)
    libors = [oracle.f_from_double(rng.uniform(0.005, 0.04, n)) for _ in range(periods + products)]
    num = oracle.f_from_double(rng.uniform(1.0, 1.3, n))
    prev_fusion, prev_jit = gpu.set_fusion(True), gpu.set_jit(gpu.JIT_SYNC)
    try:
        dev = [gpu.DeviceVector.from_host(x) for x in libors]
        dnum = gpu.DeviceVector.from_host(num)

        def record(k, short=False):
            return swaption_like_chain(lambda p: dev[p + k], 3 if short else periods, dnum, 0.02 + 0.001 * k, 0.5)

        def all_chains():
            soft = gpu.fusion_hold(2)
            chains = [record(k) for k in range(products)] + [record(k, short=True) for k in range(products)]
            gpu.fusion_hold(soft)
            return chains
        want = None
        for attempt in range(3):                              # reference: stored, then their moments (attempt 0 meets the shapes for the first time)
            chains = all_chains()
            ticket = gpu.reduce_moments_batch_begin(chains)
            want = [(m.sum, m.sumsq, m.min, m.max) for m in gpu.reduce_moments_batch_end(ticket, len(chains))]
            stored = [c.to_float32() for c in chains]
        assert all(np.isfinite(w[0]) for w in want)
        for attempt in range(3):                              # the same, values given up
            chains = all_chains()
            kept = chains[3].v1s1("MULT_S", 2.0)              # somebody else uses chain 3: it is NOT given up in effect
            gpu.give_up_values(chains)
            ticket = gpu.reduce_moments_batch_begin(chains)
            got = [(m.sum, m.sumsq, m.min, m.max) for m in gpu.reduce_moments_batch_end(ticket, len(chains))]
            assert [np.array(g).tobytes() for g in got] == [np.array(w).tobytes() for w in want], attempt
            assert_bits_equal(kept.to_float32(), oracle.f_v1s1("MULT_S", stored[3], 2.0), "a consumer of a payoff that was not given up in effect")
            assert_bits_equal(chains[3].to_float32(), stored[3], "… which is still there")
            m = chains[5].moments()                           # asked again, one at a time: from the node
            assert (m.sum, m.sumsq, m.min, m.max) == want[5]
            if attempt == 2:                                  # the shapes' kernels exist: nothing was stored for the given-up payoffs …
                gone = 0
                for k, c in enumerate(chains):
                    if k == 3: continue
                    try:
                        values = c.to_float32()               # … so reading one is an error (or, had the engine kept it, the right value)
                        assert_bits_equal(values, stored[k], "a kept value")
                    except gpu.FmhipError as e:
                        assert e.code == gpu._native.ERR_INVALID_ARGUMENT and "given up" in str(e)
                        gone += 1
                assert gone >= products, "the long chains' payoffs, at least, were not stored"
                with pytest.raises(gpu.FmhipError):
                    chains[0].v1s1("ADD_S", 1.0)              # nor can it be an operand
            del chains, kept
        gpu.flush()
    finally:
        gpu.set_jit(prev_jit)
        gpu.set_fusion(prev_fusion)


def test_moments_into_a_device_buffer_come_from_the_launches_too(gpu, oracle):
    """fmhip_reduce_moments_batch_device on PENDING vectors (the send buffer of a caller's own RCCL exchange: lmm_hip --world N): the flush
    that computes them takes their moments along, given-up values are not stored, and a one-wave kernel collects the 32-byte blocks from the
    pinned arena into the caller's buffer in the order asked — the same bits as the host-side ticket, for pending vectors, vectors
    computed earlier (a reduction launch into the arena) and vectors whose moments are known already, mixed in one call."""
    import ctypes as C
    n, periods, products = 30_011, 24, 10
    rng = np.random.default_rng(78)
    libors = [oracle.f_from_double(rng.uniform(0.005, 0.04, n)) for _ in range(periods + products)]
    num = oracle.f_from_double(rng.uniform(1.0, 1.3, n))
    prev_fusion, prev_jit = gpu.set_fusion(True), gpu.set_jit(gpu.JIT_SYNC)
    try:
        dev = [gpu.DeviceVector.from_host(x) for x in libors]
        dnum = gpu.DeviceVector.from_host(num)

        def all_chains():
            soft = gpu.fusion_hold(2)
            chains = [swaption_like_chain(lambda p: dev[p + k], 3 if k % 2 else periods, dnum, 0.02 + 0.001 * k, 0.5) for k in range(products)]
            gpu.fusion_hold(soft)
            return chains
        for attempt in range(3):
            chains = all_chains()
            ticket = gpu.reduce_moments_batch_begin(chains)
            want = np.array([(m.sum, m.sumsq, m.min, m.max) for m in gpu.reduce_moments_batch_end(ticket, len(chains))])
            del chains
            chains = all_chains()
            early = dev[0].v1s1("MULT_S", 3.0); early.to_float32()         # computed earlier, no moments yet
            known = dev[1].v1s1("ADD_S", 1.0); m_known = known.moments()    # moments known already
            m_early = None
            gpu.give_up_values(chains[:6])                                  # some given up, some not
            asked = chains[:5] + [early, known] + chains[5:]
            k = len(asked)
            handles = (C.c_int64 * k)(*[w.handle for w in asked])
            out = gpu.DeviceVector.filled(8 * k, 0.0)
            gpu._native.check(gpu.lib().fmhip_reduce_moments_batch_device(handles, k, None, C.c_void_p(out.device_ptr())))
            raw = out.to_float32().view(np.float64).reshape(k, 4)
            got = np.concatenate([raw[:5], raw[7:]])
            assert got.tobytes() == want.tobytes(), attempt
            m_early = early.moments()
            assert raw[5].tobytes() == np.array([m_early.sum, m_early.sumsq, m_early.min, m_early.max]).tobytes()
            assert raw[6].tobytes() == np.array([m_known.sum, m_known.sumsq, m_known.min, m_known.max]).tobytes()
            m = chains[2].moments()                                         # asked again on the host: from the node's slot
            assert np.array([m.sum, m.sumsq, m.min, m.max]).tobytes() == want[2].tobytes()
            del chains, asked, early, known
        gpu.flush()
    finally:
        gpu.set_jit(prev_jit)
        gpu.set_fusion(prev_fusion)
