"""Fused programs and the lazy front-end: one launch for a whole chain, bit-identical to eager execution and
to the oracle; horizontal batching; explicit program API; limits and splitting."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


def inputs(oracle, n, k=0):
    x = oracle.f_from_double(oracle.java_random_doubles(31415 + k, n))
    y = oracle.f_from_double(oracle.java_random_doubles(27182 + k, n) + 0.5)
    z = oracle.f_from_double(oracle.java_random_doubles(16180 + k, n) + 0.5)
    return x, y, z


def stream_s_oracle(o, x, y, z):
    """Canonical stream S of SURVEY.md §8(d) config 2 (12 path-ops, 3 inputs, 1 escaping output)."""
    t = o.f_v2s0("SUB", o.f_v2s0("MULT", o.f_v1s1("DIV_S", o.f_v1s1("ADD_S", x, 4.0), 2.0), y), z)
    u = o.f_v1s0("SQRT", o.f_v1s0("ABS", o.f_v1s0("LOG", o.f_v1s0("EXP", t))))
    v = o.f_v3s0("ADDPRODUCT", o.f_v1s1("FLOOR_S", o.f_v1s1("CAP_S", u, 1.5), 0.25), y, z)
    return o.f_v3s0("CHOOSE", t, v, x)


def stream_s_program(gpu):
    p = gpu.Program(3)
    x, y, z = 0, 1, 2
    t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", x, s=4.0), s=2.0), y), z)
    u = p.op("SQRT", p.op("ABS", p.op("LOG", p.op("EXP", t))))
    v = p.op("ADDPRODUCT", p.op("FLOOR_S", p.op("CAP_S", u, s=1.5), s=0.25), y, z)
    w = p.op("CHOOSE", t, v, x)
    p.output(w); p.reduce(w)
    return p.compile()


def libm_equal(got, want):
    assert (np.abs(got.astype(np.float64) - want) <= 1e-7 * (1 + np.abs(want))).all()
    assert (got != want).mean() <= 1e-4


@pytest.mark.parametrize("n", [1, 5, 1000, 100000, 1000000])
def test_stream_s_program_single(gpu, oracle, n):
    x, y, z = inputs(oracle, n)
    want = stream_s_oracle(oracle, x, y, z)
    p = stream_s_program(gpu)
    outs, mom = p.run([[gpu.DeviceVector.from_host(a) for a in (x, y, z)]])
    got = outs[0][0].to_float32()
    libm_equal(got, want)
    wm = oracle.f_moments(got)                                  # reductions of what the device produced
    assert abs(mom[0, 0, 0] - wm[0]) <= 1e-13 * np.abs(got.astype(np.float64)).sum()
    assert abs(mom[0, 0, 1] - wm[1]) <= 1e-13 * (got.astype(np.float64) ** 2).sum()
    assert mom[0, 0, 2] == wm[2] and mom[0, 0, 3] == wm[3]


def test_stream_s_batched_equals_single(gpu, oracle):
    """One launch over 7 independent (x,y,z) triples == 7 single launches (bitwise), and == oracle."""
    n, B = 30011, 7
    p = stream_s_program(gpu)
    rows = [[gpu.DeviceVector.from_host(a) for a in inputs(oracle, n, k)] for k in range(B)]
    before = gpu.pool_stats().n_kernel_launches
    outs, mom = p.run(rows)
    assert gpu.pool_stats().n_kernel_launches - before == 1      # program incl. the fused final combine, for the whole batch
    for k in range(B):
        single, m1 = p.run([rows[k]])
        assert_bits_equal(outs[k][0].to_float32(), single[0][0].to_float32(), f"row {k}")
        assert (mom[k] == m1[0]).all()
        libm_equal(outs[k][0].to_float32(), stream_s_oracle(oracle, *inputs(oracle, n, k)))


def test_lazy_chain_is_one_launch_and_bit_identical(gpu, oracle):
    n = 50000
    d = oracle.java_random_doubles(31415, n)
    f = gpu.RandomVariableHipFactory()

    def chain(x, y):
        t = x.add(4.0).div(2.0).mult(y).sub(x)
        return t.squared().cap(9.0).floor(0.25).addProduct(x, y).discount(x, 0.5).choose(t, x)

    x, y = f.createRandomVariable(0.0, d), f.createRandomVariable(0.0, d[::-1].copy())
    eager = chain(x, y).getRealizations()
    gpu.set_fusion(True)
    try:
        before = gpu.pool_stats()
        r = chain(x, y)
        assert gpu.pool_stats().n_kernel_launches == before.n_kernel_launches       # nothing ran yet
        fused = r.getRealizations()
        after = gpu.pool_stats()
        assert after.n_kernel_launches - before.n_kernel_launches == 1
        assert after.n_ops_executed - before.n_ops_executed == 10
    finally:
        gpu.set_fusion(False)
    assert (fused == eager).all()
    of = oracle.RandomVariableFloatFactory()
    want = chain(of.createRandomVariable(0.0, d), of.createRandomVariable(0.0, d[::-1].copy())).getRealizations()
    assert (fused == want).all()


def test_expectation_of_a_pending_chain_is_taken_in_its_own_launch(gpu, oracle):
    """`chain.getAverage()`: ONE launch computes the chain AND its {Σ, Σ², min, max}; the vector is there afterwards and later
    statistics are the usual stand-alone reductions.  Everything equal to the twin's values."""
    n = 30011
    d = oracle.java_random_doubles(27182, n)
    f, of = gpu.RandomVariableHipFactory(), oracle.RandomVariableFloatFactory()
    want = of.createRandomVariable(0.0, d).add(4.0).div(2.0).exp()
    x = f.createRandomVariable(0.0, d)
    gpu.set_fusion(True)
    try:
        before = gpu.pool_stats().n_kernel_launches
        r = x.add(4.0).div(2.0).exp()
        avg = r.getAverage()
        assert gpu.pool_stats().n_kernel_launches - before == 1
        assert abs(avg - want.getAverage()) <= 1e-13 * abs(want.getAverage())
        var = r.getVariance()                                           # the vector exists now: two more launches (mean, shifted squares), none for the chain
        assert gpu.pool_stats().n_kernel_launches - before <= 3
        assert abs(var - want.getVariance()) <= 1e-10 * want.getVariance()
        assert r.getMin() == want.getMin() and r.getMax() == want.getMax()
        assert (r.getRealizations() == want.getRealizations()).all()
    finally:
        gpu.set_fusion(False)


def test_lazy_escaping_intermediate_is_materialised_once(gpu, oracle):
    n = 4099
    d = oracle.java_random_doubles(5, n)
    f = gpu.RandomVariableHipFactory()
    gpu.set_fusion(True)
    try:
        x = f.createRandomVariable(0.0, d)
        t = x.mult(2.0).add(1.0)            # user keeps t
        u = t.squared().sub(3.0)
        before = gpu.pool_stats().n_kernel_launches
        ur = u.getRealizations()
        tr = t.getRealizations()            # already materialised as a second output of the same launch
        assert gpu.pool_stats().n_kernel_launches - before == 1
    finally:
        gpu.set_fusion(False)
    of = oracle.RandomVariableFloatFactory()
    xo = of.createRandomVariable(0.0, d)
    to = xo.mult(2.0).add(1.0)
    assert (tr == to.getRealizations()).all() and (ur == to.squared().sub(3.0).getRealizations()).all()


def test_flush_batches_identical_programs(gpu, oracle):
    """80 'components' with the same op stream but different vectors AND different scalars → one launch."""
    n, B = 10007, 80
    f = gpu.RandomVariableHipFactory()
    xs = [oracle.java_random_doubles(1000 + k, n) for k in range(B)]
    gpu.set_fusion(True)
    try:
        rvs = [f.createRandomVariable(0.0, x) for x in xs]
        before = gpu.pool_stats().n_kernel_launches
        res = [rv.mult(0.01 * (k + 1)).add(1.0).invert().accrue(rv, 0.5 + k) for k, rv in enumerate(rvs)]
        gpu.flush()
        assert gpu.pool_stats().n_kernel_launches - before == 1
        got = [r.getRealizations() for r in res]
    finally:
        gpu.set_fusion(False)
    of = oracle.RandomVariableFloatFactory()
    for k in range(B):
        xo = of.createRandomVariable(0.0, xs[k])
        want = xo.mult(0.01 * (k + 1)).add(1.0).invert().accrue(xo, 0.5 + k).getRealizations()
        assert (got[k] == want).all(), k


def test_fusion_hold_batches_long_chains(gpu, oracle):
    """fmhip_fusion_hold: chains longer than the engine's own execution threshold (≈ 40 pending methods) are normally
    launched one by one as each crosses it; recorded under a hold they stay pending and the flush batches the identical
    chains as rows — far fewer launches, the same bits."""
    n, B, L = 4099, 24, 60
    f = gpu.RandomVariableHipFactory()
    xs = [oracle.java_random_doubles(77 + k, n) for k in range(B)]

    def record(rvs):
        out = []
        for k, rv in enumerate(rvs):
            v = rv
            for i in range(L):
                v = v.mult(1.0 + 1e-3 * (k + 1)).add(1e-4 * i).cap(5.0) if i % 3 else v.squared().sqrt().add(0.5)
            out.append(v)
        return out

    gpu.set_fusion(True)
    try:
        rvs = [f.createRandomVariable(0.0, x) for x in xs]
        gpu.flush()
        before = gpu.pool_stats().n_kernel_launches
        res_plain = record(rvs)
        gpu.flush()
        launches_plain = gpu.pool_stats().n_kernel_launches - before
        before = gpu.pool_stats().n_kernel_launches
        with gpu.holding():
            assert gpu.fusion_hold(True) == 1                          # already held by the context manager
            res_held = record(rvs)
            assert gpu.pool_stats().n_kernel_launches == before        # nothing ran on the engine's own accord
        assert gpu.fusion_hold(False) == 0                             # the context manager restored "not held"
        assert gpu.fusion_hold(2) == 0                                 # a soft hold …
        with gpu.holding():
            pass
        assert gpu.fusion_hold(0) == 2                                 # … comes back from the context manager as a soft hold
        assert gpu.pool_stats().n_kernel_launches == before            # releasing the hold executes nothing by itself
        gpu.flush()
        launches_held = gpu.pool_stats().n_kernel_launches - before
        got_plain = [r.getRealizations() for r in res_plain]
        got_held = [r.getRealizations() for r in res_held]
    finally:
        gpu.fusion_hold(False)
        gpu.set_fusion(False)
    assert launches_plain >= B and launches_held <= 8, (launches_plain, launches_held)
    of = oracle.RandomVariableFloatFactory()
    want = [r.getRealizations() for r in record([of.createRandomVariable(0.0, x) for x in xs])]
    for k in range(B):
        assert (got_plain[k] == want[k]).all() and (got_held[k] == want[k]).all(), k


def test_long_chain_is_split_not_refused(gpu, oracle):
    """More pending ops / inputs than one launch holds: executed in several launches, same bits."""
    n = 2053
    f = gpu.RandomVariableHipFactory()
    of = oracle.RandomVariableFloatFactory()
    ds = [oracle.java_random_doubles(300 + k, n) for k in range(40)]
    gpu.set_fusion(True)
    try:
        acc = f.createRandomVariable(0.0, ds[0])
        for k in range(1, 40):
            acc = acc.addProduct(f.createRandomVariable(0.0, ds[k]), 0.5).mult(1.01)
        for _ in range(150):
            acc = acc.mult(1.001).add(0.001)
        got = acc.getRealizations()
    finally:
        gpu.set_fusion(False)
    acc = of.createRandomVariable(0.0, ds[0])
    for k in range(1, 40):
        acc = acc.addProduct(of.createRandomVariable(0.0, ds[k]), 0.5).mult(1.01)
    for _ in range(150):
        acc = acc.mult(1.001).add(0.001)
    assert (got == acc.getRealizations()).all()


def test_program_limits_and_validation(gpu):
    p = gpu.Program(1)
    v = 0
    for _ in range(200):
        v = p.op("ADD_S", v, s=1.0)
    p.output(v)
    with pytest.raises(gpu.FmhipError) as e:
        p.compile()
    assert e.value.code == -8           # FMHIP_ERR_PROGRAM_LIMIT
    q = gpu.Program(1)
    q.ops.append((21, 0, 5, -1, 0.0))   # operand id out of range
    q.output(1)
    with pytest.raises(gpu.FmhipError) as e:
        q.compile()
    assert e.value.code == -5


def test_run_into_and_in_place(gpu, oracle):
    n = 777
    x = oracle.f_from_double(oracle.java_random_doubles(9, n))
    p = gpu.Program(2)
    p.output(p.op("ADDPRODUCT_VS", 0, 1, s=0.25))
    p.compile()
    a, b = gpu.DeviceVector.from_host(x), gpu.DeviceVector.from_host(x * 2)
    p.run_into([[a, b]], [[a]])         # a ← a + b*0.25 in place
    want = oracle.f_v2s1("ADDPRODUCT_VS", x, x * 2, 0.25)
    assert_bits_equal(a.to_float32(), want, "in place")


def test_device_moments_output(gpu, oracle):
    """fmhip_reduce_moments_device: the 4 doubles land in caller-owned device memory (RCCL interop path)."""
    n = 12345
    x = oracle.f_from_double(oracle.java_random_doubles(3, n))
    v = gpu.DeviceVector.from_host(x)
    buf = gpu.DeviceVector.from_host(np.zeros(8, dtype=np.float32))         # 32 bytes of device memory
    gpu._native.check(gpu.lib().fmhip_reduce_moments_device(v.handle, 0.0, C.c_void_p(buf.device_ptr())))
    raw = buf.to_float32().view(np.float64)
    m = v.moments()
    assert raw[0] == m.sum and raw[1] == m.sumsq and raw[2] == m.min and raw[3] == m.max


def test_flush_fuses_overlapping_results_into_one_multi_output_launch(gpu, oracle):
    """Several results that share intermediates (a running sum feeding many updates — the LMM drift pattern) run as
    ONE launch with several outputs; identical components over different vectors are batched as rows."""
    n, K, G = 6007, 6, 5                       # K updates sharing a prefix sum, G independent groups
    f = gpu.RandomVariableHipFactory()
    of = oracle.RandomVariableFloatFactory()
    data = [[oracle.java_random_doubles(900 + 10 * g + k, n) for k in range(K)] for g in range(G)]
    dw = oracle.java_random_doubles(77, n) - 0.5

    def step(fac, Ls, dW, lam):
        s = None
        out = []
        for k, L in enumerate(Ls):
            y = L.mult(0.5).add(1.0).vid(0.5 * lam[k])             # λδ / (1 + δ L)
            s = y if s is None else s.add(y)                       # running sum (shared by all later components)
            out.append(L.addProduct(s, lam[k] * 0.5).addProduct(dW, lam[k]))
        return out, s

    gpu.set_fusion(True)
    try:
        groups = [[f.createRandomVariable(0.0, d) for d in row] for row in data]
        dWg = f.createRandomVariable(0.0, dw)
        before = gpu.pool_stats().n_kernel_launches
        res = [step(f, Ls, dWg, [0.01 * (k + 1) + 0.001 * g for k in range(K)]) for g, Ls in enumerate(groups)]
        gpu.flush()
        assert gpu.pool_stats().n_kernel_launches - before == 1          # 5 rows × (6 updates + carry) in one launch
        got = [([r.getRealizations() for r in outs], s.getRealizations()) for outs, s in res]
    finally:
        gpu.set_fusion(False)
    dWo = of.createRandomVariable(0.0, dw)
    for g in range(G):
        outs, s = step(of, [of.createRandomVariable(0.0, d) for d in data[g]], dWo, [0.01 * (k + 1) + 0.001 * g for k in range(K)])
        for k in range(K):
            assert (got[g][0][k] == outs[k].getRealizations()).all(), (g, k)
        assert (got[g][1] == s.getRealizations()).all()


def test_graph_clone_equals_recording_by_hand(gpu, oracle):
    """fmhip_graph_clone: a pending chain recorded ONCE and replicated with other vectors and other scalars gives, copy by copy,
    the bits of the same chain recorded by hand — and the copies run as rows of the original's launches."""
    n, copies = 20011, 5
    rng = np.random.default_rng(7)
    hosts = [[oracle.f_from_double(rng.uniform(0.1, 2.0, n)) for _ in range(3)] for _ in range(copies + 1)]
    shared_host = oracle.f_from_double(rng.normal(0.0, 1.0, n))
    scal = [[0.5 + 0.1 * j, 1.25 - 0.05 * j, 0.25, 0.3 + j] for j in range(copies + 1)]

    def record(vecs, shared, s):
        x, y, z = vecs
        t = x.v2s1("DISCOUNT", y, s[0]).v1s1("MULT_S", s[1])                  # scalars in recording order: s[0], s[1], s[2], s[3]
        u = t.v2s1("ADDPRODUCT_VS", shared, s[2]).v2s0("ADD", z)
        w = u.v1s1("FLOOR_S", s[3] - 5.0).v1s0("SQUARED")
        return [u, w]                                                          # two roots sharing t

    prev = gpu.set_fusion(True)
    try:
        with gpu.holding():
            shared = gpu.DeviceVector.from_host(shared_host)
            dev = [[gpu.DeviceVector.from_host(a) for a in row] for row in hosts]
            want = [record(dev[j], shared, scal[j]) for j in range(copies + 1)]       # by hand, every set
            want_bits = None
        gpu.flush()
        want_bits = [[v.to_float32().view(np.uint32) for v in pair] for pair in want]
        before = gpu.pool_stats().n_kernel_launches
        with gpu.holding():
            roots = record(dev[0], shared, scal[0])                                     # once …
            rec = gpu.graph_scalars(roots)
            assert list(rec) == [scal[0][0], scal[0][1], scal[0][2], scal[0][3] - 5.0]
            clone_scalars = [[s[0], s[1], s[2], s[3] - 5.0] for s in scal[1:]]
            got = gpu.graph_clone(roots, copies, leaf_from=dev[0], leaf_to=dev[1:], scalars=clone_scalars)      # … and replicated
            same = gpu.graph_clone(roots, 1)                                            # no substitutions, the original's scalars
        gpu.flush()
        launches = gpu.pool_stats().n_kernel_launches - before
        assert launches == 1                                                           # original + 6 copies: rows of ONE launch
        for k in range(2):
            assert (roots[k].to_float32().view(np.uint32) == want_bits[0][k]).all()
            assert (same[0][k].to_float32().view(np.uint32) == want_bits[0][k]).all()
            for j in range(copies):
                assert (got[j][k].to_float32().view(np.uint32) == want_bits[j + 1][k]).all(), (j, k)
        # errors: wrong number of scalars; a substituted operand inside the graph
        with gpu.holding():
            roots = record(dev[0], shared, scal[0])
            with pytest.raises(gpu.FmhipError):
                gpu.graph_clone(roots, 1, scalars=[[1.0, 2.0]])
            with pytest.raises(gpu.FmhipError):
                gpu.graph_clone(roots, 1, leaf_from=[roots[0]], leaf_to=[[dev[1][0]]])
        gpu.flush()
    finally:
        gpu.set_fusion(prev)
