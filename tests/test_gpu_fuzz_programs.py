"""Differential fuzzing of the program compiler and both execution tiers: seeded random op DAGs (1-45 ops over 1-6 inputs,
random escaping outputs and fused reductions, random batch sizes and ragged lengths) evaluated

    (1) op by op on the oracle (numpy fp32 restatement of RandomVariableFromFloatArray),
    (2) as an explicit compiled program on the interpreter tier,
    (3) as the same program on the specialised (hiprtc) tier,
    (4) as lazily fused RandomVariable-style calls.

(2) = (3) bit for bit always; (1) = (2) = (4) bit for bit for the exactly rounded opcodes (exp/log participate in the
tier comparison only).  Register allocation, accumulator scheduling, operand-order variants (_A/_B), program splitting and
multi-output escapes are all exercised by construction."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu
SCALE = int(__import__("os").environ.get("FMHIP_FUZZ_SCALE", "1"))      # one-off deep runs: FMHIP_FUZZ_SCALE=25

EXACT = [("SQUARED", 1, 0), ("SQRT", 1, 0), ("INVERT", 1, 0), ("ABS", 1, 0), ("ISNAN", 1, 0),
         ("CAP_S", 1, 1), ("FLOOR_S", 1, 1), ("ADD_S", 1, 1), ("SUB_S", 1, 1), ("BUS_S", 1, 1), ("MULT_S", 1, 1), ("DIV_S", 1, 1), ("VID_S", 1, 1),
         ("CAP", 2, 0), ("FLOOR", 2, 0), ("ADD", 2, 0), ("SUB", 2, 0), ("MULT", 2, 0), ("DIV", 2, 0),
         ("ACCRUE", 2, 1), ("DISCOUNT", 2, 1), ("ADDPRODUCT_VS", 2, 1),
         ("ADDPRODUCT", 3, 0), ("ADDRATIO", 3, 0), ("SUBRATIO", 3, 0), ("CHOOSE", 3, 0)]
LIBM = [("EXP", 1, 0), ("LOG", 1, 0)]
SCALARS = [0.5, 2.0, -1.25, 1.0 / 3.0, 4.0, 0.0, 1e-3, 8.0, -0.5]


def random_program(rng, n_in, n_ops, ops):
    prog = []                                   # (name, args, scalar)
    for i in range(n_ops):
        name, nvec, has_s = ops[rng.integers(len(ops))]
        avail = n_in + i
        # bias towards recent values (chains) but keep long-range reuse (register pressure)
        args = [int(avail - 1 - min(avail - 1, int(rng.exponential(3.0)))) for _ in range(nvec)]
        prog.append((name, args, float(SCALARS[rng.integers(len(SCALARS))]) if has_s else 0.0))
    n_vals = n_in + n_ops
    n_out = int(rng.integers(1, 4))
    outs = sorted(set([n_vals - 1] + [int(rng.integers(n_in, n_vals)) for _ in range(n_out - 1)]))
    reds = [int(rng.integers(n_in, n_vals)) for _ in range(int(rng.integers(0, 3)))]
    return prog, outs, reds


def oracle_eval(oracle, prog, inputs):
    vals = list(inputs)
    with np.errstate(all="ignore"):
        for name, args, s in prog:
            a = [vals[k] for k in args]
            if len(a) == 1:
                vals.append(oracle.f_v1s1(name, a[0], s) if name.endswith("_S") else oracle.f_v1s0(name, a[0]))
            elif len(a) == 2:
                vals.append(oracle.f_v2s1(name, a[0], a[1], s) if name in ("ACCRUE", "DISCOUNT", "ADDPRODUCT_VS") else oracle.f_v2s0(name, a[0], a[1]))
            else:
                vals.append(oracle.f_v3s0(name, a[0], a[1], a[2]))
    return vals


def build(gpu, n_in, prog, outs, reds):
    p = gpu.Program(n_in)
    for name, args, s in prog:
        p.op(name, *args, s=s)
    for o in outs:
        p.output(o)
    for r in reds:
        p.reduce(r)
    return p.compile()


def run_tier(gpu, tier, n_in, prog, outs, reds, rows):
    prev = gpu.set_jit(tier)
    try:
        p = build(gpu, n_in, prog, outs, reds)
        vecs = [[gpu.DeviceVector.from_host(x) for x in row] for row in rows]
        o, m = p.run(vecs, shifts=[0.25 * k for k in range(len(reds))] if reds else None)
        return [[v.to_float32() for v in r] for r in o], (m.tobytes() if m is not None else b""), p.tier()[0]
    finally:
        gpu.set_jit(prev)


def make_rows(oracle, rng, n_in, n, batch):
    rows = []
    for b in range(batch):
        row = []
        for k in range(n_in):
            x = oracle.f_from_double(oracle.java_random_doubles(int(rng.integers(1, 1 << 30)), n) * 3.0 - 1.0)
            if n > 8:
                x[rng.integers(n)] = np.float32(np.nan) if rng.random() < 0.3 else np.float32(0.0)
                x[rng.integers(n)] = np.float32(np.inf) if rng.random() < 0.2 else np.float32(-0.0)
            row.append(x)
        rows.append(row)
    return rows


@pytest.mark.parametrize("seed", range(40 * SCALE))
def test_random_exact_programs(gpu, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n_in = int(rng.integers(1, 7))
    n_ops = int(rng.integers(1, 46))
    prog, outs, reds = random_program(rng, n_in, n_ops, EXACT)
    n = int(rng.choice([1, 7, 255, 1024, 2049, 10007, 40001]))
    batch = int(rng.choice([1, 1, 2, 5]))
    rows = make_rows(oracle, rng, n_in, n, batch)
    try:
        o0, m0, t0 = run_tier(gpu, gpu.JIT_OFF, n_in, prog, outs, reds, rows)
    except gpu.FmhipError as e:
        assert e.code == -8                       # FMHIP_ERR_PROGRAM_LIMIT: too many live values / outputs for ONE explicit launch
        pytest.skip("program exceeds the one-launch limits (the lazy front-end would split it)")
    o1, m1, t1 = run_tier(gpu, gpu.JIT_SYNC, n_in, prog, outs, reds, rows)
    assert t0 == 0 and t1 == 1 and m0 == m1
    for b in range(batch):
        want = oracle_eval(oracle, prog, rows[b])
        for k, vid in enumerate(outs):
            assert_bits_equal(o0[b][k], o1[b][k], f"tiers seed {seed} out {vid}")
            assert_bits_equal(o1[b][k], want[vid], f"oracle seed {seed} out {vid}")


@pytest.mark.parametrize("seed", range(15 * SCALE))
def test_random_programs_with_exp_log_tiers_identical(gpu, oracle, seed):
    rng = np.random.default_rng(5000 + seed)
    n_in = int(rng.integers(1, 5))
    prog, outs, reds = random_program(rng, n_in, int(rng.integers(2, 30)), EXACT + LIBM + LIBM)
    rows = make_rows(oracle, rng, n_in, int(rng.choice([5, 1025, 30011])), int(rng.choice([1, 3])))
    try:
        o0, m0, t0 = run_tier(gpu, gpu.JIT_OFF, n_in, prog, outs, reds, rows)
    except gpu.FmhipError as e:
        assert e.code == -8
        pytest.skip("program exceeds the one-launch limits")
    o1, m1, t1 = run_tier(gpu, gpu.JIT_SYNC, n_in, prog, outs, reds, rows)
    assert m0 == m1
    for a, b in zip(o0, o1):
        for x, y in zip(a, b):
            assert_bits_equal(x, y, f"tiers seed {seed}")


class _RV:
    """Minimal lazy evaluation of a random program through the eager/lazy entry points (fmhip_call_v*)."""
    @staticmethod
    def run(gpu, prog, inputs):
        vals = [gpu.DeviceVector.from_host(x) for x in inputs]
        for name, args, s in prog:
            a = [vals[k] for k in args]
            if len(a) == 1:
                vals.append(a[0].v1s1(name, s) if name.endswith("_S") else a[0].v1s0(name))
            elif len(a) == 2:
                vals.append(a[0].v2s1(name, a[1], s) if name in ("ACCRUE", "DISCOUNT", "ADDPRODUCT_VS") else a[0].v2s0(name, a[1]))
            else:
                vals.append(a[0].v3s0(name, a[1], a[2]))
        return vals


@pytest.mark.parametrize("seed", range(25 * SCALE))
def test_random_lazy_chains_match_oracle(gpu, oracle, seed):
    """The lazy front-end with arbitrary DAGs (including ones far beyond one launch: splitting, escaping intermediates)."""
    rng = np.random.default_rng(9000 + seed)
    n_in = int(rng.integers(1, 7))
    prog, outs, _ = random_program(rng, n_in, int(rng.integers(5, 120)), EXACT)
    rows = make_rows(oracle, rng, n_in, int(rng.choice([3, 1023, 20011])), 1)
    want = oracle_eval(oracle, prog, rows[0])
    for jit in (gpu.JIT_OFF, gpu.JIT_SYNC):
        prev_j = gpu.set_jit(jit)
        prev_f = gpu.set_fusion(True)
        try:
            vals = _RV.run(gpu, prog, rows[0])
            keep = [vals[k] for k in outs]
            del vals                                   # intermediates without a live handle never touch HBM
            got = [v.to_float32() for v in keep]
        finally:
            gpu.set_fusion(prev_f)
            gpu.set_jit(prev_j)
        for g, vid in zip(got, outs):
            assert_bits_equal(g, want[vid], f"lazy seed {seed} value {vid}")
