"""Execution tiers (include/fmhip.h "execution tiers"): every compiled program runs either on the bytecode interpreter
kernel or on a specialised kernel generated from its op stream and compiled with hiprtc.  The two tiers must be
BIT-IDENTICAL (outputs and fused moments), and both equal to the oracle."""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def dv(gpu, arr):
    return gpu.DeviceVector.from_host(np.asarray(arr, dtype=np.float32))


@pytest.fixture()
def tiers(gpu):
    """Runs `fn` once per tier and returns both results; restores the mode."""
    def run(fn):
        prev = gpu.set_jit(gpu.JIT_OFF)
        try:
            a = fn()
            gpu.set_jit(gpu.JIT_SYNC)
            b = fn()
        finally:
            gpu.set_jit(prev)
        return a, b
    return run


def inputs(oracle, n, seed=1):
    x = oracle.f_from_double(oracle.java_random_doubles(seed, n))
    y = oracle.f_from_double(oracle.java_random_doubles(seed + 1, n) + 0.5)
    z = oracle.f_from_double(oracle.java_random_doubles(seed + 2, n) - 0.5)
    return x, y, z


def stream_s(gpu):
    p = gpu.Program(3)
    x, y, z = 0, 1, 2
    t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", x, s=4.0), s=2.0), y), z)
    u = p.op("SQRT", p.op("ABS", p.op("LOG", p.op("EXP", t))))
    v = p.op("ADDPRODUCT", p.op("FLOOR_S", p.op("CAP_S", u, s=1.5), s=0.25), y, z)
    w = p.op("CHOOSE", t, v, x)
    p.output(w)
    p.reduce(w)
    return p


@pytest.mark.parametrize("n", [1, 5, 1023, 2048, 2049, 8192, 100003])
def test_stream_s_tiers_identical(gpu, oracle, tiers, n):
    x, y, z = inputs(oracle, n)

    def run():
        p = stream_s(gpu).compile()
        rows = [[dv(gpu, x), dv(gpu, y), dv(gpu, z)], [dv(gpu, y), dv(gpu, x), dv(gpu, z)]]
        outs, m = p.run(rows, shifts=[0.125])
        return [o[0].to_float32() for o in outs], m, p.tier()[0]

    (o0, m0, t0), (o1, m1, t1) = tiers(run)
    assert t0 == 0 and t1 == 1                      # the second run really used the specialised kernel
    for a, b in zip(o0, o1):
        assert_bits_equal(a, b, f"tier outputs n={n}")
    assert m0.tobytes() == m1.tobytes()             # same accumulation order, same combine: identical fp64 moments


SINGLE = [("SQUARED", 1, False), ("SQRT", 1, False), ("EXP", 1, False), ("LOG", 1, False), ("INVERT", 1, False), ("ABS", 1, False),
          ("SIN", 1, False), ("COS", 1, False), ("ISNAN", 1, False),
          ("CAP_S", 1, True), ("FLOOR_S", 1, True), ("ADD_S", 1, True), ("SUB_S", 1, True), ("BUS_S", 1, True), ("MULT_S", 1, True),
          ("DIV_S", 1, True), ("VID_S", 1, True), ("POW_S", 1, True),
          ("CAP", 2, False), ("FLOOR", 2, False), ("ADD", 2, False), ("SUB", 2, False), ("MULT", 2, False), ("DIV", 2, False),
          ("ACCRUE", 2, True), ("DISCOUNT", 2, True), ("ADDPRODUCT_VS", 2, True),
          ("ADDPRODUCT", 3, False), ("ADDRATIO", 3, False), ("SUBRATIO", 3, False), ("CHOOSE", 3, False)]


@pytest.mark.parametrize("name,nvec,has_s", SINGLE)
def test_every_opcode_tiers_identical_and_match_oracle(gpu, oracle, tiers, name, nvec, has_s):
    """One program per opcode, with the operand first as the accumulator and then as a register operand
    (both micro-op forms _A/_B), on seeded data plus IEEE edge values."""
    n = 20011
    x, y, z = inputs(oracle, n, seed=7)
    edge = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754942e-38, 3.4028235e38, 88.0, -104.0, 0.5, 2.0],
                    dtype=np.float32)
    x[:edge.size] = edge; y[:edge.size] = edge[::-1]; z[:edge.size] = np.roll(edge, 3)
    s = 1.0 / 3.0

    def run():
        res = []
        for order in range(2):
            p = gpu.Program(3)
            a = p.op("ADD_S", 0, s=0.0) if order == 0 else 0        # order 0: first operand comes out of the accumulator
            b = p.op("ADD_S", 1, s=0.0) if order == 1 else 1
            args = [a, b, 2][:nvec]
            w = p.op(name, *args, s=s) if has_s else p.op(name, *args)
            p.output(w); p.reduce(w)
            p.compile()
            outs, m = p.run([[dv(gpu, x), dv(gpu, y), dv(gpu, z)]])
            res.append((outs[0][0].to_float32(), m))
        return res

    with np.errstate(all="ignore"):
        r0, r1 = tiers(run)
        for (o0, m0), (o1, m1) in zip(r0, r1):
            assert_bits_equal(o0, o1, name)
            assert m0.tobytes() == m1.tobytes()
        # and against the oracle (x + 0.0 is not the identity for -0.0: apply it on the oracle side too)
        xa = oracle.f_v1s1("ADD_S", x, 0.0)
        if nvec == 1:
            want = oracle.f_v1s1(name, xa, s) if has_s else oracle.f_v1s0(name, xa)
        elif nvec == 2:
            want = oracle.f_v2s1(name, xa, y, s) if has_s else oracle.f_v2s0(name, xa, y)
        else:
            want = oracle.f_v3s0(name, xa, y, z)
        got = r1[0][0]
        if name in ("EXP", "LOG", "SIN", "COS", "POW_S"):
            ia, ib = got.view(np.int32).astype(np.int64), want.view(np.int32).astype(np.int64)
            ok = (np.abs(ia - ib) <= 1) | (np.isnan(got) & np.isnan(want))
            assert ok.all() and ((ia != ib) & ~np.isnan(got)).mean() <= 1e-4, name
        else:
            assert_bits_equal(got, want, name)


def test_many_live_values_program(gpu, oracle, tiers):
    """A wide program (4-element / 16-register variant: 10 live values, 3 outputs, 2 reductions)."""
    n = 30001
    x, y, z = inputs(oracle, n, seed=11)

    def run():
        p = gpu.Program(3)
        vals = [p.op("ADD_S", 0, s=float(k)) for k in range(10)]
        acc = vals[0]
        for k in range(1, 10):
            acc = p.op("ADDPRODUCT", acc, vals[k], 1 if k % 2 else 2)
        e = p.op("EXP", p.op("MULT_S", acc, s=0.01))
        q = p.op("DIV", e, vals[3])
        p.output(acc); p.output(e); p.output(q)
        p.reduce(e); p.reduce(q)
        p.compile()
        outs, m = p.run([[dv(gpu, x), dv(gpu, y), dv(gpu, z)]], shifts=[0.0, 1.5])
        return [o.to_float32() for o in outs[0]], m

    (o0, m0), (o1, m1) = tiers(run)
    for a, b in zip(o0, o1):
        assert_bits_equal(a, b, "wide program")
    assert m0.tobytes() == m1.tobytes()


def test_lazy_front_end_on_specialised_kernels(gpu, oracle):
    """RandomVariable-style lazy chains with JIT_SYNC: every fused launch runs a specialised kernel; results as the oracle."""
    n = 50021
    x, y, z = inputs(oracle, n, seed=21)
    prev_jit = gpu.set_jit(gpu.JIT_SYNC)
    prev_fusion = gpu.set_fusion(True)
    try:
        before = gpu.jit_stats()
        a, b = dv(gpu, x), dv(gpu, y)
        r = a.v1s1("ADD_S", 4.0).v2s0("DIV", b).v1s0("SQUARED").v3s0("ADDPRODUCT", a, b).v1s1("FLOOR_S", 0.5)
        got = r.to_float32()
        m = r.moments(0.0)
        after = gpu.jit_stats()
    finally:
        gpu.set_fusion(prev_fusion)
        gpu.set_jit(prev_jit)
    t = oracle.f_v1s1("ADD_S", x, 4.0); t = oracle.f_v2s0("DIV", t, y); t = oracle.f_v1s0("SQUARED", t)
    t = oracle.f_v3s0("ADDPRODUCT", t, x, y); want = oracle.f_v1s1("FLOOR_S", t, 0.5)
    assert_bits_equal(got, want, "lazy chain on the specialised tier")
    assert after["compiled"] > before["compiled"] and after["failed"] == 0
    assert abs(m.sum - float(np.sum(want.astype(np.float64)))) <= 1e-9 * abs(m.sum)


def test_auto_mode_promotes_explicit_programs_in_the_background(gpu, oracle):
    n = 4099
    x, y, z = inputs(oracle, n, seed=31)
    prev = gpu.set_jit(gpu.JIT_AUTO)
    try:
        p = stream_s(gpu).compile()                 # queued at creation, does not block
        rows = [[dv(gpu, x), dv(gpu, y), dv(gpu, z)]]
        outs0, m0 = p.run(rows)                     # whichever tier is ready
        gpu.jit_wait()
        assert p.tier()[0] == 1 and p.tier()[1] > 0
        outs1, m1 = p.run(rows)
        assert_bits_equal(outs0[0][0].to_float32(), outs1[0][0].to_float32(), "auto mode")
        assert m0.tobytes() == m1.tobytes()
        assert gpu.jit_stats()["failed"] == 0 and gpu.jit_stats()["pending"] == 0
    finally:
        gpu.set_jit(prev)


def test_fast_math_programs_specialise_too(gpu, oracle, tiers):
    n = 10007
    x, y, z = inputs(oracle, n, seed=41)
    prev = gpu.set_math_mode(gpu.MATH_FAST)
    try:
        def run():
            p = stream_s(gpu).compile()
            outs, m = p.run([[dv(gpu, x), dv(gpu, y), dv(gpu, z)]])
            return outs[0][0].to_float32(), m
        (o0, m0), (o1, m1) = tiers(run)
    finally:
        gpu.set_math_mode(prev)
    assert_bits_equal(o0, o1, "fast math tiers")
    assert m0.tobytes() == m1.tobytes()


def test_persistent_code_object_cache(tmp_path):
    """A second PROCESS finds the specialised kernels of the first in $FMHIP_JIT_CACHE_DIR: no recompilation, same bits."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import importlib, sys, json, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "fm = importlib.import_module('finmath-lib-cuda-extensions_amd')\n"
        "fm.init(0); fm.set_jit(fm.JIT_SYNC)\n"
        "x = fm.DeviceVector.from_host(np.linspace(0.1, 2.0, 5003, dtype=np.float32))\n"
        "p = fm.Program(1); w = p.op('LOG', p.op('EXP', p.op('MULT_S', 0, s=1.25))); p.output(w); p.reduce(w); p.compile()\n"
        "outs, m = p.run([[x]])\n"
        "print(json.dumps({'stats': fm.jit_stats(), 'sum': float(m[0, 0, 0]), 'bits': int(outs[0][0].to_float32().view(np.uint32).sum())}))\n")
    env = dict(os.environ, FMHIP_JIT_CACHE_DIR=str(tmp_path))
    runs = []
    for _ in range(2):
        out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        runs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    assert runs[0]["stats"]["compiled"] == 1 and runs[0]["stats"]["disk_cache_hits"] == 0
    assert runs[1]["stats"]["compiled"] == 1 and runs[1]["stats"]["disk_cache_hits"] == 1
    assert runs[0]["sum"] == runs[1]["sum"] and runs[0]["bits"] == runs[1]["bits"]
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".co")]) == 1


def test_traffic_statistics_count_algorithmic_bytes(gpu, oracle):
    """fmhip_traffic_stats: 4 B x N x (inputs + outputs) x rows per program launch, reductions add nothing."""
    import ctypes as C
    n = 12345
    x, y, _ = inputs(oracle, n, seed=51)
    def stats():
        b, j = C.c_int64(0), C.c_int64(0)
        gpu._native.check(gpu.lib().fmhip_traffic_stats(C.byref(b), C.byref(j)))
        return b.value, j.value
    prev = gpu.set_jit(gpu.JIT_SYNC)
    try:
        p = gpu.Program(2)
        w = p.op("ADDPRODUCT_VS", p.op("EXP", 0), 1, s=0.5)
        p.output(w); p.reduce(w); p.compile()
        rows = [[dv(gpu, x), dv(gpu, y)] for _ in range(3)]
        b0, j0 = stats()
        p.run(rows)
        b1, j1 = stats()
        assert b1 - b0 == 4 * n * (2 + 1) * 3 and j1 - j0 == 1
        dv(gpu, x).moments()                        # stand-alone reduction: 4 B per element read, nothing written
        b2, _ = stats()
        assert b2 - b1 == 4 * n
    finally:
        gpu.set_jit(prev)


def test_kernel_pack_serves_a_cold_machine(tmp_path):
    """A machine that has never run the engine (empty code-object cache) must find the kernels of the LMM calibration in the
    build-time pack (lib/jit_pack) instead of compiling them while the calibration runs on the interpreter tier."""
    import json
    import subprocess
    lmm = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "bin", "lmm_hip")
    env = dict(os.environ, FMHIP_JIT_CACHE_DIR=str(tmp_path / "cache"), FMHIP_JIT="auto")
    env.pop("FMHIP_ROLL", None)                                         # the pack holds the rolled kernels of this workload
    out = subprocess.run([lmm, "--paths", "200000", "--mode", "evaluate", "--evaluations", "24", "--jacobian-batch", "8"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["specialised_kernels"] >= 4 and r["specialisations_from_disk_cache"] >= 1          # the user cache was empty: these came from the pack
    if r["specialisations_from_disk_cache"] < r["specialised_kernels"]:
        pytest.skip(f"kernel pack is stale: {r['specialised_kernels'] - r['specialisations_from_disk_cache']} of {r['specialised_kernels']} kernels "
                    "were compiled at run time — re-record csrc/kernel_pack.txt with benchmarks/record_kernel_pack.sh")
    off = subprocess.run([lmm, "--paths", "200000", "--mode", "evaluate", "--evaluations", "24", "--jacobian-batch", "8"], capture_output=True, text=True, timeout=600,
                         env=dict(env, FMHIP_JIT_CACHE_DIR=str(tmp_path / "cache2"), FMHIP_JIT_PACK_DIR="off"))
    assert off.returncode == 0, off.stderr
    q = json.loads(off.stdout.strip().splitlines()[-1])
    assert q["specialisations_from_disk_cache"] == 0
    assert q["model_volatility"] == r["model_volatility"]                                          # same kernels either way


def test_moments_do_not_depend_on_the_shape_of_the_launch(gpu, oracle):
    """The reduction tree (fm_kernel_parts.hpp) is a function of the data and of the vector's length: a launch of few rows gives every
    workgroup ONE unit of it (runtime.cpp: unit_launch — four times the workgroups), a launch of many rows a span of four units; both
    tiers, 4 or 8 elements per lane.  Outputs and moments of a row must not depend on which launch computed it."""
    import importlib
    fusion = importlib.import_module("test_gpu_fusion")
    for n in (100_003, 2048 * 4 * 3 + 1, 2047):
        many = 1 + (128 * 8192 + n - 1) // n                 # rows x spans > 128: the launch over all rows takes whole spans
        rows = [[gpu.DeviceVector.from_host(a) for a in fusion.inputs(oracle, n, k % 5)] for k in range(min(many, 40))]
        rows = [rows[k % len(rows)] for k in range(many)]
        results = {}
        for tier in (gpu.JIT_OFF, gpu.JIT_SYNC):
            prev = gpu.set_jit(tier)
            try:
                for name, p in (("stream S", fusion.stream_s_program(gpu)), ("4 elements per lane", four_per_lane_program(gpu))):
                    single, m1 = p.run([rows[3]])
                    batch, mb = p.run(rows)
                    results[tier, name] = (single[0][0].to_float32().view(np.uint32), m1[0].tobytes(), batch[3][0].to_float32().view(np.uint32), mb[3].tobytes())
            finally:
                gpu.set_jit(prev)
        for (tier, name), r in results.items():
            ref = results[gpu.JIT_OFF, name]
            assert (r[0] == ref[0]).all() and (r[2] == ref[0]).all(), (n, tier, name)
            assert r[1] == ref[1] and r[3] == ref[1], (n, tier, name)


def four_per_lane_program(gpu):
    """pow calls out-of-line library code: such programs run 4 elements per lane (a unit of the reduction tree is two of their passes)."""
    p = gpu.Program(3)
    w = p.op("MULT", p.op("POW_S", p.op("ADD_S", p.op("ABS", 0), s=1.0), s=1.5), 1)
    v = p.op("SUB", w, 2)
    p.output(v); p.reduce(v); p.reduce(w)
    return p.compile()
