"""A caller whose handles die when a garbage collector says so — the lifetime contract of the reference
(RandomVariableCuda.java:96-106, 293-305, 384-385: device pointers recycled through WeakReference / ReferenceQueue) and of the Java
binding (java/net/finmath/hip/DeviceVector.java: a Cleaner action per handle).  The temporary of `x.add(y).mult(z)` still HAS its
handle when the engine flushes; the release arrives late, in a burst, from another thread.  The engine must not decide what to
store by live handles: it learns which handle-only values are never used again and leaves them unstored (deferred: their recipe is
kept), computing them on demand if a handle is used after all.  Bits = the eager bits in every case; launches and bytes written
= those of the caller that frees its temporaries at once, as soon as a shape has come round again."""
import threading
import time

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

N_PATHS = 50000


def inputs(gpu, oracle, k):
    mk = lambda seed, off: gpu.DeviceVector.from_host(oracle.f_from_double(oracle.java_random_doubles(seed + k, N_PATHS) + off))
    return mk(31415, 0.0), mk(27182, 0.5), mk(16180, 0.5)


def chain(x, y, z, keep):
    """18 methods, 17 temporaries; `keep` (a list) receives every temporary — the handles a collector has not released yet — or is None:
    temporaries die at once, as in C++ or CPython."""
    def t(v):
        if keep is not None:
            keep.append(v)
        return v
    a = t(x.v1s1("ADD_S", 4.0)); a = t(a.v1s1("DIV_S", 2.0)); a = t(a.v2s0("MULT", y)); a = t(a.v2s0("SUB", z))
    b = t(a.v1s0("EXP")); b = t(b.v1s0("LOG")); b = t(b.v1s0("ABS")); b = t(b.v1s0("SQRT"))
    c = t(b.v1s1("CAP_S", 1.5)); c = t(c.v1s1("FLOOR_S", 0.25)); c = t(c.v3s0("ADDPRODUCT", y, z))
    d = t(a.v3s0("CHOOSE", c, x)); d = t(d.v2s1("ACCRUE", y, 0.5)); d = t(d.v2s1("DISCOUNT", z, 0.25))
    e = t(d.v1s0("SQUARED")); e = t(e.v2s0("ADD", a)); e = t(e.v1s1("MULT_S", 0.125))
    return e.v2s0("SUB", b)


def eager_bits(gpu, oracle, k):
    prev = gpu.set_fusion(False)
    try:
        x, y, z = inputs(gpu, oracle, k)
        return chain(x, y, z, None).to_float32()
    finally:
        gpu.set_fusion(prev)


def run_rounds(gpu, oracle, rounds, release):
    """`release(keep)` is handed the round's temporaries after the round's result has been read.  Returns per round (bits, launches, bytes written)."""
    out = []
    for k in range(rounds):
        x, y, z = inputs(gpu, oracle, k)
        gpu.flush()
        s0 = gpu.engine_stats()
        keep = [] if release is not None else None
        r = chain(x, y, z, keep)
        gpu.flush()
        bits = r.to_float32()
        s1 = gpu.engine_stats()
        out.append((bits, s1["kernel_launches"] - s0["kernel_launches"], s1["algorithmic_bytes_written"] - s0["algorithmic_bytes_written"]))
        if release is not None:
            release(keep)
    return out


@pytest.fixture()
def fused(gpu):
    gpu.purge()                                             # forgets what earlier tests taught the engine about their shapes
    prev = gpu.set_fusion(True)
    yield gpu
    gpu.flush()
    gpu.set_fusion(prev)
    gpu.purge()


def test_temporaries_released_late_never_or_from_another_thread(fused, oracle):
    gpu = fused
    rounds = 6
    want = [eager_bits(gpu, oracle, k) for k in range(rounds)]
    raii = run_rounds(gpu, oracle, rounds, None)
    for k in range(rounds):
        assert_bits_equal(raii[k][0], want[k], f"RAII round {k}")
    # (1) released late: a round's temporaries go when the NEXT round is over
    gpu.purge()
    late = []
    def release_late(keep):
        late.append(keep)
        if len(late) > 1:
            late.pop(0).clear()
    got = run_rounds(gpu, oracle, rounds, release_late)
    late.clear()
    # (2) never released while the test runs
    gpu.purge()
    never = []
    got_never = run_rounds(gpu, oracle, rounds, never.append)
    # (3) released by another thread, in bursts, 30 ms after the fact
    gpu.purge()
    queue, stop = [], threading.Event()
    lock = threading.Lock()
    def collector():
        while not stop.is_set():
            time.sleep(0.03)
            with lock:
                batch = list(queue); queue.clear()
            for keep in batch:
                keep.clear()                                # the handles are released HERE, on this thread
    th = threading.Thread(target=collector); th.start()
    def release_other_thread(keep):
        with lock:
            queue.append(keep)
    try:
        got_thread = run_rounds(gpu, oracle, rounds, release_other_thread)
    finally:
        stop.set(); th.join()
    for name, res in (("late", got), ("never", got_never), ("other thread", got_thread)):
        for k in range(rounds):
            assert_bits_equal(res[k][0], want[k], f"{name} round {k}")
        # from the second occurrence of the shape on: the launches and the bytes written of the caller that frees its temporaries at once
        for k in range(1, rounds):
            assert res[k][1] == raii[k][1], (name, k, res[k][1:], raii[k][1:])
            assert res[k][2] == raii[k][2], (name, k, res[k][1:], raii[k][1:])
    never.clear()


def test_a_deferred_value_used_after_all(fused, oracle):
    """Every temporary keeps its handle; after the shape has come round, the engine leaves them unstored.  Then one of them IS used: read,
    reduced, or an operand of a later method — computed from its recipe (eager bits), and that position is stored from then on."""
    gpu = fused
    kept = []
    for k in range(3):
        x, y, z = inputs(gpu, oracle, k)
        keep = []
        r = chain(x, y, z, keep)
        gpu.flush()
        kept.append((keep, r, x, y, z))
    s = gpu.engine_stats()
    assert s["values_deferred_now"] > 0, s
    # the eager values of round 2's temporaries
    prev = gpu.set_fusion(False)
    x, y, z = inputs(gpu, oracle, 2)
    eager = []
    chain(x, y, z, eager)
    want = [v.to_float32() for v in eager]
    gpu.set_fusion(prev)
    keep = kept[2][0]
    d0 = gpu.engine_stats()["values_demanded"]
    assert_bits_equal(keep[7].to_float32(), want[7], "a deferred temporary, read")                       # b = sqrt(abs(log(exp(a))))
    m = keep[12].moments()                                                                               # accrue(...)
    w = want[12].astype(np.float64)
    assert abs(m.sum - w.sum()) <= 1e-13 * np.abs(w).sum() and m.min == w.min() and m.max == w.max()
    later = keep[3].v2s0("MULT", keep[10])                                                               # two deferred operands of a later method
    assert_bits_equal(later.to_float32(), (want[3] * want[10]).astype(np.float32), "deferred operands")
    assert gpu.engine_stats()["values_demanded"] >= d0 + 3
    assert_bits_equal(kept[2][1].to_float32(), eager_bits(gpu, oracle, 2), "the round's result")
    # the positions that were wanted are stored by the next rounds: no demand any more
    for k in range(3, 6):
        x, y, z = inputs(gpu, oracle, k)
        keep = []
        r = chain(x, y, z, keep)
        gpu.flush()
        d1 = gpu.engine_stats()["values_demanded"]
        got = keep[7].to_float32()
        assert gpu.engine_stats()["values_demanded"] == d1, "a position that was demanded once is stored from then on"
        prev = gpu.set_fusion(False)
        e = []
        chain(*inputs(gpu, oracle, k), e)
        gpu.set_fusion(prev)
        assert_bits_equal(got, e[7].to_float32(), f"round {k}")


def test_a_vector_written_in_place_while_recipes_read_it(fused, oracle):
    """fmhip_program_run_into overwrites a vector.  Values that were left unstored and whose recipe reads that vector are computed BEFORE."""
    gpu = fused
    p = gpu.Program(1)
    p.output(p.op("MULT_S", 0, s=3.0)); p.compile()
    for k in range(4):
        x, y, z = inputs(gpu, oracle, k)
        x0 = x.to_float32()
        keep = []
        r = chain(x, y, z, keep)
        gpu.flush()
        if k < 3:
            continue
        assert gpu.engine_stats()["values_deferred_now"] > 0
        p.run_into([[y]], [[x]])                            # x := 3 y, in place
        assert_bits_equal(x.to_float32(), (y.to_float32() * np.float32(3.0)).astype(np.float32), "overwritten")
        prev = gpu.set_fusion(False)
        e = []
        chain(gpu.DeviceVector.from_host(x0), y, z, e)
        gpu.set_fusion(prev)
        assert_bits_equal(keep[0].to_float32(), e[0].to_float32(), "x + 4 with the OLD x")
        assert_bits_equal(keep[11].to_float32(), e[11].to_float32(), "choose(a, c, x) with the OLD x")


def test_pool_clean_stores_what_was_deferred(fused, oracle):
    gpu = fused
    keeps = []
    for k in range(3):
        x, y, z = inputs(gpu, oracle, k)
        keep = []
        chain(x, y, z, keep)
        gpu.flush()
        keeps.append(keep)
    assert gpu.engine_stats()["values_deferred_now"] > 0
    gpu.clean()
    assert gpu.engine_stats()["values_deferred_now"] == 0
    prev = gpu.set_fusion(False)
    e = []
    chain(*inputs(gpu, oracle, 2), e)
    gpu.set_fusion(prev)
    for i in (0, 5, 9, 16):
        assert_bits_equal(keeps[2][i].to_float32(), e[i].to_float32(), f"temporary {i}")


def test_releases_free_the_recipes(fused, oracle):
    """The release that eventually arrives frees a deferred value's recipe — and with it the vectors only the recipe kept alive."""
    gpu = fused
    base, base_bytes = gpu.pool_stats().n_live_vectors, gpu.pool_stats().bytes_in_use
    for k in range(4):
        x, y, z = inputs(gpu, oracle, k)
        keep = []
        r = chain(x, y, z, keep)
        gpu.flush()
        del x, y, z, r                                       # inputs and result go; the temporaries' recipes still read the inputs
        assert gpu.engine_stats()["values_deferred_now"] > 0 or k == 0
        keep.clear()
        assert gpu.engine_stats()["values_deferred_now"] == 0
        assert gpu.pool_stats().n_live_vectors == base
        assert gpu.pool_stats().bytes_in_use == base_bytes
