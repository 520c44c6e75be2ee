"""fmhip_graph_clone, round 3: copies of a pending graph exist as a DESCRIPTION (csrc/runtime.hpp: ReplicaGroup) and run as further
rows of the original's launches; whenever the original is no longer exactly what was replicated, or a copy is used before the
flush, the description is expanded into ordinary nodes.  Every path must give, copy by copy, the bits of the same chain recorded
by hand (= the oracle's bits, through tests/test_gpu_fusion.py's equality of lazy and eager execution)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, COPIES = 20011, 4


def make_inputs(oracle, seed):
    rng = np.random.default_rng(seed)
    hosts = [[oracle.f_from_double(rng.uniform(0.1, 2.0, N)) for _ in range(3)] for _ in range(COPIES + 1)]
    shared = oracle.f_from_double(rng.normal(0.0, 1.0, N))
    scal = [[0.5 + 0.1 * j, 1.25 - 0.05 * j, 0.25, 0.3 + j] for j in range(COPIES + 1)]
    return hosts, shared, scal


def record(vecs, shared, s, keep_inner=None):
    x, y, z = vecs
    t = x.v2s1("DISCOUNT", y, s[0]).v1s1("MULT_S", s[1])                  # scalars in recording order: s[0], s[1], s[2], s[3]
    if keep_inner is not None:
        keep_inner.append(t)
    u = t.v2s1("ADDPRODUCT_VS", shared, s[2]).v2s0("ADD", z)
    w = u.v1s1("FLOOR_S", s[3]).v1s0("SQUARED")
    return [u, w]                                                          # two roots sharing t


def long_chain(vecs, shared, s, periods=30):
    """A backward induction that does not fit one launch (one distinct input per period): the component is cut into segments and,
    with the specialised tier, its periodic stretch runs as a rolled loop."""
    value = None
    for p in range(periods):
        libor = vecs[p % 3].v1s1("MULT_S", 1.0 + 0.013 * p) if p % 5 == 0 else vecs[p % 3]
        payoff = libor.v1s1("SUB_S", s[0]).v1s1("MULT_S", s[1])
        value = payoff if value is None else value.v2s0("ADD", payoff)
        value = value.v2s1("DISCOUNT", libor, s[2])
    return [value.v1s1("FLOOR_S", 0.0).v2s0("DIV", shared.v1s0("EXP"))]


def bits(v):
    return v.to_float32().view(np.uint32)


@pytest.fixture()
def data(gpu, oracle):
    hosts, shared_host, scal = make_inputs(oracle, 11)
    prev = gpu.set_fusion(True)
    shared = gpu.DeviceVector.from_host(shared_host)
    dev = [[gpu.DeviceVector.from_host(a) for a in row] for row in hosts]
    yield gpu, dev, shared, scal
    gpu.flush()
    gpu.set_fusion(prev)


def by_hand(gpu, dev, shared, scal, chain):
    with gpu.holding():
        want = [chain(dev[j], shared, scal[j]) for j in range(COPIES + 1)]
    gpu.flush()
    return [[bits(v) for v in roots] for roots in want]


def check(roots, copies, want):
    for k in range(len(roots)):
        assert (bits(roots[k]) == want[0][k]).all(), ("original", k)
        for j in range(len(copies)):
            assert (bits(copies[j][k]) == want[j + 1][k]).all(), (j, k)


def clone_scalars(scal):
    return [list(s) for s in scal[1:]]


def test_described_copies_are_rows_of_the_originals_launch(data):
    gpu, dev, shared, scal = data
    want = by_hand(gpu, dev, shared, scal, record)
    live0 = gpu.pool_stats().n_live_vectors
    before = gpu.pool_stats().n_kernel_launches
    with gpu.holding():
        roots = record(dev[0], shared, scal[0])
        copies = gpu.graph_clone(roots, COPIES, leaf_from=dev[0], leaf_to=dev[1:], scalars=clone_scalars(scal))
        # the only new handles are the original's roots and the copies' roots (the inner values of a copy have none)
        assert gpu.pool_stats().n_live_vectors - live0 == 2 * (COPIES + 1)
    gpu.flush()
    assert gpu.pool_stats().n_kernel_launches - before == 1
    check(roots, copies, want)


def test_long_components_with_described_copies(data):
    """Components larger than one launch: segments and rolled loops serve members that have no nodes."""
    gpu, dev, shared, scal = data
    want = by_hand(gpu, dev, shared, scal, long_chain)
    for tier in (gpu.JIT_OFF, gpu.JIT_SYNC):
        prev = gpu.set_jit(tier)
        try:
            for _ in range(2):                  # second time: straight from the plan
                with gpu.holding():
                    roots = long_chain(dev[0], shared, scal[0])
                    rec = gpu.graph_scalars(roots)
                    sc = [[scal[j][0] if v == scal[0][0] else scal[j][1] if v == scal[0][1] else scal[j][2] if v == scal[0][2] else v for v in rec] for j in range(1, COPIES + 1)]
                    copies = gpu.graph_clone(roots, COPIES, leaf_from=dev[0], leaf_to=dev[1:], scalars=sc)
                gpu.flush()
                check(roots, copies, want)
        finally:
            gpu.set_jit(prev)


def test_copy_read_before_the_flush(data):
    gpu, dev, shared, scal = data
    want = by_hand(gpu, dev, shared, scal, record)
    with gpu.holding():
        roots = record(dev[0], shared, scal[0])
        copies = gpu.graph_clone(roots, COPIES, leaf_from=dev[0], leaf_to=dev[1:], scalars=clone_scalars(scal))
        assert (bits(copies[2][1]) == want[3][1]).all()          # read under the hold: this copy's expression runs on its own
    gpu.flush()
    check(roots, copies, want)


def test_operation_on_a_copy_before_the_flush(data):
    gpu, dev, shared, scal = data
    want = by_hand(gpu, dev, shared, scal, record)
    with gpu.holding():
        roots = record(dev[0], shared, scal[0])
        copies = gpu.graph_clone(roots, COPIES, leaf_from=dev[0], leaf_to=dev[1:], scalars=clone_scalars(scal))
        more = copies[1][0].v2s0("ADD", copies[3][1])             # consumes two copies' roots
    gpu.flush()
    check(roots, copies, want)
    assert (bits(more) == (want[2][0].view(np.float32) + want[4][1].view(np.float32)).view(np.uint32)).all()


def test_operation_on_the_original_after_cloning(data):
    gpu, dev, shared, scal = data
    want = by_hand(gpu, dev, shared, scal, record)
    with gpu.holding():
        roots = record(dev[0], shared, scal[0])
        copies = gpu.graph_clone(roots, COPIES, leaf_from=dev[0], leaf_to=dev[1:], scalars=clone_scalars(scal))
        more = roots[0].v1s1("ADD_S", 1.0)                        # the original's component is larger than what was replicated
    gpu.flush()
    check(roots, copies, want)
    assert (bits(more) == (want[0][0].view(np.float32) + np.float32(1.0)).view(np.uint32)).all()


def test_handle_inside_the_original(data):
    gpu, dev, shared, scal = data
    want = by_hand(gpu, dev, shared, scal, record)
    inner = []
    with gpu.holding():
        roots = record(dev[0], shared, scal[0], keep_inner=inner)    # t keeps a handle: it escapes from the original, the copies have no such value
        copies = gpu.graph_clone(roots, COPIES, leaf_from=dev[0], leaf_to=dev[1:], scalars=clone_scalars(scal))
    gpu.flush()
    check(roots, copies, want)
    assert np.isfinite(inner[0].to_float32()).all()


def test_originals_and_copies_released_before_the_flush(data):
    gpu, dev, shared, scal = data
    want = by_hand(gpu, dev, shared, scal, record)
    with gpu.holding():
        roots = record(dev[0], shared, scal[0])
        copies = gpu.graph_clone(roots, COPIES, leaf_from=dev[0], leaf_to=dev[1:], scalars=clone_scalars(scal))
        del roots                                                    # the caller keeps the copies only …
        kept = [copies[j] for j in (0, 2, 3)]
        del copies                                                   # … and not all of them
    gpu.flush()
    for row, j in zip(kept, (0, 2, 3)):
        for k in range(2):
            assert (bits(row[k]) == want[j + 1][k]).all(), (j, k)


def test_copy_of_a_copy_and_second_copy_of_the_same_graph(data):
    gpu, dev, shared, scal = data
    want = by_hand(gpu, dev, shared, scal, record)
    with gpu.holding():
        roots = record(dev[0], shared, scal[0])
        copies = gpu.graph_clone(roots, COPIES, leaf_from=dev[0], leaf_to=dev[1:], scalars=clone_scalars(scal))
        again = gpu.graph_clone(roots, 1)                            # the graph belongs to a live description: these copies are nodes
        of_copy = gpu.graph_clone(copies[1], 1, leaf_from=dev[2], leaf_to=[dev[4]], scalars=[scal[4]])   # copy 1 reads dev[2]; its copy reads dev[4] with set 4's scalars
    gpu.flush()
    check(roots, copies, want)
    for k in range(2):
        assert (bits(again[0][k]) == want[0][k]).all()
        assert (bits(of_copy[0][k]) == want[4][k]).all()


def test_roots_that_are_vectors_already_and_roots_listed_twice(data):
    gpu, dev, shared, scal = data
    want = by_hand(gpu, dev, shared, scal, record)
    with gpu.holding():
        roots = record(dev[0], shared, scal[0])
        listed = [roots[0], dev[0][1], roots[1], roots[0], shared]     # a substituted vector, a shared vector, a root twice
        copies = gpu.graph_clone(listed, COPIES, leaf_from=dev[0], leaf_to=dev[1:], scalars=clone_scalars(scal))
    gpu.flush()
    for j in range(COPIES):
        assert (bits(copies[j][0]) == want[j + 1][0]).all() and (bits(copies[j][3]) == want[j + 1][0]).all()
        assert (bits(copies[j][2]) == want[j + 1][1]).all()
        assert (bits(copies[j][1]) == bits(dev[j + 1][1])).all()
        assert (bits(copies[j][4]) == bits(shared)).all()


def test_nothing_leaks(data):
    gpu, dev, shared, scal = data
    gpu.flush()
    live0 = gpu.pool_stats().n_live_vectors
    for variant in range(3):
        with gpu.holding():
            roots = record(dev[0], shared, scal[0])
            copies = gpu.graph_clone(roots, COPIES, leaf_from=dev[0], leaf_to=dev[1:], scalars=clone_scalars(scal))
            if variant == 1:
                copies[0][0].to_float32()
            if variant == 2:
                del copies
        gpu.flush()
        del roots
        copies = None
        assert gpu.pool_stats().n_live_vectors == live0


_FAILURE_SCRIPT = r'''
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
fm.init(0)
fm.set_fusion(True)
n, copies = 257, 1100                                   # more members than one launch part takes (1024): two parts
rng = np.random.default_rng(5)
x0, y0 = rng.uniform(0.5, 1.5, n).astype(np.float32), rng.uniform(0.5, 1.5, n).astype(np.float32)
xs = [fm.DeviceVector.from_host(x0 * np.float32(1.0 + 0.001 * j)) for j in range(copies + 1)]
y = fm.DeviceVector.from_host(y0)
def chain(x):
    t = x.v2s1("ACCRUE", y, 0.5)
    return [t, t.v1s1("ADD_S", 1.0).v2s0("MULT", y)]            # two roots: the second depends on the first
want = []
for j in range(copies + 1):                              # eager reference (counted by the failure hook as well: it is armed far behind these)
    prev = fm.set_fusion(False)
    want.append([r.to_float32() for r in chain(xs[j])])
    fm.set_fusion(True)
with fm.holding():
    roots = chain(xs[0])
    clones = fm.graph_clone(roots, copies, [xs[0]], [[xs[j]] for j in range(1, copies + 1)])
st = fm.pool_stats(); allocs_before = st.n_alloc_hits + st.n_alloc_misses
failed = None
try:
    fm.flush()
except fm.FmhipError as e:
    failed = e.code
wrong = lost = right = 0
for j in range(copies + 1):
    for r, v in enumerate(roots if j == 0 else clones[j - 1]):
        try:
            got = v.to_float32()
        except fm.FmhipError as e:
            lost += 1
            continue
        if got.tobytes() == want[j][r].tobytes(): right += 1
        else: wrong += 1
print(json.dumps({"failed": failed, "wrong": wrong, "lost": lost, "right": right, "allocs_before": allocs_before}))
'''


def test_a_failure_between_the_parts_of_a_replicated_launch_never_yields_a_wrong_copy(gpu, tmp_path):
    """An allocation that fails while a replicated graph runs in several parts (more than 1024 members): the parts that ran have
    committed their buffers, the others have not.  Afterwards every value of the original and of every copy is either RIGHT (it was
    computed, or it can still be computed from the description) or reading it is an error — never a wrong number; and the group's
    holds are gone (the process ends without leaking into the next allocation).  The hook FMHIP_TEST_FAIL_ALLOC_AT makes the N-th
    allocation of the child process fail; several N cover failures in the first part, between the parts and in the second."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fail.py"
    script.write_text(_FAILURE_SCRIPT % {"root": root})
    def child(fail_at):
        env = dict(os.environ, FMHIP_TEST_FAIL_ALLOC_AT=str(fail_at), FMHIP_JIT="off")
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    probe = child(10**12)                                 # never fails: how many allocations precede the flush
    assert probe["failed"] is None and probe["wrong"] == 0 and probe["lost"] == 0
    base = probe["allocs_before"]
    seen_failure = 0
    for extra in (1, 200, 1000, 1030, 1500, 2040, 2100, 2199):      # 2 x 1101 buffers in two parts of 1024 and 77 members
        out = child(base + extra)
        assert out["wrong"] == 0, (extra, out)
        assert out["right"] + out["lost"] == 2 * 1101
        if out["failed"] is not None:
            seen_failure += 1
        else:
            assert out["lost"] == 0
    assert seen_failure >= 2, "the sweep must hit the flush at least twice"
