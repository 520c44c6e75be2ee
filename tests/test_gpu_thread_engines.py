"""An engine per caller thread (include/fmhip.h: fmhip_set_thread_engines; csrc/abi.cpp, the `te` layer): every thread records into a pending
graph of its own, on a stream of its own; a vector of another thread enters a method as a leaf that aliases the owner's storage.  In a
process of its own (the mode ends with fmhip_shutdown): eight threads run the same workload as the main thread did alone — chains over
vectors the MAIN thread created (foreign operands, some of them still pending when the first thread asks), Brownian increments generated
by whichever thread asks first, expectations in two halves, a program compiled by the main thread, vectors handed from thread to thread
and released by the other — and every result is the main thread's to the last bit; then the native LMM driver with its Jacobian columns
on four threads against the same driver on one."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LMM_HIP = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "bin", "lmm_hip")

_SCRIPT = r'''
import importlib, json, sys, threading
import numpy as np
sys.path.insert(0, %(root)r)
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
n, n_threads = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(77)
xh = rng.uniform(0.2, 1.8, n).astype(np.float32); yh = rng.uniform(0.5, 1.5, n).astype(np.float32)
fm.init(0)
fm.set_fusion(True)
x, y = fm.DeviceVector.from_host(xh), fm.DeviceVector.from_host(yh)
pending = x.v1s1("MULT_S", 1.5).v2s0("ADD", y)                # recorded by the main thread, NOT computed yet when the threads start
td = fm.TimeDiscretization(0.0, 8, 0.25)
bm = fm.BrownianMotionHip(td, 1, n, 31415)
p = fm.Program(2); a = p.op("MULT", 0, 1); b = p.op("ADD_S", a, s=1.0); p.output(b); p.reduce(b); prog = p.compile()

def workload(k):
    out = {}
    s = x
    for i in range(8):                                       # an Euler scheme: the engine groups the time steps by the increments' first use
        s = s.v2s1("DISCOUNT", y, 0.01 * (i + 1)).v3s0("ADDPRODUCT", bm.getBrownianIncrement(i, 0).realizations, y)
    out["scheme"] = s.to_float32()
    t = pending.v1s0("EXP").v1s0("LOG").v2s0("SUB", x)
    out["chain"] = t.to_float32()
    m = t.moments()
    out["moments"] = [m.sum, m.sumsq, m.min, m.max]
    both = fm.reduce_moments_batch_end(fm.reduce_moments_batch_begin([s, pending]), 2)
    out["ticket"] = [[q.sum, q.sumsq, q.min, q.max] for q in both]
    outs, moms = prog.run([[s, y], [x, t]])
    out["program"] = [o[0].to_float32() for o in outs]
    out["program_moments"] = [[float(v) for v in q] for row in moms for q in row]
    out["kept"] = s                                           # handed to the main thread, which reads and releases it
    return out

alone = workload(-1)
alone_kept = alone.pop("kept").to_float32()
assert fm.set_thread_engines(True) is False
results, errors = [None] * n_threads, []
def run(k):
    try: results[k] = workload(k)
    except Exception as e: errors.append(repr(e))
threads = [threading.Thread(target=run, args=(k,)) for k in range(n_threads)]
for t in threads: t.start()
for t in threads: t.join()
assert not errors, errors
ok = True
for k, r in enumerate(results):
    kept = r.pop("kept")
    ok &= kept.to_float32().tobytes() == alone_kept.tobytes()         # a vector of thread k's engine, read by the main thread
    del kept                                                          # … and released by it
    for key in ("scheme", "chain"): ok &= r[key].tobytes() == alone[key].tobytes()
    ok &= r["moments"] == alone["moments"] and r["ticket"] == alone["ticket"] and r["program_moments"] == alone["program_moments"]
    ok &= all(a.tobytes() == b.tobytes() for a, b in zip(r["program"], alone["program"]))
stats = fm.pool_stats()
fm.synchronize()
del x, y, pending, bm, prog, results
fm.shutdown()
fm.init(0)                                                            # the library is usable again, on one engine
z = fm.DeviceVector.from_host(xh).v1s1("ADD_S", 1.0).to_float32()
ok &= z.tobytes() == (xh + np.float32(1.0)).tobytes()
print(json.dumps({"identical": bool(ok), "launches": int(stats.n_kernel_launches)}))
'''


def test_eight_threads_with_an_engine_each_compute_what_one_thread_computes(tmp_path):
    script = tmp_path / "threads.py"
    script.write_text(_SCRIPT % {"root": ROOT})
    r = subprocess.run([sys.executable, str(script), "200003", "8"], capture_output=True, text=True, timeout=600, env=dict(os.environ, FMHIP_JIT="sync"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["identical"] and out["launches"] > 0, out


def run_lmm(*args):
    r = subprocess.run([LMM_HIP, *[str(a) for a in args]], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_the_caller_without_hints_on_four_threads_calibrates_the_same_parameters():
    """lmm_hip --finmath-like --threads 4: the columns of the Jacobian are evaluated by four threads, an engine each (what finmath-lib's
    optimiser does with its thread pool), the Brownian increments belong to whichever thread used them first.  Same evaluations, same
    parameters, same deviations as on one thread."""
    one = run_lmm("--paths", 20000, "--mode", "calibrate", "--max-iterations", 2, "--finmath-like")
    four = run_lmm("--paths", 20000, "--mode", "calibrate", "--max-iterations", 2, "--finmath-like", "--threads", 4)
    assert four["evaluations"] == one["evaluations"]
    assert four["mean_deviation"] == one["mean_deviation"] and four["rms_deviation"] == one["rms_deviation"]
    assert four["parameters"] == one["parameters"]
