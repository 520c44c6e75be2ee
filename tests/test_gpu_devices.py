"""ONE process, SEVERAL devices (include/fmhip.h: fmhip_init_devices; csrc/sharded.cpp): every vector is cut into blocks of paths, one per
device of the list, every method runs on every block, reads gather the blocks, host-side moments are the per-device moments added in
device order.  Tested on ONE GPU with the device lists [0, 0] and [0, 0, 0] (shards on separate streams of one device — the box has one):
element-wise results bit-identical to the unsharded run, moments equal to the combination of the shards' moments to the last bit and to
the unsharded moments within the reassociation bound, Brownian increments identical, the LMM objective evaluation of lmm_hip --devices
identical to the unsharded one to 1e-12.  UNMEASURED on more than one physical GPU."""
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
import importlib, json, sys
import numpy as np
sys.path.insert(0, %(root)r)
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
par = importlib.import_module("finmath-lib-cuda-extensions_amd.parallel")
devices = json.loads(sys.argv[1])
n = int(sys.argv[2])
rng = np.random.default_rng(2024)
xh = rng.uniform(0.2, 1.8, n).astype(np.float32); yh = rng.uniform(0.5, 1.5, n).astype(np.float32); zh = rng.normal(0.0, 1.0, n).astype(np.float32)
if n > 7: zh[7] = np.nan                                     # a NaN travels through min / max of the block it lies in and of the union

def workload(record):
    x, y, z = fm.DeviceVector.from_host(xh), fm.DeviceVector.from_host(yh), fm.DeviceVector.from_host(zh)
    out = {}
    for fusion in (False, True):
        prev = fm.set_fusion(fusion)
        t = x.v1s1("ADD_S", 4.0).v1s1("DIV_S", 2.0).v2s0("MULT", y).v2s0("SUB", z)
        u = t.v1s0("EXP").v1s0("LOG").v1s0("ABS").v1s0("SQRT")
        v = u.v1s1("CAP_S", 1.5).v1s1("FLOOR_S", 0.25).v3s0("ADDPRODUCT", y, x)
        w = t.v3s0("CHOOSE", v, x)
        long = x
        for k in range(70): long = long.v2s1("DISCOUNT", y, 0.01 * (k + 1)).v2s0("ADD", x)          # does not fit one launch
        key = "fused" if fusion else "eager"
        out[key] = {"w": w.to_float32(), "v": v.to_float32(), "long": long.to_float32()}
        mw, mx, ml = w.moments(), x.moments(), long.moments(shift=0.5)
        out[key]["moments"] = [[m.sum, m.sumsq, m.min, m.max] for m in (mw, mx, ml)]
        both = fm.reduce_moments_batch_end(fm.reduce_moments_batch_begin([x, long]), 2)
        out[key]["ticket"] = [[m.sum, m.sumsq, m.min, m.max] for m in both]
        fm.set_fusion(prev)
    td = fm.TimeDiscretization(0.0, 6, 0.25)
    bm = fm.BrownianMotionHip(td, 2, n, 31415)
    out["bm"] = [bm.getBrownianIncrement(i, f).realizations.to_float32() for i in (0, 5) for f in (0, 1)]
    p = fm.Program(2); a = p.op("MULT", 0, 1); b = p.op("ADD_S", a, s=1.0); p.output(b); p.reduce(b); prog = p.compile()
    rows = [[x, y], [y, x]]
    outs, moms = prog.run(rows)
    out["program"] = [o[0].to_float32() for o in outs]
    out["program_moments"] = [[float(v) for v in m] for row in moms for m in row]
    return out

fm.init(0)
single = workload(False)
# the shards' own moments, by hand on the single device: blocks of par.path_shard, reduced one by one
shard_moments = {}
for D in sorted({len(d) for d in devices}):
    per = []
    for d in range(D):
        off, cnt = par.path_shard(n, D, d)
        block = fm.DeviceVector.from_host(xh[off:off + cnt])
        m = block.moments()
        per.append([[m.sum, m.sumsq, m.min, m.max]])
    shard_moments[D] = fm.expectation_combine(np.array(per))[0].tolist()
fm.shutdown()
report = {}
for dev in devices:
    fm.init_devices(dev)
    assert fm.device_count() == len(dev)
    got = workload(True)
    fm.shutdown()
    r = {"elementwise_identical": True, "moments_rel": 0.0}
    for key in ("eager", "fused"):
        for name in ("w", "v", "long"):
            r["elementwise_identical"] &= bool((got[key][name].view(np.uint32) == single[key][name].view(np.uint32)).all())
        for ms_got, ms_one in ((got[key]["moments"], single[key]["moments"]), (got[key]["ticket"], single[key]["ticket"])):
            for a, b in zip(ms_got, ms_one):
                for c in (0, 1):
                    r["moments_rel"] = max(r["moments_rel"], abs(a[c] - b[c]) / max(abs(b[c]), 1e-300))
                r["elementwise_identical"] &= (np.isnan(a[2]) and np.isnan(b[2])) or a[2] == b[2]
                r["elementwise_identical"] &= (np.isnan(a[3]) and np.isnan(b[3])) or a[3] == b[3]
    r["bm_identical"] = all(bool((g.view(np.uint32) == s.view(np.uint32)).all()) for g, s in zip(got["bm"], single["bm"]))
    r["program_identical"] = all(bool((g.view(np.uint32) == s.view(np.uint32)).all()) for g, s in zip(got["program"], single["program"]))
    r["program_moments_rel"] = max(abs(a[0] - b[0]) / abs(b[0]) for a, b in zip(got["program_moments"], single["program_moments"]))
    mx = got["eager"]["moments"][1]
    r["x_moments_equal_combined_shard_moments"] = np.array(mx).tobytes() == np.array(shard_moments[len(dev)]).tobytes()
    report[json.dumps(dev)] = r
print(json.dumps(report))
'''


@pytest.mark.parametrize("n", [100_003, 5])
def test_a_device_list_gives_the_unsharded_results(tmp_path, n):
    script = tmp_path / "devices.py"
    script.write_text(_SCRIPT % {"root": ROOT})
    lists = [[0, 0], [0, 0, 0]]
    r = subprocess.run([sys.executable, str(script), json.dumps(lists), str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    report = json.loads(r.stdout.strip().splitlines()[-1])
    for dev in lists:
        got = report[json.dumps(dev)]
        assert got["elementwise_identical"], (dev, got)
        assert got["bm_identical"] and got["program_identical"], (dev, got)
        assert got["moments_rel"] <= 1e-12 and got["program_moments_rel"] <= 1e-12, (dev, got)
        assert got["x_moments_equal_combined_shard_moments"], (dev, got)


def test_errors_and_what_a_device_list_does_not_offer(tmp_path):
    script = tmp_path / "errors.py"
    script.write_text(r'''
import importlib, sys
import numpy as np
sys.path.insert(0, %(root)r)
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
E = fm._native
fm.init_devices([0, 0])
x = fm.DeviceVector.from_host(np.ones(1000, dtype=np.float32)); y = fm.DeviceVector.from_host(np.ones(999, dtype=np.float32))
def code(f):
    try: f()
    except fm.FmhipError as e: return e.code
    return 0
assert code(lambda: x.v2s0("ADD", y)) == E.ERR_SIZE_MISMATCH                 # checked by the front, at once
assert code(lambda: fm.init(0)) == E.ERR_INVALID_ARGUMENT
assert code(lambda: x.device_ptr()) == E.ERR_INVALID_ARGUMENT                # names ONE device
h = x.handle
z = x.v1s1("MULT_S", 2.0)
del x
assert z.to_float32()[0] == 2.0                                              # the operand was released after the operation was queued
import ctypes as C
assert fm.lib().fmhip_vec_release(C.c_int64(h)) == E.ERR_INVALID_HANDLE      # released already
st = fm.pool_stats(); assert st.n_kernel_launches > 0 and st.n_live_vectors >= 2
fm.shutdown()
fm.init(0)                                                                   # one device again, same process
assert fm.device_count() == 1
print("ok")
''' % {"root": ROOT})
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1000:] + r.stderr[-3000:]


def test_lmm_objective_evaluation_on_a_device_list():
    """lmm_hip --devices 0,0: fmhip_init_devices and nothing else changed in the driver (replication of the Jacobian batch, expectations in
    flight, values given up).  The model volatilities equal the one-device run's to 1e-12 (the expectations are sums over other blocks)."""
    def run(*extra):
        r = subprocess.run([LMM_HIP, "--paths", "20000", "--mode", "evaluate", "--evaluations", "8", "--jacobian-batch", "8", *extra], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    one = run()
    for devices in ("0,0", "0,0,0"):
        many = run("--devices", devices)
        a, b = np.array(one["model_volatility"]), np.array(many["model_volatility"])
        assert a.shape == b.shape and np.all(np.abs(a - b) <= 1e-12 * np.abs(a)), (devices, float(np.max(np.abs(a - b) / np.abs(a))))


def test_expectations_on_the_devices_of_a_device_list(tmp_path):
    """fmhip_reduce_moments_batch_devices / _batch_device / _device behind a device list.  The box has one GPU: the lists [0, 0] and
    [0, 0, 0] repeat a device index, so the shards' moments are combined on the HOST and copied back (fmhip_expectation_collective says 2,
    and why); with distinct devices the same call is one grouped RCCL all-gather and a combine kernel per device (rehearsed on the null
    device: tests/test_sanitizers_cpu.py).  Every device buffer must hold the bits fmhip_reduce_moments_batch returns on the host."""
    script = tmp_path / "collective.py"
    script.write_text(r'''
import ctypes as C, importlib, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
N = fm._native
for dev in ([0, 0], [0, 0, 0]):
    fm.init_devices(dev)
    D = len(dev)
    kind = C.c_int32(-1); why = C.create_string_buffer(256)
    N.check(fm.lib().fmhip_expectation_collective(C.byref(kind), why, 256))
    assert kind.value == 2 and b"repeats" in why.value, (kind.value, why.value)
    rng = np.random.default_rng(7)
    n = 100_003
    xs = [fm.DeviceVector.from_host(rng.normal(0.1 * k, 1.0, n).astype(np.float32)) for k in range(9)]
    prev = fm.set_fusion(True)
    vs = [x.v1s1("MULT_S", 1.5).v2s0("ADD", xs[0]).v1s0("ABS") for x in xs]             # pending: the launches that compute them take the moments along
    count = len(vs)
    handles = (C.c_int64 * count)(*[v.handle for v in vs])
    bufs = [torch.full((count * 4,), -1.0, dtype=torch.float64, device="cuda:0") for _ in range(D)]
    outs = (C.c_void_p * D)(*[b.data_ptr() for b in bufs])
    if D == 3: outs[1] = None                                                            # one device that does not want them
    N.check(fm.lib().fmhip_reduce_moments_batch_devices(handles, count, None, outs, D))
    fm.synchronize()
    host = (N.Moments * count)()
    N.check(fm.lib().fmhip_reduce_moments_batch(handles, count, None, host))
    want = np.frombuffer(host, dtype=np.float64)
    for d in range(D):
        got = bufs[d].cpu().numpy()
        if D == 3 and d == 1: assert (got == -1.0).all()
        else: assert got.tobytes() == want.tobytes(), (dev, d)
    first = torch.full((count * 4,), -1.0, dtype=torch.float64, device="cuda:0")
    N.check(fm.lib().fmhip_reduce_moments_batch_device(handles, count, None, C.c_void_p(first.data_ptr())))
    one = torch.full((4,), -1.0, dtype=torch.float64, device="cuda:0")
    N.check(fm.lib().fmhip_reduce_moments_device(C.c_int64(vs[3].handle), C.c_double(0.0), C.c_void_p(one.data_ptr())))
    fm.synchronize()
    assert first.cpu().numpy().tobytes() == want.tobytes() and one.cpu().numpy().tobytes() == want[12:16].tobytes()
    s = C.c_void_p(0)
    N.check(fm.lib().fmhip_get_stream_of(D - 1, C.byref(s))); assert s.value
    assert fm.lib().fmhip_get_stream_of(D, C.byref(s)) == N.ERR_INVALID_ARGUMENT
    assert fm.lib().fmhip_reduce_moments_batch_devices(handles, count, None, outs, D + 1) == N.ERR_INVALID_ARGUMENT
    fm.set_fusion(prev)
    del vs, xs
    fm.shutdown()
fm.init(0)                                                                               # one device: kind 0, the same entry points
kind = C.c_int32(-1)
N.check(fm.lib().fmhip_expectation_collective(C.byref(kind), None, 0)); assert kind.value == 0
x = fm.DeviceVector.from_host(np.arange(1000, dtype=np.float32))
buf = torch.zeros(4, dtype=torch.float64, device="cuda:0")
h = (C.c_int64 * 1)(x.handle); o = (C.c_void_p * 1)(buf.data_ptr())
N.check(fm.lib().fmhip_reduce_moments_batch_devices(h, 1, None, o, 1)); fm.synchronize()
assert buf.cpu().numpy()[0] == 499500.0
print("ok")
''' % {"root": ROOT})
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1000:] + r.stderr[-3000:]
