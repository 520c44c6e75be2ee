"""Row-table ring: a program relaunched with an identical table after the ring has wrapped must upload the table again
(the device copy it remembers has been overwritten by other launches).  The ring size is fixed at fmhip_init, so the case
runs in a process of its own with FMHIP_RING_BYTES=16384 (a wrap every handful of launches instead of every few thousand)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import importlib, sys
import numpy as np
sys.path.insert(0, %(root)r)
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
import oracle as o
fm.init(0)
n, B = 3001, 90                      # 90 rows x 5 words > the 368 words that travel in the kernel arguments: table path

def program(scalar):
    p = fm.Program(3)
    t = p.op("ADDPRODUCT", p.op("MULT_S", 0, s=scalar), 1, 2)
    p.output(p.op("SUB", t, 0))
    return p.compile()

def host_rows(seed):
    return [[o.f_from_double(o.java_random_doubles(seed + 3 * b + k, n) + 0.25 * k) for k in range(3)] for b in range(B)]

def want(rows, scalar):
    return [o.f_v2s0("SUB", o.f_v3s0("ADDPRODUCT", o.f_v1s1("MULT_S", r[0], scalar), r[1], r[2]), r[0]) for r in rows]

for mode in (fm.JIT_OFF, fm.JIT_SYNC):
    fm.set_jit(mode)
    P, Q = program(1.5), program(-0.75)
    ha, hb = host_rows(1000), host_rows(5000)
    A = [[fm.DeviceVector.from_host(v) for v in r] for r in ha]
    Bv = [[fm.DeviceVector.from_host(v) for v in r] for r in hb]
    out = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
    P.run_into(A, out)                                   # uploads P's table; P remembers the device copy
    wa = want(ha, 1.5)
    for b in range(B):
        assert (out[b][0].to_float32().view(np.uint32) == wa[b].view(np.uint32)).all()
    for k in range(12):                                  # 12 x 3.6 KB of other tables through a 16 KB ring: it wraps, P's slot is overwritten
        rows = Bv[k:] + Bv[:k]                           # a different table every time
        Q.run_into(rows, out)
    P.run_into(A, out)                                   # identical table as the first launch
    for b in range(B):
        got = out[b][0].to_float32()
        assert (got.view(np.uint32) == wa[b].view(np.uint32)).all(), (mode, b)
    P.run_into(A, out)                                   # and the steady-state reuse right after an upload still works
    for b in range(0, B, 17):
        assert (out[b][0].to_float32().view(np.uint32) == wa[b].view(np.uint32)).all()
print("ring wrap ok")
'''


def test_relaunch_after_ring_wrap_uploads_the_table_again():
    env = dict(os.environ, FMHIP_RING_BYTES="16384")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ring wrap ok" in r.stdout
