"""POW_S with a wave-uniform exponent: device time of the opcode alone (64 rows x 1 M paths, one read + one write per path) per exponent,
both tiers — the exponents with code of their own (fm_device_math.hpp: pow_all) against the library path."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
fm.init(0)
n, B = 1_000_000, 64
rng = np.random.default_rng(1)
rows = [[fm.DeviceVector.from_host(rng.uniform(0.25, 1.75, n).astype(np.float32))] for _ in range(B)]
outs = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
res = {}
for tier, mode in (("specialised", fm.JIT_SYNC), ("interpreter", fm.JIT_OFF)):
    prev = fm.set_jit(mode)
    for s in (2.0, 0.5, -1.0, 3.0, 4.0, 1.5, 1.25):
        p = fm.Program(1); v = p.op("POW_S", 0, s=s); p.output(v); prog = p.compile()
        for _ in range(5): prog.run_into(rows, outs, want_moments=False)
        fm.profile_enable(True)
        for _ in range(20): prog.run_into(rows, outs, want_moments=False)
        ms, count = fm.profile_read(); fm.profile_enable(False)
        us = ms * 1e3 / count
        res[f"{tier} s={s}"] = {"us": us, "GBps": 8.0 * n * B / (us * 1e-6) / 1e9, "frac": 8.0 * n * B / (us * 1e-6) / 1e9 / 8000.0}
    fm.set_jit(prev)
print(json.dumps(res, indent=1))
