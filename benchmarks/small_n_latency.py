"""Latency of the reference README's example chain `x.add(4).div(2).exp().getAverage()` as a function of the
path count, on the engine (lazy front-end on: the chain is one launch, the expectation a second one + the read-back) and on the CPU
twin (oracle, one loop + one fresh array per method).  The reference puts its CPU/GPU break-even at about 5 000 paths (README.md:26)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
fm.init(0); fm.set_fusion(True)
f = fm.RandomVariableHipFactory()
out = []
for n in (100, 1000, 5000, 20000, 100000, 1000000):
    xd = oracle.java_random_doubles(31415, n)
    x = f.createRandomVariable(0.0, xd)
    xf = oracle.f_from_double(xd)
    reps = 2000 if n <= 100000 else 300
    for _ in range(50): x.add(4.0).div(2.0).exp().getAverage()
    fm.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): g = x.add(4.0).div(2.0).exp().getAverage()
    gpu_us = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    creps = max(20, reps // 10)
    for _ in range(creps): c = oracle.f_average(oracle.f_v1s0("EXP", oracle.f_v1s1("DIV_S", oracle.f_v1s1("ADD_S", xf, 4.0), 2.0)))
    cpu_us = (time.perf_counter() - t0) / creps * 1e6
    out.append({"paths": n, "engine_us": round(gpu_us, 1), "cpu_twin_us": round(cpu_us, 1), "same_average": abs(g - c) <= 1e-12 * abs(c)})
    print(out[-1], flush=True)
print(json.dumps(out))
