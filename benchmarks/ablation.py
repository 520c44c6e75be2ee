import importlib, sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
fm.init(0)
n, B = 1_000_000, 64
bm = fm.BrownianMotionHip(fm.TimeDiscretization(0.0, B, 1.0), 3, n, 31415)
rows = []
for b in range(B):
    g = [bm.getBrownianIncrement(b, f) for f in range(3)]
    rows.append([g[0].mult(0.25).add(0.5).cap(1.0).floor(0.0).realizations, g[1].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations, g[2].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations])
del bm
outs = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
def prog(kind, red=True):
    p = fm.Program(3); x,y,z = 0,1,2
    if kind == "copy1": w = p.op("ADD", x, y); w = p.op("ADD", w, z)
    elif kind == "S":
        t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", x, s=4.0), s=2.0), y), z)
        u = p.op("SQRT", p.op("ABS", p.op("LOG", p.op("EXP", t))))
        v = p.op("ADDPRODUCT", p.op("FLOOR_S", p.op("CAP_S", u, s=1.5), s=0.25), y, z)
        w = p.op("CHOOSE", t, v, x)
    elif kind == "S_noexplog":
        t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", x, s=4.0), s=2.0), y), z)
        u = p.op("SQRT", p.op("ABS", p.op("ABS", p.op("ABS", t))))
        v = p.op("ADDPRODUCT", p.op("FLOOR_S", p.op("CAP_S", u, s=1.5), s=0.25), y, z)
        w = p.op("CHOOSE", t, v, x)
    elif kind == "simple12":
        w = p.op("ADD", x, y)
        for i in range(11): w = p.op("MULT" if i % 2 else "ADD", w, z)
    elif kind == "simple24":
        w = p.op("ADD", x, y)
        for i in range(23): w = p.op("MULT" if i % 2 else "ADD", w, z)
    elif kind == "exp1": w = p.op("EXP", p.op("ADD", x, y)); w = p.op("ADD", w, z)
    elif kind == "log1": w = p.op("LOG", p.op("ADD", x, y)); w = p.op("ADD", w, z)
    elif kind == "div1": w = p.op("DIV", p.op("ADD", x, y), z)
    elif kind == "sqrt1": w = p.op("SQRT", p.op("ADD", x, y)); w = p.op("ADD", w, z)
    p.output(w)
    if red: p.reduce(w)
    return p.compile()
import sys as _s
MODES = [("exact", fm.MATH_EXACT), ("fast", fm.MATH_FAST)]
TIERS = [("interpreter", fm.JIT_OFF), ("specialised", fm.JIT_SYNC)]
for (mname, mode), (tname, tier) in [(m, t) for m in MODES for t in TIERS]:
  fm.set_math_mode(mode)
  fm.set_jit(tier)
  print("math mode", mname, "| tier", tname)
  for kind, red in [("copy1", False), ("copy1", True), ("simple12", False), ("simple24", False), ("div1", False), ("sqrt1", False), ("exp1", False), ("log1", False), ("S_noexplog", True), ("S", True), ("S", False)]:
    if mname == "fast" and kind not in ("exp1", "log1", "S"): continue
    p = prog(kind, red)
    for _ in range(3): p.run_into(rows, outs, want_moments=False)
    fm.profile_enable(True)
    for _ in range(10): p.run_into(rows, outs, want_moments=False)
    ms, k = fm.profile_read(); fm.profile_enable(False)
    us = ms / k * 1e3
    print(f"{kind:12s} red={red!s:5s} {us:8.1f} us  {16.0*n*B/us/1e3:8.1f} GB/s", flush=True)
