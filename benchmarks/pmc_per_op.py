import importlib, sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
fm.init(0)
n, B = 1_000_000, 16
rows = [[fm.DeviceVector.filled(n, 0.5 + 0.001*b), fm.DeviceVector.filled(n, 1.0), fm.DeviceVector.filled(n, 1.25)] for b in range(B)]
outs = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
OPS = ["BASE", "ADD_S", "MULT", "DIV", "DIV_S", "SQRT", "EXP", "LOG", "CAP_S", "ADDPRODUCT", "CHOOSE", "ABS", "INVERT", "ACCRUE", "DISCOUNT", "SQUARED", "RED"]
for name in OPS:
    p = fm.Program(3); x, y, z = 0, 1, 2
    w = p.op("ADD", x, y)
    if name in ("ADD_S", "DIV_S", "CAP_S"): w = p.op(name, w, s=3.0)
    elif name in ("MULT", "DIV"): w = p.op(name, w, z)
    elif name in ("SQRT", "EXP", "LOG", "ABS", "INVERT", "SQUARED"): w = p.op(name, w)
    elif name in ("ADDPRODUCT", "CHOOSE"): w = p.op(name, w, y, z)
    elif name in ("ACCRUE", "DISCOUNT"): w = p.op(name, w, z, s=0.5)
    w = p.op("ADD", w, z)
    p.output(w)
    if name == "RED": p.reduce(w)
    p.compile()
    for _ in range(2): p.run_into(rows, outs, want_moments=False)
fm.synchronize()
print("ORDER", OPS)
