import importlib, sys, numpy as np
sys.path.insert(0, '.')
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
fm.init(0)
n, B = 1_000_000, 64
rows = [[fm.DeviceVector.filled(n, 0.5 + 0.001*b), fm.DeviceVector.filled(n, 1.0), fm.DeviceVector.filled(n, 1.25)] for b in range(B)]
outs = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
def prog(kind, red):
    p = fm.Program(3); x,y,z = 0,1,2
    if kind == "copy1": w = p.op("ADD", x, y); w = p.op("ADD", w, z)
    elif kind == "simple24":
        w = p.op("ADD", x, y)
        for i in range(23): w = p.op("MULT" if i % 2 else "ADD", w, z)
    elif kind == "S":
        t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", x, s=4.0), s=2.0), y), z)
        u = p.op("SQRT", p.op("ABS", p.op("LOG", p.op("EXP", t))))
        v = p.op("ADDPRODUCT", p.op("FLOOR_S", p.op("CAP_S", u, s=1.5), s=0.25), y, z)
        w = p.op("CHOOSE", t, v, x)
    p.output(w)
    if red: p.reduce(w)
    return p.compile()
for kind, red in [("copy1", False), ("simple24", False), ("S", False), ("S", True)]:
    p = prog(kind, red)
    for _ in range(3): p.run_into(rows, outs, want_moments=False)
fm.synchronize()
