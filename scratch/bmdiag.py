import importlib, sys, numpy as np
sys.path.insert(0, '.')
fm = importlib.import_module("finmath-lib-cuda-extensions_amd"); import oracle
fm.init(0)
td = fm.TimeDiscretization(0.0, 2, 0.1)
n=5000
bm = fm.BrownianMotionHip(td, 2, n, 1234)
want = oracle.bm_generate(1234, [td.getTimeStep(0), td.getTimeStep(1)], 2, n)
for t in range(2):
  for f in range(2):
    got = bm.getBrownianIncrement(t,f).realizations.to_float32()
    bad = np.flatnonzero(got.view(np.uint32) != want[t][f].view(np.uint32))
    print(t,f,'mismatch',bad.size, bad[:10], got[bad[:5]], want[t][f][bad[:5]])
