"""Stream S (bench.py workload: 64 x 1M paths) on one execution tier, a few launches — the target of PMC passes:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY \
        SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d <out> -o run --output-format csv -- \
        python3 $GRAFT_REPO_ROOT/benchmarks/tier_pmc.py [interp|jit] [exact|fast]

and summarise with benchmarks/pmc_summary.py <out>."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
tier = sys.argv[1] if len(sys.argv) > 1 else "jit"
math = sys.argv[2] if len(sys.argv) > 2 else "exact"
fm.init(0)
fm.set_jit(fm.JIT_SYNC if tier == "jit" else fm.JIT_OFF)
if math == "fast":
    fm.set_math_mode(fm.MATH_FAST)
n, B = 1_000_000, 64
bm = fm.BrownianMotionHip(fm.TimeDiscretization(0.0, B, 1.0), 3, n, 31415)
rows = []
for b in range(B):
    g = [bm.getBrownianIncrement(b, f) for f in range(3)]
    rows.append([g[0].mult(0.25).add(0.5).cap(1.0).floor(0.0).realizations,
                 g[1].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations,
                 g[2].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations])
del bm
outs = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
p = fm.Program(3)
x, y, z = 0, 1, 2
t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", x, s=4.0), s=2.0), y), z)
u = p.op("SQRT", p.op("ABS", p.op("LOG", p.op("EXP", t))))
v = p.op("ADDPRODUCT", p.op("FLOOR_S", p.op("CAP_S", u, s=1.5), s=0.25), y, z)
w = p.op("CHOOSE", t, v, x)
p.output(w); p.reduce(w)
p.compile()
for _ in range(12):
    p.run_into(rows, outs, want_moments=False)
fm.synchronize()
print("tier", p.tier(), "math", math)
