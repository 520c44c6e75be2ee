"""Stream S (bench.py's program: 12 fused path-ops + 4 fused reductions) across vector sizes and batch counts, specialised
tier: where the launch overhead the reference names as its bottleneck (README.md:24-28: ≫ kernel time at N <= 1e5) stops
mattering when a whole chain over a whole batch is ONE launch.

    python benchmarks/stream_sizes.py [--json out.json]"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
fm.init(0)
fm.set_jit(fm.JIT_SYNC)


def program():
    p = fm.Program(3)
    x, y, z = 0, 1, 2
    t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", x, s=4.0), s=2.0), y), z)
    u = p.op("SQRT", p.op("ABS", p.op("LOG", p.op("EXP", t))))
    v = p.op("ADDPRODUCT", p.op("FLOOR_S", p.op("CAP_S", u, s=1.5), s=0.25), y, z)
    w = p.op("CHOOSE", t, v, x)
    p.output(w); p.reduce(w)
    return p.compile()


results = []
p = program()
for n, B in ((1_000, 1), (1_000, 64), (10_000, 1), (10_000, 64), (100_000, 1), (100_000, 64), (1_000_000, 1), (1_000_000, 2), (1_000_000, 4), (1_000_000, 8), (1_000_000, 16), (1_000_000, 64),
             (1 << 20, 64), (1 << 26, 1), (8_000_000, 8)):
    bm = fm.BrownianMotionHip(fm.TimeDiscretization(0.0, B, 1.0), 3, n, 31415)
    rows = []
    for b in range(B):
        g = [bm.getBrownianIncrement(b, f) for f in range(3)]
        rows.append([g[0].mult(0.25).add(0.5).cap(1.0).floor(0.0).realizations, g[1].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations,
                     g[2].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations])
    del bm
    outs = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
    for _ in range(3):
        p.run_into(rows, outs, want_moments=False)
    fm.profile_enable(True)
    for _ in range(10):
        p.run_into(rows, outs, want_moments=False)
    ms, k = fm.profile_read()
    fm.profile_enable(False)
    us = ms / k * 1e3
    r = {"n": n, "batch": B, "kernel_us": us, "GBps": 16.0 * n * B / us / 1e3, "path_ops_per_s": 12.0 * n * B / (us * 1e-6)}
    results.append(r)
    print(f"N={n:9d} x{B:3d}  {us:9.1f} us  {r['GBps']:7.0f} GB/s  {r['path_ops_per_s']:.3e} path-ops/s", flush=True)
    del rows, outs
    fm.purge()
if "--json" in sys.argv:
    json.dump({"workload": "stream S + fused reductions, specialised tier", "results": results}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
