"""BASELINE.json configs[1] / SURVEY.md §8(d) config 2, parts (A) and (C): every opcode ALONE and the reductions alone, on
both execution tiers, as device time and algorithmic GB/s (4 B x N x (inputs + outputs); a stand-alone reduction 4 B/path).

    python benchmarks/config2_sweep.py [--json out.json] [--ops LOG,CAP_S,moments]

Three shapes: 64 independent tuples of N = 1 000 000 and of N = 2^20 per launch (working set 0.5-1 GB >> the 256 MB Infinity
Cache, i.e. the HBM-labelled figure) and one tuple of N = 2^26."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
fm.init(0)

OPS = [("SQUARED", 1, 0), ("SQRT", 1, 0), ("EXP", 1, 0), ("LOG", 1, 0), ("INVERT", 1, 0), ("ABS", 1, 0),
       ("CAP_S", 1, 1), ("FLOOR_S", 1, 1), ("ADD_S", 1, 1), ("SUB_S", 1, 1), ("BUS_S", 1, 1), ("MULT_S", 1, 1), ("DIV_S", 1, 1), ("VID_S", 1, 1), ("POW_S", 1, 1),
       ("CAP", 2, 0), ("FLOOR", 2, 0), ("ADD", 2, 0), ("SUB", 2, 0), ("MULT", 2, 0), ("DIV", 2, 0),
       ("ACCRUE", 2, 1), ("DISCOUNT", 2, 1), ("ADDPRODUCT_VS", 2, 1),
       ("ADDPRODUCT", 3, 0), ("ADDRATIO", 3, 0), ("SUBRATIO", 3, 0), ("CHOOSE", 3, 0)]


def inputs(n, B):
    bm = fm.BrownianMotionHip(fm.TimeDiscretization(0.0, B, 1.0), 3, n, 31415)
    rows = []
    for b in range(B):
        g = [bm.getBrownianIncrement(b, f) for f in range(3)]
        rows.append([g[0].mult(0.25).add(0.5).cap(1.0).floor(0.01).realizations,
                     g[1].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations,
                     g[2].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations])
    return rows


def timed(p, rows, outs, reps):
    for _ in range(2):
        p.run_into(rows, outs, want_moments=False)
    fm.profile_enable(True)
    for _ in range(reps):
        p.run_into(rows, outs, want_moments=False)
    ms, k = fm.profile_read()
    fm.profile_enable(False)
    return ms / k * 1e3


def main():
    results = []
    only = set(sys.argv[sys.argv.index("--ops") + 1].split(",")) if "--ops" in sys.argv else None
    for n, B, reps in ((1_000_000, 64, 6), (1 << 20, 64, 6), (1 << 26, 1, 6)):       # the three sizes SURVEY.md §8(d) config 2 names
        rows = inputs(n, B)
        outs = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
        for tname, tier in (("interpreter", fm.JIT_OFF), ("specialised", fm.JIT_SYNC)):
            fm.set_jit(tier)
            for name, nvec, has_s in OPS:
                if only is not None and name not in only: continue
                p = fm.Program(nvec)
                w = p.op(name, *range(nvec), s=1.25) if has_s else p.op(name, *range(nvec))
                p.output(w); p.compile()
                us = timed(p, [r[:nvec] for r in rows], outs, reps)
                gb = 4.0 * (nvec + 1) * n * B / us / 1e3
                results.append({"part": "A", "op": name, "n": n, "batch": B, "tier": tname, "us": us, "GBps": gb})
                print(f"A  {name:14s} N={n:9d} x{B:3d} {tname:12s} {us:9.1f} us {gb:8.0f} GB/s", flush=True)
            if only is not None and "moments" not in only: continue
            p = fm.Program(1); p.reduce(0); p.compile()
            us = timed(p, [r[:1] for r in rows], [[] for _ in range(B)], reps)
            gb = 4.0 * n * B / us / 1e3
            results.append({"part": "C", "op": "moments", "n": n, "batch": B, "tier": tname, "us": us, "GBps": gb})
            print(f"C  {'sum,sumsq,min,max':14s} N={n:9d} x{B:3d} {tname:12s} {us:9.1f} us {gb:8.0f} GB/s", flush=True)
        del rows, outs
        fm.purge()
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump({"workload": "config 2 (A) each opcode alone, (C) reductions alone", "results": results}, fh, indent=1)


main()
