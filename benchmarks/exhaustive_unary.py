"""Exhaustive parity of the unary opcodes over ALL 2^32 fp32 bit patterns: HIP engine (through the C-ABI) against the oracle
(C restatement of RandomVariableFromFloatArray: `(float)exp((double)x)` etc.).  Not a pytest (≈ 2-4 minutes on the GPU box):

    python benchmarks/exhaustive_unary.py [--ops EXP,LOG,SQRT,INVERT,POW_S:0.5,CAP_S:nan,DIV_S:3] [--fast] [--json out.json]

--fast measures FMHIP_MATH_FAST (hardware exp/log): there the figure of interest is max_ulp (stated bound: 2).

Reports, per opcode: elements whose bits differ, the largest difference in fp32 ulp, and NaN-ness mismatches."""
import importlib, json, os, sys, time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHUNK = 1 << 24


def oracle_chunk(args):
    op, start = args
    import oracle
    x = np.arange(start, start + CHUNK, dtype=np.uint64).astype(np.uint32).view(np.float32)
    with np.errstate(all="ignore"):
        if ":" in op:                               # "<scalar opcode>:<scalar>", e.g. POW_S:0.5, CAP_S:nan, DIV_S:3
            return oracle.f_v1s1(op.split(":")[0], x, float(op.split(":")[1]))
        return oracle.f_v1s0(op, x)


def ulp_index(a):
    i = a.view(np.int32).astype(np.int64)
    return np.where(i < 0, -(i & 0x7fffffff), i)


def main():
    ops = "EXP,LOG,SQRT,INVERT".split(",")
    if "--ops" in sys.argv:
        ops = sys.argv[sys.argv.index("--ops") + 1].split(",")
    workers = min(16, os.cpu_count() or 1)
    pool = ProcessPoolExecutor(workers)            # forked BEFORE the GPU is touched
    list(pool.map(abs, range(workers)))
    fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
    fm.init(0)
    fast = "--fast" in sys.argv
    if fast:
        fm.set_math_mode(fm.MATH_FAST)
    report = {}
    for op in ops:
        t0 = time.time()
        differ = nan_mismatch = over2 = 0
        worst = 0
        examples = []
        starts = list(range(0, 1 << 32, CHUNK))
        for start, want in zip(starts, pool.map(oracle_chunk, [(op, s) for s in starts])):
            x = np.arange(start, start + CHUNK, dtype=np.uint64).astype(np.uint32).view(np.float32)
            v = fm.DeviceVector.from_host(x)
            got = (v.v1s1(op.split(":")[0], float(op.split(":")[1])) if ":" in op else v.v1s0(op)).to_float32()
            gn, wn = np.isnan(got), np.isnan(want)
            nan_mismatch += int((gn != wn).sum())
            bad = (got.view(np.uint32) != want.view(np.uint32)) & ~(gn & wn)
            k = int(bad.sum())
            if k:
                differ += k
                d = np.abs(ulp_index(got[bad]) - ulp_index(want[bad]))
                worst = max(worst, int(d.max()))
                over2 += int((d > 2).sum())
                if len(examples) < 5:
                    j = int(np.flatnonzero(bad)[0])
                    examples.append({"x_bits": hex(int(x.view(np.uint32)[j])), "x": float(x[j]), "got": float(got[j]), "want": float(want[j])})
        report[op] = {"inputs": 1 << 32, "differ": differ, "fraction": differ / float(1 << 32), "max_ulp": worst,
                      "more_than_2_ulp": over2, "nan_mismatch": nan_mismatch, "examples": examples, "seconds": time.time() - t0}
        print(op, json.dumps(report[op]), flush=True)
    pool.shutdown()
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump({"what": "HIP engine vs oracle over all 2^32 fp32 inputs (%s math mode)" % ("fast" if fast else "exact"), "results": report}, fh, indent=1)


if __name__ == "__main__":
    main()
