"""Shape of the specialised kernel (csrc/jit.cpp: jit_shape) measured on the bench program, stream S over 64 x 1M paths:
elements per lane and pass (FMHIP_JIT_ELEMS), interleaved elements of an exp / log body (FMHIP_JIT_GROUP), occupancy
hint for the register allocator (FMHIP_JIT_WAVES), software prefetch of the next pass (FMHIP_JIT_PREFETCH).
One fresh process per configuration (the knobs are read when the kernel source is generated); all on one box, in sequence.

    python benchmarks/jit_knobs.py [--json out.json] [--configs "8,4,0,0;4,4,0,0;..."]        (elems,group,waves,prefetch[,elements per workgroup])
"""
import importlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULT = "8,4,0,0;8,4,5,0;8,1,0,0;4,4,0,0;4,2,0,0;8,4,0,1;4,4,0,1;8,4,5,1;8,4,0,0"


def child():
    import torch
    fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
    fm.init(0)
    fm.set_jit(fm.JIT_SYNC)
    n, B = 1_000_000, 64
    bm = fm.BrownianMotionHip(fm.TimeDiscretization(0.0, B, 1.0), 3, n, 31415)
    rows = []
    for b in range(B):
        g = [bm.getBrownianIncrement(b, f) for f in range(3)]
        rows.append([g[0].mult(0.25).add(0.5).cap(1.0).floor(0.0).realizations, g[1].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations,
                     g[2].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations])
    del bm
    outs = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
    sys.path.insert(0, ROOT)
    import bench
    if os.environ.get("KNOB_FAST"): fm.set_math_mode(fm.MATH_FAST)
    drop = set(filter(None, os.environ.get("KNOB_DROP", "").split("+")))      # e.g. LOG+SQRT: those methods become abs()
    if os.environ.get("KNOB_NO_REDUCE") or drop:    # variants of the stream: without the fused reductions / without some methods
        p = fm.Program(3)
        x, y, z = 0, 1, 2
        f = lambda name, a: p.op("ABS" if name in drop else name, a)
        t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", x, s=4.0), s=2.0), y), z)
        u = f("SQRT", p.op("ABS", f("LOG", f("EXP", t))))
        v = p.op("ADDPRODUCT", p.op("FLOOR_S", p.op("CAP_S", u, s=1.5), s=0.25), y, z)
        w = p.op("CHOOSE", t, v, x)
        p.output(w)
        if not os.environ.get("KNOB_NO_REDUCE"): p.reduce(w)
        p.compile()
    else:
        p = bench.build_stream_s(fm)
    partial = torch.zeros(B * 4, dtype=torch.float64, device="cuda:0")

    def run(k):
        for _ in range(k):
            p.run_into(rows, outs, want_moments=False, device_moments=partial.data_ptr())

    run(30); fm.synchronize()
    best, vals = 1e9, []
    for _ in range(4):
        fm.profile_enable(True); run(100); ms, k = fm.profile_read(); fm.profile_enable(False)
        vals.append(ms / k * 1e3)
    # sustained: ~1.5 s of back-to-back launches, HIP events on the runtime stream around chunks of 50 (as bench.py's sustained leg)
    ext = torch.cuda.ExternalStream(fm.stream_ptr(), device=torch.device("cuda", 0))
    chunks = int(1.5 / (50 * vals[-1] * 1e-6))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    evs[0].record(ext)
    for c in range(chunks):
        run(50); evs[c + 1].record(ext)
    fm.synchronize()
    us = [evs[c].elapsed_time(evs[c + 1]) * 20.0 for c in range(chunks)]
    tail = us[len(us) // 2:]
    print(json.dumps({"tier": p.tier(), "us": [round(v, 1) for v in vals], "sustained_us": round(sum(us) / len(us), 1), "sustained_second_half_us": round(sum(tail) / len(tail), 1),
                      "moments": partial[:4].tolist()}), flush=True)


def main():
    if "--child" in sys.argv:
        return child()
    configs = DEFAULT
    if "--configs" in sys.argv:
        configs = sys.argv[sys.argv.index("--configs") + 1]
    out = []
    for c in configs.split(";"):
        e, g, w, pf, *rest = c.split(",")
        env = dict(os.environ, FMHIP_JIT_ELEMS=e, FMHIP_JIT_GROUP=g, FMHIP_JIT_WAVES=w, FMHIP_JIT_PREFETCH=pf)
        if rest: env["FMHIP_ELEMS_PER_BLOCK"] = rest[0]
        if len(rest) > 1 and "noreduce" in rest[1:]: env["KNOB_NO_REDUCE"] = "1"
        for r in rest[1:]:
            if r.startswith("drop="): env["KNOB_DROP"] = r[5:]
            if r == "fast": env["KNOB_FAST"] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        rec = {"elems": int(e), "group": int(g), "waves": int(w), "prefetch": int(pf), "elems_per_block": int(rest[0]) if rest else 8192, "reduce": not (len(rest) > 1 and "noreduce" in rest[1:]), "drop": [r[5:] for r in rest[1:] if r.startswith("drop=")], "fast": "fast" in rest[1:]}
        if line: rec.update(json.loads(line[-1]))
        else: rec["error"] = (r.stderr or r.stdout)[-400:]
        print(json.dumps(rec), flush=True)
        out.append(rec)
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
