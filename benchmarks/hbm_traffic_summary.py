"""profiles/roundNN_hbm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

    python benchmarks/hbm_traffic_summary.py <fetch_dir> <write_dir> > profiles/round02_hbm_traffic.json

Picks the dominant program kernel of the run (either tier; largest grid: stream S over 64 x 1M paths), averages the counter over its launches
and applies the gfx950 corrections of MI355X_MICROARCH.md §HBM: counters are in KiB; FETCH_SIZE is doubled."""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(root, counter):
    vals = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("fm_jit_" in r["Kernel_Name"] or "fm_program_kernel" in r["Kernel_Name"]):
                vals[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    key = max(vals, key=lambda k: (k[1], len(vals[k])))
    v = vals[key]
    return key[0], {"launches": len(v), "avg_counter_KiB": sum(v) / len(v), "min": min(v), "max": max(v)}


fetch_kernel, fetch = collect(sys.argv[1], "FETCH_SIZE")
write_kernel, write = collect(sys.argv[2], "WRITE_SIZE")
assert fetch_kernel == write_kernel, (fetch_kernel, write_kernel)
n, B = 1_000_000, 64
rd = fetch["avg_counter_KiB"] * 1024 * 2
wr = write["avg_counter_KiB"] * 1024
alg = 4.0 * (3 + 1) * n * B
print(json.dumps({
    "kernel": fetch_kernel,
    "workload": {"paths": n, "batch": B, "stream": "S (12 ops, 3 in, 1 out) + fused reduction"},
    "method": "two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `python3 bench.py --workload stream --steps 5 --warmup 2 "
              "--sustained-seconds 0 --no-cpu-baseline`; FETCH_SIZE doubled per the gfx950 correction (128-B requests tallied at 64 B, "
              "MI355X_MICROARCH.md §HBM), WRITE_SIZE as is; counters are in KiB",
    "raw": {"FETCH_SIZE": fetch, "WRITE_SIZE": write},
    "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg}, indent=1))
