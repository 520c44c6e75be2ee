"""HBM traffic of the whole LMM objective-evaluation op stream against its algorithmic bytes: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
over `lmm_hip --mode evaluate --evaluations 16 --jacobian-batch 8 --warmup-evaluations 8`, summed over every fm_jit_* / fm_program_kernel launch
and corrected as MI355X_MICROARCH.md §HBM prescribes (counters in KiB, FETCH_SIZE doubled on gfx950); the algorithmic bytes are the engine's own
count for the same run (fmhip_traffic_stats: 4 B x paths x (vectors read + vectors stored) per launch), from the driver's JSON line.

    python benchmarks/lmm_hbm_traffic.py <fetch_dir> <write_dir> <driver line .json>"""
import csv, glob, json, os, sys


def total(root, counter):
    s, n = 0.0, 0
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("fm_jit_" in r["Kernel_Name"] or "fm_program_kernel" in r["Kernel_Name"] or "fm_gather" in r["Kernel_Name"]):
                s += float(r["Counter_Value"]); n += 1
    return s, n


fetch, nf = total(sys.argv[1], "FETCH_SIZE")
write, nw = total(sys.argv[2], "WRITE_SIZE")
line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
rd, wr, alg = fetch * 1024 * 2, write * 1024, float(line["algorithmic_bytes"])
print(json.dumps({"workload": "LMM objective evaluations, 1 M paths: 24 evaluations in lock-step batches of 8 (the first batch meets every shape for the first time)",
                  "launches_counted": [nf, nw], "read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr, "algorithmic_bytes_engine_count": alg,
                  "traffic_over_algorithmic": (rd + wr) / alg,
                  "method": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE); counters in KiB, FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md §HBM)"}, indent=1))
