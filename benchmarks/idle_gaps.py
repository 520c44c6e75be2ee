"""Device idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV, by length of the gap.
usage: idle_gaps.py <..._kernel_trace.csv> [label]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"{sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]}: {len(rows)} kernels, first start to last end {(t1 - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms")
count, total = collections.Counter(), collections.Counter()
end = int(rows[0]["End_Timestamp"])
for r in rows[1:]:
    s = int(r["Start_Timestamp"]); g = s - end
    if g > 0:
        k = "< 20 us" if g < 20e3 else "20-100 us" if g < 100e3 else "0.1-1 ms" if g < 1e6 else "1-10 ms" if g < 1e7 else ">= 10 ms"
        count[k] += 1; total[k] += g
    end = max(end, int(r["End_Timestamp"]))
for k in ("< 20 us", "20-100 us", "0.1-1 ms", "1-10 ms", ">= 10 ms"):
    print(f"    gaps {k:10s} {count[k]:6d}  {total[k] / 1e6:8.1f} ms")
