"""Durations and start-to-start / end-to-start gaps of consecutive kernels in a rocprofv3 --kernel-trace CSV, by kernel name and grid size.
usage: kernel_gaps.py <..._kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = defaultdict(lambda: {"dur": [], "gap": [], "period": []})
prev = None
for r in rows:
    key = (r["Kernel_Name"][:40], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g[key]["dur"].append(e - s)
    if prev and prev[0] == key:
        g[key]["gap"].append(s - prev[2]); g[key]["period"].append(s - prev[1])
    prev = (key, s, e)
med = lambda v: sorted(v)[len(v) // 2] / 1e3 if v else float("nan")
print(f"{'kernel':42s} {'wgs':>7s} {'rows':>5s} {'launches':>8s} {'duration us':>12s} {'end-to-start us':>16s} {'start-to-start us':>18s}   (medians)")
for k, v in sorted(g.items(), key=lambda kv: -len(kv[1]["dur"])):
    if len(v["dur"]) < 20: continue
    print(f"{k[0]:42s} {k[1]:7d} {k[2]:5d} {len(v['dur']):8d} {med(v['dur']):12.2f} {med(v['gap']):16.2f} {med(v['period']):18.2f}")
