"""Summarise a rocprofv3 --pmc run of benchmarks/tier_pmc.py: per-kernel averages of every counter + derived figures."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
elems = float(sys.argv[2]) if len(sys.argv) > 2 else 64e6
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list); meta = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if int(r["Grid_Size"]) < 1_000_000: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        meta[k] = (r["VGPR_Count"], r["SGPR_Count"])
for k, c in acc.items():
    m = {n: sum(v[2:]) / max(1, len(v[2:])) for n, v in c.items()}       # skip the first two launches
    if not m.get("SQ_WAVES"): continue
    d = sorted(dur[k])[len(dur[k]) // 2]
    print(f"== {k[:90]}  vgpr={meta[k][0]} sgpr={meta[k][1]}  median {d:.1f} us")
    for n, v in sorted(m.items()): print(f"   {n:24s} {v:16.0f}")
    if "SQ_INSTS_VALU" in m: print(f"   VALU wave-instructions per element: {m['SQ_INSTS_VALU'] * 64 / elems:.1f}")
    if "GRBM_GUI_ACTIVE" in m: print(f"   effective clock: {m['GRBM_GUI_ACTIVE'] / 8 / d / 1e3:.2f} GHz (GRBM_GUI_ACTIVE / 8 / duration)")
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_WAVE_CYCLES" in m:
        print(f"   VALU active / wave cycles: {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        # quad-cycles summed over all SIMDs vs shader cycles summed over 8 XCDs
        print(f"   VALU busy fraction per SIMD: {m['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / (m['GRBM_GUI_ACTIVE'] / 8):.3f}")
    for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if w in m and "SQ_WAVE_CYCLES" in m: print(f"   {w} / SQ_WAVE_CYCLES: {m[w] / m['SQ_WAVE_CYCLES']:.3f}")
