"""Per-kernel summary of the rocprofv3 --pmc passes of benchmarks/pmc_lmm_kernels.sh: for every kernel that took more than 1 % of the
device time, the sum of each counter over its launches, per launch-microsecond figures and the ratios the guide names
(MI355X_MICROARCH.md, rocprofv3 PMC slots: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ≈ WAVE_CYCLES, all in quad-cycles)."""
import csv, glob, os, sys
from collections import defaultdict

counters = defaultdict(lambda: defaultdict(float)); time_us = defaultdict(float); launches = defaultdict(int); regs = {}
for root in sys.argv[1:]:
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            counters[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (root, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                if root == sys.argv[1]:
                    time_us[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; launches[k] += 1
            regs[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"))
total = sum(time_us.values())
print(f"{sum(launches.values())} dispatches, {total / 1e3:.1f} ms of kernel time in the first pass")
for k in sorted(time_us, key=lambda k: -time_us[k]):
    if time_us[k] < 0.01 * total: continue
    c = counters[k]
    print(f"== {k[:100]}  vgprs {regs[k][0]} sgprs {regs[k][1]}  {launches[k]} launches  {time_us[k] / 1e3:.2f} ms ({100 * time_us[k] / total:.1f} %)")
    for n in sorted(c): print(f"   {n:24s} {c[n]:18.0f}")
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"):
            if n in c: print(f"   {n} / SQ_WAVE_CYCLES = {c[n] / wc:.3f}")
    if c.get("SQ_WAVES"):
        if "SQ_INSTS_VALU" in c: print(f"   VALU instructions per wave: {c['SQ_INSTS_VALU'] / c['SQ_WAVES']:.0f}")
        if "SQ_INSTS_SALU" in c: print(f"   SALU instructions per wave: {c['SQ_INSTS_SALU'] / c['SQ_WAVES']:.0f}")
        if wc: print(f"   wave lifetime: {wc * 4 / c['SQ_WAVES']:.0f} cycles (SQ_WAVE_CYCLES x 4 / SQ_WAVES)")
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
        print(f"   SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {c['SQ_ACTIVE_INST_VALU'] / c['SQ_BUSY_CYCLES']:.3f}")
    if c.get("GRBM_GUI_ACTIVE"):
        print(f"   effective clock {c['GRBM_GUI_ACTIVE'] / 8 / time_us[k] / 1e3:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the first pass)")
