# Round 5, second half: the evidence behind DESIGN.md §4.6 — run from the repo root on a GPU box, writes gpurun_out/$1
set -o pipefail
OUT=gpurun_out/${1:-r5f}
mkdir -p $OUT
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
# hint-free caller: speculation window with merged families (FMHIP_SPECULATE_PENDING)
{
for v in 5000 3000 8000 12000 20000 5000; do
FMHIP_SPECULATE_PENDING=$v timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --finmath-like | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('engine',{}); print('speculate after $v methods: %.3f s, %d launches, merged %s/%s, interp %s, mean dev %.6e' % (d['seconds'], d['kernel_launches'], e.get('merged_launches'), e.get('merged_chains'), e.get('interpreter_launches'), d['mean_deviation']))"
done
} > $OUT/speculate_sweep.txt 2>&1
cat $OUT/speculate_sweep.txt
# per-shape tables of the native and the hint-free calibration, merged and one launch per shape
for M in 1 0; do
  for mode in native hintfree; do
    extra=""; [ $mode = hintfree ] && extra="--finmath-like"
    FMHIP_MERGE_CHAINS=$M FMHIP_PROFILE_DUMP=1 timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 $extra --profile > $OUT/${mode}_merge${M}_line.json 2> $OUT/${mode}_merge${M}_by_shape.txt
  done
done
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
for mode in ("native", "hintfree"):
    for M in (1, 0):
        d = json.loads(open(f"{out}/{mode}_merge{M}_line.json").read().strip().splitlines()[-1]); e = d.get("engine", {})
        print(mode, "merge", M, "%.3f s profiled, %d launches, %.2f TB, kernel %.1f ms, %.0f GB/s, mean dev %.6e, merged %s/%s" % (d["seconds"], d["kernel_launches"], d["algorithmic_bytes"] / 1e12, d["kernel_ms_total"], d["achieved_GBps"], d["mean_deviation"], e.get("merged_launches"), e.get("merged_chains")))
PY
