# The evidence of the end of round 5 in one GPU call: rocprofv3 summaries (profile_round.sh), SQ counters, by-shape tables with the
# engine's switches on and off, the default bench line.  bash benchmarks/round5/final_evidence.sh <tag>  →  gpurun_out/<tag>/
TAG=${1:-r5q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash benchmarks/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
bash benchmarks/pmc_lmm_kernels.sh $TAG > $OUT/pmc.log 2>&1
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  for mode in native hintfree; do
    extra=""; [ $mode = hintfree ] && extra="--finmath-like"
    FMHIP_MERGE_CHAINS=$1 FMHIP_COMMON_ROWS=$2 timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 $extra > $OUT/${mode}_merge$1_common$2_line.json 2>/dev/null
    FMHIP_MERGE_CHAINS=$1 FMHIP_COMMON_ROWS=$2 FMHIP_PROFILE_DUMP=1 timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 $extra --profile > $OUT/${mode}_merge$1_common$2_profiled_line.json 2> $OUT/${mode}_merge$1_common$2_by_shape.txt
  done
done
python3 - $OUT > $OUT/ab_summary.txt <<'PY'
import json, sys
out = sys.argv[1]
for mode in ("native", "hintfree"):
    for m, c in ((1, 1), (1, 0), (0, 0)):
        d = json.loads(open(f"{out}/{mode}_merge{m}_common{c}_line.json").read().strip().splitlines()[-1]); e = d.get("engine", {})
        p = json.loads(open(f"{out}/{mode}_merge{m}_common{c}_profiled_line.json").read().strip().splitlines()[-1])
        print("%-8s merged chains %d common rows %d: %.3f s, %d launches, %.2f TB, %.3e path-ops, kernel time %.1f ms (%.0f GB/s), mean deviation %.6e, merged %s/%s, common rows %s" % (
            mode, m, c, d["seconds"], d["kernel_launches"], d["algorithmic_bytes"] / 1e12, d["path_ops"], p["kernel_ms_total"], p["achieved_GBps"], d["mean_deviation"], e.get("merged_launches"), e.get("merged_chains"), e.get("common_rows")))
PY
cat $OUT/ab_summary.txt
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc $?"
