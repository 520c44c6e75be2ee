set -o pipefail
mkdir -p gpurun_out/r5m
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
timeout -k 10 400 python -m pytest tests/test_gpu_merged_chains.py -x -q > gpurun_out/r5m/test_merged.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r5m/test_merged.txt; tail -5 gpurun_out/r5m/test_merged.txt
for M in 1 0; do
  for mode in native hintfree; do
    extra=""; [ $mode = hintfree ] && extra="--finmath-like"
    FMHIP_MERGE_CHAINS=$M timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 $extra > gpurun_out/r5m/${mode}_m${M}.json 2> gpurun_out/r5m/${mode}_m${M}.err
    FMHIP_MERGE_CHAINS=$M FMHIP_PROFILE_DUMP=1 timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 $extra --profile > gpurun_out/r5m/${mode}_m${M}_prof.json 2> gpurun_out/r5m/${mode}_m${M}_prof.txt
  done
done
python3 - <<'PY'
import json
for mode in ("native","hintfree"):
    for M in (1,0):
        for suffix in ("", "_prof"):
            try:
                d=json.loads(open(f"gpurun_out/r5m/{mode}_m{M}{suffix}.json").read().strip().splitlines()[-1])
                e=d.get("engine",{})
                print(mode, "merge", M, suffix, "%.3f s, %d ev, %d launches, %.2f TB, dev %.6e, merged %s/%s, interp %s, kernel_ms %s GBps %s" % (d["seconds"], d["evaluations"], d["kernel_launches"], d["algorithmic_bytes"]/1e12, d.get("mean_deviation", float("nan")), e.get("merged_launches"), e.get("merged_chains"), e.get("interpreter_launches"), d.get("kernel_ms_total"), d.get("achieved_GBps")))
            except Exception as ex: print(mode, M, suffix, "failed", ex)
PY
