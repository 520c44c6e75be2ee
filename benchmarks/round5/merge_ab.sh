set -o pipefail
mkdir -p gpurun_out/r5m
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
J='import json,sys
for l in sys.stdin.read().strip().splitlines():
    if not l.startswith("{"): continue
    d=json.loads(l); e=d.get("engine",{}); print("  %.3f s, %d evaluations, %d launches, %.2f TB algorithmic, mean dev %.6e, merged launches %s chains %s, interp %s, kernel_ms %s GBps %s" % (d["seconds"], d["evaluations"], d["kernel_launches"], d["algorithmic_bytes"]/1e12, d.get("mean_deviation", float("nan")), e.get("merged_launches"), e.get("merged_chains"), e.get("interpreter_launches"), d.get("kernel_ms_total"), d.get("achieved_GBps")))'
timeout -k 10 400 python -m pytest tests/test_gpu_merged_chains.py -x -q > gpurun_out/r5m/test_merged.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r5m/test_merged.txt; tail -15 gpurun_out/r5m/test_merged.txt
{
for M in 1 0; do
  echo "== native calibration, merge $M"; FMHIP_MERGE_CHAINS=$M timeout -k 10 120 $L --paths 1000000 --mode calibrate | python3 -c "$J"
  echo "== native calibration profiled, merge $M"; FMHIP_MERGE_CHAINS=$M timeout -k 10 120 $L --paths 1000000 --mode calibrate --profile | python3 -c "$J"
  echo "== hint-free calibration, merge $M"; FMHIP_MERGE_CHAINS=$M timeout -k 10 120 $L --paths 1000000 --mode calibrate --finmath-like | python3 -c "$J"
  echo "== hint-free calibration profiled, merge $M"; FMHIP_MERGE_CHAINS=$M timeout -k 10 120 $L --paths 1000000 --mode calibrate --finmath-like --profile | python3 -c "$J"
done
} > gpurun_out/r5m/lmm_merge_ab.txt 2>&1
cat gpurun_out/r5m/lmm_merge_ab.txt
