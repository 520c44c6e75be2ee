set -o pipefail
mkdir -p gpurun_out/r5m
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
S=finmath-lib-cuda-extensions_amd/bin/lmm_smile_hip
timeout -k 10 600 python -m pytest tests/test_gpu_merged_chains.py tests/test_gpu_lmm.py tests/test_gpu_lmm_smile.py tests/test_gpu_replicas.py -x -q > gpurun_out/r5m/test_small.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r5m/test_small.txt; tail -5 gpurun_out/r5m/test_small.txt
J='import json,sys
for l in sys.stdin.read().strip().splitlines():
    if not l.startswith("{"): continue
    d=json.loads(l); e=d.get("engine",{}); print("  %.3f s, %d evaluations, %d launches, %.2f TB algorithmic, mean dev %.6e, merged %s/%s, interp %s, kernel_ms %s GBps %s" % (d["seconds"], d["evaluations"], d["kernel_launches"], d["algorithmic_bytes"]/1e12, d.get("mean_deviation", float("nan")), e.get("merged_launches"), e.get("merged_chains"), e.get("interpreter_launches"), d.get("kernel_ms_total"), d.get("achieved_GBps")))'
{
for rep in 1 2; do
for SM in 1 0; do
  echo "== native, merge small $SM"; FMHIP_MERGE_SMALL=$SM timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 | python3 -c "$J"
  echo "== hint-free, merge small $SM"; FMHIP_MERGE_SMALL=$SM timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --finmath-like | python3 -c "$J"
done
done
echo "== native profiled"; FMHIP_PROFILE_DUMP=1 timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --profile 2> gpurun_out/r5m/native_small_prof.txt | python3 -c "$J"
for SP in 1 0; do for rep in 1 2; do
  echo "== smile speculate $SP"; FMHIP_SMILE_SPECULATE=$SP timeout -k 10 120 $S --paths 163840 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in d.items() if k in ('seconds','iterations','accepted_points','evaluations','speculative_evaluations_discarded','rms_deviation','mean_deviation','kernel_launches')})"
done; done
} > gpurun_out/r5m/small_and_smile.txt 2>&1
cat gpurun_out/r5m/small_and_smile.txt
