set -o pipefail
mkdir -p gpurun_out/r5m
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5m/gpu_suite.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r5m/gpu_suite.txt; tail -5 gpurun_out/r5m/gpu_suite.txt
S=finmath-lib-cuda-extensions_amd/bin/lmm_smile_hip
for M in 1 0; do
  for rep in 1 2; do
    FMHIP_MERGE_CHAINS=$M timeout -k 10 120 $S --paths 163840 > gpurun_out/r5m/smile_m${M}_${rep}.json 2> gpurun_out/r5m/smile_m${M}_${rep}.err
  done
done
python3 - <<'PY'
import json
for M in (1,0):
    for rep in (1,2):
        try:
            d=json.loads(open(f"gpurun_out/r5m/smile_m{M}_{rep}.json").read().strip().splitlines()[-1])
            print("smile merge", M, rep, {k:v for k,v in d.items() if not isinstance(v,(list,dict))})
        except Exception as ex: print("smile", M, rep, "failed", ex)
PY
