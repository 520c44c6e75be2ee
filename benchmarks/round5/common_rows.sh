set -o pipefail
mkdir -p gpurun_out/r5c
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
timeout -k 10 600 python -m pytest tests/test_gpu_merged_chains.py tests/test_gpu_lmm.py tests/test_gpu_replicas.py tests/test_gpu_fusion.py -x -q > gpurun_out/r5c/tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r5c/tests.txt; tail -6 gpurun_out/r5c/tests.txt
J='import json,sys
for l in sys.stdin.read().strip().splitlines():
    if not l.startswith("{"): continue
    d=json.loads(l); e=d.get("engine",{}); print("  %.3f s, %d evaluations, %d launches, %.2f TB algorithmic, %.3e path-ops, mean dev %.6e, merged %s/%s, common rows %s, interp %s, kernel_ms %s" % (d["seconds"], d["evaluations"], d["kernel_launches"], d["algorithmic_bytes"]/1e12, d["path_ops"], d.get("mean_deviation", float("nan")), e.get("merged_launches"), e.get("merged_chains"), e.get("common_rows"), e.get("interpreter_launches"), d.get("kernel_ms_total")))'
{
for C in 1 0 1 0; do echo "== native, common rows $C"; FMHIP_COMMON_ROWS=$C timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 | python3 -c "$J"; done
for C in 1 0; do echo "== native profiled, common rows $C"; FMHIP_COMMON_ROWS=$C timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --profile | python3 -c "$J"; done
for rep in 1 2; do echo "== hint-free (window 2000, new pack)"; timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --finmath-like | python3 -c "$J"; done
echo "== full convergence, common rows 1"; timeout -k 10 120 $L --paths 1000000 --mode calibrate | python3 -c "$J"
} > gpurun_out/r5c/common_rows.txt 2>&1
cat gpurun_out/r5c/common_rows.txt
