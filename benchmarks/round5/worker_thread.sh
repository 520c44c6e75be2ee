set -o pipefail
mkdir -p gpurun_out/r5m
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
J='import json,sys
for l in sys.stdin.read().strip().splitlines():
    if not l.startswith("{"): continue
    d=json.loads(l); e=d.get("engine",{}); print("  %.3f s, %d evaluations, %d launches, %.2f TB algorithmic, mean dev %.6e, merged launches %s" % (d["seconds"], d["evaluations"], d["kernel_launches"], d["algorithmic_bytes"]/1e12, d.get("mean_deviation", float("nan")), e.get("merged_launches")))'
{
for rep in 1 2; do
echo "== hint-free, caller thread drives the engine"; timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --finmath-like | python3 -c "$J"
echo "== hint-free, FMHIP_WORKER_THREAD=1 --devices 0"; FMHIP_WORKER_THREAD=1 timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --finmath-like --devices 0 | python3 -c "$J"
done
echo "== native, FMHIP_WORKER_THREAD=1 --devices 0"; FMHIP_WORKER_THREAD=1 timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --devices 0 | python3 -c "$J"
echo "== native"; timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 | python3 -c "$J"
} > gpurun_out/r5m/worker_thread.txt 2>&1
cat gpurun_out/r5m/worker_thread.txt
