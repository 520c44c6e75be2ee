set -o pipefail
OUT=gpurun_out/${1:-r5g}
mkdir -p $OUT
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
J='import json,sys
for l in sys.stdin.read().strip().splitlines():
    if not l.startswith("{"): continue
    d=json.loads(l); e=d.get("engine",{}); print("  %.3f s, %d evaluations, %d launches, %.2f TB algorithmic, mean dev %.6e, merged %s/%s, interp %s, kernels %s from disk %s" % (d["seconds"], d["evaluations"], d["kernel_launches"], d["algorithmic_bytes"]/1e12, d.get("mean_deviation", float("nan")), e.get("merged_launches"), e.get("merged_chains"), e.get("interpreter_launches"), d.get("specialised_kernels"), d.get("specialisations_from_disk_cache")))'
{
for S in 4 5 5 4 5; do echo "== native, $S steps per group"; timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --steps-per-launch $S | python3 -c "$J"; done
for S in 4 5 5 4 5; do echo "== hint-free, engine groups $S steps"; FMHIP_GROUP_STEPS=$S timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --finmath-like | python3 -c "$J"; done
echo "== hint-free host profile"; FMHIP_HOST_PROFILE=1 timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --finmath-like 2>&1 >/dev/null | tail -20
} > $OUT/steps_and_host.txt 2>&1
cat $OUT/steps_and_host.txt
