mkdir -p gpurun_out/r5d
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
J='import json,sys
for l in sys.stdin.read().strip().splitlines():
    if not l.startswith("{"): continue
    d=json.loads(l); e=d.get("engine",{}); print("  %.3f s, %d launches, %.2f TB, dev %.6e, merged %s/%s, interp %s, kernels %s disk %s" % (d["seconds"], d["kernel_launches"], d["algorithmic_bytes"]/1e12, d.get("mean_deviation", float("nan")), e.get("merged_launches"), e.get("merged_chains"), e.get("interpreter_launches"), d.get("specialised_kernels"), d.get("specialisations_from_disk_cache")))'
{
for M in 1 0 1 0; do echo "== devices 0,0 merge $M"; FMHIP_MERGE_CHAINS=$M timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --devices 0,0 | python3 -c "$J"; done
echo "== devices 0,0 merge 1 small 0"; FMHIP_MERGE_SMALL=0 timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --devices 0,0 | python3 -c "$J"
echo "== single, 500k paths"; timeout -k 10 120 $L --paths 500000 --mode calibrate --max-iterations 12 | python3 -c "$J"
} > gpurun_out/r5d/devlist.txt 2>&1
cat gpurun_out/r5d/devlist.txt
