# hint-free caller with merged families: speculation window, every value twice (the second run finds the first one's kernels in the user's cache)
mkdir -p gpurun_out/r5s
B=finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12 --finmath-like"
{
for v in 5000 1500 1500 2000 2000 2500 2500 3000 3000 4000 4000 5000; do
FMHIP_SPECULATE_PENDING=$v timeout -k 10 120 $B $A | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('engine',{}); print('speculate after $v methods: %.3f s, %d launches, merged %s/%s, interp %s' % (d['seconds'], d['kernel_launches'], e.get('merged_launches'), e.get('merged_chains'), e.get('interpreter_launches')))"
done
} > gpurun_out/r5s/sweep.txt 2>&1
cat gpurun_out/r5s/sweep.txt
