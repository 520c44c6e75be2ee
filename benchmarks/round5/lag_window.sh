mkdir -p gpurun_out/r5l
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
{
for rep in 1 2; do for W in 2000 5000; do for LAG in 0 20 100; do
  extra=""; [ $LAG != 0 ] && extra="--release-lag $LAG"
  FMHIP_SPECULATE_PENDING=$W timeout -k 10 120 $L --paths 1000000 --mode calibrate --max-iterations 12 --finmath-like $extra | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('engine',{}); print('window $W lag $LAG: %.3f s, %d launches, interp %s, late while waiting %s at once %s, %.3f s' % (d['seconds'], d['kernel_launches'], e.get('interpreter_launches'), e.get('late_releases_while_waiting'), e.get('late_releases_at_once'), e.get('late_release_seconds')))"
done; done; done
} > gpurun_out/r5l/lag_window.txt 2>&1
cat gpurun_out/r5l/lag_window.txt
